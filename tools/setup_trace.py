import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from gblastn_amd import api, synth
api.lib().gbn_init(1, 0)
lay = synth.SynthDb(100, 1_000_000, seed=5)
slab = torch.empty(lay.nbytes, dtype=torch.uint8, device="cuda")
api._check(api.lib().gbn_synth_fill(slab.data_ptr(), lay.nbytes, lay.seed, None))
src = api.BlastSeqSrc.from_slab((slab.data_ptr(), lay.nbytes), lay.byte_off, lay.lens, is_device=True, keep=slab)
queries, _ = synth.make_queries(5000, lay)
opt = api.default_options("megablast", db_length=50_000_000_000, db_num_seqs=50000)
Q = api.QuerySet(queries)
for r in range(4):
    t = time.perf_counter(); ps = api.BlastPrelimSearch(Q, opt, src); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("set-up returns after %.2f ms, device done after %.2f ms" % ((t1 - t) * 1e3, (t2 - t) * 1e3), file=sys.stderr)
    ps.close()
