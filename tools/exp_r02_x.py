import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gblastn_amd import api, synth
nsub = 1000
api.lib().gbn_init(1, 0)
lay = synth.SynthDb(nsub, 1_000_000, seed=12345)
slab = torch.empty(lay.nbytes, dtype=torch.uint8, device="cuda")
api._check(api.lib().gbn_synth_fill(slab.data_ptr(), lay.nbytes, lay.seed, None))
src = api.BlastSeqSrc.from_slab((slab.data_ptr(), lay.nbytes), lay.byte_off, lay.lens, is_device=True, keep=slab)
qs, _ = synth.make_queries(100, lay)
ps = api.BlastPrelimSearch(qs, api.default_options("blastn", db_length=nsub * 10**6, db_num_seqs=nsub), src)
outs = []
for k in range(3):
    r = ps.run(keep_stages=True)
    ih = r["init_hits"]
    outs.append(ih)
    print(k, len(r["seeds"]), len(ih), len(r["hsps"]))
def key(a): return set(map(tuple, a[["oid", "q_off", "s_off", "q_start", "s_start", "length", "score"]].tolist()))
A, B = key(outs[0]), key(outs[1])
print("only in 0:", len(A - B), "only in 1:", len(B - A))
d = sorted(A - B)[:10]; print(d)
d2 = sorted(B - A)[:10]; print(d2)
import collections
print("scores of diff:", collections.Counter(x[6] for x in (A ^ B)).most_common(8))
print("lengths of diff:", collections.Counter(x[5] for x in (A ^ B)).most_common(8))
