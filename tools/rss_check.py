"""Peak resident memory of a pipelined blastn search loop as the number of passes grows (the queue of host replays
must not keep finished tasks reachable).  usage: rss_check.py passes"""
import os, resource, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gblastn_amd import api, synth

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nsub = 1000
api.lib().gbn_init(1, 0)
lay = synth.SynthDb(nsub, 1_000_000, seed=4242)
slab = torch.empty(lay.nbytes, dtype=torch.uint8, device="cuda")
api._check(api.lib().gbn_synth_fill(slab.data_ptr(), lay.nbytes, lay.seed, None))
src = api.BlastSeqSrc.from_slab((slab.data_ptr(), lay.nbytes), lay.byte_off, lay.lens, is_device=True, keep=slab)
qs, _ = synth.make_queries(100, lay)
ps = api.BlastPrelimSearch(qs, api.default_options("blastn", db_length=nsub * 10**6, db_num_seqs=nsub), src)
n0 = len(ps.run()["hsps"])
for k in range(passes):
    ps.begin()
    assert len(ps.end()["hsps"]) == n0
    if k in (0, passes // 2, passes - 1):
        print("pass %d: max RSS %.0f MB" % (k, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0), flush=True)
