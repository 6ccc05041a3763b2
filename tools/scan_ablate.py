#!/usr/bin/env python
"""Time the scan stage only (gbn_scan_only) for a synthetic shard; GBN_DBG / GBN_SCAN_BINS
environment switches select ablations, TASK=blastn the blastn options.  usage: scan_ablate.py [subjects] [queries]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gblastn_amd import api, synth

nsub = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
api.lib().gbn_init(1, 0)
lay = synth.SynthDb(nsub, 1_000_000, seed=12345)
slab = torch.empty(lay.nbytes, dtype=torch.uint8, device="cuda")
api._check(api.lib().gbn_synth_fill(slab.data_ptr(), lay.nbytes, lay.seed, None))
src = api.BlastSeqSrc.from_slab((slab.data_ptr(), lay.nbytes), lay.byte_off, lay.lens, is_device=True, keep=slab)
qs, _ = synth.make_queries(nq, None)
task = os.environ.get("TASK", "megablast")
ps = api.BlastPrelimSearch(qs, api.default_options(task, db_length=nsub * 10**6, db_num_seqs=nsub), src)
print(ps.info())
ps.scan_only(repeats=1)
d = ps.scan_only(repeats=3)
n = d.scan_launches
print("GBN_DBG=%s bins=%s: scan %.2f ms/launch (bin %.2f probe %.2f rare %.2f), seeds %d, lookup_hits %d" % (
    os.environ.get("GBN_DBG"), os.environ.get("GBN_SCAN_BINS"), d.scan_kernel_ms / n, d.bin_kernel_ms / n,
    d.probe_kernel_ms / n, d.rare_kernel_ms / n, d.seeds, d.lookup_hits))
