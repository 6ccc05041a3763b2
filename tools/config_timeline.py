#!/usr/bin/env python
"""The cold start of the whole C2 config (bench.py's config_measured) once under the tracer: no records, nothing set up ahead;
gbn_db_prepare_records, two query batches set up from scratch, scanned, extended.  Run under rocprofv3 --kernel-trace and read
with tools/timeline.py (the last ~120 dispatches).  usage: config_timeline.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from concurrent.futures import ThreadPoolExecutor
from gblastn_amd import api, synth

api.lib().gbn_init(1, 0)
nsub, slen = 50000, 1_000_000
lay = synth.SynthDb(nsub, slen, seed=0x9E3779B97F4A7C15 ^ 1)
slab = torch.empty(lay.nbytes, dtype=torch.uint8, device="cuda")
api._check(api.lib().gbn_synth_fill(slab.data_ptr(), lay.nbytes, lay.seed, None))
src = api.BlastSeqSrc.from_slab((slab.data_ptr(), lay.nbytes), lay.byte_off, lay.lens, is_device=True, keep=slab)
queries, _ = synth.make_queries(10000, lay)
opt = api.default_options("megablast", db_length=nsub * slen, db_num_seqs=nsub)
qsets = [api.QuerySet(queries[k * 5000:(k + 1) * 5000]) for k in range(2)]
pool = ThreadPoolExecutor(max_workers=2, initializer=lambda: torch.cuda.set_device(0))
make = lambda k: api.BlastPrelimSearch(qsets[k], opt, src)


def config():
    src.prepare_records(opt, qsets[0])
    f = [pool.submit(make, 0), pool.submit(make, 1)]
    a = f[0].result(); a.begin()
    b = f[1].result(); b.begin()
    ra = a.end(); rb = b.end()
    a.close(); b.close()
    return len(ra["hsps"]) + len(rb["hsps"])


for r in range(3):
    api.record_cache_set_limit(-1); api.record_cache_invalidate()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = config()
    torch.cuda.synchronize()
    print("config %d: %.2f ms, %d HSPs" % (r, (time.perf_counter() - t0) * 1e3, n), file=sys.stderr)
    time.sleep(0.05)
