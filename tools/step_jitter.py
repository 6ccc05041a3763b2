#!/usr/bin/env python
"""Where a pipelined pass loses time now and then (and a soak test of the pipelined loop: every pass of a query set must return the same HSPs): the C2 loop of bench.py (two set-ups in flight, begin() on the main thread, end() +
close() on a worker) with the wall clock of every step taken apart -- waiting for the batch's set-up, begin() -- and the outliers
printed.  usage: step_jitter.py [steps] [cache: 0|1]"""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from concurrent.futures import ThreadPoolExecutor
from gblastn_amd import api, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cache = int(sys.argv[2]) if len(sys.argv) > 2 else 1
api.lib().gbn_init(1, 0)
nsub, slen = 50000, 1_000_000
lay = synth.SynthDb(nsub, slen, seed=0x9E3779B97F4A7C15 ^ 1)
slab = torch.empty(lay.nbytes, dtype=torch.uint8, device="cuda")
api._check(api.lib().gbn_synth_fill(slab.data_ptr(), lay.nbytes, lay.seed, None))
src = api.BlastSeqSrc.from_slab((slab.data_ptr(), lay.nbytes), lay.byte_off, lay.lens, is_device=True, keep=slab)
queries, _ = synth.make_queries(10000, lay)
opt = api.default_options("megablast", db_length=nsub * slen, db_num_seqs=nsub)
qsets = [api.QuerySet(queries[k * 5000:(k + 1) * 5000]) for k in range(2)]
pin = lambda: torch.cuda.set_device(0)
setup_pool = ThreadPoolExecutor(max_workers=2, initializer=pin)
closer = ThreadPoolExecutor(max_workers=1, initializer=pin)
api.record_cache_set_limit(-1 if cache else 0)
t_make = {}


def make(k):
    t0 = time.perf_counter()
    b = api.BlastPrelimSearch(qsets[k % 2], opt, src)
    t_make[k] = (time.perf_counter() - t0) * 1e3
    return b


def finish(b, k):
    t0 = time.perf_counter()
    h = b.end()["hsps"]
    t1 = time.perf_counter()
    b.close()
    return (k & 1, len(h), hash(h.tobytes())), (t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3


def run(count):
    rows, futs = [], []
    ahead, queued, prev = [setup_pool.submit(make, 0)], 1, None
    for k in range(count):
        t0 = time.perf_counter()
        b = ahead.pop(0).result()
        t1 = time.perf_counter()
        while queued < count and queued <= k + 2:
            ahead.append(setup_pool.submit(make, queued)); queued += 1
        t2 = time.perf_counter()
        b.begin()
        t3 = time.perf_counter()
        if prev is not None:
            futs.append(closer.submit(finish, prev, k - 1))
        prev = b
        rows.append((k, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (time.perf_counter() - t0) * 1e3, b.diagnostics.total_ms, b.diagnostics.scan_stage_ms))
    futs.append(closer.submit(finish, prev, count - 1))
    fin = [f.result() for f in futs]
    return rows, fin


run(6)
torch.cuda.synchronize()
t0 = time.perf_counter()
rows, fin = run(steps)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) * 1e3
tot = sorted(r[4] for r in rows)
med = tot[len(tot) // 2]
seen = {}
for sig, _, _ in fin:                       # the two query sets alternate: every pass of one gives the same HSPs, byte for byte
    seen.setdefault(sig[0], set()).add(sig[1:])
assert all(len(v) == 1 for v in seen.values()), "HSPs of one query set differ between passes: %r" % {k: sorted(v)[:4] for k, v in seen.items()}
print("HSPs per pass (even / odd batches): %s -- identical in every pass" % [sorted(v)[0][0] for _, v in sorted(seen.items())])
print("cache %d: %d steps, %.2f ms per step wall, median step %.2f, p90 %.2f, max %.2f" % (cache, steps, wall / steps, med, tot[int(len(tot) * 0.9)], tot[-1]))
print("mean: wait for set-up %.3f, submit %.3f, begin %.3f; set-up (worker) mean %.2f max %.2f; end mean %.2f max %.2f; close mean %.2f max %.2f" % (
    sum(r[1] for r in rows) / steps, sum(r[2] for r in rows) / steps, sum(r[3] for r in rows) / steps,
    sum(t_make.values()) / len(t_make), max(t_make.values()),
    sum(f[1] for f in fin) / len(fin), max(f[1] for f in fin), sum(f[2] for f in fin) / len(fin), max(f[2] for f in fin)))
print("outliers (step > 1.3 x median): k, wait for set-up, submit, begin, total | that batch's set-up | inside the library: search, its scan stage")
for r in rows:
    if r[4] > 1.3 * med:
        print("  %4d  %7.2f %6.2f %7.2f %7.2f | %.2f | %.2f %.2f" % (r[0], r[1], r[2], r[3], r[4], t_make.get(r[0], 0), r[5], r[6]))
