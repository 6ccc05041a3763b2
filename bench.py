#!/usr/bin/env python
"""bench.py -- subject Gbp scanned per second of the megablast preliminary search.

Workload (BASELINE.json configs[1], "C2"): 10,000 x 1 kb synthetic queries vs a
50 Gbp synthetic 2-bit database (50,000 subjects x 1 Mb, 12.5 GB packed) on ONE
MI355X, megablast word_size 28.  The reference batch plan applies: 5 Mb query
batches (5,000 queries, 10 M lookup words -> megablast table lut 12, stride 17),
so the config is 2 passes over the database.

A "step" = one pass of ONE query batch over the rank's resident shard, from the caller's query arrays to
merged results: set-up of the batch (concatenation, Karlin-Altschul parameters, cut-offs on the host; lookup
structures built on the device), the whole preliminary path (scan + seed, diagonal filter, ungapped X-drop,
greedy gapped, HSP rules) and the gather + top-N merge.  No result of one step is reused by another -- the headline
runs with the library's record cache switched OFF (gbn_record_cache_set_limit(0)), so every step bins the whole shard
(the north_star scan); set-up, extension stages and merge run on worker threads / a second stream underneath the
neighbouring steps' scans, and the pipeline stays primed from one timed region to the next (each region sets up the
first two query batches of the next one and finds its own first two set up; the binning kernel of a region's first
pass may have been queued by the pass before it -- a region of K passes still holds K of everything: steady state,
not a cold start; `config.config_wall_ms_measured` is the cold start of the whole config, with the library's default
policy).  Only the database shard is resident in HBM before the timed region.  value = (bases of all shards x K passes) /
max-over-ranks wall time.  `config.engine_only` gives, beside it, the engine entry point alone on reused
query batches (its lookup tables are inputs of that entry point).  With N > 1 every rank holds its own
50 Gbp shard (weak scaling, the C5 layout: volumes sharded by rank, global statistics) and rank 0 gathers
the per-shard HSP records over RCCL after every pass, inside the timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=160, help="timed passes (default: > 2 s of timed region on C2)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["C2", "C3", "C4", "shim", "cli"], default="C2",
                    help="C2 (the metric's config): megablast W=28 vs 50 Gbp; C3: blastn W=11 vs 5 Gbp, 100 kb batches; "
                         "C4: 100k queries streamed in 5 Mb batches through the host pipeline, CPU traceback overlapped with the GPU stages; "
                         "shim: the C2 shard as 100 resident blocks searched the way gblastn_amd/shim/gpu_blastn_amd_shim.cpp searches them; "
                         "cli: the documented invocation end to end -- blastn_prelim on the C2 database written as BLAST v4 volumes on disk (bench_cli.py)")
    ap.add_argument("--trace-threads", type=int, default=0, help="C4: traceback consumer threads (0: a quarter of the host cores, 4 .. 16; round 4 ran 4, "
                                                                  "with which the traceback of a batch, not the GPU, sets the pace once the records are cached)")
    ap.add_argument("--no-traceback", action="store_true", help="C4 diagnostics: the pipeline without its traceback stage")
    ap.add_argument("--subjects", type=int, default=None, help="subjects per GPU shard")
    ap.add_argument("--subject-len", type=int, default=1_000_000)
    ap.add_argument("--queries", type=int, default=10_000)
    ap.add_argument("--batch-queries", type=int, default=None)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-cores", type=int, default=0, help="processes of the CPU baseline (0: half of the host cores, at most 64)")
    ap.add_argument("--no-overlap", action="store_true", help="run every pass to completion before the next starts")
    ap.add_argument("--engine-steps", type=int, default=8, help="passes of the side measurement on reused query batches (engine entry point alone); 0: skip")
    ap.add_argument("--min-seconds", type=float, default=2.0,
                    help="the timed region of --steps passes is repeated until this much time has been measured; the line reports the "
                         "median region (ms_per_step, value) and the spread (ms_per_step_minmax, regions)")
    ap.add_argument("--no-side-workloads", action="store_true",
                    help="skip the short C3 / C4 side measurements (config.other_workloads) of the default C2 run")
    ap.add_argument("--side", action="store_true", help=argparse.SUPPRESS)    # this process IS a side measurement
    ap.add_argument("--record-cache", choices=["default", "on", "off"], default="default",
                    help="the library's record cache (bin once, probe many).  default: OFF for the C2 / C3 headline (every pass bins: "
                         "the north_star scan), ON -- the library's own default -- for C4 and the shim workload")
    ap.add_argument("--skew", action="store_true",
                    help="repeats over the synthetic database (gbn_synth_skew: 8 %% of every subject as homopolymer runs / tandem repeats, a 1,200-base family "
                         "element in one subject of fifty) and 2 %% of the queries carrying a piece of the family element; reports rescans, direct-kernel "
                         "ranges and library sorts beside the step")
    ap.add_argument("--strong", action="store_true",
                    help="N > 1: strong scaling -- the --subjects of ONE shard are divided among the ranks (fixed total work) instead of every rank holding --subjects (weak, the default)")
    a = ap.parse_args()
    if a.workload == "C3" and "--steps" not in " ".join(sys.argv):
        a.steps = 16
    if a.workload == "C4" and "--steps" not in " ".join(sys.argv):
        a.steps = 20                                    # 100,000 queries = 20 batches of 5,000
    if a.workload == "shim" and "--steps" not in " ".join(sys.argv):
        a.steps = 10
    if a.workload == "cli" and "--steps" not in " ".join(sys.argv):
        a.steps = 2
    if a.trace_threads <= 0:
        a.trace_threads = max(4, min(16, (os.cpu_count() or 16) // 4))
    if a.subjects is None:
        a.subjects = 5_000 if a.workload == "C3" else 50_000
    if a.batch_queries is None:
        a.batch_queries = 100 if a.workload == "C3" else 5_000     # 5 Mb megablast / 100 kb blastn batches
    return a


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from gblastn_amd import api, synth, shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    ndev = max(torch.cuda.device_count(), 1)
    dev = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(dev)
    shared_device = world > ndev
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared_device:
            # more ranks than devices (the builder's one-GPU box): NOT a scaling measurement, only a way to run the
            # RCCL exchange for real -- RCCL refuses two ranks of one host on one device, so each rank presents
            # itself as its own host and RCCL connects them through its socket transport (see gblastn_amd/blastn_sharded.py)
            os.environ.setdefault("NCCL_HOSTID", "gbn-rank-%d" % rank)
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
            os.environ.setdefault("NCCL_IB_DISABLE", "1")
        dist.init_process_group("nccl", device_id=dev)
    rc = api.lib().gbn_init(1, dev.index)
    if rc:
        raise SystemExit("gbn_init failed: %s" % api.lib().gbn_last_error().decode())
    os.environ.pop("GBN_RECORD_CACHE_MB", None)             # (the policy of this run is set through the API below)
    if args.workload == "cli":
        from bench_cli import bench_cli
        return bench_cli(args, api)
    cache_on = args.record_cache == "on" or (args.record_cache == "default" and args.workload in ("C4", "shim"))
    api.record_cache_set_limit(-1 if cache_on else 0)
    if args.strong and world > 1:
        args.subjects = max(1, args.subjects // world)      # the shard of a rank under strong scaling

    # ---- database shard of this rank, generated in HBM ----
    nsub, slen = args.subjects, args.subject_len
    layouts = [synth.SynthDb(nsub, slen, seed=0x9E3779B97F4A7C15 ^ (r + 1), first_oid=r * nsub, skew=args.skew)
               for r in range(world)]
    mine = layouts[rank]
    slab = torch.empty(mine.nbytes, dtype=torch.uint8, device=dev)
    api._check(api.lib().gbn_synth_fill(slab.data_ptr(), mine.nbytes, mine.seed, None))
    if args.skew:
        mine.skew_on_device(api, slab.data_ptr())
    src = api.BlastSeqSrc.from_slab((slab.data_ptr(), mine.nbytes), mine.byte_off, mine.lens,
                                    first_oid=mine.first_oid, is_device=True, keep=slab)
    total_bases_global = world * nsub * slen

    # ---- queries: replicated; planted homologs come from any shard ----
    class AnyShard:
        num, length, first_oid = world * nsub, slen, 0
        _cache = {}

        def subject_bases(self, g):
            if g not in self._cache:
                if len(self._cache) > 64:
                    self._cache.clear()
                self._cache[g] = layouts[g // nsub].subject_bases(g % nsub)
            return self._cache[g]
    AnyShard.seed = layouts[0].seed
    queries, plants = synth.make_queries(args.queries, AnyShard(), family_fraction=0.02 if args.skew else 0.0)
    task = "blastn" if args.workload == "C3" else "megablast"
    opt = api.default_options(task, db_length=total_bases_global, db_num_seqs=world * nsub)
    nbatch = (len(queries) + args.batch_queries - 1) // args.batch_queries
    npass_config = nbatch
    nbatch = min(nbatch, max(args.steps, args.warmup, 1))       # only the batches the run touches
    # query batches as the caller would hand them over: one contiguous BLASTNA array per query
    qsets = [api.QuerySet(queries[i * args.batch_queries:(i + 1) * args.batch_queries]) for i in range(nbatch)]

    if args.workload == "shim":
        return bench_shim(args, api, torch, dev, slab, mine, src, qsets, nbatch, opt, nsub, slen)
    if args.workload == "C4":
        return bench_c4(args, api, torch, dist, world, rank, dev, src, qsets, nbatch, opt, total_bases_global, nsub, slen, queries)

    # --skew: the queries are DUST-filtered as blastn filters them by default (a planted slice of a poly-A stretch would otherwise seed at
    # every scan position of every poly-A stretch of the database: 5e10 seeds); the masks are the caller's input, made once
    qmasks = [api.dust_masks(queries[i * args.batch_queries:(i + 1) * args.batch_queries]) if args.skew else None for i in range(nbatch)]

    def make(k):
        """set-up of the query batch of pass k from scratch: concatenation, Karlin-Altschul parameters,
        cut-offs on the host; lookup structures built on the device"""
        return api.BlastPrelimSearch(qsets[k % nbatch], opt, src, masks=qmasks[k % nbatch])

    def merge(nq, out):
        # exchange + merge step: gather to rank 0, replay through the per-query top-N collector
        got = shard.collect_on_root(out["hsps"], nq, opt.hitlist_size, dst=0, device=dev)
        return 0 if got is None else len(got[0])

    from concurrent.futures import ThreadPoolExecutor
    pin = lambda: torch.cuda.set_device(dev)
    merger = shard.Exchange(dev)          # the rank's one ordered channel for collectives: worker thread + its own stream
    setup_pool = ThreadPoolExecutor(max_workers=2, initializer=pin)     # two set-ups in flight
    primed, keep_primed = [], [False]     # set-ups started by one run_passes call for the next one (futures); on between timed regions

    def run_passes(count, diags):
        """`count` passes.  Nothing is carried over between passes: every pass sets its query batch up
        from scratch (worker threads, underneath the passes before it), scans the whole shard, extends,
        and is merged on rank 0 (worker thread: host work, or an RCCL gather at N > 1).  The extension
        stages and the merge of pass k overlap the scan of pass k + 1.  With --no-overlap every step runs
        to completion before the next starts, set-up included."""
        if count <= 0:
            return 0
        n = 0
        if args.no_overlap:
            for k in range(count):
                b = make(k)
                n += merge(len(b._q), b.run()); diags.append(b.diagnostics); b.close()
            return n
        def finish(b):                      # worker thread: wait for the batch's extension stages (they were queued before the
            got = merge(len(b._q), b.end())  # next batch's begin() returned: that begin() is over), merge, release the batch
            b.close()
            return got
        prev, futs = None, []
        # Two set-ups are in flight at any time.  The pipeline stays primed from one call to the next: the set-ups of the
        # first two passes of the NEXT region are started (and waited for) inside this one -- a region of K passes still
        # holds K set-ups, K scans, K extension stages and K merges, but a 20-pass region no longer begins with a set-up
        # that has nothing to hide behind (the very first call pays it, outside the timed regions: the warm-up).
        ahead = primed[:]; del primed[:]
        queued = len(ahead)
        if not ahead:
            ahead, queued = [setup_pool.submit(make, 0)], 1
        total = count + (2 if keep_primed[0] else 0)
        for k in range(count):
            b = ahead.pop(0).result()
            while queued < total and queued <= k + 2:
                ahead.append(setup_pool.submit(make, queued)); queued += 1
            b.begin()                       # waits for prev's extension stages before queueing its own
            if prev is not None:
                futs.append(merger.submit(finish, prev)); diags.append(prev.diagnostics)     # (the main thread goes straight to the next begin())
            prev = b
        futs.append(merger.submit(finish, prev)); diags.append(prev.diagnostics)
        got = n + sum(f.result() for f in futs)
        for f in ahead:                     # the next region's first batches: set up, lookup structures queued on the builder's stream
            f.result()
        primed.extend(ahead)
        return got

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn):
        sync()
        t0 = time.perf_counter()
        r = fn()
        sync()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return r, el

    # what this box's HBM does on a plain device-to-device copy (boxes differ by 10 % and more: recorded next to
    # the roofline so that box variance can be told from kernel changes)
    def copy_bandwidth():
        n = 1 << 30
        a = torch.empty(n, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
        b.copy_(a); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            b.copy_(a)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        del a, b
        return 8 * 2 * n / (ms * 1e-3) / 1e9
    box_copy = copy_bandwidth()

    # allocator warm-up, whatever --warmup says: three batches exist at a time in the pipeline below, and the
    # library keeps freed device blocks in a pool
    for b in [make(k) for k in range(3)]:
        b.close()
    t_setup = time.perf_counter()
    probe_batch = make(0)
    info = probe_batch.info()                                   # (waits for nothing: host-side fields)
    torch.cuda.synchronize()
    batch_setup_ms = (time.perf_counter() - t_setup) * 1e3      # one set-up alone, lookup structures complete
    probe_batch.close()
    keep_primed[0] = not args.no_overlap
    run_passes(max(args.warmup, 1), [])
    # The timed region = exactly --steps passes between barrier + synchronize on both sides.  A short region (the driver's
    # 20 steps = 0.3 s) is mostly the pipeline's ramp -- first set-up alone, last extension stage and merge -- and one
    # box-noise sample: it is repeated until --min-seconds are measured, the MEDIAN region is the line's ms_per_step /
    # value, minimum and maximum are reported beside it.
    regions = []
    while True:
        dg = []
        nh, el = timed(lambda: run_passes(args.steps, dg))
        regions.append((el, nh, dg))
        spent = sum(r[0] for r in regions)
        more = spent < args.min_seconds and len(regions) < 64
        if world > 1:                                           # every rank takes the same decision
            t = torch.tensor([1.0 if more else 0.0], dtype=torch.float64, device=dev)
            dist.broadcast(t, src=0)
            more = bool(t.item() > 0.5)
        if not more:
            break
    keep_primed[0] = False
    for f in primed:                        # what the last region prepared for a region that does not come
        f.result().close()
    del primed[:]
    order = sorted(range(len(regions)), key=lambda i: regions[i][0])
    elapsed, nhsp, diags = regions[order[(len(order) - 1) // 2]]
    region_ms = [r[0] / args.steps * 1e3 for r in regions]

    # ---- beside it: the engine entry point alone, on query batches set up once and reused (their lookup
    # tables are inputs of the entry point).  Not the headline number.
    engine_only = None
    if not args.no_overlap and args.engine_steps > 0 and nbatch >= 2:
        held = [make(k) for k in range(2)]

        def reuse(count):
            prev, futs = None, []
            for k in range(count):
                b = held[k % 2]
                b.begin()
                if prev is not None:
                    futs.append(merger.submit(merge, len(prev._q), prev.end()))
                prev = b
            futs.append(merger.submit(merge, len(prev._q), prev.end()))
            return sum(f.result() for f in futs)
        reuse(2)
        k0 = [(b.diagnostics.bin_kernel_ms, b.diagnostics.probe_kernel_ms, b.diagnostics.rare_kernel_ms, b.diagnostics.scan_launches) for b in held]
        _, el = timed(lambda: reuse(args.engine_steps))
        k1 = [(b.diagnostics.bin_kernel_ms, b.diagnostics.probe_kernel_ms, b.diagnostics.rare_kernel_ms, b.diagnostics.scan_launches) for b in held]
        nl = max(1, sum(a[3] - c[3] for a, c in zip(k1, k0)))
        engine_only = {"ms_per_pass": el / args.engine_steps * 1e3, "value": world * nsub * slen * args.engine_steps / el / 1e9,
                       "unit": "Gbp/s", "passes": args.engine_steps,
                       # (HIP-event time per launch of the three scan kernels with no set-up running beside them)
                       "scan_kernels_ms": [sum(a[i] - c[i] for a, c in zip(k1, k0)) / nl for i in range(3)],
                       "what": "gbn_prelim_search_begin/_end on two query batches set up once and reused alternately"}
        for b in held:
            b.close()

    # ---- the whole config MEASURED, in this process, with the library's default policy (record cache on: bin once, probe many):
    # one region = the config's query batches from the caller's arrays to merged results, nothing cached, set up or primed when
    # it starts (a cold start: the first set-up has nothing to hide behind); then the steady state of later batches over the
    # cached records.  The headline above stays what it was: every pass bins.
    config_measured = cached_pass = None
    if not args.no_overlap and args.workload == "C2" and npass_config >= 2 and nbatch >= npass_config:
        walls, binned_passes = [], []
        for _ in range(7):
            api.record_cache_set_limit(-1); api.record_cache_invalidate()      # every region starts without records (the buffers stay allocated)
            st0 = api.record_cache_stats()

            def whole_config():
                src.prepare_records(opt, qsets[0])              # the caller knows its first batch: the shard's records are binned underneath its set-up
                return run_passes(npass_config, [])
            _, el = timed(whole_config)
            st1 = api.record_cache_stats()
            walls.append(el * 1e3); binned_passes.append(st1["misses"] - st0["misses"])
        walls_sorted = sorted(walls)
        config_measured = {"ms": walls_sorted[(len(walls) - 1) // 2], "minmax": [walls_sorted[0], walls_sorted[-1]], "regions": len(walls),
                           "passes": npass_config, "passes_that_binned_for_themselves": sorted(set(binned_passes)), "record_sets_prepared_per_region": 1,
                           "what": "wall clock of the WHOLE config (%d query batches of %d, set up from scratch inside the region, scanned, extended, merged) as one "
                                   "timed region in this process, library default policy: gbn_db_prepare_records queues the shard's binning kernel when the region starts "
                                   "(it runs underneath the first batch's set-up), the record cache holds the records, both batches run probe + rare kernel only; every "
                                   "region starts with no records and nothing set up ahead" % (npass_config, args.batch_queries)}
        # later batches of a stream over the cached records (what C4 and the shim see per batch)
        keep_primed[0] = True
        run_passes(2, [])
        cr = []
        for _ in range(3):
            dgc = []
            _, el = timed(lambda: run_passes(args.steps, dgc))
            cr.append((el, dgc))
        keep_primed[0] = False
        for f in primed:
            f.result().close()
        del primed[:]
        cr.sort(key=lambda r: r[0])
        el, dgc = cr[1]
        rcs = api.record_cache_stats()
        nl = max(1, sum(d.scan_launches for d in dgc))
        cached_pass = {"ms_per_step": el / args.steps * 1e3, "ms_per_step_minmax": [cr[0][0] / args.steps * 1e3, cr[-1][0] / args.steps * 1e3], "steps": args.steps, "regions": 3,
                       "value": total_bases_global * args.steps / el / 1e9, "unit": "Gbp/s",
                       "scan_kernels_ms": [sum(d.bin_kernel_ms for d in dgc) / nl, sum(d.probe_kernel_ms for d in dgc) / nl, sum(d.rare_kernel_ms for d in dgc) / nl],
                       # the same algorithmic bytes (0.25 B per subject base and pass) over the kernels a cached pass runs
                       "records": {"form": "sorted by cell (runs)" if rcs.get("sorted_sets") else "streams", "resident_bytes": rcs["bytes"], "sorted_bytes": rcs.get("sorted_bytes"),
                                   "sort_gpu_ms": rcs.get("last_sort_us", 0) / 1e3, "sorts": rcs.get("sorts"), "passes_over_sorted_records": rcs.get("sorted_passes")},
                       "roofline": {"bound": "hbm", "kernel": "probe_runs_kernel" if rcs.get("sorted_sets") else "probe_bin_kernel", "peak": 8000.0, "unit": "GB/s",
                                    "achieved": 0.25 * sum(d.subject_bases_scanned for d in dgc) / max(sum(d.probe_kernel_ms for d in dgc), 1e-9) / 1e6,
                                    "frac": 0.25 * sum(d.subject_bases_scanned for d in dgc) / max(sum(d.probe_kernel_ms for d in dgc), 1e-9) / 1e6 / 8000.0,
                                    "scan_stage_frac": 0.25 * sum(d.subject_bases_scanned for d in dgc) / max(sum(d.scan_kernel_ms for d in dgc), 1e-9) / 1e6 / 8000.0,
                                    "traffic": 13.1e9 + 2.9e9, "traffic_what": "probe 13.1 GB + rare 2.9 GB per pass (profiles/scan_traffic.json): 1.28 x the algorithmic 12.5 GB, against 3.8 x for a pass that bins"},
                       "what": "the same step as the headline (set-up from scratch, scan, extension, merge) with the record cache ON and the shard's records "
                               "resident: the binning kernel does not run -- NOT the headline metric (that one bins in every pass)"}
        api.record_cache_set_limit(0)

    # ---- roofline of the dominant kernel (scan+seed), from HIP events in the library ----
    scan_ms = sum(d.scan_kernel_ms for d in diags)
    launches = sum(d.scan_launches for d in diags)
    scanned = sum(d.subject_bases_scanned for d in diags)
    seeds = sum(d.seeds for d in diags)
    lookup_hits = sum(d.lookup_hits for d in diags)
    algo_bytes = 0.25 * scanned
    bin_ms = sum(d.bin_kernel_ms for d in diags)
    probe_ms = sum(d.probe_kernel_ms for d in diags)
    rare_ms = sum(d.rare_kernel_ms for d in diags)
    # dominant kernel: the binning kernel when the partitioned scan is used, else the direct scan
    dom_name, dom_ms = ("scan_bin_kernel", bin_ms) if bin_ms > 0 else (slice_kernel_name(info) if info.get("scan_path") == 2 else "scan_seed_kernel", scan_ms)
    # the binning kernel has stride-specialised variants; this is the name rocprof shows
    dom_label = dom_name + ("_s%d" % info["scan_step"] if bin_ms > 0 and info["scan_step"] in (1, 2, 4, 17, 18, 21) else "")
    achieved = algo_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    stage_achieved = algo_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    traffic = None; traffic_tag = ""
    tf = os.path.join(ROOT, "profiles", "scan_traffic.json")
    # the committed PMC passes were taken on the default workload (C2, full shard); other shapes: null
    per_launch = algo_bytes / max(launches, 1)
    if os.path.exists(tf):
        try:
            tj = json.load(open(tf))
            traffic_tag = str(tj.get("_source", "")).split(":")[0] or "PMC passes"
            if args.workload == "C2" and abs(per_launch - 12.5e9) < 1e6:
                traffic = tj.get(dom_name, {}).get("hbm_bytes_per_launch")
            elif args.workload == "C3" and abs(per_launch - 2.5e8) < 1e6:       # a launch = one subject range of 1,000 x 1 Mb
                traffic = tj.get(dom_label, {}).get("hbm_bytes_per_launch")
                import re
                m = re.search(r"profiles/[A-Za-z0-9_]+\.csv", str(tj.get(dom_label, {}).get("_note", "")))
                if m:
                    traffic_tag = m.group(0)
        except Exception:
            traffic = None

    # GPU time per kernel (class) and launch of the scan: HIP events of the library around every kernel of the scan stage and
    # around the kernel classes of the stages behind it (GbnDiagnostics.kernel_ms)
    by_kernel = {dom_label: dom_ms / max(launches, 1)}
    if bin_ms > 0:
        by_kernel["probe_bin_kernel"] = probe_ms / max(launches, 1); by_kernel["probe_rare_kernel"] = rare_ms / max(launches, 1)
    for i, name in enumerate(api.GbnDiagnostics.KERNEL_CLASSES):
        t = sum(d.kernel_ms[i] for d in diags) / max(launches, 1)
        if t > 0:
            by_kernel[name] = t
    top_name = max(by_kernel, key=by_kernel.get)
    valu = valu_roofline(args.workload, elapsed / args.steps * 1e3, launches / max(args.steps, 1))

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.cpu_seconds > 0:
        cpu = cpu_baseline(args, queries[:args.batch_queries], opt, mine)

    # ---- the other single-GPU configs beside it (BASELINE.json configs[2], [3]): short runs of this same script in
    # processes of their own once this one's passes are done, their key figures under config.other_workloads
    others = None
    if rank == 0 and world == 1 and args.workload == "C2" and not args.no_side_workloads and not args.side:
        others = side_workloads(dev.index)

    if rank == 0:
        value = total_bases_global * args.steps / elapsed / 1e9
        line = {
            "metric": "subject Gbp scanned/sec (%s preliminary search, DB bases x passes / wall)" % task,
            "value": value, "unit": "Gbp/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_minmax": [min(region_ms), max(region_ms)], "regions": len(regions),
            "regions_what": "timed regions of exactly `steps` passes each (barrier + synchronize either side); ms_per_step and value are the median region's; "
                            "the pipeline stays primed between regions: each region also sets up the first two query batches of the next one (and finds its own first two set up)",
            "higher_is_better": True, "scaling": "strong" if (args.strong and world > 1) else "weak", "vs_baseline": None,
            "dtype": "u8 (2-bit packed bases, int32 scores)", "data": "synthetic",
            "config": {
                "workload": "%s: %d x 1 kb queries vs %.1f Gbp synthetic 2-bit DB per GPU, %s W=%d"
                            % (args.workload, len(queries), nsub * slen / 1e9, task, opt.word_size),
                "stage_ms_per_pass": {k: sum(getattr(d, k) for d in diags) / max(launches, 1)
                                      for k in ["scan_stage_ms", "seed_stage_ms", "gapped_stage_ms", "host_stage_ms"]},
                "stage_ms_per_pass_what": "HOST wall clock per stage, from its first launch to the stream synchronisation that ends it. scan_stage_ms does not "
                                          "contain a binning kernel that the pass before queued ahead (it ran before this pass's host clock started); "
                                          "roofline.scan_stage.avg_ms is GPU time (HIP events) of all three scan kernels of a pass, wherever they ran",
                "config_wall_ms": npass_config * elapsed / args.steps * 1e3,
                "config_wall_ms_what": "passes_per_config x ms_per_step: every pass bins (record cache off)",
                "config_wall_ms_measured": None if not config_measured else config_measured["ms"],
                "config_measured": config_measured,
                "cached_pass": cached_pass,
                "batch_setup_ms": batch_setup_ms, "engine_only": engine_only, "init_hits_per_pass": sum(d.good_init_extends for d in diags) / max(launches, 1),
                "batch_plan": {"queries_per_batch": args.batch_queries, "passes_per_config": npass_config,
                               "lut": info["lut_width"], "scan_step": info["scan_step"],
                               "lut_type": info["lut_type"], "diag_container": info["container"]},
                "subjects_per_gpu": nsub, "subject_len": slen,
                "parallelism": "db-shard x%d (volumes by rank, RCCL gather of HSP records)%s" % (
                    world, " -- %d ranks SHARE one device: exchange exercised, not a scaling number" % world if shared_device else ""),
                "record_cache": "on" if cache_on else "OFF for value / ms_per_step / roofline: every pass runs the binning kernel (config_measured and cached_pass switch it on, as the library does by default)",
                "query_batches": "set up from scratch in every step (inside the timed region); no result reused between steps",
                "pipeline": "off" if args.no_overlap else
                            "set-up of pass k+1/k+2 (worker threads) and seed/gapped stages + merge of pass k (second HIP stream + host threads) overlap the scan of pass k+1",
                "hsps_per_pass": nhsp / max(args.steps, 1),
                "seeds_per_pass": seeds / max(launches, 1),
                "lookup_hits_per_pass": lookup_hits / max(launches, 1),
                "other_workloads": others,
                "skew": None if not args.skew else {
                    "what": "gbn_synth_skew over the shard (8 % of every subject homopolymer runs / tandem repeats, a 1,200-base family element in one subject "
                            "of fifty), 2 % of the queries carry a piece of the element; parity of this shape against the oracle: "
                            "tests/test_workload_size_gpu.py::test_skewed_shard_against_the_oracle",
                    "ranges_per_pass": sum(d.ranges for d in diags) / max(args.steps, 1), "scan_launches_per_pass": launches / max(args.steps, 1),
                    "rescans_per_pass": sum(d.scan_rescans for d in diags) / max(args.steps, 1),
                    "direct_kernel_ranges_per_pass": sum(d.direct_ranges for d in diags) / max(args.steps, 1),
                    "library_sort_launches_per_pass": sum(d.library_sorts for d in diags) / max(args.steps, 1),
                    "seeds_per_pass": seeds / max(args.steps, 1), "init_hits_per_pass": sum(d.good_init_extends for d in diags) / max(args.steps, 1)},
            },
            "roofline": {"bound": "hbm", "kernel": dom_label,
                         "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "box_copy_GBps": box_copy,
                         "frac": achieved / 8000.0, "traffic": traffic,
                         "traffic_source": ("profiles/scan_traffic.json (%s): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, "
                                            "corrected as MI355X_MICROARCH.md prescribes; a constant of that profile, not a counter of this run"
                                            % traffic_tag) if traffic is not None else None,
                         "algorithmic_bytes_per_launch": algo_bytes / max(launches, 1),
                         "avg_launch_ms": dom_ms / max(launches, 1), "launches": launches,
                         "scan_stage": {"kernels": dom_label + " + probe_bin_kernel + probe_rare_kernel"
                                        if bin_ms > 0 else dom_name,
                                        "avg_ms": scan_ms / max(launches, 1),
                                        "avg_ms_by_kernel": [bin_ms / max(launches, 1), probe_ms / max(launches, 1),
                                                             rare_ms / max(launches, 1)],
                                        "achieved": stage_achieved, "frac": stage_achieved / 8000.0},
                         "gpu_ms_per_launch_by_kernel": by_kernel,
                         "dominant_kernel_by_gpu_time": {"kernel": top_name, "avg_ms_per_launch": by_kernel[top_name],
                                                         "what": "HIP-event time per scan launch (= subject range), kernels running next to other streams' work included"},
                         "valu": valu},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def valu_roofline(workload, ms_per_step, launches_per_step):
    """VALU issue roofline of a whole step: wave-instructions per step (SQ_INSTS_VALU summed over the kernels of the committed
    rocprofv3 --pmc pass named in profiles/valu_counts.json -- a constant of that profile, not a counter of this run) against
    what 1,024 SIMDs issue in ms_per_step.  tools/valu_microbench.hip (profiles/r04_valu_microbench.txt): a wave64 add / sub /
    and / or / xor / mov / lshr issues every 2.4 cycles, everything else integer (shifts left, bfe, alignbit, perm, cmp, min /
    max, cndmask, three-operand forms, DPP) every 4.4; the fraction is given for both."""
    tf = os.path.join(ROOT, "profiles", "valu_counts.json")
    if not os.path.exists(tf):
        return None
    try:
        e = json.load(open(tf)).get(workload)
        if not e:
            return None
        n = float(e["valu_wave_instructions_per_launch"]) * launches_per_step
        clock = 2.4e9
        per_ms = lambda cyc: 1024 * clock / cyc * 1e-3
        return {"bound": "valu issue", "wave_instructions_per_step": n, "source": e.get("source"),
                "frac_at_4.4_cycles": n / (per_ms(4.4) * ms_per_step), "frac_at_2.4_cycles": n / (per_ms(2.4) * ms_per_step),
                "peak_wave_instructions_per_ms": [per_ms(4.4), per_ms(2.4)], "by_kernel_per_launch": e.get("by_kernel")}
    except Exception:
        return None


def slice_kernel_name(info):
    """the slice scan's kernel as rocprof names it: tables of more than 2^20 cells (lut 11, 12) are scanned through the folded
    filter + rank tables (scan_fold_kernel) unless GBN_SLICE_FOLD=0 asks for a pass per slice"""
    folded = 2 * int(info.get("lut_width", 0)) > 20 and os.environ.get("GBN_SLICE_FOLD", "1") != "0"
    if folded and os.environ.get("GBN_SCAN_ORDERED", "1") != "0":
        return "scan_fold_ordered_kernel"       # ... with the seeds in scan order
    return "scan_fold_kernel" if folded else "scan_slice_kernel"


def side_workloads(device_index):
    """C3 (blastn W=11, 100 kb batches vs 5 Gbp) and C4 (5 Mb batches streamed through the host pipeline with the
    traceback overlapped) for >= 1 s of timed region each: ms per pass / batch, Gbp/s, dominant kernel and its fraction of
    the HBM roofline.  Each is `python bench.py --workload ...` in a process of its own (its full line is what that command
    prints); a failure is reported, it does not fail the C2 line."""
    import subprocess
    out = {}
    for wl, steps in (("C3", "32"), ("C4", "80"), ("shim", "10")):
        # C3 carries a CPU baseline of its own (the oracle on the same 100-query batch, ~6 s per host core); C4's preliminary
        # search is C2's -- its baseline is the C2 line's
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", wl, "--steps", steps, "--warmup", "2",
               "--engine-steps", "0", "--min-seconds", "1.0", "--side"] + (["--cpu-seconds", "6"] if wl == "C3" else ["--no-cpu-baseline"])
        env = dict(os.environ); env["HIP_VISIBLE_DEVICES"] = env.get("HIP_VISIBLE_DEVICES", "")
        if not env["HIP_VISIBLE_DEVICES"]:
            del env["HIP_VISIBLE_DEVICES"]
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
            j = json.loads(p.stdout.strip().splitlines()[-1])
            if wl == "shim":
                out[wl] = j
                continue
            r = j["roofline"]
            out[wl] = {"workload": j["config"]["workload"], "ms_per_step": j["ms_per_step"], "ms_per_step_minmax": j.get("ms_per_step_minmax"),
                       "steps": j["steps"], "regions": j.get("regions"), "value": j["value"], "unit": j["unit"],
                       "step_is": "one 100-query batch over the 5 Gbp shard (5 subject ranges)" if wl == "C3" else "one 5,000-query batch from the caller's arrays to its final alignments",
                       "record_cache": j["config"].get("record_cache"),
                       "dominant_kernel": (r.get("dominant_kernel_by_gpu_time") or {}).get("kernel", r.get("kernel")),
                       "dominant_kernel_avg_launch_ms": (r.get("dominant_kernel_by_gpu_time") or {}).get("avg_ms_per_launch", r.get("avg_launch_ms")),
                       "gpu_ms_per_launch_by_kernel": r.get("gpu_ms_per_launch_by_kernel"),
                       "launches_per_step": (r.get("launches") or 0) / max(j["steps"], 1),
                       "scan_kernel": r.get("kernel"), "scan_kernel_hbm_frac": r.get("frac"), "valu": r.get("valu"),
                       "stage_ms_per_launch": j["config"].get("stage_ms_per_pass"),
                       "cpu_baseline": j.get("cpu_baseline") if wl == "C3" else "the preliminary search is C2's: see this line's cpu_baseline",
                       "command": "python bench.py --workload %s --steps %s" % (wl, steps)}
        except Exception as e:      # noqa
            out[wl] = {"error": repr(e)[:300]}
    return out


def bench_c4(args, api, torch, dist, world, rank, dev, src, qsets, nbatch, opt, total_bases_global, nsub, slen, queries):
    """C4: query batches streamed through the C++ host pipeline (gblastn_amd_host.hpp CSearchPipeline behind its C
    ABI): set-up thread, preliminary search on the GPU, `--trace-threads` traceback consumers -- a step is one
    5 Mb batch from the caller's arrays to its final alignments (edit scripts, identities, e-values)."""
    def run(count):
        pipe = api.SearchPipeline(opt, src, trace_threads=args.trace_threads, traceback=not args.no_traceback, overlap=not args.no_overlap)
        sub = got = 0; nfinal = 0; diags = []
        while got < count:
            while sub < count and sub < got + 8:
                pipe.submit(qsets[sub % nbatch]); sub += 1
            if sub == count:
                pipe.finish()
            k, res, d = pipe.next(read=False)
            diags.append(d); got += 1
        pipe.close()
        return diags

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    run(max(args.warmup, 2))
    regions = []
    while True:
        sync(); t0 = time.perf_counter()
        dg = run(args.steps)
        sync(); el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); el = float(t.item())
        regions.append((el, dg))
        if (sum(r[0] for r in regions) >= args.min_seconds and len(regions) >= 2) or len(regions) >= 64 or world > 1:
            break
    order = sorted(range(len(regions)), key=lambda i: regions[i][0])
    elapsed, diags = regions[order[(len(order) - 1) // 2]]
    region_ms = [r[0] / args.steps * 1e3 for r in regions]
    # final alignments of one batch, counted once outside the timed region
    pipe = api.SearchPipeline(opt, src, trace_threads=args.trace_threads, traceback=True, overlap=False)
    pipe.submit(qsets[0]); pipe.finish(); _, res, _ = pipe.next(); pipe.close()
    rec = res[0]
    scan_ms = sum(d.scan_kernel_ms for d in diags); launches = sum(d.scan_launches for d in diags)
    bin_ms = sum(d.bin_kernel_ms for d in diags); probe_ms = sum(d.probe_kernel_ms for d in diags); rare_ms = sum(d.rare_kernel_ms for d in diags)
    scanned = sum(d.subject_bases_scanned for d in diags)
    algo = 0.25 * scanned
    by_kernel = {"scan_bin_kernel_s17": bin_ms / max(launches, 1), "probe_bin_kernel": probe_ms / max(launches, 1), "probe_rare_kernel": rare_ms / max(launches, 1)}
    for i, name in enumerate(api.GbnDiagnostics.KERNEL_CLASSES):
        t = sum(d.kernel_ms[i] for d in diags) / max(launches, 1)
        if t > 0:
            by_kernel[name] = t
    cache = api.record_cache_stats()
    if rank == 0:
        line = {
            "metric": "subject Gbp scanned/sec (megablast, query batches streamed through preliminary search + overlapped CPU traceback)",
            "value": total_bases_global * args.steps / elapsed / 1e9, "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "ms_per_step_minmax": [min(region_ms), max(region_ms)], "regions": len(regions),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 (2-bit packed bases, int32 scores)", "data": "synthetic",
            "config": {"workload": "C4: %d queries streamed in %d batches of %d x 1 kb (cycling over %d distinct queries) vs %.1f Gbp per GPU, megablast W=%d, traceback on %d host threads"
                                   % (args.steps * args.batch_queries, args.steps, args.batch_queries, len(queries), nsub * slen / 1e9, opt.word_size, args.trace_threads),
                       "pipeline": "set-up thread -> preliminary search (GPU) -> traceback threads; overlapped" if not args.no_overlap else "one batch at a time",
                       "record_cache": ("on (the library's default: the shard's scan records are binned by the first batch and stay resident; later batches run "
                                        "probe + rare kernel only) -- %d passes served from the cache, %d binned, %.1f GB of records resident"
                                        % (cache["hits"], cache["misses"], cache["bytes"] / 1e9)) if cache["limit"] > 0 else "off (--record-cache off): every batch bins",
                       "stage_ms_per_pass": {k: sum(getattr(d, k) for d in diags) / max(launches, 1)
                                             for k in ["scan_stage_ms", "seed_stage_ms", "gapped_stage_ms", "host_stage_ms"]},
                       "final_hsps_per_batch": int(len(rec)), "final_identity_mean": float((rec["num_ident"] / np.maximum(rec["align_length"], 1)).mean()) if len(rec) else None,
                       "gapped_alignments_per_batch": int((rec["gaps"] > 0).sum()) if len(rec) else 0},
            "roofline": {"bound": "hbm", "kernel": "probe_bin_kernel" if cache["limit"] > 0 else "scan_bin_kernel_s17",
                         "achieved": algo / ((probe_ms if cache["limit"] > 0 else bin_ms) * 1e-3) / 1e9 if bin_ms + probe_ms else 0.0, "peak": 8000.0, "unit": "GB/s",
                         "frac": (algo / ((probe_ms if cache["limit"] > 0 else bin_ms) * 1e-3) / 1e9 / 8000.0) if bin_ms + probe_ms else 0.0, "traffic": None,
                         "avg_launch_ms": (probe_ms if cache["limit"] > 0 else bin_ms) / max(launches, 1), "launches": launches,
                         "scan_stage": {"avg_ms": scan_ms / max(launches, 1), "avg_ms_by_kernel": [bin_ms / max(launches, 1), probe_ms / max(launches, 1), rare_ms / max(launches, 1)],
                                        "achieved": algo / (scan_ms * 1e-3) / 1e9 if scan_ms else 0.0, "frac": (algo / (scan_ms * 1e-3) / 1e9 / 8000.0) if scan_ms else 0.0},
                         "gpu_ms_per_launch_by_kernel": by_kernel,
                         "dominant_kernel_by_gpu_time": {"kernel": max(by_kernel, key=by_kernel.get), "avg_ms_per_launch": max(by_kernel.values())},
                         "valu": valu_roofline("C4", elapsed / args.steps * 1e3, launches / max(args.steps, 1))},
            "cpu_baseline": None,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


def bench_shim(args, api, torch, dev, slab, mine, src, qsets, nbatch, opt, nsub, slen):
    """The drop-in boundary's real path, timed: what Blast_gpu_RunPreliminarySearchWithInterrupt of
    gblastn_amd/shim/gpu_blastn_amd_shim.cpp does per query batch, through the same C-ABI calls (the shim itself needs the
    configured toolkit to compile).  The shard is 100 OID chunks (a hundredth of the database each, the reference's chunk:
    GB/gpu_blastn_pre_search_engine.cpp:1243); one "call" = set the batch up (gbn_batch_new_masked), then for every group of
    chunks: gbn_block_cache_find per chunk, gbn_block_view over the group's blocks, gbn_prelim_search_begin, and -- for the
    group before -- gbn_prelim_search_end + gbn_results_emit_lists into a counting sink; free the batch.  Calls follow each other
    without any overlap between them, as CPrelimSearchRunner issues them.
      warm:  the blocks are resident (every call after a thread's first) -- per group size, incl. 1 chunk per group through
             the synchronous gbn_prelim_search_lists (round 4's shim loop) and the whole shard as one begin / end
      cold:  the first call: every block is uploaded from host memory (gbn_db_new from a host slab) when the loop reaches it,
             under the search of the group before"""
    import ctypes as C
    L = api.lib()
    nchunk = 100
    per = nsub // nchunk
    name = b"bench-shim-db"
    boff, lens = np.asarray(mine.byte_off, dtype=np.int64), np.asarray(mine.lens, dtype=np.int32)
    oid_arr = [np.arange(k * per, (k + 1) * per, dtype=np.int32) + mine.first_oid for k in range(nchunk)]
    spans = []
    for k in range(nchunk):
        a = int(boff[k * per]) - 16
        z = int(boff[(k + 1) * per - 1]) + (int(lens[(k + 1) * per - 1]) + 3) // 4 + 128
        spans.append((a, z))

    def new_block(k, host=None):
        a, z = spans[k]
        h = C.c_void_p()
        off = np.ascontiguousarray(boff[k * per:(k + 1) * per] - a)
        ln = np.ascontiguousarray(lens[k * per:(k + 1) * per])
        ptr = host.ctypes.data if host is not None else slab.data_ptr() + a
        api._check(L.gbn_db_new(C.byref(h), ptr, z - a, per, off.ctypes.data, ln.ctypes.data, int(oid_arr[k][0]), 0 if host is not None else 1))
        kept = C.c_void_p()
        api._check(L.gbn_block_cache_insert(name, oid_arr[k].ctypes.data, per, h, C.byref(kept)))
        return kept

    def get_block(k, host_of=None):
        out = C.c_void_p()
        api._check(L.gbn_block_cache_find(name, oid_arr[k].ctypes.data, per, C.byref(out)))
        if out.value:
            return out
        return new_block(k, None if host_of is None else host_of(k))

    res = [C.c_void_p(), C.c_void_p()]
    for r in res:
        api._check(L.gbn_results_new(C.byref(r)))
    sink = C.cast(L.gbn_debug_counting_sink, api.GbnHspListFn)
    counts = (C.c_longlong * 2)()
    views_refused = [0]

    def call(qs, group, diag, host_of=None, style="pipelined"):
        """one Blast_gpu_RunPreliminarySearchWithInterrupt"""
        b = C.c_void_p()
        none = (C.c_int32 * 1)()
        api._check(L.gbn_batch_new_masked(C.byref(b), C.byref(opt), len(qs), qs.ptrs, qs.lens, 0, none, none, none, 1))
        cur, in_flight = 0, False
        for g0 in range(0, nchunk, group):
            blocks = [get_block(k, host_of) for k in range(g0, min(g0 + group, nchunk))]
            if style == "lists":                        # round 4: one synchronous search per chunk
                for blk in blocks:
                    api._check(L.gbn_prelim_search_lists(b, blk, sink, counts, C.byref(diag), None, None))
                continue
            arr = (C.c_void_p * len(blocks))(*[x.value for x in blocks])
            view = C.c_void_p()
            one = L.gbn_block_view(arr, len(blocks), C.byref(view)) == 0 and view.value
            if not one:
                views_refused[0] += 1                   # (slabs too far apart for one view: block by block, as the shim does)
            for target in ([view] if one else blocks):
                L.gbn_results_clear(res[cur])
                api._check(L.gbn_prelim_search_begin(b, target, res[cur], C.byref(diag), None, None))
                if in_flight:
                    api._check(L.gbn_prelim_search_end(res[cur ^ 1]))
                    api._check(L.gbn_results_emit_lists(res[cur ^ 1], sink, counts))
                in_flight = True; cur ^= 1
        if in_flight:
            api._check(L.gbn_prelim_search_end(res[cur ^ 1]))
            api._check(L.gbn_results_emit_lists(res[cur ^ 1], sink, counts))
        L.gbn_batch_free(b)

    def measure(group, steps, style="pipelined", regions=3):
        # every grouping has the record cache to itself (what the groupings before left is freed outside the clock); round 4's loop
        # runs the way round 4 ran it: no record cache, every search bins its block
        api.record_cache_set_limit(0)
        if style != "lists":
            api.record_cache_set_limit(-1)
        call(qsets[0], group, api.GbnDiagnostics(), style=style)      # the records / views of this grouping exist
        out = []
        for _ in range(regions):
            d = api.GbnDiagnostics(); counts[0] = counts[1] = 0
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(steps):
                call(qsets[(k + 1) % nbatch], group, d, style=style)
            torch.cuda.synchronize()
            out.append(((time.perf_counter() - t0) / steps * 1e3, d, counts[0] / steps, counts[1] / steps))
        out.sort(key=lambda r: r[0])
        ms, d, nl, nh = out[(len(out) - 1) // 2]
        n = max(1, d.scan_launches)
        return {"ms_per_batch": ms, "ms_per_batch_minmax": [out[0][0], out[-1][0]], "regions": regions, "batches": steps, "searches_per_batch": d.scan_launches / steps,
                "gbp_per_s": nsub * slen / ms / 1e6, "lists_per_batch": nl, "hsps_per_batch": nh,
                "scan_kernels_ms_per_batch": [d.bin_kernel_ms / steps, d.probe_kernel_ms / steps, d.rare_kernel_ms / steps]}

    # ---- cold: a fresh block cache, host-resident database.  The host copy is taken before the clock starts.
    host_slab = torch.empty(mine.nbytes, dtype=torch.uint8, pin_memory=False)
    host_slab.copy_(slab); torch.cuda.synchronize()
    hs = host_slab.numpy()
    host_of = lambda k: hs[spans[k][0]:spans[k][1]]
    default_group = 100          # what the shim searches at a time with ONE leased GPU: every chunk the iterator still has, as one view
    up0 = L.gbn_debug_db_bytes_uploaded()
    d = api.GbnDiagnostics(); counts[0] = counts[1] = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    call(qsets[0], default_group, d, host_of=host_of)
    torch.cuda.synchronize()
    cold_ms = (time.perf_counter() - t0) * 1e3
    uploaded = L.gbn_debug_db_bytes_uploaded() - up0
    cold = {"ms_first_batch": cold_ms, "uploaded_GB": uploaded / 1e9, "effective_upload_GBps": uploaded / 1e9 / (cold_ms * 1e-3),
            "what": "first call on an empty block cache: every block uploaded from pageable host memory by a synchronous copy when the loop reaches it, "
                    "then ONE search over the view of all %d blocks, which bins its records on the way" % default_group}
    del hs, host_slab
    warm = {}
    for group, style, tag in ((default_group, "pipelined", "group_100_chunks (the shim's default with one GPU: one view over all blocks)"), (25, "pipelined", "group_25_chunks (four searches per batch)"),
                              (3, "pipelined", "group_3_chunks (the group size with eight leased GPUs; here ONE GPU searches all 34 groups -- each of eight would search four)"), (1, "pipelined", "group_1_chunk (pipelined begin / end per chunk)"),
                              (1, "lists", "round_4_loop (one synchronous gbn_prelim_search_lists per chunk)")):
        warm[tag] = measure(group, args.steps if group > 3 else max(2, args.steps // 5), style=style)
    # the same batches against the whole shard made the usual way (one GbnDb over the slab), set-up included, nothing overlapped
    api.record_cache_set_limit(0); api.record_cache_set_limit(-1)
    ps = api.BlastPrelimSearch(qsets[0], opt, src); ps.run(); ps.close()
    out = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(args.steps):
            ps = api.BlastPrelimSearch(qsets[(k + 1) % nbatch], opt, src); ps.begin(); ps.end(); ps.close()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / args.steps * 1e3)
    out.sort()
    api.record_cache_set_limit(-1)
    head = warm["group_100_chunks (the shim's default with one GPU: one view over all blocks)"]
    cache = api.record_cache_stats()
    line = {"metric": "ms per query batch through the shim-shaped loop (C2 shard as 100 resident OID-chunk blocks, megablast 5,000 x 1 kb per batch)",
            "value": head["ms_per_batch"], "unit": "ms", "higher_is_better": False, "n_gpus": 1, "steps": args.steps, "warmup": 1,
            "ms_per_step": head["ms_per_batch"], "scaling": "weak", "vs_baseline": None, "dtype": "u8 (2-bit packed bases, int32 scores)", "data": "synthetic",
            "config": {"workload": "shim: %d x 1 kb queries per call vs %.1f Gbp as %d blocks of %d subjects, megablast W=%d" % (args.batch_queries, nsub * slen / 1e9, nchunk, per, opt.word_size),
                       "call_is": "gbn_batch_new_masked + the loop over the chunks' groups (block cache look-ups, view, begin; end + lists of the group before) + gbn_batch_free; "
                                  "calls back to back, nothing of one call overlaps the next (the set-up of a batch, %s ms alone, is not hidden as in the C2 headline)" % "4-5",
                       "record_cache": "on (library default): %d passes served from the cache, %d binned, %.1f GB of records resident at the end" % (cache["hits"], cache["misses"], cache["bytes"] / 1e9),
                       "warm": warm, "cold": cold, "views_refused": views_refused[0],
                       "whole_shard_same_calls_ms": {"ms_per_batch": out[1], "minmax": [out[0], out[-1]], "what": "set-up + gbn_prelim_search_begin / _end over ONE GbnDb of the whole shard + free, back to back"}},
            "roofline": None, "cpu_baseline": None}
    for r in res:
        L.gbn_results_free(r)
    L.gbn_release_db_memory()
    print(json.dumps(line))


def _cpu_worker(job):
    """One host core: the oracle over subjects w, w + cores, ... of the rank-0 shard for ~seconds."""
    w, cores, seconds, queries, optd, nsub, slen, seed, first_oid = job
    from oracle import orc
    from tests import util
    from gblastn_amd import api, synth
    gopt = api.default_options("megablast")
    for k, v in optd.items():
        setattr(gopt, k, v)
    layout = synth.SynthDb(nsub, slen, seed=seed, first_oid=first_oid)
    s = orc.Search(util.oracle_options(gopt), queries)
    done, t, i = 0, 0.0, w
    while t < seconds and i < layout.num:
        packed = layout.subject_packed(i)
        t0 = time.perf_counter()
        s.subject(packed, layout.length)
        t += time.perf_counter() - t0
        done += layout.length
        i += cores
    return done, t


def cpu_baseline(args, batch_queries, gopt, layout):
    """The oracle (a scalar C port of the reference algorithm: lookup word cut from the packed bytes,
    presence-vector test before the table, as the reference's scanners do) on this box's host cores:
    the same query batch, a bounded sample of the same shard's subjects, one process per core (the
    reference shares OID chunks among threads the same way, x_LaunchMultiThreadedSearch)."""
    import multiprocessing as mp
    from gblastn_amd import api
    cores = args.cpu_cores if args.cpu_cores > 0 else max(1, min(64, (os.cpu_count() or 1) // 2))
    optd = {f: getattr(gopt, f) for f, _ in api.GbnOptions._fields_}
    jobs = [(w, cores, args.cpu_seconds, batch_queries, optd, layout.num, layout.length, layout.seed, layout.first_oid)
            for w in range(cores)]
    ctx = mp.get_context("spawn")                       # no HIP state in the children
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, jobs)
    done = sum(r[0] for r in res)
    t = max(r[1] for r in res)
    single = max(r[0] / r[1] for r in res if r[1] > 0) / 1e9
    return {"value": done / t / 1e9 if t > 0 else 0.0, "unit": "Gbp/s", "cores": cores, "kind": "port",
            "single_core_value": single,
            "sample": "%d subjects (%.0f Mbp) of the rank-0 shard spread over %d processes, one query batch of %.2f Mb, %.1f s each"
                      % (done // layout.length, done / 1e6, cores, sum(len(q) for q in batch_queries) / 1e6, t)}


if __name__ == "__main__":
    main()
