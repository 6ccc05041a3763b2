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
import types

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
    ap.add_argument("--trace-threads", type=int, default=0, help="C4: traceback consumer threads (0: the CPUs granted to this process -- gbn_host_cpus --, 4 .. 16; round 4 ran 4, "
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
    if a.subjects is None:
        a.subjects = 5_000 if a.workload == "C3" else 50_000
    if a.batch_queries is None:
        a.batch_queries = 100 if a.workload == "C3" else 5_000     # 5 Mb megablast / 100 kb blastn batches
    return a


def open_run(args):
    """Everything the timed region needs, made OUTSIDE it: devices and process group, the rank's shard generated in HBM, the queries, the
    options, the query batches as the caller hands them over, and the closures that run passes (run_passes) and time them (timed).
    Returns a namespace of them -- or the exit status of a workload that is a program of its own (cli, shim, C4)."""
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
        # the ranks of one node share what the node (or its container) grants: every rank sizes its host pools from ITS share of the CPUs,
        # not from all of them (gbn_host_cpus reads GBN_HOST_CPUS at its first use, which is below)
        api.share_cpus_among_local_ranks()
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

    return types.SimpleNamespace(**{k: v for k, v in locals().items() if k != 'args'}, args=args)


def warm_up(R):
    """allocator warm-up, one set-up timed alone, --warmup passes (all untimed); sets R.info / R.batch_setup_ms"""
    args, keep_primed, make, run_passes, torch = R.args, R.keep_primed, R.make, R.run_passes, R.torch
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
    R.info, R.batch_setup_ms = info, batch_setup_ms


def timed_regions(R):
    """THE TIMED REGION, repeated: exactly --steps passes between barrier + synchronize on both sides (timed), the median region reported"""
    args, dev, dist, keep_primed, primed, run_passes, timed, torch = R.args, R.dev, R.dist, R.keep_primed, R.primed, R.run_passes, R.timed, R.torch
    world, = R.world,
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

    return regions, elapsed, nhsp, diags, region_ms


def measure_engine_only(R):
    """beside the headline: the engine entry point alone on reused query batches"""
    args, make, merge, merger, nbatch, nsub, slen, timed, world = R.args, R.make, R.merge, R.merger, R.nbatch, R.nsub, R.slen, R.timed, R.world
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

    return engine_only


def cached_roofline(rcs, dgc):
    """roofline block of the pass over cached records.  Stream form: the algorithmic bytes of the scan (0.25 B per subject base) over the
    probe kernel's time, as for the headline.  Sorted form: that figure would exceed the HBM peak -- the kernel reads the runs of the cells
    the batch occupies out of a query-independent index, never the subject -- so `achieved` / `frac` are the bytes the kernel MOVES
    (profiles/scan_traffic.json, from profiles/r06c_cached_pmc.csv) over its time, and the algorithmic rate stands beside them."""
    algo = 0.25 * sum(d.subject_bases_scanned for d in dgc)
    probe_ms = max(sum(d.probe_kernel_ms for d in dgc), 1e-9); scan_ms = max(sum(d.scan_kernel_ms for d in dgc), 1e-9)
    nl = max(1, sum(d.scan_launches for d in dgc))
    if not rcs.get("sorted_sets"):
        return {"bound": "hbm", "kernel": "probe_bin_kernel", "peak": 8000.0, "unit": "GB/s", "achieved": algo / probe_ms / 1e6, "frac": algo / probe_ms / 1e6 / 8000.0,
                "scan_stage_frac": algo / scan_ms / 1e6 / 8000.0, "traffic": 13.1e9 + 2.9e9,
                "traffic_what": "probe 13.1 GB + rare 2.9 GB per pass (profiles/scan_traffic.json): 1.28 x the algorithmic 12.5 GB, against 3.8 x for a pass that bins"}
    probe_b, rare_b = 5.75e9, 2.73e9
    try:
        tj = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "scan_traffic.json")))
        probe_b = float(tj["probe_runs_kernel"]["hbm_bytes_per_launch"]); rare_b = float(tj["probe_rare_kernel (over sorted records)"]["hbm_bytes_per_launch"])
    except Exception:
        pass
    return {"bound": "hbm", "kernel": "probe_runs_kernel", "peak": 8000.0, "unit": "GB/s",
            "achieved": probe_b * nl / probe_ms / 1e6, "frac": probe_b * nl / probe_ms / 1e6 / 8000.0,
            "achieved_is": "HBM bytes the kernel moves per launch (traffic, profiles/r06c_cached_pmc.csv) / its average launch time: this kernel has no algorithmic bytes "
                           "in the sense of the scan's 0.25 B per subject base -- it reads the runs of the cells the batch occupies out of a resident index, not the subject",
            "scan_stage_frac": (probe_b + rare_b) * nl / scan_ms / 1e6 / 8000.0,
            "subject_bytes_equivalent_rate": algo / probe_ms / 1e6,
            "subject_bytes_equivalent_what": "0.25 B per subject base / the probe kernel's time, GB/s: above the HBM peak because the pass does not touch what a scan of the subject would",
            "traffic": probe_b + rare_b,
            "traffic_what": "probe_runs_kernel %.2f GB + rare kernel %.2f GB per pass (constants of profiles/scan_traffic.json, not counters of this run): %.2f x the 12.5 GB a scan of "
                            "the subject reads, against 3.8 x for a pass that bins and 1.28 x for a pass over records in stream form" % (probe_b / 1e9, rare_b / 1e9, (probe_b + rare_b) / 12.5e9)}


def measure_config_and_cached_pass(R):
    """beside the headline: the whole config as one cold region with the library's default policy, and later batches over cached records"""
    api, args, keep_primed, nbatch, npass_config, opt, primed, qsets, run_passes = R.api, R.args, R.keep_primed, R.nbatch, R.npass_config, R.opt, R.primed, R.qsets, R.run_passes
    src, timed, total_bases_global = R.src, R.timed, R.total_bases_global
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
                       "roofline": cached_roofline(rcs, dgc),
                       "what": "the same step as the headline (set-up from scratch, scan, extension, merge) with the record cache ON and the shard's records "
                               "resident: the binning kernel does not run -- NOT the headline metric (that one bins in every pass)"}
        # ... and a batch of ONE 1 kb query over the resident shard (a table of 4^8 cells, stride 21: a record set of its own): what a pass
        # costs when the batch occupies a handful of cells -- stream form reads every record, the sorted form the runs of those cells
        try:
            one = api.BlastPrelimSearch(api.QuerySet(R.queries[:1]), opt, src)
            for _ in range(4):                              # bins, is hit, is sorted at the second hit, is hit again
                one.run()
            d0 = (one.diagnostics.bin_kernel_ms, one.diagnostics.probe_kernel_ms, one.diagnostics.rare_kernel_ms, one.diagnostics.scan_launches)
            n1 = 40
            _, el1 = timed(lambda: [one.run() for _ in range(n1)])
            d1 = (one.diagnostics.bin_kernel_ms, one.diagnostics.probe_kernel_ms, one.diagnostics.rare_kernel_ms, one.diagnostics.scan_launches)
            st1 = api.record_cache_stats()
            cached_pass["one_query_pass"] = {"ms_per_pass": el1 / n1 * 1e3, "passes": n1, "table": one.info(),
                                             "scan_kernels_ms": [(d1[i] - d0[i]) / max(d1[3] - d0[3], 1) for i in range(3)],
                                             "sorted_sets": st1.get("sorted_sets"), "resident_bytes_all_sets": st1["bytes"],
                                             "what": "gbn_prelim_search of a set-up 1-query batch over the cached records of its table shape, run to completion one after the other"}
            one.close()
        except Exception as e:      # noqa
            cached_pass["one_query_pass"] = {"error": repr(e)[:200]}
        api.record_cache_set_limit(0)

    return config_measured, cached_pass


def report(R, regions, elapsed, nhsp, diags, region_ms, engine_only, config_measured, cached_pass):
    """roofline of the dominant kernel from the library's HIP events, CPU baseline, side workloads, and the one JSON line"""
    api, args, batch_setup_ms, box_copy, cache_on, dev, dist, info, mine = R.api, R.args, R.batch_setup_ms, R.box_copy, R.cache_on, R.dev, R.dist, R.info, R.mine
    npass_config, nsub, opt, queries, rank, shared_device, slen, task, total_bases_global, world = R.npass_config, R.nsub, R.opt, R.queries, R.rank, R.shared_device, R.slen, R.task, R.total_bases_global, R.world
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
                         "box_copy_GBps": box_copy, "box": box_facts(dev),
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


def box_facts(dev):
    """what may tell one box from another (the binning kernel's time differs 7.0-7.7 ms between boxes): the device's clocks as the runtime
    reports them and, when rocm-smi answers, the power cap and the current clocks -- best effort, never fails the line"""
    out = {}
    try:
        import torch
        p = torch.cuda.get_device_properties(dev)
        for k in ("name", "gcnArchName", "multi_processor_count", "clock_rate", "memory_clock_rate", "memory_bus_width", "total_memory", "L2_cache_size"):
            if hasattr(p, k):
                out[k] = getattr(p, k)
    except Exception as e:      # noqa
        out["props_error"] = repr(e)[:100]
    try:
        import subprocess
        r = subprocess.run(["rocm-smi", "--showpower", "--showmaxpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=20)
        j = json.loads(r.stdout)
        card = j.get("card%d" % (dev.index or 0)) or next(iter(j.values()))
        out["rocm_smi"] = {k: v for k, v in card.items() if any(t in k.lower() for t in ("power", "sclk", "mclk", "fclk"))}
    except Exception as e:      # noqa
        out["rocm_smi_error"] = repr(e)[:100]
    return out


def main():
    args = parse()
    R = open_run(args)
    if not isinstance(R, types.SimpleNamespace):
        return R
    warm_up(R)
    regions, elapsed, nhsp, diags, region_ms = timed_regions(R)
    engine_only = measure_engine_only(R)
    config_measured, cached_pass = measure_config_and_cached_pass(R)
    report(R, regions, elapsed, nhsp, diags, region_ms, engine_only, config_measured, cached_pass)


from bench_side import valu_roofline, slice_kernel_name, side_workloads, bench_c4, bench_shim, cpu_baseline  # noqa: E402


if __name__ == "__main__":
    main()
