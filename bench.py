#!/usr/bin/env python
"""bench.py -- subject Gbp scanned per second of the megablast preliminary search.

Workload (BASELINE.json configs[1], "C2"): 10,000 x 1 kb synthetic queries vs a
50 Gbp synthetic 2-bit database (50,000 subjects x 1 Mb, 12.5 GB packed) on ONE
MI355X, megablast word_size 28.  The reference batch plan applies: 5 Mb query
batches (5,000 queries, 10 M lookup words -> megablast table lut 12, stride 17),
so the config is 2 passes over the database.

A "step" = one pass of the whole preliminary path (scan+seed, diagonal filter,
ungapped X-drop, greedy gapped, HSP rules) of ONE query batch over the rank's
resident shard.  Lookup structures of both batches and the database are
resident in HBM before the timed region (the reference's boundary receives
them ready-made too).  value = (bases of all shards x K passes) / max-over-ranks
wall time.  With N > 1 every rank holds its own 50 Gbp shard (weak scaling, the
C5 layout: volumes sharded by rank, global statistics) and rank 0 gathers the
per-shard HSP records over RCCL after every pass, inside the timed region.
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
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["C2", "C3"], default="C2",
                    help="C2 (the metric's config): megablast W=28 vs 50 Gbp; C3: blastn W=11 vs 5 Gbp, 100 kb batches")
    ap.add_argument("--subjects", type=int, default=None, help="subjects per GPU shard")
    ap.add_argument("--subject-len", type=int, default=1_000_000)
    ap.add_argument("--queries", type=int, default=10_000)
    ap.add_argument("--batch-queries", type=int, default=None)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-cores", type=int, default=0, help="processes of the CPU baseline (0: half of the host cores, at most 64)")
    ap.add_argument("--no-overlap", action="store_true", help="run every pass to completion before the next starts")
    ap.add_argument("--stream-steps", type=int, default=16, help="passes of the streamed variant (fresh batch set-up per pass, overlapped); 0: skip")
    ap.add_argument("--reuse-binning", action="store_true",
                    help="NOT the headline metric: keep the query-independent scan records of the shard in HBM and "
                         "skip the binning kernel for later batches with the same table shape (a database index)")
    a = ap.parse_args()
    if a.subjects is None:
        a.subjects = 50_000 if a.workload == "C2" else 5_000
    if a.batch_queries is None:
        a.batch_queries = 5_000 if a.workload == "C2" else 100     # 5 Mb megablast / 100 kb blastn batches
    return a


def main():
    args = parse()
    os.environ["GBN_REUSE_BINNING"] = "1" if args.reuse_binning else "0"
    import torch
    import torch.distributed as dist
    from gblastn_amd import api, synth, shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    rc = api.lib().Blast_gpu_Init(1, local_rank)
    if rc:
        raise SystemExit("Blast_gpu_Init failed: %s" % api.lib().gbn_last_error().decode())

    # ---- database shard of this rank, generated in HBM ----
    nsub, slen = args.subjects, args.subject_len
    layouts = [synth.SynthDb(nsub, slen, seed=0x9E3779B97F4A7C15 ^ (r + 1), first_oid=r * nsub)
               for r in range(world)]
    mine = layouts[rank]
    slab = torch.empty(mine.nbytes, dtype=torch.uint8, device=dev)
    api._check(api.lib().gbn_synth_fill(slab.data_ptr(), mine.nbytes, mine.seed, None))
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
    queries, plants = synth.make_queries(args.queries, AnyShard())
    task = "megablast" if args.workload == "C2" else "blastn"
    opt = api.default_options(task, db_length=total_bases_global, db_num_seqs=world * nsub)
    nbatch = (len(queries) + args.batch_queries - 1) // args.batch_queries
    npass_config = nbatch
    nbatch = min(nbatch, max(args.steps, args.warmup, 1))       # only the batches the run touches
    t_setup = time.perf_counter()
    batches = [api.BlastPrelimSearch(queries[i * args.batch_queries:(i + 1) * args.batch_queries], opt, src)
               for i in range(nbatch)]
    torch.cuda.synchronize()
    # outside the timed region (the lookup table is an input of the preliminary search engine): host set-up
    # (concatenation, Karlin-Altschul, cut-offs) + lookup structures built on the device, per query batch
    batch_setup_ms = (time.perf_counter() - t_setup) * 1e3 / max(nbatch, 1)
    info = batches[0].info()

    def merge(b, out):
        # exchange + merge step: gather to rank 0, replay through the per-query top-N collector
        got = shard.collect_on_root(out["hsps"], len(b._q), opt.hitlist_size, dst=0, device=dev)
        return 0 if got is None else len(got[0])

    from concurrent.futures import ThreadPoolExecutor
    merger = ThreadPoolExecutor(max_workers=1, initializer=lambda: torch.cuda.set_device(dev))

    def run_passes(first, count):
        """`count` passes, software-pipelined when there are two batches to alternate: the gapped
        stage + host acceptance + gather/merge of pass k overlap the scan of pass k + 1.  Every pass
        is complete (merged on rank 0) when this returns."""
        n = 0
        if count <= 0:
            return 0
        if nbatch < 2 or args.no_overlap:
            for k in range(first, first + count):
                b = batches[k % nbatch]
                n += merge(b, b.run())
            return n
        # begin(k) returns when scan k is done and its extension stages are in flight; the gather + top-N
        # merge of pass k-1 (host work, or an RCCL gather at N > 1) runs on a worker thread underneath scan k+1
        prev, futs = None, []
        for k in range(first, first + count):
            b = batches[k % nbatch]
            b.begin()                       # waits for prev's extension stages before queueing its own
            if prev is not None:
                futs.append(merger.submit(merge, prev, prev.end()))
            prev = b
        futs.append(merger.submit(merge, prev, prev.end()))
        return n + sum(f.result() for f in futs)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_passes(0, args.warmup)
    sync()
    for b in batches:
        b.diagnostics = api.GbnDiagnostics()
    t0 = time.perf_counter()
    nhsp = run_passes(args.warmup, args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- the same job streamed: every pass sets its query batch up from scratch (host set-up + lookup
    # structures built on the device), on a worker thread underneath the previous pass, as
    # `blastn_prelim -mode 2` does.  Reported beside `value`, which times the engine entry point alone
    # (the lookup table is one of its inputs).
    streamed = None
    if not args.no_overlap and args.stream_steps > 0:
        qsets = [queries[i * args.batch_queries:(i + 1) * args.batch_queries] for i in range(nbatch)]
        qsets = [[np.ascontiguousarray(q, dtype=np.uint8) for q in qs] for qs in qsets]
        # two set-ups in flight: a 5 Mb megablast batch takes ~25 ms to set up, a 50 Gbp pass ~16 ms
        setup_pool = ThreadPoolExecutor(max_workers=2, initializer=lambda: torch.cuda.set_device(dev))
        make = lambda k: api.BlastPrelimSearch(qsets[k % nbatch], opt, src)

        def stream(count):
            prev, futs = None, []
            ahead = [setup_pool.submit(make, k) for k in range(min(2, count))]
            for k in range(count):
                b = ahead.pop(0).result()
                if k + 2 < count:
                    ahead.append(setup_pool.submit(make, k + 2))
                b.begin()
                if prev is not None:
                    out = prev.end(); futs.append(merger.submit(merge, prev, out)); prev.close()
                prev = b
            out = prev.end(); futs.append(merger.submit(merge, prev, out)); prev.close()
            return sum(f.result() for f in futs)
        stream(3); sync()
        ts = time.perf_counter()
        stream(args.stream_steps); sync()
        el = time.perf_counter() - ts
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        streamed = {"ms_per_pass": el / args.stream_steps * 1e3,
                    "value": world * nsub * slen * args.stream_steps / el / 1e9, "unit": "Gbp/s", "passes": args.stream_steps,
                    "what": "every pass builds its query batch from scratch (host set-up + lookup tables on the device), two set-ups in flight on worker threads underneath the passes; the first set-up is inside the timed region"}

    # ---- roofline of the dominant kernel (scan+seed), from HIP events in the library ----
    scan_ms = sum(b.diagnostics.scan_kernel_ms for b in batches)
    launches = sum(b.diagnostics.scan_launches for b in batches)
    scanned = sum(b.diagnostics.subject_bases_scanned for b in batches)
    seeds = sum(b.diagnostics.seeds for b in batches)
    lookup_hits = sum(b.diagnostics.lookup_hits for b in batches)
    algo_bytes = 0.25 * scanned
    bin_ms = sum(b.diagnostics.bin_kernel_ms for b in batches)
    probe_ms = sum(b.diagnostics.probe_kernel_ms for b in batches)
    rare_ms = sum(b.diagnostics.rare_kernel_ms for b in batches)
    # dominant kernel: the binning kernel when the partitioned scan is used, else the direct scan
    dom_name, dom_ms = ("scan_bin_kernel", bin_ms) if bin_ms > 0 else ("scan_seed_kernel", scan_ms)
    # the binning kernel has stride-specialised variants; this is the name rocprof shows
    dom_label = dom_name + ("_s%d" % info["scan_step"] if bin_ms > 0 and info["scan_step"] in (1, 2, 17, 18) else "")
    achieved = algo_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    stage_achieved = algo_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    traffic = None
    tf = os.path.join(ROOT, "profiles", "scan_traffic.json")
    # the committed PMC passes were taken on the default workload (C2, full shard); other shapes: null
    if os.path.exists(tf) and args.workload == "C2" and abs(algo_bytes / max(launches, 1) - 12.5e9) < 1e6:
        try:
            traffic = json.load(open(tf)).get(dom_name, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.cpu_seconds > 0:
        cpu = cpu_baseline(args, queries[:args.batch_queries], opt, mine)

    if rank == 0:
        value = total_bases_global * args.steps / elapsed / 1e9
        line = {
            "metric": "subject Gbp scanned/sec (%s preliminary search, DB bases x passes / wall)" % task,
            "value": value, "unit": "Gbp/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 (2-bit packed bases, int32 scores)", "data": "synthetic",
            "config": {
                "workload": "%s: %d x 1 kb queries vs %.1f Gbp synthetic 2-bit DB per GPU, %s W=%d"
                            % (args.workload, len(queries), nsub * slen / 1e9, task, opt.word_size),
                "stage_ms_per_pass": {k: sum(getattr(b.diagnostics, k) for b in batches) / max(launches, 1)
                                      for k in ["scan_stage_ms", "seed_stage_ms", "gapped_stage_ms", "host_stage_ms"]},
                "batch_setup_ms": batch_setup_ms, "streamed": streamed, "init_hits_per_pass": sum(b.diagnostics.good_init_extends for b in batches) / max(launches, 1),
                "batch_plan": {"queries_per_batch": args.batch_queries, "passes_per_config": npass_config,
                               "lut": info["lut_width"], "scan_step": info["scan_step"],
                               "lut_type": info["lut_type"], "diag_container": info["container"]},
                "subjects_per_gpu": nsub, "subject_len": slen,
                "parallelism": "db-shard x%d (volumes by rank, RCCL gather of HSP records)" % world,
                "binning_reused_across_batches": bool(args.reuse_binning),
                "pipeline": "off" if (nbatch < 2 or args.no_overlap) else
                            "gapped stage + merge of pass k overlap the scan of pass k+1 (second HIP stream + host thread)",
                "hsps_per_pass": nhsp / max(args.steps, 1),
                "seeds_per_pass": seeds / max(launches, 1),
                "lookup_hits_per_pass": lookup_hits / max(launches, 1),
            },
            "roofline": {"bound": "hbm", "kernel": dom_label,
                         "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic,
                         "algorithmic_bytes_per_launch": algo_bytes / max(launches, 1),
                         "avg_launch_ms": dom_ms / max(launches, 1), "launches": launches,
                         "scan_stage": {"kernels": dom_label + " + probe_bin_kernel + probe_rare_kernel"
                                        if bin_ms > 0 else "scan_seed_kernel",
                                        "avg_ms": scan_ms / max(launches, 1),
                                        "avg_ms_by_kernel": [bin_ms / max(launches, 1), probe_ms / max(launches, 1),
                                                             rare_ms / max(launches, 1)],
                                        "achieved": stage_achieved, "frac": stage_achieved / 8000.0}},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


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
    """The oracle (a scalar C port of the reference algorithm) on this box's host cores: the same
    query batch, a bounded sample of the same shard's subjects, one process per core (the reference
    shares OID chunks among threads the same way, x_LaunchMultiThreadedSearch)."""
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
            "sample": "%d subjects (%.0f Mbp) of the rank-0 shard spread over %d processes, one 5 Mb query batch, %.1f s each"
                      % (done // layout.length, done / 1e6, cores, t)}


if __name__ == "__main__":
    main()
