"""bench_side.py -- what bench.py measures BESIDE its timed region: the other single-GPU configs as side runs (C3, C4, the shim-shaped
loop, the documented invocation, the repeat-rich database), the C4 and shim workloads themselves, the VALU roofline of the blastn shape,
and the CPU baseline (the oracle timed on the host cores: the only place besides tests/ and smoke() that touches oracle/)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
BENCH = os.path.join(ROOT, "bench.py")


def valu_roofline(workload, ms_per_step, launches_per_step):
    """VALU issue roofline of a whole step: wave-instructions per step (SQ_INSTS_VALU summed over the kernels of the committed
    rocprofv3 --pmc pass named in profiles/valu_counts.json -- a constant of that profile, refreshed every round, not a counter of
    this run) against ONE peak: 1,024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md: a wave64 VALU op
    issues over two cycles) = 1.2288e12 wave-instructions per second.  Beside the fraction, as explanation and not as other
    denominators: what tools/valu_microbench.hip measures per instruction class on this part (profiles/r04_valu_microbench.txt: add /
    sub / and / or / xor / mov / lshr every 2.4 cycles, everything else integer -- shifts left, bfe, alignbit, perm, cmp, min / max,
    cndmask, three-operand forms, DPP -- every 4.4)."""
    tf = os.path.join(ROOT, "profiles", "valu_counts.json")
    if not os.path.exists(tf):
        return None
    try:
        e = json.load(open(tf)).get(workload)
        if not e:
            return None
        n = float(e["valu_wave_instructions_per_launch"]) * launches_per_step
        peak = 1024 * 2.4e9 / 2.0
        return {"bound": "valu issue", "wave_instructions_per_step": n, "source": e.get("source"),
                "peak_wave_instructions_per_s": peak, "peak_what": "1,024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md)",
                "frac": n / (peak * ms_per_step * 1e-3),
                "measured_issue_cycles_by_class": {"add/sub/and/or/xor/mov/lshr": 2.4, "other integer (shl, bfe, alignbit, perm, cmp, min/max, cndmask, 3-operand, DPP)": 4.4,
                                                   "source": "tools/valu_microbench.hip, profiles/r04_valu_microbench.txt -- why the step cannot reach frac 1: its instruction mix issues at 4.0 cycles on average"},
                "by_kernel_per_launch": e.get("by_kernel")}
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
    for wl, steps in (("C3", "32"), ("C4", "80"), ("shim", "10"), ("cli", "3"), ("skew", "20")):
        # C3 carries a CPU baseline of its own (the oracle on the same 100-query batch, ~6 s per host core); C4's preliminary
        # search is C2's -- its baseline is the C2 line's
        cmd = [sys.executable, BENCH, "--workload", wl, "--steps", steps, "--warmup", "2",
               "--engine-steps", "0", "--min-seconds", "1.0", "--side"] + (["--cpu-seconds", "6"] if wl == "C3" else ["--no-cpu-baseline"])
        if wl == "skew":        # the C2 step over a repeat-rich shard (profiles/r06_skew.txt), the library's default policy
            cmd = [sys.executable, BENCH, "--skew", "--record-cache", "on", "--steps", steps, "--warmup", "6", "--engine-steps", "0", "--min-seconds", "0.5",
                   "--side", "--no-cpu-baseline", "--no-side-workloads"]
        env = dict(os.environ); env["HIP_VISIBLE_DEVICES"] = env.get("HIP_VISIBLE_DEVICES", "")
        if not env["HIP_VISIBLE_DEVICES"]:
            del env["HIP_VISIBLE_DEVICES"]
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
            j = json.loads(p.stdout.strip().splitlines()[-1])
            if wl in ("shim", "cli"):
                out[wl] = j
                continue
            if wl == "skew":
                out[wl] = {"workload": j["config"]["workload"] + " with gbn_synth_skew over it", "ms_per_step": j["ms_per_step"], "steps": j["steps"],
                           "uniform_step_is": "this line's cached_pass (record cache on)", "skew": j["config"].get("skew"),
                           "hsps_per_pass": j["config"].get("hsps_per_pass"), "gpu_ms_per_launch_by_kernel": j["roofline"].get("gpu_ms_per_launch_by_kernel"),
                           "scan_kernels_ms": j["roofline"]["scan_stage"]["avg_ms_by_kernel"], "command": "python bench.py --skew --record-cache on --steps %s --warmup 6" % steps,
                           "warmup_what": "six passes: the set's one-time work -- binned again with per-bin stream capacities when the first pass's streams overflow (a new 25 GB "
                                          "buffer: up to 2 s when the driver has to hand out memory just freed), sorted by cell at its second hit -- is over before the timed "
                                          "regions; with two, one run in three had it inside (17.9 / 24.6 / 112 ms per step)"}
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
    if args.trace_threads <= 0:
        # (the CPUs this process is granted, not the hardware threads it sees: 16 of 256 on the GPU boxes.  Measured there: 8 threads
        # 5.6 ms per batch, 16 threads 4.0 -- a 5,000-query batch's 2,469 alignments are some 50 ms of CPU)
        args.trace_threads = max(4, min(16, api.host_cpus()))
    window = int(os.environ.get("GBN_C4_WINDOW", "8"))          # batches the caller keeps in the pipeline

    def run(count):
        pipe = api.SearchPipeline(opt, src, trace_threads=args.trace_threads, traceback=not args.no_traceback, overlap=not args.no_overlap)
        sub = got = 0; nfinal = 0; diags = []
        while got < count:
            while sub < count and sub < got + window:
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
    if os.environ.get("GBN_CPU_ACCOUNT") and rank == 0:
        # CPU time of the library's host threads by what they did (csrc/gbn_host.hpp GBN_CPU_*), per batch, next to the process's total
        import ctypes as C
        L = api.lib(); ms = (C.c_double * 16)(); L.gbn_debug_cpu_account.argtypes = [C.POINTER(C.c_double), C.c_int]
        n = L.gbn_debug_cpu_account(ms, 16)
        names = ["set-up (calling thread)", "set-up (pool workers)", "search thread (begin: scan stage)", "extension stage thread", "host replay",
                 "end + collector", "traceback (calling thread)", "traceback (workers)", "submit (copy of the queries)", "  of the workers: unpacking the subject stretches", "  of the workers: start points", "  of the workers: alignment (incl. what follows in the loop)", "  of the workers: rescoring, identities, containment (to the end of the list)", "  of the calling thread: per-query order and output"]
        batches = max(args.warmup, 2) + args.steps * len(regions)
        tot = time.process_time() * 1e3
        print("[cpu account] %d batches; process CPU %.0f ms in all (incl. start-up)" % (batches, tot), file=sys.stderr)
        for i in range(n):
            print("[cpu account]   %-36s %8.2f ms per batch" % (names[i], ms[i] / batches), file=sys.stderr)
    # final alignments of one batch, counted once outside the timed region
    pipe = api.SearchPipeline(opt, src, trace_threads=args.trace_threads, traceback=True, overlap=False)
    pipe.submit(qsets[0]); pipe.finish(); _, res, _ = pipe.next(); pipe.close()
    rec = res[0]
    scan_ms = sum(d.scan_kernel_ms for d in diags); launches = sum(d.scan_launches for d in diags)
    bin_ms = sum(d.bin_kernel_ms for d in diags); probe_ms = sum(d.probe_kernel_ms for d in diags); rare_ms = sum(d.rare_kernel_ms for d in diags)
    scanned = sum(d.subject_bases_scanned for d in diags)
    algo = 0.25 * scanned
    cache = api.record_cache_stats()
    # (the kernel that went over the cached records: probe_runs_kernel once the set is sorted by cell -- from its second hit on)
    probe_name = "probe_runs_kernel" if cache.get("sorted_passes", 0) > 0 else "probe_bin_kernel"
    if probe_name == "probe_runs_kernel":
        # (the pass over sorted records reads the runs of the occupied cells of a resident index, not the subject: its roofline is the bytes
        # it moves -- profiles/scan_traffic.json, a constant of profiles/r06c_cached_pmc.csv -- over its time; bench.py cached_roofline)
        try:
            tj = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "scan_traffic.json")))
            algo = float(tj["probe_runs_kernel"]["hbm_bytes_per_launch"]) * launches
        except Exception:
            algo = 5.75e9 * launches
    by_kernel = {"scan_bin_kernel_s17": bin_ms / max(launches, 1), probe_name: probe_ms / max(launches, 1), "probe_rare_kernel": rare_ms / max(launches, 1)}
    for i, name in enumerate(api.GbnDiagnostics.KERNEL_CLASSES):
        t = sum(d.kernel_ms[i] for d in diags) / max(launches, 1)
        if t > 0:
            by_kernel[name] = t
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
                                        "probe + rare kernel only) -- %d passes served from the cache (%d of them over records sorted by cell), %d binned, %.1f GB of records resident"
                                        % (cache["hits"], cache.get("sorted_passes", 0), cache["misses"], cache["bytes"] / 1e9)) if cache["limit"] > 0 else "off (--record-cache off): every batch bins",
                       "stage_ms_per_pass": {k: sum(getattr(d, k) for d in diags) / max(launches, 1)
                                             for k in ["scan_stage_ms", "seed_stage_ms", "gapped_stage_ms", "host_stage_ms"]},
                       "final_hsps_per_batch": int(len(rec)), "final_identity_mean": float((rec["num_ident"] / np.maximum(rec["align_length"], 1)).mean()) if len(rec) else None,
                       "gapped_alignments_per_batch": int((rec["gaps"] > 0).sum()) if len(rec) else 0},
            "roofline": {"bound": "hbm", "kernel": probe_name if cache["limit"] > 0 else "scan_bin_kernel_s17",
                         "achieved": algo / ((probe_ms if cache["limit"] > 0 else bin_ms) * 1e-3) / 1e9 if bin_ms + probe_ms else 0.0, "peak": 8000.0, "unit": "GB/s",
                         "frac": (algo / ((probe_ms if cache["limit"] > 0 else bin_ms) * 1e-3) / 1e9 / 8000.0) if bin_ms + probe_ms else 0.0, "traffic": None,
                         "achieved_is": ("bytes probe_runs_kernel moves per launch (profiles/scan_traffic.json) / its time" if probe_name == "probe_runs_kernel"
                                         else "0.25 B per subject base / the kernel's time"),
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
    # one process per CPU this job is granted (gbn_host_cpus: hardware threads cut down to affinity and cgroup quota -- the GPU boxes
    # show 256 hardware threads and grant 16 CPUs; rounds 1-5 started 64 processes there, which the quota ran as sixteen)
    cores = args.cpu_cores if args.cpu_cores > 0 else max(1, min(64, api.host_cpus()))
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

