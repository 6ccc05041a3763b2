"""GPU: the three kernel launchers driven through the C ABI (gbn_launch_scan_seed / _ungapped / _gapped with the
parameter blocks of include/gblastn_amd_kernels.h, device buffers owned by this test), checked against the
DEFINITIONS of tests/test_oracle_definitions.py (brute-force seeds, whole-matrix X-drop DP) and against the oracle;
plus the shim-side pieces of the boundary: gbn_prelim_search_lists, the shard builder and the shard cache."""
import ctypes as C
import numpy as np
import pytest
import torch
from gblastn_amd import api
from oracle import orc
from tests import util
from tests.test_oracle_definitions import brute_force_seeds, seed_case, dp_pairs, dp_by_definition

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def scan_through_the_launcher(ps, src, lens):
    """every subject of the shard tiled by the rule of gblastn_amd_kernels.h -> seeds (DEV_SEED_DT, launch order)"""
    L = api.lib()
    P = api.GbnScanParams()
    api._check(L.gbn_batch_scan_params(ps._b, src._h, C.byref(P)))
    offs, _ = api.layout_slab(lens)
    tiles = []
    for s, n in enumerate(lens):
        if n < P.lut:
            continue
        npos = (n - P.lut) // P.step + 1
        for p in range(0, npos, api.GBN_TILE_POS):
            tiles.append((s, p * P.step, min(api.GBN_TILE_POS, npos - p), offs[s] >> 4))
    t = dev(np.array(tiles, dtype=api.TILE_DT).view(np.int32))
    cap = 1 << 20
    seeds = torch.zeros(cap * 4, dtype=torch.int32, device="cuda")
    ctr = torch.zeros(2, dtype=torch.int64, device="cuda")
    P.tiles = t.data_ptr(); P.ntiles = len(tiles)
    P.seeds = seeds.data_ptr(); P.seed_cap = cap; P.seed_count = ctr.data_ptr(); P.raw_hits = ctr.data_ptr() + 8
    api._check(L.gbn_launch_scan_seed(C.byref(P), min(len(tiles), 2048), stream()))
    torch.cuda.synchronize()
    n, raw = (int(x) for x in ctr.cpu())
    assert 0 < n <= cap and raw >= n
    return seeds.cpu().numpy().view(api.DEV_SEED_DT)[:n].copy(), seeds, P


def scan_order(sd, descending):
    return np.lexsort((-sd["q_pos"] if descending else sd["q_pos"], sd["s_scan"], sd["subj"]))


@pytest.mark.parametrize("nq,lut,step", [(16, 11, 18), (160, 12, 17)])
def test_scan_launcher_seeds_equal_the_definition(nq, lut, step):
    sub, queries = seed_case(nq, 40 + nq)
    sub2 = np.random.default_rng(3).integers(0, 4, 9000).astype(np.uint8)
    sub2[4000:4700] = queries[0][150:850] & 3          # (the query's ambiguity code becomes a base: extensions stop there)
    subjects = [sub, sub2]
    opt = api.default_options("megablast", db_length=10**7, db_num_seqs=10)
    src = api.BlastSeqSrc.from_packed([(orc.pack_ncbi2na(s), len(s)) for s in subjects])
    ps = api.BlastPrelimSearch(queries, opt, src)
    assert (ps.info()["lut_width"], ps.info()["scan_step"]) == (lut, step)
    sd, _, _ = scan_through_the_launcher(ps, src, [len(s) for s in subjects])
    sd = sd[scan_order(sd, True)]
    S = orc.Search(util.oracle_options(opt), queries)
    qcat = S.query_concat()
    for k, s in enumerate(subjects):
        want = brute_force_seeds(qcat, S.contexts, s, 28, lut, step, descending=True)
        g = sd[sd["subj"] == k]
        got = list(zip((g["q_pos"] - g["ext_left"]).tolist(), (g["s_scan"] - g["ext_left"]).tolist()))
        assert got == want and len(want) > 0
    # the whole search reports the same seeds (keep_stages)
    r = ps.run(keep_stages=True)
    assert len(r["seeds"]) == len(sd)


@pytest.mark.parametrize("task,nq", [("megablast", 2), ("megablast", 200), ("blastn", 4)])
def test_ungapped_launcher_equals_the_oracle(task, nq):
    """seeds from the scan launcher, ordered here as gblastn_amd_kernels.h prescribes, through gbn_launch_ungapped:
    the initial hits equal the oracle's word finder (diagonal array for few queries, hash container for many)"""
    L = api.lib()
    db, queries, plants, subjects, opt = util.small_case(6, 30000, nq, qlen=600, seed=21, task=task)
    src = api.BlastSeqSrc.from_packed(subjects)
    ps = api.BlastPrelimSearch(queries, opt, src)
    sd, d_seeds, _ = scan_through_the_launcher(ps, src, [n for _, n in subjects])
    hashc, diag_len, desc = C.c_int32(), C.c_int32(), C.c_int32()
    api._check(L.gbn_batch_diag_layout(ps._b, C.byref(hashc), C.byref(diag_len), C.byref(desc)))
    assert hashc.value == (1 if nq == 200 else 0)       # CORE/blast_extend.c:85-140: hash container from 8000 query bases up
    order = scan_order(sd, bool(desc.value))
    o = sd[order]
    q0, s0 = o["q_pos"] - o["ext_left"], o["s_scan"] - o["ext_left"]
    slot = ((s0 - q0) & 511) if hashc.value else ((s0 + diag_len.value - q0) & (diag_len.value - 1))
    gbits = 9 if hashc.value else int(diag_len.value).bit_length() - 1
    key = (o["subj"].astype(np.uint64) << np.uint64(gbits)) | slot.astype(np.uint64)
    grp = np.argsort(key, kind="stable")
    n = len(sd)
    X = api.GbnExtParams()
    api._check(L.gbn_batch_ext_params(ps._b, src._h, C.byref(X)))
    idx = dev(order[grp].astype(np.uint32).view(np.int32)); kg = dev(key[grp].view(np.int64))
    cd = torch.zeros(n, dtype=torch.int32, device="cuda"); cl = torch.zeros(n, dtype=torch.int32, device="cuda")
    rh = torch.zeros(n + 1, dtype=torch.int32, device="cuda"); ctr = torch.zeros(2, dtype=torch.int64, device="cuda")
    cap = n
    ih = torch.zeros(cap * 8, dtype=torch.int32, device="cuda")
    X.seeds = d_seeds.data_ptr(); X.idx = idx.data_ptr(); X.key_group = kg.data_ptr(); X.n = n
    X.cell_diag = cd.data_ptr(); X.cell_level = cl.data_ptr(); X.run_heads = rh.data_ptr(); X.run_count = ctr.data_ptr() + 8
    X.group_bits = gbits
    X.ihits = ih.data_ptr(); X.ihit_count = ctr.data_ptr(); X.ihit_cap = cap
    api._check(L.gbn_launch_ungapped(C.byref(X), stream()))
    torch.cuda.synchronize()
    nih = int(ctr.cpu()[0])
    got = ih.cpu().numpy().view(api.DEV_IHIT_DT)[:nih]
    ora, _ = util.oracle_run(opt, queries, subjects)
    total = 0
    for k, r in enumerate(ora):
        g = got[got["subj"] == k]
        g = g[np.lexsort((g["seq"], g["q_start"], -g["length"], g["s_start"], -g["score"]))]    # Blast_InitHitListSortByScore
        w = r["init_hits"]
        assert len(g) == len(w), k
        for f in ("q_off", "s_off", "q_start", "s_start", "length", "score"):
            assert np.array_equal(g[f], w[f]), (k, f)
        total += len(w)
    assert total > 0


def gapped_through_the_launcher(queries, subjects, hits, opt, greedy):
    """hits: (pair index, q offset inside its query, s offset); one query / one subject per pair"""
    L = api.lib()
    src = api.BlastSeqSrc.from_packed([(orc.pack_ncbi2na(s), len(s)) for s in subjects])
    ps = api.BlastPrelimSearch(queries, opt, src)
    ctx = ps.contexts
    G = api.GbnGapParams()
    api._check(L.gbn_batch_gap_params(ps._b, src._h, C.byref(G)))
    ih = np.zeros(len(hits), dtype=api.DEV_IHIT_DT)
    for i, (p, qo, so) in enumerate(hits):
        ih[i] = (p, ctx[2 * p].query_offset + qo, so, ctx[2 * p].query_offset + qo, so, 4, 8, i)
    d_ih = dev(ih.view(np.int32)); out = torch.zeros(len(hits) * 8, dtype=torch.int32, device="cuda")
    blocks = (len(hits) + 63) // 64
    scratch = torch.zeros(blocks * 64 * G.scratch_per_thread, dtype=torch.int32, device="cuda")
    G.ihits = d_ih.data_ptr(); G.first = 0; G.n = len(hits); G.out = out.data_ptr(); G.scratch = scratch.data_ptr(); G.max_blocks = blocks
    api._check(L.gbn_launch_gapped(C.byref(G), 1 if greedy else 0, stream()))
    torch.cuda.synchronize()
    return out.cpu().numpy().view(api.DEV_GAPPED_DT), G


@pytest.mark.parametrize("X_bits,go,ge", [(30, 5, 2), (12, 5, 2), (30, 2, 1), (30, 2, 2)])
def test_gapped_launcher_dp_equals_the_matrix_definition(X_bits, go, ge):
    """>= 200 pairs through dynprog_wave_kernel (+ the scratch kernel for the ones it leaves);
    each extension equals the whole-matrix X-drop definition and the oracle's packed DP"""
    pairs = dp_pairs(240, 5)
    assert len(pairs) >= 200
    opt = api.default_options("blastn", db_length=10**7, db_num_seqs=10, xdrop_gap_bits=float(X_bits))
    if (go, ge) != (5, 2):
        opt.reward, opt.penalty, opt.gap_open, opt.gap_extend = 1, -2, go, ge
    queries = [p[0] for p in pairs]; subjects = [p[1] for p in pairs]
    hits = [(i, p[2], p[3]) for i, p in enumerate(pairs)]
    got, G = gapped_through_the_launcher(queries, subjects, hits, opt, greedy=False)
    assert not (got["score"] == -2**31 + 1).any()           # GBN_GAP_REDO never leaves the launcher
    long_ones = 0
    for i, (q, s, qo, so) in enumerate(pairs):
        want = dp_by_definition(q, s, qo, so, opt.reward, opt.penalty, G.gap_open, G.gap_extend, G.xdrop)
        o = orc.gapped_extend(q, s, qo, so, G.xdrop, opt.reward, opt.penalty, G.gap_open, G.gap_extend)
        g = dict(q_offset=int(got[i]["q_start"]), q_end=int(got[i]["q_stop"]), s_offset=int(got[i]["s_start"]),
                 s_end=int(got[i]["s_stop"]), score=int(got[i]["score"]))
        assert g == want == o, i
        long_ones += g["q_end"] - g["q_offset"] > 60
    assert long_ones >= 50


def test_gapped_launcher_greedy_equals_the_oracle():
    pairs = dp_pairs(220, 6)
    for kw in (dict(), dict(reward=1, penalty=-2, gap_open=2, gap_extend=2)):
        opt = api.default_options("megablast", db_length=10**7, db_num_seqs=10, **kw)
        queries = [p[0] for p in pairs]; subjects = [p[1] for p in pairs]
        got, G = gapped_through_the_launcher(queries, subjects, [(i, p[2], p[3]) for i, p in enumerate(pairs)], opt, greedy=True)
        for i, (q, s, qo, so) in enumerate(pairs):
            o = orc.gapped_extend(q, s, qo, so, G.xdrop, opt.reward, opt.penalty, G.gap_open, G.gap_extend, greedy=True)
            g = dict(q_offset=int(got[i]["q_start"]), q_end=int(got[i]["q_stop"]), s_offset=int(got[i]["s_start"]),
                     s_end=int(got[i]["s_stop"]), score=int(got[i]["score"]))
            assert g == o, (i, kw)


def test_search_lists_shard_builder_and_cache():
    """the library side of the shim: subjects handed over one by one, the shard kept per caller handle, HSPs delivered
    as one list per subject -- equal to gbn_prelim_search on a shard made the usual way"""
    L = api.lib()
    db, queries, plants, subjects, opt = util.small_case(12, 20000, 8, qlen=700, seed=5)
    sb = C.c_void_p()
    api._check(L.gbn_shard_builder_new(C.byref(sb), len(subjects)))
    for p, n in subjects:
        a = np.ascontiguousarray(p[:(n + 3) // 4])
        api._check(L.gbn_shard_builder_add(sb, a.ctypes.data, n))
    h = C.c_void_p()
    api._check(L.gbn_shard_builder_finish(sb, C.byref(h)))
    L.gbn_shard_builder_free(sb)
    key = C.c_void_p(0xBEEF0)
    assert L.gbn_db_cache_find(key) is None
    api._check(L.gbn_db_cache_insert(key, h))
    assert L.gbn_db_cache_insert(key, h) != 0
    assert L.gbn_db_cache_find(key) == h.value

    ps = api.BlastPrelimSearch(queries, opt)
    lists = []

    @api.GbnHspListFn
    def sink(arg, oid, hsps, n):
        a = np.ctypeslib.as_array(C.cast(hsps, C.POINTER(C.c_uint8)), shape=(n * api.HSP_DT.itemsize,)).view(api.HSP_DT).copy()
        lists.append((oid, a))
        return 0
    d = api.GbnDiagnostics()
    api._check(L.gbn_prelim_search_lists(ps._b, h, sink, None, C.byref(d), None, None))
    want = api.BlastPrelimSearch(queries, opt, api.BlastSeqSrc.from_packed(subjects)).run()["hsps"]
    assert len(want) > 0 and [o for o, _ in lists] == sorted(set(want["oid"].tolist()))
    got = np.concatenate([a for _, a in lists])
    assert got.tobytes() == want.tobytes()
    for oid, a in lists:
        assert (a["oid"] == oid).all() and (np.diff(a["score"]) <= 0).all()

    @api.GbnHspListFn
    def failing(arg, oid, hsps, n):
        return 1
    assert L.gbn_prelim_search_lists(ps._b, h, failing, None, None, None, None) != 0
    L.gbn_release_db_memory()                     # frees the cached shard
    assert L.gbn_db_cache_find(key) is None


@pytest.mark.parametrize("task,nq,seed", [("megablast", 24, 3), ("blastn", 9, 5), ("blastn", 6, 8)])
def test_hash_path_of_the_hip_kernels_equals_the_definitions(task, nq, seed):
    """the checkers of tests/hash_path_definitions.py (exact per-diagonal map + prefix-sum ungapped extension; brute-force
    containment) against what the HIP path delivers: seed list, initial-hit list, HSP list, extension counter -- with the
    engine's own kernel choice and with the two-kernel seed stage (seed_ext_kernel + diag_replay_kernel) forced"""
    from oracle import orc
    from tests.test_hash_path_oracle import hash_case, check_filter_and_extension, check_acceptance
    sub, queries, oopt, S = hash_case(task, nq, seed)
    kw = dict(task="blastn", word_size=11) if task == "blastn" else {}
    gopt = api.default_options(task if task == "megablast" else "blastn", db_length=10**7, db_num_seqs=10, **({"word_size": 11} if task == "blastn" else {}))
    packed = orc.pack_ncbi2na(sub)
    src = api.BlastSeqSrc.from_packed([(packed, len(sub))])
    ps = api.BlastPrelimSearch(queries, gopt, src)
    assert ps.info()["container"] == 1
    g = ps.run(keep_stages=True)
    seeds = list(zip(g["seeds"]["q_off"].tolist(), g["seeds"]["s_off"].tolist()))
    nwin, want = check_filter_and_extension(S, oopt, sub, g["init_hits"])            # seeds by brute force
    check_filter_and_extension(S, oopt, sub, g["init_hits"], seeds=seeds)            # ... and from the kernel's own seed list
    assert len(want) >= 3
    nacc, nshared = check_acceptance(S, oopt, sub, g["init_hits"], g["hsps"], got_extensions=int(ps.diagnostics.gapped_extensions))
    assert nacc >= 2


def test_shard_with_explicit_oids_and_holes():
    """what the shim builds from a BlastSeqSrc iterator over an OID list: the HSPs carry the OIDs that were given, the
    traceback stage finds its subjects by them"""
    L = api.lib()
    db, queries, plants, subjects, opt = util.small_case(10, 20000, 8, qlen=700, seed=11)
    oids = [3, 4, 9, 20, 21, 22, 40, 41, 77, 100]
    sb = C.c_void_p()
    api._check(L.gbn_shard_builder_new(C.byref(sb), len(subjects)))
    for (p, n), oid in zip(subjects, oids):
        a = np.ascontiguousarray(p[:(n + 3) // 4])
        api._check(L.gbn_shard_builder_add_oid(sb, oid, a.ctypes.data, n))
    h = C.c_void_p()
    api._check(L.gbn_shard_builder_finish(sb, C.byref(h)))
    L.gbn_shard_builder_free(sb)
    src = api.BlastSeqSrc(h)
    assert L.gbn_db_device(h) == L.gbn_current_device() >= 0
    got = api.BlastPrelimSearch(queries, opt, src).run()["hsps"]
    want = api.BlastPrelimSearch(queries, opt, api.BlastSeqSrc.from_packed(subjects)).run()["hsps"].copy()
    assert len(want) > 0
    want["oid"] = np.array(oids, dtype=np.int32)[want["oid"]]
    assert got.tobytes() == want.tobytes()
    # traceback over the same shard: subjects are looked up by OID
    ps = api.BlastPrelimSearch(queries, opt, src)
    col = api.BlastHSPCollector(len(queries), opt.hitlist_size)
    col.write(ps.run()["hsps"])
    kept, starts, _ = col.close()
    rec, ops, qs = api.BlastTracebackSearch(ps, src).run(kept, starts)
    ref_src = api.BlastSeqSrc.from_packed(subjects)
    ps2 = api.BlastPrelimSearch(queries, opt, ref_src)
    col2 = api.BlastHSPCollector(len(queries), opt.hitlist_size)
    col2.write(ps2.run()["hsps"])
    kept2, starts2, _ = col2.close()
    rec2, ops2, qs2 = api.BlastTracebackSearch(ps2, ref_src).run(kept2, starts2)
    assert len(rec) == len(rec2) > 0 and ops == ops2
    assert rec["oid"].tolist() == [oids[o] for o in rec2["oid"].tolist()]
    assert rec["score"].tolist() == rec2["score"].tolist()
    src.close(); ref_src.close()


def test_two_host_threads_search_two_shards_concurrently():
    """the re-entrant contract of the reference's entry point (N search threads, API/prelim_search_runner.hpp:135-166):
    two threads, each with a shard and batches of its own, in flight at the same time == the same searches one after
    the other.  (One GPU here: both threads lease device 0 and the engine serialises them call by call; with two
    GPUs each thread calls gbn_use_device for its own.)"""
    import threading
    L = api.lib()
    cases = [util.small_case(12, 60000, 10, qlen=800, seed=21), util.small_case(9, 80000, 12, qlen=600, seed=22, task="blastn", word_size=11)]
    serial = []
    for db, queries, plants, subjects, opt in cases:
        src = api.BlastSeqSrc.from_packed(subjects)
        serial.append(api.BlastPrelimSearch(queries, opt, src).run()["hsps"].tobytes())
        src.close()
    assert all(len(x) > 0 for x in serial)
    out = [None, None]; err = []

    def work(k):
        try:
            api._check(L.gbn_use_device(L.gbn_current_device()))
            db, queries, plants, subjects, opt = cases[k]
            src = api.BlastSeqSrc.from_packed(subjects)
            res = []
            for it in range(6):
                ps = api.BlastPrelimSearch(queries, opt, src)
                if it % 2: ps.begin(); res.append(ps.end()["hsps"].tobytes())
                else: res.append(ps.run()["hsps"].tobytes())
            out[k] = res
            src.close()
        except Exception as e:      # noqa
            err.append(repr(e))
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    assert not err, err
    for k in range(2):
        assert all(r == serial[k] for r in out[k]), k


def test_parity_suite_through_the_two_kernel_seed_stage():
    """seed_ext_kernel + diag_replay_kernel take over from GBN_DIAG_COMPACT_MIN seeds per launch (2^20): the parity
    cases are far smaller, so the whole stage-by-stage comparison with the oracle runs once more with the threshold
    at 1 (every launch goes through the two kernels: hash and array containers, masked queries, short words)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ); env["GBN_DIAG_COMPACT_MIN"] = "1"
    # (the tests that start child processes of their own stay out: they carry their own two-kernel cases)
    p = util.run_child([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py",
                        "tests/test_gpu_definitions.py::test_hash_path_of_the_hip_kernels_equals_the_definitions",
                        "-x", "-q", "-m", "gpu and not spawns"], cwd=root, env=env, timeout=900)
    assert " passed" in p.stdout and "failed" not in p.stdout


def test_slice_scan_cases_in_the_forms_before_the_ordered_one():
    """Tables of more than one slice of presence bits (lut 11, 12 as wide as the word) are scanned through the folded filter
    with the seeds in scan order (scan_fold_ordered_kernel).  GBN_SCAN_ORDERED=0: folded, seeds in no particular order
    (scan_fold_kernel, the full composite-key sort behind it); GBN_SLICE_FOLD=0: a slice per pass (scan_slice_kernel).  The
    blastn cases of the parity suite once more through each, stage by stage against the oracle -- with the threshold of
    the composite-key stage at 1 in a third leg, so that the small cases take the shortened sort as well"""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for setting in ({"GBN_SCAN_ORDERED": "0"}, {"GBN_SLICE_FOLD": "0"}, {"GBN_DIAG_COMPACT_MIN": "1"}):
        env = dict(os.environ); env.update(setting)
        p = util.run_child([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-x", "-q", "-m", "gpu and not spawns",
                            "-k", "slice or lut11_stride1 or word12 or randomised or ragged"], cwd=root, env=env, timeout=900)
        assert " passed" in p.stdout and "failed" not in p.stdout, setting


def test_parity_cases_with_poisoned_device_blocks():
    """GBN_POISON=<byte>: every device block the engine's pool hands out is filled with that byte first.  The randomised
    shapes, the ragged inputs and the pipelined cases once more with it: a kernel that reads memory nobody wrote would
    depend on what the device memory held before."""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for byte, compact in (("165", None), ("255", "1")):
        env = dict(os.environ); env["GBN_POISON"] = byte
        if compact: env["GBN_DIAG_COMPACT_MIN"] = compact
        p = util.run_child([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-x", "-q", "-m", "gpu and not spawns",
                            "-k", "randomised or ragged or begin_end or known_answers"], cwd=root, env=env, timeout=900)
        assert " passed" in p.stdout and "failed" not in p.stdout


TIE_CASE = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from gblastn_amd import api
from oracle import orc
from tests import util
rng = np.random.default_rng(8)
R = rng.integers(0, 4, 60).astype(np.uint8)
nq = 12
qs = [rng.integers(0, 4, 900).astype(np.uint8) for _ in range(nq)]
opt = api.default_options(sys.argv[1], db_length=10**7, db_num_seqs=10)
S0 = orc.Search(util.oracle_options(opt), qs)
off = [S0.contexts[2 * k].query_offset for k in range(nq)]
for k in range(nq):                                    # the same 60-mer at concatenated offsets congruent modulo 512
    a = (100 + off[0] - off[k]) %% 512
    if a < 10: a += 512
    if a + 60 > 900: a -= 512
    assert 0 <= a <= 840
    qs[k][a:a + 60] = R
subs = []
for j in range(3):
    s = rng.integers(0, 4, 20000).astype(np.uint8)
    for p in (1000, 7000 + j, 15000): s[p:p + 60] = R
    subs.append(s)
subjects = [(orc.pack_ncbi2na(s), len(s)) for s in subs]
src = api.BlastSeqSrc.from_packed(subjects)
got = api.BlastPrelimSearch(qs, opt, src).run(keep_stages=True)
ora, S = util.oracle_run(opt, qs, [(np.concatenate([p, np.zeros(16, np.uint8)]), n) for p, n in subjects])
assert S.info()["container"] == 1
sd = ora[0]["seeds"]
ties = sum(1 for i in range(1, len(sd)) if sd["s_off"][i] == sd["s_off"][i - 1] and (sd["q_off"][i] - sd["q_off"][i - 1]) %% 512 == 0)
assert ties >= 20, ties                               # seeds of one (subject, slot, scan position): the sort leaves them unordered
util.compare_stages(got, ora)
print("TIES_OK", ties, len(got["init_hits"]), len(got["hsps"]))
'''


@pytest.mark.parametrize("task", ["megablast", "blastn"])
def test_seeds_of_one_slot_and_position_are_ordered_after_the_sort(task):
    """the composite-key sort stops at (subject, slot, scan position); seeds that agree on all three -- the same word
    at query positions congruent modulo the slot count -- are put into table order by seed_ext_kernel"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ); env["GBN_DIAG_COMPACT_MIN"] = "1"
    p = util.run_child([sys.executable, "-c", TIE_CASE % root, task], cwd=root, env=env, timeout=600)
    assert "TIES_OK" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]


BIG_N = r'''
import sys, hashlib, numpy as np, torch
sys.path.insert(0, %r)
from gblastn_amd import api, synth
nsub = 120
api.lib().gbn_init(1, 0)
lay = synth.SynthDb(nsub, 1_000_000, seed=777)
slab = torch.empty(lay.nbytes, dtype=torch.uint8, device="cuda")
api._check(api.lib().gbn_synth_fill(slab.data_ptr(), lay.nbytes, lay.seed, None))
src = api.BlastSeqSrc.from_slab((slab.data_ptr(), lay.nbytes), lay.byte_off, lay.lens, is_device=True, keep=slab)
qs, _ = synth.make_queries(100, lay)
ps = api.BlastPrelimSearch(qs, api.default_options("blastn", db_length=nsub * 10**6, db_num_seqs=nsub), src)
out = []
for k in range(int(sys.argv[1])):
    r = ps.run(keep_stages=True)
    ih = np.sort(r["init_hits"], order=["oid", "q_off", "s_off", "score"])
    out.append((len(r["seeds"]), len(ih), hashlib.sha1(ih.tobytes()).hexdigest(), hashlib.sha1(r["hsps"].tobytes()).hexdigest()))
assert all(o == out[0] for o in out), out
print("BIGN", *out[0])
'''


def test_many_seeds_composite_sort_is_deterministic_and_equals_the_two_sort_path():
    """5.6 million seeds in one launch (blastn shape): the composite-key path (one sort -- of keys that carry their value in
    the low bits, or of (key, value) pairs with GBN_SEED_CKEYS=2 --, seed_ext_kernel + diag_replay_kernel) gives the same
    initial hits and HSPs run after run -- a race here once changed 8 % of the
    exact ungapped extensions from run to run -- and the same as the two-sort path (GBN_SEED_CKEYS=0)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for name, env_add, runs in (("composite", {}, "3"), ("pairs", {"GBN_SEED_CKEYS": "2"}, "1"), ("two sorts", {"GBN_SEED_CKEYS": "0"}, "1")):
        env = dict(os.environ); env.update(env_add)
        p = util.run_child([sys.executable, "-c", BIG_N % root, runs], cwd=root, env=env, timeout=900)
        assert "BIGN" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]
        res[name] = p.stdout.split("BIGN", 1)[1].split()
    assert int(res["composite"][0]) > (1 << 22) and int(res["composite"][1]) > 10000
    assert res["composite"] == res["two sorts"] and res["pairs"] == res["two sorts"]


def test_queue_of_host_replays_keeps_nothing_alive():
    """pipelined blastn passes hand the replay of their gapped extensions to a queue of host tasks, each waiting for its
    predecessor: a finished task must let go of it (and of its copies of the extensions), or the peak resident memory grows
    by megabytes per pass.  60 passes over a 1 Gbp shard (87,000 extensions each): flat after the first."""
    import os, re, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = util.run_child([sys.executable, os.path.join(root, "tools", "rss_check.py"), "60"], cwd=root, timeout=600)
    rss = [float(x) for x in re.findall(r"max RSS (\d+) MB", p.stdout)]
    assert len(rss) == 3 and rss[2] - rss[0] < 100, p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("task,n,kw", [("megablast", 12, {}), ("blastn", 8, {"word_size": 8})])
def test_device_built_tables_of_the_de_bruijn_queries(task, n, kw):
    """The reference's lookup-table known answers (UT/ntlookup_unit_test.cpp:468-556) through the HIP table builder: the
    (n, 4) de Bruijn query holds every n-mer exactly once per strand, so the table is the megablast one with 4^12 cells
    (the standard one with 4^8 for blastn, word 8), and EVERY scan position of any subject finds exactly two entries,
    one per strand -- a count the scan reports.  Results equal the oracle's as everywhere else."""
    from tests.test_oracle_golden import _debruijn_query
    buf, ln = _debruijn_query(n)
    q = buf[1:1 + ln].copy()
    rng = np.random.default_rng(5)
    slen = 400_000 if task == "megablast" else 30_000
    subj = rng.integers(0, 4, slen, dtype=np.uint8)
    subj[1000:1400] = q[5000:5400]
    packed = orc.pack_ncbi2na(subj)
    opt = api.default_options(task, db_length=slen, db_num_seqs=1, **kw)
    ps = api.BlastPrelimSearch([q], opt, api.BlastSeqSrc.from_packed([(packed, slen)]))
    gpu = ps.run(keep_stages=True)
    info = ps.info()
    want = (3, 12, 17) if task == "megablast" else (2, 8, 1)
    assert (info["lut_type"], info["lut_width"], info["scan_step"]) == want
    positions = (slen - info["lut_width"]) // info["scan_step"] + 1
    assert int(ps.diagnostics.lookup_hits) == 2 * positions
    ora, s = util.oracle_run(opt, [q], [(packed, slen)])
    util.compare_stages(gpu, ora)
    assert len(gpu["hsps"]) >= 1


@pytest.mark.gpu
def test_block_cache_under_two_threads_and_changing_chunkings():
    """The shim's resident database (gbn_block_cache_*): N search threads pull OID chunks from ONE shared
    bookmark, as the reference's CPrelimSearchThreads do (API/prelim_search_runner.hpp:135-166), so which thread gets which
    chunk changes from query batch to query batch.  Blocks are keyed by what they hold: the first batch uploads every block
    once, the second -- other threads, other order -- uploads NOTHING, and the lists of both equal one search over the
    whole database."""
    import threading
    L = api.lib()
    L.gbn_release_db_memory()
    nsub, slen, chunk = 23, 30000, 4
    db, queries, plants, subjects, opt = util.small_case(nsub, slen, 10, qlen=800, seed=21)
    want = api.BlastPrelimSearch(queries, opt, api.BlastSeqSrc.from_packed(subjects)).run()["hsps"]
    assert len(want) > 0

    def get_block(oids):
        arr = np.asarray(oids, dtype=np.int32)
        h = C.c_void_p()
        api._check(L.gbn_block_cache_find(b"synthetic_db", arr.ctypes.data, len(arr), C.byref(h)))
        if h.value:
            return h
        sb = C.c_void_p()
        api._check(L.gbn_shard_builder_new(C.byref(sb), len(arr)))
        for o in oids:
            p, n = subjects[o]
            a = np.ascontiguousarray(p[:(n + 3) // 4])
            api._check(L.gbn_shard_builder_add_oid(sb, int(o), a.ctypes.data, n))
        api._check(L.gbn_shard_builder_finish(sb, C.byref(h)))
        L.gbn_shard_builder_free(sb)
        kept = C.c_void_p()
        api._check(L.gbn_block_cache_insert(b"synthetic_db", arr.ctypes.data, len(arr), h, C.byref(kept)))
        return kept

    def one_batch(nthreads, delays):
        bookmark = [0]; mu = threading.Lock(); lists = []; errs = []

        def worker(t):
            try:
                L.gbn_use_device(0)
                ps = api.BlastPrelimSearch(queries, opt)            # the thread's own query batch, as in the shim
                got = []

                @api.GbnHspListFn
                def sink(arg, oid, hsps, n):
                    a = np.ctypeslib.as_array(C.cast(hsps, C.POINTER(C.c_uint8)), shape=(n * api.HSP_DT.itemsize,)).view(api.HSP_DT).copy()
                    got.append((oid, a))
                    return 0
                import time
                while True:
                    with mu:
                        first = bookmark[0]; bookmark[0] += chunk
                    if first >= nsub:
                        break
                    time.sleep(delays[t])                           # (shifts which thread is back first for the next chunk)
                    blk = get_block(list(range(first, min(first + chunk, nsub))))
                    api._check(L.gbn_prelim_search_lists(ps._b, blk, sink, None, None, None, None))
                with mu:
                    lists.extend(got)
                ps.close()
            except Exception as e:      # noqa
                errs.append(repr(e))
        th = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
        [t.start() for t in th]; [t.join() for t in th]
        assert not errs, errs
        lists.sort(key=lambda x: x[0])
        return np.concatenate([a for _, a in lists]) if lists else np.zeros(0, dtype=api.HSP_DT)

    up0 = L.gbn_debug_db_bytes_uploaded()
    first = one_batch(2, [0.0, 0.02])
    up1 = L.gbn_debug_db_bytes_uploaded()
    second = one_batch(3, [0.03, 0.0, 0.01])
    up2 = L.gbn_debug_db_bytes_uploaded()
    assert up1 - up0 >= nsub * slen // 4            # every block went up once ...
    assert up2 == up1                               # ... and never again
    assert first.tobytes() == want.tobytes() and second.tobytes() == want.tobytes()
    L.gbn_release_db_memory()


@pytest.mark.gpu
def test_switches_are_read_at_every_use(monkeypatch):
    """The library's A/B switches are read from the environment at every use (gbn_dev.h: gbn::switch_value; rounds 1-3 cached
    several in function-local statics): a process runs the same search under several settings -- here the two-kernel seed
    stage forced on small inputs with the composite-key sort on and off, and the direct-probe scan -- and gets the same
    stages every time, equal to the oracle's."""
    db, queries, plants, subjects, opt = util.small_case(6, 150_000, 16, task="blastn")
    src = api.BlastSeqSrc.from_packed(subjects)
    ora, s = util.oracle_run(opt, queries, subjects)
    seen = []
    for env in ({}, {"GBN_DIAG_COMPACT_MIN": "1"}, {"GBN_DIAG_COMPACT_MIN": "1", "GBN_SEED_CKEYS": "0"}, {"GBN_SCAN_BINS": "1"}, {}):
        for k in ("GBN_DIAG_COMPACT_MIN", "GBN_SEED_CKEYS", "GBN_SCAN_BINS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ps = api.BlastPrelimSearch(queries, opt, src)
        gpu = ps.run(keep_stages=True)
        util.compare_stages(gpu, ora)
        seen.append((ps.info()["scan_path"], gpu["hsps"].tobytes()))
        ps.close()
    assert len(set(h for _, h in seen)) == 1
    assert seen[3][0] != seen[0][0]                  # the direct-probe scan did run in the fourth round


def _bits_for(below):
    k = 1
    while k < 63 and (1 << k) < below:
        k += 1
    return k


@pytest.mark.gpu
@pytest.mark.parametrize("nsubj,nseg,mean,container_hash,diag_len,q_desc", [
    (7, 64, 30_000, 1, 0, 0),            # a few subjects of many chunks each
    (300, 4096, 3_000, 1, 0, 1),         # every segment table entry in use, most subjects inside one chunk
    (5, 16, 50_000, 0, 2048, 0),         # diagonal array: 2,048 slots
    (4096, 700, 70, 1, 0, 0),            # as many subjects as a launch takes, most with a handful of seeds, many without
    (1, 3, 4096, 1, 0, 0),               # one subject, exactly one chunk ...
    (1, 3, 4097, 1, 0, 1),               # ... and one seed more
    (3, 4096, 2, 1, 0, 0),               # nearly all segments empty
])
def test_seed_order_kernels_equal_a_stable_sort(nsubj, nseg, mean, container_hash, diag_len, q_desc):
    """csrc/seed_order.hip (the seeds of an ordered scan by (subject, slot), scan order inside -- the order in which the diagonal
    container meets them, CORE/na_ungapped.c:611-922) against numpy's stable sort of the same composite keys: segments of
    uneven length, empty ones, subjects without seeds, subjects across many segments and segments across many subjects."""
    rng = np.random.default_rng(nsubj * 1000 + nseg)
    qlen, max_len, subj_base = 200_001, 1_000_000, 17
    counts = rng.poisson(mean, nsubj) if mean > 100 else rng.integers(0, 2 * mean + 1, nsubj)
    if nsubj in (1,):
        counts[:] = mean
    counts[rng.random(nsubj) < 0.2] = 0 if nsubj > 1 else counts[0]
    n = int(counts.sum())
    subj = np.repeat(np.arange(nsubj, dtype=np.int32) + subj_base, counts)
    s_scan = np.concatenate([np.sort(rng.integers(0, max_len, c)) for c in counts]).astype(np.int32) if n else np.zeros(0, np.int32)
    q_pos = rng.integers(0, qlen, n).astype(np.int32)
    ext_left = rng.integers(0, 17, n).astype(np.int32)
    seeds = np.stack([subj, s_scan, q_pos, ext_left], axis=1).astype(np.int32)
    cuts = np.sort(rng.integers(0, n + 1, nseg - 1)) if n else np.zeros(nseg - 1, np.int64)
    if nseg > 1000:
        cuts[: nseg // 2] = cuts[nseg // 2]           # a run of empty segments in front
    bounds = np.concatenate([[0], cuts, [n]])
    seg_count = np.diff(bounds).astype(np.uint32)
    seg_cap = int(max(seg_count.max(), 1)) + 5
    seg = np.full((nseg, seg_cap, 4), -7, np.int32)
    for g in range(nseg):
        seg[g, : seg_count[g]] = seeds[bounds[g]: bounds[g + 1]]
    reported = seg_count.copy()
    out = np.zeros(max(n, 1), np.uint64)
    n_out = C.c_int64(0)
    L = api.lib()
    api._check(L.gbn_debug_seed_order(seg.ctypes.data, reported.ctypes.data, nseg, seg_cap, nsubj, subj_base, container_hash, diag_len,
                                      qlen, max_len, q_desc, out.ctypes.data, C.byref(n_out)))
    assert n_out.value == n
    q_bits, s_bits = min(32, _bits_for(qlen + 1)), _bits_for(max_len + 1)
    gb = 9 if container_hash else _bits_for(max(diag_len, 2))
    qh_bits = max(0, q_bits - gb)
    v_bits = 8 + qh_bits
    qkey = ((1 << q_bits) - 1 - q_pos.astype(np.int64)) if q_desc else q_pos.astype(np.int64)
    slot = ((s_scan.astype(np.int64) - q_pos) & 511) if container_hash else ((s_scan.astype(np.int64) + diag_len - q_pos) & (diag_len - 1))
    group = ((subj.astype(np.int64) - subj_base) << gb) | slot
    val = ext_left.astype(np.int64) | ((qkey >> gb if qh_bits else 0) << 8)
    key = ((((group << s_bits) | s_scan) << v_bits) | val).astype(np.uint64)
    expect = key[np.argsort(group, kind="stable")]
    assert np.array_equal(out[:n], expect)


def test_skewed_synthetic_database_equals_its_numpy_form():
    """bench.py --skew: gbn_synth_skew over a gbn_synth_fill slab = synth.SynthDb(skew=True).subject_packed, subject by subject (the
    oracle's input for the parity sample of the skewed workload)"""
    import torch
    from gblastn_amd import synth
    db = synth.SynthDb(120, 400_000, seed=0x1234567, first_oid=1000, skew=True)
    slab = torch.empty(db.nbytes, dtype=torch.uint8, device="cuda")
    api._check(api.lib().gbn_synth_fill(slab.data_ptr(), db.nbytes, db.seed, None))
    db.skew_on_device(api, slab.data_ptr())
    host = slab.cpu().numpy()
    fam = 0
    for i in range(db.num):
        want = db.subject_packed(i, pad=0)
        got = host[db.byte_off[i]: db.byte_off[i] + len(want)]
        assert np.array_equal(got, want), i
    plain = synth.SynthDb(120, 400_000, seed=0x1234567, first_oid=1000)
    changed = sum(int((plain.subject_packed(i, pad=0) != db.subject_packed(i, pad=0)).sum()) for i in range(10))
    assert changed > 10 * 100_000 * 0.05            # ~8 % of the bytes of every subject
