"""CPU: the product's host set-up (C++: Karlin-Altschul blocks, cut-offs, effective
lengths, lookup-table choice) against the oracle, and C-ABI load/export checks.
No compute calls -- there is no GPU here."""
import ctypes as C
import os
import re
import numpy as np
import pytest
from gblastn_amd import api
from oracle import orc
from tests import util

CTX_INT = ["query_offset", "query_length", "frame", "query_index", "is_valid", "length_adjustment",
           "eff_searchsp", "x_dropoff", "cutoff_score", "reduced_cutoff", "gap_cutoff_score",
           "gap_cutoff_score_max"]
CTX_F64 = ["lambda_u", "K_u", "logK_u", "H_u"]


def bits(x):
    return np.float64(x).view(np.uint64)


@pytest.mark.parametrize("task,nq,qlen,dbl,dbn", [
    ("megablast", 1, 1000, 10_000_000, 10),
    ("megablast", 40, 1000, 50_000_000_000, 50_000),
    ("megablast", 400, 1000, 400_000_000_000, 400_000),
    ("blastn", 3, 700, 5_000_000_000, 5_000),
    ("blastn", 30, 1000, 5_000_000_000, 5_000),
])
def test_setup_matches_oracle(task, nq, qlen, dbl, dbn):
    rng = np.random.default_rng(nq * 7 + qlen)
    qs = [rng.integers(0, 4, qlen, dtype=np.uint8) for _ in range(nq)]
    qs[0][5] = 14                       # an N: not counted in the composition, never indexed
    gopt = api.default_options(task, db_length=dbl, db_num_seqs=dbn)
    b = api.BlastPrelimSearch(qs, gopt, upload=False)
    s = orc.Search(util.oracle_options(gopt), qs)
    gi, oi = b.info(), s.info()
    for k in ["lut_type", "lut_width", "scan_step", "container", "gap_x_dropoff"]:
        assert gi[k] == oi[k], k
    gc, oc = b.contexts, s.contexts
    assert len(gc) == len(oc) == 2 * nq
    for g, o in zip(gc, oc):
        for f in CTX_INT:
            assert getattr(g, f) == getattr(o, f), f
        for f in CTX_F64:
            assert bits(getattr(g, f)) == bits(getattr(o, f)), f       # bit-identical doubles


def test_unsupported_options_are_errors_not_fallbacks():
    q = [np.zeros(100, dtype=np.uint8)]
    with pytest.raises(api.BlastError):
        api.BlastPrelimSearch(q, api.default_options("megablast", reward=2, penalty=-1), upload=False)
    with pytest.raises(api.BlastError):     # gap costs outside the Karlin-Altschul tables
        api.BlastPrelimSearch([np.arange(100, dtype=np.uint8) % 4],
                              api.default_options("megablast", gap_open=1, gap_extend=3), upload=False)


def test_c_abi_exports_every_declared_symbol():
    L = api.lib()
    hdr = open(os.path.join(os.path.dirname(api._HERE), "include", "gblastn_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b((?:gbn_|Blast_gpu_|gpu_Release)\w*)\s*\(", hdr))
    assert len(declared) > 30
    for name in sorted(declared):
        assert hasattr(L, name), "declared in include/gblastn_amd.h but not exported: " + name
    for name in api.EXPORTS:
        assert name in declared, "exported but undeclared: " + name


def test_no_gpu_is_reported_not_emulated():
    # on a box without a HIP device the engine must fail loudly; there is no CPU path
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = api.lib()
    assert L.gbn_init(1, -1) != 0
    assert b"HIP device" in L.gbn_last_error()
    assert L.gbn_init(0, -1) != 0          # use_gpu = false is refused as well
    with pytest.raises(api.BlastError):
        api.BlastPrelimSearch([np.zeros(64, dtype=np.uint8)], api.default_options("megablast"))
    # the per-device entry points: nothing to lease, nothing in flight, nothing to release
    assert L.gbn_device_count() == 0
    assert L.gbn_use_device(0) != 0 and L.gbn_use_device(-1) != 0
    assert L.gbn_current_device() == -1
    assert L.gbn_prelim_search_end(None) == 0
    assert L.gbn_debug_check_guards() == 0
    L.gbn_release(); L.gbn_release_db_memory()


def test_product_never_imports_oracle():
    root = os.path.dirname(api._HERE)
    for dirpath, _, files in os.walk(os.path.join(root, "gblastn_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                txt = open(os.path.join(dirpath, f)).read()
                for pat in [r"^\s*import\s+oracle", r"^\s*from\s+oracle\b", r"import\s+orc\b",
                            r"#include\s*[\"<].*orc(_int)?\.h", r"liborc"]:
                    assert not re.search(pat, txt, flags=re.M), (f, pat)


def test_masked_setup_matches_oracle_and_rejects_bad_masks():
    rng = np.random.default_rng(3)
    qs = [rng.integers(0, 4, 1000, dtype=np.uint8) for _ in range(12)]
    # heavy masking moves the table-size estimate below a lookup-choice threshold
    masks = [(i, 100, 899) for i in range(12)]
    gopt = api.default_options("megablast", db_length=10**9, db_num_seqs=1000)
    plain = api.BlastPrelimSearch(qs, gopt, upload=False).info()
    b = api.BlastPrelimSearch(qs, gopt, upload=False, masks=masks).info()
    o = orc.Search(util.oracle_options(gopt), qs, masks=masks).info()
    for k in ["lut_type", "lut_width", "scan_step", "container"]:
        assert b[k] == o[k], k
    assert (plain["lut_type"], plain["lut_width"]) != (b["lut_type"], b["lut_width"])
    for bad in ([(0, 10, 5)], [(0, 10, 2000)], [(12, 1, 2)], [(0, 10, 50), (0, 40, 60)], [(-1, 0, 1)]):
        with pytest.raises(api.BlastError):
            api.BlastPrelimSearch(qs, gopt, upload=False, masks=bad)


def test_batch_setup_in_a_forked_child():
    """the set-up's worker pool lives with the library; a child made by fork() (multiprocessing's default start method)
    has the parent's pool object and none of its threads.  The parent sets a batch up (the pool exists), forks, and the child sets
    one up by itself -- the per-context loops of 800 contexts are spread over workers -- and reports the same cut-offs."""
    import os
    rng = np.random.default_rng(3)
    qs = [rng.integers(0, 4, 1000, dtype=np.uint8) for _ in range(400)]
    gopt = api.default_options("megablast", db_length=50_000_000_000, db_num_seqs=50_000)
    want = [c.cutoff_score for c in api.BlastPrelimSearch(qs, gopt, upload=False).contexts]
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        try:
            got = [c.cutoff_score for c in api.BlastPrelimSearch(qs, gopt, upload=False).contexts]
            os.write(w, b"ok" if got == want else b"differs")
        finally:
            os._exit(0)
    os.close(w)
    import select
    ready, _, _ = select.select([r], [], [], 60)
    out = os.read(r, 16) if ready else b"timeout"
    if not ready:
        os.kill(pid, 9)
    os.waitpid(pid, 0)
    assert out == b"ok", out


def test_host_cpus_follow_the_affinity_mask_and_the_cgroup_quota():
    """gbn_host_cpus: what the library sizes its host pools from.  Never more than the hardware threads or the affinity mask, the
    cgroup's quota when there is one (the GPU boxes: 16 CPUs of quota on 256 hardware threads -- pools sized from the hardware
    threads had the scheduler stop the whole process for the rest of a period), GBN_HOST_CPUS overrides."""
    import math, subprocess, sys
    n = api.host_cpus()
    want = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            want = min(want, max(1, math.ceil(int(q) / int(per))))
    except OSError:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                want = min(want, max(1, math.ceil(q / per)))
        except OSError:
            pass
    assert n == want and n >= 1
    assert api.granted_cpus() == n                      # (the launchers' copy of the rule, without the library)
    out = subprocess.run([sys.executable, "-c", "from gblastn_amd import api; print(api.host_cpus())"], env=dict(os.environ, GBN_HOST_CPUS="3"),
                         capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.stdout.strip() == "3", out.stderr[-500:]


def test_setup_of_many_compositions_matches_oracle_whatever_the_thread_share():
    """The per-context Karlin-Altschul parameters are solved once per score distribution a thread has seen and found again by the
    distribution's exact bits (csrc/stat.cpp, round 6): a batch of very different compositions -- uniform, AT-rich, one letter only,
    half ambiguity codes, a query of Ns alone -- against the oracle, bit for bit, set up on the whole pool and on one thread
    (gbn_set_setup_threads)."""
    rng = np.random.default_rng(99)
    qs = []
    for i in range(300):
        kind = i % 6
        if kind == 0: q = rng.integers(0, 4, 600 + i, dtype=np.uint8)
        elif kind == 1: q = rng.choice(np.array([0, 3, 0, 3, 1, 2], dtype=np.uint8), 500 + i)
        elif kind == 2: q = np.full(400 + i, i % 4, dtype=np.uint8)
        elif kind == 3: q = rng.integers(0, 4, 700, dtype=np.uint8); q[rng.random(700) < 0.5] = 14
        elif kind == 4: q = rng.choice(np.array([0, 0, 0, 1, 2, 3], dtype=np.uint8), 900)
        else: q = rng.integers(0, 4, 300 + 3 * i, dtype=np.uint8); q[::7] = np.uint8(4 + i % 10)
        qs.append(q)
    qs[17] = np.full(200, 14, dtype=np.uint8)           # no valid context
    gopt = api.default_options("blastn", db_length=5_000_000_000, db_num_seqs=5_000)
    s = orc.Search(util.oracle_options(gopt), qs)
    L = api.lib()
    L.gbn_set_setup_threads.argtypes = [C.c_int32]; L.gbn_set_setup_threads.restype = None
    try:
        for share in (0, 1, 3):
            L.gbn_set_setup_threads(share)
            b = api.BlastPrelimSearch(qs, gopt, upload=False)
            gc, oc = b.contexts, s.contexts
            assert len(gc) == len(oc) == 2 * len(qs)
            for g, o in zip(gc, oc):
                for f in CTX_INT:
                    assert getattr(g, f) == getattr(o, f), (share, f)
                for f in CTX_F64:
                    assert bits(getattr(g, f)) == bits(getattr(o, f)), (share, f)
            b.close()
    finally:
        L.gbn_set_setup_threads(0)


def test_local_ranks_share_the_granted_cpus():
    """One process per GPU: a launcher hands every rank of a node its share of the CPUs the node grants (api.share_cpus_among_local_ranks
    sets GBN_HOST_CPUS before the library sizes its pools); an explicit GBN_HOST_CPUS stays."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "from gblastn_amd import api; api.share_cpus_among_local_ranks(); print(api.host_cpus(), api.granted_cpus())"
    env = {k: v for k, v in os.environ.items() if k not in ("GBN_HOST_CPUS", "LOCAL_WORLD_SIZE", "WORLD_SIZE")}
    one = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=root).stdout.split()
    assert one[0] == one[1]                                             # a single process: everything it is granted
    four = subprocess.run([sys.executable, "-c", code], env=dict(env, LOCAL_WORLD_SIZE="4"), capture_output=True, text=True, cwd=root).stdout.split()
    assert int(four[0]) == max(2, int(four[1]) // 4)
    kept = subprocess.run([sys.executable, "-c", code], env=dict(env, LOCAL_WORLD_SIZE="4", GBN_HOST_CPUS="5"), capture_output=True, text=True, cwd=root).stdout.split()
    assert kept[0] == "5"
