"""-m gpu: the HIP path, called through the C ABI, against the CPU oracle."""
import os
import numpy as np
import pytest
from gblastn_amd import api, synth
from tests import util

pytestmark = pytest.mark.gpu


def run_case(nsub, slen, nq, task="megablast", qlen=1000, expect=None, **kw):
    db, queries, plants, subjects, opt = util.small_case(nsub, slen, nq, qlen=qlen, task=task, **kw)
    src = api.BlastSeqSrc.from_packed(subjects)
    ps = api.BlastPrelimSearch(queries, opt, src)
    info = ps.info()
    if expect:
        for k, v in expect.items():
            assert info[k] == v, (k, info)
    gpu = ps.run(keep_stages=True)
    ora, s = util.oracle_run(opt, queries, subjects)
    oi = s.info()
    assert (oi["lut_type"], oi["lut_width"], oi["scan_step"], oi["container"]) == \
           (info["lut_type"], info["lut_width"], info["scan_step"], info["container"])
    util.compare_stages(gpu, ora)
    d = ps.diagnostics
    assert d.lookup_hits == s.stats.lookup_hits
    assert d.good_init_extends == s.stats.good_init_extends
    assert d.gapped_extensions == s.stats.gapped_extensions
    assert d.good_extensions == s.stats.good_extensions
    nh = len(gpu["hsps"])
    return nh, plants


@pytest.fixture(params=["auto", "direct"], autouse=True)
def scan_variant(request, monkeypatch):
    """Every case runs with the engine's own choice of scan kernel (key-range partitioned
    for lut >= 10) and with the direct-probe kernel forced."""
    if request.param == "direct":
        monkeypatch.setenv("GBN_SCAN_BINS", "1")
    else:
        monkeypatch.delenv("GBN_SCAN_BINS", raising=False)


def test_megablast_small_query_smallna_lut8():
    # C1-shaped: one 1 kb query -> small table lut 8, stride 21, diag array
    nh, _ = run_case(6, 200_000, 1, planted_fraction=1.0,
                     expect=dict(lut_type=1, lut_width=8, scan_step=21, container=0))
    assert nh >= 1


def test_megablast_mb_lut11_diag_hash():
    # 16 x 1 kb -> 31,968 entries -> megablast table lut 11, stride 18, hash container
    nh, plants = run_case(8, 150_000, 16, expect=dict(lut_type=3, lut_width=11, scan_step=18, container=1))
    assert nh >= len(plants) // 2


def test_megablast_mb_lut12():
    # > 300,000 entries -> lut 12, stride 17 (the C2 table)
    nh, plants = run_case(6, 120_000, 160, expect=dict(lut_type=3, lut_width=12, scan_step=17, container=1))
    assert nh >= 1


@pytest.mark.parametrize("go,ge", [(2, 2), (1, 2), (0, 2), (3, 1), (1, 1)])
def test_megablast_affine_greedy(go, ge):
    # BLAST_AffineGreedyAlign: every affine gap cost of the 1/-2 table
    nh, plants = run_case(6, 100_000, 12, gap_open=go, gap_extend=ge)
    assert nh >= 1


def test_blastn_small_lut8_stride4():
    nh, _ = run_case(4, 60_000, 2, task="blastn", planted_fraction=1.0,
                     expect=dict(lut_type=1, lut_width=8, scan_step=4, container=0))
    assert nh >= 1


def test_blastn_mb_lut11_stride1():
    nh, _ = run_case(3, 40_000, 8, task="blastn",
                     expect=dict(lut_type=3, lut_width=11, scan_step=1, container=1))
    assert nh >= 1


def test_blastn_word12_lut12_sixteen_slices_folded():
    """blastn W=12 with 460 kb of query: a lut-12 table as wide as the word (CORE/blast_nalookup.c:130-180: from 900,000
    entries up), sixteen slices of presence bits -- scan_fold_kernel keeps their OR in LDS (more than half full here) and
    finds the entry lists through the rank tables (GbnScanParams::pvx / pstart)"""
    nh, _ = run_case(3, 50_000, 460, task="blastn", word_size=12, planted_fraction=0.05,
                     expect=dict(lut_width=12, scan_step=1, container=1))
    assert nh >= 1


def test_reference_known_answers_on_gpu():
    """The reference's own known answers through the HIP path."""
    import os
    from oracle import orc
    g = os.path.join(os.path.dirname(__file__), "golden")
    packed, n = orc.read_blastdb_v4_nucl(os.path.join(g, "nt.41646578"))[0]
    q = orc.unpack_ncbi2na(packed, n)[54:561].copy()
    src = api.BlastSeqSrc.from_packed([(packed, n)])
    ps = api.BlastPrelimSearch([q], api.default_options("megablast", db_length=n, db_num_seqs=1), src)
    h = ps.run()["hsps"]
    assert len(h) == 1 and h[0]["context"] == 0
    assert (h[0]["q_offset"], h[0]["q_end"] - 1, h[0]["s_offset"], h[0]["s_end"] - 1) == (0, 506, 54, 560)
    a = orc.encode_blastna(orc.read_fasta(os.path.join(g, "greedy1a.fsa"))[0])
    b = orc.encode_blastna(orc.read_fasta(os.path.join(g, "greedy1b.fsa"))[0])
    src = api.BlastSeqSrc.from_packed([(orc.pack_ncbi2na(b), len(b))])
    ps = api.BlastPrelimSearch([a], api.default_options("megablast"), src)
    h = ps.run()["hsps"]
    assert len(h) == 1 and h[0]["score"] == 619
    ps = api.BlastPrelimSearch([a], api.default_options("megablast", reward=10, penalty=-25,
                               xdrop_gap_bits=100.0, xdrop_gap_final_bits=100.0), src)
    h = ps.run()["hsps"]
    assert len(h) == 1 and h[0]["score"] == 6034


def _mutate(rng, seq, subs, indels):
    s = seq.copy()
    for _ in range(subs):
        p = int(rng.integers(0, len(s))); s[p] = (s[p] + 1 + rng.integers(0, 3)) & 3
    for _ in range(indels):
        p = int(rng.integers(5, len(s) - 5))
        s = np.delete(s, p) if rng.random() < 0.5 else np.insert(s, p, rng.integers(0, 4))
    return s.astype(np.uint8)


def ragged_case(task, rng_seed, with_n=True):
    """Edge cases the reference's tests exercise: subjects shorter than the lookup word, lengths that
    are not multiples of 4, hits touching either end of a subject, queries shorter than the word
    size, ambiguity codes (N = 14 in BLASTNA) inside queries, hits on the minus strand."""
    rng = np.random.default_rng(rng_seed)
    lens = [2, 3, 7, 11, 12, 27, 28, 29, 31, 64, 257, 1001, 4099, 30_003, 50_001, 1]
    subs = [rng.integers(0, 4, n, dtype=np.uint8) for n in lens]
    queries = []
    # homologs at the very start / very end of a subject, and spanning a whole short subject
    queries.append(_mutate(rng, subs[13][:700], 12, 2))                         # starts at subject base 0
    queries.append(_mutate(rng, subs[14][-650:], 10, 1))                        # ends at the last base
    queries.append(np.concatenate([rng.integers(0, 4, 200, dtype=np.uint8), subs[11], rng.integers(0, 4, 200, dtype=np.uint8)]))
    queries.append((3 - _mutate(rng, subs[13][9000:9800], 15, 2))[::-1].copy())  # minus strand
    queries.append(rng.integers(0, 4, 20, dtype=np.uint8))                      # shorter than word size 28
    queries.append(subs[9][:40].copy())                                         # tiny query, tiny subject
    q = _mutate(rng, subs[13][20000:21000], 8, 1)
    if with_n:
        q[100:104] = 14; q[500] = 14; q[777] = 4                                # N runs, single N, another ambiguity code
    queries.append(q)
    q = _mutate(rng, subs[12][1000:1900], 30, 3)
    if with_n:
        q[::97] = 14
    queries.append(q)
    from oracle import orc
    subjects = [(orc.pack_ncbi2na(s), len(s)) for s in subs]
    opt = api.default_options(task, db_length=int(sum(lens)), db_num_seqs=len(lens))
    return queries, subjects, opt


@pytest.mark.parametrize("task", ["megablast", "blastn"])
def test_ragged_subjects_short_queries_and_ambiguity_codes(task):
    queries, subjects, opt = ragged_case(task, 2024)
    src = api.BlastSeqSrc.from_packed(subjects)
    ps = api.BlastPrelimSearch(queries, opt, src)
    gpu = ps.run(keep_stages=True)
    ora, s = util.oracle_run(opt, queries, subjects)
    util.compare_stages(gpu, ora)
    assert len(gpu["hsps"]) >= 5
    d = ps.diagnostics
    assert (d.lookup_hits, d.good_init_extends, d.gapped_extensions, d.good_extensions) == \
           (s.stats.lookup_hits, s.stats.good_init_extends, s.stats.gapped_extensions, s.stats.good_extensions)


def test_first_oid_offsets_and_two_shards_equal_one():
    # a database cut into two shards (global statistics) gives the records of the whole database
    db, queries, plants, subjects, opt = util.small_case(8, 80_000, 12)
    whole = api.BlastPrelimSearch(queries, opt, api.BlastSeqSrc.from_packed(subjects)).run()["hsps"]
    a = api.BlastPrelimSearch(queries, opt, api.BlastSeqSrc.from_packed(subjects[:3], first_oid=0)).run()["hsps"]
    b = api.BlastPrelimSearch(queries, opt, api.BlastSeqSrc.from_packed(subjects[3:], first_oid=3)).run()["hsps"]
    both = np.concatenate([a, b])
    assert np.array_equal(both, whole)


def test_full_size_c2_pass_against_the_oracle_subject_by_subject():
    """BASELINE.json configs[1] at full size: 50,000 x 1 Mb subjects, one 5 Mb query batch (lut 12,
    stride 17, diagonal hash).  Subjects are independent given the global statistics, so the pass is
    checked bit for bit against the oracle on every subject that produced an HSP plus a sample of the
    others, and for determinism (two passes give identical records)."""
    import torch
    nsub, slen, nq = 50_000, 1_000_000, 5_000
    db = synth.SynthDb(nsub, slen, seed=0x9E3779B97F4A7C15 ^ 1)
    slab = torch.empty(db.nbytes, dtype=torch.uint8, device="cuda")
    api._check(api.lib().gbn_synth_fill(slab.data_ptr(), db.nbytes, db.seed, None))
    src = api.BlastSeqSrc.from_slab((slab.data_ptr(), db.nbytes), db.byte_off, db.lens, is_device=True, keep=slab)
    queries, plants = synth.make_queries(nq, db)
    opt = api.default_options("megablast", db_length=nsub * slen, db_num_seqs=nsub)
    ps = api.BlastPrelimSearch(queries, opt, src)
    assert ps.info()["lut_width"] == 12 and ps.info()["scan_step"] == 17 and ps.info()["container"] == 1
    first = ps.run()["hsps"]
    again = ps.run()["hsps"]
    assert np.array_equal(first, again)
    assert np.all(np.diff(first["oid"]) >= 0)
    assert ps.diagnostics.subject_bases_scanned == 2 * nsub * slen
    hit_oids = np.unique(first["oid"])
    assert len(hit_oids) >= 0.9 * len(plants)          # planted homologs are found
    rng = np.random.default_rng(3)
    sample = np.union1d(hit_oids, rng.choice(nsub, 25, replace=False))
    from oracle import orc
    s = orc.Search(util.oracle_options(opt), queries)
    # the reference's container semantics: ONE diagonal container carried from subject to subject in OID order
    # (CORE/blast_extend.c:166-190); the HIP path starts every subject fresh -- DESIGN.md "a6", tests/test_diag_carry.py
    s.carry_diag(True)
    for oid in sample.tolist():                         # (ascending)
        o = s.subject(db.subject_packed(oid), slen)
        g = first[first["oid"] == oid]
        assert len(g) == len(o["hsps"]), (oid, len(g), len(o["hsps"]))
        for f in ["context", "q_offset", "q_end", "q_gapped_start", "s_offset", "s_end", "s_gapped_start", "score"]:
            assert np.array_equal(g[f], o["hsps"][f]), (oid, f)
        assert np.array_equal(g["evalue"].view(np.uint64), o["hsps"]["evalue"].view(np.uint64)), oid


def test_pipelined_begin_end_equals_run():
    # two query batches alternating, gapped stage of pass k overlapping the scan of pass k+1
    db, queries, plants, subjects, opt = util.small_case(8, 150_000, 40)
    src = api.BlastSeqSrc.from_packed(subjects)
    a = api.BlastPrelimSearch(queries[:20], opt, src)
    b = api.BlastPrelimSearch(queries[20:], opt, src)
    want = [a.run()["hsps"], b.run()["hsps"]]
    assert len(want[0]) + len(want[1]) >= 4
    got, prev = [], None
    for k in range(6):
        cur = (a, b)[k % 2]
        cur.begin()
        if prev is not None:
            got.append(prev.end()["hsps"])
        prev = cur
    got.append(prev.end()["hsps"])
    for k, g in enumerate(got):
        assert np.array_equal(g, want[k % 2]), k
    assert np.array_equal(a.run()["hsps"], want[0])     # the synchronous call still works afterwards


# option surface of the path: word sizes (every lookup choice of BlastChooseNaLookupTable that a
# batch of this size reaches), scoring systems of the Karlin-Altschul tables, X-drops, e-value and
# diagonal-separation settings, raw-score cut-off, stock-NCBI word_size 11 rule
OPTION_SWEEP = [
    ("megablast", dict(word_size=16)),
    ("megablast", dict(word_size=20)),
    ("megablast", dict(word_size=24)),
    ("megablast", dict(word_size=32)),
    ("megablast", dict(word_size=64)),
    ("megablast", dict(word_size=12)),
    ("megablast", dict(reward=1, penalty=-3)),
    ("megablast", dict(reward=1, penalty=-1, gap_open=3, gap_extend=2)),
    ("megablast", dict(reward=2, penalty=-3, gap_open=5, gap_extend=2)),
    ("megablast", dict(xdrop_gap_bits=40.0, xdrop_ungap_bits=10.0)),
    ("megablast", dict(evalue=1e-20, min_diag_separation=0)),
    ("megablast", dict(cutoff_score=200)),
    ("blastn", dict(word_size=7)),
    ("blastn", dict(word_size=9)),
    ("blastn", dict(word_size=10)),
    ("blastn", dict(word_size=11, lut11_gblastn_rule=0)),
    ("blastn", dict(word_size=15)),
    ("blastn", dict(reward=1, penalty=-2, gap_open=2, gap_extend=2)),
    ("blastn", dict(reward=4, penalty=-5, gap_open=12, gap_extend=8)),
    ("blastn", dict(reward=1, penalty=-4, gap_open=5, gap_extend=2)),
    ("blastn", dict(evalue=1e-3, min_diag_separation=10, xdrop_gap_bits=20.0)),
    ("blastn", dict(greedy=1, gap_open=0, gap_extend=0)),       # greedy preliminary stage with blastn scoring
]


@pytest.mark.parametrize("task,kw", OPTION_SWEEP, ids=lambda v: v if isinstance(v, str) else "-".join("%s=%s" % i for i in v.items()))
def test_option_sweep(task, kw):
    nq = 24 if task == "megablast" else 6
    nh, plants = run_case(5, 60_000, nq, task=task, planted_fraction=0.6, **kw)
    if kw.get("cutoff_score", 0) == 0 and kw.get("evalue", 10) >= 1e-10:
        assert nh >= 1


@pytest.mark.spawns
def test_reused_binning_gives_identical_results(monkeypatch):
    # the record cache (default policy): the scan records of a shard serve every later query batch of the same table shape;
    # GBN_RECORD_CACHE_MB=0 switches it off (every pass bins for itself)
    import subprocess, sys, os, json
    code = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
from gblastn_amd import api
from tests import util
db, queries, plants, subjects, opt = util.small_case(12, 1_000_000, 320)
src = api.BlastSeqSrc.from_packed(subjects)
a = api.BlastPrelimSearch(queries[:160], opt, src); b = api.BlastPrelimSearch(queries[160:], opt, src)
assert a.info()["lut_width"] == 12
out = []
for k in range(4):
    cur = (a, b)[k %% 2]
    before = cur.diagnostics.bin_kernel_ms
    h = cur.run()["hsps"]
    out.append([h.tobytes().hex(), cur.diagnostics.bin_kernel_ms - before])
print(json.dumps(out))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for flag in ("0", "1"):
        env = dict(os.environ); env.pop("GBN_SCAN_BINS", None); env.pop("GBN_RECORD_CACHE_MB", None)
        if flag == "0":
            env["GBN_RECORD_CACHE_MB"] = "0"
        p = util.run_child([sys.executable, "-c", code], env=env, timeout=600)
        res[flag] = json.loads(p.stdout.strip().splitlines()[-1])
    assert [r[0] for r in res["0"]] == [r[0] for r in res["1"]]
    # with reuse only the first pass runs the binning kernel
    assert sum(r[1] for r in res["1"][1:]) < 0.6 * sum(r[1] for r in res["0"][1:])


@pytest.mark.spawns
def test_subject_ranges_do_not_change_results():
    # the engine cuts a shard into subject ranges (GBN_RANGE_GIB, position-id width); results must not depend on it
    import subprocess, sys, os, json
    code = r'''
import sys, json
sys.path.insert(0, %r)
from gblastn_amd import api
from tests import util
db, queries, plants, subjects, opt = util.small_case(40, 60_000_0, 200)
src = api.BlastSeqSrc.from_packed(subjects)
ps = api.BlastPrelimSearch(queries, opt, src)
h = ps.run()["hsps"]
launches, hits = int(ps.diagnostics.scan_launches), int(ps.diagnostics.lookup_hits)
# the pipelined entry points: every range's seed + extension stages run on the second stream
# underneath the scan of the next range, results appended range by range
ps.begin(); h2 = ps.end()["hsps"]
assert h2.tobytes() == h.tobytes()
print(json.dumps([h.tobytes().hex(), launches, hits]))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    # (the last two: every launch of the seed stage through seed_ext_kernel + diag_replay_kernel, see
    # test_parity_suite_through_the_two_kernel_seed_stage -- that run leaves the tests with child processes to this file)
    for tag, env_add in (("one", {}), ("many", {"GBN_RANGE_MIB": "1"}), ("tiles", {"GBN_RANGE_TILES": "3"}),
                         ("one-2k", {"GBN_DIAG_COMPACT_MIN": "1"}), ("tiles-2k", {"GBN_RANGE_TILES": "3", "GBN_DIAG_COMPACT_MIN": "1"})):
        env = dict(os.environ); env.update(env_add)
        p = util.run_child([sys.executable, "-c", code], env=env, timeout=600)
        out[tag] = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["one"][1] == 1 and out["many"][1] > 3 and out["tiles"][1] > 3
    assert out["one"][0] == out["many"][0] and out["one"][2] == out["many"][2]
    assert out["one"] == out["one-2k"] and out["tiles"] == out["tiles-2k"]
    assert out["one"][0] == out["tiles"][0] and out["one"][2] == out["tiles"][2]


def test_interrupt_callback_stops_between_ranges():
    import ctypes as C
    db, queries, plants, subjects, opt = util.small_case(8, 100_000, 8)
    src = api.BlastSeqSrc.from_packed(subjects)
    ps = api.BlastPrelimSearch(queries, opt, src)
    calls = []
    CB = C.CFUNCTYPE(C.c_int, C.c_void_p)
    cb = CB(lambda p: (calls.append(1), 1)[1])
    L = api.lib()
    rc = L.gbn_prelim_search(ps._b, src._h, ps._r, C.byref(ps.diagnostics), 0, C.cast(cb, C.c_void_p), None)
    assert rc != 0 and b"interrupt" in L.gbn_last_error() and len(calls) == 1
    assert len(ps.run()["hsps"]) >= 1                   # the engine is usable afterwards


def random_masks(rng, queries, per_query=3, edge_cases=True):
    """Soft masks like DUST produces them: a few intervals per query, some touching the ends,
    some abutting, some shorter than a lookup word."""
    masks = []
    for qi, q in enumerate(queries):
        n = len(q)
        cuts = sorted(set(int(x) for x in rng.integers(0, n, 2 * per_query)))
        iv = [(cuts[i], cuts[i + 1] - 1) for i in range(0, len(cuts) - 1, 2) if cuts[i + 1] - 1 >= cuts[i]]
        if edge_cases and qi % 3 == 0 and iv:
            iv[0] = (0, iv[0][1])                          # the query starts masked
        if edge_cases and qi % 4 == 1 and iv:
            iv[-1] = (iv[-1][0], n - 1)                    # ... or ends masked
        if edge_cases and qi % 5 == 2 and len(iv) >= 2 and iv[0][1] + 1 < iv[1][0]:
            iv[1] = (iv[0][1] + 1, iv[1][1])               # two abutting masks
        masks += [(qi, a, b) for a, b in iv]
    return masks


@pytest.mark.parametrize("task,nq,kw", [
    ("megablast", 1, {}),                                   # small table, lut 8, stride 21
    ("megablast", 16, {}),                                  # megablast table lut 11
    ("megablast", 160, {}),                                 # lut 12: partitioned scan
    ("megablast", 16, dict(word_size=16)),
    ("blastn", 4, {}),                                      # lut 8, word 11
    ("blastn", 8, {}),                                      # lut 11 = word size: no re-check needed
    ("blastn", 8, dict(word_size=15)),
])
def test_soft_query_masks(task, nq, kw):
    rng = np.random.default_rng(nq * 31 + len(task))
    db, queries, plants, subjects, opt = util.small_case(6, 100_000, nq, task=task, planted_fraction=1.0, **kw)
    masks = random_masks(rng, queries)
    src = api.BlastSeqSrc.from_packed(subjects)
    ps = api.BlastPrelimSearch(queries, opt, src, masks=masks)
    gpu = ps.run(keep_stages=True)
    ora, s = util.oracle_run(opt, queries, subjects, masks=masks)
    oi, gi = s.info(), ps.info()
    assert (oi["lut_type"], oi["lut_width"], oi["scan_step"]) == (gi["lut_type"], gi["lut_width"], gi["scan_step"])
    util.compare_stages(gpu, ora)
    d = ps.diagnostics
    assert (d.lookup_hits, d.good_init_extends, d.gapped_extensions, d.good_extensions) == \
           (s.stats.lookup_hits, s.stats.good_init_extends, s.stats.gapped_extensions, s.stats.good_extensions)
    # masking must matter in this case: fewer seeds than without masks, and still some HSPs
    plain = api.BlastPrelimSearch(queries, opt, src).run(keep_stages=True)
    assert len(gpu["seeds"]) < len(plain["seeds"])
    assert len(gpu["hsps"]) >= 1


def test_set_up_buffers_handed_on_from_batch_to_batch():
    """A batch's concatenation buffer and its pinned staging buffer are kept when the batch goes and handed to the next one
    (from 1 MB up: batch.cpp qbuf_take, Engine::stage_idle).  A batch of other queries, lengths and masks in buffers a longer
    one has just left behind: same stages as the oracle's -- every byte that is not a query base (pads, separators) is written
    by the set-up itself, nothing of the batch before shows through."""
    rng = np.random.default_rng(77)
    db, qa, _, subjects, opt = util.small_case(4, 60_000, 700, qlen=1000, seed=3, planted_fraction=0.2)
    src = api.BlastSeqSrc.from_packed(subjects)
    for q in qa[::7]:
        q[rng.integers(0, len(q), 5)] = 14                          # ambiguity codes in the first batch's buffer
    a = api.BlastPrelimSearch(qa, opt, src)
    a.run(); a.close()
    _, qb, _, _, _ = util.small_case(4, 60_000, 560, qlen=1100, seed=3, planted_fraction=0.5)
    qb = [q[: int(rng.integers(800, 1100))] for q in qb]           # ragged: other offsets, a buffer three quarters as long (within the hand-on rule)
    masks = [(i, 10, 40) for i in range(0, len(qb), 5)]
    b = api.BlastPrelimSearch(qb, opt, src, masks=masks)
    gpu = b.run(keep_stages=True)
    ora, s = util.oracle_run(opt, qb, subjects, masks=masks)
    util.compare_stages(gpu, ora)
    assert len(gpu["hsps"]) >= 50
    d = b.diagnostics
    assert (d.lookup_hits, d.good_init_extends) == (s.stats.lookup_hits, s.stats.good_init_extends)
    b.close(); src.close()


@pytest.mark.parametrize("task,period", [("blastn", 12), ("blastn", 14), ("megablast", 30)])
def test_query_masks_every_few_bases(task, period):
    """One masked base every `period`: 70-85 indexed stretches begin inside one 1,024-position block of the table
    builder's enumeration kernel (more than the 64 whose starts it keeps in LDS: its search-on branch), each just
    long enough for a word (blastn: 11 and 13 bases for word size 11; megablast: 29 for 28)."""
    db, queries, plants, subjects, opt = util.small_case(6, 100_000, 6, task=task, planted_fraction=1.0)
    masks = [(qi, a, a) for qi, q in enumerate(queries) for a in range(qi % period, len(q), period)]
    src = api.BlastSeqSrc.from_packed(subjects)
    ps = api.BlastPrelimSearch(queries, opt, src, masks=masks)
    gpu = ps.run(keep_stages=True)
    ora, s = util.oracle_run(opt, queries, subjects, masks=masks)
    oi, gi = s.info(), ps.info()
    assert (oi["lut_type"], oi["lut_width"], oi["scan_step"]) == (gi["lut_type"], gi["lut_width"], gi["scan_step"])
    util.compare_stages(gpu, ora)
    assert len(gpu["seeds"]) >= 1
    d = ps.diagnostics
    assert (d.lookup_hits, d.good_init_extends) == (s.stats.lookup_hits, s.stats.good_init_extends)


@pytest.mark.parametrize("task", ["megablast", "blastn"])
def test_default_dust_filtering_end_to_end(task):
    """blastn's default: DUST the queries, mask at hash.  Queries and subjects share low-complexity
    stretches (poly-A, CA repeats) that would seed everywhere without the filter."""
    from oracle import orc
    rng = np.random.default_rng(11)
    rep = [np.zeros(60, dtype=np.uint8), np.tile(np.array([1, 0], dtype=np.uint8), 40),
           np.tile(np.array([2, 3, 3], dtype=np.uint8), 25)]
    subs = []
    for i in range(5):
        s = rng.integers(0, 4, 60_000, dtype=np.uint8)
        for k in range(12):
            p = int(rng.integers(0, len(s) - 200)); r = rep[k % 3]; s[p:p + len(r)] = r
        subs.append(s)
    queries = []
    for i in range(6):
        src = subs[i % 5]; a = int(rng.integers(1000, 50_000))
        q = src[a:a + 900].copy()
        for k in range(2):
            p = int(rng.integers(50, 800)); r = rep[(i + k) % 3]; q[p:p + len(r)] = r[:len(q) - p]
        q[rng.integers(0, 900, 15)] = rng.integers(0, 4, 15)
        queries.append(q)
    subjects = [(orc.pack_ncbi2na(s), len(s)) for s in subs]
    opt = api.default_options(task, db_length=sum(len(s) for s in subs), db_num_seqs=len(subs))
    masks = api.dust_masks(queries)
    assert len(masks) >= 6
    srcdb = api.BlastSeqSrc.from_packed(subjects)
    ps = api.BlastPrelimSearch(queries, opt, srcdb, masks=masks)
    gpu = ps.run(keep_stages=True)
    omasks = [(qi, a, b) for qi, q in enumerate(queries) for a, b in orc.dust(q)]
    assert omasks == masks
    ora, s = util.oracle_run(opt, queries, subjects, masks=omasks)
    util.compare_stages(gpu, ora)
    unfiltered = api.BlastPrelimSearch(queries, opt, srcdb).run(keep_stages=True)
    assert len(gpu["seeds"]) < len(unfiltered["seeds"]) and len(gpu["hsps"]) >= 6


@pytest.mark.parametrize("split_mb", ["256", "1"])
def test_skewed_subjects_fill_few_bins(split_mb, monkeypatch):
    """Real genomes have megabase satellite arrays: whole tiles of scan positions fall into one or two
    bins of the partitioned scan (hundreds of complete lines of one bin per tile, empty runs for all
    other bins over many consecutive tiles).  Subjects: long poly-A / short-period repeats with
    unique islands that the queries hit; 200 queries -> lut 12, 512 bins.  The engine notices that the
    streams of a few bins overflow, halves the subject range until the repeat-rich subjects are in
    small ranges (threshold GBN_SKEW_SPLIT_MB) and scans those with the direct-probe kernel; two random
    subjects in the shard stay on the partitioned path."""
    from oracle import orc
    monkeypatch.setenv("GBN_SKEW_SPLIT_MB", split_mb)
    rng = np.random.default_rng(77)
    subs, islands = [], []
    for i in range(4):
        parts, pos = [], 0
        for k in range(10):
            kind = (i + k) % 3
            n = int(rng.integers(150_000, 400_000))
            if kind == 0:
                parts.append(np.zeros(n, dtype=np.uint8))                                   # poly-A: one lookup word only
            elif kind == 1:
                parts.append(np.tile(rng.integers(0, 4, int(rng.integers(2, 40)), dtype=np.uint8), n // 2 + 40)[:n])   # tandem repeat
            else:
                parts.append(rng.integers(0, 4, n // 8, dtype=np.uint8))                    # unique island
                islands.append((i, pos, len(parts[-1])))
            pos += len(parts[-1])
        subs.append(np.concatenate(parts))
    subs.insert(0, rng.integers(0, 4, 2_000_000, dtype=np.uint8))          # two ordinary subjects around them
    subs.append(rng.integers(0, 4, 2_000_000, dtype=np.uint8))
    islands = [(i + 1, p, n) for i, p, n in islands] + [(0, 0, 2_000_000), (5, 0, 2_000_000)]
    queries = []
    for qi in range(200):
        if qi % 2 == 0:                                       # a mutated piece of a unique island
            si, p0, ln = islands[qi % len(islands)]
            a = p0 + int(rng.integers(0, ln - 1000))
            q = subs[si][a:a + 1000].copy()
            q[rng.integers(0, 1000, 20)] = rng.integers(0, 4, 20)
        else:
            q = rng.integers(0, 4, 1000, dtype=np.uint8)
        queries.append(q)
    subjects = [(orc.pack_ncbi2na(s), len(s)) for s in subs]
    opt = api.default_options("megablast", db_length=sum(len(s) for s in subs), db_num_seqs=len(subs))
    src = api.BlastSeqSrc.from_packed(subjects)
    ps = api.BlastPrelimSearch(queries, opt, src)
    assert ps.info()["lut_width"] == 12
    gpu = ps.run(keep_stages=True)
    ora, s = util.oracle_run(opt, queries, subjects)
    util.compare_stages(gpu, ora)
    assert ps.diagnostics.lookup_hits == s.stats.lookup_hits and len(gpu["hsps"]) >= 50


@pytest.mark.spawns
def test_device_built_lookup_tables_equal_the_host_builder():
    """The lookup structures are built on the device (lutbuild.hip); GBN_HOST_LOOKUP=1 selects the host
    builder they replaced.  Same HSPs, seeds and lookup-hit counts for every table kind: megablast chains
    (descending offsets), small-NA (one-byte and general extension), the standard table a crowded small-NA
    table falls back to, direct mode (lut = word), and a masked batch."""
    import subprocess, sys, os, json
    code = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
from gblastn_amd import api
from tests import util
out = []
db, queries, plants, subjects, opt = util.small_case(10, 300_000, 60)
src = api.BlastSeqSrc.from_packed(subjects)
cases = [("megablast", {}, queries), ("megablast", {"word_size": 16}, queries[:3]), ("blastn", {}, queries[:8]),
         ("blastn", {"word_size": 7}, queries[:2]), ("blastn", {}, queries), ("blastn", {"word_size": 12}, queries[:1])]
for task, kw, qs in cases:
    o = api.default_options(task, db_length=opt.db_length, db_num_seqs=opt.db_num_seqs, **kw)
    ps = api.BlastPrelimSearch(qs, o, src)
    h = ps.run()["hsps"]
    out.append([ps.info(), h.tobytes().hex(), int(ps.diagnostics.lookup_hits), int(ps.diagnostics.seeds)])
qs = queries[:6]
ps = api.BlastPrelimSearch(qs, api.default_options("blastn", db_length=opt.db_length, db_num_seqs=opt.db_num_seqs), src,
                           masks=[(0, 100, 400), (3, 0, 50), (3, 700, 900)])
h = ps.run()["hsps"]
out.append([ps.info(), h.tobytes().hex(), int(ps.diagnostics.lookup_hits), int(ps.diagnostics.seeds)])
print(json.dumps(out))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for flag in ("0", "1"):
        env = dict(os.environ); env["GBN_HOST_LOOKUP"] = flag
        p = util.run_child([sys.executable, "-c", code], env=env, timeout=900)
        res[flag] = json.loads(p.stdout.strip().splitlines()[-1])
    kinds = set()
    for dev, host in zip(res["0"], res["1"]):
        assert dev == host, (dev[0], host[0])
        kinds.add((dev[0]["lut_type"], dev[0]["lut_width"]))
    assert len(kinds) >= 3 and sum(len(r[1]) for r in res["0"]) > 0


import os as _os
_FUZZ = [101, 202, 303, 404, 505, 606] + list(range(int(_os.environ.get("GBN_FUZZ_BASE", "1000")), int(_os.environ.get("GBN_FUZZ_BASE", "1000")) + int(_os.environ.get("GBN_FUZZ_EXTRA", "0"))))


@pytest.mark.parametrize("seed", _FUZZ)
def test_randomised_shapes_against_the_oracle(seed, monkeypatch):
    """Random task / word size / scoring / batch size / subject count per seed, every stage compared with the
    oracle: exercises the device-built tables of every kind, the stride variants of the binning kernel
    (1, 2, 4, 17, 18, 21 and the generic one), 2..512 bins, masks, and both gapped kernels."""
    rng = np.random.default_rng(seed)
    for _ in range(5):
        task = "megablast" if rng.random() < 0.5 else "blastn"
        nq = int(rng.choice([1, 2, 7, 30, 120]))
        nsub, slen = int(rng.integers(2, 9)), int(rng.choice([30_000, 120_000, 400_000]))
        kw = {}
        if task == "megablast":
            kw["word_size"] = int(rng.choice([28, 28, 20, 32, 16, 48]))
        else:
            kw["word_size"] = int(rng.choice([11, 11, 9, 7, 13, 15]))
            if rng.random() < 0.3:
                kw.update(reward=1, penalty=-3, gap_open=5, gap_extend=2)
        db, queries, plants, subjects, opt = util.small_case(nsub, slen, nq, seed=int(rng.integers(1, 1 << 30)),
                                                             planted_fraction=0.6, task=task, **kw)
        masks = None
        if rng.random() < 0.4:
            masks = sorted((int(q), int(a), int(a + rng.integers(5, 200))) for q, a in
                           zip(rng.choice(nq, min(nq, 3), replace=False), rng.integers(0, 700, 3)))
        src = api.BlastSeqSrc.from_packed(subjects)
        ps = api.BlastPrelimSearch(queries, opt, src, masks=masks)
        got = ps.run(keep_stages=True)
        ora, osearch = util.oracle_run(opt, queries, subjects, masks=masks)
        util.compare_stages(got, ora)
        d, st = ps.diagnostics, osearch.stats
        assert (d.lookup_hits, d.good_init_extends, d.gapped_extensions, d.good_extensions) == \
               (st.lookup_hits, st.good_init_extends, st.gapped_extensions, st.good_extensions), (task, kw, ps.info())
        # the pipelined entry points over subject ranges of a few tiles: seed + extension stages of a range
        # on the second stream underneath the scan of the next one
        monkeypatch.setenv("GBN_RANGE_TILES", str(int(rng.integers(2, 9))))
        ps.begin()
        assert np.array_equal(ps.end()["hsps"], got["hsps"]), (task, kw, ps.info())
        monkeypatch.delenv("GBN_RANGE_TILES")


def test_slice_scan_with_repeats_that_overflow_a_segment():
    """blastn W=11 with a table as wide as the word: scan_slice_kernel (the presence bits sliced through the LDS, every
    workgroup writing seeds into a segment of its own sized for random subjects).  A subject that is mostly one
    short-period repeat which the queries carry too gives a hundred times the seeds a random subject would: the
    segments overflow, the engine scans again with longer ones.  Stage by stage against the oracle; a second search
    over random subjects only takes the short segments again."""
    from oracle import orc
    rng = np.random.default_rng(2024)
    unit = rng.integers(0, 4, 7, dtype=np.uint8)
    queries = [rng.integers(0, 4, 1000, dtype=np.uint8) for _ in range(12)]
    for qi in (0, 3, 7):
        queries[qi][200:500] = np.tile(unit, 50)[:300]
    rep = rng.integers(0, 4, 60_000, dtype=np.uint8)
    rep[5_000:45_000] = np.tile(unit, 6000)[:40_000]                     # 40 kb of the repeat
    rep[50_000:50_600] = queries[5][100:700]
    plain = [rng.integers(0, 4, 40_000, dtype=np.uint8) for _ in range(3)]
    plain[1][1000:1700] = queries[9][200:900]
    subs = [plain[0], rep, plain[1], plain[2]]
    subjects = [(orc.pack_ncbi2na(s), len(s)) for s in subs]
    opt = api.default_options("blastn", db_length=sum(len(s) for s in subs), db_num_seqs=len(subs))
    ps = api.BlastPrelimSearch(queries, opt, api.BlastSeqSrc.from_packed(subjects))
    info = ps.info()
    sliced = os.environ.get("GBN_SCAN_BINS") != "1"                    # (the fixture's other leg: the direct-probe kernel)
    assert (info["lut_width"], info["scan_step"], info["scan_path"]) == (11, 1, 2 if sliced else 1), info
    gpu = ps.run(keep_stages=True)
    ora, s = util.oracle_run(opt, queries, subjects)
    util.compare_stages(gpu, ora)
    assert len(gpu["seeds"]) > 1_000_000 and ps.diagnostics.lookup_hits == s.stats.lookup_hits
    assert ps.diagnostics.scan_launches >= (2 if sliced else 1)         # the repeat made the engine scan twice
    ps.begin(); assert ps.end()["hsps"].tobytes() == gpu["hsps"].tobytes()
    ps2 = api.BlastPrelimSearch(queries, opt, api.BlastSeqSrc.from_packed([subjects[0], subjects[2], subjects[3]]))
    gpu2 = ps2.run(keep_stages=True)
    ora2, s2 = util.oracle_run(opt, queries, [subjects[0], subjects[2], subjects[3]])
    util.compare_stages(gpu2, ora2)
    assert ps2.diagnostics.scan_launches == 1


@pytest.mark.parametrize("task", ["megablast", "blastn"])
def test_tables_of_many_cells_and_few_words(task):
    """A batch long enough for a table of 4^11 cells in which most queries are nothing but N: a few thousand words among four
    million cells, long runs of empty cells (the device builder reads cell_start off the sorted word list and fills the
    runs a wave at a time, lutbuild.hip), and one query batch without a single word.  Stages and counters equal the oracle's."""
    rng = np.random.default_rng(5 if task == "megablast" else 6)
    db, queries, plants, subjects, opt = util.small_case(5, 120_000, 30, task=task, seed=21)
    for i in range(len(queries)):
        if i % 6:
            queries[i] = np.full(len(queries[i]), 14, dtype=np.uint8)            # N throughout
    src = api.BlastSeqSrc.from_packed(subjects)
    ps = api.BlastPrelimSearch(queries, opt, src)
    if task == "megablast":                                                     # (blastn counts the words before it chooses: a table of 4^8 cells)
        assert ps.info()["lut_width"] >= 10, ps.info()
    gpu = ps.run(keep_stages=True)
    ora, s = util.oracle_run(opt, queries, subjects)
    util.compare_stages(gpu, ora)
    assert ps.diagnostics.lookup_hits == s.stats.lookup_hits and len(gpu["hsps"]) > 0
    ps.close()
    # (a batch of nothing but N has no valid context: the set-up refuses it as BLAST_MainSetUp does; one query of 29 good bases
    # among them makes a batch whose table has two words)
    nothing = [np.full(1000, 14, dtype=np.uint8) for _ in range(30)]
    with pytest.raises(api.BlastError):
        api.BlastPrelimSearch(nothing, opt, src)
    nothing[7] = nothing[7].copy(); nothing[7][400:429] = rng.integers(0, 4, 29, dtype=np.uint8)
    ps = api.BlastPrelimSearch(nothing, opt, src)
    gpu = ps.run(keep_stages=True)
    ora, s = util.oracle_run(opt, nothing, subjects)
    util.compare_stages(gpu, ora)
    assert ps.diagnostics.lookup_hits == s.stats.lookup_hits
    ps.close()


def test_pipelined_passes_bin_ahead_for_one_another(monkeypatch):
    """Record cache OFF (the north_star scan: every pass bins): gbn_prelim_search_begin over one range of a shard with a
    megablast-shaped batch queues the NEXT pass's binning kernel (the records depend on the shard and the table's shape only)
    behind its own kernels -- once the pass before it had the same shape (a repeat has been seen); a next pass of the same shape
    finds its records there, one of another shape does not and bins for itself (a miss, after which nothing is queued ahead until
    a shape repeats).  Passes of two shapes in turn: every pass gives what a search on its own gives, and the passes that
    could use the records binned ahead did (GBN_BIN_AHEAD=0: none)."""
    monkeypatch.setenv("GBN_RECORD_CACHE_MB", "0")
    db, queries, plants, subjects, opt = util.small_case(8, 300_000, 40, task="megablast", seed=11)
    src = api.BlastSeqSrc.from_packed(subjects)
    qa, qb, qc = queries[:18], queries[18:36], queries[36:40]        # 18 kb, 18 kb (lut 11, stride 18) and 4 kb (another table)
    want = {}
    shapes = {}
    for name, q in (("a", qa), ("b", qb), ("c", qc)):
        ps = api.BlastPrelimSearch(q, opt, src)
        shapes[name] = (ps.info()["lut_width"], ps.info()["scan_step"], ps.info()["scan_path"])
        want[name] = ps.run()["hsps"].tobytes()
        ps.close()
    assert shapes["a"] == shapes["b"] != shapes["c"], shapes
    # (the direct-probe leg of the fixture has nothing to bin ahead)
    binned = shapes["a"][2] == 0 and os.environ.get("GBN_SCAN_BINS", "0") in ("", "0")
    L = api.lib()
    for ahead in ("1", "0"):
        monkeypatch.setenv("GBN_BIN_AHEAD", ahead)
        h0, m0 = L.gbn_debug_bin_ahead_hits(), L.gbn_debug_bin_ahead_misses()
        prev = None
        for name, q in (("a", qa), ("b", qb), ("a", qa), ("b", qb), ("c", qc), ("b", qb), ("a", qa), ("a", qa)):
            ps = api.BlastPrelimSearch(q, opt, src)
            ps.begin()
            if prev is not None:
                assert prev[1].end()["hsps"].tobytes() == want[prev[0]], (ahead, prev[0])
                prev[1].close()
            prev = (name, ps)
        assert prev[1].end()["hsps"].tobytes() == want[prev[0]]
        prev[1].close()
        hits, misses = L.gbn_debug_bin_ahead_hits() - h0, L.gbn_debug_bin_ahead_misses() - m0
        # a (first of its shape) b (a repeat: bins ahead) a (hit) b (hit) c (miss: nobody wants b's records) b (no repeat seen) a (a
        # repeat: bins ahead) a (hit); with the switch off only the records the first leg's last pass left behind are used
        assert hits == ((3 if ahead == "1" else 1) if binned else 0), (ahead, hits, shapes)
        assert misses == ((1 if ahead == "1" else 0) if binned else 0), (ahead, misses)
    # the shard goes away while a binning kernel queued ahead may still read it; a search over another shard follows at once
    monkeypatch.setenv("GBN_BIN_AHEAD", "1")
    for _ in range(2):
        ps = api.BlastPrelimSearch(qa, opt, src)
        ps.begin(); assert ps.end()["hsps"].tobytes() == want["a"]
        ps.close()
    src.close()
    db2, queries2, _, subjects2, opt2 = util.small_case(5, 200_000, 18, task="megablast", seed=12)
    src2 = api.BlastSeqSrc.from_packed(subjects2)
    ps2 = api.BlastPrelimSearch(queries2, opt2, src2)
    ps2.begin(); got2 = ps2.end()
    ora2, _ = util.oracle_run(opt2, queries2, subjects2)
    util.compare_stages(got2, ora2)
    ps2.close()


def test_binned_ahead_then_a_slice_scan_with_few_seeds(monkeypatch):
    """Record cache off.  Pipelined passes of a binned blastn shape (small batches: lut 8) bin ahead for one another; then a
    larger batch whose table is as wide as the word (lut 11 = word 11: the slice scan, seeds left in per-workgroup
    segments) with fewer than 2^20 seeds follows on the same engine.  Its seeds are put back to back on the engine's stream
    and copied for the asynchronous stage BEHIND that kernel (round 4 copied them on the second stream, unordered against it,
    whenever a binning kernel had been queued ahead).  Every pass equals a search on its own."""
    from oracle import orc
    monkeypatch.setenv("GBN_RECORD_CACHE_MB", "0")
    monkeypatch.setenv("GBN_BIN_AHEAD", "1")
    monkeypatch.delenv("GBN_SCAN_BINS", raising=False)
    rng = np.random.default_rng(31)
    subs = [rng.integers(0, 4, 40_000, dtype=np.uint8) for _ in range(6)]
    small = [rng.integers(0, 4, 700, dtype=np.uint8) for _ in range(3)]
    big = [rng.integers(0, 4, 1000, dtype=np.uint8) for _ in range(14)]
    for k, q in enumerate(small + big):
        subs[k % 6][1000 + 300 * k:1000 + 300 * k + 250] = q[50:300]
    subjects = [(orc.pack_ncbi2na(x), len(x)) for x in subs]
    opt = api.default_options("blastn", db_length=sum(len(x) for x in subs), db_num_seqs=len(subs))
    src = api.BlastSeqSrc.from_packed(subjects)
    want = {}
    for name, q in (("small", small), ("big", big)):
        ps = api.BlastPrelimSearch(q, opt, src)
        want[name] = (ps.run()["hsps"].tobytes(), ps.info()["scan_path"])
        ps.close()
    assert want["small"][1] in (0, 1) and want["big"][1] == 2, (want["small"][1], want["big"][1])
    prev = None
    for name, q in (("small", small), ("small", small), ("small", small), ("big", big), ("small", small), ("big", big)):
        ps = api.BlastPrelimSearch(q, opt, src)
        ps.begin()
        if prev is not None:
            assert prev[1].end()["hsps"].tobytes() == want[prev[0]][0], prev[0]
            prev[1].close()
        prev = (name, ps)
    assert prev[1].end()["hsps"].tobytes() == want[prev[0]][0]
    prev[1].close()


def test_seeds_ordered_by_the_counting_sort_and_by_the_library_sort(monkeypatch):
    """blastn W=11, table as wide as the word: the ordered scan's seeds go to the diagonal filter through seed_order.hip (a
    counting sort per subject and slot, keys built on the way; default) or through seed_ckeys_kernel + the library's radix
    sort (GBN_SEED_ORDER=0, rounds 2-3).  Subjects of very different lengths, some too short for a word, a homolog in every
    fifth: the HSPs of both ways are the oracle's, and the kernel-class timers say which way ran."""
    from oracle import orc
    rng = np.random.default_rng(77)
    queries = [rng.integers(0, 4, int(n), dtype=np.uint8) for n in rng.integers(600, 1400, 14)]
    lens = [int(x) for x in rng.integers(3, 30_000, 40)]
    lens[3], lens[17], lens[39] = 5, 10, 11
    subs = [rng.integers(0, 4, n, dtype=np.uint8) for n in lens]
    for i in range(0, 40, 5):
        if lens[i] > 2000:
            q = queries[i % len(queries)]
            subs[i][500:500 + 400] = q[100:500]
    subjects = [(orc.pack_ncbi2na(x), len(x)) for x in subs]
    opt = api.default_options("blastn", db_length=sum(lens), db_num_seqs=len(subs))
    ora, osearch = util.oracle_run(opt, queries, subjects)
    monkeypatch.setenv("GBN_DIAG_COMPACT_MIN", "1")
    monkeypatch.delenv("GBN_SCAN_BINS", raising=False)
    seen = {}
    for order in ("1", "0"):
        monkeypatch.setenv("GBN_SEED_ORDER", order)
        ps = api.BlastPrelimSearch(queries, opt, api.BlastSeqSrc.from_packed(subjects))
        info = ps.info()
        assert (info["lut_width"], info["scan_step"], info["scan_path"]) == (11, 1, 2), info
        gpu = ps.run(keep_stages=False)
        util.compare_stages(gpu, ora)
        km = dict(zip(api.GbnDiagnostics.KERNEL_CLASSES, list(ps.diagnostics.kernel_ms)))
        seen[order] = (gpu["hsps"].tobytes(), km, int(ps.diagnostics.library_sorts), int(ps.diagnostics.ranges))
        ps.close()
    assert seen["1"][0] == seen["0"][0] and len(seen["1"][0]) > 0
    # the counting sort never calls the radix sort; the other path calls it at least once per range.  (The key kernel's timer
    # is no witness of that path any more: when the scan wrote composite keys itself -- GBN_SEED_KEYS, buffers permitting --
    # no key kernel runs on either path.)
    keys = [k for k in api.GbnDiagnostics.KERNEL_CLASSES if "key" in k][0]
    assert seen["1"][2] == 0 and seen["1"][1][keys] == 0.0, seen["1"]
    assert seen["0"][2] >= seen["0"][3] >= 1, seen["0"]


@pytest.mark.parametrize("task,period", [("megablast", 5), ("megablast", 13), ("blastn", 3)])
def test_repeats_in_the_queries_give_long_cells(task, period):
    """Queries that carry a short-period repeat put hundreds of offsets into a few cells of the table, in descending
    order for the megablast table and ascending for the others (the device builder's sort key: lutbuild.hip); the seed
    list -- in the reference's order, which is the order inside the cells -- the initial hits and the HSPs equal the
    oracle's.  (Written with round 4's counting-sort builder, which passed it and was not kept: profiles/r04_lut_builder.txt.)"""
    from oracle import orc
    rng = np.random.default_rng(100 + period)
    unit = rng.integers(0, 4, period, dtype=np.uint8)
    nq = 24 if task == "megablast" else 14
    queries = [rng.integers(0, 4, 1000, dtype=np.uint8) for _ in range(nq)]
    for qi in range(0, nq, 3):
        queries[qi][100:900] = np.tile(unit, 800 // period + 1)[:800]
    subj = rng.integers(0, 4, 120_000, dtype=np.uint8)
    subj[10_000:10_400] = np.tile(unit, 400 // period + 1)[:400]         # the repeat: every seed of it hits the long cells
    subj[50_000:50_700] = queries[1][150:850]
    subjects = [(orc.pack_ncbi2na(subj), len(subj))]
    opt = api.default_options(task, db_length=len(subj), db_num_seqs=1)
    ps = api.BlastPrelimSearch(queries, opt, api.BlastSeqSrc.from_packed(subjects))
    gpu = ps.run(keep_stages=True)
    ora, s = util.oracle_run(opt, queries, subjects)
    util.compare_stages(gpu, ora)
    assert ps.diagnostics.lookup_hits == s.stats.lookup_hits and len(gpu["seeds"]) > 500


def test_rare_queue_segments_that_overflow_are_scanned_again(monkeypatch):
    """The probe kernel queues its survivors in a segment per workgroup, sized for random subjects; a segment that overflows
    (repeat-rich ranges at full size) is counted past its end and the range is scanned again with the room the counts ask
    for.  GBN_RARE_SEG=1 makes a small search take that way: same stages as the oracle's, one scan launch more per range --
    single searches and pipelined passes (whose binning kernel may have run ahead) alike."""
    db, queries, plants, subjects, opt = util.small_case(8, 300_000, 40, task="megablast", seed=5, planted_fraction=0.8)
    src = api.BlastSeqSrc.from_packed(subjects)
    ora, s = util.oracle_run(opt, queries, subjects)
    ps = api.BlastPrelimSearch(queries, opt, src)
    binned = ps.info()["scan_path"] == 0 and os.environ.get("GBN_SCAN_BINS", "0") in ("", "0")
    gpu = ps.run(keep_stages=True)
    util.compare_stages(gpu, ora)
    base = ps.diagnostics.scan_launches
    want = gpu["hsps"].tobytes()
    ps.close()
    monkeypatch.setenv("GBN_RARE_SEG", "1")
    ps = api.BlastPrelimSearch(queries, opt, src)
    gpu = ps.run(keep_stages=True)
    util.compare_stages(gpu, ora)
    if binned:
        assert ps.diagnostics.scan_launches > base, (ps.diagnostics.scan_launches, base)
    for _ in range(3):                                  # pipelined passes
        ps.begin(); assert ps.end()["hsps"].tobytes() == want
    ps.close(); src.close()


@pytest.mark.parametrize("wave", ["1", "0"])
def test_greedy_extension_a_wave_per_hit_and_its_fallback(wave, monkeypatch):
    """megablast, gap costs 0 / 0: greedy_wave_kernel (a workgroup of two waves per initial hit, the diagonals of a distance
    across the lanes, offsets in LDS) against the oracle -- homologs of every kind on both strands: exact, substitutions only,
    indels, ends of the query and of the subject inside the alignment, an ambiguity code in the query, and long diverged ones
    (12 kb at 4-6 % differences: more distance than the kernel's LDS window holds, so those halves go to the thread-per-hit
    kernel behind the GBN_GAP_REDO mark).  GBN_GREEDY_WAVE=0: the thread-per-hit kernel alone; both equal the oracle."""
    from oracle import orc
    monkeypatch.setenv("GBN_GREEDY_WAVE", wave)
    rng = np.random.default_rng(123)
    comp = np.array([3, 2, 1, 0], dtype=np.uint8)

    def mutate(x, sub, indel):
        y = x.copy()
        m = rng.random(len(y)) < sub
        y[m] = (y[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) & 3
        out, i = [], 0
        for pos in sorted(rng.choice(len(y), indel, replace=False).tolist()) if indel else []:
            out.append(y[i:pos]); i = pos
            if rng.random() < 0.5:
                out.append(rng.integers(0, 4, int(rng.integers(1, 4)), dtype=np.uint8))      # insertion
            else:
                i = min(len(y), pos + int(rng.integers(1, 4)))                                # deletion
        out.append(y[i:])
        return np.concatenate(out)
    nsub, slen = 10, 60_000
    subs = [rng.integers(0, 4, slen, dtype=np.uint8) for _ in range(nsub)]
    queries = []
    plans = [(900, 0.0, 0), (900, 0.03, 0), (900, 0.02, 3), (1500, 0.05, 6), (700, 0.0, 1), (12_000, 0.04, 10), (12_000, 0.06, 25), (3000, 0.08, 0),
             (900, 0.01, 2), (900, 0.05, 2), (2500, 0.03, 8), (600, 0.0, 0)]
    for k, (n, sub, indel) in enumerate(plans):
        s = subs[k % nsub]
        at = int(rng.integers(0, slen - n))
        if k == 4:
            at = 0                                                   # the subject's start inside the alignment
        if k == 8:
            at = slen - n                                            # ... and its end
        piece = mutate(s[at:at + n], sub, indel)
        if k % 2:
            piece = comp[piece[::-1]]
        flank = int(rng.integers(0, 300)) if k not in (0, 11) else 0  # (no flank: the query's ends inside the alignment)
        qq = np.concatenate([rng.integers(0, 4, flank, dtype=np.uint8), piece, rng.integers(0, 4, flank, dtype=np.uint8)])
        if k == 9:
            qq[len(qq) // 2] = 14                                     # an ambiguity code (N) in the middle
        queries.append(qq)
    subjects = [(orc.pack_ncbi2na(x), len(x)) for x in subs]
    opt = api.default_options("megablast", db_length=nsub * slen, db_num_seqs=nsub)
    src = api.BlastSeqSrc.from_packed(subjects)
    ps = api.BlastPrelimSearch(queries, opt, src)
    gpu = ps.run(keep_stages=True)
    ora, s = util.oracle_run(opt, queries, subjects)
    util.compare_stages(gpu, ora)
    assert len(gpu["hsps"]) >= len(plans) - 1
    assert ps.diagnostics.gapped_extensions == s.stats.gapped_extensions
    km = dict(zip(api.GbnDiagnostics.KERNEL_CLASSES, list(ps.diagnostics.kernel_ms)))
    assert (km["dynprog_wave_kernel / greedy_wave_kernel"] > 0) == (wave == "1")
    ps.close(); src.close()


@pytest.mark.parametrize("case", ["mb_hash", "smallna_array", "mb_lut11", "ragged"])
def test_few_seeds_ordered_by_one_workgroup(case, monkeypatch):
    """Without keep_stages the few seeds of a megablast-shaped search go to the diagonal filter through seed_sort.hip (one
    launch of one workgroup: indices sorted by subject | slot | scan position | query key, never stored); GBN_SMALL_SORT=0 takes
    the two library sorts of rounds 1-4.  Both give the oracle's HSPs -- hash and array container, tables whose chains come
    out descending (megablast) and ascending (small-NA), subjects of very different lengths with repeats of the query in them
    (many seeds of one diagonal slot, several per scan position)."""
    from oracle import orc
    rng = np.random.default_rng({"mb_hash": 1, "smallna_array": 2, "mb_lut11": 3, "ragged": 4}[case])
    if case == "smallna_array":
        nq, qlen, nsub, slen = 1, 1000, 6, 200_000          # C1's shape: small-NA table, diagonal array
    elif case == "mb_lut11":
        nq, qlen, nsub, slen = 20, 1000, 8, 300_000
    elif case == "ragged":
        nq, qlen, nsub, slen = 40, 900, 30, 0
    else:
        nq, qlen, nsub, slen = 400, 1000, 10, 400_000
    queries = [rng.integers(0, 4, qlen, dtype=np.uint8) for _ in range(nq)]
    lens = [int(x) for x in rng.integers(40, 120_000, nsub)] if case == "ragged" else [slen] * nsub
    subs = [rng.integers(0, 4, n, dtype=np.uint8) for n in lens]
    for k in range(min(max(nq, 8), 60)):
        s = subs[k % nsub]
        if len(s) < 2000:
            continue
        q = queries[k % nq]
        piece = q[100:100 + 600].copy()
        m = rng.random(len(piece)) < 0.02
        piece[m] = (piece[m] + 1) & 3
        for rep in range(1 + k % 3):                              # the same stretch several times: seeds of one slot at many scan positions
            at = int(rng.integers(0, len(s) - 700))
            s[at:at + len(piece)] = piece
    subjects = [(orc.pack_ncbi2na(x), len(x)) for x in subs]
    opt = api.default_options("megablast", db_length=sum(lens), db_num_seqs=nsub)
    ora, osearch = util.oracle_run(opt, queries, subjects)
    want = np.concatenate([o["hsps"] for o in ora]) if ora else None
    assert sum(len(o["hsps"]) for o in ora) >= 3
    src = api.BlastSeqSrc.from_packed(subjects)
    seen = {}
    for sw in ("1", "0"):
        monkeypatch.setenv("GBN_SMALL_SORT", sw)
        ps = api.BlastPrelimSearch(queries, opt, src)
        if case == "smallna_array":
            assert ps.info()["container"] == 0 and ps.info()["lut_type"] != 3
        got = ps.run()["hsps"]
        util.compare_stages({"hsps": got}, ora)
        ps.begin(); got2 = ps.end()["hsps"]
        assert got2.tobytes() == got.tobytes()
        km = dict(zip(api.GbnDiagnostics.KERNEL_CLASSES, list(ps.diagnostics.kernel_ms)))
        seen[sw] = (got.tobytes(), km["seed keys"])
        ps.close()
    assert seen["1"][0] == seen["0"][0]
    if not os.environ.get("GBN_DIAG_COMPACT_MIN"):                # (a run that sends every launch through the composite-key stage sorts that way whatever this switch says)
        assert seen["1"][1] == 0 and seen["0"][1] > 0             # (the key kernels belong to the library sorts only)
    src.close()
