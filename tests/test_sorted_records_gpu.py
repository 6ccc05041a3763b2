"""-m gpu: the SORTED form of a cached record set (csrc/scan_runs.hip) against the oracle and against
the stream form, through the C ABI.

A set that keeps being hit is sorted by cell once; later passes read the runs of the cells their batch occupies and
nothing else -- the presence test in front of the table access (MB_ACCESS_HITS / s_BlastMBLookupRetrieve,
CORE/blast_nascan.c:1413-1461).  Every case here runs a batch three times over one resident shard with
GBN_RUNS_AFTER=0: the first pass bins (stream form), the second finds the set complete and sorts it, the third probes
the runs again; seeds, initial hits, HSPs and the lookup-hit count of every pass equal the oracle's."""
import os

import numpy as np
import pytest

from gblastn_amd import api
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def sorted_at_first_hit(monkeypatch):
    monkeypatch.delenv("GBN_RECORD_CACHE_MB", raising=False)
    monkeypatch.delenv("GBN_SCAN_BINS", raising=False)
    monkeypatch.setenv("GBN_REC_RUNS", "2")                 # whatever the shard's size (the engine sorts only sets whose runs are long)
    monkeypatch.setenv("GBN_RUNS_AFTER", "0")
    api.record_cache_set_limit(-1)
    yield
    api.record_cache_set_limit(-1)


def three_passes(queries, opt, subjects, masks=None, expect_sorted=True):
    """miss (bins), hit (sorts, probes the runs), hit (probes the runs): each against the oracle"""
    src = api.BlastSeqSrc.from_packed(subjects)
    ora, s = util.oracle_run(opt, queries, subjects, masks=masks)
    ps = api.BlastPrelimSearch(queries, opt, src, masks=masks)
    partitioned = ps.info()["scan_path"] == 0
    st0 = api.record_cache_stats()
    first = None
    for k in range(3):
        ps.diagnostics = api.GbnDiagnostics()                               # (they add up over the runs of one search object)
        gpu = ps.run(keep_stages=True)
        util.compare_stages(gpu, ora)
        d = ps.diagnostics
        assert d.lookup_hits == s.stats.lookup_hits, (k, d.lookup_hits, s.stats.lookup_hits)
        assert (d.good_init_extends, d.gapped_extensions, d.good_extensions) == \
               (s.stats.good_init_extends, s.stats.gapped_extensions, s.stats.good_extensions)
        if first is None:
            first = gpu["hsps"].tobytes()
        assert gpu["hsps"].tobytes() == first
    st = api.record_cache_stats()
    if partitioned and expect_sorted:
        assert st["sorts"] - st0["sorts"] >= 1 and st["sorted_passes"] - st0["sorted_passes"] >= 2, (st0, st)
    ps.close(); src.close()
    return st["sorted_passes"] - st0["sorted_passes"]


@pytest.mark.parametrize("nsub,slen,nq,task,kw", [
    (6, 120_000, 160, "megablast", {}),                 # lut 12, stride 17, 512 bins (the C2 table)
    (8, 150_000, 16, "megablast", {}),                  # lut 11, stride 18, 128 bins
    (6, 200_000, 1, "megablast", {}),                   # lut 8, stride 21
    (4, 60_000, 2, "blastn", {}),                       # lut 8, stride 4
    (5, 300_000, 40, "megablast", dict(word_size=20)),  # lut 12 / 11 with a shorter word: other fingerprint widths
    (3, 400_000, 30, "megablast", dict(word_size=16)),
])
def test_table_shapes(nsub, slen, nq, task, kw):
    db, queries, plants, subjects, opt = util.small_case(nsub, slen, nq, task=task, planted_fraction=0.7, **kw)
    three_passes(queries, opt, subjects)


def test_repeats_in_the_queries_cells_of_many_entries():
    """cells with two entries (both fingerprints of the table word), with three and more (the side lists, read from global
    memory here) and cells that always take the rare path"""
    from oracle import orc
    rng = np.random.default_rng(31)
    unit = rng.integers(0, 4, 7, dtype=np.uint8)
    queries = [rng.integers(0, 4, 1000, dtype=np.uint8) for _ in range(180)]
    for qi in range(0, 180, 3):
        queries[qi][100:900] = np.tile(unit, 800 // 7 + 1)[:800]
    for qi in range(1, 180, 9):
        queries[qi] = queries[1].copy()                                 # twenty copies of one query: cells of twenty entries
    subj = [rng.integers(0, 4, 400_000, dtype=np.uint8) for _ in range(3)]
    subj[0][10_000:10_400] = np.tile(unit, 400 // 7 + 1)[:400]
    subj[1][50_000:50_700] = queries[1][150:850]
    subj[2][70_000:70_900] = queries[5][50:950]
    subjects = [(orc.pack_ncbi2na(x), len(x)) for x in subj]
    opt = api.default_options("megablast", db_length=sum(len(x) for x in subj), db_num_seqs=3)
    three_passes(queries, opt, subjects)


def test_ragged_subjects_and_masks():
    from tests.test_gpu_parity import ragged_case
    queries, subjects, opt = ragged_case("megablast", 5)[:3]
    three_passes(queries, opt, subjects)
    db, queries, plants, subjects, opt = util.small_case(6, 200_000, 60, task="megablast", planted_fraction=0.8, seed=3)
    masks = [(0, 100, 300), (3, 0, 999), (7, 500, 520)]
    three_passes(queries, opt, subjects, masks=masks)


def test_skewed_subjects_one_cell_larger_than_a_round():
    """poly-A and short-period stretches: the records of one cell exceed what the place kernel stages per round (the
    single-cell path), sub-bins are sorted in rounds over ranges of their cells -- as long as the streams do not overflow
    the set is cached and sorted like any other"""
    from oracle import orc
    rng = np.random.default_rng(9)
    subs = []
    for i in range(3):
        a = rng.integers(0, 4, 1_500_000, dtype=np.uint8)
        a[200_000:200_000 + 60_000] = 0                                              # poly-A: 3,500 scan positions in one cell per stretch
        a[700_000:760_000] = np.tile(rng.integers(0, 4, 3, dtype=np.uint8), 20_000)
        subs.append(a)
    queries = [rng.integers(0, 4, 1000, dtype=np.uint8) for _ in range(170)]
    for qi in range(0, 170, 5):
        p = int(rng.integers(0, 1_400_000))
        queries[qi] = subs[qi % 3][p:p + 1000].copy()
    queries[3][200:260] = 0                                                          # the poly-A cell is occupied
    subjects = [(orc.pack_ncbi2na(x), len(x)) for x in subs]
    opt = api.default_options("megablast", db_length=sum(len(x) for x in subs), db_num_seqs=3)
    three_passes(queries, opt, subjects, expect_sorted=False)


def test_rare_queue_overflow_over_sorted_records(monkeypatch):
    db, queries, plants, subjects, opt = util.small_case(8, 300_000, 40, task="megablast", seed=5, planted_fraction=0.8)
    src = api.BlastSeqSrc.from_packed(subjects)
    ora, s = util.oracle_run(opt, queries, subjects)
    ps = api.BlastPrelimSearch(queries, opt, src)
    util.compare_stages(ps.run(keep_stages=True), ora)
    util.compare_stages(ps.run(keep_stages=True), ora)                  # sorted now
    ps.diagnostics = api.GbnDiagnostics()
    ps.run()
    base = ps.diagnostics.scan_launches
    monkeypatch.setenv("GBN_RARE_SEG", "1")
    st0 = api.record_cache_stats()
    ps.diagnostics = api.GbnDiagnostics()
    gpu = ps.run(keep_stages=True)
    util.compare_stages(gpu, ora)
    st = api.record_cache_stats()
    assert ps.diagnostics.scan_launches > base and st["hits"] - st0["hits"] == 1 and st["sorted_passes"] - st0["sorted_passes"] >= 2
    ps.close(); src.close()


def test_other_batches_over_one_sorted_set_and_the_stream_form_switch(monkeypatch):
    """batches of one table shape share the sorted set; a batch that occupies a handful of cells reads a handful of runs;
    GBN_REC_RUNS=0 keeps the stream form (the A/B switch) and gives the same bytes"""
    db, queries, plants, subjects, opt = util.small_case(8, 300_000, 60, task="megablast", seed=11)
    src = api.BlastSeqSrc.from_packed(subjects)
    qa, qb, qc = queries[:24], queries[24:48], queries[48:60]
    api.record_cache_set_limit(0)
    want = {}
    for name, q in (("a", qa), ("b", qb), ("c", qc)):
        ps = api.BlastPrelimSearch(q, opt, src); want[name] = ps.run()["hsps"].tobytes(); ps.close()
    api.record_cache_set_limit(-1)
    st0 = api.record_cache_stats()
    for name, q in (("a", qa), ("b", qb), ("a", qa), ("b", qb)):
        ps = api.BlastPrelimSearch(q, opt, src); assert ps.run()["hsps"].tobytes() == want[name]; ps.close()
    st = api.record_cache_stats()
    assert st["sorts"] - st0["sorts"] == 1 and st["sorted_sets"] >= 1 and st["sorted_bytes"] > 0 and st["sorted_passes"] - st0["sorted_passes"] == 3
    assert st["bytes"] < 2 * st["sorted_bytes"]                         # the streams went back to the pool
    api.record_cache_invalidate()
    for mode in ("0", "1"):                                             # never; the default: only sets whose runs are long
        monkeypatch.setenv("GBN_REC_RUNS", mode)
        api.record_cache_invalidate()
        st0 = api.record_cache_stats()
        for name, q in (("a", qa), ("b", qb), ("a", qa)):
            ps = api.BlastPrelimSearch(q, opt, src); assert ps.run()["hsps"].tobytes() == want[name]; ps.close()
        st = api.record_cache_stats()
        assert st["sorts"] == st0["sorts"] and st["sorted_sets"] == 0 and st["hits"] - st0["hits"] == 2, (mode, st0, st)
    src.close()


def test_default_policy_sorts_at_the_second_hit(monkeypatch):
    """(of a set whose runs are long -- the size test is switched off here; test_full_size_* run the policy as it is)"""
    monkeypatch.delenv("GBN_RUNS_AFTER", raising=False)
    db, queries, plants, subjects, opt = util.small_case(8, 300_000, 40, task="megablast", seed=13)
    src = api.BlastSeqSrc.from_packed(subjects)
    ps = api.BlastPrelimSearch(queries, opt, src)
    want = None
    sorts = []
    for k in range(4):
        s0 = api.record_cache_stats()["sorts"]
        h = ps.run()["hsps"].tobytes()
        want = want or h
        assert h == want
        sorts.append(api.record_cache_stats()["sorts"] - s0)
    assert sorts == [0, 0, 1, 0], sorts                                 # miss, hit in stream form, hit that sorts, hit over runs
    ps.close(); src.close()


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_randomised_shapes(seed):
    rng = np.random.default_rng(seed)
    for _ in range(4):
        task = "megablast" if rng.random() < 0.7 else "blastn"
        nq = int(rng.choice([1, 7, 30, 120, 200]))
        nsub, slen = int(rng.integers(2, 9)), int(rng.choice([30_000, 120_000, 400_000]))
        kw = dict(word_size=int(rng.choice([28, 28, 20, 32, 16, 48]))) if task == "megablast" else dict(word_size=int(rng.choice([11, 13, 15])))
        db, queries, plants, subjects, opt = util.small_case(nsub, slen, nq, seed=int(rng.integers(1, 1 << 30)),
                                                             planted_fraction=0.6, task=task, **kw)
        three_passes(queries, opt, subjects, expect_sorted=False)
