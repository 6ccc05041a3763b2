"""BLAST database v4 reader (C++, through the C ABI) against the values the reference's own
tests pin for its unit-test databases (UT/seqsrc_unit_test.cpp:240-311) and against the
test-only Python reader of the oracle."""
import os
import numpy as np
import pytest
from gblastn_amd import api
from oracle import orc

G = os.path.join(os.path.dirname(__file__), "golden")


def test_seqn_known_values_of_the_reference_tests():
    db = api.BlastDb(os.path.join(G, "seqn"))
    assert db.num_volumes == 1
    assert db.num_seqs == 2004                      # kNumSeqs
    assert db.total_length == 943942                # kTotLen
    assert db.max_length == 875                     # kMaxLen
    assert db.total_length // db.num_seqs == 471    # kAvgLen
    lens = np.array([db.seq_length(i) for i in range(db.num_seqs)])
    assert lens.sum() == 943942 and lens.max() == 875
    assert lens[1000:2000].sum() == 478404          # kTotLenRange, OIDs [1000, 2000)
    assert lens[1000:2000].max() == 839             # kMaxLenRange
    assert db.seq_length(1500) == 715               # kSeqLength
    packed, n = db.ncbi2na(1500)
    assert n == 715 and packed[:4].tolist() == [159, 145, 213, 43]         # kNcbi2naSeqBytes
    assert db.blastna(1500, sentinels=True)[:5].tolist() == [15, 2, 1, 3, 3]   # kBlastnaSeqBytes
    assert db.seq_length(2004) < 0 and db.seq_length(-1) < 0


def test_every_sequence_equals_the_python_reader():
    for name in ["seqn", "nt.41646578"]:
        ref = orc.read_blastdb_v4_nucl(os.path.join(G, name))
        db = api.BlastDb(os.path.join(G, name))
        assert db.num_seqs == len(ref)
        for oid, (packed, n) in enumerate(ref):
            got, gn = db.ncbi2na(oid)
            assert gn == n
            assert np.array_equal(got, packed[:(n + 3) // 4]), oid


def test_ambiguity_runs_are_applied():
    db = api.BlastDb(os.path.join(G, "seqn"))
    seen = 0
    for oid in range(db.num_seqs):
        st, ln, val = db.ambiguities(oid)
        if len(st) == 0:
            continue
        seen += 1
        n = db.seq_length(oid)
        seq = db.blastna(oid)
        plain = orc.unpack_ncbi2na(db.ncbi2na(oid)[0], n)
        amb = np.zeros(n, dtype=bool)
        for a, l, v in zip(st, ln, val):
            assert 0 <= a and a + l <= n and 1 <= l <= 4096 and 1 <= v <= 15
            amb[a:a + l] = True
        assert np.array_equal(seq[~amb], plain[~amb])       # untouched outside the runs
        assert np.all(seq[amb] > 3)                          # ambiguity codes inside them
    assert seen > 0                                          # the fixture does contain ambiguous sequences


def test_alias_file_with_two_volumes():
    db = api.BlastDb(os.path.join(G, "two_vols"))
    assert db.num_volumes == 2 and db.num_seqs == 2005
    assert db.total_length == 943942 + 3300
    assert (db.stat_num_seqs, db.stat_length) == (2005, 947242)
    assert db.title == "two volumes of the unit-test data"
    assert db.volume_range(0) == (0, 2004) and db.volume_range(1) == (2004, 1)
    assert db.seq_length(2004) == 3300
    one = api.BlastDb(os.path.join(G, "nt.41646578"))
    assert np.array_equal(db.ncbi2na(2004)[0], one.ncbi2na(0)[0])
    assert db.max_length == 3300


def test_errors_are_loud():
    with pytest.raises(api.BlastError):
        api.BlastDb(os.path.join(G, "no_such_db"))
    with pytest.raises(api.BlastError):
        api.BlastDb(os.path.join(G, "filtered"))            # GILIST filtering is not supported
    with pytest.raises(api.BlastError):
        api.BlastDb(os.path.join(G, "greedy1a"))            # not a database at all


@pytest.mark.gpu
def test_search_a_database_loaded_from_files():
    # the reference's prelim-search known answer (UT/prelimsearch_unit_test.cpp:169-203) through the
    # file reader, and a sharded seqn search against the oracle
    db = api.BlastDb(os.path.join(G, "two_vols"))
    q = orc.unpack_ncbi2na(*db.ncbi2na(2004))[54:561].copy()
    src = db.load_shard(2004, 1)
    ps = api.BlastPrelimSearch([q], api.default_options("megablast", db_length=3300, db_num_seqs=1), src)
    h = ps.run()["hsps"]
    assert len(h) == 1 and h[0]["oid"] == 2004
    assert (h[0]["q_offset"], h[0]["q_end"] - 1, h[0]["s_offset"], h[0]["s_end"] - 1) == (0, 506, 54, 560)
    # whole database in two shards, queries cut from database sequences (with ambiguity codes kept)
    from tests import util
    queries = [db.blastna(i) for i in (5, 700, 1500, 1999)]
    opt = api.default_options("megablast", db_length=db.stat_length, db_num_seqs=db.stat_num_seqs)
    a = api.BlastPrelimSearch(queries, opt, db.load_shard(0, 1000)).run()["hsps"]
    b = api.BlastPrelimSearch(queries, opt, db.load_shard(1000, 1005)).run(keep_stages=False)["hsps"]
    got = np.concatenate([a, b])
    subjects = [db.ncbi2na(i) for i in range(db.num_seqs)]
    subjects = [(np.concatenate([p, np.zeros(16, dtype=np.uint8)]), n) for p, n in subjects]
    ora, s = util.oracle_run(opt, queries, subjects)
    assert {5, 700, 1500, 1999} <= set(got["oid"].tolist())
    util.compare_stages(dict(hsps=got), ora)


@pytest.mark.gpu
def test_shard_uploaded_in_pieces_equals_the_one_slab_form(monkeypatch):
    """gbn_blastdb_load_shard fills pinned pieces from the volumes' files (pread) on worker threads and uploads them while
    others are being filled (gbn_db_new_streamed); GBN_LOAD_ONE_SLAB=1 is the loader of rounds 1-5 -- one host slab, one copy.
    Same search results either way, over both volumes of the alias, ambiguity runs attached."""
    db = api.BlastDb(os.path.join(G, "two_vols"))
    queries = [db.blastna(i) for i in (5, 700, 1500, 1999, 2004)]
    opt = api.default_options("megablast", db_length=db.stat_length, db_num_seqs=db.stat_num_seqs)
    got = {}
    for mode in ("pieces", "slab"):
        if mode == "slab":
            monkeypatch.setenv("GBN_LOAD_ONE_SLAB", "1")
        src = db.load_shard()
        got[mode] = api.BlastPrelimSearch(queries, opt, src).run()["hsps"].tobytes()
        src.close()
    assert got["pieces"] == got["slab"] and len(got["pieces"]) > 0
