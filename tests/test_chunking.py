"""Subjects longer than the engine's MAX_DBSEQ_LEN are searched in chunks (CORE/blast_engine.c:218-262, :455-540) and
the chunks' HSP lists are merged (Blast_HSPListsMerge, CORE/blast_hits.c:2545-2716).  G-BLASTN builds with
MAX_DBSEQ_LEN = 200,000,000 (COREI/blast_gapalign.h:54-55); the tests lower it to a few thousand bases.
CPU: the oracle's restatement against properties of the rule.  GPU: the product (chunk copies in the shard, lists
merged on the host, traceback over the re-assembled sequence) equals the oracle."""
import numpy as np
import pytest
from oracle import orc
from tests.test_traceback_oracle import mutate

MAXLEN = 4000


def case(seed=3):
    rng = np.random.default_rng(seed)
    queries = [rng.integers(0, 4, 900).astype(np.uint8) for _ in range(4)]
    long1 = rng.integers(0, 4, 12000).astype(np.uint8)
    long1[1000:1600] = mutate(rng, queries[0][100:700], subs=12, indels=0)              # inside chunk 0
    long1[3600:4400] = mutate(rng, queries[1][50:850], subs=20, indels=0)               # across the end of chunk 0 (4000) and the start of chunk 1 (3900)
    long1[7700:8300] = mutate(rng, queries[2][200:800], subs=10, indels=0)[:600]        # across 7800 (chunk 2 starts) -- chunk 1 ends at 7900
    long1[11650:11950] = queries[3][300:600]                                           # across 11700 (the short last chunk)
    short = rng.integers(0, 4, 2500).astype(np.uint8); short[500:1200] = queries[0][150:850]
    long2 = rng.integers(0, 4, 9000).astype(np.uint8)
    long2[3950:4350] = (3 - queries[1][200:600])[::-1]                                  # minus strand, starts inside the overlap strip
    return queries, [long1, short, long2]


def pad(s):
    return np.concatenate([orc.pack_ncbi2na(s), np.zeros(16, np.uint8)])


def test_oracle_chunks_follow_the_rule():
    queries, subjects = case()
    S = orc.Search(orc.default_options(True, db_length=10**7, db_num_seqs=10), queries)
    whole = S.subject(pad(subjects[0]), len(subjects[0]))["hsps"]
    parts = S.subject_chunked(pad(subjects[0]), len(subjects[0]), MAXLEN)
    assert len(whole) == 4 and len(parts) == 4
    key = lambda h: (int(h["context"]), int(h["q_offset"]))
    w = {int(h["context"]): h for h in whole}; p = {int(h["context"]): h for h in parts}
    assert set(w) == set(p)
    # an alignment inside one chunk, clear of the overlap strips, is what the whole subject gives
    inside = [c for c in w if w[c]["s_end"] < 3900]
    assert len(inside) == 1
    for f in ("q_offset", "q_end", "s_offset", "s_end", "score"):
        assert w[inside[0]][f] == p[inside[0]][f]
    # one that crosses a chunk boundary comes back as ONE HSP with the joint extent and the better of the two scores
    for c in w:
        if c in inside:
            continue
        assert p[c]["s_offset"] == w[c]["s_offset"] and p[c]["s_end"] == w[c]["s_end"]
        assert p[c]["q_offset"] == w[c]["q_offset"] and p[c]["q_end"] == w[c]["q_end"]
        assert 0 < p[c]["score"] < w[c]["score"]
    # a subject no longer than MAX_DBSEQ_LEN is one chunk
    a = S.subject(pad(subjects[1]), len(subjects[1]))["hsps"]
    b = S.subject_chunked(pad(subjects[1]), len(subjects[1]), MAXLEN)
    assert a.tobytes() == b.tobytes() and len(a) >= 1


@pytest.mark.gpu
@pytest.mark.parametrize("task", ["megablast", "blastn"])
def test_product_chunks_equal_the_oracle(task):
    from gblastn_amd import api
    from tests import util
    queries, subjects = case()
    api.set_max_dbseq_len(MAXLEN)
    try:
        src = api.BlastSeqSrc.from_packed([(orc.pack_ncbi2na(s), len(s)) for s in subjects])
        assert src.num_seqs == 3                                    # sequences, not chunks
        opt = api.default_options(task, db_length=10**7, db_num_seqs=10)
        ps = api.BlastPrelimSearch(queries, opt, src)
        got = ps.run()["hsps"]
        S = orc.Search(util.oracle_options(opt), queries)
        want, want_oid = [], []
        for oid, s in enumerate(subjects):
            h = S.subject_chunked(pad(s), len(s), MAXLEN)
            want.append(h); want_oid += [oid] * len(h)
        want = np.concatenate(want); want_oid = np.array(want_oid)
        assert len(want) >= 6
        assert np.array_equal(got["oid"], want_oid)
        for f in ("context", "q_offset", "q_end", "q_gapped_start", "s_offset", "s_end", "s_gapped_start", "score"):
            assert np.array_equal(got[f], want[f]), f
        assert np.array_equal(got["evalue"].view(np.uint64), want["evalue"].view(np.uint64))
        # the pipelined entry points merge as well
        ps.begin(); got2 = ps.end()["hsps"]
        assert got2.tobytes() == got.tobytes()
        # traceback over the merged lists reads the sequence back from its chunk copies
        col = api.BlastHSPCollector(len(queries), 10); col.write(got); hs, st, lq = col.close()
        tb = api.BlastTracebackSearch(ps, src)
        rec, ops, qs = tb.run(hs, st)
        ocol = orc.Collector(len(queries), 10)
        for oid in range(len(subjects)):
            m = want[want_oid == oid]
            if len(m):
                ocol.write(oid, [dict(zip(m.dtype.names, x)) for x in m])
        n = 0
        fin_by = {}
        for oid, q, lst in ocol.close():
            fin = S.traceback(subjects[oid], [dict(zip(orc.Collector.FIELDS, h)) for h in lst])
            for f in fin:
                fin_by[(oid, f["context"], f["q_offset"])] = f
        assert len(rec) == len(fin_by) and len(rec) >= 6
        for r in rec:
            f = fin_by[(int(r["oid"]), int(r["context"]), int(r["q_offset"]))]
            for k in ("q_end", "s_offset", "s_end", "score", "num_ident", "align_length", "gaps"):
                assert int(r[k]) == int(f[k]), k
    finally:
        api.set_max_dbseq_len()


@pytest.mark.gpu
def test_a_subject_longer_than_the_reference_limit_at_full_size():
    """MAX_DBSEQ_LEN as G-BLASTN builds it (200,000,000): one subject of 230 Mbp is two chunks, the second one
    starting DBSEQ_CHUNK_OVERLAP bases before the limit.  Alignments inside either chunk, across the boundary and
    at the very end of the subject equal the oracle's chunked search (the largest single subject the tests hold)."""
    from gblastn_amd import api
    from tests import util
    rng = np.random.default_rng(77)
    L, n = 200_000_000, 230_000_000
    queries = [rng.integers(0, 4, 1000).astype(np.uint8) for _ in range(4)]
    subj = rng.integers(0, 4, n, dtype=np.uint8)
    subj[5_000_000:5_000_700] = mutate(rng, queries[0][100:800], subs=10, indels=0)
    subj[L - 300:L + 400] = mutate(rng, queries[1][150:850], subs=12, indels=0)       # across the end of chunk 0
    subj[L + 20_000_000:L + 20_000_600] = (3 - queries[2][200:800])[::-1]             # minus strand, chunk 1
    subj[n - 500:n] = queries[3][400:900]                                             # runs to the last base
    small = rng.integers(0, 4, 50_000, dtype=np.uint8); small[1000:1600] = queries[0][300:900]
    subjects = [small, subj]
    api.set_max_dbseq_len()
    src = api.BlastSeqSrc.from_packed([(orc.pack_ncbi2na(s), len(s)) for s in subjects])
    opt = api.default_options("megablast", db_length=n + 50_000, db_num_seqs=2)
    ps = api.BlastPrelimSearch(queries, opt, src)
    got = ps.run()["hsps"]
    S = orc.Search(util.oracle_options(opt), queries)
    want = [S.subject_chunked(pad(s), len(s), L) for s in subjects]
    want_oid = np.concatenate([[i] * len(h) for i, h in enumerate(want)]); want = np.concatenate(want)
    assert len(want) >= 5 and (want["s_end"][want_oid == 1] > L).any()
    assert np.array_equal(got["oid"], want_oid)
    for f in ("context", "q_offset", "q_end", "q_gapped_start", "s_offset", "s_end", "s_gapped_start", "score"):
        assert np.array_equal(got[f], want[f]), f
    assert np.array_equal(got["evalue"].view(np.uint64), want["evalue"].view(np.uint64))
    ps.begin(); assert ps.end()["hsps"].tobytes() == got.tobytes()
