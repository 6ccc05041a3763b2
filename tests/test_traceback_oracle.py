"""CPU: the oracle's traceback stage (oracle/orc_traceback.c) pinned

* on the reference's known answers that are reproducible offline (greedy1a/b -> 619 / 6034,
  UT/bl2seq_unit_test.cpp:1620-1691 are final, post-traceback scores; nt.41646578 self hit,
  UT/prelimsearch_unit_test.cpp:169-203 / UT/traceback_unit_test.cpp:742),
* at definition level: an edit script must re-score to the reported score over the reported span,
  ALIGN_EX must equal its score-only twin Blast_SemiGappedAlign (same cells, same pruning), and with an
  X-drop larger than any score both must equal the full O(NM) affine-gap matrix evaluated cell by cell.
"""
import os
import numpy as np
import pytest
from oracle import orc

G = os.path.join(os.path.dirname(__file__), "golden")
DEL, SUB, INS = 0, 3, 6


def rescore(q, s, r, reward, penalty, gap_open, gap_extend):
    """walk an edit script: (score, query bases used, subject bases used, identities)"""
    qi, si, score, ident = r["q_start"], r["s_start"], 0, 0
    for op, n in r["ops"]:
        if op == SUB:
            for _ in range(n):
                if q[qi] == s[si] and q[qi] < 4:
                    score += reward; ident += 1
                else:
                    score += penalty
                qi += 1; si += 1
        elif op == DEL:
            score -= gap_open + gap_extend * n; si += n
        elif op == INS:
            score -= gap_open + gap_extend * n; qi += n
        else:
            raise AssertionError("op %d" % op)
    return score, qi, si, ident


def mutate(rng, seq, subs, indels):
    out = list(seq)
    for _ in range(indels):
        p = int(rng.integers(5, len(out) - 5))
        if rng.random() < 0.5:
            del out[p:p + int(rng.integers(1, 4))]
        else:
            out[p:p] = list(rng.integers(0, 4, int(rng.integers(1, 4))))
    out = np.array(out, dtype=np.uint8)
    for p in rng.integers(0, len(out), subs):
        out[p] = (out[p] + 1 + rng.integers(0, 3)) & 3
    return out


def find_anchor(q, s, around, k=14):
    """a (query, subject) position inside a common exact k-mer, query position near `around`"""
    index = {}
    for j in range(len(s) - k + 1):
        index.setdefault(bytes(s[j:j + k]), j)
    for d in range(0, 200):
        for qs in (around + d, around - d):
            if 0 <= qs <= len(q) - k and bytes(q[qs:qs + k]) in index:
                return qs + k // 2, index[bytes(q[qs:qs + k])] + k // 2
    raise AssertionError("no anchor")


def affine_full_dp(q, s, reward, penalty, go, ge):
    """best score of an alignment that starts at (0, 0) and ends anywhere: the whole matrix, three states"""
    n, m = len(q), len(s)
    NEG = -10**9
    H = np.full((n + 1, m + 1), NEG, dtype=np.int64); E = H.copy(); F = H.copy()
    H[0, 0] = 0
    for j in range(1, m + 1):
        E[0, j] = -(go + ge * j); H[0, j] = E[0, j]
    for i in range(1, n + 1):
        F[i, 0] = -(go + ge * i); H[i, 0] = F[i, 0]
        for j in range(1, m + 1):
            E[i, j] = max(E[i, j - 1] - ge, H[i, j - 1] - go - ge)
            F[i, j] = max(F[i - 1, j] - ge, H[i - 1, j] - go - ge)
            d = H[i - 1, j - 1] + (reward if q[i - 1] == s[j - 1] else penalty)
            H[i, j] = max(d, E[i, j], F[i, j])
    return int(H.max())


@pytest.mark.parametrize("seed", range(6))
def test_align_ex_equals_score_only_twin_and_rescoring(seed):
    rng = np.random.default_rng(100 + seed)
    q = rng.integers(0, 4, 600).astype(np.uint8)
    s_core = mutate(rng, q[100:500], subs=int(rng.integers(0, 30)), indels=int(rng.integers(0, 6)))
    s = np.concatenate([rng.integers(0, 4, 150).astype(np.uint8), s_core, rng.integers(0, 4, 150).astype(np.uint8)])
    opt = orc.default_options(False, db_length=10**6, db_num_seqs=10)         # blastn: 2 / -3, gaps 5 / 2, DP traceback
    S = orc.Search(opt, [q])
    M = S.matrix()
    for x in (16, 33, 60, 110):
        qs, ss = find_anchor(q, s, 300)                                       # a start point inside the planted region
        r = S.align_traceback(0, s, qs, ss, x)
        sc, qe, se, _ = rescore(q, s, r, opt.reward, opt.penalty, opt.gap_open, opt.gap_extend)
        assert sc == r["score"] and (qe, se) == (r["q_stop"], r["s_stop"])
        assert all(n > 0 for _, n in r["ops"]) and r["ops"][0][0] == SUB and r["ops"][-1][0] == SUB
        # the score-only twin over the same two quadrants (CORE/blast_gapalign.c:4040-4093)
        left, la, lb = orc.semi_gapped_score(M, q, s, qs + 1, ss + 1, x, opt.gap_open, opt.gap_extend, True)
        right, ra, rb = orc.semi_gapped_score(M, q[qs:], s[ss:], len(q) - qs - 1, len(s) - ss - 1, x, opt.gap_open, opt.gap_extend, False)
        assert left + right == r["score"]
        assert (qs - la + 1, ss - lb + 1) == (r["q_start"], r["s_start"])
        assert (qs + ra + 1, ss + rb + 1) == (r["q_stop"], r["s_stop"])


@pytest.mark.parametrize("seed", range(4))
def test_align_ex_with_unbounded_xdrop_is_the_full_affine_matrix(seed):
    rng = np.random.default_rng(200 + seed)
    q = rng.integers(0, 4, 70).astype(np.uint8)
    s = mutate(rng, q, subs=int(rng.integers(0, 12)), indels=int(rng.integers(0, 4)))
    opt = orc.default_options(False, db_length=10**6, db_num_seqs=10)
    S = orc.Search(opt, [q])
    M = S.matrix()
    X = 10**6
    # one quadrant, forward: rows q[1..], columns s[1..] (the start point itself is the left extension's)
    sc, a, b = orc.semi_gapped_score(M, q, s, len(q) - 1, len(s) - 1, X, opt.gap_open, opt.gap_extend, False)
    assert sc == affine_full_dp(q[1:], s[1:], opt.reward, opt.penalty, opt.gap_open, opt.gap_extend)
    r = S.align_traceback(0, s, 0, 0, X)
    full = sc + int(M[q[0]][s[0]]) if M[q[0]][s[0]] > 0 else None
    if full is not None:
        assert r["score"] == full


@pytest.mark.parametrize("seed", range(5))
def test_greedy_traceback_rescoring_and_agreement_with_dp(seed):
    rng = np.random.default_rng(300 + seed)
    q = rng.integers(0, 4, 800).astype(np.uint8)
    s_core = mutate(rng, q[100:700], subs=int(rng.integers(0, 25)), indels=int(rng.integers(0, 5)))
    s = np.concatenate([rng.integers(0, 4, 120).astype(np.uint8), s_core, rng.integers(0, 4, 120).astype(np.uint8)])
    opt = orc.default_options(True, db_length=10**6, db_num_seqs=10)           # megablast 1 / -2, gaps 0 / 0
    S = orc.Search(opt, [q])
    qs, ss = find_anchor(q, s, 400)
    r = S.align_traceback(0, s, qs, ss, 80, greedy=True)
    # linear greedy costs: a gap of length n costs n * (reward / 2 - penalty) (CORE/blast_hits.c:380-387), in half points
    sc2, qe, se, _ = rescore(q, s, r, 2 * opt.reward, 2 * opt.penalty, 0, opt.reward - 2 * opt.penalty)
    assert (qe, se) == (r["q_stop"], r["s_stop"])
    assert sc2 // 2 == r["score"]


def _greedy_pair():
    a = orc.encode_blastna(orc.read_fasta(os.path.join(G, "greedy1a.fsa"))[0])
    b = orc.encode_blastna(orc.read_fasta(os.path.join(G, "greedy1b.fsa"))[0])
    return a, b


def test_MegablastGreedyTraceback2_final_alignment():
    # UT/bl2seq_unit_test.cpp:1620-1672: the test reads score 619 from the FINAL alignment (after traceback)
    a, b = _greedy_pair()
    S = orc.Search(orc.default_options(True), [a])
    pre = S.subject(orc.pack_ncbi2na(b), len(b))["hsps"]
    fin = S.traceback(b, pre)
    assert len(fin) == 1 and fin[0]["score"] == 619
    sc2, qe, se, ident = rescore(a, b, dict(q_start=fin[0]["q_offset"], s_start=fin[0]["s_offset"], ops=fin[0]["ops"]), 2, -4, 0, 5)
    assert sc2 // 2 == 619 and (qe, se) == (fin[0]["q_end"], fin[0]["s_end"]) and ident == fin[0]["num_ident"]
    # 10 / -25 scoring, X-drop 100 / 100 (:1674-1690)
    opt = orc.default_options(True, reward=10, penalty=-25, xdrop_gap_bits=100.0, xdrop_gap_final_bits=100.0)
    S = orc.Search(opt, [a])
    fin = S.traceback(b, S.subject(orc.pack_ncbi2na(b), len(b))["hsps"])
    assert len(fin) == 1 and fin[0]["score"] == 6034


def test_nt41646578_slice_traceback():
    # the 507-base slice of UT/prelimsearch_unit_test.cpp:169-203 is an exact copy: one ungapped final HSP
    db = orc.read_blastdb_v4_nucl(os.path.join(G, "nt.41646578"))
    packed, n = db[0]
    sub = orc.unpack_ncbi2na(packed, n)
    q = sub[54:561].copy()
    S = orc.Search(orc.default_options(True, db_length=n, db_num_seqs=1), [q])
    pre = S.subject(packed, n)["hsps"]
    plus = [h for h in pre if h["context"] == 0]
    fin = S.traceback(sub, plus)
    assert fin[0]["score"] == 507 and fin[0]["ops"] == [(SUB, 507)] and fin[0]["num_ident"] == 507
    assert (fin[0]["q_offset"], fin[0]["q_end"], fin[0]["s_offset"], fin[0]["s_end"]) == (0, 507, 54, 561)
    assert fin[0]["gaps"] == 0 and fin[0]["align_length"] == 507


@pytest.mark.parametrize("megablast", [True, False])
def test_traceback_lists_are_consistent(megablast):
    """whole stage on planted homologs: every final HSP re-scores to its score, lists are sorted, no HSP is
    enveloped by a better one, e-values follow the scores"""
    rng = np.random.default_rng(7 if megablast else 8)
    sub = rng.integers(0, 4, 30000).astype(np.uint8)
    queries = []
    for k in range(6):
        core = mutate(rng, sub[2000 + 4000 * k: 2000 + 4000 * k + 700], subs=int(rng.integers(0, 25)), indels=int(rng.integers(0, 4)))
        queries.append(np.concatenate([rng.integers(0, 4, 100).astype(np.uint8), core, rng.integers(0, 4, 100).astype(np.uint8)]))
    opt = orc.default_options(megablast, db_length=10**7, db_num_seqs=50)
    S = orc.Search(opt, queries)
    pre = S.subject(orc.pack_ncbi2na(sub), len(sub))["hsps"]
    assert len(pre) >= 6
    ctxs = S.contexts
    total = 0
    for qi in range(len(queries)):
        mine = [h for h in pre if h["context"] // 2 == qi]
        fin = S.traceback(sub, mine)
        total += len(fin)
        qcat = S.query_concat()
        for f in fin:
            c = ctxs[f["context"]]
            qs = qcat[c.query_offset:c.query_offset + c.query_length]
            if megablast:
                sc2, qe, se, ident = rescore(qs, sub, dict(q_start=f["q_offset"], s_start=f["s_offset"], ops=f["ops"]), 2, -4, 0, 5)
                assert sc2 // 2 == f["score"]
            else:
                sc, qe, se, ident = rescore(qs, sub, dict(q_start=f["q_offset"], s_start=f["s_offset"], ops=f["ops"]), opt.reward, opt.penalty, opt.gap_open, opt.gap_extend)
                assert sc & ~1 == f["score"]                                    # blastn rounds odd scores down
            assert (qe, se) == (f["q_end"], f["s_end"]) and ident == f["num_ident"]
        assert [f["score"] for f in fin] == sorted([f["score"] for f in fin], reverse=True)
    assert total >= 6


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("costs", [(1, -2, 2, 2), (1, -3, 5, 2), (2, -3, 5, 2), (1, -2, 0, 2)])
def test_affine_greedy_traceback_twin_rescoring_and_optimum(seed, costs):
    """BLAST_AffineGreedyAlign with an edit block (CORE/greedy_align.c:755-1236): the same score and end points as
    its score-only twin on the packed subject (the preliminary stage's aligner, pinned in test_oracle_definitions),
    the score of its own edit script, and -- under an X-drop nothing is pruned by -- the optimum of the whole affine
    matrix on either side of the start point"""
    reward, penalty, go, ge = costs
    rng = np.random.default_rng(900 + seed)
    q = rng.integers(0, 4, 700).astype(np.uint8)
    s_core = mutate(rng, q[80:620], subs=int(rng.integers(0, 20)), indels=int(rng.integers(0, 5)))
    s = np.concatenate([rng.integers(0, 4, 90).astype(np.uint8), s_core, rng.integers(0, 4, 90).astype(np.uint8)])
    opt = orc.default_options(True, db_length=10**6, db_num_seqs=10)
    opt.reward, opt.penalty, opt.gap_open, opt.gap_extend = reward, penalty, go, ge
    S = orc.Search(opt, [q])
    qs, ss = find_anchor(q, s, 350)
    for X in (30, 10000):
        r = S.align_traceback(0, s, qs, ss, X, greedy=True)
        twin = orc.gapped_extend(q, s, qs, ss, X, reward, penalty, go, ge, greedy=True)
        assert (r["q_start"], r["q_stop"], r["s_start"], r["s_stop"], r["score"]) == \
               (twin["q_offset"], twin["q_end"], twin["s_offset"], twin["s_end"], twin["score"])
        sc2, qe, se, _ = rescore(q, s, r, reward, penalty, go, ge)
        assert (qe, se) == (r["q_stop"], r["s_stop"])
        # (s_ReduceGaps may trade a pair of opposite gaps for mismatches: the script never scores below the aligner's)
        assert sc2 >= r["score"]
        assert any(op != SUB for op, _ in r["ops"]) or r["score"] == sc2
    full = affine_full_dp(q[:qs][::-1], s[:ss][::-1], reward, penalty, go, ge) + affine_full_dp(q[qs:], s[ss:], reward, penalty, go, ge)
    assert r["score"] == full
