"""CPU: the oracle's restatements of the paths C2 / C3 actually run, checked against DEFINITIONS that share
no structure with them ("widen the oracle pin"):

* seeds of the megablast table (lut 11 and 12, hash container sized batches) = brute-force enumeration of
  (scan position, query position) pairs by the rule of CORE/na_ungapped.c:1025-1144 -- this restates the
  four fixtures of UT/ntscan_unit_test.cpp:560-790 (every hit's word equals the subject's, none is missed,
  order = scan order, chain order inside a position), not only ScanCheckHits;
* the packed-subject X-drop DP (s_BlastAlignPackedNucl) = the whole affine matrix evaluated cell by cell with the
  reference's pruning rule, on pairs whose band shrinks and regrows (CORE/blast_gapalign.c:2957-3052);
* affine greedy = that matrix's optimum when the X-drop is large enough for both to be optimal.
The same checkers run against the HIP kernels in tests/test_gpu_definitions.py."""
import numpy as np
import pytest
from oracle import orc
from tests.test_traceback_oracle import mutate, find_anchor


# ---------------------------------------------------------------- seeds by definition
def brute_force_seeds(qcat, ctxs, subj, word, lut, step, descending):
    """(q, s) of every seed in the order the reference reports them: scan positions s = 0, step, ... ascending;
    at one position the query offsets of the cell in table order (megablast chains: descending); a lookup hit
    becomes a seed when the exact match extends to `word` bases with at most word - lut of them on the left."""
    index = {}
    for c in ctxs:
        if c.query_length < word:
            continue
        for q in range(c.query_offset, c.query_offset + c.query_length - lut + 1):
            w = qcat[q:q + lut]
            if (w < 4).all():
                index.setdefault(bytes(w), []).append(q)
    ext_to = word - lut
    out = []
    qn, sn = len(qcat), len(subj)
    for s in range(0, sn - lut + 1, step):
        qs = index.get(bytes(subj[s:s + lut]))
        if not qs:
            continue
        for q in (reversed(qs) if descending else qs):
            left = 0
            while left < min(ext_to, s) and q - left - 1 >= 0 and qcat[q - left - 1] == subj[s - left - 1]:
                left += 1
            if left < ext_to:
                need = ext_to - left
                if s + lut + need > sn:
                    continue
                r = 0
                while r < need and q + lut + r < qn and qcat[q + lut + r] == subj[s + lut + r]:
                    r += 1
                if r < need:
                    continue
            out.append((q - left, s - left))
    return out


def seed_case(nq, seed):
    rng = np.random.default_rng(seed)
    sub = rng.integers(0, 4, 60000).astype(np.uint8)
    queries = []
    for k in range(nq):
        if k % 3 == 0:      # a homolog: seeds that verify
            a = int(rng.integers(0, len(sub) - 1000))
            core = mutate(rng, sub[a:a + 700], subs=int(rng.integers(0, 12)), indels=int(rng.integers(0, 3)))
            q = np.concatenate([rng.integers(0, 4, 150).astype(np.uint8), core, rng.integers(0, 4, 150).astype(np.uint8)])
        else:
            q = rng.integers(0, 4, 1000).astype(np.uint8)
        if k % 5 == 0:      # an ambiguity code: words over it are not indexed
            q[int(rng.integers(0, len(q)))] = 14
        queries.append(q)
    return sub, queries


@pytest.mark.parametrize("nq,lut,step", [(16, 11, 18), (160, 12, 17)])
def test_megablast_seeds_equal_the_definition(nq, lut, step):
    sub, queries = seed_case(nq, 40 + nq)
    S = orc.Search(orc.default_options(True, db_length=10**7, db_num_seqs=10), queries)
    info = S.info()
    assert (info["lut_type"], info["lut_width"], info["scan_step"], info["container"]) == (3, lut, step, 1)
    r = S.subject(orc.pack_ncbi2na(sub), len(sub))
    qcat = S.query_concat()
    want = brute_force_seeds(qcat, S.contexts, sub, 28, lut, step, descending=True)
    got = list(zip(r["seeds"]["q_off"].tolist(), r["seeds"]["s_off"].tolist()))
    assert len(want) >= nq // 4
    assert got == want


# ---------------------------------------------------------------- X-drop DP by definition
def xdrop_matrix(q, s, reward, penalty, go, ge, X):
    """Rows = subject letters (a), columns = query letters (b), as s_BlastAlignPackedNucl walks them.  The whole
    (M+1) x (N+1) matrix exists; a cell is LIVE when the rule keeps it: visited left to right inside the row's
    window, dropped when best-so-far minus its score exceeds X (the window's left edge moves past leading dropped
    cells, the right edge to the last kept cell, then out again while a horizontal gap stays within X).
    Returns (best score, a, b) with a / b the numbers of subject / query letters consumed."""
    N, M = len(q), len(s)
    NEG = -(1 << 30)
    goe = go + ge
    X = max(X, goe)
    best = np.full(N + 2, NEG, dtype=np.int64); gap = np.full(N + 2, NEG, dtype=np.int64)
    best[0] = 0; gap[0] = -goe
    b_size = 1; sc = -goe
    while b_size <= N and sc >= -X:
        best[b_size] = sc; gap[b_size] = sc - goe; sc -= ge; b_size += 1
    first = 0; top = 0; at = (0, 0)
    for a in range(1, M + 1):
        score = NEG; gap_row = NEG; last = first
        row_first = first
        for b in range(row_first, b_size):
            gap_col = int(gap[b])
            nxt = int(best[b]) + (reward if (b < N and q[b] == s[a - 1] and q[b] < 4) else penalty) if b < N else NEG * 2
            score = max(score, gap_col, gap_row)
            if top - score > X:
                if b == first:
                    first += 1
                else:
                    best[b] = NEG
            else:
                last = b
                if score > top:
                    top = score; at = (a, b)
                gap_row -= ge; gap_col -= ge
                gap[b] = max(score - goe, gap_col)
                gap_row = max(score - goe, gap_row)
                best[b] = score
            score = nxt
        if first == b_size:
            break
        if last < b_size - 1:
            b_size = last + 1
        else:
            while gap_row >= top - X and b_size <= N:
                best[b_size] = gap_row; gap[b_size] = gap_row - goe; gap_row -= ge; b_size += 1
        if b_size <= N:
            best[b_size] = NEG; gap[b_size] = NEG; b_size += 1
    return top, at[0], at[1]


def dp_by_definition(q, s, q_off, s_off, reward, penalty, go, ge, X):
    """s_BlastDynProgNtGappedAlignment: the left part over the reversed prefixes up to the next 4-aligned subject
    base, the right part from there"""
    adj = 4 - (s_off % 4)
    ql, sl = q_off + adj, s_off + adj
    if ql > len(q) or sl > len(s):
        ql -= 4; sl -= 4
    left, la, lb = xdrop_matrix(q[:ql][::-1], s[:sl][::-1], reward, penalty, go, ge, X)
    out = dict(q_offset=ql - lb, s_offset=sl - la)
    right = 0
    if ql < len(q) and sl < len(s):
        right, ra, rb = xdrop_matrix(q[ql:], s[sl:], reward, penalty, go, ge, X)
        out.update(q_end=ql + rb, s_end=sl + ra)
    else:
        out.update(q_end=ql, s_end=sl)
    out["score"] = left + right
    return out


def dp_pairs(n, seed):
    """pairs with planted homology of varying quality: clean stretches (the band shrinks to the diagonal), indel
    clusters and noisy stretches (it regrows), and pure chance hits"""
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        q = rng.integers(0, 4, int(rng.integers(120, 420))).astype(np.uint8)
        if k % 4 == 3:
            s = rng.integers(0, 4, len(q) + 40).astype(np.uint8)
            s[50:61] = q[40:51]
            out.append((q, s, 45, 55)); continue
        core = mutate(rng, q[20:len(q) - 20], subs=int(rng.integers(0, len(q) // 12)), indels=int(rng.integers(0, 5)))
        s = np.concatenate([rng.integers(0, 4, 30).astype(np.uint8), core, rng.integers(0, 4, 30).astype(np.uint8)])
        try:
            qo, so = find_anchor(q, s, len(q) // 2, k=11)
        except AssertionError:
            continue
        out.append((q, s, qo, so))
    return out


def test_packed_xdrop_dp_equals_the_matrix_definition():
    pairs = dp_pairs(240, 5)
    assert len(pairs) >= 200
    shrunk = regrown = 0
    for q, s, qo, so in pairs:
        for X, go, ge in ((33, 5, 2), (16, 5, 2), (40, 2, 1)):
            got = orc.gapped_extend(q, s, qo, so, X, 2, -3, go, ge)
            want = dp_by_definition(q, s, qo, so, 2, -3, go, ge, X)
            assert got == want, (len(q), qo, so, X, go, ge)
        span = got["q_end"] - got["q_offset"]
        shrunk += span > 60; regrown += abs((got["q_end"] - got["q_offset"]) - (got["s_end"] - got["s_offset"])) > 0
    assert shrunk >= 50 and regrown >= 20          # long extensions and ones that kept an indel


def gotoh_optimum(q, s, reward, penalty, go, ge):
    """no pruning, no window: max over all (a, b) of the best affine-gap alignment of s[:a] with q[:b]
    (textbook three-matrix recurrence, a row at a time; the horizontal gap is a prefix maximum)"""
    N = len(q)
    NEG = -(1 << 40)
    k = np.arange(N + 1, dtype=np.int64)
    H = np.where(k == 0, 0, -go - ge * k); E = np.full(N + 1, NEG, dtype=np.int64)
    top = 0
    for a in range(1, len(s) + 1):
        sub = np.where((q == s[a - 1]) & (q < 4), reward, penalty).astype(np.int64)
        E = np.maximum(H - go - ge, E - ge)                      # vertical gap: from the row above
        D = np.full(N + 1, NEG, dtype=np.int64); D[1:] = H[:-1] + sub
        D[0] = -go - ge * a
        G = np.maximum(D, E)                                     # everything but the horizontal gap
        # F[b] = max_{j<b} (G'[j] - go - ge (b - j)) where G' includes F itself; opening from F is never better
        pm = np.maximum.accumulate(G + ge * k)
        F = np.full(N + 1, NEG, dtype=np.int64); F[1:] = pm[:-1] - go - ge * k[1:]
        H = np.maximum(G, F)
        top = max(top, int(H.max()))
    return top


def test_packed_xdrop_dp_without_pruning_is_the_affine_optimum():
    for q, s, qo, so in dp_pairs(80, 9):
        adj = 4 - (so % 4); ql, sl = qo + adj, so + adj
        if ql >= len(q) or sl >= len(s):
            continue
        for go, ge in ((5, 2), (2, 1), (0, 3)):
            got = orc.gapped_extend(q, s, qo, so, 10**6, 2, -3, go, ge)
            want = gotoh_optimum(q[:ql][::-1], s[:sl][::-1], 2, -3, go, ge) + gotoh_optimum(q[ql:], s[sl:], 2, -3, go, ge)
            assert got["score"] == want


# ---------------------------------------------------------------- affine greedy against the matrix optimum
def test_affine_greedy_equals_the_dp_optimum_under_a_large_xdrop():
    rng = np.random.default_rng(17)
    checked = 0
    for k in range(60):
        q = rng.integers(0, 4, int(rng.integers(150, 300))).astype(np.uint8)
        core = mutate(rng, q[10:len(q) - 10], subs=int(rng.integers(0, 8)), indels=int(rng.integers(0, 3)))
        s = np.concatenate([rng.integers(0, 4, 15).astype(np.uint8), core, rng.integers(0, 4, 15).astype(np.uint8)])
        try:
            qo, so = find_anchor(q, s, len(q) // 2, k=12)
        except AssertionError:
            continue
        for reward, penalty, go, ge in ((1, -2, 2, 2), (1, -3, 5, 2), (2, -3, 5, 2)):
            X = 10000
            g = orc.gapped_extend(q, s, qo, so, X, reward, penalty, go, ge, greedy=True)
            # the optimum of an extension anchored at (qo, so) in both directions, whole matrices
            left = gotoh_optimum(q[:qo][::-1], s[:so][::-1], reward, penalty, go, ge)
            right = gotoh_optimum(q[qo:], s[so:], reward, penalty, go, ge)
            assert g["score"] == left + right, (k, reward, penalty, go, ge)
            checked += 1
    assert checked >= 120
