"""Definition-level checkers for the paths behind the diagonal HASH container (batches whose concatenated query is
longer than 8,000 bases: C2, C3, C4), sharing no structure with the oracle's restatement or the HIP kernels:

* one-hit diagonal filter + ungapped extension (CORE/na_ungapped.c:778-922, :152-351): the seed list (already pinned on
  its own definition, tests/test_oracle_definitions.py) replayed with an EXACT per-diagonal `last_hit` map -- no 512
  buckets, no chains, no stale-slot reuse -- and an ungapped extension written on prefix sums and running maxima
  instead of the reference's running `sum`.  The hash container differs from the exact map in one situation only:
  a cell that has expired (`s_off - level > 1 - word`) may be overwritten by another diagonal of its bucket, after
  which a later seed of the first diagonal that lies below the lost level is extended instead of skipped.  Such seeds
  ("in the window": skipped by the exact map with level - s_off < word) are listed; everything else must be equal.
* the acceptance loop of BLAST_GetGappedScore (CORE/blast_gapalign.c:3351-3548): containment of an initial hit by
  brute force over ALL accepted HSPs (s_HSPIsContained, CORE/blast_itree.c:814-852) instead of the interval tree.
  The tree differs from "all accepted" only through its common-endpoint eviction (CORE/blast_itree.c:280-500): cases
  where two accepted HSPs of one strand share a start or an end point are listed, everything else must be equal.
Both are run against the oracle (tests/test_hash_path_oracle.py, CPU) and the HIP path (tests/test_gpu_definitions.py)."""
import numpy as np


def context_of(ctxs, q):
    lo = 0
    for i, c in enumerate(ctxs):
        if c.query_offset <= q:
            lo = i
    return lo


def _side(step_scores, X):
    """X-drop walk over 4-base step scores by prefix sums: returns (score gained, steps of the best prefix).
    sum_t of the reference = P_t - M_t with P the prefix sums and M their running maximum (0 to start with);
    the walk stops at the first t with P_t - M_t < X; the best prefix is the first to attain the final maximum."""
    if len(step_scores) == 0:
        return 0, 0
    P = np.cumsum(np.asarray(step_scores, dtype=np.int64))
    M = np.maximum.accumulate(np.maximum(P, 0))
    stop = np.nonzero(P - M < X)[0]
    k = int(stop[0]) if len(stop) else len(P) - 1
    best = int(M[k])
    if best <= 0:
        return 0, 0
    return best, int(np.nonzero(P[:k + 1] == best)[0][0]) + 1


def ungapped_by_definition(qcat, subj, q_off, s_off, s_match_end, X, reduced_cutoff, matrix, reward, penalty):
    """s_NuclUngappedExtend / s_NuclUngappedExtendExact -> (q_start, s_start, length, score); X < 0."""
    qn, sn = len(qcat), len(subj)

    def step_score(q4, s4):
        # score_table[q_byte ^ s_byte] with q_byte put together from the unpacked codes (codes above 3 spill)
        qb = ((int(q4[0]) << 6) | (int(q4[1]) << 4) | (int(q4[2]) << 2) | int(q4[3])) & 0xff
        sb = (int(s4[0]) << 6) | (int(s4[1]) << 4) | (int(s4[2]) << 2) | int(s4[3])
        x = qb ^ sb
        mism = sum(1 for g in range(4) if (x >> (2 * g)) & 3)
        return reward * (4 - mism) + penalty * mism
    ln = (4 - (s_off % 4)) % 4
    q_ext, s_ext = q_off + ln, s_off + ln
    nl = min(q_ext, s_ext) // 4
    left = []
    for i in range(nl):
        left.append(step_score(qcat[q_ext - 4 * i - 4:q_ext - 4 * i], subj[s_ext - 4 * i - 4:s_ext - 4 * i]))
        if len(left) >= 64 and sum(left[-16:]) < 4 * X:       # (far past any stop: enough material for the walk)
            break
    gl, bl = _side(left, X)
    nr = min(qn - q_ext, sn - s_ext) // 4
    right = []
    for i in range(nr):
        right.append(step_score(qcat[q_ext + 4 * i:q_ext + 4 * i + 4], subj[s_ext + 4 * i:s_ext + 4 * i + 4]))
        if len(right) >= 64 and sum(right[-16:]) < 4 * X:
            break
    gr, br = _side(right, X)
    score = gl + gr
    uq, us = q_ext - 4 * bl, s_ext - 4 * bl
    if score < reduced_cutoff:
        new_q = q_ext + 4 * br - 1 if br else q_ext
        return uq, us, max(s_match_end - us, new_q - uq + 1), score
    # exact: base by base from (q_off, s_off) itself with the matrix
    n = min(q_off, s_off)
    sl = [int(matrix[qcat[q_off - 1 - i], subj[s_off - 1 - i]]) for i in range(n)]
    g1, b1 = _side(sl, X)
    n = min(qn - q_off, sn - s_off)
    sr = [int(matrix[qcat[q_off + i], subj[s_off + i]]) for i in range(n)]
    g2, b2 = _side(sr, X)
    return q_off - b1, s_off - b1, b1 + b2, g1 + g2


def diag_filter_by_definition(seeds, qcat, ctxs, subj, word, matrix, reward, penalty):
    """seeds: [(q_off, s_off)] of word-sized matches in scan order.  -> (init hits [(q_off, s_off, q_start, s_start,
    length, score)] in the order they are saved, seeds in the window [(q_off, s_off)])"""
    level = {}
    hits, window = [], []
    for q_off, s_off in seeds:
        d = s_off - q_off
        lv = level.get(d, 0)
        if s_off < lv:
            if lv - s_off < word:
                window.append((q_off, s_off))
            continue
        c = ctxs[context_of(ctxs, q_off)]
        u = ungapped_by_definition(qcat, subj, q_off, s_off, s_off + word, -c.x_dropoff, c.reduced_cutoff, matrix, reward, penalty)
        if u[3] >= c.cutoff_score:
            hits.append((q_off, s_off) + u)
            level[d] = u[1] + u[2]
        else:
            level[d] = s_off + word
    return hits, window


def sort_init_hits(hits):
    """Blast_InitHitListSortByScore (CORE/blast_extend.c:279-315): score desc, s_start asc, length desc, q_start asc, stable"""
    return sorted(hits, key=lambda h: (-h[5], h[3], -h[4], h[2]))


def contained(inh, tree, mds):
    """s_HSPIsContained: dicts with context, q_offset, q_end, s_offset, s_end, score (query offsets context-relative)"""
    if inh["context"] != tree["context"] or inh["score"] > tree["score"]:
        return False

    def inside(a, b, c, d, e, f):
        return a <= c <= b and d <= f <= e
    if not (inside(tree["q_offset"], tree["q_end"], inh["q_offset"], tree["s_offset"], tree["s_end"], inh["s_offset"]) and
            inside(tree["q_offset"], tree["q_end"], inh["q_end"], tree["s_offset"], tree["s_end"], inh["s_end"])):
        return False
    if mds == 0:
        return True

    def close(q1, s1, q2, s2):
        return abs((q1 - s1) - (q2 - s2)) < mds
    return close(tree["q_offset"], tree["s_offset"], inh["q_offset"], inh["s_offset"]) or \
        close(tree["q_end"], tree["s_end"], inh["q_end"], inh["s_end"])


def acceptance_by_definition(sorted_hits, ctxs, extend, mds):
    """sorted_hits: initial hits in list order (concatenated query coordinates); extend(context, hit) -> dict of the
    gapped HSP (context-relative) or None below the cutoff.  -> (accepted HSPs in acceptance order, number of gapped
    extensions, common-endpoint pairs among the accepted)"""
    accepted, n_ext = [], 0
    for h in sorted_hits:
        ci = context_of(ctxs, h[0])
        qs = ctxs[ci].query_offset
        probe = dict(context=ci, score=h[5], q_offset=h[2] - qs, q_end=h[2] - qs + h[4], s_offset=h[3], s_end=h[3] + h[4])
        if any(contained(probe, a, mds) for a in accepted):
            continue
        n_ext += 1
        g = extend(ci, h)
        if g is not None:
            accepted.append(g)
    shared = []
    for i in range(len(accepted)):
        for j in range(i):
            a, b = accepted[i], accepted[j]
            if a["context"] == b["context"] and ((a["q_offset"], a["s_offset"]) == (b["q_offset"], b["s_offset"]) or
                                                 (a["q_end"], a["s_end"]) == (b["q_end"], b["s_end"])):
                shared.append((j, i))
    return accepted, n_ext, shared
