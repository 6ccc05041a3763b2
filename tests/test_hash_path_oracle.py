"""CPU: the oracle on hash-container sized batches against the definition-level checkers of tests/hash_path_definitions.py
(one-hit filter + ungapped extension with an exact per-diagonal map; acceptance loop with brute-force containment)."""
import numpy as np
import pytest
from oracle import orc
from tests import hash_path_definitions as hp
from tests.test_oracle_definitions import brute_force_seeds, seed_case


def hash_case(task, nq, seed):
    sub, queries = seed_case(nq, seed)
    if task == "blastn":
        opt = orc.default_options(False, db_length=10**7, db_num_seqs=10, word_size=11)
    else:
        opt = orc.default_options(True, db_length=10**7, db_num_seqs=10)
    S = orc.Search(opt, queries)
    return sub, queries, opt, S


def check_filter_and_extension(S, opt, sub, got_init_hits, seeds=None):
    """got_init_hits: records with q_off, s_off, q_start, s_start, length, score in LIST order (sorted).  Returns the
    number of seeds in the window (where the hash container may differ from the exact map)."""
    info = S.info()
    assert info["container"] == 1
    qcat = S.query_concat()
    ctxs = S.contexts
    if seeds is None:
        seeds = brute_force_seeds(qcat, ctxs, sub, opt.word_size, info["lut_width"], info["scan_step"], descending=info["lut_type"] == 3)
    want, window = hp.diag_filter_by_definition(seeds, qcat, ctxs, sub, opt.word_size, S.matrix(), opt.reward, opt.penalty)
    want = hp.sort_init_hits(want)
    got = [tuple(int(h[f]) for f in ("q_off", "s_off", "q_start", "s_start", "length", "score")) for h in got_init_hits]
    if not window:
        assert got == want
    else:       # seeds in the window may or may not have been extended by the hash container: everything else equal
        win = set(window)
        assert [h for h in got if (h[0], h[1]) not in win] == [h for h in want if (h[0], h[1]) not in win]
    return len(window), want


@pytest.mark.parametrize("task,nq,seed", [("megablast", 24, 3), ("blastn", 9, 5), ("blastn", 6, 8)])
def test_one_hit_hash_filter_and_ungapped_extension_equal_the_definition(task, nq, seed):
    sub, queries, opt, S = hash_case(task, nq, seed)
    r = S.subject(orc.pack_ncbi2na(sub), len(sub))
    nwin, want = check_filter_and_extension(S, opt, sub, r["init_hits"])
    assert len(want) >= 3
    # the definition's input is the oracle's own seed list as well (pinned on its definition in test_oracle_definitions)
    got_seeds = list(zip(r["seeds"]["q_off"].tolist(), r["seeds"]["s_off"].tolist()))
    check_filter_and_extension(S, opt, sub, r["init_hits"], seeds=got_seeds)


def check_acceptance(S, opt, sub, init_hits, got_hsps, got_extensions=None):
    """init_hits in list order; got_hsps: the subject's final list.  The definition's accepted HSPs go through the same
    list rules (purge, rounding, sort, e-value reap -- pinned on UT/blasthits_unit_test.cpp elsewhere) by way of the
    comparison being made on the accepted set BEFORE them: every final HSP must be an accepted one, and the accepted
    ones that are missing must have been purged for a common endpoint or reaped."""
    ctxs = S.contexts
    qcat = S.query_concat()
    info = S.info()

    def extend(ci, h):
        c = ctxs[ci]
        q = qcat[c.query_offset:c.query_offset + c.query_length]
        if opt.greedy:
            q_off = h[2] - c.query_offset + h[4] // 2; s_off = h[3] + h[4] // 2
        else:
            q_off, s_off = h[0] - c.query_offset, h[1]
            if h[3] + h[4] >= s_off + 8:
                q_off += 3; s_off += 3
        g = orc.gapped_extend(q, sub, q_off, s_off, info["gap_x_dropoff"], opt.reward, opt.penalty, opt.gap_open, opt.gap_extend, greedy=bool(opt.greedy))
        if g["score"] < c.gap_cutoff_score:
            return None
        g["context"] = ci
        return g
    hits = [tuple(int(h[f]) for f in ("q_off", "s_off", "q_start", "s_start", "length", "score")) for h in init_hits]
    accepted, n_ext, shared = hp.acceptance_by_definition(hits, ctxs, extend, opt.min_diag_separation)
    if got_extensions is not None and not shared:
        assert got_extensions == n_ext
    acc = set((a["context"], a["q_offset"], a["q_end"], a["s_offset"], a["s_end"]) for a in accepted)
    round_down = opt.reward % 2 == 0                     # (reward 2: odd scores rounded down, CORE/blast_hits.c:2734-2750)
    score_of = {(a["context"], a["q_offset"], a["q_end"], a["s_offset"], a["s_end"]): (a["score"] & ~1 if round_down else a["score"]) for a in accepted}
    got = [(int(h["context"]), int(h["q_offset"]), int(h["q_end"]), int(h["s_offset"]), int(h["s_end"])) for h in got_hsps]
    if not shared:
        for k, h in zip(got, got_hsps):
            assert k in acc and int(h["score"]) == score_of[k], k
        # an accepted HSP that is not in the final list: reaped by e-value (none is purged: no common endpoints)
        missing = acc - set(got)
        assert all(score_of[k] < max(int(h["score"]) for h in got_hsps) for k in missing) if len(got_hsps) else True
    return len(accepted), len(shared)


@pytest.mark.parametrize("task,nq,seed", [("megablast", 24, 3), ("blastn", 9, 5), ("blastn", 6, 8)])
def test_acceptance_loop_equals_brute_force_containment(task, nq, seed):
    sub, queries, opt, S = hash_case(task, nq, seed)
    before = S.stats.gapped_extensions
    r = S.subject(orc.pack_ncbi2na(sub), len(sub))
    nacc, nshared = check_acceptance(S, opt, sub, r["init_hits"], r["hsps"], got_extensions=int(S.stats.gapped_extensions - before))
    assert nacc >= 2
