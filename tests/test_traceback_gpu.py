"""-m gpu: the traceback stage (host threads behind the C ABI, gbn_traceback_run) on the lists the HIP
preliminary stage + collector produce, against the oracle's traceback of the oracle's lists: final
coordinates, scores, e-value bits, identities, gap counts and the edit scripts themselves."""
import numpy as np
import pytest
from oracle import orc
from gblastn_amd import api
from tests import util

pytestmark = pytest.mark.gpu


def oracle_final(opt, queries, subjects):
    """oracle: preliminary search -> collector -> traceback per (oid, query) list -> {(query, oid): [dict]}"""
    ora, s = util.oracle_run(opt, queries, subjects)
    col = orc.Collector(len(queries), opt.hitlist_size)
    for oid, o in enumerate(ora):
        if len(o["hsps"]):
            col.write(oid, [dict(zip(o["hsps"].dtype.names, r)) for r in o["hsps"]])
    out = {}
    for oid, q, hs in col.close():
        bases = orc.unpack_ncbi2na(subjects[oid][0], subjects[oid][1])
        fin = s.traceback(bases, [dict(zip(orc.Collector.FIELDS, h)) for h in hs])
        if fin:
            out[(q, oid)] = fin
    return out, s


def product_final(opt, queries, subjects, threads=0):
    src = api.BlastSeqSrc.from_packed(subjects)
    ps = api.BlastPrelimSearch(queries, opt, src)
    pre = ps.run()["hsps"]
    col = api.BlastHSPCollector(len(queries), opt.hitlist_size)
    col.write(pre)
    hsps, starts, _ = col.close()
    tb = api.BlastTracebackSearch(ps, src)
    rec, ops, qstarts = tb.run(hsps, starts, threads=threads)
    out = {}
    for r, o in zip(rec, ops):
        out.setdefault((int(r["context"]) // 2, int(r["oid"])), []).append((r, o))
    return out, rec, qstarts, ps


def compare(prod, ora):
    assert set(prod.keys()) == set(ora.keys()), (sorted(prod.keys())[:5], sorted(ora.keys())[:5])
    n = 0
    for key, lst in ora.items():
        got = prod[key]
        assert len(got) == len(lst), key
        for (r, ops), f in zip(got, lst):
            for a, b in [("context", "context"), ("q_offset", "q_offset"), ("q_end", "q_end"), ("s_offset", "s_offset"),
                         ("s_end", "s_end"), ("score", "score"), ("num_ident", "num_ident"), ("gaps", "gaps"),
                         ("gap_opens", "gap_opens"), ("align_length", "align_length"),
                         ("q_gapped_start", "q_gapped_start"), ("s_gapped_start", "s_gapped_start")]:
                assert int(r[a]) == int(f[b]), (key, a, int(r[a]), int(f[b]))
            assert ops == f["ops"], key
            assert np.float64(r["evalue"]).view(np.uint64) == np.float64(f["evalue"]).view(np.uint64)
            assert np.float64(r["bit_score"]).view(np.uint64) == np.float64(f["bit_score"]).view(np.uint64)
            n += 1
    return n


@pytest.mark.parametrize("task", ["megablast", "blastn"])
def test_traceback_equals_oracle(task):
    db, queries, plants, subjects, opt = util.small_case(6, 150_000, 24 if task == "megablast" else 8, task=task, planted_fraction=0.8)
    ora, _ = oracle_final(opt, queries, subjects)
    prod, rec, qstarts, ps = product_final(opt, queries, subjects)
    n = compare(prod, ora)
    assert n >= len(plants) // 2
    # per query: subjects in the order of the reference's results (best e-value, best score, oid descending)
    for qi in range(len(queries)):
        seg = rec[qstarts[qi]:qstarts[qi + 1]]
        oids = [int(o) for o in seg["oid"]]
        firsts = [i for i in range(len(oids)) if i == 0 or oids[i] != oids[i - 1]]
        best = [float(seg["evalue"][i:(firsts[k + 1] if k + 1 < len(firsts) else len(oids))].min()) for k, i in enumerate(firsts)]
        assert all(best[i] <= best[i + 1] * (1 + 1e-6) for i in range(len(best) - 1))


def test_traceback_long_subject_uses_a_window():
    # subjects of 90,000 bases and more are traced inside the stretch an extension can reach (AdjustSubjectRange)
    db, queries, plants, subjects, opt = util.small_case(2, 400_000, 12, planted_fraction=1.0)
    ora, _ = oracle_final(opt, queries, subjects)
    prod, *_ = product_final(opt, queries, subjects, threads=3)
    assert compare(prod, ora) >= 6


def test_traceback_option_sweep():
    for kw in [dict(reward=2, penalty=-3, gap_open=5, gap_extend=2, greedy=0, xdrop_gap_bits=30.0),
               dict(reward=1, penalty=-3), dict(reward=1, penalty=-1, gap_open=3, gap_extend=2, greedy=0, xdrop_gap_bits=30.0),
               # megablast with explicit gap costs: affine greedy, preliminary (HIP) and with traceback (host)
               dict(reward=1, penalty=-2, gap_open=2, gap_extend=2), dict(reward=1, penalty=-3, gap_open=5, gap_extend=2),
               dict(reward=2, penalty=-3, gap_open=5, gap_extend=2, greedy=1)]:
        db, queries, plants, subjects, opt = util.small_case(3, 120_000, 10, planted_fraction=0.9, **kw)
        ora, _ = oracle_final(opt, queries, subjects)
        prod, *_ = product_final(opt, queries, subjects)
        assert compare(prod, ora) >= 3


def test_pipeline_streams_batches_with_traceback_overlapped():
    """C4 shape in small: 24 query batches streamed through the host pipeline (set-up, preliminary search on
    the GPU, traceback consumer threads, all overlapped); every batch's rows equal the oracle's."""
    nsub, slen, nbatch, per = 4, 120_000, 24, 6
    db, queries, plants, subjects, opt = util.small_case(nsub, slen, nbatch * per, planted_fraction=0.6)
    src = api.BlastSeqSrc.from_packed(subjects)
    pipe = api.SearchPipeline(opt, src, trace_threads=3, traceback=True, overlap=True)
    for k in range(nbatch):
        assert pipe.submit(queries[k * per:(k + 1) * per]) == k
    pipe.finish()
    got = {}
    while True:
        r = pipe.next()
        if r is None:
            break
        k, (rec, ops, qstarts), d = r
        assert k == len(got)                                  # submission order
        got[k] = (rec, ops)
    assert len(got) == nbatch
    pipe.close()
    total = 0
    for k in range(nbatch):
        ora, _ = oracle_final(opt, queries[k * per:(k + 1) * per], subjects)
        prod = {}
        for r, o in zip(*got[k]):
            prod.setdefault((int(r["context"]) // 2, int(r["oid"])), []).append((r, o))
        total += compare(prod, ora)
    assert total >= nbatch * per // 3


def test_greedy_traceback_with_ambiguity_codes_and_long_exact_runs():
    """Round 6: the greedy half compares match runs eight bases at a time (a byte is a mismatch where the letters differ or the
    query's is no base) and walks its rows through pointers.  Queries with ambiguity codes inside and at the ends of their homologies,
    homologies that are exact for hundreds of bases (runs that cross many 8-byte words from every alignment of the two sequences),
    short queries (runs that end within the last 8 bases of a sequence) -- rows and edit scripts against the oracle's."""
    db, queries, plants, subjects, opt = util.small_case(5, 130_000, 30, planted_fraction=1.0)
    rng = np.random.default_rng(5)
    queries = [np.array(q, dtype=np.uint8) for q in queries]
    bases = [orc.unpack_ncbi2na(s[0], s[1]) for s in subjects]
    for i, q in enumerate(queries):
        kind = i % 5
        if kind == 0:                                   # ambiguity codes sprinkled over the query
            q[rng.integers(0, len(q), 6)] = 14
        elif kind == 1:                                 # an exact copy of a subject stretch, offset by every residue mod 8
            s = bases[i % len(bases)]; at = 1000 + 37 * i + (i // 5)
            q[:] = s[at:at + len(q)]
        elif kind == 2:                                 # exact copy, reverse strand, with one N in the middle and one at the very end
            s = bases[i % len(bases)]; at = 5000 + 11 * i
            q[:] = (3 - s[at:at + len(q)])[::-1]
            q[len(q) // 2] = 14; q[-1] = 14
        elif kind == 3:                                 # a short query: its runs end inside the last word
            queries[i] = q[:61 + i].copy()
        # kind 4: as planted
    ora, _ = oracle_final(opt, queries, subjects)
    prod, *_ = product_final(opt, queries, subjects, threads=2)
    assert compare(prod, ora) >= 20
