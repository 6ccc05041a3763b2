"""a14: per-query top-N hit lists (the HSP stream's collector writer).  CPU only:
the oracle restatement against the behaviour the reference's own tests pin, and the
product's C++ collector (through the C ABI) against the oracle."""
import numpy as np
import pytest
from gblastn_amd import api
from oracle import orc


def rec(oid, context, score, evalue, q0=0, s0=0, length=50):
    return dict(oid=oid, context=context, q_offset=q0, q_end=q0 + length, q_gapped_start=q0 + 1,
                s_offset=s0, s_end=s0 + length, s_gapped_start=s0 + 1, score=score, evalue=evalue)


def to_array(recs):
    a = np.zeros(len(recs), dtype=api.HSP_DT)
    for i, r in enumerate(recs):
        for k, v in r.items():
            a[i][k] = v
    return a


def run_oracle(nq, hitlist, subjects):
    col = orc.Collector(nq, hitlist)
    for oid, recs in subjects:
        assert col.write(oid, recs) == 0
    return col.close()


def run_product(nq, hitlist, subjects, one_call=False):
    col = api.BlastHSPCollector(nq, hitlist)
    if one_call:
        col.write(to_array([r for _, recs in subjects for r in recs]))
    else:
        for _, recs in subjects:
            col.write(to_array(recs))
    hsps, starts, queries = col.close()
    out = []
    for i, q in enumerate(queries):
        seg = hsps[starts[i]:starts[i + 1]]
        out.append((int(seg[0]["oid"]), int(q), [tuple(h[f].item() for f in orc.Collector.FIELDS) for h in seg]))
    return out


def test_prelim_hitlist_size():
    # SBlastHitsParametersNew: min(2N, N+50), at least 10
    for n, want in [(500, 550), (1, 10), (10, 20), (50, 100), (51, 101), (250, 300)]:
        assert orc.lib().orc_prelim_hitlist_size(n, 1) == want
        assert api.lib().gbn_prelim_hitlist_size(n) == want


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_reference_stream_test_shape(impl):
    # UT/hspstream_unit_test.cpp:56-165 (collector flavour): 1000 HSP lists of 10 queries x 1 HSP,
    # score rand % 100, e-value 0 -> min(1000, prelim_hitlist_size = 550) lists per query, read out
    # by ascending oid with one HSP each
    rng = np.random.default_rng(5)
    nq = 10
    subjects = []
    for oid in rng.permutation(1000):       # 40 writer threads: arbitrary arrival order
        sc = int(rng.integers(0, 100))
        subjects.append((int(oid), [rec(int(oid), 2 * q, sc, 0.0) for q in range(nq)]))
    out = (run_oracle if impl == "oracle" else run_product)(nq, 500, subjects)
    assert len(out) == 550 * nq
    assert [o for o, _, _ in out] == sorted(o for o, _, _ in out)
    assert all(len(h) == 1 for _, _, h in out)
    for q in range(nq):
        kept = sorted(h[0][7] for _, qq, h in out if qq == q)
        allsc = sorted(r[0]["score"] for _, r in subjects)
        assert kept == allsc[-550:]         # the 550 best scores survive


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_reference_multiseq_collector(impl):
    # UT/hspstream_unit_test.cpp:175-228: odd subjects, one query, read back in oid order
    subjects = [(i, [rec(i, 0, i, 0.0)]) for i in range(1, 10, 2)]
    out = (run_oracle if impl == "oracle" else run_product)(1, 500, subjects)
    assert [o for o, _, _ in out] == [1, 3, 5, 7, 9]


@pytest.mark.parametrize("seed", range(12))
def test_product_equals_oracle_on_random_streams(seed):
    rng = np.random.default_rng(100 + seed)
    nq = int(rng.integers(1, 6))
    hitlist = int(rng.choice([1, 5, 12, 30]))
    nsub = int(rng.integers(20, 400))
    # few distinct scores / e-values -> many ties and fuzzy-equal e-values (relative 1e-7 .. 1e-5)
    base_e = 10.0 ** rng.integers(-30, 1, size=6).astype(np.float64)
    subjects = []
    for oid in sorted(rng.choice(5000, size=nsub, replace=False).tolist()):
        recs = []
        for _ in range(int(rng.integers(1, 2 * nq + 2))):
            k = int(rng.integers(0, 6))
            e = float(base_e[k] * (1 + rng.choice([0, 1e-7, -1e-7, 3e-6, 1e-5])))
            recs.append(rec(oid, int(rng.integers(0, 2 * nq)), int(40 + 10 * (5 - k) + rng.integers(0, 3)), e,
                            q0=int(rng.integers(0, 500)), s0=int(rng.integers(0, 5000))))
        recs.sort(key=lambda r: (-r["score"], r["s_offset"], -r["s_end"], r["q_offset"], -r["q_end"]))
        subjects.append((oid, recs))
    want = run_oracle(nq, hitlist, subjects)
    assert run_product(nq, hitlist, subjects) == want
    assert run_product(nq, hitlist, subjects, one_call=True) == want
    cap = orc.lib().orc_prelim_hitlist_size(hitlist, 1)
    for q in range(nq):
        assert sum(1 for _, qq, _ in want if qq == q) <= cap


def test_write_after_close_is_an_error():
    col = api.BlastHSPCollector(2, 5)
    col.write(to_array([rec(3, 1, 50, 1e-9)]))
    hsps, starts, queries = col.close()
    assert len(hsps) == 1 and list(queries) == [0] and list(starts) == [0, 1]
    with pytest.raises(api.BlastError):
        col.write(to_array([rec(4, 1, 50, 1e-9)]))
    with pytest.raises(api.BlastError):
        api.BlastHSPCollector(2, 5).write(to_array([rec(4, 9, 50, 1e-9)]))     # context beyond the batch
