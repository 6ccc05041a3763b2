"""CPU: the oracle against every known-answer the reference's own unit tests hold for
this path and that is reproducible offline."""
import ctypes as C
import math
import os
import numpy as np
from oracle import orc

G = os.path.join(os.path.dirname(__file__), "golden")


def test_prelimsearch_BuildCStd_seg_blastn():
    # UT/prelimsearch_unit_test.cpp:169-203 -- query = bases 54..560 of gi|41646578,
    # megablast prelim vs data/nt.41646578: first segment q 0..506 <-> s 54..560, plus/plus
    db = orc.read_blastdb_v4_nucl(os.path.join(G, "nt.41646578"))
    assert len(db) == 1 and db[0][1] == 3300
    packed, n = db[0]
    q = orc.unpack_ncbi2na(packed, n)[54:561].copy()
    s = orc.Search(orc.default_options(True, db_length=n, db_num_seqs=1), [q])
    r = s.subject(packed, n)
    h = r["hsps"]
    assert len(h) >= 1
    assert h[0]["context"] == 0                       # plus strand
    assert (h[0]["q_offset"], h[0]["q_end"] - 1) == (0, 506)
    assert (h[0]["s_offset"], h[0]["s_end"] - 1) == (54, 560)
    assert h[0]["score"] == 507
    info = s.info()
    assert (info["lut_type"], info["lut_width"], info["scan_step"]) == (1, 8, 21)


def _greedy_pair():
    a = orc.encode_blastna(orc.read_fasta(os.path.join(G, "greedy1a.fsa"))[0])
    b = orc.encode_blastna(orc.read_fasta(os.path.join(G, "greedy1b.fsa"))[0])
    return a, b


def test_bl2seq_MegablastGreedyTraceback2_score_619():
    # UT/bl2seq_unit_test.cpp:1620-1672: megablast defaults, gap 0/0, no DUST -> one
    # alignment, score 619 (bl2seq => per-subject statistics, db_num_seqs = 0)
    a, b = _greedy_pair()
    s = orc.Search(orc.default_options(True), [a])
    r = s.subject(orc.pack_ncbi2na(b), len(b))
    assert len(r["hsps"]) == 1
    h = r["hsps"][0]
    assert h["score"] == 619
    assert (h["q_offset"], h["q_end"], h["s_offset"], h["s_end"]) == (159, 874, 30, 739)


def test_bl2seq_MegablastGreedyTraceback2_score_6034():
    # UT/bl2seq_unit_test.cpp:1674-1690: reward 10, penalty -25, X-drop 100/100 -> 6034
    a, b = _greedy_pair()
    opt = orc.default_options(True, reward=10, penalty=-25, xdrop_gap_bits=100.0,
                              xdrop_gap_final_bits=100.0)
    s = orc.Search(opt, [a])
    r = s.subject(orc.pack_ncbi2na(b), len(b))
    assert len(r["hsps"]) == 1 and r["hsps"][0]["score"] == 6034


def _karlin():
    return orc.OrcKarlin()


def test_scoreblk_NuclGappedCalc():
    # UT/scoreblk_unit_test.cpp:354-530
    L = orc.lib()
    ideal = _karlin()
    assert L.orc_karlin_ideal(1, -2, C.byref(ideal)) == 0
    k = _karlin(); rd = C.c_int(0)
    assert L.orc_karlin_nucl_gapped(3, 1, 1, -2, C.byref(ideal), C.byref(k), C.byref(rd)) == 0
    assert rd.value == 0
    assert math.isclose(k.Lambda, 1.32, rel_tol=1e-5) and math.isclose(k.K, 0.57, rel_tol=1e-5)
    assert math.isclose(k.logK, -0.562, rel_tol=1e-3)
    al, be = C.c_double(), C.c_double()
    L.orc_nucl_alpha_beta(1, -2, 3, 1, C.byref(ideal), 1, C.byref(al), C.byref(be))
    assert math.isclose(al.value, 1.3, rel_tol=1e-5) and math.isclose(be.value, -1.0, rel_tol=1e-5)
    # gap costs in the "infinite" regime copy the ungapped block
    assert L.orc_karlin_nucl_gapped(4, 2, 1, -2, C.byref(ideal), C.byref(k), C.byref(rd)) == 0
    assert (k.Lambda, k.K, k.logK) == (ideal.Lambda, ideal.K, ideal.logK)
    L.orc_nucl_alpha_beta(1, -2, 4, 2, C.byref(ideal), 1, C.byref(al), C.byref(be))
    assert math.isclose(al.value, ideal.Lambda / ideal.H, rel_tol=1e-12) and be.value == 0.0
    # scaled-up scores
    assert L.orc_karlin_nucl_gapped(30, 10, 10, -20, C.byref(ideal), C.byref(k), C.byref(rd)) == 0
    assert math.isclose(k.Lambda, 0.132, rel_tol=1e-5) and math.isclose(k.K, 0.57, rel_tol=1e-5)
    # 2/-7 rounds odd scores down
    assert L.orc_karlin_nucl_gapped(4, 2, 2, -7, C.byref(ideal), C.byref(k), C.byref(rd)) == 0
    assert rd.value == 1
    assert math.isclose(k.Lambda, 0.675, rel_tol=1e-5) and math.isclose(k.K, 0.62, rel_tol=1e-5)
    assert math.isclose(k.logK, -0.478036, rel_tol=1e-5)
    # unsupported gap costs / substitution scores
    assert L.orc_karlin_nucl_gapped(3, 2, 4, -5, C.byref(ideal), C.byref(k), C.byref(rd)) == 1
    assert L.orc_karlin_nucl_gapped(1, 3, 1, -2, C.byref(ideal), C.byref(k), C.byref(rd)) == 1
    assert L.orc_karlin_nucl_gapped(1, 3, 2, -1, C.byref(ideal), C.byref(k), C.byref(rd)) == -1
    # ungapped alpha/beta for 2/-3
    L.orc_nucl_alpha_beta(2, -3, 0, 0, C.byref(ideal), 0, C.byref(al), C.byref(be))
    assert math.isclose(al.value, ideal.Lambda / ideal.H, rel_tol=1e-12) and be.value == -2.0


def test_scoreblk_EqualRewardPenaltyLHtoK():
    # UT/scoreblk_unit_test.cpp:337-352: reward 2 / penalty -2 -> ideal K = 1/3
    k = _karlin()
    assert orc.lib().orc_karlin_ideal(2, -2, C.byref(k)) == 0
    assert abs(k.K - 1.0 / 3) < 1e-6


def test_gapped_tables_defaults():
    # megablast 1/-2 gap 0/0 and blastn 2/-3 gap 5/2
    L = orc.lib()
    ideal = _karlin(); L.orc_karlin_ideal(1, -2, C.byref(ideal))
    k = _karlin(); rd = C.c_int(0)
    L.orc_karlin_nucl_gapped(0, 0, 1, -2, C.byref(ideal), C.byref(k), C.byref(rd))
    assert (k.Lambda, k.K, k.H, rd.value) == (1.28, 0.46, 0.85, 0)
    L.orc_karlin_ideal(2, -3, C.byref(ideal))
    L.orc_karlin_nucl_gapped(5, 2, 2, -3, C.byref(ideal), C.byref(k), C.byref(rd))
    assert (k.Lambda, k.K, k.H, rd.value) == (0.625, 0.41, 0.78, 1)


def _hsps(rows):
    arr = (orc.OrcHSP * len(rows))()
    for i, (qo, qe, so, se, sc) in enumerate(rows):
        arr[i].context = 0; arr[i].q_offset = qo; arr[i].q_end = qe
        arr[i].s_offset = so; arr[i].s_end = se; arr[i].score = sc
    return arr


def test_blasthits_testCheckHSPCommonEndpoints():
    # UT/blasthits_unit_test.cpp:1114-1171: 9 HSPs -> 3 survivors (original indices 4, 0, 6)
    scores = [1044, 995, 965, 219, 160, 125, 110, 107, 103]
    qo = [2, 2, 2, 236, 88, 259, 278, 259, 278]
    qe = [322, 336, 300, 322, 182, 322, 341, 341, 341]
    so = [7, 7, 7, 194, 2, 194, 197, 194, 197]
    se = [292, 293, 301, 292, 96, 292, 260, 260, 266]
    arr = _hsps(list(zip(qo, qe, so, se, scores)))
    n = orc.lib().orc_hsplist_purge_common_endpoints(arr, 9)
    assert n == 3
    for i, orig in enumerate([4, 0, 6]):
        assert (arr[i].score, arr[i].q_offset, arr[i].s_offset, arr[i].q_end, arr[i].s_end) == \
               (scores[orig], qo[orig], so[orig], qe[orig], se[orig])


def test_blasthits_sort_by_score_rule():
    # CORE/blast_hits.c:1182-1208: score desc, s.offset asc, s.end desc, q.offset asc, q.end desc
    rows = [(10, 50, 30, 70, 40), (5, 45, 20, 60, 40), (5, 45, 20, 65, 40), (0, 90, 0, 90, 90),
            (6, 45, 20, 65, 40)]
    arr = _hsps(rows)
    orc.lib().orc_hsplist_sort_by_score(arr, len(rows))
    got = [(a.q_offset, a.q_end, a.s_offset, a.s_end, a.score) for a in arr]
    assert got == [(0, 90, 0, 90, 90), (5, 45, 20, 65, 40), (6, 45, 20, 65, 40),
                   (5, 45, 20, 60, 40), (10, 50, 30, 70, 40)]


def test_blastn_word_size4_invariants():
    # UT/bl2seq_unit_test.cpp:2246-2302 (NucleotideBlastWordSize4): structural invariants
    a = orc.encode_blastna(orc.read_fasta(os.path.join(G, "blastn_size4a.fsa"))[0])
    b = orc.encode_blastna(orc.read_fasta(os.path.join(G, "blastn_size4b.fsa"))[0])
    opt = orc.default_options(False, word_size=4, reward=1, penalty=-1, evalue=10000.0)
    s = orc.Search(opt, [a])
    assert s.info()["lut_width"] == 4 and s.info()["scan_step"] == 1
    r = s.subject(orc.pack_ncbi2na(b), len(b))
    for h in r["hsps"]:
        assert 0 <= h["q_offset"] < h["q_end"] <= len(a)
        assert 0 <= h["s_offset"] < h["s_end"] <= len(b)
        assert h["q_offset"] <= h["q_gapped_start"] <= h["q_end"]
        assert h["s_offset"] <= h["s_gapped_start"] <= h["s_end"]


def test_lookup_every_scan_hit_is_a_word_match():
    # UT/ntscan_unit_test.cpp:717 property: every seed's word matches exactly on both sides
    rng = np.random.default_rng(3)
    q = rng.integers(0, 4, 900, dtype=np.uint8)
    subj = rng.integers(0, 4, 30000, dtype=np.uint8)
    subj[5000:5600] = q[100:700]
    for mb, ws in [(True, 28), (False, 11)]:
        s = orc.Search(orc.default_options(mb, db_length=len(subj), db_num_seqs=1), [q])
        r = s.subject(orc.pack_ncbi2na(subj), len(subj))
        qc = s.query_concat()
        assert len(r["seeds"]) > 0
        for sd in r["seeds"]:
            assert np.array_equal(qc[sd["q_off"]:sd["q_off"] + ws], subj[sd["s_off"]:sd["s_off"] + ws])


def test_lookup_table_choice_thresholds():
    # CORE/blast_nalookup.c:51-189 incl. G-BLASTN's word_size==11 patch vs stock
    rng = np.random.default_rng(1)

    def choice(mb, nq, **kw):
        qs = [rng.integers(0, 4, 1000, dtype=np.uint8) for _ in range(nq)]
        s = orc.Search(orc.default_options(mb, db_length=10**7, db_num_seqs=10, **kw), qs)
        i = s.info()
        return i["lut_type"], i["lut_width"], i["scan_step"], i["container"]
    assert choice(True, 1) == (1, 8, 21, 0)
    assert choice(True, 4) == (1, 8, 21, 1)        # 7,992 entries < 8,500; 8,007 bases > 8,000
    assert choice(True, 5) == (3, 11, 18, 1)
    assert choice(True, 160) == (3, 12, 17, 1)
    assert choice(False, 2) == (1, 8, 4, 0)
    assert choice(False, 8) == (3, 11, 1, 1)
    assert choice(False, 8, lut11_gblastn_rule=0) == (3, 10, 2, 1)


def test_affine_greedy_reduces_to_linear_when_costs_coincide():
    # no offline golden exists for BLAST_AffineGreedyAlign (UT/bl2seq_unit_test.cpp:1575 needs
    # GenBank).  Cross-check of the two restated code paths: with reward 2 / penalty -4 the
    # non-affine greedy charges reward/2 - penalty = 5 per gap base, so the affine routine
    # with gap_open 0 / gap_extend 5 must return the same box, score and seed.
    L = orc.lib()
    rng = np.random.default_rng(11)
    for trial in range(20):
        subj = rng.integers(0, 4, 3000, dtype=np.uint8)
        q = subj[500:2300].copy()
        for _ in range(25):
            p = int(rng.integers(5, len(q) - 5)); q[p] = (q[p] + 1) & 3
        for _ in range(6):
            p = int(rng.integers(5, len(q) - 5))
            q = np.delete(q, p) if rng.random() < 0.5 else np.insert(q, p, rng.integers(0, 4))
        packed = orc.pack_ncbi2na(subj)
        qa = np.ascontiguousarray(q, dtype=np.uint8)
        a, b = orc.OrcHSP(), orc.OrcHSP()
        assert L.orc_greedy_extend(qa.ctypes.data, len(qa), packed.ctypes.data, len(subj), 900, 1400, 30,
                                   2, -4, 0, 0, C.byref(a)) == 0
        assert L.orc_greedy_extend(qa.ctypes.data, len(qa), packed.ctypes.data, len(subj), 900, 1400, 30,
                                   2, -4, 0, 5, C.byref(b)) == 0
        for f in ["q_offset", "q_end", "s_offset", "s_end", "score", "q_gapped_start", "s_gapped_start"]:
            assert getattr(a, f) == getattr(b, f), (trial, f, getattr(a, f), getattr(b, f))


# ---- the reference's offline known answers for the tables and containers C2 / C3 run on ------------------------------

def _debruijn_query(n):
    """UT/ntlookup_unit_test.cpp:147-170 (debruijnInit): the (n, 4) de Bruijn sequence followed by its first n - 1
    letters, 4^n + n - 1 bases holding every n-mer exactly once, with a sentinel byte either side."""
    L = orc.lib()
    L.orc_debruijn.argtypes = [C.c_int32, C.c_int32, C.c_void_p]
    ln = 4 ** n + n - 1
    buf = np.full(ln + 2, 15, dtype=np.uint8)
    L.orc_debruijn(n, 4, buf.ctypes.data + 1)
    buf[1 + 4 ** n:1 + ln] = buf[1:n]
    return buf, ln


def _lookup_probe(opt, buf, ln):
    L = orc.lib()
    L.orc_lookup_probe.argtypes = [C.POINTER(orc.OrcOptions), C.c_void_p, C.c_int32, C.c_void_p]
    out = np.zeros(12, dtype=np.int64)
    assert L.orc_lookup_probe(C.byref(opt), buf.ctypes.data + 1, ln, out.ctypes.data) == 0
    return dict(zip(["type", "cells", "word", "lut", "step", "pv_bts", "longest_chain", "single", "empty",
                     "chained", "pv_not_full", "overflow"], [int(v) for v in out]))


def test_debruijn_generator_holds_every_word_once():
    # CORE/lookup_util.c:100-190 restated (orc_debruijn): every n-mer of the cyclic sequence is distinct
    for n in (2, 3, 5, 8):
        buf, ln = _debruijn_query(n)
        seq = buf[1:1 + ln].astype(np.int64)
        assert seq.max() <= 3
        code = np.zeros(4 ** n, dtype=np.int64)
        for k in range(n):
            code = code * 4 + seq[k:k + 4 ** n]
        assert len(np.unique(code)) == 4 ** n


def test_ntlookup_testStdLookupTableDebruijn():
    # UT/ntlookup_unit_test.cpp:468-509: blastn, word_size 8, the (8, 4) de Bruijn query of 65,543 bases ->
    # eNaLookupTable, 65,536 cells, every cell exactly one entry, longest chain 1, nothing in the overflow array,
    # every presence bit set
    buf, ln = _debruijn_query(8)
    assert ln == 65536 + 7
    t = _lookup_probe(orc.default_options(False, word_size=8), buf, ln)
    assert t["type"] == 2                       # ORC_LUT_NA = eNaLookupTable
    assert t["cells"] == 65536 and t["lut"] == 8 and t["word"] == 8
    assert t["single"] == 65536 and t["empty"] == 0
    assert t["longest_chain"] == 1 and t["overflow"] == 0
    assert t["pv_not_full"] == 0


def test_ntlookup_testMegablastLookupTableDebruijn():
    # UT/ntlookup_unit_test.cpp:511-556: megablast defaults, the (12, 4) de Bruijn query of 16,777,227 bases ->
    # eMBLookupTable, 4^12 cells, word 28, the reference's chain estimate 2 ("an overestimate, should be 1"),
    # pv_array_bts 10, next_pos all 0 (no word shares a cell), every presence bit set
    buf, ln = _debruijn_query(12)
    assert ln == 4 ** 12 + 11
    t = _lookup_probe(orc.default_options(True), buf, ln)
    assert t["type"] == 3                       # ORC_LUT_MB = eMBLookupTable
    assert t["cells"] == 16777216 and t["word"] == 28 and t["lut"] == 12 and t["step"] == 17
    assert t["longest_chain"] == 2
    assert t["pv_bts"] == 10
    assert t["chained"] == 0 and t["single"] == 16777216 and t["empty"] == 0
    assert t["pv_not_full"] == 0


def test_blastdiag_ExtendWordExit_known_answers():
    # UT/blastdiag_unit_test.cpp:45-150 (testDiagClear, testDiagUpdateFull, testDiagUpdateNotFull): query 100,
    # window 20 -> array of 128 cells; what Blast_ExtendWordExit does to the offset and the cells, i.e. the
    # bookkeeping orc_search_carry_diag runs on (oracle/orc_wordfinder.c: orc_extend_word_exit)
    L = orc.lib()
    L.orc_extend_word_exit.argtypes = [C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]
    INT4_MAX, window, slen, n = 2 ** 31 - 1, 20, 100, 128
    assert n == 1 << int(np.ceil(np.log2(100 + window)))     # s_BlastDiagTableNew: the power of two past qlen + window
    for start, cleared in [(INT4_MAX // 4, True), (INT4_MAX // 4 + 1000, True), (100, False)]:
        last_hit = np.full(n, 40, dtype=np.int32); flag = np.ones(n, dtype=np.uint32)
        off = C.c_int32(start)
        r = L.orc_extend_word_exit(C.byref(off), window, slen, last_hit.ctypes.data, flag.ctypes.data, n)
        if cleared:
            assert r == 1 and off.value == window
            assert (last_hit == -window).all() and (flag == 0).all()
        else:
            assert r == 0 and off.value == start + slen + window
            assert (last_hit == 40).all() and (flag == 1).all()


def test_blastoptions_testExtensionParamsNew():
    # UT/blastoptions_unit_test.cpp:761-810: gapped Lambda 1.30 (MakeSomeValidKBP :741-757), X-drop 20 / 22 bits ->
    # 10 / 11 raw; 25 / 22 -> 13 / 13 (the final value is never below the preliminary one)
    L = orc.lib()
    L.orc_extension_params.argtypes = [C.c_double, C.c_double, C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    x, xf = C.c_int32(), C.c_int32()
    L.orc_extension_params(1.30, 20.0, 22.0, C.byref(x), C.byref(xf))
    assert (x.value, xf.value) == (10, 11)
    L.orc_extension_params(1.30, 25.0, 22.0, C.byref(x), C.byref(xf))
    assert (x.value, xf.value) == (13, 13)
    # and through the search set-up: blastn defaults 30 / 100 bits at the gapped Lambda of 2 / -3, 5 / 2
    s = orc.Search(orc.default_options(False, db_length=10**6, db_num_seqs=1), [np.zeros(50, dtype=np.uint8)])
    lam = orc.lib().orc_gap_lambda(s._h)
    assert s.info()["gap_x_dropoff"] == int(30 * math.log(2) / lam)


def test_blastoptions_GetNucleotideGapExistenceExtendParams():
    # UT/blastoptions_unit_test.cpp:184-231
    L = orc.lib()
    L.orc_nucl_gap_params.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]

    def params(reward, penalty, ex, ext):
        a, b = C.c_int(ex), C.c_int(ext)
        st = L.orc_nucl_gap_params(reward, penalty, C.byref(a), C.byref(b))
        return st, a.value, b.value
    assert params(0, 3, -1, -1)[0] == -1
    assert params(1, -3, 0, 0) == (0, 0, 0)         # megablast linear values
    assert params(1, -3, -1, -1) == (0, 2, 2)
    assert params(2, -5, -1, -1) == (0, 4, 4)
    assert params(1, -2, -1, -1) == (0, 2, 2)


def masks_from_table_cover(s, queries):
    """what gblastn_amd/shim s_MasksFromTable does with the reference's table when it keeps no masked_locations
    (lookup word = search word): the complement, per plus strand, of the positions inside an indexed word"""
    L = orc.lib()
    L.orc_search_indexed_cover.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_query_concat_len.restype = C.c_int32; L.orc_query_concat_len.argtypes = [C.c_void_p]
    cover = np.zeros(L.orc_query_concat_len(s._h), dtype=np.uint8)
    L.orc_search_indexed_cover(s._h, cover.ctypes.data)
    ctx = s.contexts
    masks = []
    for qi in range(len(queries)):
        c = ctx[2 * qi]
        cov = cover[c.query_offset:c.query_offset + c.query_length]
        p = 0
        while p < len(cov):
            if cov[p]:
                p += 1; continue
            e = p
            while e + 1 < len(cov) and not cov[e + 1]:
                e += 1
            masks.append((qi, p, e)); p = e + 1
    return masks


def test_masks_rebuilt_from_the_table_give_the_table_back():
    # blastn, 16 queries: word 11 = lut 11 (the C3 shape), where the reference keeps no masked_locations.  Soft masks
    # (DUST-like intervals, one short stretch between two masks, one next to an ambiguity code) -> table -> cover ->
    # masks' -> the same seeds, initial hits and HSPs
    rng = np.random.default_rng(17)
    queries = [rng.integers(0, 4, 1000, dtype=np.uint8) for _ in range(16)]
    queries[3][500] = 14                                  # an ambiguity code
    subj = rng.integers(0, 4, 60000, dtype=np.uint8)
    for k, q in enumerate(queries[:8]):
        subj[2000 + 3000 * k:2000 + 3000 * k + 600] = np.where(q[200:800] > 3, 0, q[200:800])
    masks = [(0, 250, 330), (0, 336, 400), (1, 0, 120), (2, 900, 999), (3, 480, 495), (5, 300, 700)]
    opt = orc.default_options(False, db_length=len(subj), db_num_seqs=1)
    a = orc.Search(opt, queries, masks=masks)
    assert (a.info()["lut_width"], a.info()["scan_step"]) == (11, 1)
    rebuilt = masks_from_table_cover(a, queries)
    assert rebuilt != sorted(masks)                       # (short stretches and the ambiguity widen them)
    b = orc.Search(opt, queries, masks=rebuilt)
    packed = orc.pack_ncbi2na(subj)
    ra, rb = a.subject(packed, len(subj)), b.subject(packed, len(subj))
    assert len(ra["seeds"]) > 1000 and len(ra["hsps"]) >= 8
    for k in ("seeds", "init_hits", "hsps"):
        assert ra[k].tobytes() == rb[k].tobytes(), k
    # and an unmasked search differs: the masks matter in this case
    rc = orc.Search(opt, queries).subject(packed, len(subj))
    assert len(rc["seeds"]) != len(ra["seeds"])
