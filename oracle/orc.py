"""ctypes binding of the CPU oracle (oracle/liborc.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package gblastn_amd/.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OrcOptions(C.Structure):
    _fields_ = [("word_size", C.c_int32), ("reward", C.c_int32), ("penalty", C.c_int32),
                ("gap_open", C.c_int32), ("gap_extend", C.c_int32), ("greedy", C.c_int32),
                ("xdrop_ungap_bits", C.c_double), ("gap_trigger_bits", C.c_double),
                ("xdrop_gap_bits", C.c_double), ("xdrop_gap_final_bits", C.c_double),
                ("evalue", C.c_double), ("min_diag_separation", C.c_int32),
                ("hitlist_size", C.c_int32), ("cutoff_score", C.c_int32),
                ("lut11_gblastn_rule", C.c_int32), ("db_length", C.c_int64),
                ("db_num_seqs", C.c_int32)]


class OrcContext(C.Structure):
    _fields_ = [("query_offset", C.c_int32), ("query_length", C.c_int32), ("frame", C.c_int32),
                ("query_index", C.c_int32), ("is_valid", C.c_int32),
                ("length_adjustment", C.c_int32), ("eff_searchsp", C.c_int64),
                ("lambda_u", C.c_double), ("K_u", C.c_double), ("logK_u", C.c_double),
                ("H_u", C.c_double), ("x_dropoff", C.c_int32), ("cutoff_score", C.c_int32),
                ("reduced_cutoff", C.c_int32), ("gap_cutoff_score", C.c_int32),
                ("gap_cutoff_score_max", C.c_int32)]


class OrcSeed(C.Structure):
    _fields_ = [("q_off", C.c_int32), ("s_off", C.c_int32)]


class OrcInitHit(C.Structure):
    _fields_ = [("q_off", C.c_int32), ("s_off", C.c_int32), ("q_start", C.c_int32),
                ("s_start", C.c_int32), ("length", C.c_int32), ("score", C.c_int32)]


class OrcHSP(C.Structure):
    _fields_ = [("context", C.c_int32), ("q_offset", C.c_int32), ("q_end", C.c_int32),
                ("q_gapped_start", C.c_int32), ("s_offset", C.c_int32), ("s_end", C.c_int32),
                ("s_gapped_start", C.c_int32), ("score", C.c_int32), ("evalue", C.c_double)]


class OrcStats(C.Structure):
    _fields_ = [("lookup_hits", C.c_int64), ("init_extends", C.c_int64),
                ("good_init_extends", C.c_int64), ("gapped_extensions", C.c_int64),
                ("good_extensions", C.c_int64), ("seqs_passed", C.c_int64)]


class OrcKarlin(C.Structure):
    _fields_ = [("Lambda", C.c_double), ("K", C.c_double), ("logK", C.c_double), ("H", C.c_double)]


class OrcEditScript(C.Structure):
    _fields_ = [("op", C.POINTER(C.c_uint8)), ("num", C.POINTER(C.c_int32)), ("size", C.c_int32)]


class OrcTbHSP(C.Structure):
    _fields_ = [("hsp", OrcHSP), ("esp", OrcEditScript), ("num_ident", C.c_int32), ("align_length", C.c_int32),
                ("gaps", C.c_int32), ("gap_opens", C.c_int32), ("bit_score", C.c_double)]


class OrcGapOut(C.Structure):
    _fields_ = [("q_start", C.c_int32), ("q_stop", C.c_int32), ("s_start", C.c_int32), ("s_stop", C.c_int32),
                ("score", C.c_int32), ("seed_q", C.c_int32), ("seed_s", C.c_int32)]


def build(force=False):
    """Compile oracle/liborc.so with the committed Makefile."""
    so = os.path.join(_HERE, "liborc.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(f) > os.path.getmtime(so) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_search_new_masked.restype = C.c_void_p
        L.orc_search_new_masked.argtypes = [C.POINTER(OrcOptions), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int32),
                                            C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.orc_search_new.restype = C.c_void_p
        L.orc_search_new.argtypes = [C.POINTER(OrcOptions), C.c_int, C.POINTER(C.c_void_p),
                                     C.POINTER(C.c_int32)]
        L.orc_search_free.argtypes = [C.c_void_p]
        L.orc_default_options.argtypes = [C.POINTER(OrcOptions), C.c_int]
        L.orc_search_subject.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(OrcStats)]
        for name, rt in [("orc_num_contexts", C.c_int32), ("orc_lut_type", C.c_int32),
                         ("orc_lut_width", C.c_int32), ("orc_scan_step", C.c_int32),
                         ("orc_diag_container", C.c_int32), ("orc_gap_x_dropoff", C.c_int32),
                         ("orc_gap_x_dropoff_final", C.c_int32), ("orc_gap_lambda", C.c_double),
                         ("orc_gap_K", C.c_double), ("orc_query_concat_len", C.c_int32),
                         ("orc_num_seeds", C.c_int32), ("orc_num_init_hits", C.c_int32),
                         ("orc_num_hsps", C.c_int32)]:
            getattr(L, name).restype = rt
            getattr(L, name).argtypes = [C.c_void_p]
        L.orc_contexts.restype = C.POINTER(OrcContext); L.orc_contexts.argtypes = [C.c_void_p]
        L.orc_seeds.restype = C.POINTER(OrcSeed); L.orc_seeds.argtypes = [C.c_void_p]
        L.orc_init_hits.restype = C.POINTER(OrcInitHit); L.orc_init_hits.argtypes = [C.c_void_p]
        L.orc_hsps.restype = C.POINTER(OrcHSP); L.orc_hsps.argtypes = [C.c_void_p]
        L.orc_query_concat.restype = C.POINTER(C.c_uint8); L.orc_query_concat.argtypes = [C.c_void_p]
        L.orc_karlin_ungapped.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double),
                                          C.POINTER(C.c_double), C.POINTER(OrcKarlin)]
        L.orc_karlin_ideal.argtypes = [C.c_int, C.c_int, C.POINTER(OrcKarlin)]
        L.orc_karlin_nucl_gapped.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.POINTER(OrcKarlin), C.POINTER(OrcKarlin),
                                             C.POINTER(C.c_int)]
        L.orc_nucl_alpha_beta.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(OrcKarlin),
                                          C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_length_adjustment.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double,
                                            C.c_int32, C.c_int64, C.c_int32, C.POINTER(C.c_int32)]
        L.orc_hsplist_purge_common_endpoints.restype = C.c_int32
        L.orc_hsplist_purge_common_endpoints.argtypes = [C.POINTER(OrcHSP), C.c_int32]
        L.orc_hsplist_sort_by_score.argtypes = [C.POINTER(OrcHSP), C.c_int32]
        L.orc_dust.restype = C.c_int32
        L.orc_dust.argtypes = [C.c_void_p, C.c_int32, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int32]
        L.orc_prelim_hitlist_size.argtypes = [C.c_int, C.c_int]
        L.orc_collector_new.restype = C.c_void_p; L.orc_collector_new.argtypes = [C.c_int32, C.c_int32, C.c_int]
        L.orc_collector_write.argtypes = [C.c_void_p, C.c_int32, C.POINTER(OrcHSP), C.c_int32]
        L.orc_collector_close.restype = C.c_int64; L.orc_collector_close.argtypes = [C.c_void_p]
        L.orc_collector_list.restype = C.c_int32
        L.orc_collector_list.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                         C.POINTER(C.POINTER(OrcHSP))]
        L.orc_collector_free.argtypes = [C.c_void_p]
        L.orc_greedy_extend.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, C.POINTER(OrcHSP)]
        L.orc_dynprog_extend.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(OrcHSP)]
        L.orc_traceback_hsp_list.restype = C.c_int32
        L.orc_traceback_hsp_list.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(OrcHSP), C.c_int32,
                                             C.POINTER(C.POINTER(OrcTbHSP))]
        L.orc_traceback_free.argtypes = [C.POINTER(OrcTbHSP), C.c_int32]
        for f in (L.orc_tb_dynprog, L.orc_tb_greedy):
            f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                          C.POINTER(OrcGapOut), C.POINTER(OrcEditScript)]
        L.orc_semi_gapped_score.restype = C.c_int32
        L.orc_semi_gapped_score.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32),
                                            C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32, C.c_int]
        L.orc_matrix.restype = C.POINTER(C.c_int32); L.orc_matrix.argtypes = [C.c_void_p]
        L.orc_esp_free.argtypes = [C.POINTER(OrcEditScript)]
        _LIB = L
    return _LIB


# ---------------------------------------------------------------- sequences
_IUPAC = np.full(256, 15, dtype=np.uint8)
for _i, _ch in enumerate("ACGTRYMKWSBDHVN-"):
    _IUPAC[ord(_ch)] = _i
    _IUPAC[ord(_ch.lower())] = _i


def encode_blastna(s):
    """IUPAC string -> BLASTNA codes (CORE/blast_encoding.c:80-96)."""
    return _IUPAC[np.frombuffer(s.encode("ascii"), dtype=np.uint8)].copy()


def read_fasta(path):
    seqs, cur = [], []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line.startswith(">"):
                if cur:
                    seqs.append("".join(cur)); cur = []
            elif line:
                cur.append(line)
    if cur:
        seqs.append("".join(cur))
    return seqs


def pack_ncbi2na(bases, pad=16):
    """Array of 0..3 codes -> NCBI2na (4 bases/byte, base 0 in bits 7..6), plus pad bytes."""
    b = np.asarray(bases, dtype=np.uint8) & 3
    n = len(b)
    full = np.zeros(((n + 3) // 4) * 4, dtype=np.uint8)
    full[:n] = b
    q = full.reshape(-1, 4)
    out = (q[:, 0] << 6) | (q[:, 1] << 4) | (q[:, 2] << 2) | q[:, 3]
    return np.concatenate([out.astype(np.uint8), np.zeros(pad, dtype=np.uint8)])


def unpack_ncbi2na(packed, n):
    p = np.asarray(packed, dtype=np.uint8)[: (n + 3) // 4]
    out = np.empty((len(p), 4), dtype=np.uint8)
    out[:, 0] = p >> 6; out[:, 1] = (p >> 4) & 3; out[:, 2] = (p >> 2) & 3; out[:, 3] = p & 3
    return out.reshape(-1)[:n]


def read_blastdb_v4_nucl(prefix):
    """Minimal BLAST DB v4 nucleotide volume reader (.nin/.nsq), following
    seqdb_reader/index_files.txt:62-120 and sequence_files.txt.  Returns a list of
    (packed_bytes, length) ignoring ambiguity runs."""
    import struct
    nin = open(prefix + ".nin", "rb").read()
    nsq = np.frombuffer(open(prefix + ".nsq", "rb").read(), dtype=np.uint8)
    ver, typ = struct.unpack(">ii", nin[:8])
    assert ver == 4 and typ == 0
    off = 8
    (tl,) = struct.unpack(">i", nin[off:off + 4]); off += 4 + tl
    (dl,) = struct.unpack(">i", nin[off:off + 4]); off += 4 + dl
    (noids,) = struct.unpack(">i", nin[off:off + 4]); off += 4
    (vol_len,) = struct.unpack("<q", nin[off:off + 8]); off += 8
    (max_len,) = struct.unpack(">i", nin[off:off + 4]); off += 4
    n1 = noids + 1
    hdr = struct.unpack(">%di" % n1, nin[off:off + 4 * n1]); off += 4 * n1
    seq = struct.unpack(">%di" % n1, nin[off:off + 4 * n1]); off += 4 * n1
    amb = struct.unpack(">%di" % n1, nin[off:off + 4 * n1]); off += 4 * n1
    out = []
    for i in range(noids):
        data = nsq[seq[i]:amb[i]]
        whole = len(data) - 1
        rem = int(data[-1]) & 3         # last byte: number of valid bases in it (low 2 bits)
        length = whole * 4 + rem
        packed = np.concatenate([data, np.zeros(16, dtype=np.uint8)]).copy()
        # zero the count bits of the final byte so it is pure sequence data
        packed[whole] &= 0xFC
        out.append((packed, length))
    assert sum(l for _, l in out) == vol_len
    return out


# ---------------------------------------------------------------- search wrapper
def default_options(megablast=True, db_length=0, db_num_seqs=0, **kw):
    o = OrcOptions()
    lib().orc_default_options(C.byref(o), 1 if megablast else 0)
    o.db_length = db_length
    o.db_num_seqs = db_num_seqs
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class Search:
    def __init__(self, opt, queries, masks=None):
        """queries: list of uint8 BLASTNA arrays (plus strand); masks: [(query index, from, to)],
        inclusive plus-strand intervals, sorted and non-overlapping per query (soft masking)."""
        self._L = lib()
        self._q = [np.ascontiguousarray(q, dtype=np.uint8) for q in queries]
        ptrs = (C.c_void_p * len(queries))(*[q.ctypes.data for q in self._q])
        lens = (C.c_int32 * len(queries))(*[len(q) for q in self._q])
        self.opt = opt
        masks = sorted(masks or [])
        n = len(masks)
        mq = (C.c_int32 * max(n, 1))(*[m[0] for m in masks])
        mf = (C.c_int32 * max(n, 1))(*[m[1] for m in masks])
        mt = (C.c_int32 * max(n, 1))(*[m[2] for m in masks])
        self._h = self._L.orc_search_new_masked(C.byref(opt), len(queries), ptrs, lens, n, mq, mf, mt)
        if not self._h:
            raise RuntimeError("orc_search_new failed")
        self.stats = OrcStats()

    def close(self):
        if self._h:
            self._L.orc_search_free(self._h); self._h = None

    def __del__(self):
        self.close()

    @property
    def contexts(self):
        n = self._L.orc_num_contexts(self._h)
        p = self._L.orc_contexts(self._h)
        return [p[i] for i in range(n)]

    def info(self):
        L, h = self._L, self._h
        return dict(lut_type=L.orc_lut_type(h), lut_width=L.orc_lut_width(h),
                    scan_step=L.orc_scan_step(h), container=L.orc_diag_container(h),
                    gap_x_dropoff=L.orc_gap_x_dropoff(h),
                    gap_x_dropoff_final=L.orc_gap_x_dropoff_final(h),
                    gap_lambda=L.orc_gap_lambda(h), gap_K=L.orc_gap_K(h),
                    qlen=L.orc_query_concat_len(h))

    def query_concat(self):
        n = self._L.orc_query_concat_len(self._h)
        p = self._L.orc_query_concat(self._h)
        return np.ctypeslib.as_array(p, shape=(n,)).copy()

    def subject_chunked(self, packed, length, max_len):
        """a subject longer than the engine's MAX_DBSEQ_LEN: searched in chunks, lists merged -> hsps only"""
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        self._L.orc_search_subject_chunked.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(OrcStats)]
        rc = self._L.orc_search_subject_chunked(self._h, packed.ctypes.data, length, max_len, C.byref(self.stats))
        assert rc == 0
        n = self._L.orc_num_hsps(self._h)
        if n == 0:
            return np.zeros(0, dtype=HSP_DT)
        return np.frombuffer(C.string_at(self._L.orc_hsps(self._h), n * C.sizeof(OrcHSP)), dtype=HSP_DT).copy()

    def carry_diag(self, on=True):
        """the diagonal container carried across subjects (reference behaviour) instead of fresh per subject"""
        self._L.orc_search_carry_diag.argtypes = [C.c_void_p, C.c_int]
        self._L.orc_search_carry_diag(self._h, 1 if on else 0)

    def subject(self, packed, length):
        """Run one subject; returns dict(seeds, init_hits, hsps) as numpy structured arrays."""
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        assert len(packed) >= (length + 3) // 4 + 4
        self._L.orc_search_subject(self._h, packed.ctypes.data, length, C.byref(self.stats))
        L, h = self._L, self._h

        def grab(nf, pf, ctype, dt):
            n = nf(h)
            if n == 0:
                return np.zeros(0, dtype=dt)
            buf = C.string_at(pf(h), n * C.sizeof(ctype))
            return np.frombuffer(buf, dtype=dt).copy()
        seeds = grab(L.orc_num_seeds, L.orc_seeds, OrcSeed, SEED_DT)
        ih = grab(L.orc_num_init_hits, L.orc_init_hits, OrcInitHit, IHIT_DT)
        hs = grab(L.orc_num_hsps, L.orc_hsps, OrcHSP, HSP_DT)
        return dict(seeds=seeds, init_hits=ih, hsps=hs)


    # ---- traceback stage (orc_traceback.c) ----
    def matrix(self):
        return np.ctypeslib.as_array(self._L.orc_matrix(self._h), shape=(16, 16)).copy()

    def traceback(self, subject_blastna, hsps):
        """Blast_TracebackFromHSPList for the preliminary HSPs (records of HSP_DT, sorted by score) of ONE query
        against one subject given as BLASTNA codes.  -> list of dicts (HSP fields + ops, num_ident, ...)."""
        sub = np.ascontiguousarray(subject_blastna, dtype=np.uint8)
        arr = (OrcHSP * max(len(hsps), 1))()
        for i, h in enumerate(hsps):
            for f in Collector.FIELDS:
                setattr(arr[i], f, h[f].item() if hasattr(h[f], "item") else h[f])
        out = C.POINTER(OrcTbHSP)()
        n = self._L.orc_traceback_hsp_list(self._h, sub.ctypes.data, len(sub), arr, len(hsps), C.byref(out))
        res = []
        for i in range(n):
            t = out[i]
            d = {f: getattr(t.hsp, f) for f in Collector.FIELDS}
            d.update(num_ident=t.num_ident, align_length=t.align_length, gaps=t.gaps, gap_opens=t.gap_opens,
                     bit_score=t.bit_score, ops=[(int(t.esp.op[k]), int(t.esp.num[k])) for k in range(t.esp.size)])
            res.append(d)
        self._L.orc_traceback_free(out, n)
        return res

    def align_traceback(self, context, subject_blastna, q_start, s_start, x_dropoff, greedy=False):
        """One gapped extension with traceback from (q_start, s_start) (context-relative query offset)."""
        sub = np.ascontiguousarray(subject_blastna, dtype=np.uint8)
        r, e = OrcGapOut(), OrcEditScript()
        f = self._L.orc_tb_greedy if greedy else self._L.orc_tb_dynprog
        f(self._h, context, sub.ctypes.data, len(sub), q_start, s_start, x_dropoff, C.byref(r), C.byref(e))
        ops = [(int(e.op[k]), int(e.num[k])) for k in range(e.size)]
        self._L.orc_esp_free(C.byref(e))
        return dict(q_start=r.q_start, q_stop=r.q_stop, s_start=r.s_start, s_stop=r.s_stop, score=r.score, ops=ops)


def gapped_extend(query, subj_bases, q_off, s_off, xdrop, reward, penalty, gap_open, gap_extend, greedy=False):
    """one score-only gapped extension (the preliminary stage's aligners) -> dict of the HSP fields"""
    q = np.ascontiguousarray(query, dtype=np.uint8)
    packed = pack_ncbi2na(subj_bases)
    h = OrcHSP()
    f = lib().orc_greedy_extend if greedy else lib().orc_dynprog_extend
    rc = f(q.ctypes.data, len(q), packed.ctypes.data, len(subj_bases), q_off, s_off, xdrop, reward, penalty, gap_open, gap_extend, C.byref(h))
    assert rc == 0
    return dict(q_offset=h.q_offset, q_end=h.q_end, s_offset=h.s_offset, s_end=h.s_end, score=h.score)


def semi_gapped_score(matrix, A, B, M, N, x_dropoff, gap_open, gap_extend, reverse):
    """Blast_SemiGappedAlign, score only -> (score, a_offset, b_offset)"""
    m = np.ascontiguousarray(matrix, dtype=np.int32)
    a = np.ascontiguousarray(A, dtype=np.uint8); b = np.ascontiguousarray(B, dtype=np.uint8)
    ao, bo = C.c_int32(), C.c_int32()
    sc = lib().orc_semi_gapped_score(m.ctypes.data, a.ctypes.data, b.ctypes.data, M, N, C.byref(ao), C.byref(bo),
                                     x_dropoff, gap_open, gap_extend, 1 if reverse else 0)
    return sc, ao.value, bo.value


SEED_DT = np.dtype([("q_off", "<i4"), ("s_off", "<i4")])
IHIT_DT = np.dtype([("q_off", "<i4"), ("s_off", "<i4"), ("q_start", "<i4"), ("s_start", "<i4"),
                    ("length", "<i4"), ("score", "<i4")])
HSP_DT = np.dtype([("context", "<i4"), ("q_offset", "<i4"), ("q_end", "<i4"),
                   ("q_gapped_start", "<i4"), ("s_offset", "<i4"), ("s_end", "<i4"),
                   ("s_gapped_start", "<i4"), ("score", "<i4"), ("evalue", "<f8")])


class Collector:
    """orc_collect.c: the reference's collector writer, fed one subject's HSP list at a time."""
    FIELDS = ["context", "q_offset", "q_end", "q_gapped_start", "s_offset", "s_end", "s_gapped_start", "score", "evalue"]

    def __init__(self, num_queries, hitlist_size=500, gapped=True):
        self._c = lib().orc_collector_new(num_queries, hitlist_size, 1 if gapped else 0)

    def write(self, oid, hsps):
        """hsps: sequence of dicts / records with FIELDS"""
        arr = (OrcHSP * len(hsps))()
        for i, h in enumerate(hsps):
            for f in self.FIELDS:
                setattr(arr[i], f, h[f].item() if hasattr(h[f], "item") else h[f])
        return lib().orc_collector_write(self._c, oid, arr, len(hsps))

    def close(self):
        """-> list of (oid, query, [tuple(FIELDS)...]) in (oid, query) ascending order"""
        L = lib()
        n = L.orc_collector_close(self._c)
        out = []
        for i in range(n):
            oid, q, p = C.c_int32(), C.c_int32(), C.POINTER(OrcHSP)()
            k = L.orc_collector_list(self._c, i, C.byref(oid), C.byref(q), C.byref(p))
            out.append((oid.value, q.value, [tuple(getattr(p[j], f) for f in self.FIELDS) for j in range(k)]))
        return out

    def __del__(self):
        try:
            if self._c:
                lib().orc_collector_free(self._c); self._c = None
        except Exception:
            pass


def dust(seq, level=20, window=64, linker=1):
    """Symmetric DUST intervals [(from, to)] of a BLASTNA sequence (orc_dust.c)."""
    a = np.ascontiguousarray(seq, dtype=np.uint8)
    cap = len(a) // 2 + 4
    f = np.zeros(cap, dtype=np.int32); t = np.zeros(cap, dtype=np.int32)
    n = lib().orc_dust(a.ctypes.data, len(a), level, window, linker, f.ctypes.data, t.ctypes.data, cap)
    return [(int(f[i]), int(t[i])) for i in range(n)]
