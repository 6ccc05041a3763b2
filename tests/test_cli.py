"""blastn_prelim: the reference's blastn command line on this library's stages (C++ host classes over the C ABI).
CPU: it is built and refuses what it cannot do.  GPU: FASTA queries against the reference's own `seqn` test
database -- the twelve standard tabular columns of the final alignments equal the rows the ORACLE produces
(its DUST, preliminary search, collector and traceback), in all three -mode settings; `-stage prelim` gives
exactly the rows the library calls give (a consistency property, not oracle parity)."""
import os
import subprocess
import sys
import numpy as np
import pytest
from gblastn_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "gblastn_amd", "bin", "blastn_prelim")
DB = os.path.join(ROOT, "tests", "golden", "seqn")
IUPAC = "ACGTRYMKWSBDHVN-"


def test_cli_is_built_and_refuses_a_cpu_run(tmp_path):
    assert os.path.exists(CLI), "run __graft_entry__.build() (make -C gblastn_amd/csrc)"
    p = subprocess.run([CLI, "-help"], capture_output=True, text=True)
    assert p.returncode == 0 and "-use_gpu true" in p.stdout
    q = tmp_path / "q.fa"; q.write_text(">q\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n")
    p = subprocess.run([CLI, "-db", DB, "-query", str(q), "-use_gpu", "false"], capture_output=True, text=True)
    assert p.returncode != 0 and "no CPU path" in p.stderr


def _queries(db):
    """a few subjects of the database as queries: one verbatim, one with substitutions and an indel,
    one reverse-complemented, one with a low-complexity insert (DUST has something to mask)"""
    rng = np.random.default_rng(11)
    out = []
    a = db.blastna(1500).copy(); out.append(("exact_1500", a))
    b = db.blastna(10).copy()
    pos = rng.choice(len(b), 12, replace=False); b[pos] = (b[pos] + 1 + rng.integers(0, 3, 12)) % 4
    b = np.concatenate([b[:300], b[303:]]); out.append(("mutated_10", b))
    c = db.blastna(777).copy(); c = (3 - np.minimum(c, 3))[::-1].copy(); out.append(("revcomp_777", c.astype(np.uint8)))
    d = db.blastna(42).copy(); d = np.concatenate([d[:200], np.zeros(60, np.uint8), d[200:]]); out.append(("polyA_42", d))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("task,mode,batch", [("megablast", "1", None), ("blastn", "1", None), ("megablast", "2", None),
                                             ("megablast", "2", "900"), ("blastn", "2", "900")])
def test_cli_rows_equal_the_library_calls(tmp_path, task, mode, batch):
    """batch: BATCH_SIZE override (the reference's experimentation knob) -- one query per batch, so that
    -mode 2 really pipelines: batch k+1 is set up on a second thread and its lookup tables are built on the
    device while batch k is searched"""
    db = api.BlastDb(DB)
    qs = _queries(db)
    fa = tmp_path / "q.fa"
    fa.write_text("".join(">%s some description\n%s\n" % (n, "".join(IUPAC[int(x)] for x in s)) for n, s in qs))
    out = tmp_path / "out.tsv"
    env = dict(os.environ)
    if batch:
        env["BATCH_SIZE"] = batch
    p = subprocess.run([CLI, "-db", DB, "-query", str(fa), "-task", task, "-use_gpu", "true", "-mode", mode, "-stage", "prelim",
                        "-evalue", "1e-3", "-max_target_seqs", "5", "-out", str(out)], capture_output=True, text=True, timeout=600, env=env)
    if batch:
        assert " in 1 batches" not in p.stderr and "batches" in p.stderr, p.stderr
    assert p.returncode == 0, p.stderr[-2000:]
    rows = [l.split("\t") for l in out.read_text().splitlines()]
    assert rows, p.stderr

    # the same search through the library: DUST masks, database statistics, top-N collector
    seqs = [s for _, s in qs]
    opt = api.default_options(task, db_length=db.total_length, db_num_seqs=db.num_seqs, evalue=1e-3, hitlist_size=5)
    ps = api.BlastPrelimSearch(seqs, opt, db.load_shard(), masks=api.dust_masks(seqs))
    col = api.BlastHSPCollector(len(seqs), 5)
    col.write(ps.run()["hsps"])
    hsps, starts, lq = col.close()
    want = set()
    for l in range(len(lq)):
        for h in hsps[starts[l]:starts[l + 1]]:
            qlen = len(seqs[lq[l]]); minus = int(h["context"]) & 1
            qs_, qe_ = (qlen - h["q_end"] + 1, qlen - h["q_offset"]) if minus else (h["q_offset"] + 1, h["q_end"])
            ss_, se_ = (h["s_end"], h["s_offset"] + 1) if minus else (h["s_offset"] + 1, h["s_end"])
            want.add((qs[lq[l]][0], int(h["oid"]), int(qs_), int(qe_), int(ss_), int(se_), int(h["score"]), "minus" if minus else "plus"))
    got = set((r[0], int(r[1]), int(r[2]), int(r[3]), int(r[4]), int(r[5]), int(r[8]), r[9]) for r in rows)
    assert got == want
    # every query finds the subject it was made from, on the strand it was made from
    best = {}
    for r in rows:
        best.setdefault(r[0], r)
    assert int(best["exact_1500"][1]) == 1500 and best["exact_1500"][9] == "plus"
    assert int(best["mutated_10"][1]) == 10 and int(best["revcomp_777"][1]) == 777 and best["revcomp_777"][9] == "minus"
    assert int(best["polyA_42"][1]) == 42
    # bit scores follow from the raw scores (gapped Karlin-Altschul parameters of the task)
    import ctypes as C, math
    lam, K = C.c_double(), C.c_double()
    f = api.lib().gbn_batch_karlin_gapped
    f.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    assert f(ps._b, C.byref(lam), C.byref(K)) == 0 and lam.value > 0 and K.value > 0
    for r in rows:
        assert abs(float(r[7]) - (lam.value * int(r[8]) - math.log(K.value)) / math.log(2.0)) < 0.051


def _evalue_string(e):
    """objtools/align_format/align_format_util.cpp:694-713"""
    if e < 1.0e-180: return "0.0"
    if e < 1.0e-99: return "%2.0e" % e
    if e < 0.0009: return "%3.0e" % e
    if e < 0.1: return "%4.3f" % e
    if e < 1.0: return "%3.2f" % e
    if e < 10.0: return "%2.1f" % e
    return "%5.0f" % e


def _bits_string(s):
    if s > 9999: return "%4.3e" % s
    if s > 99.9: return "%4d" % int(s)
    return "%4.1f" % s


def _oracle_rows(db, qs, task, evalue, hitlist):
    """the whole search on the CPU oracle: DUST masks, every subject, collector, traceback, result order"""
    from oracle import orc
    from tests import util
    seqs = [s for _, s in qs]
    masks = []
    for i, s in enumerate(seqs):
        masks += [(i, f, t) for f, t in orc.dust(s)]
    gopt = api.default_options(task, db_length=db.total_length, db_num_seqs=db.num_seqs, evalue=evalue, hitlist_size=hitlist)
    S = orc.Search(util.oracle_options(gopt), seqs, masks=masks)
    col = orc.Collector(len(seqs), hitlist)
    subj = {}
    for oid in range(db.num_seqs):
        packed, n = db.ncbi2na(oid)
        r = S.subject(np.concatenate([packed, np.zeros(16, np.uint8)]), n)
        if len(r["hsps"]):
            col.write(oid, [dict(zip(r["hsps"].dtype.names, x)) for x in r["hsps"]])
            subj[oid] = db.blastna(oid)                 # with its ambiguity codes, as the traceback stage sees it
    per_query = {}
    for oid, q, hs in col.close():
        fin = S.traceback(subj[oid], [dict(zip(orc.Collector.FIELDS, h)) for h in hs])
        if fin:
            per_query.setdefault(q, []).append((oid, fin))
    rows = []
    ctxs = S.contexts
    for q in sorted(per_query):
        lists = per_query[q]
        # CORE/blast_hits.c:2757-2788: best e-value (exact ties broken by score, then higher oid first)
        lists.sort(key=lambda l: (min(f["evalue"] for f in l[1]), -l[1][0]["score"], -l[0]))
        for oid, fin in lists[:hitlist]:
            for f in fin:
                minus = f["context"] & 1
                qlen = ctxs[f["context"]].query_length
                qs_, qe_ = (qlen - f["q_end"] + 1, qlen - f["q_offset"]) if minus else (f["q_offset"] + 1, f["q_end"])
                ss_, se_ = (f["s_end"], f["s_offset"] + 1) if minus else (f["s_offset"] + 1, f["s_end"])
                al = f["align_length"]
                rows.append([qs[q][0], "gnl|BL_ORD_ID|%d" % oid, "%.2f" % (100.0 * f["num_ident"] / al), str(al),
                             str(al - f["num_ident"] - f["gaps"]), str(f["gap_opens"]), str(qs_), str(qe_), str(ss_), str(se_),
                             _evalue_string(f["evalue"]), _bits_string(f["bit_score"])])
    return rows


@pytest.mark.gpu
@pytest.mark.parametrize("task,mode,batch", [("megablast", "1", None), ("blastn", "1", None), ("megablast", "2", "900"),
                                             ("blastn", "0", "900")])
def test_cli_final_rows_equal_the_oracle(tmp_path, task, mode, batch):
    db = api.BlastDb(DB)
    qs = _queries(db)
    fa = tmp_path / "q.fa"
    fa.write_text("".join(">%s some description\n%s\n" % (n, "".join(IUPAC[int(x)] for x in s)) for n, s in qs))
    out = tmp_path / "out.tsv"
    env = dict(os.environ)
    if batch:
        env["BATCH_SIZE"] = batch
    p = subprocess.run([CLI, "-db", DB, "-query", str(fa), "-task", task, "-use_gpu", "true", "-mode", mode,
                        "-evalue", "1e-3", "-max_target_seqs", "5", "-out", str(out), "-trace_t_num", "3"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    got = [l.split("\t") for l in out.read_text().splitlines()]
    want = _oracle_rows(db, qs, task, 1e-3, 5)
    assert got == want
    assert len(got) >= 4 and all(len(r) == 12 for r in got)
    first = {}
    for r in got:
        first.setdefault(r[0], r)
    # (subject 1500 carries ambiguity codes: the shard has 2 bits per base, the traceback stage puts the codes back)
    assert len(db.ambiguities(1500)[0]) > 0
    assert first["exact_1500"][1] == "gnl|BL_ORD_ID|1500" and first["exact_1500"][2] == "100.00" and first["exact_1500"][5] == "0"
    assert first["mutated_10"][1] == "gnl|BL_ORD_ID|10" and int(first["mutated_10"][5]) >= 1 and int(first["mutated_10"][4]) >= 8
    assert first["revcomp_777"][1] == "gnl|BL_ORD_ID|777" and int(first["revcomp_777"][8]) > int(first["revcomp_777"][9])


@pytest.mark.gpu
@pytest.mark.parametrize("task,stage,mode,batch", [("megablast", "traceback", "1", None), ("blastn", "traceback", "2", "900"),
                                                   ("megablast", "prelim", "2", "900"), ("blastn", "prelim", "1", None)])
def test_cli_search_threads_over_database_parts_give_the_rows_of_one(tmp_path, task, stage, mode, batch):
    """-gpu_id -1 / -num_threads N: the volumes of the alias dealt to N search threads (one pipeline and one resident
    shard each, on the node's GPUs round robin -- both on the one GPU of this box), per-part results merged per query in
    the process (gbn_traceback_merge / one collector): the rows of one thread over the whole database."""
    db = api.BlastDb(DB)
    qs = _queries(db)
    fa = tmp_path / "q.fa"
    fa.write_text("".join(">%s some description\n%s\n" % (n, "".join(IUPAC[int(x)] for x in s)) for n, s in qs))
    two = os.path.join(ROOT, "tests", "golden", "two_vols")
    env = dict(os.environ)
    if batch:
        env["BATCH_SIZE"] = batch
    rows = {}
    for n in ("1", "2"):
        out = tmp_path / ("out%s.tsv" % n)
        p = subprocess.run([CLI, "-db", two, "-query", str(fa), "-task", task, "-use_gpu", "true", "-gpu_id", "-1", "-num_threads", n,
                            "-mode", mode, "-stage", stage, "-evalue", "1e-3", "-max_target_seqs", "5", "-out", str(out)],
                           capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        rows[n] = out.read_text().splitlines()
        if stage == "prelim":       # the scaling self-check: a line per part and the run's total (what a node of N GPUs prints its 1 / N table with)
            tab = [l for l in p.stderr.splitlines() if l.startswith("# ")]
            parts = [l.split() for l in tab if l[2].isdigit()]
            assert len(parts) == int(n) and all(int(x[4]) >= 1 and float(x[6]) > 0 for x in parts), p.stderr[-1500:]
            total = [l for l in tab if l.startswith("# total:")]
            assert len(total) == 1 and ("%d parts" % int(n)) in total[0] and "Gbp/s" in total[0]
    assert len(rows["1"]) >= 4 and rows["1"] == rows["2"]


def _mixer_sizes(hits_per_batch, asked_first=10000, target=2000000, most=4999000):
    """CBatchSizeMixer::GetBatchSize transcribed (APP/blast_app_util.cpp:67-86, .hpp:54-73): the sizes asked for, batch by batch,
    given the hits (good_init_extends) every batch came back with; batches run one at a time"""
    ratio, size, out = -1.0, asked_first, [asked_first]
    for hits in hits_per_batch:
        if hits > 0:
            r = 1.0 * hits / size
            ratio = r if ratio < 0 else 0.3 * r + 0.7 * ratio
            want = 1.0 * target / ratio
            size = int(want) if want < 2147483648.0 else -2147483648     # (Int4)double beyond Int4 on x86-64: INT_MIN, i.e. the 100-base floor
            if size > most:
                size, ratio = most, -1.0
            elif size < 100:
                size, ratio = 100, -1.0
        elif hits == 0:
            size, ratio = most, -1.0
        out.append(size)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["1", "2"])
def test_cli_batch_plan_of_the_reference_and_unchanged_rows(tmp_path, mode):
    """blastn starts with a 10,000-base sample and steers towards 2 M extensions per batch (CBatchSizeMixer; APP/blastn_app.cpp:360-386);
    a batch takes whole queries until it has the bases asked for (CBlastInput::GetNextSeqBatch).  46 queries of ~1 kb against the
    unit-test database with target / cap scaled down by the test's knobs so that three and more batches form: the sizes the program asked
    for equal the transcription fed with the hits the program reports, and the rows equal those of one fixed batch."""
    db = api.BlastDb(DB)
    rng = np.random.default_rng(5)
    qs = []
    for k in range(46):
        s = db.blastna(int(rng.integers(0, 2000))).copy()
        s = s[:min(len(s), 1000)]
        qs.append(("q%02d" % k, np.minimum(s, 3).astype(np.uint8)))
    fa = tmp_path / "q.fa"
    fa.write_text("".join(">%s\n%s\n" % (n, "".join(IUPAC[int(x)] for x in s)) for n, s in qs))
    rows = {}
    env = dict(os.environ); env.pop("BATCH_SIZE", None)
    env["GBN_CLI_MIXER_TARGET"] = "40"; env["GBN_CLI_MIXER_MOST"] = "12000"      # (tests: the mixer's two constants)
    for plan in ("mixer", "fixed"):
        out = tmp_path / (plan + ".tsv")
        p = subprocess.run([CLI, "-db", DB, "-query", str(fa), "-use_gpu", "true", "-gpu_id", "0", "-mode", mode, "-batch_plan", plan, "-evalue", "1e-3",
                            "-max_target_seqs", "5", "-out", str(out), "-timing", "true"], capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        rows[plan] = out.read_text().splitlines()
        line = [l for l in p.stderr.splitlines() if l.startswith("blastn_prelim: batch plan " + plan)]
        assert len(line) == 1, p.stderr[-1500:]
        plan_items = [x.split("/") for x in line[0].split(": ")[-1].split()]
        asked = [int(x[0]) for x in plan_items]; taken = [int(x[1]) for x in plan_items]
        hits = [int(x) for x in [l for l in p.stderr.splitlines() if l.startswith("blastn_prelim: hits per batch")][0].split(":")[-1].split()]
        if plan == "fixed":
            assert asked == [5000000] and taken == [46]
        else:
            assert len(asked) >= 3 and asked[0] == 10000 and sum(taken) == 46, (asked, taken, hits)
            if mode == "1":     # one batch at a time: exactly the reference's sequence
                assert asked == _mixer_sizes(hits, target=40, most=12000)[:len(asked)], (asked, hits)
            lens = [len(s) for _, s in qs]; i = 0
            for a_, t_ in zip(asked, taken):        # whole queries until the bases asked for are there
                acc = n = 0
                while i + n < len(lens) and acc < a_:
                    acc += lens[i + n]; n += 1
                assert n == t_; i += n
    assert len(rows["mixer"]) >= 40 and rows["mixer"] == rows["fixed"]


@pytest.mark.gpu
@pytest.mark.parametrize("stage", ["traceback", "prelim"])
def test_cli_eight_search_threads_on_one_device_give_the_rows_of_one(tmp_path, stage):
    """The C5 width inside one process: -gpu_id -1 -num_threads 8 deals the database's volumes (ten here, written by
    tools/make_synth_blastdb.py) to EIGHT search threads -- a pipeline and a resident shard each, all on the one device of this box, as
    eight GPUs of a node would each hold theirs (GB/gpu_blast_multi_gpu_utils.cpp:105-139) -- and merges their results per query:
    the rows of one thread over the whole database, the batch plan included (the parts' hits are summed for the mixer)."""
    d = tmp_path / "db"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synth_blastdb.py"), str(d), "--subjects", "100", "--subject-len", "200000",
                        "--volumes", "10", "--queries", "60", "--planted-fraction", "0.7"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-1000:]
    rows = {}
    for n in ("1", "8"):
        out = tmp_path / ("out%s.tsv" % n)
        p = subprocess.run([CLI, "-db", str(d / "c2db"), "-query", str(d / "queries.fa"), "-use_gpu", "true", "-gpu_id", "-1", "-num_threads", n,
                            "-mode", "2", "-stage", stage, "-out", str(out)], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        rows[n] = out.read_text().splitlines()
        if stage == "prelim":
            parts = [l.split() for l in p.stderr.splitlines() if l.startswith("# ") and l[2].isdigit()]
            assert len(parts) == int(n), p.stderr[-1500:]
    assert len(rows["1"]) >= 30 and rows["1"] == rows["8"]
