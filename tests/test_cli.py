"""blastn_prelim: the reference's blastn command line for the preliminary stage (C++ over the C ABI).
CPU: it is built and refuses what it cannot do.  GPU: FASTA queries against the reference's own `seqn`
test database give exactly the rows the library calls give (same masks, same statistics, same collector)."""
import os
import subprocess
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
    p = subprocess.run([CLI, "-db", DB, "-query", str(fa), "-task", task, "-use_gpu", "true", "-mode", mode,
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
