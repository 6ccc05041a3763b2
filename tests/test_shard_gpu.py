"""GPU: the database sharded by volume over two ranks that share the one device of the test box, exchanging over RCCL
("nccl" backend; the launcher makes each rank its own RCCL host, see gblastn_amd/blastn_sharded.py) -- the rows of
the sharded search equal the rows of the one-rank search and of the C++ command line on the whole database, batch
order kept while four batches are in flight.  Also the collectives of gather_records on a one-rank nccl group."""
import os
import subprocess
import sys
import numpy as np
import pytest
from gblastn_amd import api

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
CLI = os.path.join(ROOT, "gblastn_amd", "bin", "blastn_prelim")
IUPAC = "ACGTRYMKWSBDHVN-"


def make_case(tmp_path):
    (tmp_path / "three.nal").write_text("TITLE three volumes\nDBLIST %s/seqn %s/nt.41646578 %s/seqn\n" % (G, G, G))
    db = api.BlastDb(str(tmp_path / "three"))
    assert db.num_volumes == 3 and db.num_seqs == 4009
    rng = np.random.default_rng(2)
    with open(tmp_path / "q.fa", "w") as f:
        for oid in (5, 700, 1500, 2004, 2500, 3999, 42, 1234):
            s = db.blastna(oid)[:900].copy()
            pos = rng.choice(len(s), 6, replace=False); s[pos] = (s[pos] + 1) % 4
            f.write(">q%d\n%s\n" % (oid, "".join(IUPAC[int(x)] for x in s)))
    return str(tmp_path / "three"), str(tmp_path / "q.fa")


def launch(nproc, db, fa, out, task, backend="nccl", port=29711, batch=None, extra_env=None, expect_rc=0):
    env = dict(os.environ); env["MASTER_ADDR"] = "127.0.0.1"; env["PYTHONPATH"] = ROOT
    if batch:
        env["BATCH_SIZE"] = str(batch)
    env.update(extra_env or {})
    cmd = [sys.executable]
    if nproc > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % nproc, "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    cmd += ["-m", "gblastn_amd.blastn_sharded", "-db", db, "-query", fa, "-out", out, "-task", task, "-evalue", "1e-3",
            "-max_target_seqs", "5", "-backend", backend]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    if expect_rc:
        assert p.returncode != 0, p.stderr[-3000:]
        return None, p.stderr
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    return open(out).read().splitlines(), p.stderr


@pytest.mark.parametrize("task", ["megablast", "blastn"])
def test_two_ranks_over_rccl_equal_one_rank_and_the_cli(tmp_path, task):
    db, fa = make_case(tmp_path)
    one, _ = launch(1, db, fa, str(tmp_path / "one.tsv"), task)
    # BATCH_SIZE 1000: a query per batch -> the batches queue up behind one another on the Exchange thread
    two, err = launch(2, db, fa, str(tmp_path / "two.tsv"), task, batch=1000)
    import re
    m = re.search(r"in (\d+) batches on 2 ranks \(3 volumes, 0\.\.2 here\)", err)
    assert m and int(m.group(1)) >= 4, err[-300:]
    assert len(one) >= 16 and two == one
    p = subprocess.run([CLI, "-db", db, "-query", fa, "-task", task, "-use_gpu", "true", "-evalue", "1e-3", "-max_target_seqs", "5",
                        "-out", str(tmp_path / "cli.tsv")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert open(tmp_path / "cli.tsv").read().splitlines() == one
    # duplicates of a subject in the two seqn volumes tie: the higher oid comes first (CORE/blast_hits.c:2757-2788)
    rows = [r.split("\t") for r in one if r.startswith("q5\t")]
    assert [rows[0][1], rows[1][1]] == ["gnl|BL_ORD_ID|2010", "gnl|BL_ORD_ID|5"]


def test_three_ranks_with_an_idle_one(tmp_path):
    """more ranks than one of them has volumes for is legal: the empty rank takes part in every exchange"""
    db, fa = make_case(tmp_path)
    one, _ = launch(1, db, fa, str(tmp_path / "one.tsv"), "megablast")
    (tmp_path / "two.nal").write_text("TITLE two\nDBLIST %s/seqn %s/nt.41646578\n" % (G, G))
    a, _ = launch(1, str(tmp_path / "two"), fa, str(tmp_path / "a.tsv"), "megablast")
    b, err = launch(3, str(tmp_path / "two"), fa, str(tmp_path / "b.tsv"), "megablast", port=29713, batch=3000)
    assert a == b and len(a) > 0 and "on 3 ranks" in err


def test_collectives_of_a_one_rank_nccl_group():
    code = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
from gblastn_amd import api, shard
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29717", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
rec = np.zeros(1000, dtype=api.HSP_DT); rec["oid"] = np.arange(1000); rec["score"] = 77
ex = shard.Exchange(dev)
futs = [ex.submit(shard.gather_records, rec[:k], 0, dev, None, True) for k in (1000, 0, 17)]
outs = [f.result() for f in futs]
assert [len(o) for o in outs] == [1000, 0, 17] and outs[0].tobytes() == rec.tobytes()
assert ex.stream is not None and ex.stream != torch.cuda.default_stream(dev)
ex.close(); dist.destroy_process_group(); print("NCCL_ONE_RANK_OK")
''' % ROOT
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "NCCL_ONE_RANK_OK" in p.stdout, p.stdout[-1000:] + p.stderr[-3000:]


def test_eight_ranks_on_one_device_uneven_volumes_an_idle_rank_and_a_failing_one(tmp_path):
    """The C5 protocol at its full width without an 8-GPU node: EIGHT ranks over RCCL on the one device of the box.
    (a) ten volumes over eight ranks (2, 2, 1, 1, 1, 1, 1, 1): rows equal the one-rank rows, several batches in flight;
    (b) seven volumes over eight ranks: the last rank holds nothing and still takes part in every exchange;
    (c) rank 5's preliminary search of the second batch fails: every rank ends, with the failure reported, within the timeout
        -- nobody waits in a gather or a broadcast for a rank that dropped out;
    (d) the C++ command line with eight search threads sharing the device (-gpu_id -1 -num_threads 8): the same rows.
    Not a scaling measurement (one device): bench.py --gpus 8 on a real node is the driver's."""
    db3, fa = make_case(tmp_path)
    vols = " ".join(["%s/seqn %s/nt.41646578" % (G, G)] * 5)
    (tmp_path / "ten.nal").write_text("TITLE ten volumes\nDBLIST %s\n" % vols)
    ten = str(tmp_path / "ten")
    one, _ = launch(1, ten, fa, str(tmp_path / "one.tsv"), "megablast")
    eight, err = launch(8, ten, fa, str(tmp_path / "eight.tsv"), "megablast", port=29721, batch=1000)
    assert len(one) >= 16 and eight == one and "on 8 ranks (10 volumes, 0..2 here)" in err, err[-400:]
    vols7 = " ".join(["%s/seqn %s/nt.41646578" % (G, G)] * 3) + " %s/seqn" % G
    (tmp_path / "seven.nal").write_text("TITLE seven volumes\nDBLIST %s\n" % vols7)
    seven = str(tmp_path / "seven")
    a, _ = launch(1, seven, fa, str(tmp_path / "a.tsv"), "megablast")
    b, err = launch(8, seven, fa, str(tmp_path / "b.tsv"), "megablast", port=29723, batch=3000)
    assert a == b and len(a) > 0 and "on 8 ranks" in err
    _, err = launch(8, ten, fa, str(tmp_path / "f.tsv"), "megablast", port=29725, batch=1000, extra_env={"GBN_TEST_FAIL": "5:1"}, expect_rc=3)
    assert "injected failure" in err and "failed on another rank" in err, err[-1500:]
    p = subprocess.run([CLI, "-db", ten, "-query", fa, "-task", "megablast", "-use_gpu", "true", "-gpu_id", "-1", "-num_threads", "8",
                        "-evalue", "1e-3", "-max_target_seqs", "5", "-out", str(tmp_path / "cli8.tsv")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert open(tmp_path / "cli8.tsv").read().splitlines() == one
