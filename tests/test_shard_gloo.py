"""CPU, world_size 2 over gloo: the N > 1 path -- volume sharding and the
variable-length gather of per-shard HSP records to rank 0."""
import os
import sys
import subprocess
import numpy as np
from gblastn_amd import shard

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch.distributed as dist
from gblastn_amd import shard, api
dist.init_process_group("gloo")
r, w = dist.get_rank(), dist.get_world_size()
lo, hi = shard.shard_bounds(13, w, r)                 # 13 volumes over 2 ranks
n = 0 if r == 1 and os.environ.get("EMPTY_RANK1") else (hi - lo) * 3
rec = np.zeros(n, dtype=api.HSP_DT)
rec["oid"] = lo * 1000 + np.arange(n)
rec["score"] = 100 + r
rec["evalue"] = 1e-5 * (r + 1)
out = shard.gather_records(rec, dst=0)
if r == 0:
    lo1, hi1 = shard.shard_bounds(13, w, 1)
    n1 = 0 if os.environ.get("EMPTY_RANK1") else (hi1 - lo1) * 3
    assert out is not None and len(out) == n + n1, (len(out), n, n1)
    assert np.all(np.diff(out["oid"]) > 0)            # rank order == ascending global OID
    assert set(out["score"].tolist()) <= {100, 101}
    print("GATHER_OK", len(out))
else:
    assert out is None
# exchange + merge: rank 0 replays both shards through the per-query top-N collector
def shard_records(rank):
    lo, hi = shard.shard_bounds(13, w, rank)
    k = (hi - lo) * 3
    a = np.zeros(k, dtype=api.HSP_DT)
    a["oid"] = lo * 1000 + np.arange(k); a["context"] = np.arange(k) % 4
    a["score"] = 60 + (np.arange(k) * 7 + rank) % 23; a["q_end"] = 50; a["s_end"] = 50
    a["evalue"] = 10.0 ** (-(a["score"] - 50.0) / 3)
    return a
got = shard.collect_on_root(shard_records(r), num_queries=2, hitlist_size=5, dst=0)
if r == 0:
    col = api.BlastHSPCollector(2, 5)
    col.write(np.concatenate([shard_records(x) for x in range(w)]))
    want = col.close()
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    assert len(got[2]) == 2 * api.lib().gbn_prelim_hitlist_size(5)
    print("COLLECT_OK", len(got[0]))
else:
    assert got is None
# bench.py issues the merges of consecutive passes from one worker thread while the main thread scans
from concurrent.futures import ThreadPoolExecutor
pool = ThreadPoolExecutor(max_workers=1)
futs = [pool.submit(shard.collect_on_root, shard_records(r), 2, 5, 0) for _ in range(3)]
outs = [f.result() for f in futs]
dist.barrier()
if r == 0:
    assert all(all(np.array_equal(a, b) for a, b in zip(o, want)) for o in outs)
    print("THREADED_OK", len(outs))
else:
    assert all(o is None for o in outs)
dist.destroy_process_group()
'''


def run(extra_env=None):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1"})
    if extra_env:
        env.update(extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29611", "-c", WORKER, root]
    # torch.distributed.run has no -c: write the worker to a temp file instead
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(WORKER)
        path = f.name
    cmd = cmd[:-3] + [path, root]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    os.unlink(path)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "GATHER_OK" in p.stdout and "COLLECT_OK" in p.stdout and "THREADED_OK" in p.stdout
    return p.stdout


def test_shard_bounds_cover_all_volumes():
    for n, w in [(13, 2), (100, 8), (7, 8), (0, 4)]:
        got = [shard.shard_bounds(n, w, r) for r in range(w)]
        assert got[0][0] == 0 and got[-1][1] == n
        for (a, b), (c, d) in zip(got, got[1:]):
            assert b == c and a <= b
        sizes = [b - a for a, b in got]
        assert max(sizes) - min(sizes) <= 1


def test_gather_two_ranks_gloo():
    run()


def test_gather_with_an_empty_shard():
    run({"EMPTY_RANK1": "1"})


# ---------------------------------------------------------------------------------------------------------------
# the whole per-batch protocol of ShardedSearch (gather -> collector -> broadcast of the surviving lists ->
# traceback by the owner -> gather -> merge) with record generators in place of the two GPU stages; batches are
# pipelined through the Exchange thread and must come back in submission order, equal to one rank doing it all
PROTOCOL_WORKER = r'''
import os, sys, types
sys.path.insert(0, sys.argv[1])
import numpy as np, torch.distributed as dist
from gblastn_amd import shard, api
world = int(os.environ.get("WORLD_SIZE", "1"))
import datetime, threading
if world > 1:
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=60))
r = dist.get_rank() if world > 1 else 0
NOID = 90
def excepthook(t, v, tb):                                 # a failed rank must not leave the others waiting
    import traceback; traceback.print_exception(t, v, tb); sys.stdout.flush(); sys.stderr.flush(); os._exit(1)
sys.excepthook = excepthook

def make_shard(w, rank):
    lo, hi = shard.shard_bounds(9, w, rank)               # 9 volumes of 10 subjects
    return types.SimpleNamespace(first_oid=lo * 10, num_oids=(hi - lo) * 10, db_length=10**9, db_num_seqs=NOID, src=None)

def stages(sh):
    def search(queries, masks):
        b, nq = queries                                    # (batch number, number of queries): the test's "queries"
        rng_all = np.random.default_rng(100 + b)
        recs = []
        for oid in range(NOID):                            # every rank draws the same stream, keeps its own subjects
            k = int(rng_all.integers(0, 4))
            a = np.zeros(k, dtype=api.HSP_DT)
            a["oid"] = oid; a["context"] = rng_all.integers(0, 2 * nq, k)
            a["score"] = np.sort(rng_all.integers(30, 90, k))[::-1]
            a["q_offset"] = rng_all.integers(0, 50, k); a["q_end"] = a["q_offset"] + 40
            a["s_offset"] = rng_all.integers(0, 500, k); a["s_end"] = a["s_offset"] + 40
            a["evalue"] = 10.0 ** (-(a["score"] - 20.0) / 4)
            if sh.first_oid <= oid < sh.first_oid + sh.num_oids:
                recs.append(a)
        return (b, nq), (np.concatenate(recs) if recs else np.zeros(0, dtype=api.HSP_DT))
    def trace(token, hsps, starts):
        b, nq = token
        per_q = [[] for _ in range(nq)]
        for l in range(len(starts) - 1):
            for h in hsps[starts[l]:starts[l + 1]]:
                assert sh.first_oid <= h["oid"] < sh.first_oid + sh.num_oids      # only the owner traces a subject
                t = np.zeros(1, dtype=api.TB_DT)
                for f in api.HSP_DT.names:
                    t[f] = h[f]
                t["score"] = h["score"] + 1; t["num_ident"] = 39; t["align_length"] = 40; t["bit_score"] = 2.0 * h["score"]
                per_q[int(h["context"]) // 2].append(t)
        qs = np.zeros(nq + 1, dtype="<i8"); out = []
        for q in range(nq):
            out += per_q[q]; qs[q + 1] = qs[q] + len(per_q[q])
        return (np.concatenate(out) if out else np.zeros(0, dtype=api.TB_DT)), qs
    return search, trace

opt = api.default_options("megablast", hitlist_size=6)
batches = [(0, 3), (1, 7), (2, 1), (3, 5)]
sh = make_shard(world, r)
se, tr = stages(sh)
if os.environ.get("FAIL_STAGE"):
    # a local stage fails on ONE rank for ONE batch: every rank's future of that batch raises, the batches around it
    # come through, nobody hangs in a collective (the process group's timeout is 60 s: a hang fails the test)
    which, bad_rank, bad_batch = os.environ["FAIL_STAGE"], world - 1, 1
    def se_f(queries, masks):
        if which == "search" and r == bad_rank and queries[0] == bad_batch: raise RuntimeError("injected search failure")
        return se(queries, masks)
    def tr_f(token, hsps, starts):
        if which == "trace" and r == bad_rank and token[0] == bad_batch: raise MemoryError("injected traceback failure")
        return tr(token, hsps, starts)
    S = shard.ShardedSearch(sh, opt, search=se_f, trace=tr_f)
    futs = [S.submit(bq, num_queries=bq[1]) for bq in batches]
    outcome = []
    for f in futs:
        try: f.result(); outcome.append("ok")
        except api.BlastError as e: outcome.append("failed: " + str(e))
    S._pending = []; S.close()
    assert [o.startswith("failed") for o in outcome] == [False, True, False, False], outcome
    assert ("injected" in outcome[1]) == (r == bad_rank), outcome
    sys.stdout.write("FAILURE_OK %d %d\n" % (world, r)); sys.stdout.flush()       # (one write: the ranks share the pipe)
    dist.barrier(); dist.destroy_process_group(); sys.exit(0)
S = shard.ShardedSearch(sh, opt, search=se, trace=tr)
for bq in batches:
    S.submit(bq, num_queries=bq[1])                       # all four in flight behind one another
res = S.results()
S.close()
if r == 0:
    one = make_shard(1, 0)
    se1, tr1 = stages(one)
    for bq, got in zip(batches, res):
        tok, local = se1(bq, None)
        col = api.BlastHSPCollector(bq[1], opt.hitlist_size); col.write(local); hs, st, lq = col.close(); col.free()
        rec, qs = tr1(tok, hs, np.asarray(st, dtype="<i8"))
        want = shard.merge_final([(rec, qs)], bq[1], opt.hitlist_size)
        assert np.array_equal(got[1], want[1]), (bq, got[1], want[1])
        assert got[0].tobytes() == want[0].tobytes(), bq
        assert len(want[0]) > 0
        # at most hitlist_size subjects per query, best e-value first
        for q in range(bq[1]):
            rows = got[0][got[1][q]:got[1][q + 1]]
            oids = [int(o) for i, o in enumerate(rows["oid"]) if i == 0 or rows["oid"][i - 1] != o]
            assert len(oids) == len(set(oids)) <= opt.hitlist_size
    print("PROTOCOL_OK", world, sum(len(g[0]) for g in res))
else:
    assert all(g is None for g in res)
if world > 1:
    dist.barrier(); dist.destroy_process_group()
'''


def run_protocol(nproc, port, fail_stage=None):
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(PROTOCOL_WORKER)
        path = f.name
    env = dict(os.environ); env["MASTER_ADDR"] = "127.0.0.1"
    if fail_stage:
        env["FAIL_STAGE"] = fail_stage
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % nproc,
           "--master-addr", "127.0.0.1", "--master-port", str(port), path, root]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    os.unlink(path)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    if fail_stage:
        assert p.stdout.count("FAILURE_OK") == nproc, p.stdout[-2000:]
    else:
        assert "PROTOCOL_OK %d" % nproc in p.stdout


def test_a_failing_local_stage_is_reported_on_every_rank_and_nobody_hangs():
    run_protocol(2, 29625, fail_stage="trace")
    run_protocol(3, 29627, fail_stage="search")


def test_sharded_search_protocol_two_ranks_gloo():
    run_protocol(2, 29621)


def test_sharded_search_protocol_more_ranks_than_some_volumes_gloo():
    run_protocol(4, 29623)          # 9 volumes over 4 ranks: 3 + 2 + 2 + 2


def test_volume_shards_cover_the_database():
    from gblastn_amd import api
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "two_vols")
    parts = [shard.VolumeShard(G, 2, r, load=False) for r in range(2)]
    assert [(p.first_oid, p.num_oids) for p in parts] == [(0, 2004), (2004, 1)]
    assert all((p.db_length, p.db_num_seqs) == (947242, 2005) for p in parts)      # NSEQ / LENGTH of the alias: global
    three = [shard.VolumeShard(G, 3, r, load=False) for r in range(3)]
    assert [p.num_oids for p in three] == [2004, 1, 0] and three[2].src is None
    assert parts[0].owns(2003) and not parts[0].owns(2004) and parts[1].owns(2004)


def test_sharded_search_protocol_eight_ranks_gloo():
    run_protocol(8, 29629)          # 9 volumes over 8 ranks: 2 + 1 + 1 + 1 + 1 + 1 + 1 + 1 (the C5 width)
    run_protocol(8, 29631, fail_stage="search")


SUBGROUP_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, datetime, torch.distributed as dist
from gblastn_amd import shard, api
dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=60))
r = dist.get_rank()
grp = dist.new_group([1, 2])                    # a group that does not start at global rank 0
if r in (1, 2):
    gr = dist.get_rank(grp)
    rec = np.zeros(5 + gr, dtype=api.HSP_DT); rec["oid"] = 100 * gr + np.arange(5 + gr); rec["score"] = 7 + gr
    for dst in (0, 1):                           # ranks OF THE GROUP
        got = shard.gather_records(rec, dst=dst, group=grp)
        if gr == dst:
            assert got is not None and len(got) == 11 and got["oid"].tolist() == list(range(5)) + [100 + i for i in range(6)], got["oid"]
        else:
            assert got is None
        parts = shard.gather_parts(rec, dst=dst, group=grp)
        assert (parts is None) == (gr != dst) and (parts is None or [len(x) for x in parts] == [5, 6])
    for src in (0, 1):
        out = shard.broadcast_records(rec if gr == src else None, api.HSP_DT, src=src, group=grp)
        assert len(out) == 5 + src and int(out["score"][0]) == 7 + src
    try:
        shard.gather_records(rec, dst=2, group=grp); raise SystemExit("dst outside the group was accepted")
    except ValueError:
        pass
    print("SUBGROUP_OK", gr)
dist.barrier(); dist.destroy_process_group()
'''


def test_gather_and_broadcast_in_a_subgroup_that_does_not_start_at_rank_zero():
    """dst / src of the exchange helpers are ranks OF THE GROUP; torch's calls take global ranks (round 4 mixed the two)."""
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(SUBGROUP_WORKER); path = f.name
    env = dict(os.environ); env["MASTER_ADDR"] = "127.0.0.1"
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
                        "--master-port", "29633", path, root], env=env, capture_output=True, text=True, timeout=300)
    os.unlink(path)
    assert p.returncode == 0 and p.stdout.count("SUBGROUP_OK") == 2, p.stdout[-2000:] + p.stderr[-3000:]
