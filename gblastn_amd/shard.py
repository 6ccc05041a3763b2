"""Database sharding across the GPUs of one node.

Subjects (BLAST volumes) are independent work units: each rank keeps its own contiguous block of volumes resident in
HBM, queries are replicated, every rank uses the GLOBAL database length / sequence count for its statistics, and
global OID = shard base + local OID.  Per query batch the ranks exchange, over torch.distributed -- RCCL on GPUs
("nccl" backend), gloo in the CPU tests:

  1. a variable-length gather of the per-shard preliminary HSP records to rank 0 (counts first, then payload);
     rank 0 replays them in ascending OID order through the per-query top-N collector;
  2. a broadcast of the lists that survived: every rank runs the traceback of ITS subjects (the sequences are in
     its HBM, the host work spreads over the ranks);
  3. a gather of the final records; rank 0 merges them per query (gbn_traceback_merge).

Every rank issues these collectives from ONE worker thread in submission order (Exchange), on that thread's own
device stream, so that they overlap the next batch's preliminary search without ever reordering between ranks.
The reference has no counterpart: its threads merge through one mutex-guarded BlastHSPStream
(CORE/blast_hspstream.c:316-365) and map host threads to GPUs (GB/gpu_blast_multi_gpu_utils.cpp:105-139).
"""
import ctypes as C
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(num_volumes, world_size, rank):
    """Contiguous volumes -> ranks; the first (num_volumes % world_size) ranks get one more."""
    base, extra = divmod(num_volumes, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _active(group):
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def gather_records(records, dst=0, device=None, group=None, force=False):
    """Gather a 1-D structured numpy array from every rank to `dst`.

    Returns the concatenation in rank order on `dst` (ascending global OID when shards are contiguous blocks of
    OIDs), None elsewhere.  force: run the collectives even in a group of one (exercises the backend).

    One all-gather of the byte counts (ONE read-back per exchange: the sizes are needed on the host to size the
    receive buffer), then point-to-point transfers of exactly those sizes into their places of one buffer on `dst`
    -- nothing is padded to the largest shard, nothing is sent by a rank that found nothing (round 3: a padded
    `gather` sized by the largest shard and a `.item()` per rank)."""
    if not (_active(group) or (force and dist.is_initialized())):
        return records
    # `dst` is a rank OF THE GROUP (0 .. world - 1), like every rank in this module; torch's point-to-point calls take
    # global ranks: peer() converts (round 4 compared the group rank with `dst` and sent to the global rank `dst`: in a
    # sub-group that does not start at global rank 0 nobody received)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if not 0 <= dst < world:
        raise ValueError("gather_records: dst = %d is not a rank of this group of %d" % (dst, world))
    dev = device if device is not None else torch.device("cpu")
    raw = np.ascontiguousarray(records).view(np.uint8).reshape(-1)
    mine = torch.tensor([raw.size], dtype=torch.int64, device=dev)
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    try:
        dist.all_gather_into_tensor(counts, mine, group=group)
    except (RuntimeError, NotImplementedError):        # a backend without the flat form
        parts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        counts = torch.cat(parts)
    sizes = [int(v) for v in counts.tolist()]
    peer = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    if rank != dst:
        if raw.size:        # (a batched point-to-point op: RCCL runs it next to the group's other work instead of serialising it)
            for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, torch.from_numpy(raw.copy()).to(dev), peer(dst), group)]):
                q.wait()
        return None
    out = torch.empty(max(sum(sizes), 1), dtype=torch.uint8, device=dev)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    if raw.size:
        out[int(offs[rank]):int(offs[rank + 1])] = torch.from_numpy(raw.copy()).to(dev)
    ops = [dist.P2POp(dist.irecv, out[int(offs[r]):int(offs[r + 1])], peer(r), group)
           for r in range(world) if r != dst and sizes[r] > 0]
    for q in (dist.batch_isend_irecv(ops) if ops else []):
        q.wait()
    return out[:int(offs[world])].cpu().numpy().view(records.dtype)


def gather_parts(records, dst=0, device=None, group=None):
    """as gather_records, but `dst` gets the list of the ranks' arrays"""
    if not _active(group):
        return [records]
    n = np.array([len(records)], dtype=np.int64)
    counts = gather_records(n, dst=dst, device=device, group=group)
    cat = gather_records(records, dst=dst, device=device, group=group)
    if cat is None:
        return None
    cut = np.concatenate([[0], np.cumsum(counts)])
    return [cat[cut[i]:cut[i + 1]] for i in range(len(counts))]


def broadcast_records(records, dtype, src=0, device=None, group=None):
    """A 1-D structured array from `src` to every rank (size first, then payload); `records` is ignored elsewhere."""
    if not _active(group):
        return records
    rank = dist.get_rank(group)                         # `src`: a rank of the group; dist.broadcast wants the global one
    src_global = dist.get_global_rank(group, src) if group is not None else src
    dev = device if device is not None else torch.device("cpu")
    raw = np.ascontiguousarray(records).view(np.uint8).reshape(-1) if rank == src else np.zeros(0, dtype=np.uint8)
    n = torch.tensor([raw.size], dtype=torch.int64, device=dev)
    dist.broadcast(n, src=src_global, group=group)
    size = int(n.item())
    buf = torch.from_numpy(raw.copy()).to(dev) if rank == src else torch.zeros(size, dtype=torch.uint8, device=dev)
    if size:
        dist.broadcast(buf, src=src_global, group=group)
    return buf.cpu().numpy().view(dtype) if size else np.zeros(0, dtype=dtype)


def collect_on_root(records, num_queries, hitlist_size, dst=0, device=None, group=None):
    """One query batch's first exchange + merge step: gather every shard's preliminary HSP records to `dst` and
    replay them there, in ascending global OID order, through the per-query top-N collector (SURVEY.md 8e;
    CORE/blast_hspstream.c:316-365 is the reference's merge point).

    Returns (hsps, list_starts, list_queries) on `dst`, None elsewhere."""
    from . import api
    merged = gather_records(records, dst=dst, device=device, group=group)
    if merged is None:
        return None
    col = api.BlastHSPCollector(num_queries, hitlist_size)
    try:
        col.write(merged)
        return col.close()
    finally:
        col.free()


def merge_final(parts, num_queries, hitlist_size):
    """parts: [(records TB_DT, query_starts)] of the shards' traceback stages -> (records, query_starts) of the
    whole database (gbn_traceback_merge: per query by best e-value, best score, oid; at most hitlist_size)"""
    from . import api
    L = api.lib()
    recs = [np.ascontiguousarray(r, dtype=api.TB_DT) for r, _ in parts]
    qss = [np.ascontiguousarray(q, dtype="<i8") for _, q in parts]
    n = len(parts)
    hp = (C.c_void_p * max(n, 1))(*[r.ctypes.data for r in recs]); qp = (C.c_void_p * max(n, 1))(*[q.ctypes.data for q in qss])
    out = np.zeros(sum(len(r) for r in recs), dtype=api.TB_DT)
    oqs = np.zeros(num_queries + 1, dtype="<i8")
    L.gbn_traceback_merge.restype = C.c_int64
    L.gbn_traceback_merge.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    w = L.gbn_traceback_merge(n, hp, qp, num_queries, hitlist_size, out.ctypes.data, oqs.ctypes.data)
    if w < 0:
        raise api.BlastError(L.gbn_last_error().decode())
    return out[:w], oqs


class Exchange:
    """The one ordered channel a rank uses for its collectives: a single worker thread (FIFO) with its own device
    stream.  Whatever the main thread does meanwhile (the next batch's kernels run on the engine's own streams),
    every rank issues the collectives of batch k before those of batch k + 1."""

    def __init__(self, device=None, group=None):
        self.device, self.group = device, group
        self.stream = None

        def pin():
            if device is not None and device.type == "cuda":
                torch.cuda.set_device(device)
                self.stream = torch.cuda.Stream(device)
        self._pool = ThreadPoolExecutor(max_workers=1, initializer=pin)

    def submit(self, fn, *args, **kw):
        def run():
            if self.stream is not None:
                with torch.cuda.stream(self.stream):
                    r = fn(*args, **kw)
                    self.stream.synchronize()
                    return r
            return fn(*args, **kw)
        return self._pool.submit(run)

    def close(self):
        self._pool.shutdown(wait=True)


class VolumeShard:
    """This rank's block of the volumes of a BLAST database (.nal alias or a single volume), resident in HBM, with
    the statistics of the WHOLE database (BlastSeqSrcGetTotLen / GetNumSeqs span all volumes,
    GB/gpu_blastn_pre_search_engine.cpp:1242; NSEQ / LENGTH of the alias file win when given)."""

    def __init__(self, name, world_size=1, rank=0, load=True):
        from . import api
        self.db = api.BlastDb(name)
        self.world_size, self.rank = world_size, rank
        v0, v1 = shard_bounds(self.db.num_volumes, world_size, rank)
        self.volumes = (v0, v1)
        if v1 > v0:
            first, _ = self.db.volume_range(v0)
            last, n = self.db.volume_range(v1 - 1)
            self.first_oid, self.num_oids = first, last + n - first
        else:       # more ranks than volumes: nothing to search here, the rank still takes part in the exchanges
            self.first_oid, self.num_oids = self.db.num_seqs, 0
        self.db_length, self.db_num_seqs = self.db.stat_length, self.db.stat_num_seqs
        self.src = self.db.load_shard(self.first_oid, self.num_oids) if (load and self.num_oids) else None

    def owns(self, oid):
        return self.first_oid <= oid < self.first_oid + self.num_oids


class ShardedSearch:
    """Query batches against a database sharded by volume: preliminary search and traceback on every rank's own
    subjects, results of the whole database on rank 0, batch by batch in submission order.

    search / trace are the two local stages -- by default the library's (BlastPrelimSearch / BlastTracebackSearch
    on shard.src); the gloo tests replace them with record generators to exercise the exchanges without a GPU:
        search(queries, masks) -> (token, preliminary records HSP_DT grouped by oid)
        trace(token, hsps, list_starts) -> (records TB_DT, query_starts)"""

    def __init__(self, shard, options, device=None, group=None, trace_threads=0, search=None, trace=None):
        from . import api
        self.api, self.shard, self.opt = api, shard, options
        self.opt.db_length, self.opt.db_num_seqs = shard.db_length, shard.db_num_seqs      # global statistics on every rank
        self.device, self.group, self.trace_threads = device, group, trace_threads
        self.ex = Exchange(device, group)
        self._search = search or self._local_search
        self._trace = trace or self._local_trace
        self._pending = []

    # ---- the local stages
    def _local_search(self, queries, masks):
        if self.shard.src is None:
            return None, np.zeros(0, dtype=self.api.HSP_DT)
        ps = self.api.BlastPrelimSearch(queries, self.opt, self.shard.src, masks=masks)
        return ps, ps.run()["hsps"]

    def _local_trace(self, ps, hsps, starts):
        nq = len(ps._q) if ps is not None else 0
        if ps is None or len(starts) <= 1:
            return np.zeros(0, dtype=self.api.TB_DT), None
        tb = self.api.BlastTracebackSearch(ps, self.shard.src)
        try:
            rec, _, qs = tb.run(hsps, starts, threads=self.trace_threads)
        finally:
            tb.close()
        return rec, qs

    # ---- the exchanges of one batch (worker thread)
    def _any_failed(self, failed):
        """one status word per batch: a rank whose local stage failed has taken part in every collective with an empty
        payload; here every rank learns of it"""
        if not _active(self.group):
            return failed
        dev = self.device if self.device is not None else torch.device("cpu")
        t = torch.tensor([1 if failed else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return bool(int(t.item()))

    def _finish(self, token, local, nq, err=None):
        """The collectives of a batch are issued by every rank whatever happens to its LOCAL stages (collector on rank 0,
        selection and traceback of its own lists, final merge): a failure there is kept, the rank goes on with empty
        payloads, and after the last exchange an all-reduce of the status makes every rank raise -- nobody is left
        waiting in a gather or a broadcast for a rank that has dropped out."""
        api, dev, grp = self.api, self.device, self.group
        empty_h, empty_s = np.zeros(0, dtype=api.HSP_DT), np.zeros(1, dtype="<i8")
        # 1: preliminary records -> rank 0 -> per-query top-N
        merged = gather_records(local, dst=0, device=dev, group=grp)
        root = merged is not None
        got = None
        if root:
            try:
                if err is None:
                    col = api.BlastHSPCollector(nq, self.opt.hitlist_size)
                    try:
                        col.write(merged); got = col.close()
                    finally:
                        col.free()
            except Exception as e:      # noqa
                err = "collector on rank 0: %r" % (e,)
            if got is None:
                got = (empty_h, empty_s, None)
        # 2: the lists that survived -> every rank
        hsps = broadcast_records(got[0] if root else None, api.HSP_DT, src=0, device=dev, group=grp)
        starts = broadcast_records(np.asarray(got[1], dtype="<i8") if root else None, np.dtype("<i8"), src=0, device=dev, group=grp)
        rec, qs = np.zeros(0, dtype=api.TB_DT), None
        try:
            if err is None:
                # the lists of this rank's subjects: contiguous, the collector orders lists by (oid, query)
                if len(starts) > 1:
                    first_oid = hsps["oid"][starts[:-1]]
                    mine = np.nonzero((first_oid >= self.shard.first_oid) & (first_oid < self.shard.first_oid + self.shard.num_oids))[0]
                else:
                    mine = np.zeros(0, dtype=np.int64)
                if len(mine):
                    a, b = int(mine[0]), int(mine[-1]) + 1
                    if b - a != len(mine):
                        raise api.BlastError("the lists of this rank's subjects are not contiguous")
                    sel_h = hsps[starts[a]:starts[b]]; sel_s = starts[a:b + 1] - starts[a]
                else:
                    sel_h = hsps[:0]; sel_s = empty_s
                rec, qs = self._trace(token, sel_h, sel_s)
        except Exception as e:      # noqa
            err = "traceback stage: %r" % (e,); rec, qs = np.zeros(0, dtype=api.TB_DT), None
        if qs is None:
            qs = np.zeros(nq + 1, dtype="<i8")
        # 3: final records -> rank 0 -> per-query merge
        recs = gather_parts(np.ascontiguousarray(rec, dtype=api.TB_DT), dst=0, device=dev, group=grp)
        qss = gather_parts(np.ascontiguousarray(qs, dtype="<i8"), dst=0, device=dev, group=grp)
        try:
            if hasattr(token, "close"):
                token.close()
        except Exception:           # noqa
            pass
        out = None
        if recs is not None and err is None:
            try:
                out = merge_final(list(zip(recs, qss)), nq, self.opt.hitlist_size)
            except Exception as e:  # noqa
                err = "final merge: %r" % (e,)
        if self._any_failed(err is not None):
            raise api.BlastError(err if err is not None else "this batch failed on another rank")
        return out

    def submit(self, queries, masks=None, num_queries=None):
        """preliminary search of this batch now (caller's thread); its exchanges, traceback and merge are queued
        behind those of the batches before it and overlap whatever the caller does next.  A search that fails here is
        reported by the batch's future (on every rank), not by this call: the rank still owes the others its exchanges."""
        err = None
        try:
            token, local = self._search(queries, masks)
        except Exception as e:      # noqa
            token, local, err = None, np.zeros(0, dtype=self.api.HSP_DT), "preliminary search: %r" % (e,)
        fut = self.ex.submit(self._finish, token, local, len(queries) if num_queries is None else num_queries, err)
        self._pending.append(fut)
        return fut

    def results(self):
        """finished batches in submission order: (records TB_DT, query_starts) on rank 0, None elsewhere"""
        out = [f.result() for f in self._pending]
        self._pending = []
        return out

    def close(self):
        self.ex.close()
