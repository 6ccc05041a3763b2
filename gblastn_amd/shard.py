"""Database sharding across the GPUs of one node.

Subjects (BLAST volumes) are independent work units: each rank keeps its own
contiguous block of volumes resident in HBM, queries are replicated, every
rank uses the GLOBAL database length / sequence count for its statistics, and
global OID = shard base + local OID.  The only exchange is one variable-length
gather of per-shard preliminary HSP records to rank 0 per query batch
(counts first, then payload) over torch.distributed -- RCCL on GPUs ("nccl"
backend), gloo in the CPU tests.  The reference has no counterpart: its
threads merge through one mutex-guarded BlastHSPStream
(CORE/blast_hspstream.c:316-365).
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(num_volumes, world_size, rank):
    """Contiguous volumes -> ranks; the first (num_volumes % world_size) ranks get one more."""
    base, extra = divmod(num_volumes, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_records(records, dst=0, device=None, group=None):
    """Gather a 1-D structured numpy array from every rank to `dst`.

    Returns the concatenation in rank order on `dst` (ascending global OID when
    shards are contiguous blocks of OIDs), None elsewhere."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return records
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    raw = np.ascontiguousarray(records).view(np.uint8).reshape(-1)
    count = torch.tensor([raw.size], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, count, group=group)
    sizes = [int(c.item()) for c in counts]
    cap = max(max(sizes), 1)
    buf = torch.zeros(cap, dtype=torch.uint8, device=dev)
    if raw.size:
        buf[:raw.size] = torch.from_numpy(raw.copy()).to(dev)
    if rank == dst:
        parts = [torch.zeros(cap, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.gather(buf, parts, dst=dst, group=group)
        out = [p[:n].cpu().numpy() for p, n in zip(parts, sizes)]
        cat = np.concatenate(out) if out else np.zeros(0, dtype=np.uint8)
        return cat.view(records.dtype)
    dist.gather(buf, None, dst=dst, group=group)
    return None


def collect_on_root(records, num_queries, hitlist_size, dst=0, device=None, group=None):
    """One query batch's exchange + merge step: gather every shard's preliminary HSP records to
    `dst` and replay them there, in ascending global OID order, through the per-query top-N
    collector (CORE/blast_hspstream.c:316-365 is the reference's merge point).

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
