"""blastn over a database sharded by volume, one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        -m gblastn_amd.blastn_sharded -db NAME -query Q.fa [-task megablast|blastn] [-evalue E]
        [-max_target_seqs N] [-dust yes|no] [-out FILE] [-trace_t_num T]

Rank r keeps volumes shard_bounds(num_volumes, N, r) of NAME(.nal) resident in its GPU's HBM; every rank reads the
queries, searches its shard with the statistics of the whole database and runs the traceback of its own subjects;
rank 0 merges and prints the twelve standard tabular columns -- the rows `blastn_prelim -db NAME` prints on one GPU
(the multi-GPU counterpart of GB/gpu_blast_multi_gpu_utils.cpp:105-139, which maps host threads to GPUs).
Started without torch.distributed.run it is a single rank."""
import argparse
import os
import sys
import numpy as np

IUPAC = {c: i for i, c in enumerate("ACGTRYMKWSBDHVN-")}
IUPAC.update({c.lower(): i for c, i in list(IUPAC.items())})
IUPAC["U"] = IUPAC["u"] = 3


def read_fasta(path):
    out, name, parts = [], None, []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            if line.startswith(">"):
                if name is not None:
                    out.append((name, "".join(parts)))
                name, parts = line[1:].split()[0] if len(line) > 1 else "Query_%d" % (len(out) + 1), []
            else:
                parts.append(line)
    if name is not None:
        out.append((name, "".join(parts)))
    return [(n, np.array([IUPAC.get(c, 14) for c in s], dtype=np.uint8)) for n, s in out]


def evalue_string(e):
    """objtools/align_format/align_format_util.cpp:694-713"""
    if e < 1.0e-180: return "0.0"
    if e < 1.0e-99: return "%2.0e" % e
    if e < 0.0009: return "%3.0e" % e
    if e < 0.1: return "%4.3f" % e
    if e < 1.0: return "%3.2f" % e
    if e < 10.0: return "%2.1f" % e
    return "%5.0f" % e


def bits_string(s):
    if s > 9999: return "%4.3e" % s
    if s > 99.9: return "%4d" % int(s)
    return "%4.1f" % s


def rows_of(names, lens, rec, query_starts):
    """the 12 std columns of one batch's final records (query-strand coordinates as blastn prints them)"""
    out = []
    for q in range(len(names)):
        for r in rec[query_starts[q]:query_starts[q + 1]]:
            minus = int(r["context"]) & 1
            qlen = lens[q]
            qs, qe = (qlen - r["q_end"] + 1, qlen - r["q_offset"]) if minus else (r["q_offset"] + 1, r["q_end"])
            ss, se = (r["s_end"], r["s_offset"] + 1) if minus else (r["s_offset"] + 1, r["s_end"])
            al = int(r["align_length"])
            out.append("%s\tgnl|BL_ORD_ID|%d\t%.2f\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%s\t%s" % (
                names[q], r["oid"], 100.0 * r["num_ident"] / al, al, al - r["num_ident"] - r["gaps"], r["gap_opens"],
                qs, qe, ss, se, evalue_string(float(r["evalue"])), bits_string(float(r["bit_score"]))))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(prog="blastn_sharded", prefix_chars="-")
    ap.add_argument("-db", required=True); ap.add_argument("-query", required=True)
    ap.add_argument("-task", default="megablast", choices=["megablast", "blastn"])
    ap.add_argument("-evalue", type=float, default=10.0); ap.add_argument("-max_target_seqs", type=int, default=500)
    ap.add_argument("-dust", default="yes", choices=["yes", "no"]); ap.add_argument("-out", default="-")
    ap.add_argument("-trace_t_num", type=int, default=0)
    ap.add_argument("-backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    a = ap.parse_args(argv)

    import torch
    import torch.distributed as dist
    from gblastn_amd import api, shard
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("blastn_sharded: no HIP device (there is no CPU path)")
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", local_rank % ndev)       # more ranks than devices: they share (tests on one GPU)
    torch.cuda.set_device(dev)
    if world > 1:
        api.share_cpus_among_local_ranks()              # (every rank its share of the CPUs the node grants: csrc gbn_host_cpus)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > ndev and a.backend == "nccl":
            # ranks that share a device (tests on a one-GPU box): RCCL refuses two ranks of one host on one device
            # ("Duplicate GPU detected"), so each rank presents itself as a host of its own and RCCL connects them
            # through its socket transport -- same collectives, same ordering rules, no xGMI
            os.environ.setdefault("NCCL_HOSTID", "gbn-rank-%d" % rank)
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
            os.environ.setdefault("NCCL_IB_DISABLE", "1")
        dist.init_process_group(a.backend, device_id=dev if a.backend == "nccl" else None)
    api._check(api.lib().gbn_init(1, dev.index))
    sh = shard.VolumeShard(a.db, world, rank)
    opt = api.default_options(a.task, evalue=a.evalue, hitlist_size=a.max_target_seqs)
    S = shard.ShardedSearch(sh, opt, device=dev if a.backend == "nccl" else None, trace_threads=a.trace_t_num)
    queries = read_fasta(a.query)
    batch_bases = 5000000 if a.task == "megablast" else 100000      # GetQueryBatchSize, blastinput/blast_input_aux.cpp:66-124
    if os.environ.get("BATCH_SIZE"):
        batch_bases = max(1, int(os.environ["BATCH_SIZE"]))
    batches, cur, acc = [], [], 0
    for q in queries:
        if cur and acc + len(q[1]) > batch_bases:
            batches.append(cur); cur, acc = [], 0
        cur.append(q); acc += len(q[1])
    if cur:
        batches.append(cur)
    # tests: GBN_TEST_FAIL="rank:batch" makes that rank's preliminary search of that batch raise -- the protocol must end on every
    # rank with the failure reported, nobody waiting in a collective (tests/test_shard_gpu.py, eight ranks)
    fail_rank, fail_batch = (int(x) for x in os.environ["GBN_TEST_FAIL"].split(":")) if os.environ.get("GBN_TEST_FAIL") else (-1, -1)
    if fail_rank == rank:
        real = S._search

        def failing(queries, masks, _n=[0]):
            _n[0] += 1
            if _n[0] - 1 == fail_batch:
                raise RuntimeError("injected failure of the preliminary search (GBN_TEST_FAIL)")
            return real(queries, masks)
        S._search = failing
    for b in batches:
        seqs = [s for _, s in b]
        S.submit(seqs, masks=api.dust_masks(seqs) if a.dust == "yes" else None)
    failed = None
    try:
        res = S.results()
    except api.BlastError as e:
        failed = str(e)
    if failed is not None:
        sys.stderr.write("blastn_sharded: rank %d: a batch failed: %s\n" % (rank, failed))
        S.close()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return 3
    if rank == 0:
        out = sys.stdout if a.out == "-" else open(a.out, "w")
        n = 0
        for b, (rec, qs) in zip(batches, res):
            for line in rows_of([nm for nm, _ in b], [len(s) for _, s in b], rec, qs):
                out.write(line + "\n"); n += 1
        if out is not sys.stdout:
            out.close()
        sys.stderr.write("blastn_sharded: %d queries in %d batches on %d ranks (%d volumes, %d..%d here), %d rows\n" % (
            len(queries), len(batches), world, sh.db.num_volumes, sh.volumes[0], sh.volumes[1], n))
    S.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
