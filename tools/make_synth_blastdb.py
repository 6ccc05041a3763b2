#!/usr/bin/env python
"""Write a synthetic nucleotide BLAST database (format version 4: .nin / .nsq volumes + a .nal alias) and a FASTA file of
queries with planted homologies -- the input of the reference's documented invocation
(`blastn -db ... -query ... -outfmt 7 -use_gpu true`, shell/g.m.sh:10) at BASELINE.json's C2 size, for
`bench.py --workload cli`.

    python tools/make_synth_blastdb.py OUTDIR [--subjects 50000] [--subject-len 1000000] [--volumes 13] [--queries 10000]

Layout of a volume as csrc/dbreader.cpp and oracle/orc.py: read_blastdb_v4_nucl read it (seqdb_reader/index_files.txt:62-120):
.nin = version 4, type 0 (nucleotide), title, date, number of OIDs, volume length (little-endian 8 bytes), longest sequence,
then the header-, sequence- and ambiguity-offset arrays (big-endian 32 bit, OIDs + 1 entries each); .nsq = a NUL byte, then per
sequence its NCBI2na bytes (4 bases per byte, base 0 in bits 7..6) whose last byte holds the number of valid bases of that
byte in its low 2 bits (a byte of its own when the length is a multiple of 4).  No ambiguity runs, no .nhr (the search reads
neither).  Deterministic in --seed."""
import argparse
import os
import struct
import sys
import time

import numpy as np


def write_volume(prefix, packed_rows, seq_len, title):
    """packed_rows: uint8 [n, seq_len / 4] (seq_len a multiple of 4)"""
    n, nb = packed_rows.shape
    assert nb * 4 == seq_len
    rec = nb + 1                                            # + the byte that says "0 valid bases in me"
    with open(prefix + ".nsq", "wb") as f:
        f.write(b"\0")
        body = np.zeros((n, rec), dtype=np.uint8)
        body[:, :nb] = packed_rows
        f.write(body.tobytes())
    seq_off = 1 + rec * np.arange(n + 1, dtype=np.int64)
    amb_off = seq_off.copy(); amb_off[:n] += rec            # no ambiguity data: it starts (and ends) where the next sequence starts
    assert seq_off[-1] < (1 << 31)
    date = time.strftime("%b %d, %Y  %I:%M %p").encode()
    with open(prefix + ".nin", "wb") as f:
        f.write(struct.pack(">ii", 4, 0))
        for s in (title.encode(), date):
            f.write(struct.pack(">i", len(s))); f.write(s)
        f.write(struct.pack(">i", n)); f.write(struct.pack("<q", n * seq_len)); f.write(struct.pack(">i", seq_len))
        f.write(np.zeros(n + 1, dtype=">i4").tobytes())      # header offsets (no .nhr)
        f.write(seq_off.astype(">i4").tobytes())
        f.write(amb_off.astype(">i4").tobytes())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("--name", default="c2db")
    ap.add_argument("--subjects", type=int, default=50_000)
    ap.add_argument("--subject-len", type=int, default=1_000_000)
    ap.add_argument("--volumes", type=int, default=13)
    ap.add_argument("--queries", type=int, default=10_000)
    ap.add_argument("--query-len", type=int, default=1000)
    ap.add_argument("--planted-fraction", type=float, default=0.5)
    ap.add_argument("--seed", type=int, default=20260930)
    a = ap.parse_args()
    assert a.subject_len % 4 == 0
    os.makedirs(a.outdir, exist_ok=True)
    rng = np.random.default_rng(a.seed)
    nb = a.subject_len // 4
    # which queries carry a homology, and from where (subject, offset)
    planted = rng.random(a.queries) < a.planted_fraction
    p_subj = rng.integers(0, a.subjects, a.queries)
    p_off = rng.integers(0, a.subject_len - a.query_len, a.queries) & ~3
    queries = [None] * a.queries
    t0 = time.time()
    names = []
    for v in range(a.volumes):
        s0, s1 = v * a.subjects // a.volumes, (v + 1) * a.subjects // a.volumes
        rows = rng.integers(0, 256, size=(s1 - s0, nb), dtype=np.uint8)
        name = "%s.%02d" % (a.name, v)
        write_volume(os.path.join(a.outdir, name), rows, a.subject_len, "synthetic volume %d" % v)
        names.append(name)
        for qi in np.nonzero(planted & (p_subj >= s0) & (p_subj < s1))[0]:
            b = rows[p_subj[qi] - s0, p_off[qi] // 4: p_off[qi] // 4 + a.query_len // 4 + 1]
            bases = np.stack([b >> 6, (b >> 4) & 3, (b >> 2) & 3, b & 3], axis=1).reshape(-1)[:a.query_len].copy()
            m = rng.integers(0, a.query_len, a.query_len // 50)             # 2 % substitutions
            bases[m] = (bases[m] + 1 + rng.integers(0, 3, len(m))) & 3
            if rng.random() < 0.5:                                           # half of them on the minus strand
                bases = (3 - bases)[::-1]
            queries[qi] = bases
        del rows
        print("volume %d / %d written (%.0f s)" % (v + 1, a.volumes, time.time() - t0), file=sys.stderr)
    with open(os.path.join(a.outdir, a.name + ".nal"), "w") as f:
        f.write("#\n# synthetic database of tools/make_synth_blastdb.py\n#\nTITLE synthetic %d x %d bases\nDBLIST %s\nNSEQ %d\nLENGTH %d\n"
                % (a.subjects, a.subject_len, " ".join(names), a.subjects, a.subjects * a.subject_len))
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    with open(os.path.join(a.outdir, "queries.fa"), "wb") as f:
        for qi in range(a.queries):
            q = queries[qi] if queries[qi] is not None else rng.integers(0, 4, a.query_len, dtype=np.uint8)
            f.write(b">q%05d%s\n" % (qi, b" planted_in_%d_at_%d" % (p_subj[qi], p_off[qi]) if queries[qi] is not None else b""))
            s = letters[q].tobytes()
            for i in range(0, len(s), 70):
                f.write(s[i:i + 70]); f.write(b"\n")
    print("database %s: %d volumes, %d sequences, %.1f Gbp; %d queries (%d planted) in %.0f s"
          % (os.path.join(a.outdir, a.name), a.volumes, a.subjects, a.subjects * a.subject_len / 1e9, a.queries, int(planted.sum()), time.time() - t0), file=sys.stderr)


if __name__ == "__main__":
    main()
