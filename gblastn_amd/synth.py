"""Synthetic workloads of BASELINE.md section 2: i.i.d. uniform 2-bit databases and
1 kb queries (80 % random, 20 % carrying one planted homolog).

The database byte stream is defined by the xorshift64* generator that
gbn_synth_fill runs on the device (csrc/kernels.hip: synth_fill_kernel): one
stream per 4 KiB chunk.  synth_bytes_numpy() is its bit-identical numpy form,
used to materialise subjects on the host (oracle input, planted queries).
"""
import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_MUL = np.uint64(0x2545F4914F6CDD1D)
CHUNK_WORDS = 512


def synth_bytes_numpy(byte_start, nbytes, seed):
    """Bytes [byte_start, byte_start+nbytes) of the stream gbn_synth_fill(seed) writes."""
    with np.errstate(over="ignore"):
        w0 = byte_start // 8
        w1 = (byte_start + nbytes + 7) // 8
        c0, c1 = w0 // CHUNK_WORDS, (w1 + CHUNK_WORDS - 1) // CHUNK_WORDS
        chunks = np.arange(c0, c1, dtype=np.uint64)
        x = np.uint64(seed) ^ (GOLDEN * (chunks + np.uint64(1)))
        x ^= x >> np.uint64(30); x *= _M1; x ^= x >> np.uint64(27); x *= _M2; x ^= x >> np.uint64(31)
        x[x == 0] = GOLDEN
        out = np.empty((len(chunks), CHUNK_WORDS), dtype=np.uint64)
        for w in range(CHUNK_WORDS):
            x ^= x >> np.uint64(12); x ^= x << np.uint64(25); x ^= x >> np.uint64(27)
            out[:, w] = x * _MUL
        raw = out.reshape(-1).view(np.uint8)
        lo = byte_start - c0 * CHUNK_WORDS * 8
        return raw[lo:lo + nbytes].copy()


def unpack_bases(packed, n):
    p = np.asarray(packed, dtype=np.uint8)[: (n + 3) // 4]
    out = np.empty((len(p), 4), dtype=np.uint8)
    out[:, 0] = p >> 6; out[:, 1] = (p >> 4) & 3; out[:, 2] = (p >> 2) & 3; out[:, 3] = p & 3
    return out.reshape(-1)[:n]


def pack_bases(bases, pad=16):
    b = np.asarray(bases, dtype=np.uint8) & 3
    n = len(b)
    full = np.zeros(((n + 3) // 4) * 4, dtype=np.uint8)
    full[:n] = b
    q = full.reshape(-1, 4)
    out = (q[:, 0] << 6) | (q[:, 1] << 4) | (q[:, 2] << 2) | q[:, 3]
    return np.concatenate([out.astype(np.uint8), np.zeros(pad, dtype=np.uint8)])


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = np.uint64(x) + GOLDEN
        x = (x ^ (x >> np.uint64(30))) * _M1
        x = (x ^ (x >> np.uint64(27))) * _M2
        return x ^ (x >> np.uint64(31))


def family_element(seed):
    """the 300 packed bytes (1,200 bases) of the database's interspersed repeat family (synth_skew_kernel)"""
    out = np.empty(300, dtype=np.uint8)
    for k in range(300):
        out[k] = (int(_splitmix64(np.uint64(seed) ^ np.uint64(0xFA111 + (k >> 3)))) >> (8 * (k & 7))) & 0xff
    return out


def skew_subject(raw, nb, oid, seed):
    """synth_skew_kernel on one subject's packed bytes (raw[:nb], modified in place): four stretches of nb // 50 bytes as homopolymer
    runs or tandem repeats, and in one subject of fifty a slightly changed copy of the family element"""
    if nb < 2000:
        return raw
    rl = nb // 50
    with np.errstate(over="ignore"):
        for r in range(4):
            h = int(_splitmix64(np.uint64(seed) ^ _splitmix64(np.uint64(oid * 4 + r))))
            start = h % (nb - rl)
            h2 = int(_splitmix64(np.uint64(h)))
            if ((h >> 40) & 1) == 0:
                raw[start:start + rl] = (0x55 * ((h >> 42) & 3)) & 0xff
            else:
                period = 1 + ((h >> 44) % 6)
                unit = np.array([(h2 >> (8 * j)) & 0xff for j in range(period)], dtype=np.uint8)
                raw[start:start + rl] = np.tile(unit, rl // period + 1)[:rl]
        he = int(_splitmix64(np.uint64(seed) ^ _splitmix64(np.uint64(oid) ^ np.uint64(0xE1E1E1E1))))
        if he % 50 == 0:
            at = int(_splitmix64(np.uint64(he))) % (nb - 300)
            e = family_element(seed).copy()
            for k in range(300):
                hm = int(_splitmix64(np.uint64(he) ^ np.uint64(k + 1)))
                if (hm & 15) == 0:
                    e[k] ^= (((hm >> 4) & 3) << (2 * ((hm >> 6) & 3))) & 0xff
            raw[at:at + 300] = e
    return raw


class SynthDb:
    """Layout of a synthetic shard: `num` subjects of `length` bases, 16-byte aligned.  skew: repeats written over the uniform
    bases (gbn_synth_skew / skew_subject)."""

    def __init__(self, num, length, seed=0, first_oid=0, skew=False):
        self.skew = skew
        self.num, self.length, self.seed, self.first_oid = num, length, seed, first_oid
        self.stride = (((length + 3) // 4) + 15) // 16 * 16
        self.front = 16
        self.nbytes = self.front + num * self.stride + 128
        self.nbytes = (self.nbytes + 7) // 8 * 8
        self.byte_off = self.front + np.arange(num, dtype=np.int64) * self.stride
        self.lens = np.full(num, length, dtype=np.int32)

    def subject_packed(self, i, pad=16):
        """Packed bytes of subject i (local index) exactly as the device slab holds them."""
        nb = (self.length + 3) // 4
        raw = synth_bytes_numpy(int(self.byte_off[i]), nb + pad, self.seed)
        if self.skew:
            skew_subject(raw, nb, self.first_oid + i, self.seed)
        return raw

    def skew_on_device(self, api, slab_ptr):
        """the same repeats over the slab gbn_synth_fill wrote"""
        api._check(api.lib().gbn_synth_skew(slab_ptr, int(self.byte_off[0]), int(self.stride), (self.length + 3) // 4, self.num, self.first_oid, self.seed, None))

    def subject_bases(self, i):
        return unpack_bases(self.subject_packed(i, pad=0), self.length)

    def host_slab(self):
        return synth_bytes_numpy(0, self.nbytes, self.seed)


_COMP = np.array([3, 2, 1, 0], dtype=np.uint8)


def make_queries(nq, db, qlen=1000, first_query_id=0, planted_fraction=0.2, family_fraction=0.0):
    """Queries per BASELINE.md: seed 42 + query_id; planted ones copy a 300-900 base
    slice of a chosen subject with 0-5 % substitutions and 0-2 single-base indels,
    random strand.  Returns (list of uint8 BLASTNA arrays, list of plant records)."""
    queries, plants = [], []
    for k in range(nq):
        qid = first_query_id + k
        rng = np.random.default_rng(42 + qid)
        q = rng.integers(0, 4, size=qlen, dtype=np.uint8)
        if rng.random() < planted_fraction and db is not None and db.num > 0:
            subj = int(rng.integers(0, db.num))
            ln = int(rng.integers(300, 901))
            ln = min(ln, db.length, qlen)
            s0 = int(rng.integers(0, db.length - ln + 1))
            piece = db.subject_bases(subj)[s0:s0 + ln].copy()
            rate = rng.random() * 0.05
            mut = rng.random(ln) < rate
            piece[mut] = (piece[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) & 3
            for _ in range(int(rng.integers(0, 3))):
                pos = int(rng.integers(1, len(piece) - 1))
                if rng.random() < 0.5:
                    piece = np.delete(piece, pos)
                else:
                    piece = np.insert(piece, pos, rng.integers(0, 4, dtype=np.uint8))
            if rng.random() < 0.5:
                piece = _COMP[piece[::-1]]
            piece = piece[:qlen]
            q0 = int(rng.integers(0, qlen - len(piece) + 1))
            q[q0:q0 + len(piece)] = piece
            plants.append(dict(query=qid, subject=db.first_oid + subj, s0=s0, length=ln, q0=q0))
        elif family_fraction > 0 and rng.random() < family_fraction:
            # a 600-base piece of the database's repeat family, 2 % substitutions: hits every copy (one subject in fifty)
            fam = unpack_bases(family_element(db.seed), 1200)
            a = int(rng.integers(0, 600))
            piece = fam[a:a + 600].copy()
            mut = rng.random(600) < 0.02
            piece[mut] = (piece[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) & 3
            q0 = int(rng.integers(0, qlen - 600 + 1))
            q[q0:q0 + 600] = piece
        queries.append(q)
    return queries, plants
