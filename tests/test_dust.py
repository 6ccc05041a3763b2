"""Symmetric DUST: the product's implementation (C ABI) against the oracle restatement on many
sequences, and the filter's invariants.  (The reference's own DUST tests fetch GenBank entries.)"""
import numpy as np
import pytest
from gblastn_amd import api
from oracle import orc


def product(seq, **kw):
    return [(a, b) for _, a, b in api.dust_masks([seq], **kw)]


def low_complexity_mix(rng, n):
    parts = []
    while sum(len(p) for p in parts) < n:
        k = rng.integers(0, 6)
        if k == 0:
            parts.append(np.full(int(rng.integers(5, 90)), rng.integers(0, 4), dtype=np.uint8))          # homopolymer
        elif k == 1:
            u = rng.integers(0, 4, int(rng.integers(2, 7)), dtype=np.uint8)
            parts.append(np.tile(u, int(rng.integers(3, 30))))                                            # tandem repeat
        elif k == 2:
            parts.append(rng.choice(np.array([0, 3], dtype=np.uint8), int(rng.integers(10, 120))))        # two-letter stretch
        elif k == 3:
            parts.append(np.full(int(rng.integers(1, 12)), 14, dtype=np.uint8))                           # run of N
        else:
            parts.append(rng.integers(0, 4, int(rng.integers(20, 400)), dtype=np.uint8))                  # complex
    return np.concatenate(parts)[:n]


@pytest.mark.parametrize("seed", range(25))
def test_product_equals_oracle(seed):
    rng = np.random.default_rng(seed)
    seq = low_complexity_mix(rng, int(rng.integers(4, 3000)))
    for level, window, linker in [(20, 64, 1), (10, 32, 5), (40, 64, 1), (20, 16, 1)]:
        want = orc.dust(seq, level, window, linker)
        got = product(seq, level=level, window=window, linker=linker)
        assert got == want, (seed, level, window, linker)


def test_invariants():
    rng = np.random.default_rng(99)
    seq = low_complexity_mix(rng, 20000)
    iv = product(seq)
    assert iv, "a sequence full of repeats must be masked somewhere"
    for (a, b), (c, d) in zip(iv, iv[1:]):
        assert a <= b and b + 1 < c                       # ascending, disjoint, not even abutting
    assert 0 <= iv[0][0] and iv[-1][1] < len(seq)
    assert product(np.zeros(200, dtype=np.uint8)) == [(0, 199)]
    assert product(np.zeros(3, dtype=np.uint8)) == []
    # a complex sequence stays essentially unmasked
    rnd = rng.integers(0, 4, 20000, dtype=np.uint8)
    masked = sum(b - a + 1 for a, b in product(rnd))
    assert masked < 0.02 * len(rnd)
    # out-of-range parameters fall back to the defaults, as in the reference
    assert product(seq, level=1000, window=3, linker=0) == iv


def test_masks_plug_into_the_search_setup():
    rng = np.random.default_rng(5)
    qs = [low_complexity_mix(rng, 1000) for _ in range(6)]
    masks = api.dust_masks(qs)
    assert masks
    opt = api.default_options("megablast", db_length=10**9, db_num_seqs=1000)
    api.BlastPrelimSearch(qs, opt, upload=False, masks=masks)     # accepted: sorted, disjoint, in range
