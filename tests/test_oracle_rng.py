"""Pin the oracle's integer RNG (reference backends/embree/lcg_rng.ih:4-59) against independent
implementations: scikit-learn's MurmurHash3_x86_32 for the hash mixing, the Numerical Recipes
LCG sequence, and exact rational arithmetic for the uint32 -> float conversion."""
import struct

import numpy as np
import pytest

from tests import kat_inputs as K

M32 = 0xFFFFFFFF


def _fmix_inverse(h):
    def unxorshift(x, s):
        y = x
        for _ in range(32 // s + 1):
            y = x ^ (y >> s)
        return y & M32
    h = unxorshift(h, 16)
    h = (h * pow(0xc2b2ae35, -1, 2**32)) & M32
    h = unxorshift(h, 13)
    h = (h * pow(0x85ebca6b, -1, 2**32)) & M32
    h = unxorshift(h, 16)
    return h


def _fmix(h):
    h ^= h >> 16
    h = (h * 0x85ebca6b) & M32
    h ^= h >> 13
    h = (h * 0xc2b2ae35) & M32
    h ^= h >> 16
    return h


def test_sklearn_murmur_is_the_published_one():
    from sklearn.utils import murmurhash3_32
    # published MurmurHash3_x86_32 vectors
    assert murmurhash3_32(b"", 0, positive=True) == 0
    assert murmurhash3_32(b"", 1, positive=True) == 0x514E28B7
    assert murmurhash3_32(b"\xff\xff\xff\xff", 0, positive=True) == 0x76293B50
    assert murmurhash3_32(b"\x21\x43\x65\x87", 0, positive=True) == 0xF55B516B
    assert murmurhash3_32(b"hello", 0, positive=True) == 0x248BFA47


def test_get_rng_matches_murmur3_of_pixel_and_frame(oracle):
    """get_rng(p, f) = fmix(h) where MurmurHash3_x86_32(le32(p) + le32(f), seed 0) = fmix(h ^ 8):
    the reference omits the length xor, everything else is the standard hash."""
    from sklearn.utils import murmurhash3_32
    rec = K.rng_records()
    out = oracle.kat(K.KAT_RNG, rec, 17).view(np.uint32)
    for (p, f), row in zip(rec.view(np.uint32), out):
        canonical = murmurhash3_32(struct.pack("<II", int(p), int(f)), 0, positive=True)
        h = _fmix_inverse(canonical) ^ 8
        assert _fmix(h) == int(row[0])


def test_lcg_sequence_and_float_conversion(oracle):
    rec = K.rng_records()
    out = oracle.kat(K.KAT_RNG, rec, 17)
    bits = out.view(np.uint32)
    for row_b, row_f in zip(bits, out):
        s = int(row_b[0])
        for k in range(8):
            s = (s * 1664525 + 1013904223) & M32
            assert int(row_b[1 + 2 * k]) == s
            # ldexp((float)u32, -32): round-to-nearest-even conversion, exact scaling
            expect = np.float32(np.float64(np.float32(np.uint32(s))) * 2.0 ** -32)
            assert row_f[2 + 2 * k] == expect
    # Numerical Recipes: from state 0 the LCG yields 1013904223, 1196435762, 3519870697, 2868466484
    s, seq = 0, []
    for _ in range(4):
        s = (s * 1664525 + 1013904223) & M32
        seq.append(s)
    assert seq == [1013904223, 1196435762, 3519870697, 2868466484]


def test_randomf_can_reach_one(oracle):
    """Quirk Q2: states >= 0xFFFFFF80 convert to exactly 1.0f."""
    assert np.float32(np.uint32(0xFFFFFF80)) * np.float32(2.0 ** -32) == np.float32(1.0)
    assert np.float32(np.uint32(0xFFFFFF7F)) * np.float32(2.0 ** -32) < np.float32(1.0)


def test_russian_roulette_against_an_independent_float32_derivation(oracle):
    """render_embree.ispc:327-335 as the oracle states it (CRT_KAT_ROULETTE), against numpy float32 arithmetic with the reference's
    select-max `a < b ? b : a` (embree_sycl/render_embree_kernel.inl:284-287: sycl::max) and the LCG above: q, the decision, the
    divided throughput and the RNG state after one draw -- every bit, NaN and inf throughputs included."""
    rec = K.roulette_records(4000)
    out = oracle.kat(K.KAT_ROULETTE, rec, 6)
    tp = rec[:, :3].astype(np.float32)
    sel = lambda a, b: np.where(a < b, b, a).astype(np.float32)  # noqa: E731
    with np.errstate(all="ignore"):
        q = sel(np.full(len(rec), 0.05, np.float32), np.float32(1.0) - sel(tp[:, 0], sel(tp[:, 1], tp[:, 2])))
        state = (rec[:, 3].view(np.uint32).astype(np.uint64) * 1664525 + 1013904223) & M32
        draw = (state.astype(np.uint32).astype(np.float32).astype(np.float64) * 2.0 ** -32).astype(np.float32)
        ended = draw < q
        after = np.where(ended[:, None], tp, tp / (np.float32(1.0) - q)[:, None]).astype(np.float32)
    assert np.array_equal(out[:, 0], ended.astype(np.float32))
    assert np.array_equal(out[:, 4].view(np.uint32), state.astype(np.uint32))
    assert np.array_equal(out[:, 5].view(np.uint32), q.view(np.uint32))
    # (a NaN's payload / sign after a division is the platform's; which values are NaN and every other bit must agree)
    got, want = out[:, 1:4], after
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(got.view(np.uint32)[~np.isnan(want)], want.view(np.uint32)[~np.isnan(want)])
