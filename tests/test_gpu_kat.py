"""GPU parity: device shading functions vs the CPU oracle on the same seeded records.

Integer work (RNG, sRGB8 quantisation, texel addressing) must be bit-exact. Floating point:
+ - * / sqrt are evaluated in the same order with contraction off, so results differ only
through libm transcendentals (pow, log, sin, cos, atan2, acos); tolerance 1e-5 relative +
1e-6 absolute, stated per test.
"""
import numpy as np
import pytest

from chameleonrt_amd import scenes
from chameleonrt_amd.render_hip import RenderHIP
from tests import kat_inputs as K

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pair(oracle, hip_lib):
    sc = scenes.instanced_grove()
    r = RenderHIP()
    r.initialize(64, 64)
    r.set_scene(sc)
    yield r, oracle.OracleScene(sc), sc
    r.close()


def _close(a, b, rtol, atol):
    both_nan = np.isnan(a) & np.isnan(b)
    both_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
    ok = both_nan | both_inf | (np.abs(a - b) <= atol + rtol * np.abs(b))
    return ok


def test_rng_bit_exact(pair):
    r, o, _ = pair
    rec = K.rng_records()
    g = r.kat(K.KAT_RNG, rec, 17).view(np.uint32)
    c = o.kat(K.KAT_RNG, rec, 17).view(np.uint32)
    assert np.array_equal(g, c)


def test_disney_eval(pair):
    r, o, _ = pair
    rec = K.disney_eval_records(20000)
    g, c = r.kat(K.KAT_DISNEY_EVAL, rec, 4), o.kat(K.KAT_DISNEY_EVAL, rec, 4)
    ok = _close(g, c, 2e-5, 1e-6)
    assert ok.all(), f"{(~ok).sum()} mismatches, worst {np.nanmax(np.abs(g - c))}"


def test_disney_eval_is_bit_exact_once_the_oracle_forms_schlick_like_the_product(pair, oracle):
    """The 2e-5 bar above, explained rather than assumed. The product forms schlick_weight's fifth power as (x^2)^2 * x where the
    reference (disney_bsdf.ih:74-76) and the oracle call pow(x, 5) (a documented deviation). With the
    oracle's TEST switch making it multiply too, every BRDF value in which no other transcendental takes part must equal the
    device's TO THE BIT: that is every record without a clear-coat contribution (the clear-coat lobe's GTR1 has a log; with
    clearcoat = 0 it is multiplied away). The pdf always contains that log (gtr_1_pdf is one of its summands): it stays
    under the tolerance. So the gap of test_disney_eval = the fifth power + libm's log, nothing else."""
    r, o, _ = pair
    rec = K.disney_eval_records(20000)
    g = r.kat(K.KAT_DISNEY_EVAL, rec, 4)
    c_pow = o.kat(K.KAT_DISNEY_EVAL, rec, 4)
    oracle.lib().orc_set_schlick_by_multiplication(1)
    try:
        c_mul = o.kat(K.KAT_DISNEY_EVAL, rec, 4)
    finally:
        oracle.lib().orc_set_schlick_by_multiplication(0)
    no_coat = rec[:, 10] == 0.0
    assert no_coat.sum() > 5000
    gb, cb = g[no_coat, :3].view(np.uint32), c_mul[no_coat, :3].view(np.uint32)
    same = (gb == cb) | (np.isnan(g[no_coat, :3]) & np.isnan(c_mul[no_coat, :3]))
    assert same.all(), f"{(~same).any(axis=1).sum()} of {no_coat.sum()} clear-coat-free records differ in the BRDF with pow out of the picture"
    # and the switch matters: against the reference's pow the same records are NOT all bit-equal (else the test shows nothing)
    assert (g[no_coat, :3].view(np.uint32) != c_pow[no_coat, :3].view(np.uint32)).any()
    assert _close(g, c_mul, 2e-5, 1e-6).all()


def test_disney_sample(pair):
    r, o, _ = pair
    rec = K.disney_sample_records(20000)
    g, c = r.kat(K.KAT_DISNEY_SAMPLE, rec, 8), o.kat(K.KAT_DISNEY_SAMPLE, rec, 8)
    assert np.array_equal(g[:, 7].view(np.uint32), c[:, 7].view(np.uint32)), "RNG consumption differs"
    # w_i: unit vectors, absolute 1e-5; f and pdf relative 2e-4. Near-specular lobes (roughness -> 0,
    # alpha clamped to 0.001) amplify a 1-ulp sin/cos/pow difference by ~1/alpha^2, so a small
    # fraction of the records may exceed the tolerance
    ok = _close(g[:, 3:6], c[:, 3:6], 0, 1e-5).all(axis=1) & _close(g[:, [0, 1, 2, 6]], c[:, [0, 1, 2, 6]], 2e-4, 1e-6).all(axis=1)
    assert ok.mean() > 0.995, f"{(~ok).sum()} of {len(ok)} records out of tolerance"


def test_sample_direct_light_around_its_occlusion_queries(pair):
    """render_embree.ispc:105-181 without the two rtcOccluded calls (CRT_KAT_NEE): light pick and sample, both pdfs, the
    MIS weights, the BSDF-sample branch and its test against the quad -- the part of next-event estimation k_shade runs
    (nee_setup) before k_trace_shadow resolves the visibilities. RNG consumption and the light-sample ray are bit-exact
    (+ - * / sqrt only); the contributions carry the BSDF's transcendentals."""
    r, o, sc = pair
    rec = K.nee_records(20000, sc.lights[0])
    g, c = r.kat(K.KAT_NEE, rec, 17), o.kat(K.KAT_NEE, rec, 17)
    assert np.array_equal(g[:, 3:7].view(np.uint32), c[:, 3:7].view(np.uint32)), "light-sample ray (direction, distance)"
    same_b = g[:, 7] == c[:, 7]
    assert same_b.mean() > 0.999 and (c[:, 7] == 1).mean() > 0.02, (same_b.mean(), (c[:, 7] == 1).mean())
    assert np.array_equal(g[same_b, 15].view(np.uint32), c[same_b, 15].view(np.uint32)), "RNG consumption differs"
    assert np.array_equal(g[same_b, 16], c[same_b, 16]), "occlusion rays counted"
    assert (c[:, 0:3] != 0).any(axis=1).mean() > 0.3, "too few records with a light-sample contribution"
    ok_a = _close(g[:, 0:3], c[:, 0:3], 2e-4, 1e-6).all(axis=1)
    b = same_b & (c[:, 7] == 1)
    ok_cb = _close(g[b, 8:11], c[b, 8:11], 2e-4, 1e-6).all(axis=1)
    ok_dir = _close(g[b, 11:14], c[b, 11:14], 0, 1e-5).all(axis=1)
    ok_t = _close(g[b, 14], c[b, 14], 2e-5, 1e-6)  # exact ops, but along a direction that carries sin / cos ulps
    msg = f"c_a {(~ok_a).sum()} / {len(ok_a)}, c_b {(~ok_cb).sum()}, w_i {(~ok_dir).sum()}, t {(~ok_t).sum()} / {int(b.sum())} out of tolerance"
    assert ok_a.mean() > 0.995 and (ok_cb & ok_dir & ok_t).mean() > 0.99, msg


def test_russian_roulette_bit_exact(pair):
    """render_embree.ispc:327-335 (CRT_KAT_ROULETTE): q, the draw, the decision and throughput / (1 - q) -- + - * / and compares
    only, so EVERY output bit must match, NaN payloads included; 20 000 records over dim, bright (q = 0.05), switch-over, zero,
    negative and non-finite throughputs (NaN / inf in any subset of the channels: the reference's max is the select
    `a < b ? b : a`, not fmax -- [NaN, 0.5, 0.2] gives q = 0.05, [0.5, NaN, 0.2] gives q = 0.5)."""
    r, o, _ = pair
    rec = K.roulette_records(20000)
    g, c = r.kat(K.KAT_ROULETTE, rec, 6), o.kat(K.KAT_ROULETTE, rec, 6)
    assert np.array_equal(g.view(np.uint32), c.view(np.uint32))
    # the oracle's own answers on the position-dependent cases (so that both sides agreeing on something else would show)
    assert c[0, 5] == np.float32(0.05) and c[1, 5] == np.float32(0.5) and c[2, 5] == np.float32(0.5)
    assert c[6, 5] == np.float32(0.05) and c[8, 5] == np.float32(1.0) and c[8, 0] == 1.0  # q = 1: lcg_randomf < 1 always ends the path
    ended = c[:, 0] == 1.0
    assert 0.2 < ended.mean() < 0.8  # both outcomes are exercised


def test_lights(pair):
    r, o, _ = pair
    rec = K.light_records(5000)
    g, c = r.kat(K.KAT_LIGHT, rec, 9), o.kat(K.KAT_LIGHT, rec, 9)
    assert np.array_equal(g.view(np.uint32), c.view(np.uint32)), "light functions use only + - * /: bit-exact"


def test_texture(pair):
    r, o, sc = pair
    rec = K.texture_records(20000, len(sc.textures))
    g, c = r.kat(K.KAT_TEXTURE, rec, 5), o.kat(K.KAT_TEXTURE, rec, 5)
    assert np.array_equal(g.view(np.uint32), c.view(np.uint32)), "bilinear fetch is + - * / floor: bit-exact"


def test_textures_of_awkward_sizes(oracle, hip_lib):
    """Texels are stored in 8 x 4 tiles (crt_types.h tex_slot): sizes that are no multiple of the tile, non-square,
    narrower / lower than one tile, every channel count -- through the generic lookup (KAT_TEXTURE) and through the
    whole-texel path of unpack_material (4-channel textures), bit for bit against the oracle, which reads rows."""
    from chameleonrt_amd.scene import Image, LINEAR, SRGB
    sc = scenes.instanced_grove()
    rng = np.random.default_rng(5)
    sizes = [(13, 7), (5, 9), (20, 33)]  # (width, height), replacing the grove's 4-, 3- and 4-channel textures
    assert [t.channels for t in sc.textures] == [4, 3, 4]
    for k, (w, h) in enumerate(sizes):
        old = sc.textures[k]
        sc.textures[k] = Image(w, h, old.channels, rng.integers(0, 256, size=(h, w, old.channels), dtype=np.uint8),
                               old.color_space)
    for w, h, ch in [(1, 1, 1), (9, 4, 2), (3, 70, 1), (130, 2, 4)]:
        sc.textures.append(Image(w, h, ch, rng.integers(0, 256, size=(h, w, ch), dtype=np.uint8), SRGB if ch > 2 else LINEAR))
    r = RenderHIP()
    r.initialize(64, 64)
    r.set_scene(sc)
    o = oracle.OracleScene(sc)
    rec = K.texture_records(40000, len(sc.textures), seed=21)
    g, c = r.kat(K.KAT_TEXTURE, rec, 5), o.kat(K.KAT_TEXTURE, rec, 5)
    assert np.array_equal(g.view(np.uint32), c.view(np.uint32))
    rec = K.unpack_records(20000, len(sc.materials), seed=22)
    g, c = r.kat(K.KAT_UNPACK_MATERIAL, rec, 14), o.kat(K.KAT_UNPACK_MATERIAL, rec, 14)
    assert np.array_equal(g.view(np.uint32), c.view(np.uint32))
    r.close()


def test_unpack_material(pair):
    r, o, sc = pair
    rec = K.unpack_records(5000, len(sc.materials))
    g, c = r.kat(K.KAT_UNPACK_MATERIAL, rec, 14), o.kat(K.KAT_UNPACK_MATERIAL, rec, 14)
    assert np.array_equal(g.view(np.uint32), c.view(np.uint32))


def test_miss_shader(pair):
    r, o, _ = pair
    rec = K.dir_records(20000)
    g, c = r.kat(K.KAT_MISS, rec, 3), o.kat(K.KAT_MISS, rec, 3)
    # the checker flips where atan2/acos cross a tenth: allow a handful of boundary cases
    assert (g != c).any(axis=1).mean() < 2e-4


def test_ortho_basis(pair):
    r, o, _ = pair
    rec = K.dir_records(5000)
    g, c = r.kat(K.KAT_ORTHO_BASIS, rec, 6), o.kat(K.KAT_ORTHO_BASIS, rec, 6)
    assert np.array_equal(g.view(np.uint32), c.view(np.uint32))


def test_srgb8(pair):
    r, o, _ = pair
    rec = K.srgb_records(20000)
    g, c = r.kat(K.KAT_SRGB8, rec, 1), o.kat(K.KAT_SRGB8, rec, 1)
    # powf differs by ulps: a value within an ulp of a quantisation step may land on either side
    assert np.abs(g - c).max() <= 1 and (g != c).mean() < 1e-3
