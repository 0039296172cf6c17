"""GPU parity at the scale of the benchmark configurations (BASELINE.json C3 / C4).

Round 1's parity cases stopped at a few thousand triangles, so the code only big scenes reach was
unverified on hardware: the HBM part of the traversal stack (depth > the entries kept in LDS), deep
trees, leaf references near 2^28, 128 materials / 64 textures in one shading wave, a TLAS of > 1000
instances over several BLASes, multi-pass frames. Two kinds of test:

* FRAMES of reduced C3 / C4 content (hundreds of thousands of triangles, every material / texture /
  instancing feature of the full scene) against the CPU oracle under the image tolerance of
  tests/parity.py: the oracle's cost grows with pixels x spp, not with triangles.
* RAYS through the FULL C3 and C4 trees (6.7 M / ~6 M unique, 10 M instanced triangles): 10^5 probe rays
  and their secondary rays, HIP kernel vs (a) the oracle walking the product's own arrays -- hits
  AND node / triangle visit counts, the roofline input -- and (b) the oracle tracing its OWN BVH,
  built by independent code from the same scene: ids equal, t bit-identical. On the instanced C4 tree the
  deepest stack a probe ray needs must exceed the LDS part, i.e. the HBM slab is exercised.
"""
import os

import numpy as np
import pytest

from chameleonrt_amd import core, scenes
from chameleonrt_amd.render_hip import PreparedScene, RenderHIP
from tests.parity import MAX_DIVERGED, camera_of, compare_images, probe_rays, slot_triangles

pytestmark = pytest.mark.gpu

FRAME_CASES = {
    # C3 content: voxel city, ~0.6 M tiny axis-aligned triangles, 64 flat materials, deep tree
    "rungholt_n400_640x360_2spp": (lambda: scenes.rungholt_like(spp=2, n=400), 640, 360, 2),
    # C4 content: 128 materials over every lobe incl. three dielectrics, 64 textures (sRGB + parameter maps),
    # flattened trees + instanced shrubs (TLAS over 5 BLASes), UVs in [-2, 3]
    "sanmiguel_instanced_640x360_2spp": (
        lambda: scenes.sanmiguel_like(spp=2, detail=0.05, tex_size=256, n_trees=200, leaves_per_tree=600,
                                      n_instanced=100, glass=True), 640, 360, 2),
    # the flattened, OBJ-like variant of the same content (single level)
    "sanmiguel_flat_480x270_4spp": (
        lambda: scenes.sanmiguel_like(spp=4, detail=0.05, tex_size=128, n_trees=120, leaves_per_tree=600), 480, 270, 2),
}


@pytest.mark.parametrize("case", list(FRAME_CASES))
def test_frame_parity_at_scale(case, oracle, hip_lib):
    make, w, h, frames = FRAME_CASES[case]
    sc = make()
    assert sc.total_tris() > 250_000
    r = RenderHIP()
    r.initialize(w, h)
    r.set_scene(sc)
    o = oracle.OracleRenderer(sc, w, h)
    e, d, u, fovy = camera_of(sc)
    try:
        for f in range(frames):
            st = r.render(e, d, u, fovy, f == 0, True)
            ost = o.render(e, d, u, fovy, f == 0)
            diverged, mean_rel = compare_images(r.accum(), o.accum())
            assert diverged <= MAX_DIVERGED, f"frame {f}: {diverged:.5f} of the pixels diverged"
            assert mean_rel <= 1e-4, f"frame {f}: mean relative error {mean_rel:.3g}"
            assert (r.ray_counts() != o.ray_counts()).mean() <= MAX_DIVERGED
            assert abs(int(st.rays) - int(ost.rays)) <= max(16, int(2 * MAX_DIVERGED * ost.rays))
            g8 = r.img.view(np.uint8).reshape(h, w, 4).astype(int)
            c8 = o.framebuffer().view(np.uint8).reshape(h, w, 4).astype(int)
            assert ((np.abs(g8 - c8) > 1).any(axis=2)).mean() <= 2 * MAX_DIVERGED
    finally:
        r.close()


def test_multi_pass_frame_equals_single_pass(hip_lib, monkeypatch):
    """A frame cut into several passes (path capacity < pixel-samples, as C5 needs on one GPU) is the
    same image bit for bit: the RNG is keyed by pixel and sample, never by the pass."""
    sc = scenes.sponza_like(spp=4, detail=0.05, tex_size=64)
    e, d, u, fovy = camera_of(sc)
    imgs = []
    for cap in (None, "20000"):
        if cap:
            monkeypatch.setenv("CRT_HIP_MAX_PATHS", cap)
        r = RenderHIP()
        r.initialize(256, 144)
        r.set_scene(sc)
        for f in range(2):
            st = r.render(e, d, u, fovy, f == 0, True)
        imgs.append((r.accum().copy(), r.ray_counts().copy(), int(st.rays)))
        r.close()
    assert np.array_equal(imgs[0][0], imgs[1][0], equal_nan=True)
    assert np.array_equal(imgs[0][1], imgs[1][1]) and imgs[0][2] == imgs[1][2]


FULL = {
    # textures do not take part in traversal: generated small so the test spends its time on the trees
    "C3_rungholt_full": lambda: scenes.make_workload("C3")[0],
    "C4_sanmiguel_full": lambda: scenes.make_workload("C4", tex_size=32)[0],            # the library's choice: a world tree
    "C4_sanmiguel_full_two_level": lambda: scenes.make_workload("C4", tex_size=32)[0],  # CRT_HIP_LEVELS=two
}


@pytest.fixture(scope="module", params=list(FULL))
def full(request, oracle, hip_lib):
    sc = FULL[request.param]()
    if request.param.endswith("two_level"):
        os.environ["CRT_HIP_LEVELS"] = "two"
    try:
        ps = PreparedScene(sc)
    finally:
        os.environ.pop("CRT_HIP_LEVELS", None)
    bvh = ps.bvh()
    assert bvh["levels"] == {"C3_rungholt_full": 0, "C4_sanmiguel_full": 2, "C4_sanmiguel_full_two_level": 1}[request.param]
    r = RenderHIP(flags=core.FLAG_COUNTERS)
    r.initialize(64, 64)
    r.set_prepared_scene(ps)
    ps.close()
    yield sc, r, bvh, oracle.OracleScene(sc)
    r.close()


def _same_hits(g, c):
    for k in ("inst", "geom", "prim"):
        assert np.array_equal(g[k], c[k]), k
    hit = c["inst"] >= 0
    assert np.array_equal(g["t"][hit].view(np.uint32), c["t"][hit].view(np.uint32))
    return hit


def test_full_tree_rays(full, oracle):
    sc, r, bvh, o = full
    assert slot_triangles(bvh).sum() > 5_000_000
    org, dirs = probe_rays(sc, 100_000, seed=31)
    # primary-like rays: kernel == walk of the product's arrays (hits + visit counts) == the oracle's own BVH
    g = r.trace(org, dirs, 0.0, 1e20, closest=True)
    w = oracle.walk_product_bvh(bvh, org, dirs, 0.0, 1e20, closest=True)
    c = o.trace(org, dirs, 0.0, 1e20, closest=True, brute_force=False)
    hit = _same_hits(g, w)
    _same_hits(g, c)
    assert np.array_equal(g["u"][hit].view(np.uint32), c["u"][hit].view(np.uint32))
    assert np.array_equal(g["v"][hit].view(np.uint32), c["v"][hit].view(np.uint32))
    assert hit.mean() > 0.3
    assert (g["stats"].closest_nodes, g["stats"].closest_tris) == (w["nodes"], w["tris"])
    deepest = w["max_stack"]
    g0u = g["u"]
    # incoherent secondary rays leaving the surfaces (tnear = EPSILON): the expensive kind
    p = org[hit] + c["t"][hit, None] * dirs[hit]
    d2 = np.random.default_rng(32).normal(size=p.shape).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    g = r.trace(p, d2, 1e-4, 1e20, closest=True)
    w = oracle.walk_product_bvh(bvh, p, d2, 1e-4, 1e20, closest=True)
    c = o.trace(p, d2, 1e-4, 1e20, closest=True, brute_force=False)
    _same_hits(g, w)
    _same_hits(g, c)
    c2 = c
    assert (g["stats"].closest_nodes, g["stats"].closest_tris) == (w["nodes"], w["tris"])
    deepest = max(deepest, w["max_stack"])
    # occlusion rays over finite segments
    tmax = (np.random.default_rng(33).random(len(p)) * 20).astype(np.float32)
    g = r.trace(p, d2, 1e-4, tmax, closest=False)
    w = oracle.walk_product_bvh(bvh, p, d2, 1e-4, tmax, closest=False)
    c = o.trace(p, d2, 1e-4, tmax, closest=False, brute_force=False)
    assert np.array_equal(g["t"], w["t"]) and np.array_equal(g["t"], c["t"])
    assert (g["stats"].shadow_nodes, g["stats"].shadow_tris) == (w["nodes"], w["tris"])
    deepest = max(deepest, w["max_stack"])
    # the same three ray sets through the PRODUCTION instantiations (the kernels a frame launches, no counters)
    gp = r.trace(org, dirs, 0.0, 1e20, closest=True, production=True)
    hit0 = _same_hits(gp, o.trace(org, dirs, 0.0, 1e20, closest=True, brute_force=False))
    assert np.array_equal(gp["u"][hit0].view(np.uint32), g0u[hit0].view(np.uint32))
    _same_hits(r.trace(p, d2, 1e-4, 1e20, closest=True, production=True), c2)
    assert np.array_equal(r.trace(p, d2, 1e-4, tmax, closest=False, production=True)["t"], c["t"])
    assert deepest <= bvh["stack_need"]
    print(f"\n{sc.name}: deepest traversal stack {deepest}, {bvh['lds_stack']} entries in LDS, {bvh['stack_need']} provided for")
    if bvh["two_level"]:
        # the two-level kernels keep 15 entries in LDS: the instanced C4 tree must go beyond them, i.e. exercise the
        # HBM part of the stack (the single-level kernels keep 16, those of a world tree 14, which their trees may never exceed)
        assert deepest > bvh["lds_stack"], f"deepest stack {deepest}: the HBM part of the traversal stack was never used"
