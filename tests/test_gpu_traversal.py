"""GPU parity: BVH4 traversal + triangle intersection (kernels K2/K4).

The closest hit is defined as the lexicographic minimum of (t, inst, geom, prim) over all
valid triangle hits, so the HIP kernel walking its own SAH BVH must return EXACTLY what the
oracle gets by testing every triangle with no BVH at all: ids equal, t/u/v bit-identical.
"""
import os

import numpy as np
import pytest

from chameleonrt_amd import core, scenes
from chameleonrt_amd.render_hip import RenderHIP
from tests.parity import probe_rays

pytestmark = pytest.mark.gpu

SCENES = {
    "cornell": lambda: scenes.cornell(),
    "grove_two_level": lambda: scenes.instanced_grove(),
    "sponza_small": lambda: scenes.sponza_like(detail=0.02, tex_size=32),
    "single_triangle": lambda: _single_triangle(),
}


def _single_triangle():
    from chameleonrt_amd.scene import Camera, Geometry, Instance, Mesh, ParameterizedMesh, Scene, disney_material, obj_default_light
    g = Geometry(np.array([[-1, 0, 0], [1, 0, 0], [0, 1.5, 0]], np.float32), np.array([[0, 1, 2]], np.uint32), None)
    m = np.eye(4, dtype=np.float32)
    m[:3, 3] = [0.2, -0.1, -1.0]  # a single NON-identity instance: ray is transformed once
    return Scene(meshes=[Mesh([g])], parameterized_meshes=[ParameterizedMesh(0, [0])],
                 instances=[Instance(m.T.reshape(16), 0)], materials=[disney_material()],
                 lights=[obj_default_light()],
                 cameras=[Camera(np.array([0, 0.5, 3], np.float32), np.zeros(3, np.float32),
                                 np.array([0, 1, 0], np.float32), 50.0)])


@pytest.fixture(scope="module", params=list(SCENES))
def pair(request, oracle, hip_lib):
    sc = SCENES[request.param]()
    r = RenderHIP(flags=core.FLAG_COUNTERS)
    r.initialize(64, 64)
    if request.param == "grove_two_level":  # the top-level tree over instances; the world tree: tests/test_gpu_world_tree.py
        os.environ["CRT_HIP_LEVELS"] = "two"
    try:
        r.set_scene(sc)
    finally:
        os.environ.pop("CRT_HIP_LEVELS", None)
    assert r.bvh()["levels"] == (1 if request.param == "grove_two_level" else 0)
    yield r, oracle.OracleScene(sc), sc
    r.close()


def test_closest_hit_matches_brute_force(pair):
    r, o, sc = pair
    org, dirs = probe_rays(sc, 30000, seed=1)
    g = r.trace(org, dirs, 0.0, 1e20, closest=True)
    c = o.trace(org, dirs, 0.0, 1e20, closest=True, brute_force=True)
    for k in ("inst", "geom", "prim"):
        assert np.array_equal(g[k], c[k]), k
    hit = c["inst"] >= 0
    assert hit.sum() > 100
    for k in ("t", "u", "v"):
        assert np.array_equal(g[k][hit].view(np.uint32), c[k][hit].view(np.uint32)), k


def _same_as(g, c, with_uv=True):
    for k in ("inst", "geom", "prim"):
        assert np.array_equal(g[k], c[k]), k
    hit = c["inst"] >= 0
    for k in ("t", "u", "v") if with_uv else ("t",):
        assert np.array_equal(g[k][hit].view(np.uint32), c[k][hit].view(np.uint32)), k
    return hit


def test_production_kernels_match_brute_force(pair):
    """The kernels a FRAME launches -- k_trace_closest / k_trace_shadow without counters, with ClosestSource /
    ShadowSource, fed through the frame's queue records and read back from its hit records / radiance buffer
    (crt_hip_trace_rays with CRT_HIP_TRACE_PRODUCTION) -- give the brute-force answer bit for bit, like the
    instrumented diagnostic instantiation the other tests of this file run: primary rays (tnear = 0), secondary
    rays leaving surfaces (tnear = EPSILON), occlusion over finite segments."""
    r, o, sc = pair
    org, dirs = probe_rays(sc, 30000, seed=1)
    c = o.trace(org, dirs, 0.0, 1e20, closest=True, brute_force=True)
    hit = _same_as(r.trace(org, dirs, 0.0, 1e20, closest=True, production=True), c)
    assert hit.sum() > 100
    p = org[hit] + c["t"][hit, None] * dirs[hit]
    d2 = np.random.default_rng(17).normal(size=p.shape).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    _same_as(r.trace(p, d2, 1e-4, 1e20, closest=True, production=True), o.trace(p, d2, 1e-4, 1e20, closest=True, brute_force=True))
    tmax = np.random.default_rng(18).random(len(p)).astype(np.float32) * 10
    g = r.trace(p, d2, 1e-4, tmax, closest=False, production=True)
    assert np.array_equal(g["t"], o.trace(p, d2, 1e-4, tmax, closest=False, brute_force=True)["t"])


def test_tnear_epsilon_and_finite_tfar(pair):
    r, o, sc = pair
    org, dirs = probe_rays(sc, 20000, seed=2)
    tmax = np.random.default_rng(3).random(len(org)).astype(np.float32) * 8
    g = r.trace(org, dirs, 1e-4, tmax, closest=True)
    c = o.trace(org, dirs, 1e-4, tmax, closest=True, brute_force=True)
    assert np.array_equal(g["prim"], c["prim"]) and np.array_equal(g["inst"], c["inst"])
    hit = c["inst"] >= 0
    assert np.array_equal(g["t"][hit].view(np.uint32), c["t"][hit].view(np.uint32))


def test_occlusion_matches_brute_force(pair):
    r, o, sc = pair
    org, dirs = probe_rays(sc, 30000, seed=4)
    tmax = np.random.default_rng(5).random(len(org)).astype(np.float32) * 10
    g = r.trace(org, dirs, 1e-4, tmax, closest=False)
    c = o.trace(org, dirs, 1e-4, tmax, closest=False, brute_force=True)
    assert np.array_equal(g["t"], c["t"])


def test_secondary_rays_from_surfaces(pair):
    """Rays leaving surfaces at grazing angles with tnear = EPSILON (self-intersection zone)."""
    r, o, sc = pair
    org, dirs = probe_rays(sc, 20000, seed=6)
    first = o.trace(org, dirs, 0.0, 1e20, closest=True, brute_force=False)
    hit = first["inst"] >= 0
    p = org[hit] + first["t"][hit, None] * dirs[hit]
    rng = np.random.default_rng(7)
    d2 = rng.normal(size=p.shape).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    g = r.trace(p, d2, 1e-4, 1e20, closest=True)
    c = o.trace(p, d2, 1e-4, 1e20, closest=True, brute_force=True)
    assert np.array_equal(g["prim"], c["prim"]) and np.array_equal(g["inst"], c["inst"])
    h2 = c["inst"] >= 0
    assert np.array_equal(g["t"][h2].view(np.uint32), c["t"][h2].view(np.uint32))


def test_axis_parallel_rays(pair):
    """Rays with exactly zero direction components (1/d = inf): they occur about once per 2e7 rays
    in a frame, must give the brute-force answer, and must not disable box culling (a regression
    here made single rays walk 10^4 nodes)."""
    r, o, sc = pair
    org, _ = probe_rays(sc, 6000, seed=9)
    rng = np.random.default_rng(10)
    dirs = rng.normal(size=org.shape).astype(np.float32)
    dirs[:2000, 0] = 0.0
    dirs[2000:4000, 1] = 0.0
    dirs[4000:, 2] = 0.0
    dirs[::7, 1] = 0.0  # some with two zero components
    dirs[dirs.sum(axis=1) == 0] = [1, 0, 0]
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    g = r.trace(org, dirs, 0.0, 1e20, closest=True)
    c = o.trace(org, dirs, 0.0, 1e20, closest=True, brute_force=True)
    assert np.array_equal(g["prim"], c["prim"]) and np.array_equal(g["inst"], c["inst"])
    hit = c["inst"] >= 0
    assert np.array_equal(g["t"][hit].view(np.uint32), c["t"][hit].view(np.uint32))
    n_nodes, n_tris = r.bvh()["nodes"].shape[0], r.bvh()["tris"].shape[0]
    assert g["stats"].closest_nodes / len(org) < max(64.0, 0.05 * n_nodes), "box culling is off for axis-parallel rays"


def test_counters_match_oracle_walk_of_the_same_bvh(pair, oracle):
    """The instrumented kernels' node/triangle counts (the roofline input) equal the oracle walking
    the product's own BVH arrays with the documented visit rule -- closest-hit and occlusion rays,
    single-level scenes, a single non-identity instance, and the two-level (TLAS -> instance frame ->
    BLAS, sentinel on the stack) walk. The walker's hits must also be the kernel's hits."""
    r, _, sc = pair
    bvh = r.bvh()
    org, dirs = probe_rays(sc, 5000, seed=8)
    g = r.trace(org, dirs, 0.0, 1e20, closest=True)
    w = oracle.walk_product_bvh(bvh, org, dirs, 0.0, 1e20, closest=True)
    assert (g["stats"].closest_nodes, g["stats"].closest_tris, g["stats"].closest_slots) == (w["nodes"], w["tris"], w["slots"])
    for k in ("inst", "geom", "prim"):
        assert np.array_equal(g[k], w[k]), k
    hit = w["inst"] >= 0
    assert np.array_equal(g["t"][hit].view(np.uint32), w["t"][hit].view(np.uint32))
    assert w["max_stack"] <= bvh["stack_need"]
    tmax = np.random.default_rng(11).random(len(org)).astype(np.float32) * 10
    g = r.trace(org, dirs, 1e-4, tmax, closest=False)
    w = oracle.walk_product_bvh(bvh, org, dirs, 1e-4, tmax, closest=False)
    assert (g["stats"].shadow_nodes, g["stats"].shadow_tris, g["stats"].shadow_slots) == (w["nodes"], w["tris"], w["slots"])
    assert np.array_equal(g["t"], w["t"])
