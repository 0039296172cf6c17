"""CPU: seeded random triangle soups through the host builder, the node packer and the oracle's walker of the packed arrays:
every hit as brute force finds it, for closest-hit and for occlusion rays.

The fixed scenes of test_prepared_scene.py are well-behaved stand-ins; these are not: coordinates from 1e-3 to 1e4, soups that
are flat on an axis (a frame axis with no extent), needle and zero-area triangles, clusters far apart (node scales from one grid
unit to the whole frame inside one tree), a handful to thousands of triangles. Role in the reference: rtcCommitScene +
rtcIntersect / rtcOccluded (embree_utils.cpp:63-76, render_embree.ispc:236-249), which the product's tree and visit rule stand
in for.
"""
import numpy as np
import pytest

from chameleonrt_amd.render_hip import PreparedScene
from chameleonrt_amd.scene import Camera, Geometry, Instance, Mesh, ParameterizedMesh, Scene, disney_material, obj_default_light

F = np.float32


def _soup(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([3, 17, 200, 1500, 6000]))
    scale = float(10.0 ** rng.uniform(-3, 4))
    kind = seed % 5
    centres = rng.normal(size=(n, 1, 3))
    if kind == 1:  # clusters far apart
        centres = centres * 0.01 + rng.integers(0, 4, size=(n, 1, 3)) * 50.0
    size = 10.0 ** rng.uniform(-3, 0, size=(n, 1, 1))
    p = centres + rng.normal(size=(n, 3, 3)) * size
    if kind == 2:  # flat on one axis
        p[:, :, int(rng.integers(0, 3))] = 0.25
    if kind == 3:  # needles and zero-area triangles among the rest
        p[::7, 2] = p[::7, 1]
        p[3::11, 2] = p[3::11, 0] + (p[3::11, 1] - p[3::11, 0]) * 1e-6
    p = (p * scale).astype(F)
    geom = Geometry(p.reshape(-1, 3), np.arange(3 * n, dtype=np.uint32).reshape(-1, 3), None)
    sc = Scene(meshes=[Mesh([geom])], parameterized_meshes=[ParameterizedMesh(0, [0])], instances=[Instance(np.eye(4, dtype=F).reshape(16), 0)],
               materials=[disney_material()], lights=[obj_default_light()],
               cameras=[Camera(np.array([0, 0, 5 * scale], F), np.zeros(3, F), np.array([0, 1, 0], F), 50.0)])
    lo, hi = p.reshape(-1, 3).min(axis=0), p.reshape(-1, 3).max(axis=0)
    m = 4000
    # rays between random points of the (slightly enlarged) bounding box, and rays starting ON triangles (the bounce-ray case)
    a = lo + (hi - lo + 1e-3 * scale) * (rng.random((m, 3)) * 1.4 - 0.2)
    b = lo + (hi - lo + 1e-3 * scale) * (rng.random((m, 3)) * 1.4 - 0.2)
    tri = p[rng.integers(0, n, size=m // 2)]
    w = rng.dirichlet(np.ones(3), size=m // 2)[:, :, None]
    a[: m // 2] = (tri * w).sum(axis=1)
    d = b - a
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)
    return sc, a.astype(F), d.astype(F), scale


@pytest.mark.parametrize("seed", range(20))
def test_random_soup_is_walked_like_brute_force(seed, oracle, monkeypatch):
    monkeypatch.delenv("CRT_HIP_LEVELS", raising=False)
    monkeypatch.delenv("CRT_BVH_SPLITS", raising=False)
    monkeypatch.setenv("CRT_BVH_REINSERT", "2" if seed % 4 == 3 else "0")
    sc, org, dirs, scale = _soup(seed)
    ps = PreparedScene(sc)
    bvh = ps.bvh()
    ps.close()
    o = oracle.OracleScene(sc)
    tnear = 1e-4
    w = oracle.walk_product_bvh(bvh, org, dirs, tnear, 1e20, closest=True)
    c = o.trace(org, dirs, tnear, 1e20, closest=True, brute_force=True)
    for k in ("inst", "geom", "prim"):
        assert np.array_equal(w[k], c[k]), (seed, k)
    hit = c["inst"] >= 0
    assert np.array_equal(w["t"][hit].view(np.uint32), c["t"][hit].view(np.uint32)), seed
    assert w["max_stack"] <= bvh["stack_need"]
    tmax = (np.random.default_rng(seed + 100).random(len(org)) * 3 * scale).astype(F)
    s = oracle.walk_product_bvh(bvh, org, dirs, tnear, tmax, closest=False)
    cs = o.trace(org, dirs, tnear, tmax, closest=False, brute_force=True)
    assert np.array_equal(s["t"], cs["t"]), seed
