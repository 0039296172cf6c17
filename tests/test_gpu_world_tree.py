"""GPU: the WORLD TREE of an instanced scene (CRT_HIP_LEVELS=world; crt_types.h LEVELS_WORLD_TREE, traverse.h INST_TRIS).

One tree in world space over per-instance copies of the triangle records; a lane transforms its ray into an
instance's object space only to test a triangle of it. Required here: the kernels' hits equal brute force over the
instances bit for bit (ids, t, u, v, occlusion), their node / triangle counts equal the oracle's walk of the same
arrays, and a frame rendered from the world tree equals the frame rendered from the two-level structure bit for
bit (same hits -> same shading arithmetic -> same accumulated radiance, ray counts and RGBA8).
"""
import numpy as np
import pytest

from chameleonrt_amd import core, scenes
from chameleonrt_amd.render_hip import RenderHIP
from tests.parity import awkward_instances, camera_of, probe_rays, slot_triangles

pytestmark = pytest.mark.gpu

SCENES = {
    "grove": (lambda: scenes.instanced_grove(), 320, 200),
    "sanmiguel_small_instanced": (lambda: scenes.sanmiguel_like(detail=0.02, tex_size=64, n_trees=100, leaves_per_tree=300,
                                                                n_instanced=64, glass=True, spp=2), 320, 180),
    # mirrored / non-uniformly scaled / coinciding instances, a shared mesh under two material tables (tests/parity.py)
    "awkward_instances": (awkward_instances, 256, 160),
}


def _renderer(sc, levels, monkeypatch, w, h):
    monkeypatch.setenv("CRT_HIP_LEVELS", levels)
    r = RenderHIP(flags=core.FLAG_COUNTERS)
    r.initialize(w, h)
    r.set_scene(sc)
    monkeypatch.delenv("CRT_HIP_LEVELS")
    return r


@pytest.mark.parametrize("name", list(SCENES))
def test_world_tree_hits_counters_and_frames(name, oracle, hip_lib, monkeypatch):
    gen, w, h = SCENES[name]
    sc = gen()
    r = _renderer(sc, "world", monkeypatch, w, h)
    bvh = r.bvh()
    assert bvh["levels"] == 2 and slot_triangles(bvh).sum() == sc.total_tris()
    o = oracle.OracleScene(sc)
    org, dirs = probe_rays(sc, 30000, seed=41)
    g = r.trace(org, dirs, 0.0, 1e20, closest=True)
    c = o.trace(org, dirs, 0.0, 1e20, closest=True, brute_force=True)
    for k in ("inst", "geom", "prim"):
        assert np.array_equal(g[k], c[k]), k
    hit = c["inst"] >= 0
    assert (c["inst"][hit] == 0).any() and (c["inst"][hit] > 0).any()
    if name == "awkward_instances":
        assert set(np.unique(c["inst"][hit])) == {0, 1, 2, 3, 4}  # of the coinciding pair (4, 5) the lower id wins every tie
    for k in ("t", "u", "v"):
        assert np.array_equal(g[k][hit].view(np.uint32), c[k][hit].view(np.uint32)), k
    wk = oracle.walk_product_bvh(bvh, org, dirs, 0.0, 1e20, closest=True)
    assert (g["stats"].closest_nodes, g["stats"].closest_tris) == (wk["nodes"], wk["tris"])
    tmax = np.random.default_rng(42).random(len(org)).astype(np.float32) * 10
    g = r.trace(org, dirs, 1e-4, tmax, closest=False)
    c = o.trace(org, dirs, 1e-4, tmax, closest=False, brute_force=True)
    assert np.array_equal(g["t"], c["t"])
    wk = oracle.walk_product_bvh(bvh, org, dirs, 1e-4, tmax, closest=False)
    assert (g["stats"].shadow_nodes, g["stats"].shadow_tris) == (wk["nodes"], wk["tris"])
    # frames: world tree == two-level structure, bit for bit
    e, d, u, fov = camera_of(sc)
    frames = {}
    for levels, rr in (("world", r), ("two", _renderer(sc, "two", monkeypatch, w, h))):
        for f in range(2):
            rr.render(e, d, u, fov, f == 0, True)
        frames[levels] = (rr.accum().copy(), rr.ray_counts().copy(), rr.img.copy())
        rr.close()
    a, b = frames["world"], frames["two"]
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), "accumulated radiance"
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert np.isfinite(a[0]).any() and a[1].sum() > 0
