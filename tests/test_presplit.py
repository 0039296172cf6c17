"""CPU: spatial pre-splitting of loosely boxed leaf items (chameleonrt_amd/csrc/presplit.h, opt-in CRT_BVH_SPLITS=<fraction>).

A triangle that several leaves reference must still be found exactly as brute force finds it -- the closest-hit rule is the
lexicographic minimum of (t, instance, geomID, primID), so the same hit reached through two leaf-slot copies is one hit --
and in the textbook case (long diagonal beams through a cloud of small triangles) the tree must get cheaper to walk. Checked
with the oracle's walker of the product's arrays (what the kernels must equal, tests/test_gpu_traversal.py), in the
structures the library builds: one instance, two-level, world tree with transformed and padded instances.
Role in the reference: rtcCommitScene (embree_utils.cpp:63-76); the reference builds at Embree's default quality (no splits).
"""
import numpy as np
import pytest

from chameleonrt_amd.render_hip import PreparedScene
from chameleonrt_amd.scene import Camera, Geometry, Instance, Mesh, ParameterizedMesh, Scene, disney_material, obj_default_light
from tests.parity import probe_rays, slot_triangles

F = np.float32


def _beams_and_confetti(seed=5, n_beams=150, n_small=12000, instanced=False):
    rng = np.random.default_rng(seed)
    # long thin quads (two triangles sharing an edge) at random orientations: boxes ~100x their own area
    c = (rng.random((n_beams, 3)) - 0.5) * 16
    d = rng.normal(size=(n_beams, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    w = np.cross(d, rng.normal(size=(n_beams, 3)))
    w /= np.linalg.norm(w, axis=1, keepdims=True)
    L, W = 9.0, 0.04
    quad = np.stack([c - L * d - W * w, c + L * d - W * w, c + L * d + W * w, c - L * d + W * w], axis=1)  # (n, 4, 3)
    beams = Geometry(quad.reshape(-1, 3).astype(F), (np.arange(n_beams)[:, None, None] * 4 + np.array([[0, 1, 2], [0, 2, 3]])).reshape(-1, 3).astype(np.uint32), None)
    p = (rng.random((n_small, 1, 3)) - 0.5) * 18 + rng.normal(size=(n_small, 3, 3)) * 0.08
    confetti = Geometry(p.reshape(-1, 3).astype(F), np.arange(3 * n_small, dtype=np.uint32).reshape(-1, 3), None)
    eye = np.eye(4, dtype=F)
    meshes = [Mesh([beams, confetti])]
    pms = [ParameterizedMesh(0, [0, 0])]
    insts = [Instance(eye.T.reshape(16), 0)]
    if instanced:  # the same beams again as a rotated, non-uniformly scaled, shifted instance: padded world boxes
        a = 0.7
        m = np.array([[np.cos(a) * 1.3, 0, np.sin(a), 3.0], [0, 0.8, 0, -1.0], [-np.sin(a) * 1.3, 0, np.cos(a), 2.0], [0, 0, 0, 1]], F)
        meshes.append(Mesh([beams]))
        pms.append(ParameterizedMesh(1, [0]))
        insts.append(Instance(m.T.reshape(16), 1))
    return Scene(meshes=meshes, parameterized_meshes=pms, instances=insts, materials=[disney_material()], lights=[obj_default_light()],
                 cameras=[Camera(np.array([0, 2, 20], F), np.zeros(3, F), np.array([0, 1, 0], F), 50.0)])


def _walk(oracle, sc, monkeypatch, splits, levels=None):
    monkeypatch.setenv("CRT_BVH_SPLITS", splits)
    if levels:
        monkeypatch.setenv("CRT_HIP_LEVELS", levels)
    else:
        monkeypatch.delenv("CRT_HIP_LEVELS", raising=False)
    ps = PreparedScene(sc)
    bvh = ps.bvh()
    ps.close()
    org, dirs = probe_rays(sc, 20000, seed=11, spread=0.6)
    w = oracle.walk_product_bvh(bvh, org, dirs, 0.0, 1e20, closest=True)
    # occlusion segments towards a point above the scene
    tmax = np.full(len(org), 25.0, F)
    s = oracle.walk_product_bvh(bvh, org, dirs, 1e-4, tmax, closest=False)
    return bvh, org, dirs, tmax, w, s


@pytest.mark.parametrize("structure", ["one_instance", "two_level", "world_tree"])
def test_split_tree_finds_what_brute_force_finds_and_is_cheaper(structure, oracle, monkeypatch):
    sc = _beams_and_confetti(instanced=structure != "one_instance")
    levels = {"one_instance": None, "two_level": "two", "world_tree": "world"}[structure]
    b0, org, dirs, tmax, w0, s0 = _walk(oracle, sc, monkeypatch, "0", levels)
    b1, _, _, _, w1, s1 = _walk(oracle, sc, monkeypatch, "1.0", levels)
    n0, n1 = b0["tris"].shape[0], b1["tris"].shape[0]
    assert n1 > n0 and n1 <= 2 * n0 + 1, "the budget bounds the extra leaf slots"
    assert slot_triangles(b0).sum() == sc.total_tris() if structure != "two_level" else True
    o = oracle.OracleScene(sc)
    c = o.trace(org, dirs, 0.0, 1e20, closest=True, brute_force=True)
    for w in (w0, w1):
        for k in ("inst", "geom", "prim"):
            assert np.array_equal(w[k], c[k]), (structure, k)
        hit = c["inst"] >= 0
        assert hit.sum() > 1000
        assert np.array_equal(w["t"][hit].view(np.uint32), c["t"][hit].view(np.uint32)), structure
    cs = o.trace(org, dirs, 1e-4, tmax, closest=False, brute_force=True)
    assert np.array_equal(s0["t"], cs["t"]) and np.array_equal(s1["t"], cs["t"])
    lines0, lines1 = w0["nodes"] + w0["slots"], w1["nodes"] + w1["slots"]
    print(f"\n{structure}: {n0} -> {n1} leaf slots, lines per closest-hit ray {lines0 / len(org):.2f} -> {lines1 / len(org):.2f}, "
          f"per occlusion ray {(s0['nodes'] + s0['slots']) / len(org):.2f} -> {(s1['nodes'] + s1['slots']) / len(org):.2f}")
    assert lines1 < 0.8 * lines0, "the textbook case for spatial splits did not get cheaper"


def test_off_by_default_and_nothing_to_split_in_axis_aligned_quads(oracle, monkeypatch):
    from chameleonrt_amd import scenes
    sc = scenes.cornell()
    monkeypatch.delenv("CRT_BVH_SPLITS", raising=False)
    ps = PreparedScene(sc)
    n_default = ps.bvh()["tris"].shape[0]
    ps.close()
    monkeypatch.setenv("CRT_BVH_SPLITS", "1.0")
    ps = PreparedScene(sc)
    n_split = ps.bvh()["tris"].shape[0]
    ps.close()
    assert n_split == n_default  # walls and boxes are axis-aligned quads: their boxes are their geometry
