"""CPU: the packed tree as it is uploaded (RenderHIP.bvh() / PreparedScene.bvh(): 64-byte PNode records, 64-byte leaf slots)
is a bounding hierarchy of its own triangles: every vertex of every leaf slot lies inside EVERY child box on the path from
the root to its leaf, decoded the way the kernels decode it (origin + byte * 2^e grid units of the tree's 16-bit frame).

The walker tests (test_prepared_scene.py, test_gpu_traversal.py) show that rays find what brute force finds; this one checks the
arrays themselves, independently of any ray: a box that failed to contain its geometry would only show up there if a probe ray
happened to graze it. Single trees in object space and world trees (slots stay in object space and carry their instance; the
boxes are in world space, padded for transformed instances). Role in the reference: the BVH rtcCommitScene builds
(embree_utils.cpp:63-76).
"""
import numpy as np
import pytest

from chameleonrt_amd import scenes
from chameleonrt_amd.render_hip import PreparedScene
from tests.parity import node_boxes, node_refs, node_used

SCENES = {
    "cornell": (lambda: scenes.cornell(), None),
    "sponza_small": (lambda: scenes.sponza_like(detail=0.05, tex_size=32), None),
    "rungholt_small": (lambda: scenes.rungholt_like(n=96), None),
    "grove_world_tree": (lambda: scenes.instanced_grove(), "world"),
    "sanmiguel_small_world_tree": (lambda: scenes.sanmiguel_like(detail=0.01, tex_size=16, n_trees=40, leaves_per_tree=150,
                                                                 n_instanced=30, glass=True), "world"),
}


@pytest.mark.parametrize("name,reinsert", [(n, "0") for n in SCENES] + [("sponza_small", "2"), ("sanmiguel_small_world_tree", "2")])
def test_every_vertex_lies_inside_every_box_above_it(name, reinsert, monkeypatch):
    make, levels = SCENES[name]
    monkeypatch.setenv("CRT_BVH_REINSERT", reinsert)  # (2: the tree after two passes of insertion-based re-optimisation)
    monkeypatch.delenv("CRT_BVH_SPLITS", raising=False)  # (a pre-split slot's box covers only its part of the triangle)
    if levels:
        monkeypatch.setenv("CRT_HIP_LEVELS", levels)
    else:
        monkeypatch.delenv("CRT_HIP_LEVELS", raising=False)
    sc = make()
    ps = PreparedScene(sc)
    bvh = ps.bvh()
    ps.close()
    assert bvh["levels"] == (2 if levels else 0)
    nodes, slots = bvh["nodes"], bvh["tris"]
    frame = np.asarray(bvh["frame"], np.float64)
    base, step = frame[:3], frame[3:]
    boxes = node_boxes(nodes)  # (n, 4, 3, 2) in grid units
    lo = base + boxes[..., 0] * step
    hi = base + boxes[..., 1] * step
    refs, used = node_refs(nodes), node_used(nodes)
    verts = np.ascontiguousarray(slots[:, :12]).view(np.float32).astype(np.float64).reshape(-1, 4, 3)
    prim1 = np.ascontiguousarray(slots[:, 14]).view(np.uint32)
    tag = np.ascontiguousarray(slots[:, 15]).view(np.uint32)
    o2w = [np.asarray(it.transform, np.float64).reshape(4, 4).T for it in sc.instances]  # column-major in the record
    span = float(np.max(hi[used]) - np.min(lo[used]))
    eps = 1e-6 * span  # fp32 evaluation of base + q * step on the device vs float64 here
    seen = np.zeros(len(slots), bool)
    stack = [(int(bvh["root"]), np.full(3, -np.inf), np.full(3, np.inf))]
    visited_nodes = 0
    while stack:
        n, plo, phi = stack.pop()
        visited_nodes += 1
        for c in range(4):
            if not used[n, c]:
                continue
            clo, chi = np.maximum(plo, lo[n, c]), np.minimum(phi, hi[n, c])  # inside every box so far = inside their intersection
            r = int(refs[n, c])
            if r >= 0:
                stack.append((r, clo, chi))
                continue
            first, count = (~r & 0xffffffff) >> 3, ((~r) & 7) + 1
            for s in range(first, first + count):
                assert not seen[s], f"leaf slot {s} is referenced twice"
                seen[s] = True
                v = verts[s, :3] if prim1[s] == 0xffffffff else verts[s]
                if levels:
                    m = o2w[int(tag[s] >> 1)]
                    v = v @ m[:3, :3].T + m[:3, 3]
                assert (v >= clo - eps).all() and (v <= chi + eps).all(), (name, n, c, s, v, clo, chi)
    assert seen.all(), "a leaf slot is not reachable from the root"
    assert visited_nodes == len(nodes), "a node is not reachable from the root (or reachable twice)"
