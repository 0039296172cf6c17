"""CPU: insertion-based re-optimisation of the host-built binary tree (chameleonrt_amd/csrc/bvh_builder.cpp Reinserter, opt-in
CRT_BVH_REINSERT=<passes>; Bittner, Hapala, Havran 2013).

Subtrees are taken out and put back where they cost the least summed surface area; leaves keep their item ranges. The result
must be a valid tree over exactly the same leaf slots -- every hit as brute force finds it, in every structure the library
builds -- must not be dearer to walk than the tree it started from on a scene where a top-down build has something to repair
(clusters of small triangles interleaved with large ones), and must be the same tree whatever the number of build threads.
Role in the reference: the quality of rtcCommitScene's tree (embree_utils.cpp:63-76); priced by tools/tree_cost.py
(C4 -6.2 % line visits per ray, C3 -0.5 %).
"""
import os
import re
import subprocess

import numpy as np
import pytest

from chameleonrt_amd.render_hip import PreparedScene
from tests.parity import probe_rays, slot_triangles
from tests.test_presplit import _beams_and_confetti

F = np.float32


def _prepared(sc, monkeypatch, passes, levels=None, threads=None, piece=None):
    monkeypatch.setenv("CRT_BVH_REINSERT", passes)
    if piece:
        monkeypatch.setenv("CRT_BVH_REINSERT_PIECE", piece)
    else:
        monkeypatch.delenv("CRT_BVH_REINSERT_PIECE", raising=False)
    monkeypatch.delenv("CRT_BVH_SPLITS", raising=False)
    if levels:
        monkeypatch.setenv("CRT_HIP_LEVELS", levels)
    else:
        monkeypatch.delenv("CRT_HIP_LEVELS", raising=False)
    if threads:
        monkeypatch.setenv("CRT_HIP_BUILD_THREADS", threads)
    else:
        monkeypatch.delenv("CRT_HIP_BUILD_THREADS", raising=False)
    ps = PreparedScene(sc)
    bvh = ps.bvh()
    ps.close()
    return bvh


@pytest.mark.parametrize("structure", ["one_instance", "two_level", "world_tree"])
def test_reinserted_tree_finds_what_brute_force_finds(structure, oracle, monkeypatch):
    sc = _beams_and_confetti(instanced=structure != "one_instance")
    levels = {"one_instance": None, "two_level": "two", "world_tree": "world"}[structure]
    b0 = _prepared(sc, monkeypatch, "0", levels)
    b1 = _prepared(sc, monkeypatch, "3", levels)
    assert b1["tris"].shape[0] == b0["tris"].shape[0] and slot_triangles(b1).sum() == slot_triangles(b0).sum()
    assert not np.array_equal(b1["nodes"], b0["nodes"]), "three passes moved nothing in a scene of beams through confetti"
    org, dirs = probe_rays(sc, 20000, seed=13, spread=0.6)
    o = oracle.OracleScene(sc)
    c = o.trace(org, dirs, 0.0, 1e20, closest=True, brute_force=True)
    tmax = np.full(len(org), 25.0, F)
    cs = o.trace(org, dirs, 1e-4, tmax, closest=False, brute_force=True)
    lines = []
    for b in (b0, b1):
        w = oracle.walk_product_bvh(b, org, dirs, 0.0, 1e20, closest=True)
        for k in ("inst", "geom", "prim"):
            assert np.array_equal(w[k], c[k]), (structure, k)
        hit = c["inst"] >= 0
        assert hit.sum() > 1000 and np.array_equal(w["t"][hit].view(np.uint32), c["t"][hit].view(np.uint32)), structure
        s = oracle.walk_product_bvh(b, org, dirs, 1e-4, tmax, closest=False)
        assert np.array_equal(s["t"], cs["t"])
        assert w["max_stack"] <= b["stack_need"] and s["max_stack"] <= b["stack_need"]
        lines.append((w["nodes"] + w["slots"] + s["nodes"] + s["slots"]) / len(org))
    print(f"\n{structure}: lines per ray (closest + occlusion) {lines[0]:.2f} -> {lines[1]:.2f}")
    assert lines[1] <= 1.02 * lines[0], "the re-optimised tree is dearer to walk than the one it started from"


def test_thread_count_does_not_change_the_reinserted_tree(monkeypatch):
    sc = _beams_and_confetti()
    for piece in (None, "512"):  # one piece (the whole tree); ~50 pieces swept by several workers + the nodes above the cut
        a = _prepared(sc, monkeypatch, "2", threads="1", piece=piece)
        b = _prepared(sc, monkeypatch, "2", threads="5", piece=piece)
        assert np.array_equal(a["nodes"], b["nodes"]) and np.array_equal(a["tris"], b["tris"])


def test_tree_swept_in_pieces_finds_what_brute_force_finds(oracle, monkeypatch):
    """A pass over a large tree is cut into independent pieces (searches confined to a piece, the nodes above the cut swept
    afterwards over the whole tree): forced here on a small tree."""
    sc = _beams_and_confetti(instanced=True)
    b0 = _prepared(sc, monkeypatch, "0", "world")
    b1 = _prepared(sc, monkeypatch, "3", "world", threads="4", piece="256")
    assert b1["tris"].shape[0] == b0["tris"].shape[0] and not np.array_equal(b1["nodes"], b0["nodes"])
    org, dirs = probe_rays(sc, 20000, seed=17, spread=0.6)
    c = oracle.OracleScene(sc).trace(org, dirs, 0.0, 1e20, closest=True, brute_force=True)
    w0 = oracle.walk_product_bvh(b0, org, dirs, 0.0, 1e20, closest=True)
    w1 = oracle.walk_product_bvh(b1, org, dirs, 0.0, 1e20, closest=True)
    for k in ("inst", "geom", "prim"):
        assert np.array_equal(w1[k], c[k]), k
    hit = c["inst"] >= 0
    assert np.array_equal(w1["t"][hit].view(np.uint32), c["t"][hit].view(np.uint32)) and w1["max_stack"] <= b1["stack_need"]
    lines0, lines1 = (w0["nodes"] + w0["slots"]) / len(org), (w1["nodes"] + w1["slots"]) / len(org)
    print(f"\nworld tree swept in pieces of <= 256 nodes: lines per closest-hit ray {lines0:.2f} -> {lines1:.2f}")
    assert lines1 <= 1.02 * lines0


def test_builder_check_with_reinsertion(tmp_path):
    """The native structural check of the builder (tests/native/bvh_check.cpp: every item in exactly one leaf, boxes contain
    their children, references in range, quantised and packed boxes conservative) on a tree that went through two passes,
    and its summed-area cost against the plain build's."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "chameleonrt_amd", "csrc")
    exe = str(tmp_path / "bvh_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", csrc, os.path.join(root, "tests", "native", "bvh_check.cpp"),
                           os.path.join(csrc, "bvh_builder.cpp"), "-o", exe])
    cost = {}
    for passes in ("0", "2"):
        env = dict(os.environ, CRT_BVH_REINSERT=passes)
        p = subprocess.run([exe, "60000", "4", "11", "0"], capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0 and "errors 0" in p.stdout, p.stdout + p.stderr
        cost[passes] = float(re.search(r"sah_nodes ([0-9.]+)", p.stdout).group(1))
    print(f"\nexpected node visits per ray through the root: {cost['0']:.2f} -> {cost['2']:.2f}")
    assert cost["2"] <= cost["0"] * 1.001
