"""GPU parity: the traversal kernels on the seeded triangle soups of tests/test_packed_tree_fuzz.py (flat on an axis, needles and
zero-area triangles, clusters far apart, coordinates from 1e-3 to 1e4, re-inserted trees): ids and t / u / v bit-identical to brute
force for closest-hit rays, t for occlusion rays, for the instrumented AND the production instantiations, and the instrumented
kernels' node / slot counts equal to the oracle's walk of the same packed arrays.
"""
import numpy as np
import pytest

from chameleonrt_amd import core
from chameleonrt_amd.render_hip import RenderHIP
from tests.test_packed_tree_fuzz import _soup

pytestmark = pytest.mark.gpu

F = np.float32


@pytest.mark.parametrize("seed", range(10))
def test_kernels_on_a_random_soup(seed, oracle, hip_lib, monkeypatch):
    monkeypatch.delenv("CRT_HIP_LEVELS", raising=False)
    monkeypatch.delenv("CRT_BVH_SPLITS", raising=False)
    monkeypatch.setenv("CRT_BVH_REINSERT", "2" if seed % 4 == 3 else "0")
    sc, org, dirs, scale = _soup(seed)
    r = RenderHIP(flags=core.FLAG_COUNTERS)
    try:
        r.initialize(64, 64)
        r.set_scene(sc)
        bvh = r.bvh()
        o = oracle.OracleScene(sc)
        tnear = 1e-4
        c = o.trace(org, dirs, tnear, 1e20, closest=True, brute_force=True)
        w = oracle.walk_product_bvh(bvh, org, dirs, tnear, 1e20, closest=True)
        hit = c["inst"] >= 0
        for production in (False, True):
            g = r.trace(org, dirs, tnear, 1e20, closest=True, production=production)
            for k in ("inst", "geom", "prim"):
                assert np.array_equal(g[k], c[k]), (seed, production, k)
            for k in ("t", "u", "v"):
                assert np.array_equal(g[k][hit].view(np.uint32), c[k][hit].view(np.uint32)), (seed, production, k)
            if not production:
                assert (g["stats"].closest_nodes, g["stats"].closest_slots) == (w["nodes"], w["slots"]), seed
        tmax = (np.random.default_rng(seed + 100).random(len(org)) * 3 * scale).astype(F)
        cs = o.trace(org, dirs, tnear, tmax, closest=False, brute_force=True)
        ws = oracle.walk_product_bvh(bvh, org, dirs, tnear, tmax, closest=False)
        g = r.trace(org, dirs, tnear, tmax, closest=False)
        assert np.array_equal(g["t"], cs["t"]), seed
        assert (g["stats"].shadow_nodes, g["stats"].shadow_slots) == (ws["nodes"], ws["slots"]), seed
        assert np.array_equal(r.trace(org, dirs, tnear, tmax, closest=False, production=True)["t"], cs["t"]), seed
    finally:
        r.close()
