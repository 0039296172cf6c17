"""GPU: kernels on a tree whose leaves reference triangles more than once (CRT_BVH_SPLITS, chameleonrt_amd/csrc/presplit.h).

The same hit reached through two copies of a leaf slot must count once: exact ties in t are the RULE on such a tree, not
the exception, so this is the test of the kernels' lexicographic tie rule (traverse.h tie_break) at scale -- production
instantiations, against brute force, bit for bit -- and of a frame rendered on it against the oracle (whose own BVH has
no splits). CPU half: tests/test_presplit.py."""
import numpy as np
import pytest

from chameleonrt_amd.render_hip import RenderHIP
from tests.parity import MAX_DIVERGED, camera_of, compare_images, probe_rays
from tests.test_presplit import _beams_and_confetti

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("levels", [None, "two", "world"], ids=["one_instance", "two_level", "world_tree"])
def test_kernels_on_a_split_tree(levels, oracle, hip_lib, monkeypatch):
    sc = _beams_and_confetti(instanced=levels is not None)
    sc.samples_per_pixel = 2
    monkeypatch.setenv("CRT_BVH_SPLITS", "1.0")
    if levels:
        monkeypatch.setenv("CRT_HIP_LEVELS", levels)
    else:
        monkeypatch.delenv("CRT_HIP_LEVELS", raising=False)
    w, h = 320, 200
    r = RenderHIP()
    r.initialize(w, h)
    r.set_scene(sc)
    n_slots = r.bvh()["tris"].shape[0]
    assert n_slots > 1.9 * (150 * (2 if levels else 1) + 12000), "the beams were not split"
    o = oracle.OracleScene(sc)
    org, dirs = probe_rays(sc, 60000, seed=13, spread=0.6)
    c = o.trace(org, dirs, 0.0, 1e20, closest=True, brute_force=True)
    for production in (False, True):
        g = r.trace(org, dirs, 0.0, 1e20, closest=True, production=production)
        hit = c["inst"] >= 0
        assert hit.sum() > 3000
        for k in ("geom", "prim") + (("inst",) if not production or levels else ()):
            assert np.array_equal(g[k], c[k]), (k, production)
        for k in ("t", "u", "v"):
            assert np.array_equal(g[k][hit].view(np.uint32), c[k][hit].view(np.uint32)), (k, production)
    tmax = np.full(len(org), 25.0, np.float32)
    cs = o.trace(org, dirs, 1e-4, tmax, closest=False, brute_force=True)
    for production in (False, True):
        assert np.array_equal(r.trace(org, dirs, 1e-4, tmax, closest=False, production=production)["t"], cs["t"])
    e, d, u, fovy = camera_of(sc)
    orr = oracle.OracleRenderer(sc, w, h)
    for f in range(2):
        st = r.render(e, d, u, fovy, f == 0, True)
        ost = orr.render(e, d, u, fovy, f == 0)
    diverged, mean_rel = compare_images(r.accum(), orr.accum())
    assert diverged <= MAX_DIVERGED and mean_rel <= 1e-4, (diverged, mean_rel)
    assert abs(int(st.rays) - int(ost.rays)) <= max(16, int(2 * MAX_DIVERGED * ost.rays))
    r.close()
