"""GPU parity on the benchmark configurations AS CONFIGURED (BASELINE.json C3 / C4 / C5): full resolution, the
configuration's samples per pixel, the full triangle count, 2048^2 textures (the 1 GB tiled texel blob), 33 M paths in
one pass (98.9 % of the path capacity), and -- C5 -- the multi-pass split at the library's DEFAULT capacity.

The oracle's cost grows with pixels x spp, so it renders a fixed SAMPLE of the reference's 64x64 tiles spread over the
image (orc_render_tiles) and the same pixels of the HIP path's full frame are compared under the image tolerance of
tests/parity.py: accumulated radiance, per-pixel ray counts (REPORT_RAY_STATS) and RGBA8. bench.py repeats the check on
the sample of its own run (`parity` in the JSON line). Reference: render_embree.ispc:198-355 on BASELINE.json's configs.
"""
import os

import numpy as np
import pytest

from chameleonrt_amd import scenes
from chameleonrt_amd.render_hip import PreparedScene, RenderHIP
from tests.parity import MAX_DIVERGED, camera_of, compare_images

pytestmark = pytest.mark.gpu

# name: (workload, CRT_HIP_LEVELS or None, number of sampled tiles, expected levels, expected passes per frame[, CRT_HIP_LANES])
CASES = {
    # Sponza-like 1280x720, 4 spp: the one configuration whose frames (3.7 M paths) the library cuts for two pass lanes by
    # default (crt_core.cpp lanes_tunable) -- both cuts, forced, as configured
    "C2_one_lane": ("C2", None, 32, 0, 1, "1"),
    "C2_two_lanes": ("C2", None, 32, 0, 2, "2"),
    "C3": ("C3", None, 32, 0, 1),
    "C4": ("C4", None, 32, 2, 1),            # the library's choice for the instanced scene: a world tree
    "C4_two_level": ("C4", "two", 32, 1, 1),
    "C5": ("C5", None, 24, 2, 16),           # 3840x2160 x 64 spp = 531 M paths = 16 passes of the default 32 Mi capacity
}
_scene_cache = {}


def _scene(workload):
    gen, kw, w, h, spp = scenes.WORKLOADS[workload]
    key = (gen.__name__, tuple(sorted(kw.items())))
    if key not in _scene_cache:
        _scene_cache.clear()  # one 10 M-triangle scene at a time
        _scene_cache[key] = gen(spp=spp, **kw)
    sc = _scene_cache[key]
    sc.samples_per_pixel = spp
    return sc, w, h, spp


def _tile_sample(w, h, n):
    ntx, nty = (w + 63) // 64, (h + 63) // 64
    ntiles = ntx * nty
    stride = max(1, ntiles // n)
    return list(range(stride // 2, ntiles, stride))[:n], ntx


@pytest.mark.parametrize("case", list(CASES))
def test_full_config_tile_parity(case, oracle, hip_lib, monkeypatch):
    workload, levels, n_tiles, want_levels, want_passes = CASES[case][:5]
    lanes = CASES[case][5] if len(CASES[case]) > 5 else None
    monkeypatch.delenv("CRT_HIP_MAX_PATHS", raising=False)  # the DEFAULT path capacity
    monkeypatch.delenv("CRT_HIP_LANES", raising=False)
    if lanes:
        monkeypatch.setenv("CRT_HIP_LANES", lanes)  # read when the context is created
    sc, w, h, spp = _scene(workload)
    if levels:
        monkeypatch.setenv("CRT_HIP_LEVELS", levels)
    ps = PreparedScene(sc)
    assert ps.levels() == want_levels
    total_paths = ((w + 63) // 64) * ((h + 63) // 64) * 4096 * spp
    if not lanes:
        assert -(-total_paths // (32 << 20)) == want_passes
    r = RenderHIP()
    r.initialize(w, h)
    r.set_prepared_scene(ps)
    ps.close()
    e, d, u, fovy = camera_of(sc)
    tiles, ntx = _tile_sample(w, h, n_tiles)
    mask = np.zeros((h, w), bool)
    for t in tiles:
        mask[(t // ntx) * 64:(t // ntx) * 64 + 64, (t % ntx) * 64:(t % ntx) * 64 + 64] = True
    o = oracle.OracleRenderer(sc, w, h)
    try:
        st = r.render(e, d, u, fovy, True, True)
        o.render_tiles(e, d, u, fovy, True, tiles)
        assert st.rays > 2 * w * h * spp  # a real frame: at least a closest-hit and an occlusion ray per path on average
        assert st.passes == want_passes and st.pass_lanes == (int(lanes) if lanes else 1)
        g, c = r.accum()[mask][:, None, :], o.accum()[mask][:, None, :]
        diverged, mean_rel = compare_images(g, c)
        print(f"\n{case}: {len(tiles)} tiles, {int(mask.sum())} pixels x {spp} spp: {diverged:.6f} diverged, mean rel {mean_rel:.2e}, "
              f"{st.rays} rays in {st.render_time_ms:.1f} ms")
        assert diverged <= MAX_DIVERGED, f"{diverged:.5f} of the sampled pixels diverged"
        assert mean_rel <= 1e-4, f"mean relative error {mean_rel:.3g}"
        gc, cc = r.ray_counts()[mask].astype(np.int64), o.ray_counts()[mask].astype(np.int64)
        # a pixel's count differs when ONE of its spp paths diverged: spp times likelier than a diverged pixel mean
        assert (gc != cc).mean() <= MAX_DIVERGED * max(1, spp // 4)
        assert abs(int(gc.sum()) - int(cc.sum())) <= max(16, int(2 * MAX_DIVERGED * cc.sum()))
        g8 = r.img.view(np.uint8).reshape(h, w, 4)[mask].astype(int)
        c8 = o.framebuffer().view(np.uint8).reshape(h, w, 4)[mask].astype(int)
        assert (np.abs(g8 - c8) > 1).any(axis=1).mean() <= 2 * MAX_DIVERGED
        assert (g8[:, 3] == 255).all()
    finally:
        r.close()
