"""GPU: frames of scenes with RANDOM Disney materials (tests/parity.py random_material_grove: every parameter over its whole
range, glass on a third of the materials, 1 ... 3 random quad lights, odd framebuffer sizes) against the oracle -- which the CPU
suite holds to the reference's own kernel bit for bit on the very same scenes (tests/test_oracle_pinned.py). The image bar of
DESIGN.md section 2; the same non-finite pixels; ray counts equal except on diverged pixels; with and without the opt-in elision."""
import numpy as np
import pytest

from chameleonrt_amd import core
from chameleonrt_amd.render_hip import RenderHIP
from tests.parity import MAX_DIVERGED, camera_of, compare_images, random_material_grove

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(8))
def test_random_material_frames_match_the_oracle(seed, oracle, hip_lib):
    sc, w, h = random_material_grove(seed)
    cam = camera_of(sc)
    o = oracle.OracleRenderer(sc, w, h)
    for f in range(2):
        ost = o.render(*cam, f == 0)
    ref, ref_counts = o.accum(), o.ray_counts()
    for flags in (0, core.FLAG_ELIDE_UNUSED_SHADOW_RAYS):
        r = RenderHIP(flags=flags)
        r.initialize(w, h)
        r.set_scene(sc)
        for f in range(2):
            st = r.render(*cam, f == 0, True)
        acc, counts = r.accum(), r.ray_counts()
        r.close()
        diverged, mean_rel = compare_images(acc, ref)
        # (small frames: one diverged path is 1 / (w h) of the image, so the bar is a count here, not the 0.1 % of large frames)
        n_div = round(diverged * w * h)
        assert n_div <= max(3, MAX_DIVERGED * w * h) and mean_rel <= 1e-4, (seed, flags, n_div, mean_rel)
        # (non-finite pixels -- the reference's glass -- must be the same pixels; NaN against inf is the order of a sum of infinities)
        assert np.array_equal(~np.isfinite(acc).all(axis=2), ~np.isfinite(ref).all(axis=2)) or n_div > 0
        assert (counts != ref_counts).sum() <= 4 * max(1, n_div)
        assert abs(int(st.rays) - int(ost.rays)) <= 64 * max(1, n_div)
