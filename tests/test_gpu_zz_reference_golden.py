"""GPU parity against frames rendered by the REFERENCE's own kernel code.

tests/golden/ref_*.npz hold accumulated radiance, RGBA8 and per-pixel ray statistics produced by
backends/embree_sycl/render_embree_kernel.inl compiled from the reference tree (oracle/_ref,
tests/golden/make_ref_golden.py; the CPU oracle reproduces them bit for bit,
tests/test_oracle_pinned.py). Here the HIP path renders the same scenes, cameras and frame counts
and is held to the image tolerance of tests/parity.py:

    |hip - reference| <= 1e-4 + 1e-3 * |reference| per pixel; pixels whose path diverged (an ulp in a
    libm transcendental flips a discrete decision) <= max(0.1 %, 8 pixels) of these small frames;
    mean relative error <= 1e-4; non-finite pixels must coincide; RGBA8 within 1 LSB and ray
    statistics equal except on diverged pixels.

(This file sorts last on purpose: it is the newest test of the suite.)"""
import os

import numpy as np
import pytest

from chameleonrt_amd import scenes
from chameleonrt_amd.render_hip import RenderHIP
from tests import ref_lib
from tests.parity import MAX_DIVERGED, camera_of, compare_images

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("frame", ref_lib.GOLDEN_FRAMES, ids=[f[0] for f in ref_lib.GOLDEN_FRAMES])
def test_hip_frames_match_the_reference_kernel(frame, hip_lib):
    name, kwargs, w, h, frames = frame
    g = np.load(os.path.join(GOLDEN, f"ref_{name}.npz"))
    sc = getattr(scenes, name)(**kwargs)
    r = RenderHIP()
    r.initialize(w, h)
    r.set_scene(sc)
    e, d, u, fovy = camera_of(sc)
    try:
        for f in range(frames):
            r.render(e, d, u, fovy, f == 0, True)
        allowed = max(MAX_DIVERGED, 8.0 / (w * h))
        diverged, mean_rel = compare_images(r.accum(), g["accum"])
        assert diverged <= allowed, f"{diverged:.5f} of the pixels diverged from the reference kernel"
        assert mean_rel <= 1e-4, f"mean relative error {mean_rel:.3g}"
        counts = r.ray_counts().astype(np.int64)
        assert (counts != g["ray_stats"].astype(np.int64)).mean() <= allowed
        g8 = r.img.view(np.uint8).reshape(h, w, 4).astype(int)
        assert (g8[..., 3] == 255).all()
        assert ((np.abs(g8 - g["framebuffer"].astype(int)) > 1).any(axis=2)).mean() <= 2 * allowed
    finally:
        r.close()
