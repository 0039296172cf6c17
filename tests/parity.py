"""Shared parity helpers for the GPU tests (HIP core vs CPU oracle)."""
import numpy as np

from chameleonrt_amd.camera import camera_of, look_at  # noqa: F401  (camera_of re-exported for the tests)

# Image tolerance (SURVEY §8c / north star "within a stated float tolerance"): the two
# implementations evaluate the same expressions in the same order and differ only in libm
# transcendentals (a few ulp) and in the order the per-sample radiance sums are added, so a
# pixel must agree to  |gpu - cpu| <= 1e-4 + 1e-3 * |cpu|.  A path whose ulp-level difference
# flips a discrete decision (shadow edge, Russian roulette, checker boundary) diverges
# completely; such pixels are counted and must stay below 0.1 % of the image.
ABS_TOL, REL_TOL, MAX_DIVERGED = 1e-4, 1e-3, 1e-3


def probe_rays(scene, n, seed=0, spread=0.3):
    """Half camera-cone rays, half uniformly random directions from points around the scene."""
    rng = np.random.default_rng(seed)
    e, d, u, _ = camera_of(scene)
    org = np.tile(e, (n, 1)).astype(np.float32)
    dirs = rng.normal(size=(n, 3)).astype(np.float32)
    dirs[: n // 2] = d + spread * rng.normal(size=(n // 2, 3)).astype(np.float32)
    org[n // 2:] += (rng.normal(size=(n - n // 2, 3)) * 0.5).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    return org, dirs.astype(np.float32)


def compare_images(gpu, cpu):
    """Returns (fraction of diverged pixels, mean relative error over agreeing pixels)."""
    # the reference's arithmetic produces inf/NaN radiance in a few corner cases (quirks Q2, Q6:
    # zero-length normalisations, 1/(1-1)); a pixel that is non-finite in BOTH images agrees
    nf_g, nf_c = ~np.isfinite(gpu).all(axis=2), ~np.isfinite(cpu).all(axis=2)
    both_nf = nf_g & nf_c
    with np.errstate(invalid="ignore"):
        err = np.abs(gpu - cpu)
        tol = ABS_TOL + REL_TOL * np.abs(cpu)
        bad = ((err > tol).any(axis=2) | (nf_g != nf_c)) & ~both_nf
    good = ~bad & ~both_nf
    mean_rel = float(err[good].mean() / max(1e-12, np.abs(cpu[good]).mean()))
    return float(bad.mean()), mean_rel
