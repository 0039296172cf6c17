"""Generate tests/golden/ref_*.npz: frames rendered by the REFERENCE's own kernel code.

The reference ships no golden vectors; these are outputs of its per-pixel kernel
(backends/embree_sycl/render_embree_kernel.inl: camera ray, path loop, Disney BSDF, NEE + MIS,
Russian roulette, running mean, sRGB8, ray statistics) compiled from /root/reference into
oracle/_ref/libref.so (`make -C oracle ref`; see oracle/ref_driver.cpp for what is substituted:
Embree's two ray queries, GLM's matrix inverse and SYCL's math functions). They travel to the GPU
box, where /root/reference does not exist. Run in the development container:

    make -C oracle ref && python tests/golden/make_ref_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from chameleonrt_amd import scenes  # noqa: E402
from tests import ref_lib  # noqa: E402
from tests.parity import camera_of  # noqa: E402


def main():
    for name, kwargs, w, h, frames in ref_lib.GOLDEN_FRAMES:
        sc = getattr(scenes, name)(**kwargs)
        e, d, u, fovy = camera_of(sc)
        acc, fb, rs = ref_lib.render(sc, w, h, e, d, u, fovy, frames)
        out = os.path.join(HERE, f"ref_{name}.npz")
        np.savez_compressed(out, accum=acc, framebuffer=fb, ray_stats=rs, width=w, height=h, frames=frames)
        print(out, os.path.getsize(out), "bytes; finite pixels", float(np.isfinite(acc).all(axis=2).mean()))


if __name__ == "__main__":
    main()
