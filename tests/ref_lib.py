"""ctypes binding of oracle/_ref/libref.so: the REFERENCE's own per-pixel kernel
(backends/embree_sycl/render_embree_kernel.inl) compiled from /root/reference by `make -C oracle ref`.
TEST INFRASTRUCTURE ONLY. Exists only where the reference tree does (the development container);
everything that must also run on the GPU box uses the golden vectors generated from it
(tests/golden/ref_*.npz, tests/golden/make_ref_golden.py)."""
import ctypes as C
import os

import numpy as np

from chameleonrt_amd.scene import PackedScene, Scene, SceneDesc

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(_ROOT, "oracle", "_ref", "libref.so")
_LIB = None

# the frames the golden vectors hold: (name, scene factory kwargs, width, height, frames)
GOLDEN_FRAMES = [
    ("cornell", dict(spp=2), 64, 48, 2),
    ("sponza_like", dict(detail=0.02, tex_size=32), 96, 64, 2),
    ("instanced_grove", dict(), 96, 64, 2),
]


def available() -> bool:
    return os.path.exists(PATH)


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(PATH)
        fp = C.POINTER(C.c_float)
        L.ref_render.argtypes = [C.POINTER(SceneDesc), C.c_int, C.c_int, fp, fp, fp, C.c_float, C.c_int, fp,
                                 C.POINTER(C.c_uint8), C.POINTER(C.c_uint16)]
        _LIB = L
    return _LIB


def render(scene: Scene, width: int, height: int, pos, dir, up, fovy: float, frames: int):
    """Accumulate `frames` frames with the reference kernel. Returns (accum HxWx3 f32, framebuffer
    HxWx4 u8, ray_stats HxW u16 of the last frame)."""
    fp = C.POINTER(C.c_float)
    packed = PackedScene(scene)
    acc = np.zeros((height, width, 3), np.float32)
    fb = np.zeros((height, width, 4), np.uint8)
    rs = np.zeros((height, width), np.uint16)
    a = [np.ascontiguousarray(x, np.float32) for x in (pos, dir, up)]
    rc = lib().ref_render(packed.ptr(), width, height, *[x.ctypes.data_as(fp) for x in a], float(fovy), frames,
                          acc.ctypes.data_as(fp), fb.ctypes.data_as(C.POINTER(C.c_uint8)),
                          rs.ctypes.data_as(C.POINTER(C.c_uint16)))
    if rc != 0:
        raise RuntimeError(f"ref_render failed: {rc}")
    return acc, fb, rs
