"""ctypes binding of the C-ABI in include/crt_hip.h (libcrt_hip_core.so).

This is plumbing only: every call goes straight to the HIP core. There is no Python or CPU
implementation of the render path behind it -- if the shared library is missing or no GPU is
present, the functions raise.
"""
import ctypes as C
import os

import numpy as np

from .scene import SceneDesc

_HERE = os.path.dirname(os.path.abspath(__file__))
# CRT_HIP_LIB selects a tuning-variant build of the same library (tools/variants.py); CRT_HIP_SPEED=1 the fast-math
# "speed mode" build (chameleonrt_amd/build.py build_fast: what the reference's --opt=fast-math ISPC build is to its
# kernels; NOT the build any parity statement is about)
LIB_PATH = os.environ.get("CRT_HIP_LIB") or os.path.join(
    _HERE, "libcrt_hip_core_fast.so" if os.environ.get("CRT_HIP_SPEED") == "1" else "libcrt_hip_core.so")

FLAG_COUNTERS = 1
FLAG_TIMING = 2
FLAG_REFINE_IN_BACKGROUND = 8  # opt-in: set_scene returns with a quickly built tree, the full-quality one is swapped in later (include/crt_hip.h)
FLAG_ELIDE_UNUSED_SHADOW_RAYS = 4  # opt-in: occlusion rays whose result cannot reach the image are counted, not traced (include/crt_hip.h)
TRACE_PRODUCTION = 2  # crt_hip_trace_rays: run the kernels a frame launches (include/crt_hip.h)

# every symbol include/crt_hip.h declares
EXPORTS = [
    "crt_hip_abi_version", "crt_hip_device_count", "crt_hip_create", "crt_hip_destroy",
    "crt_hip_last_error", "crt_hip_name", "crt_hip_set_stream", "crt_hip_set_partition",
    "crt_hip_initialize", "crt_hip_set_scene", "crt_hip_render", "crt_hip_framebuffer",
    "crt_hip_device_framebuffer", "crt_hip_read_accum", "crt_hip_read_ray_counts", "crt_hip_frame_id", "crt_hip_tile_buffer",
    "crt_hip_assemble_tiles", "crt_hip_trace_rays", "crt_hip_kat", "crt_hip_bvh_info",
    "crt_hip_bvh_copy", "crt_hip_bvh_layout", "crt_hip_bvh_copy_instances", "crt_hip_prepare_scene",
    "crt_hip_prepared_scene_world_instance", "crt_hip_world_instance",
    "crt_hip_prepare_scene_on", "crt_hip_free_prepared_scene", "crt_hip_set_prepared_scene", "crt_hip_save_prepared_scene",
    "crt_hip_load_prepared_scene", "crt_hip_prepared_scene_info", "crt_hip_prepared_scene_copy",
    "crt_hip_child_order", "crt_hip_lds_stack_entries", "crt_hip_prepared_scene_set_spp", "crt_hip_debug_copy_queue",
    "crt_hip_render_begin", "crt_hip_render_end", "crt_hip_refine_state",
]


class RenderStats(C.Structure):
    """crt_render_stats (RenderStats of util/render_backend.h:7-10 + roofline inputs)."""
    _fields_ = [("render_time_ms", C.c_float), ("rays_per_second", C.c_float), ("rays", C.c_uint64),
                ("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("closest_ms", C.c_float),
                ("shadow_ms", C.c_float), ("shade_ms", C.c_float), ("closest_nodes", C.c_uint64),
                ("closest_tris", C.c_uint64), ("shadow_nodes", C.c_uint64), ("shadow_tris", C.c_uint64),
                # ABI 2: per path-loop iteration (MAX_PATH_DEPTH = 5)
                ("closest_rays_bounce", C.c_uint64 * 5), ("shadow_rays_bounce", C.c_uint64 * 5),
                ("closest_ms_bounce", C.c_float * 5), ("shadow_ms_bounce", C.c_float * 5), ("shade_ms_bounce", C.c_float * 5),
                ("raygen_ms", C.c_float), ("accumulate_ms", C.c_float),
                ("closest_slots", C.c_uint64), ("shadow_slots", C.c_uint64),
                ("passes", C.c_uint32), ("pass_lanes", C.c_uint32),  # ABI 3
                ("shadow_rays_elided", C.c_uint64)]  # ABI 4 (FLAG_ELIDE_UNUSED_SHADOW_RAYS)

    def as_dict(self):
        return {k: (list(getattr(self, k)) if hasattr(getattr(self, k), "__len__") else getattr(self, k)) for k, _ in self._fields_}


class CoreError(RuntimeError):
    pass


_lib = None


def load():
    """Load libcrt_hip_core.so; raises if it has not been built (python -m chameleonrt_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CoreError(f"{LIB_PATH} not found: build it with `python -m chameleonrt_amd.build` "
                        "(there is no CPU fallback for the render path)")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 and this
    # library links the system one with the same SONAME. Whichever is loaded first serves both,
    # and torch only works on top of its own copy -- so when torch is installed, let it load first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    fp, u32p, i32p, vp = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.c_void_p
    L.crt_hip_abi_version.restype = C.c_int
    L.crt_hip_device_count.restype = C.c_int
    L.crt_hip_create.restype = vp
    L.crt_hip_create.argtypes = [C.c_int, C.c_uint32]
    L.crt_hip_destroy.argtypes = [vp]
    L.crt_hip_destroy.restype = None
    L.crt_hip_last_error.restype = C.c_char_p
    L.crt_hip_last_error.argtypes = [vp]
    L.crt_hip_name.restype = C.c_char_p
    L.crt_hip_name.argtypes = [vp]
    L.crt_hip_set_stream.argtypes = [vp, vp]
    L.crt_hip_set_partition.argtypes = [vp, C.c_int, C.c_int]
    L.crt_hip_initialize.argtypes = [vp, C.c_int, C.c_int]
    L.crt_hip_set_scene.argtypes = [vp, C.POINTER(SceneDesc)]
    L.crt_hip_render.argtypes = [vp, fp, fp, fp, C.c_float, C.c_int, C.c_int, C.POINTER(RenderStats)]
    L.crt_hip_render_begin.argtypes = [vp, fp, fp, fp, C.c_float, C.c_int, C.c_int]
    L.crt_hip_render_begin.restype = C.c_int
    L.crt_hip_render_end.argtypes = [vp, C.POINTER(RenderStats)]
    L.crt_hip_render_end.restype = C.c_int
    L.crt_hip_framebuffer.restype = u32p
    L.crt_hip_framebuffer.argtypes = [vp]
    L.crt_hip_device_framebuffer.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.crt_hip_device_framebuffer.restype = C.c_int
    L.crt_hip_read_accum.argtypes = [vp, fp]
    L.crt_hip_read_ray_counts.argtypes = [vp, u32p]
    L.crt_hip_frame_id.restype = C.c_uint32
    L.crt_hip_frame_id.argtypes = [vp]
    L.crt_hip_tile_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.crt_hip_assemble_tiles.argtypes = [vp, vp, C.c_int, C.c_int]
    L.crt_hip_trace_rays.argtypes = [vp, C.c_uint64, fp, fp, fp, fp, C.c_int, fp, fp, fp, i32p, i32p, i32p,
                                     C.POINTER(RenderStats)]
    L.crt_hip_kat.argtypes = [vp, C.c_int, C.c_uint64, fp, C.c_int, fp, C.c_int]
    L.crt_hip_debug_copy_queue.argtypes = [vp, C.c_int, C.c_uint64, C.c_uint64, fp]
    L.crt_hip_debug_copy_queue.restype = C.c_int
    L.crt_hip_bvh_info.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), i32p, fp]
    L.crt_hip_bvh_copy.argtypes = [vp, vp, vp]
    L.crt_hip_bvh_layout.argtypes = [vp, i32p, u32p, u32p, u32p, i32p]
    L.crt_hip_prepared_scene_world_instance.argtypes = [vp]
    L.crt_hip_prepared_scene_world_instance.restype = C.c_int32
    L.crt_hip_world_instance.argtypes = [vp]
    L.crt_hip_world_instance.restype = C.c_int32
    L.crt_hip_bvh_copy_instances.argtypes = [vp, vp]
    L.crt_hip_prepare_scene.restype = vp
    L.crt_hip_prepare_scene.argtypes = [C.POINTER(SceneDesc), C.c_int]
    L.crt_hip_prepare_scene_on.restype = vp
    L.crt_hip_prepare_scene_on.argtypes = [C.POINTER(SceneDesc), C.c_int, C.c_int]
    L.crt_hip_free_prepared_scene.argtypes = [vp]
    L.crt_hip_free_prepared_scene.restype = None
    L.crt_hip_set_prepared_scene.argtypes = [vp, vp]
    L.crt_hip_save_prepared_scene.argtypes = [vp, C.c_char_p]
    L.crt_hip_load_prepared_scene.restype = vp
    L.crt_hip_load_prepared_scene.argtypes = [C.c_char_p]
    u64p = C.POINTER(C.c_uint64)
    L.crt_hip_prepared_scene_info.argtypes = [vp, u64p, u64p, u64p, i32p, fp, i32p, u32p, u32p, C.POINTER(C.c_double)]
    L.crt_hip_prepared_scene_copy.argtypes = [vp, vp, vp, vp]
    L.crt_hip_child_order.restype = C.c_int
    L.crt_hip_refine_state.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.crt_hip_refine_state.restype = C.c_int
    L.crt_hip_prepared_scene_set_spp.argtypes = [vp, C.c_uint32]
    L.crt_hip_prepared_scene_set_spp.restype = C.c_int
    L.crt_hip_lds_stack_entries.restype = C.c_uint32
    L.crt_hip_lds_stack_entries.argtypes = [C.c_int]
    for fn in ("crt_hip_set_stream", "crt_hip_set_partition", "crt_hip_initialize", "crt_hip_set_scene",
               "crt_hip_render", "crt_hip_read_accum", "crt_hip_read_ray_counts", "crt_hip_tile_buffer",
               "crt_hip_assemble_tiles", "crt_hip_trace_rays", "crt_hip_kat", "crt_hip_bvh_info",
               "crt_hip_bvh_copy", "crt_hip_bvh_layout", "crt_hip_bvh_copy_instances", "crt_hip_set_prepared_scene",
               "crt_hip_save_prepared_scene", "crt_hip_prepared_scene_info", "crt_hip_prepared_scene_copy"):
        getattr(L, fn).restype = C.c_int
    _lib = L
    return L


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def check(ctx, rc, what):
    """Error codes -> exceptions, the reference's error convention."""
    if rc != 0:
        msg = load().crt_hip_last_error(ctx)
        raise CoreError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
