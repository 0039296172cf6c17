"""ctypes binding of the CPU oracle (oracle/liborc.so). TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

from chameleonrt_amd.scene import PackedScene, Scene, SceneDesc

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


class OrcStats(C.Structure):
    _fields_ = [("render_time_ms", C.c_double), ("rays_per_second", C.c_double),
                ("rays", C.c_uint64), ("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64)]


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle")])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_ROOT, "oracle", "liborc.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        fp, u32p, i32p, vp = (C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.c_void_p)
        L.orc_scene_create.restype = vp
        L.orc_scene_create.argtypes = [C.POINTER(SceneDesc)]
        L.orc_scene_destroy.argtypes = [vp]
        L.orc_scene_num_triangles.restype = C.c_uint64
        L.orc_scene_num_triangles.argtypes = [vp]
        L.orc_renderer_create.restype = vp
        L.orc_renderer_create.argtypes = [vp, C.c_int, C.c_int, C.c_int]
        L.orc_renderer_destroy.argtypes = [vp]
        L.orc_render.argtypes = [vp, fp, fp, fp, C.c_float, C.c_int, C.c_int, C.c_int, C.POINTER(OrcStats)]
        L.orc_render_tiles.argtypes = [vp, fp, fp, fp, C.c_float, C.c_int, i32p, C.c_int, C.POINTER(OrcStats)]
        L.orc_framebuffer.restype = u32p
        L.orc_framebuffer.argtypes = [vp]
        L.orc_read_accum.argtypes = [vp, fp]
        L.orc_read_ray_counts.argtypes = [vp, u32p]
        L.orc_num_tiles.argtypes = [vp]
        L.orc_trace_rays.argtypes = [vp, C.c_uint64, fp, fp, fp, fp, C.c_int, C.c_int, fp, fp, fp,
                                     i32p, i32p, i32p, C.POINTER(OrcStats)]
        L.orc_walk_foreign_bvh.argtypes = [vp, vp, vp, C.c_uint64, C.c_int32, C.c_int32, fp, C.c_int, C.c_uint64, fp, fp, fp, fp,
                                           C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                           C.POINTER(C.c_uint32), fp, i32p, i32p, i32p, C.POINTER(C.c_uint64), C.c_int, C.POINTER(C.c_uint64)]
        L.orc_kat.argtypes = [vp, C.c_int, C.c_uint64, fp, C.c_int, fp, C.c_int]
        L.orc_set_schlick_by_multiplication.argtypes = [C.c_int]
        L.orc_set_schlick_by_multiplication.restype = None
        _LIB = L
    return _LIB


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class OracleScene:
    def __init__(self, scene: Scene):
        self.packed = PackedScene(scene)
        self.h = lib().orc_scene_create(self.packed.ptr())
        if not self.h:
            raise RuntimeError("orc_scene_create failed")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_scene_destroy(self.h)
            self.h = None

    def trace(self, org, dirs, tmin, tmax, closest=True, brute_force=False):
        org = np.ascontiguousarray(org, np.float32)
        dirs = np.ascontiguousarray(dirs, np.float32)
        n = org.shape[0]
        tmin = np.ascontiguousarray(np.broadcast_to(np.asarray(tmin, np.float32), (n,)))
        tmax = np.ascontiguousarray(np.broadcast_to(np.asarray(tmax, np.float32), (n,)))
        t, u, v = (np.zeros(n, np.float32) for _ in range(3))
        inst, geom, prim = (np.zeros(n, np.int32) for _ in range(3))
        st = OrcStats()
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        rc = lib().orc_trace_rays(self.h, n, _fp(org), _fp(dirs), _fp(tmin), _fp(tmax), int(closest),
                                  int(brute_force), _fp(t), _fp(u), _fp(v), ip(inst), ip(geom), ip(prim),
                                  C.byref(st))
        assert rc == 0
        return dict(t=t, u=u, v=v, inst=inst, geom=geom, prim=prim, nodes=st.nodes_visited,
                    tris=st.tris_tested)

    def kat(self, fn, rec_in, n_out):
        return kat(fn, rec_in, n_out, self.h)


def walk_product_bvh(bvh, org, dirs, tmin, tmax, closest=True):
    """Walk the PRODUCT's BVH arrays (RenderHIP.bvh(): packed 4-wide nodes, triangle and instance
    records) on the CPU with the product's documented visit rule. Returns node / triangle visit counts
    (what the instrumented kernels must report), the deepest stack any ray needed, and the hits."""
    org = np.ascontiguousarray(org, np.float32)
    dirs = np.ascontiguousarray(dirs, np.float32)
    n = org.shape[0]
    tmin = np.ascontiguousarray(np.broadcast_to(np.asarray(tmin, np.float32), (n,)))
    tmax = np.ascontiguousarray(np.broadcast_to(np.asarray(tmax, np.float32), (n,)))
    nv, tt, ms, ne, ls = C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_uint64(), C.c_uint64()
    t = np.zeros(n, np.float32)
    inst, geom, prim = (np.zeros(n, np.int32) for _ in range(3))
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib().orc_walk_foreign_bvh(vp(bvh["nodes"]), vp(bvh["tris"]), vp(bvh["instances"]), bvh["n_instances"],
                                    bvh.get("world_inst", -1), bvh["root"], _fp(bvh["frame"]), bvh["child_order"], n, _fp(org), _fp(dirs),
                                    _fp(tmin), _fp(tmax), int(closest), C.byref(nv), C.byref(tt), C.byref(ms),
                                    _fp(t), ip(inst), ip(geom), ip(prim), C.byref(ne), int(bvh.get("levels", -1)), C.byref(ls))
    assert rc == 0
    return dict(nodes=nv.value, tris=tt.value, max_stack=ms.value, t=t, inst=inst, geom=geom, prim=prim,
                inst_entries=ne.value, slots=ls.value)


def kat(fn, rec_in, n_out, scene_handle=None):
    rec_in = np.ascontiguousarray(rec_in, np.float32)
    out = np.zeros((rec_in.shape[0], n_out), np.float32)
    rc = lib().orc_kat(scene_handle, fn, rec_in.shape[0], _fp(rec_in), rec_in.shape[1], _fp(out), n_out)
    assert rc == 0
    return out


class OracleRenderer:
    """RenderEmbree-shaped wrapper: initialize / set_scene / render (reference
    backends/embree/render_embree.h:11-44)."""

    def __init__(self, scene: Scene, width: int, height: int, n_threads: int = 0):
        self.scene = OracleScene(scene)
        self.w, self.h_ = width, height
        self.r = lib().orc_renderer_create(self.scene.h, width, height, n_threads)
        if not self.r:
            raise RuntimeError("orc_renderer_create failed")

    def __del__(self):
        if getattr(self, "r", None):
            lib().orc_renderer_destroy(self.r)
            self.r = None

    def num_tiles(self):
        return lib().orc_num_tiles(self.r)

    def render(self, pos, dir, up, fovy, camera_changed, tile_begin=0, tile_end=-1):
        st = OrcStats()
        a = [np.ascontiguousarray(x, np.float32) for x in (pos, dir, up)]
        rc = lib().orc_render(self.r, _fp(a[0]), _fp(a[1]), _fp(a[2]), float(fovy), int(camera_changed),
                              tile_begin, tile_end, C.byref(st))
        assert rc == 0
        return st

    def render_tiles(self, pos, dir, up, fovy, camera_changed, tile_ids):
        """The frame restricted to the reference's 64x64 tiles named in tile_ids (row-major tile ids)."""
        st = OrcStats()
        a = [np.ascontiguousarray(x, np.float32) for x in (pos, dir, up)]
        ids = np.ascontiguousarray(tile_ids, np.int32)
        rc = lib().orc_render_tiles(self.r, _fp(a[0]), _fp(a[1]), _fp(a[2]), float(fovy), int(camera_changed),
                                    ids.ctypes.data_as(C.POINTER(C.c_int32)), len(ids), C.byref(st))
        assert rc == 0
        return st

    def accum(self):
        out = np.zeros((self.h_, self.w, 3), np.float32)
        lib().orc_read_accum(self.r, _fp(out))
        return out

    def ray_counts(self):
        out = np.zeros((self.h_, self.w), np.uint32)
        lib().orc_read_ray_counts(self.r, out.ctypes.data_as(C.POINTER(C.c_uint32)))
        return out

    def framebuffer(self):
        p = lib().orc_framebuffer(self.r)
        return np.ctypeslib.as_array(p, shape=(self.h_, self.w)).copy()
