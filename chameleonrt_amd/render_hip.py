"""RenderHIP -- host-side mirror of the reference's ``RenderBackend`` for the HIP core.

Same names, argument meaning and error behaviour as the reference interface
(util/render_backend.h:12-32; the Embree implementation backends/embree/render_embree.h:11-44):

    r = RenderHIP()              # make_renderer(display)
    r.initialize(w, h)           # RenderBackend::initialize
    r.set_scene(scene)           # RenderBackend::set_scene (copies what it needs)
    stats = r.render(pos, dir, up, fovy, camera_changed, readback_framebuffer)
    r.img                        # W*H RGBA8, row 0 = top (RenderBackend::img)

Errors are exceptions (the reference throws std::runtime_error). Everything is executed by
libcrt_hip_core.so on the GPU; this class holds no rendering logic.
"""
import ctypes as C

import numpy as np

from . import core
from .scene import PackedScene, Scene


class RenderHIP:
    def __init__(self, device: int = 0, flags: int = 0, rank: int = 0, world: int = 1, stream=None):
        self._lib = core.load()
        self._ctx = self._lib.crt_hip_create(device, flags)
        if not self._ctx:
            raise core.CoreError("crt_hip_create failed: " + self._lib.crt_hip_last_error(None).decode())
        self.samples_per_pixel = 1
        self.width = self.height = 0
        self.rank, self.world = rank, world
        if world > 1:
            core.check(self._ctx, self._lib.crt_hip_set_partition(self._ctx, rank, world), "set_partition")
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.crt_hip_destroy(self._ctx)
            self._ctx = None

    __del__ = close

    def name(self) -> str:
        return self._lib.crt_hip_name(self._ctx).decode()

    def set_stream(self, stream_handle: int):
        core.check(self._ctx, self._lib.crt_hip_set_stream(self._ctx, C.c_void_p(stream_handle)), "set_stream")

    def initialize(self, fb_width: int, fb_height: int):
        core.check(self._ctx, self._lib.crt_hip_initialize(self._ctx, fb_width, fb_height), "initialize")
        self.width, self.height = fb_width, fb_height

    def set_scene(self, scene: Scene):
        packed = PackedScene(scene)
        core.check(self._ctx, self._lib.crt_hip_set_scene(self._ctx, packed.ptr()), "set_scene")
        self.samples_per_pixel = scene.samples_per_pixel

    def set_prepared_scene(self, prepared: "PreparedScene"):
        """Device half of set_scene: upload a scene prepared once per node (multi-GPU)."""
        core.check(self._ctx, self._lib.crt_hip_set_prepared_scene(self._ctx, prepared.handle), "set_prepared_scene")
        self.samples_per_pixel = prepared.samples_per_pixel

    def refine_state(self):
        """(state, quick_ms, full_ms) of FLAG_REFINE_IN_BACKGROUND (crt_hip_refine_state): 0 none, 1 building, 2 ready, 3 in use."""
        q, f = C.c_double(), C.c_double()
        st = self._lib.crt_hip_refine_state(self._ctx, C.byref(q), C.byref(f))
        if st < 0:
            raise core.CoreError(self._lib.crt_hip_last_error(self._ctx).decode())
        return st, q.value, f.value

    def render(self, pos, dir, up, fovy, camera_changed, readback_framebuffer=False) -> core.RenderStats:
        a = [np.ascontiguousarray(v, np.float32) for v in (pos, dir, up)]
        st = core.RenderStats()
        core.check(self._ctx, self._lib.crt_hip_render(self._ctx, core.fptr(a[0]), core.fptr(a[1]), core.fptr(a[2]),
                                                       float(fovy), int(bool(camera_changed)),
                                                       int(bool(readback_framebuffer)), C.byref(st)), "render")
        return st

    def render_begin(self, pos, dir, up, fovy, camera_changed, readback_framebuffer=False):
        """Enqueue a frame without waiting for it (crt_hip_render_begin); at most two may be in flight."""
        a = [np.ascontiguousarray(v, np.float32) for v in (pos, dir, up)]
        core.check(self._ctx, self._lib.crt_hip_render_begin(self._ctx, core.fptr(a[0]), core.fptr(a[1]), core.fptr(a[2]), float(fovy),
                                                             int(bool(camera_changed)), int(bool(readback_framebuffer))), "render_begin")

    def render_end(self) -> core.RenderStats:
        """Wait for the oldest frame in flight and return its statistics (crt_hip_render_end)."""
        st = core.RenderStats()
        core.check(self._ctx, self._lib.crt_hip_render_end(self._ctx, C.byref(st)), "render_end")
        return st

    @property
    def img(self) -> np.ndarray:
        p = self._lib.crt_hip_framebuffer(self._ctx)
        return np.ctypeslib.as_array(p, shape=(self.height, self.width))

    def device_framebuffer(self):
        """(device pointer, pitch in bytes) of the row-major RGBA8 image in HBM: the display-interop hand-off."""
        p, pitch = C.c_void_p(), C.c_size_t()
        core.check(self._ctx, self._lib.crt_hip_device_framebuffer(self._ctx, C.byref(p), C.byref(pitch)), "device_framebuffer")
        return p.value, pitch.value

    def frame_id(self) -> int:
        return self._lib.crt_hip_frame_id(self._ctx)

    # ---- parity / diagnostic reads ------------------------------------------------------
    def accum(self) -> np.ndarray:
        out = np.zeros((self.height, self.width, 3), np.float32)
        core.check(self._ctx, self._lib.crt_hip_read_accum(self._ctx, core.fptr(out)), "read_accum")
        return out

    def ray_counts(self) -> np.ndarray:
        out = np.zeros((self.height, self.width), np.uint32)
        core.check(self._ctx, self._lib.crt_hip_read_ray_counts(
            self._ctx, out.ctypes.data_as(C.POINTER(C.c_uint32))), "read_ray_counts")
        return out

    def trace(self, org, dirs, tmin, tmax, closest=True, production=False):
        """production: through the kernels a frame launches (k_trace_closest / k_trace_shadow, no counters) instead of
        the instrumented diagnostic kernel; closest hits then need tmax = 1e20 and tmin = 0 or EPSILON."""
        org = np.ascontiguousarray(org, np.float32)
        dirs = np.ascontiguousarray(dirs, np.float32)
        n = org.shape[0]
        tmin = np.ascontiguousarray(np.broadcast_to(np.asarray(tmin, np.float32), (n,)))
        tmax = np.ascontiguousarray(np.broadcast_to(np.asarray(tmax, np.float32), (n,)))
        t, u, v = (np.zeros(n, np.float32) for _ in range(3))
        inst, geom, prim = (np.zeros(n, np.int32) for _ in range(3))
        st = core.RenderStats()
        ip = lambda x: x.ctypes.data_as(C.POINTER(C.c_int32))
        core.check(self._ctx, self._lib.crt_hip_trace_rays(
            self._ctx, n, core.fptr(org), core.fptr(dirs), core.fptr(tmin), core.fptr(tmax),
            int(closest) | (core.TRACE_PRODUCTION if production else 0), core.fptr(t), core.fptr(u), core.fptr(v), ip(inst), ip(geom), ip(prim), C.byref(st)), "trace_rays")
        return dict(t=t, u=u, v=v, inst=inst, geom=geom, prim=prim, stats=st)

    def kat(self, fn: int, rec_in, n_out: int) -> np.ndarray:
        rec_in = np.ascontiguousarray(rec_in, np.float32)
        out = np.zeros((rec_in.shape[0], n_out), np.float32)
        core.check(self._ctx, self._lib.crt_hip_kat(self._ctx, fn, rec_in.shape[0], core.fptr(rec_in),
                                                    rec_in.shape[1], core.fptr(out), n_out), "kat")
        return out

    def bvh(self):
        nn, nt, ni, tl = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int32()
        frame = np.zeros(6, np.float32)
        core.check(self._ctx, self._lib.crt_hip_bvh_info(self._ctx, C.byref(nn), C.byref(nt), C.byref(ni),
                                                         C.byref(tl), core.fptr(frame)), "bvh_info")
        nodes = np.zeros((nn.value, 16), np.uint32)  # 64-byte 4-wide nodes
        tris = np.zeros((nt.value, 16), np.uint32)  # 64-byte leaf slots: 4 vertices (float bits), geomID | selectors, primIDs, tag
        core.check(self._ctx, self._lib.crt_hip_bvh_copy(self._ctx, nodes.ctypes.data_as(C.c_void_p),
                                                         tris.ctypes.data_as(C.c_void_p)), "bvh_copy")
        root, child_order = C.c_int32(), C.c_int32()
        n_top, need, lds = C.c_uint32(), C.c_uint32(), C.c_uint32()
        core.check(self._ctx, self._lib.crt_hip_bvh_layout(self._ctx, C.byref(root), C.byref(n_top), C.byref(need),
                                                           C.byref(lds), C.byref(child_order)), "bvh_layout")
        insts = np.zeros((ni.value, 32), np.uint32)  # 128-byte instance records
        core.check(self._ctx, self._lib.crt_hip_bvh_copy_instances(self._ctx, insts.ctypes.data_as(C.c_void_p)),
                   "bvh_copy_instances")
        return dict(nodes=nodes, tris=tris, instances=insts, n_instances=ni.value, two_level=tl.value == 1, levels=tl.value,
                    frame=frame, root=root.value, n_top_nodes=n_top.value, stack_need=need.value,
                    lds_stack=lds.value, child_order=child_order.value,
                    world_inst=self._lib.crt_hip_world_instance(self._ctx))

    # ---- multi-GPU tile assembly ---------------------------------------------------------
    def tile_buffer(self):
        p, n = C.c_void_p(), C.c_size_t()
        core.check(self._ctx, self._lib.crt_hip_tile_buffer(self._ctx, C.byref(p), C.byref(n)), "tile_buffer")
        return p.value, n.value

    def assemble_tiles(self, gathered_device_ptr: int, world: int, readback: bool = True):
        core.check(self._ctx, self._lib.crt_hip_assemble_tiles(self._ctx, C.c_void_p(gathered_device_ptr), world,
                                                               int(readback)), "assemble_tiles")


class PreparedScene:
    """Host half of set_scene (BVH build, texture linearisation), done once and uploaded to every
    GPU of the node: crt_hip_prepare_scene / save / load (include/crt_hip.h)."""

    def __init__(self, scene: Scene = None, path: str = None, n_threads: int = 0, build_device: int = -1):
        """build_device >= 0: the BLAS of large meshes is built on that GPU (linear BVH) instead of by the host SAH builder."""
        self._lib = core.load()
        if scene is not None:
            packed = PackedScene(scene)
            self.handle = self._lib.crt_hip_prepare_scene_on(packed.ptr(), n_threads, build_device)
            self.samples_per_pixel = scene.samples_per_pixel
        else:
            self.handle = self._lib.crt_hip_load_prepared_scene(path.encode())
            self.samples_per_pixel = None  # the caller knows (it is stored in the prepared scene too)
        if not self.handle:
            raise core.CoreError("prepare/load scene failed: " + self._lib.crt_hip_last_error(None).decode())

    def bvh(self):
        """The host-built traversal arrays, same dict as RenderHIP.bvh() (no device involved)."""
        nn, nt, ni, tl, root = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int32(), C.c_int32()
        n_top, need, ms = C.c_uint32(), C.c_uint32(), C.c_double()
        frame = np.zeros(6, np.float32)
        rc = self._lib.crt_hip_prepared_scene_info(self.handle, C.byref(nn), C.byref(nt), C.byref(ni), C.byref(tl),
                                                   core.fptr(frame), C.byref(root), C.byref(n_top), C.byref(need),
                                                   C.byref(ms))
        assert rc == 0
        nodes = np.zeros((nn.value, 16), np.uint32)
        tris = np.zeros((nt.value, 16), np.uint32)  # 64-byte leaf slots: 4 vertices (float bits), geomID | selectors, primIDs, tag
        insts = np.zeros((ni.value, 32), np.uint32)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        assert self._lib.crt_hip_prepared_scene_copy(self.handle, vp(nodes), vp(tris), vp(insts)) == 0
        return dict(nodes=nodes, tris=tris, instances=insts, n_instances=ni.value, two_level=tl.value == 1, levels=tl.value,
                    frame=frame, root=root.value, n_top_nodes=n_top.value, stack_need=need.value,
                    child_order=self._lib.crt_hip_child_order(), lds_stack=self._lib.crt_hip_lds_stack_entries(int(tl.value)),
                    build_ms=ms.value, world_inst=self._lib.crt_hip_prepared_scene_world_instance(self.handle))

    def levels(self) -> int:
        """0: one instance; 1: two-level (top-level tree over instances); 2: world tree (include/crt_hip.h)."""
        tl = C.c_int32()
        assert self._lib.crt_hip_prepared_scene_info(self.handle, None, None, None, C.byref(tl), None, None, None, None, None) == 0
        return tl.value

    def set_samples_per_pixel(self, spp: int):
        assert self._lib.crt_hip_prepared_scene_set_spp(self.handle, spp) == 0
        self.samples_per_pixel = spp

    def save(self, path: str):
        rc = self._lib.crt_hip_save_prepared_scene(self.handle, path.encode())
        if rc != 0:
            raise core.CoreError("save_prepared_scene failed: " + self._lib.crt_hip_last_error(None).decode())

    def close(self):
        if getattr(self, "handle", None):
            self._lib.crt_hip_free_prepared_scene(self.handle)
            self.handle = None

    __del__ = close
