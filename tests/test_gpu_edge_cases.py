"""GPU: edge cases of the render path against the oracle -- degenerate framebuffers, frames in
which every ray misses (all later queues are empty), many samples per pixel, resize
(`initialize` again, main.cpp:284-285) and scene replacement (`set_scene` again)."""
import numpy as np
import pytest

from chameleonrt_amd import scenes
from chameleonrt_amd.render_hip import RenderHIP
from tests.parity import MAX_DIVERGED, camera_of, compare_images

pytestmark = pytest.mark.gpu


def _check(r, o, e, d, u, fovy, frames=2):
    for f in range(frames):
        st = r.render(e, d, u, fovy, f == 0, True)
        ost = o.render(e, d, u, fovy, f == 0)
    diverged, mean_rel = compare_images(r.accum(), o.accum())
    assert diverged <= max(MAX_DIVERGED, 1.5 / r.accum()[..., 0].size) and mean_rel <= 1e-4
    assert abs(int(st.rays) - int(ost.rays)) <= max(4, int(2 * MAX_DIVERGED * ost.rays))
    return st


@pytest.mark.parametrize("size", [(1, 1), (1, 70), (65, 1), (63, 65), (129, 7)])
def test_degenerate_framebuffers(size, oracle, hip_lib):
    w, h = size
    sc = scenes.cornell(spp=3)
    r = RenderHIP()
    r.initialize(w, h)
    r.set_scene(sc)
    o = oracle.OracleRenderer(sc, w, h)
    _check(r, o, *camera_of(sc))
    assert r.img.shape == (h, w) and (r.img.view(np.uint8).reshape(h, w, 4)[..., 3] == 255).all()
    r.close()


def test_every_ray_misses(oracle, hip_lib):
    """Camera looking away from the scene: one closest-hit ray per sample, nothing else; the image
    is the checkerboard miss shader (render_embree.ispc:184-196)."""
    sc = scenes.cornell(spp=2)
    w, h = 96, 80
    r = RenderHIP()
    r.initialize(w, h)
    r.set_scene(sc)
    o = oracle.OracleRenderer(sc, w, h)
    e, d, u, fovy = camera_of(sc)
    st = _check(r, o, e + np.float32([0, 0, 10]), -d, u, fovy)
    assert int(st.rays) == w * h * 2 and int(st.shadow_rays) == 0
    # 2 frames x 2 spp of values in {0.1, 0.5}: every pixel is a multiple of 0.1 in [0.1, 0.5]
    a = r.accum()
    assert a.min() >= 0.1 - 1e-6 and a.max() <= 0.5 + 1e-6 and np.abs(a * 10 - np.round(a * 10)).max() < 1e-4
    r.close()


def test_many_samples_per_pixel(oracle, hip_lib):
    sc = scenes.cornell(spp=64)  # BASELINE config C5's spp
    w, h = 48, 40
    r = RenderHIP()
    r.initialize(w, h)
    r.set_scene(sc)
    o = oracle.OracleRenderer(sc, w, h)
    st = _check(r, o, *camera_of(sc), frames=1)
    assert int(st.closest_rays) >= w * h * 64
    assert r.ray_counts().max() > 255  # would not fit the reference's counters if they were 8-bit; 16-bit ok
    r.close()


def test_resize_and_scene_replacement(oracle, hip_lib):
    sc1, sc2 = scenes.cornell(spp=1), scenes.instanced_grove()
    r = RenderHIP()
    r.initialize(64, 64)
    r.set_scene(sc1)
    e, d, u, fovy = camera_of(sc1)
    r.render(e, d, u, fovy, True, False)
    r.initialize(100, 50)  # window resize: new accumulation, same scene
    assert r.frame_id() == 0
    o = oracle.OracleRenderer(sc1, 100, 50)
    _check(r, o, e, d, u, fovy)
    r.set_scene(sc2)  # new scene on the same context (two-level, textures, different spp)
    assert r.frame_id() == 0 and r.samples_per_pixel == sc2.samples_per_pixel
    o2 = oracle.OracleRenderer(sc2, 100, 50)
    _check(r, o2, *camera_of(sc2))
    r.close()


def test_overlapped_launch_schedule_is_bit_identical(hip_lib, monkeypatch):
    """CRT_HIP_OVERLAP=1 puts the occlusion launch of bounce b on a second stream next to the
    closest-hit launch of bounce b+1 (they are independent). Same kernels, same order of the
    radiance updates per path: accumulated radiance, ray counts and the 8-bit image must not change
    by a single bit. Two-level scene, so both traversal stacks (and both spill slabs) are in use."""
    sc = scenes.instanced_grove()
    w, h = 160, 96
    e, d, u, fovy = camera_of(sc)
    out = []
    for overlap in ("0", "1"):
        monkeypatch.setenv("CRT_HIP_OVERLAP", overlap)  # read when the context is created
        r = RenderHIP()
        r.initialize(w, h)
        r.set_scene(sc)
        for f in range(3):
            st = r.render(e, d, u, fovy, f == 0, True)
        out.append((r.accum().copy(), r.ray_counts().copy(), r.img.copy(), int(st.rays)))
        r.close()
    a, b = out
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3]


def test_pass_lanes_are_bit_identical(hip_lib, monkeypatch):
    """With the overlapped schedule a frame is cut into passes that run on independent lanes (own queues, counters,
    streams, spill slabs: crt_core.cpp PassLane) so that one lane's launches fill the tails of the other's. Pixels are
    independent and a pass covers whole pixel slots: 1, 2 and 3 lanes, a frame cut in two by the lanes themselves
    or three by the lanes themselves (1 M paths) and one cut into many small passes, give the same radiance, ray counts
    and image bit for bit."""
    sc = scenes.instanced_grove(spp=4)
    e, d, u, fovy = camera_of(sc)
    for (w, h), max_paths in (((640, 384), None), ((192, 128), "30000")):
        if max_paths:
            monkeypatch.setenv("CRT_HIP_MAX_PATHS", max_paths)
        out = []
        for lanes in ("1", "2", "3"):
            monkeypatch.setenv("CRT_HIP_LANES", lanes)  # read when the context is created
            r = RenderHIP()
            r.initialize(w, h)
            r.set_scene(sc)
            for f in range(2):
                st = r.render(e, d, u, fovy, f == 0, True)
            out.append((r.accum().copy(), r.ray_counts().copy(), r.img.copy(), int(st.rays)))
            r.close()
        for b in out[1:]:
            assert np.array_equal(out[0][0].view(np.uint32), b[0].view(np.uint32))
            assert np.array_equal(out[0][1], b[1]) and np.array_equal(out[0][2], b[2]) and out[0][3] == b[3]
        assert out[0][3] > 0


def test_device_framebuffer_is_the_image_without_a_host_round_trip(hip_lib):
    """Display interop hand-off: with readback = False nothing reaches the host image, and the
    device pointer holds exactly what readback = True would have delivered."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    sc = scenes.cornell(spp=2)
    e, d, u, fovy = camera_of(sc)
    r = RenderHIP()
    r.initialize(200, 136)
    r.set_scene(sc)
    r.render(e, d, u, fovy, True, False)
    assert not r.img.any(), "readback = False must not touch the host image"
    ptr, pitch = r.device_framebuffer()
    assert pitch == 200 * 4
    dev = np.zeros((136, 200), np.uint32)
    assert hip.hipMemcpy(dev.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(dev.nbytes), 2) == 0  # device to host
    r.render(e, d, u, fovy, True, True)  # the same frame again, read back the reference's way
    assert np.array_equal(dev, r.img) and (dev.view(np.uint8).reshape(136, 200, 4)[..., 3] == 255).all()
    r.close()


def test_tile_buffer_alternates_with_a_moving_camera(hip_lib):
    """The compact tile buffer a multi-GPU gather reads alternates with every RENDERED frame -- also when every frame
    has camera_changed set (frame_id is then 0 each time): the asynchronous gather of frame f may still be reading its
    buffer while frame f+1 is accumulated, so two consecutive frames must never share one (round-2 advisor finding)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")  # resolves to the runtime the core already uses
    sc = scenes.cornell(spp=1)
    e, d, u, fovy = camera_of(sc)
    r = RenderHIP(rank=0, world=2)
    r.initialize(128, 128)
    r.set_scene(sc)
    ptrs, images = [], []
    for f in range(4):
        e2 = e + np.float32(0.05 * f) * np.array([1, 0, 0], np.float32)
        r.render(e2, d, u, fovy, True, False)
        assert r.frame_id() == 1
        p, n = r.tile_buffer()
        ptrs.append(p)
        buf = np.zeros(n // 4, np.uint32)
        assert hip.hipMemcpy(buf.ctypes.data_as(C.c_void_p), C.c_void_p(p), C.c_size_t(n), 2) == 0  # device -> host
        images.append(buf)
    assert ptrs[0] != ptrs[1] and ptrs[0] == ptrs[2] and ptrs[1] == ptrs[3]
    # and each buffer holds the frame that was rendered into it last (the four cameras give four different images)
    assert not np.array_equal(images[0], images[1]) and not np.array_equal(images[1], images[2])
    r.close()


@pytest.mark.parametrize("all_empty", [True, False])
@pytest.mark.parametrize("levels", ["two", "world"])
def test_meshes_without_triangles_render(all_empty, levels, oracle, hip_lib, monkeypatch):
    """Instances of a mesh without triangles hit nothing, and a scene in which no instance has a triangle renders the miss
    shader's checkerboard (one ray per path) -- in both structures for scenes with several instances, like the oracle."""
    from tests.test_prepared_scene import _scene_with_empty_meshes
    monkeypatch.setenv("CRT_HIP_LEVELS", levels)
    sc = _scene_with_empty_meshes(all_empty)
    w, h = 96, 64
    r = RenderHIP()
    r.initialize(w, h)
    r.set_scene(sc)
    o = oracle.OracleRenderer(sc, w, h)
    e, d, u, fovy = camera_of(sc)
    st = _check(r, o, e, d, u, fovy, frames=2)
    if all_empty:
        assert int(st.rays) == w * h * sc.samples_per_pixel
    r.close()


def test_pipelined_frames_equal_synchronous_frames(hip_lib):
    """crt_hip_render_begin / _end (include/crt_hip.h): the frame loop of the multi-GPU bench enqueues frame f+1 before it
    collects frame f so that the GPU never waits for the host between frames. Same launches in the same order on the same
    stream: accumulated radiance, ray counts, RGBA8 and every frame's ray statistics are those of crt_hip_render, bit
    for bit -- for a frame size the library cuts for pass lanes by trial (its trial frames complete inside render_begin)
    and for one it never cuts. A third frame in flight, and re-configuring with frames in flight, are refused."""
    from chameleonrt_amd import core
    sc = scenes.instanced_grove(spp=4)
    e, d, u, fovy = camera_of(sc)
    for w, h in ((256, 192), (640, 400)):  # 0.2 M paths (never cut) / 1 M paths (tried with one lane and with two)
        a = RenderHIP()
        a.initialize(w, h)
        a.set_scene(sc)
        sync = [a.render(e, d, u, fovy, f == 0, False).rays for f in range(9)]
        b = RenderHIP()
        b.initialize(w, h)
        b.set_scene(sc)
        piped = []
        b.render_begin(e, d, u, fovy, True, False)
        for f in range(1, 9):
            b.render_begin(e, d, u, fovy, False, False)  # frame f goes in while frame f-1 is still uncollected
            if f == 3:
                with pytest.raises(core.CoreError, match="two frames are in flight"):
                    b.render_begin(e, d, u, fovy, False, False)
                with pytest.raises(core.CoreError, match="in flight"):
                    b.initialize(w, h)
            piped.append(b.render_end().rays)
            if f == 4:  # ONE frame pending now: a frame WITH host readback is refused (one host image), so are the debugging entry points
                with pytest.raises(core.CoreError, match="one host image"):
                    b.render_begin(e, d, u, fovy, False, True)
                with pytest.raises(core.CoreError, match="in flight"):
                    b.bvh()
        st = b.render_end()
        piped.append(st.rays)
        with pytest.raises(core.CoreError, match="without a frame in flight"):
            b.render_end()
        assert piped == sync
        assert st.render_time_ms > 0 and st.passes >= 1
        assert np.array_equal(a.accum().view(np.uint32), b.accum().view(np.uint32))
        assert np.array_equal(a.ray_counts(), b.ray_counts())
        # the synchronous entry point works again once nothing is in flight
        assert b.render(e, d, u, fovy, False, True).rays == a.render(e, d, u, fovy, False, True).rays
        assert np.array_equal(a.img, b.img)
        a.close()
        b.close()
