"""GPU: the C++ side of the drop-in boundary, executed.

`oracle/_ref/crt_bench` (tools/crt_bench.cpp, built by `make -C oracle ref`) is `./chameleonrt hip <scene>` minus the
window: the reference's OWN plugin loader (`RenderPlugin("crt_hip")`, util/render_plugin.cpp:14-60, compiled from where it
lies) dlopens `oracle/_ref/libcrt_hip.so` -- backends/hip/render_hip_plugin.cpp + render_hip.cpp, compiled against the
reference's real util/render_plugin.h, render_backend.h, scene.h, mesh.h, material.h, lights.h, display/gldisplay.h, imgui.h
-- dlsyms `populate_plugin_functions`, make_renderer hands back a `RenderBackend` and the `-benchmark-frames` loop of
main.cpp:293-345 runs on it. Named a scene file, it loads it with the reference's own importer
(`Scene(fname, MaterialMode)`, util/scene.cpp:49-72 compiled from where it lies) and hands that `Scene` to
`set_scene(const Scene &)` exactly as main.cpp:185-214 does. Stand-ins only for GLM and <SDL.h>, which the reference
itself fetches from outside its tree.

The bar: its image equals, bit for bit, what the ctypes path renders from chameleonrt_amd's own importers (obj_io /
gltf_io / crts_io) for the same file, camera and frame count -- both front ends sit on the same C-ABI and the importers are
pinned to each other (tests/test_importers_pinned.py). With >= 2 devices the CRT_HIP_DEVICES=2 run also goes through
ncclCommInitAll, the grouped ncclSend/ncclRecv gather and kernel K8 (render_hip.cpp); on a 1-GPU box that part is skipped.
"""
import os
import subprocess

import numpy as np
import pytest

from chameleonrt_amd import core
from chameleonrt_amd.render_hip import RenderHIP
from chameleonrt_amd.scene import (Camera, Geometry, Instance, Mesh, ParameterizedMesh, Scene, disney_material,
                                   quad_light)

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "oracle", "_ref", "crt_bench")
SCENES = os.path.join(ROOT, "tests", "golden", "scenes")
F = np.float32


def _norm(v):  # tools/crt_bench.cpp norm(): v * (1 / sqrt(dot)), all in fp32
    v = np.asarray(v, F)
    c = F(1.0) / np.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2], dtype=F)
    return (v * c).astype(F)


def _cross(a, b):
    a, b = np.asarray(a, F), np.asarray(b, F)
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], F)


def _quad(a, b, c, d):
    return Geometry(np.array([a, b, c, d], F), np.array([[0, 1, 2], [0, 2, 3]], np.uint32),
                    np.array([[0, 0], [1, 0], [1, 1], [0, 1]], F))


def _box(c, h):
    v = [[c[0] + (h[0] if i & 1 else -h[0]), c[1] + (h[1] if i & 2 else -h[1]), c[2] + (h[2] if i & 4 else -h[2])]
         for i in range(8)]
    f = [[0, 2, 1], [1, 2, 3], [4, 5, 6], [5, 7, 6], [0, 1, 4], [1, 5, 4], [2, 6, 3], [3, 6, 7], [0, 4, 2], [2, 4, 6],
         [1, 3, 5], [3, 7, 5]]
    return Geometry((np.array(v, np.float64)).astype(F), np.array(f, np.uint32), None)


def crt_bench_scene(spp):
    """The scene tools/crt_bench.cpp builds in code, value for value."""
    h1 = np.array([0.3, 0.3, 0.3], F)
    geoms = [_quad((-1, 0, 1), (1, 0, 1), (1, 0, -1), (-1, 0, -1)), _quad((-1, 2, -1), (1, 2, -1), (1, 2, 1), (-1, 2, 1)),
             _quad((-1, 0, -1), (1, 0, -1), (1, 2, -1), (-1, 2, -1)), _quad((-1, 0, 1), (-1, 0, -1), (-1, 2, -1), (-1, 2, 1)),
             _quad((1, 0, -1), (1, 0, 1), (1, 2, 1), (1, 2, -1))]
    for c, h in ((np.array([0.33, 0.3, 0.35], F), h1), (np.array([-0.35, 0.6, -0.3], F), np.array([0.3, 0.6, 0.3], F))):
        v = np.array([[c[0] + (h[0] if i & 1 else -h[0]), c[1] + (h[1] if i & 2 else -h[1]),
                       c[2] + (h[2] if i & 4 else -h[2])] for i in range(8)], F)  # fp32 adds like the C++
        geoms.append(Geometry(v, _box(c, h).indices, None))

    def diffuse(r, g, b):
        m = np.zeros(16, F)
        m[0:3] = (r, g, b)
        m[5] = 1.0   # roughness
        m[12] = 1.5  # ior
        return m

    n = _norm([0.5, -0.8, -0.5])
    v_x = _norm(_cross([1, 0, 0], n))
    v_y = _norm(_cross(n, v_x))
    light = quad_light([20.0, 20.0, 20.0, 20.0], (F(-10.0) * n).astype(F), n, v_x, v_y, 5.0, 5.0)
    sc = Scene(meshes=[Mesh(geoms)], parameterized_meshes=[ParameterizedMesh(0, [0, 0, 0, 1, 2, 0, 0])],
               instances=[Instance(np.eye(4, dtype=F).reshape(16), 0)],
               materials=[diffuse(0.73, 0.73, 0.73), diffuse(0.65, 0.05, 0.05), diffuse(0.12, 0.45, 0.15)],
               lights=[light], cameras=[Camera(np.array([0, 1, 3.4], F), np.array([0, 1, 0], F), np.array([0, 1, 0], F), 40.0)],
               samples_per_pixel=spp)
    eye = np.array([0, 1, 3.4], F)
    d = _norm(np.array([0, 1, 0], F) - eye)
    up = _norm(_cross(_norm(_cross(d, [0, 1, 0])), d))
    return sc, eye, d, up


def _run_bench(tmp_path, devices, w, h, spp, frames, extra=()):
    out = str(tmp_path / f"bench_{devices}.ppm")
    env = dict(os.environ, CRT_HIP_DEVICES=str(devices))
    p = subprocess.run([BENCH, *extra, "-img", str(w), str(h), "-spp", str(spp), "-benchmark-frames", str(frames), "-ppm", out],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert "Benchmarked %d frames" % frames in p.stdout and "Rays per-second" in p.stdout, p.stdout
    with open(out, "rb") as f:
        assert f.readline().split() == [b"P6", str(w).encode(), str(h).encode(), b"255"]
        rgb = np.frombuffer(f.read(), np.uint8).reshape(h, w, 3)
    return rgb, p.stdout


def _camera_line(out):
    """The camera crt_bench handed to render(), as it printed it (hex floats: the same bits for the ctypes path)."""
    line = [ln for ln in out.splitlines() if ln.startswith("camera:")][0]
    v = np.array([float.fromhex(t) for t in line.split()[1:]], F)
    return v[0:3], v[3:6], v[6:9], float(v[9])


def test_crt_bench_image_equals_ctypes_path(tmp_path, hip_lib):
    if not os.path.exists(BENCH):
        pytest.skip("oracle/_ref/crt_bench not built (`make -C oracle ref`, where /root/reference exists)")
    w, h, spp, frames = 256, 192, 2, 3
    rgb, out = _run_bench(tmp_path, 1, w, h, spp, frames)
    assert "HIP wavefront path tracer" in out
    sc, eye, d, up = crt_bench_scene(spp)
    r = RenderHIP()
    r.initialize(w, h)
    r.set_scene(sc)
    for f in range(frames):
        r.render(eye, d, up, 40.0, f == 0, True)
    mine = r.img.view(np.uint8).reshape(h, w, 4)[..., :3].copy()
    r.close()
    assert np.array_equal(rgb, mine), f"{(rgb != mine).any(axis=2).mean():.5f} of the pixels differ"
    if hip_lib.crt_hip_device_count() >= 2:  # the RCCL tile gather + K8 of the plugin, executed
        rgb2, out2 = _run_bench(tmp_path, 2, w, h, spp, frames)
        assert "x2" in out2
        assert np.array_equal(rgb2, rgb)


_NAVE = ["-eye", "-10", "3", "0.5", "-center", "10", "4", "0", "-up", "0", "1", "0", "-fov", "60"]
CASES = {  # file, material mode, camera arguments (main.cpp:131-152; OBJ / glTF carry no camera and the default (0,0,5) is blind)
    "obj_atrium": ("atrium.obj", "default", _NAVE),        # 16 textures, 24 materials, generated light
    "obj_atrium_wd": ("atrium.obj", "white_diffuse", _NAVE),
    # concave / self-intersecting / degenerate polygons: tinyobjloader's ear clipping on the reference side, its restatement on ours
    "obj_polygons": ("polygons.obj", "default", ["-eye", "2", "2", "12", "-center", "2", "1", "0", "-fov", "50"]),
    "glb_scene": ("scene.glb", "default", ["-eye", "5.5", "8", "24", "-center", "5.5", "6.5", "6", "-fov", "50"]),  # TRS nodes
    "crts_grove": ("grove.crts", "default", []),           # instanced meshes, its own camera (scene.cpp:594-603) and lights
}


@pytest.mark.parametrize("case", list(CASES))
def test_reference_loaded_scene_through_the_plugin_equals_ctypes_path(case, tmp_path, hip_lib):
    """Scene(fname, mode) by the REFERENCE's importer -> plugin -> set_scene(const Scene &) == our importers -> ctypes."""
    if not os.path.exists(BENCH):
        pytest.skip("oracle/_ref/crt_bench not built (`make -C oracle ref`, where /root/reference exists)")
    from chameleonrt_amd.crts_io import load_crts
    from chameleonrt_amd.gltf_io import load_gltf
    from chameleonrt_amd.obj_io import load_obj
    fname, mode, camera = CASES[case]
    path = os.path.join(SCENES, fname)
    w, h, spp, frames = 320, 200, 2, 2
    extra = [path] + (["-mat-mode", "white_diffuse"] if mode == "white_diffuse" else []) + camera
    rgb, out = _run_bench(tmp_path, 1, w, h, spp, frames, extra)
    assert "# Total Triangles:" in out and "HIP wavefront path tracer" in out
    eye, d, up, fovy = _camera_line(out)
    load = {"obj": load_obj, "glb": load_gltf, "crts": load_crts}[fname.rsplit(".", 1)[1]]
    sc = load(path, material_mode=mode)
    sc.samples_per_pixel = spp
    r = RenderHIP()
    r.initialize(w, h)
    r.set_scene(sc)
    for f in range(frames):
        r.render(eye, d, up, fovy, f == 0, True)
    mine = r.img.view(np.uint8).reshape(h, w, 4)[..., :3].copy()
    r.close()
    assert mine.reshape(-1, 3).std(axis=0).max() > 5, "the camera does not see the scene: the comparison would be empty"
    differ = (rgb != mine).any(axis=2).mean()
    # (the .glb case included: its TRS node matrices are multiplied in GLM's order by both importers, gltf_io._mat4_mul)
    assert differ == 0.0, f"{differ:.5f} of the pixels differ"
