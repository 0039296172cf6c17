"""CPU: synthetic scene generators are deterministic and have the stated shape."""
import numpy as np

from chameleonrt_amd import scenes
from chameleonrt_amd.scene import PackedScene, textured_param


def _digest(sc):
    h = 0
    for m in sc.meshes:
        for g in m.geometries:
            h ^= hash((g.vertices.tobytes(), g.indices.tobytes(), None if g.uvs is None else g.uvs.tobytes()))
    return h


def test_cornell_is_34_triangles():
    sc = scenes.cornell()
    assert sc.total_tris() == 34 and len(sc.meshes[0].geometries) == 7
    assert len(sc.lights) == 1 and sc.lights[0].shape == (20,)
    # quirk Q14: emission 20, normal = normalize(0.5, -0.8, -0.5), position = -10 * normal, w = h = 5
    l = sc.lights[0]
    assert np.allclose(l[0:3], 20) and np.allclose(l[4:7], -10 * l[8:11]) and l[15] == 5 and l[19] == 5


def test_generators_are_deterministic():
    a = scenes.sponza_like(detail=0.02, tex_size=32)
    b = scenes.sponza_like(detail=0.02, tex_size=32)
    assert _digest(a) == _digest(b)
    assert all(np.array_equal(x.img, y.img) for x, y in zip(a.textures, b.textures))
    assert all(np.array_equal(x, y) for x, y in zip(a.materials, b.materials))
    c = scenes.instanced_grove()
    d = scenes.instanced_grove()
    assert _digest(c) == _digest(d) and len(c.instances) == 65


def test_sponza_like_shape():
    sc = scenes.sponza_like(tex_size=64)
    assert 240_000 < sc.total_tris() < 285_000
    assert len(sc.textures) == 16 and len(sc.materials) == 24
    uv = np.concatenate([g.uvs for g in sc.meshes[0].geometries])
    assert uv.min() < -1.5 and uv.max() > 2.5  # wrap + negative texture coordinates are exercised


def test_texture_handle_encoding():
    f = textured_param(5, 2)
    bits = int(np.array([f], np.float32).view(np.uint32)[0])
    assert bits & 0x80000000 and (bits >> 29) & 3 == 2 and bits & 0x1FFFFFFF == 5


def test_packed_scene_layout():
    sc = scenes.instanced_grove()
    p = PackedScene(sc)
    d = p.desc
    assert d.n_meshes == 4 and d.n_instances == 65 and d.n_geometries == 6
    assert d.meshes[1].first_geometry == 1 and d.meshes[1].n_geometries == 2
    assert d.samples_per_pixel == sc.samples_per_pixel


def test_real_assets_are_used_when_the_scene_directory_has_them(tmp_path, monkeypatch):
    """the synthetic stand-ins are generated "unless real assets are found under $CRT_SCENE_DIR" -- a
    directory per asset (cornell, sponza, rungholt, san-miguel) holding the scene file and, optionally, camera.json."""
    import json
    import os
    from chameleonrt_amd.obj_io import load_obj, save_obj
    d = tmp_path / "cornell"
    d.mkdir()
    save_obj(scenes.cornell(), str(d / "CornellBox-Original.obj"))
    monkeypatch.setenv("CRT_SCENE_DIR", str(tmp_path))
    sc, w, h, spp = scenes.make_workload("C1")
    assert sc.name == "real:CornellBox-Original.obj" and (w, h, spp) == (512, 512, 1) and sc.samples_per_pixel == 1
    ref = load_obj(str(d / "CornellBox-Original.obj"))
    assert sc.total_tris() == ref.total_tris() == 34
    assert np.allclose(sc.cameras[0].position, [0, 0, 5]) and sc.cameras[0].fov_y == 65.0  # main.cpp:122-125
    (d / "camera.json").write_text(json.dumps({"eye": [0, 1, 3.4], "center": [0, 1, 0], "fovy": 40}))
    sc, _, _, _ = scenes.make_workload("C1")
    assert np.allclose(sc.cameras[0].position, [0, 1, 3.4]) and sc.cameras[0].fov_y == 40.0
    # no directory for the asset: the stand-in, as before
    sc3, _, _, _ = scenes.make_workload("C2", detail=0.01, tex_size=8)
    assert not sc3.name.startswith("real:")
    monkeypatch.delenv("CRT_SCENE_DIR")
    assert scenes.make_workload("C1")[0].name == "cornell"
