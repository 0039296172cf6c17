"""CPU: CRTS scene ingest (chameleonrt_amd/crts_io.py) against what the reference's loader makes
of such a file (util/scene.cpp:417-624). A hand-written file checks the loader on its own (object
kinds, textured parameters, ParameterizedMesh look-up, generated light, -mat-mode); a round trip
through save_crts checks the writer; and a round-tripped scene must render like the original in
the oracle."""
import io
import json
import struct

import numpy as np
import pytest

from chameleonrt_amd import scenes
from chameleonrt_amd.crts_io import crts_default_light, load_crts, save_crts
from chameleonrt_amd.scene import LINEAR, SRGB


def _png(rgba):
    from PIL import Image as PILImage
    buf = io.BytesIO()
    PILImage.fromarray(rgba, "RGBA").save(buf, format="PNG")
    return buf.getvalue()


def _write(path, header, blob):
    h = json.dumps(header).encode()
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(h)) + h + bytes(blob))


def _handmade(path, with_light=True):
    blob = bytearray()
    views = []

    def add(a, t):
        views.append({"byte_offset": len(blob), "byte_length": a.nbytes, "type": t})
        blob.extend(a.tobytes())
        return len(views) - 1

    pos = np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]])
    idx = np.uint32([[0, 1, 2], [2, 1, 3]])
    uv = np.float32([[0, 0], [1, 0], [0, 1], [1, 1]])
    tex = np.zeros((2, 3, 4), np.uint8)
    tex[0, 0] = [255, 0, 0, 255]  # top-left texel of the FILE
    tex[1, 2] = [0, 0, 255, 128]
    meshes = [{"positions": add(pos, "VEC3_F32"), "indices": add(idx, "VEC3_U32"), "texcoords": add(uv, "VEC2_F32")},
              {"positions": add(pos * 2, "VEC3_F32"), "indices": add(idx, "VEC3_U32")}]
    images = [{"name": "albedo", "view": add(np.frombuffer(_png(tex), np.uint8), "UINT_8"), "color_space": "SRGB"},
              {"name": "params", "view": add(np.frombuffer(_png(tex), np.uint8), "UINT_8"), "color_space": "LINEAR"}]
    base = {"metallic": 0.1, "specular": 0.2, "roughness": 0.3, "specular_tint": 0.4, "anisotropic": 0.5, "sheen": 0.6,
            "sheen_tint": 0.7, "clearcoat": 0.8, "clearcoat_roughness": 0.9, "ior": 1.45, "transmission": 0.25}
    materials = [dict(base, base_color=[0.1, 0.2, 0.3]),
                 dict(base, base_color=[1, 1, 1], base_color_texture=0, roughness_texture={"texture": 1, "channel": 2})]
    eye = np.eye(4, dtype=np.float32)
    shifted = eye.copy()
    shifted[:3, 3] = [5, 6, 7]
    col_major = lambda m: [float(x) for x in m.T.reshape(16)]
    objects = [{"type": "MESH", "matrix": col_major(eye), "mesh": 0, "material": 1},
               {"type": "MESH", "matrix": col_major(shifted), "mesh": 0, "material": 1},  # same ParameterizedMesh
               {"type": "MESH", "matrix": col_major(shifted), "mesh": 0, "material": 0},  # new one: other material
               {"type": "MESH", "matrix": col_major(eye), "mesh": 1, "material": 0}]
    cam = eye.copy()
    cam[:3, 3] = [0, 1, 4]
    objects.append({"type": "CAMERA", "matrix": col_major(cam), "fov_y": 59.0})
    if with_light:
        lm = np.diag(np.float32([2, 3, 4, 1]))  # scaled axes: the loader normalises them
        lm[:3, 3] = [0, 5, 0]
        objects.append({"type": "LIGHT", "matrix": col_major(lm), "color": [1.0, 0.5, 0.25], "energy": 8.0, "size": [1.5, 0.5]})
    _write(path, {"buffer_views": views, "meshes": meshes, "images": images, "materials": materials, "objects": objects}, blob)
    return tex


def test_loader_follows_the_reference(tmp_path):
    p = str(tmp_path / "hand.crts")
    tex = _handmade(p)
    sc = load_crts(p, samples_per_pixel=3)
    assert sc.samples_per_pixel == 3
    # one geometry per mesh; uvs only where the file has them
    assert [len(m.geometries) for m in sc.meshes] == [1, 1]
    assert sc.meshes[0].geometries[0].uvs.shape == (4, 2) and sc.meshes[1].geometries[0].uvs is None
    assert sc.meshes[1].geometries[0].vertices.max() == 2.0 and sc.meshes[0].geometries[0].indices.dtype == np.uint32
    # images: forced to 4 channels, flipped (row 0 of the Image is the BOTTOM row of the file), colour space kept
    assert [(t.width, t.height, t.channels, t.color_space) for t in sc.textures] == [(3, 2, 4, SRGB), (3, 2, 4, LINEAR)]
    assert np.array_equal(np.asarray(sc.textures[0].img).reshape(2, 3, 4), tex[::-1])
    # materials: renamed parameters, texture handles in the float bits (util/texture_channel_mask.h)
    m0, m1 = sc.materials
    assert np.allclose(m0[:14], [0.1, 0.2, 0.3, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.45, 0.25])
    bits = lambda x: int(np.float32(x).view(np.uint32))
    assert bits(m1[0]) == 0x80000000 | 0 and bits(m1[5]) == 0x80000000 | (2 << 29) | 1
    assert m1[1] == 1.0 and m1[2] == 1.0
    # instances: ParameterizedMesh looked up by (mesh, material) in order of first use
    assert [(pm.mesh_id, pm.material_ids) for pm in sc.parameterized_meshes] == [(0, [1]), (0, [0]), (1, [0])]
    assert [i.parameterized_mesh_id for i in sc.instances] == [0, 0, 1, 2]
    assert np.array_equal(sc.instances[1].transform[12:15], [5, 6, 7])  # column-major translation
    # camera: position = column 3, looks down -z at distance 10, fov_y / 1.18
    c = sc.cameras[0]
    assert np.allclose(c.position, [0, 1, 4]) and np.allclose(c.center, [0, 1, -6]) and np.allclose(c.up, [0, 1, 0])
    assert c.fov_y == pytest.approx(59.0 / 1.18, rel=1e-6)
    # light: emission = color * energy (w = 1), faces -z, axes normalised, size from the file
    l = sc.lights[0]
    assert len(sc.lights) == 1
    assert np.allclose(l[0:4], [8, 4, 2, 1]) and np.allclose(l[4:8], [0, 5, 0, 1]) and np.allclose(l[8:11], [0, 0, -1])
    assert np.allclose(l[12:15], [1, 0, 0]) and np.allclose(l[16:19], [0, 1, 0]) and (l[15], l[19]) == (1.5, 0.5)


def test_generated_light_and_material_mode(tmp_path):
    p = str(tmp_path / "nolight.crts")
    _handmade(p, with_light=False)
    sc = load_crts(p)
    assert len(sc.lights) == 1 and np.array_equal(sc.lights[0], crts_default_light())
    assert np.allclose(sc.lights[0][0:4], 10.0) and (sc.lights[0][15], sc.lights[0][19]) == (5.0, 5.0)
    # -mat-mode white_diffuse: no materials are read, every object gets the one default material, and
    # objects that differ only in material now share a ParameterizedMesh
    wd = load_crts(p, material_mode="white_diffuse")
    assert len(wd.materials) == 1 and np.allclose(wd.materials[0][:3], 0.9)
    assert [(pm.mesh_id, pm.material_ids) for pm in wd.parameterized_meshes] == [(0, [0]), (1, [0])]
    assert [i.parameterized_mesh_id for i in wd.instances] == [0, 0, 0, 1]
    assert len(wd.textures) == 2  # images are still loaded (scene.cpp:481-511)


def test_malformed_files_are_rejected(tmp_path):
    p = str(tmp_path / "bad.crts")
    with open(p, "wb") as f:
        f.write(struct.pack("<Q", 1 << 40) + b"{}")
    with pytest.raises(ValueError):
        load_crts(p)
    _write(p, {"buffer_views": [], "meshes": [], "images": [], "materials": [],
               "objects": [{"type": "EMPTY", "matrix": [0.0] * 16}]}, b"")
    with pytest.raises(RuntimeError):
        load_crts(p)
    _write(p, {"buffer_views": [{"byte_offset": 0, "byte_length": 64, "type": "VEC3_F32"}],
               "meshes": [{"positions": 0, "indices": 0}], "images": [], "materials": [], "objects": []}, b"\0" * 8)
    with pytest.raises(ValueError):
        load_crts(p)


@pytest.mark.parametrize("name", ["cornell", "instanced_grove"])
def test_round_trip_renders_like_the_original(name, tmp_path, oracle):
    """save_crts -> load_crts keeps geometry, instancing, materials (textured parameters included),
    textures, lights and the camera: the oracle renders the same image. Meshes with several
    geometries come back as one mesh per geometry, which changes Embree's geomID / instance
    numbering but not a single hit."""
    from tests.parity import camera_of, compare_images
    sc = getattr(scenes, name)()
    p = str(tmp_path / f"{name}.crts")
    save_crts(sc, p)
    back = load_crts(p, samples_per_pixel=sc.samples_per_pixel)
    assert back.total_tris() == sc.total_tris() and len(back.textures) == len(sc.textures)
    assert len(back.materials) == len(sc.materials) and len(back.lights) == len(sc.lights)
    for a, b in zip(sc.materials, back.materials):
        assert np.array_equal(np.asarray(a, np.float32)[:14].view(np.uint32), np.asarray(b, np.float32)[:14].view(np.uint32))
    for a, b in zip(sc.textures, back.textures):
        ia = np.asarray(a.img).reshape(a.height, a.width, a.channels)
        ib = np.asarray(b.img).reshape(b.height, b.width, 4)
        assert b.color_space == a.color_space and np.array_equal(ia[..., :min(3, a.channels)], ib[..., :min(3, a.channels)])
    w, h = 64, 40
    images = []
    for s in (sc, back):
        o = oracle.OracleRenderer(s, w, h)
        o.render(*camera_of(sc), True)  # the ORIGINAL camera for both: the file stores it as a matrix
        images.append(o.accum())
    diverged, mean_rel = compare_images(images[1], images[0])
    assert diverged <= 2.0 / (w * h) and mean_rel <= 1e-5
    # and the camera survives the matrix form up to rounding
    e0, d0, u0, f0 = camera_of(sc)
    e1, d1, u1, f1 = camera_of(back)
    assert np.allclose(e0, e1, atol=1e-6) and np.allclose(d0, d1, atol=1e-5) and f1 == pytest.approx(f0, rel=1e-5)
