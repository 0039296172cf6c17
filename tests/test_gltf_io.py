"""CPU: glTF 2.0 scene ingest (chameleonrt_amd/gltf_io.py) against what the reference's importer
makes of such files (util/scene.cpp:230-415, util/flatten_gltf.cpp). The files are written by hand
here (JSON + base64 / GLB), independently of any writer."""
import base64
import io
import json
import struct

import numpy as np
import pytest

from chameleonrt_amd.gltf_io import load_gltf, node_transform
from chameleonrt_amd.scene import LINEAR, SRGB, obj_default_light


def _png(rgba):
    from PIL import Image as PILImage
    buf = io.BytesIO()
    PILImage.fromarray(rgba, "RGBA").save(buf, format="PNG")
    return buf.getvalue()


class Builder:
    """Minimal glTF writer for the tests: one buffer, views and accessors appended in order."""

    def __init__(self):
        self.blob = bytearray()
        self.doc = {"asset": {"version": "2.0"}, "bufferViews": [], "accessors": [], "meshes": [], "nodes": [],
                    "scenes": [{"nodes": []}], "scene": 0}

    def view(self, raw: bytes, stride=None):
        while len(self.blob) % 4:
            self.blob.append(0)
        v = {"buffer": 0, "byteOffset": len(self.blob), "byteLength": len(raw)}
        if stride:
            v["byteStride"] = stride
        self.blob.extend(raw)
        self.doc["bufferViews"].append(v)
        return len(self.doc["bufferViews"]) - 1

    def accessor(self, view, ctype, kind, count, offset=0):
        self.doc["accessors"].append({"bufferView": view, "componentType": ctype, "type": kind, "count": count,
                                      "byteOffset": offset})
        return len(self.doc["accessors"]) - 1

    def finish(self, path, glb=False):
        self.doc["buffers"] = [{"byteLength": len(self.blob)}]
        if glb:
            js = json.dumps(self.doc).encode()
            js += b" " * (-len(js) % 4)
            blob = bytes(self.blob) + b"\0" * (-len(self.blob) % 4)
            total = 12 + 8 + len(js) + 8 + len(blob)
            with open(path, "wb") as f:
                f.write(b"glTF" + struct.pack("<II", 2, total))
                f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
                f.write(struct.pack("<II", len(blob), 0x004E4942) + blob)
        else:
            self.doc["buffers"][0]["uri"] = "data:application/octet-stream;base64," + base64.b64encode(bytes(self.blob)).decode()
            with open(path, "w") as f:
                json.dump(self.doc, f)


QUAD_POS = np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]])
QUAD_UV = np.float32([[0, 0], [1, 0], [0, 1], [1, 1]])
QUAD_IDX = [0, 1, 2, 2, 1, 3]


def _scene_file(path, glb=False):
    b = Builder()
    # primitive 0: interleaved position + uv (byteStride 20), 16-bit indices
    inter = np.concatenate([QUAD_POS, QUAD_UV], axis=1).astype(np.float32)
    vi = b.view(inter.tobytes(), stride=20)
    a_pos, a_uv = b.accessor(vi, 5126, "VEC3", 4), b.accessor(vi, 5126, "VEC2", 4, offset=12)
    a_i16 = b.accessor(b.view(np.uint16(QUAD_IDX).tobytes()), 5123, "SCALAR", 6)
    # primitive 1: separate positions, 32-bit indices, no uvs, no material
    a_pos2 = b.accessor(b.view((QUAD_POS * 3).tobytes()), 5126, "VEC3", 4)
    a_i32 = b.accessor(b.view(np.uint32(QUAD_IDX).tobytes()), 5125, "SCALAR", 6)
    b.doc["meshes"] = [{"primitives": [{"attributes": {"POSITION": a_pos, "TEXCOORD_0": a_uv}, "indices": a_i16, "material": 0},
                                       {"attributes": {"POSITION": a_pos2}, "indices": a_i32}]},
                       {"primitives": [{"attributes": {"POSITION": a_pos2}, "indices": a_i32, "material": 1}]}]
    tex = np.zeros((2, 2, 4), np.uint8)
    tex[0, 0] = [200, 10, 20, 255]  # top-left texel of the file
    png = _png(tex)
    b.doc["images"] = [{"name": "albedo", "bufferView": b.view(png), "mimeType": "image/png"},
                       {"name": "orm", "uri": "data:image/png;base64," + base64.b64encode(png).decode()},
                       {"name": "unused", "uri": "data:image/png;base64," + base64.b64encode(png).decode()}]
    b.doc["textures"] = [{"source": 1}, {"source": 0}]  # texture index != image index
    b.doc["materials"] = [{"pbrMetallicRoughness": {"baseColorFactor": [0.5, 0.6, 0.7, 1.0], "metallicFactor": 0.25,
                                                     "roughnessFactor": 0.75, "baseColorTexture": {"index": 1},
                                                     "metallicRoughnessTexture": {"index": 0}}},
                          {"pbrMetallicRoughness": {"baseColorFactor": [0.1, 0.2, 0.3, 1.0]}}]
    b.doc["nodes"] = [{"mesh": 0, "translation": [1, 2, 3], "scale": [2, 2, 2]},
                      {"mesh": 1, "matrix": [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 7, 8, 9, 1]},
                      {"name": "empty"}]
    b.doc["scenes"][0]["nodes"] = [0, 1, 2]
    b.finish(path, glb)
    return tex


@pytest.mark.parametrize("glb", [False, True], ids=["gltf", "glb"])
def test_loader_follows_the_reference(tmp_path, glb):
    p = str(tmp_path / ("scene.glb" if glb else "scene.gltf"))
    tex = _scene_file(p, glb)
    sc = load_gltf(p, samples_per_pixel=2)
    # a glTF mesh is a Mesh and the ParameterizedMesh of the same index; one Geometry per primitive
    assert [len(m.geometries) for m in sc.meshes] == [2, 1]
    g0, g1 = sc.meshes[0].geometries
    assert np.array_equal(g0.vertices, QUAD_POS) and np.array_equal(g0.uvs, QUAD_UV)  # de-interleaved
    assert np.array_equal(g0.indices, np.uint32(QUAD_IDX).reshape(2, 3)) and g0.indices.dtype == np.uint32
    assert g1.uvs is None and np.array_equal(g1.vertices, QUAD_POS * 3)
    # the primitive without a material gets the generated default one (validate_materials)
    assert [(pm.mesh_id, pm.material_ids) for pm in sc.parameterized_meshes] == [(0, [0, 2]), (1, [1])]
    assert len(sc.materials) == 3 and np.allclose(sc.materials[2][:3], 0.9)
    # images: RGBA8, not flipped; sRGB only where used as base colour (via texture -> source)
    assert [t.color_space for t in sc.textures] == [SRGB, LINEAR, LINEAR]
    assert np.array_equal(np.asarray(sc.textures[0].img).reshape(2, 2, 4), tex)
    bits = lambda x: int(np.float32(x).view(np.uint32))
    m0, m1 = sc.materials[0], sc.materials[1]
    assert bits(m0[0]) == 0x80000000 | 0 and (m0[1], m0[2]) == (np.float32(0.6), np.float32(0.7))
    assert bits(m0[3]) == 0x80000000 | (2 << 29) | 1 and bits(m0[5]) == 0x80000000 | (1 << 29) | 1  # blue, green
    assert np.allclose(m1[:3], [0.1, 0.2, 0.3]) and m1[3] == 1.0 and m1[5] == 1.0  # glTF factor defaults
    assert m1[4] == 0.0 and m1[12] == 1.5  # everything else: DisneyMaterial defaults
    # instances: T * R * S, column-major; nodes without a mesh are ignored
    assert [i.parameterized_mesh_id for i in sc.instances] == [0, 1]
    t0 = sc.instances[0].transform.reshape(4, 4).T
    assert np.allclose(t0, [[2, 0, 0, 1], [0, 2, 0, 2], [0, 0, 2, 3], [0, 0, 0, 1]])
    assert np.allclose(sc.instances[1].transform[12:15], [7, 8, 9])
    assert len(sc.lights) == 1 and np.array_equal(sc.lights[0], obj_default_light()) and not sc.cameras
    # -mat-mode white_diffuse: no images, no materials
    wd = load_gltf(p, material_mode="white_diffuse")
    assert not wd.textures and len(wd.materials) == 1
    assert [pm.material_ids for pm in wd.parameterized_meshes] == [[0, 0], [0]]


def test_rotation_follows_glm():
    s = np.float32(np.sqrt(0.5))
    m = node_transform({"rotation": [0.0, 0.0, float(s), float(s)]})  # 90 degrees about z
    assert np.allclose(m[:3, :3], [[0, -1, 0], [1, 0, 0], [0, 0, 1]], atol=1e-6)
    m = node_transform({"translation": [1, 0, 0], "rotation": [0.0, 0.0, float(s), float(s)], "scale": [2, 1, 1]})
    assert np.allclose(m @ np.float32([1, 0, 0, 1]), [1, 2, 0, 1], atol=1e-6)  # scale, then rotate, then translate


def test_multi_level_graph_is_flattened(tmp_path):
    """flatten_gltf.cpp:45-95: depth-first pre-order, transforms multiplied down the path, only
    nodes with a mesh survive as instances."""
    b = Builder()
    a_pos = b.accessor(b.view(QUAD_POS.tobytes()), 5126, "VEC3", 4)
    a_idx = b.accessor(b.view(np.uint16(QUAD_IDX).tobytes()), 5123, "SCALAR", 6)
    b.doc["meshes"] = [{"primitives": [{"attributes": {"POSITION": a_pos}, "indices": a_idx}]}] * 2
    b.doc["nodes"] = [{"translation": [10, 0, 0], "children": [1, 3]},        # 0: group
                      {"mesh": 0, "scale": [2, 2, 2], "children": [2]},       # 1
                      {"mesh": 1, "translation": [0, 1, 0]},                  # 2: child of 1
                      {"mesh": 1, "translation": [0, 0, 5]},                  # 3: child of 0
                      {"mesh": 0}]                                            # 4: second root
    b.doc["scenes"][0]["nodes"] = [0, 4]
    p = str(tmp_path / "tree.gltf")
    b.finish(p)
    sc = load_gltf(p)
    assert [i.parameterized_mesh_id for i in sc.instances] == [0, 1, 1, 0]
    tr = [i.transform.reshape(4, 4).T for i in sc.instances]
    assert np.allclose(tr[0], [[2, 0, 0, 10], [0, 2, 0, 0], [0, 0, 2, 0], [0, 0, 0, 1]])
    assert np.allclose(tr[1][:3, 3], [10, 2, 0]) and np.allclose(np.diag(tr[1])[:3], 2)  # parent scale applies
    assert np.allclose(tr[2][:3, 3], [10, 0, 5]) and np.allclose(tr[3], np.eye(4))


def test_unsupported_content_is_rejected(tmp_path):
    def file_with(prim_patch=None, idx_ctype=5123, idx_dtype=np.uint16):
        b = Builder()
        a_pos = b.accessor(b.view(QUAD_POS.tobytes()), 5126, "VEC3", 4)
        a_idx = b.accessor(b.view(idx_dtype(QUAD_IDX).tobytes()), idx_ctype, "SCALAR", 6)
        prim = {"attributes": {"POSITION": a_pos}, "indices": a_idx}
        prim.update(prim_patch or {})
        b.doc["meshes"] = [{"primitives": [prim]}]
        b.doc["nodes"] = [{"mesh": 0}]
        b.doc["scenes"][0]["nodes"] = [0]
        p = str(tmp_path / "bad.gltf")
        b.finish(p)
        return p

    with pytest.raises(RuntimeError, match="Only triangles"):
        load_gltf(file_with({"mode": 1}))
    with pytest.raises(RuntimeError, match="index component type"):
        load_gltf(file_with(idx_ctype=5121, idx_dtype=np.uint8))
    p = str(tmp_path / "garbage.gltf")
    with open(p, "w") as f:
        f.write("{ not json")
    with pytest.raises(RuntimeError, match="TinyGLTF Error"):
        load_gltf(p)


def test_loaded_scene_renders_in_the_oracle(tmp_path, oracle):
    """End to end: a glTF file goes through the same scene packing as every other scene."""
    from chameleonrt_amd.scene import Camera
    from tests.parity import camera_of
    p = str(tmp_path / "scene.gltf")
    _scene_file(p)
    sc = load_gltf(p)
    sc.cameras = [Camera(np.float32([2, 2, 12]), np.float32([2, 3, 0]), np.float32([0, 1, 0]), 45.0)]
    o = oracle.OracleRenderer(sc, 48, 32)
    st = o.render(*camera_of(sc), True)
    a = o.accum()
    assert np.isfinite(a).all() and int(st.shadow_rays) > 0  # some primary rays hit the quads and sample the light
