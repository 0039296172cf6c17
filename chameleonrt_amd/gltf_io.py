"""glTF 2.0 scene ingest for the headless harness, following what the reference's
importer makes of a `.gltf` / `.glb` file (`Scene::load_gltf`, util/scene.cpp:230-415, on top of
tinygltf, and util/flatten_gltf.cpp):

  * the default scene (index 0 if the file names none) is used; if any of its root nodes has
    children the whole graph is flattened first: every node that carries a mesh, camera or skin
    becomes a root node with the product of the transforms on its path, in depth-first pre-order
    (flatten_gltf.cpp:45-95). A node's transform is its `matrix`, or T * R * S from `translation`,
    `rotation` (x, y, z, w) and `scale` (flatten_gltf.cpp:9-30);
  * a glTF mesh bundles geometry and materials, so each one becomes a Mesh AND the
    ParameterizedMesh with the same index: one Geometry per primitive (triangles only), POSITION,
    TEXCOORD_0 if present, 16- or 32-bit indices; normals are not read; a primitive without a
    material gets -1 and later the generated default material (scene.cpp:257-330, 935-958);
  * images are decoded to 8-bit RGBA, NOT flipped (glTF's texture origin is the top-left corner);
    they are LINEAR unless a material uses them as base colour, which makes them SRGB
    (scene.cpp:329-352, 361-367);
  * materials: base colour, metallic and roughness factors; `baseColorTexture` replaces the colour,
    `metallicRoughnessTexture` supplies metallic from its blue and roughness from its green channel
    (handles encoded per util/texture_channel_mask.h); every other Disney parameter keeps its
    default (scene.cpp:354-390);
  * every root node with a mesh becomes an Instance of that ParameterizedMesh; cameras and lights
    in the file are ignored and the same quad light as for OBJ is generated (scene.cpp:392-414).

Not supported, like the reference: sparse accessors, non-float positions / texture coordinates,
8-bit indices, non-indexed primitives, 16-bit images.
"""
from __future__ import annotations

import base64
import io
import json
import os
import struct
from typing import List, Optional

import numpy as np

from .scene import (LINEAR, SRGB, Geometry, Image, Instance, Mesh, ParameterizedMesh, Scene, disney_material,
                    obj_default_light, textured_param)

_COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT2": 4, "MAT3": 9, "MAT4": 16}


class _Model:
    def __init__(self, path: str):
        self.dir = os.path.dirname(os.path.abspath(path))
        with open(path, "rb") as f:
            data = f.read()
        self.bin_chunk: Optional[bytes] = None
        if data[:4] == b"glTF":  # GLB container: 12-byte header, then chunks (length, type, payload)
            version, length = struct.unpack_from("<II", data, 4)
            if version != 2 or length > len(data):
                raise RuntimeError(f"TinyGLTF Error loading {path} error: bad GLB header")
            off, doc = 12, None
            while off + 8 <= length:
                clen, ctype = struct.unpack_from("<II", data, off)
                payload = data[off + 8:off + 8 + clen]
                if ctype == 0x4E4F534A:  # 'JSON'
                    doc = json.loads(payload.decode("utf-8"))
                elif ctype == 0x004E4942 and self.bin_chunk is None:  # 'BIN\0'
                    self.bin_chunk = payload
                off += 8 + clen
            if doc is None:
                raise RuntimeError(f"TinyGLTF Error loading {path} error: no JSON chunk")
            self.doc = doc
        else:
            self.doc = json.loads(data.decode("utf-8"))
        self.buffers: List[Optional[bytes]] = [None] * len(self.doc.get("buffers", []))

    def _uri(self, uri: str) -> bytes:
        if uri.startswith("data:"):
            return base64.b64decode(uri.split(",", 1)[1])
        with open(os.path.join(self.dir, uri), "rb") as f:
            return f.read()

    def buffer(self, i: int) -> bytes:
        if self.buffers[i] is None:
            b = self.doc["buffers"][i]
            self.buffers[i] = self._uri(b["uri"]) if "uri" in b else self.bin_chunk
            if self.buffers[i] is None:
                raise RuntimeError("glTF buffer without data")
        return self.buffers[i]

    def view_bytes(self, view_id: int) -> bytes:
        v = self.doc["bufferViews"][view_id]
        off = v.get("byteOffset", 0)
        return self.buffer(v["buffer"])[off:off + v["byteLength"]]

    def accessor(self, acc_id: int) -> np.ndarray:
        """Elements of an accessor as (count, components), honouring byteOffset and byteStride
        (util/buffer_view.cpp, gltf_types.cpp)."""
        a = self.doc["accessors"][acc_id]
        if "sparse" in a or "bufferView" not in a:
            raise RuntimeError("sparse / viewless glTF accessors are not supported")
        dt = np.dtype(_COMPONENT[a["componentType"]])
        ncomp = _NCOMP[a["type"]]
        v = self.doc["bufferViews"][a["bufferView"]]
        elem = dt.itemsize * ncomp
        stride = v.get("byteStride", 0) or elem
        raw = self.view_bytes(a["bufferView"])
        start = a.get("byteOffset", 0)
        count = a["count"]
        if count and start + (count - 1) * stride + elem > len(raw):
            raise RuntimeError("glTF accessor runs past its buffer view")
        base = np.frombuffer(raw, dtype=np.uint8)
        rows = np.lib.stride_tricks.as_strided(base[start:], shape=(count, elem), strides=(stride, 1))
        return np.ascontiguousarray(rows).view(dt).reshape(count, ncomp)

    def image(self, i: int) -> Image:
        from .image_io import decode_rgba
        im = self.doc["images"][i]
        data = self._uri(im["uri"]) if "uri" in im else self.view_bytes(im["bufferView"])

        def check(pil):
            if pil.mode.startswith("I;16") or pil.mode == "I":
                raise RuntimeError("Unsupported image pixel type")  # scene.cpp:335-338
        a = decode_rgba(bytes(data), check)  # tinygltf asks stb for 4 components; no flip
        return Image(a.shape[1], a.shape[0], 4, a, LINEAR, im.get("name", ""))


def _mat4_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """a * b as GLM computes it (glm/detail/type_mat4x4.inl, operator*(mat4, mat4), GLM 0.9.9.8 per the reference's
    cmake/glm.cmake): every result column is ((A0 * b0 + A1 * b1) + A2 * b2) + A3 * b3 with A_k the columns of a and b_k the
    entries of b's column, in float32, left to right -- not a BLAS product, whose summation order is its own business. With
    this the flattened node matrices of util/flatten_gltf.cpp:21,26,50 come out bit for bit. Matrices are [row, column]."""
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    r = np.empty((4, 4), np.float32)
    for c in range(4):
        r[:, c] = ((a[:, 0] * b[0, c] + a[:, 1] * b[1, c]) + a[:, 2] * b[2, c]) + a[:, 3] * b[3, c]
    return r


def node_transform(n: dict) -> np.ndarray:
    """read_node_transform (flatten_gltf.cpp:9-30): 4x4 float32, indexed [row, column]."""
    if n.get("matrix"):
        return np.asarray(n["matrix"], np.float32).reshape(4, 4).T.copy()  # glm::make_mat4: column-major
    m = np.eye(4, dtype=np.float32)
    eye = np.eye(4, dtype=np.float32)
    if n.get("scale"):
        # glm::scale(v) = scale(mat4(1), v): column i = identity column i * v[i] -- the zeros of a column take the SIGN of its
        # factor (0 * -2 = -0), and the products below carry those signs on
        sx, sy, sz = (np.float32(v) for v in n["scale"])
        m = np.stack([eye[:, 0] * sx, eye[:, 1] * sy, eye[:, 2] * sz, eye[:, 3]], axis=1).astype(np.float32)
    if n.get("rotation"):
        x, y, z, w = (np.float32(v) for v in n["rotation"])
        one, two = np.float32(1), np.float32(2)
        rot = np.eye(4, dtype=np.float32)  # glm::mat4_cast
        rot[0, 0], rot[1, 0], rot[2, 0] = one - two * (y * y + z * z), two * (x * y + w * z), two * (x * z - w * y)
        rot[0, 1], rot[1, 1], rot[2, 1] = two * (x * y - w * z), one - two * (x * x + z * z), two * (y * z + w * x)
        rot[0, 2], rot[1, 2], rot[2, 2] = two * (x * z + w * y), two * (y * z - w * x), one - two * (x * x + y * y)
        m = _mat4_mul(rot, m)
    if n.get("translation"):
        # glm::translate(v) = translate(mat4(1), v): column 3 = ((I0 * v0 + I1 * v1) + I2 * v2) + I3, the other columns the identity's
        tx, ty, tz = (np.float32(v) for v in n["translation"])
        t = eye.copy()
        t[:, 3] = ((eye[:, 0] * tx + eye[:, 1] * ty) + eye[:, 2] * tz) + eye[:, 3]
        m = _mat4_mul(t, m)
    return m


def _root_nodes(doc: dict) -> List[dict]:
    """The default scene's nodes, flattened like flatten_gltf if the graph has more than one level."""
    scene = doc["scenes"][doc.get("scene", 0) if doc.get("scene", -1) != -1 else 0]
    nodes = doc.get("nodes", [])
    roots = [nodes[i] for i in scene.get("nodes", [])]
    if all(not r.get("children") for r in roots):
        return roots
    flat: List[dict] = []

    def visit(node: dict, parent: np.ndarray):
        transform = _mat4_mul(parent, node_transform(node))
        if "mesh" in node or "camera" in node or "skin" in node:
            out = {k: v for k, v in node.items() if k not in ("children", "scale", "rotation", "translation", "matrix")}
            out["matrix"] = [float(x) for x in transform.T.reshape(16)]
            flat.append(out)
        for c in node.get("children", []):
            visit(nodes[c], transform)

    for r in roots:
        visit(r, np.eye(4, dtype=np.float32))
    return flat


def load_gltf(path: str, material_mode: str = "default", samples_per_pixel: int = 1) -> Scene:
    try:
        model = _Model(path)
    except (ValueError, KeyError, OSError) as exc:
        raise RuntimeError(f"TinyGLTF Error loading {path} error: {exc}") from exc
    doc = model.doc
    default_mode = material_mode == "default"
    sc = Scene(samples_per_pixel=samples_per_pixel, name=path)

    for m in doc.get("meshes", []):
        geoms, material_ids = [], []
        for p in m["primitives"]:
            material_ids.append(p.get("material", -1) if default_mode else -1)
            if p.get("mode", 4) != 4:
                raise RuntimeError("Unsupported primitive mode! Only triangles are supported")
            pos = model.accessor(p["attributes"]["POSITION"])
            if pos.dtype != np.float32 or pos.shape[1] != 3:
                raise RuntimeError("POSITION must be float VEC3")
            uvs = None
            if "TEXCOORD_0" in p["attributes"]:
                uvs = model.accessor(p["attributes"]["TEXCOORD_0"])
                if uvs.dtype != np.float32 or uvs.shape[1] != 2:
                    raise RuntimeError("TEXCOORD_0 must be float VEC2")
            if "indices" not in p:
                raise RuntimeError("non-indexed primitives are not supported")
            idx = model.accessor(p["indices"])
            if idx.dtype not in (np.uint16, np.uint32):
                raise RuntimeError("Unsupported index component type")
            idx = idx.reshape(-1)
            idx = idx[:3 * (len(idx) // 3)].astype(np.uint32).reshape(-1, 3)
            geoms.append(Geometry(pos.copy(), idx, None if uvs is None else uvs.copy()))
        sc.parameterized_meshes.append(ParameterizedMesh(len(sc.meshes), material_ids))
        sc.meshes.append(Mesh(geoms))

    if default_mode:
        sc.textures = [model.image(i) for i in range(len(doc.get("images", [])))]
        tex_source = [t.get("source", -1) for t in doc.get("textures", [])]
        for m in doc.get("materials", []):
            pbr = m.get("pbrMetallicRoughness", {})
            mat = disney_material()
            mat[0:3] = np.float32(pbr.get("baseColorFactor", [1.0, 1.0, 1.0, 1.0])[:3])
            mat[3] = np.float32(pbr.get("metallicFactor", 1.0))
            mat[5] = np.float32(pbr.get("roughnessFactor", 1.0))
            if "baseColorTexture" in pbr:
                tid = tex_source[pbr["baseColorTexture"]["index"]]
                sc.textures[tid].color_space = SRGB
                mat[0] = textured_param(tid)
            if "metallicRoughnessTexture" in pbr:  # glTF: metallic is the blue channel, roughness the green one
                tid = tex_source[pbr["metallicRoughnessTexture"]["index"]]
                sc.textures[tid].color_space = LINEAR
                mat[3] = textured_param(tid, 2)
                mat[5] = textured_param(tid, 1)
            sc.materials.append(mat)

    for n in _root_nodes(doc):
        if "mesh" in n:
            sc.instances.append(Instance(node_transform(n).T.reshape(16).copy(), int(n["mesh"])))

    if any(mid == -1 for p in sc.parameterized_meshes for mid in p.material_ids):  # validate_materials
        default_id = len(sc.materials)
        sc.materials.append(disney_material())
        for p in sc.parameterized_meshes:
            p.material_ids = [default_id if mid == -1 else mid for mid in p.material_ids]
    sc.lights = [obj_default_light()]  # scene.cpp:404-414: same generated light as for OBJ
    return sc
