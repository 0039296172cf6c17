"""CRTS scene ingest for the headless harness: ChameleonRT's own binary scene
format, written by its Blender exporter and read by `Scene::load_crts` (util/scene.cpp:417-624).

File layout: `uint64 json_size | json header | binary blob`. The header holds

    buffer_views  [{byte_offset, byte_length, type}]         offsets into the blob; `type` names an
                                                              element type (util/gltf_types.cpp:144-215)
    meshes        [{positions, indices, texcoords?}]          buffer-view ids; ONE geometry per mesh
    images        [{name, view, color_space}]                 encoded image files (PNG/JPEG bytes),
                                                              loaded flipped and forced to 4 channels
    materials     [{base_color, base_color_texture?, metallic, metallic_texture?: {texture, channel}, ...}]
    objects       [{type: MESH | LIGHT | CAMERA, matrix[16] column-major, ...}]

and `load_crts` makes of it exactly what the reference does:
  * every mesh becomes a Mesh with one Geometry (scene.cpp:431-479; normals are not read);
  * MESH objects become Instances; their ParameterizedMesh is looked up by the pair
    (mesh id, material id) in order of first use (scene.cpp:557-583);
  * material parameters may be textured: the handle is encoded in the float's bits
    (util/texture_channel_mask.h); the exporter's names `anisotropic`, `clearcoat_roughness` and
    `transmission` map to anisotropy, clearcoat_gloss and specular_transmission (scene.cpp:513-553);
  * LIGHT objects are quad lights spanned by the matrix' x / y columns, facing -z, with
    emission = color * energy (scene.cpp:584-594); CAMERA objects look down -z with
    fov_y / 1.18 (scene.cpp:595-606);
  * without any light one is generated: emission 10, 5 x 5, like the OBJ one otherwise
    (scene.cpp:611-623); `-mat-mode white_diffuse` drops materials and gives every object the
    default one (scene.cpp:513, 569-571, 935-958).

`save_crts` writes a Scene in that format (a mesh with several geometries becomes one CRTS mesh
per geometry, each instanced with its own material), so the synthetic benchmark scenes can be fed
to a real ChameleonRT build and a file it wrote can be read back.
"""
from __future__ import annotations

import io
import json
import struct
from typing import Dict, List, Tuple

import numpy as np

from .scene import (LINEAR, SRGB, Camera, Geometry, Image, Instance, Mesh, ParameterizedMesh, Scene, _normalize,
                    disney_material, ortho_basis, quad_light, textured_param)

# element types a buffer view can name (util/gltf_types.cpp:144-215): numpy dtype and components
_SCALARS = {"I8": np.int8, "U8": np.uint8, "I16": np.int16, "U16": np.uint16, "I32": np.int32, "U32": np.uint32,
            "F32": np.float32, "F64": np.float64}
_DTYPES: Dict[str, Tuple[type, int]] = {
    "INT_8": (np.int8, 1), "UINT_8": (np.uint8, 1), "INT_16": (np.int16, 1), "UINT_16": (np.uint16, 1),
    "INT_32": (np.int32, 1), "UINT_32": (np.uint32, 1), "FLOAT_32": (np.float32, 1), "FLOAT_64": (np.float64, 1)}
for _n in (2, 3, 4):
    for _k, _t in _SCALARS.items():
        _DTYPES[f"VEC{_n}_{_k}"] = (_t, _n)

# (name in the file, slot in the 16-float DisneyMaterial), scene.cpp:541-553
_FLOAT_PARAMS = [("metallic", 3), ("specular", 4), ("roughness", 5), ("specular_tint", 6), ("anisotropic", 7),
                 ("sheen", 8), ("sheen_tint", 9), ("clearcoat", 10), ("clearcoat_roughness", 11), ("ior", 12),
                 ("transmission", 13)]


def crts_default_light() -> np.ndarray:
    """The light load_crts generates for a file without one (scene.cpp:611-623)."""
    n = _normalize([0.5, -0.8, -0.5])
    v_x, v_y = ortho_basis(n)
    l = quad_light([10.0, 10.0, 10.0, 10.0], (np.float32(-10.0) * n).astype(np.float32), n, v_x, v_y, 5.0, 5.0)
    l[7] = np.float32(-0.0)  # position = -10.f * vec4(normal, 0): w is minus zero (scene.cpp:617)
    return l


def _view(header: dict, blob: memoryview, view_id: int, want: Tuple[type, int]) -> np.ndarray:
    v = header["buffer_views"][view_id]
    if v["type"] not in _DTYPES:
        raise ValueError(f"unknown buffer view type {v['type']!r}")
    dt, ncomp = _DTYPES[v["type"]]
    raw = blob[v["byte_offset"]:v["byte_offset"] + v["byte_length"]]
    if len(raw) != v["byte_length"]:
        raise ValueError("buffer view runs past the end of the file")
    a = np.frombuffer(raw, dtype=dt)
    if want[1] > 1:
        # Accessor<vecN> steps by the view's element stride and reads N components of the target type
        if (np.dtype(dt).itemsize, ncomp) != (np.dtype(want[0]).itemsize, want[1]):
            raise ValueError(f"buffer view {view_id} is {v['type']}, expected {want[1]} x {np.dtype(want[0]).name}")
        a = a.view(want[0]).reshape(-1, want[1])
    return a


def _decode_image(data: bytes, name: str, color_space: str) -> Image:
    from .image_io import decode_rgba
    try:
        a = decode_rgba(bytes(data))  # stbi_load_from_memory(..., 4)
    except Exception as exc:  # the reference throws too (scene.cpp:497-500)
        raise RuntimeError(f"Failed to load {name}") from exc
    a = a[::-1].copy()  # stbi_set_flip_vertically_on_load(1)
    return Image(a.shape[1], a.shape[0], 4, a, LINEAR if color_space == "LINEAR" else SRGB, name)


def load_crts(path: str, material_mode: str = "default", samples_per_pixel: int = 1) -> Scene:
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 8:
        raise ValueError("not a CRTS file")
    (json_size,) = struct.unpack_from("<Q", data, 0)
    if 8 + json_size > len(data):
        raise ValueError("CRTS header runs past the end of the file")
    header = json.loads(data[8:8 + json_size].decode("utf-8"))
    blob = memoryview(data)[8 + json_size:]
    default_mode = material_mode == "default"
    sc = Scene(samples_per_pixel=samples_per_pixel, name=path)

    for m in header.get("meshes", []):
        pos = _view(header, blob, m["positions"], (np.float32, 3)).astype(np.float32).copy()
        idx = _view(header, blob, m["indices"], (np.uint32, 3)).astype(np.uint32).copy()
        uvs = None
        if "texcoords" in m:
            uvs = _view(header, blob, m["texcoords"], (np.float32, 2)).astype(np.float32).copy()
        sc.meshes.append(Mesh([Geometry(pos, idx, uvs)]))

    for img in header.get("images", []):
        raw = _view(header, blob, img["view"], (np.uint8, 1))
        sc.textures.append(_decode_image(raw.tobytes(), img["name"], img["color_space"]))

    if default_mode:
        for m in header.get("materials", []):
            mat = disney_material()
            mat[0:3] = np.asarray(m["base_color"], np.float32)[:3]
            if "base_color_texture" in m:
                mat[0] = textured_param(int(m["base_color_texture"]))
            for name, slot in _FLOAT_PARAMS:
                mat[slot] = np.float32(m[name])
                tex = m.get(name + "_texture")
                if tex is not None:
                    mat[slot] = textured_param(int(tex["texture"]), int(tex["channel"]))
            sc.materials.append(mat)

    pm_ids: Dict[Tuple[int, int], int] = {}
    for n in header.get("objects", []):
        matrix = np.asarray(n["matrix"], np.float32).reshape(16)  # glm::make_mat4: column-major
        col = lambda c: matrix[4 * c:4 * c + 4]
        kind = n["type"]
        if kind == "MESH":
            mesh_id = int(n["mesh"])
            mat_id = int(n["material"]) if default_mode else -1
            key = (mesh_id, mat_id)
            if key not in pm_ids:
                pm_ids[key] = len(sc.parameterized_meshes)
                sc.parameterized_meshes.append(ParameterizedMesh(mesh_id, [mat_id]))
            sc.instances.append(Instance(matrix.copy(), pm_ids[key]))
        elif kind == "LIGHT":
            color = np.asarray(n["color"], np.float32)[:3] * np.float32(n["energy"])
            normal = -_normalize(col(2))  # vec4 normalise, like glm::normalize(glm::column(matrix, 2))
            sc.lights.append(quad_light([color[0], color[1], color[2], 1.0], col(3), normal, _normalize(col(0))[:3],
                                        _normalize(col(1))[:3], float(n["size"][0]), float(n["size"][1])))
            sc.lights[-1][7] = col(3)[3]  # position and normal keep their w (glm::vec4 members)
            sc.lights[-1][11] = normal[3]
        elif kind == "CAMERA":
            position = col(3)[:3].copy()
            direction = _normalize(-col(2))[:3]
            sc.cameras.append(Camera(position, (position + direction * np.float32(10.0)).astype(np.float32),
                                     _normalize(col(1))[:3], float(np.float32(n["fov_y"]) / np.float32(1.18))))
        else:
            raise RuntimeError("Unsupported object type: not a mesh or camera?")

    # validate_materials (scene.cpp:935-958): objects without a material share one default material
    if any(mid == -1 for p in sc.parameterized_meshes for mid in p.material_ids):
        default_id = len(sc.materials)
        sc.materials.append(disney_material())
        for p in sc.parameterized_meshes:
            p.material_ids = [default_id if mid == -1 else mid for mid in p.material_ids]
    if not sc.lights:
        sc.lights.append(crts_default_light())
    return sc


def save_crts(scene: Scene, path: str) -> None:
    """Write `scene` so that `load_crts` (ours or the reference's) reads the same scene back."""
    from PIL import Image as PILImage
    views: List[dict] = []
    blob = bytearray()

    def add_view(a: np.ndarray, type_name: str) -> int:
        while len(blob) % 8:
            blob.append(0)
        raw = np.ascontiguousarray(a).tobytes()
        views.append({"byte_offset": len(blob), "byte_length": len(raw), "type": type_name})
        blob.extend(raw)
        return len(views) - 1

    # one CRTS mesh per geometry
    mesh_of: Dict[Tuple[int, int], int] = {}
    meshes = []
    for mi, mesh in enumerate(scene.meshes):
        for gi, g in enumerate(mesh.geometries):
            entry = {"positions": add_view(np.asarray(g.vertices, np.float32), "VEC3_F32"),
                     "indices": add_view(np.asarray(g.indices, np.uint32), "VEC3_U32")}
            if g.uvs is not None:
                entry["texcoords"] = add_view(np.asarray(g.uvs, np.float32), "VEC2_F32")
            mesh_of[(mi, gi)] = len(meshes)
            meshes.append(entry)

    images = []
    for t, im in enumerate(scene.textures):
        a = np.asarray(im.img, np.uint8).reshape(im.height, im.width, im.channels)[::-1]
        mode = {1: "L", 3: "RGB", 4: "RGBA"}[im.channels]
        buf = io.BytesIO()
        PILImage.fromarray(a[..., 0] if im.channels == 1 else a, mode).save(buf, format="PNG")
        images.append({"name": im.name or f"texture{t}", "view": add_view(np.frombuffer(buf.getvalue(), np.uint8), "UINT_8"),
                       "color_space": "LINEAR" if im.color_space == LINEAR else "SRGB"})

    def handle(x) -> Tuple[bool, int, int]:
        bits = int(np.asarray([x], np.float32).view(np.uint32)[0])
        return bool(bits & 0x80000000), bits & 0x1FFFFFFF, (bits >> 29) & 0x3

    materials = []
    for m in scene.materials:
        m = np.asarray(m, np.float32)
        entry: dict = {}
        textured, tex_id, _ = handle(m[0])
        if textured:
            entry["base_color"] = [1.0, float(m[1]), float(m[2])]  # r is replaced by the handle on load
            entry["base_color_texture"] = tex_id
        else:
            entry["base_color"] = [float(x) for x in m[0:3]]
        for name, slot in _FLOAT_PARAMS:
            textured, tex_id, channel = handle(m[slot])
            entry[name] = 0.0 if textured else float(m[slot])
            if textured:
                entry[name + "_texture"] = {"texture": tex_id, "channel": channel}
        materials.append(entry)

    objects = []
    for inst in scene.instances:
        pm = scene.parameterized_meshes[inst.parameterized_mesh_id]
        for gi in range(len(scene.meshes[pm.mesh_id].geometries)):
            objects.append({"type": "MESH", "name": f"object{len(objects)}",
                            "matrix": [float(x) for x in np.asarray(inst.transform, np.float32).reshape(16)],
                            "mesh": mesh_of[(pm.mesh_id, gi)], "material": int(pm.material_ids[gi])})
    for l in scene.lights:
        l = np.asarray(l, np.float32)
        normal = l[8:11]
        matrix = np.zeros(16, np.float32)
        matrix[0:3], matrix[4:7], matrix[8:11], matrix[12:15], matrix[15] = l[12:15], l[16:19], -normal, l[4:7], 1.0
        objects.append({"type": "LIGHT", "name": f"light{len(objects)}", "matrix": [float(x) for x in matrix],
                        "color": [float(x) for x in l[0:3]], "energy": 1.0, "size": [float(l[15]), float(l[19])]})
    for c in scene.cameras:
        pos, center, up = (np.asarray(x, np.float32) for x in (c.position, c.center, c.up))
        back = -_normalize(center - pos)  # the camera looks down its -z axis
        x_axis = _normalize(np.cross(up, back))
        matrix = np.zeros(16, np.float32)
        matrix[0:3], matrix[4:7], matrix[8:11], matrix[12:15], matrix[15] = x_axis, _normalize(up), back, pos, 1.0
        objects.append({"type": "CAMERA", "name": f"camera{len(objects)}", "matrix": [float(x) for x in matrix],
                        "fov_y": float(np.float32(c.fov_y) * np.float32(1.18))})

    header = json.dumps({"buffer_views": views, "meshes": meshes, "images": images, "materials": materials,
                         "objects": objects}).encode("utf-8")
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(header)))
        f.write(header)
        f.write(bytes(blob))
