"""Host-side scene model handed to ``set_scene``.

Python mirror of the plain structs of the reference's scene data model -- ``Geometry`` /
``Mesh`` / ``ParameterizedMesh`` / ``Instance`` (util/mesh.h:6-47), ``DisneyMaterial`` and
``Image`` (util/material.h:9-46), ``QuadLight`` (util/lights.h:6-18), ``Camera``
(util/camera.h:5-8) and ``Scene`` (util/scene.h:23-32) -- plus the conversion to the flat
``crt_scene_desc`` of include/crt_hip.h that crosses the C-ABI.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

LINEAR, SRGB = 0, 1  # util/material.h:9

TEXTURED_PARAM_MASK = 0x80000000  # util/texture_channel_mask.h:16


def textured_param(tex_id: int, channel: int = 0) -> np.float32:
    """Encode a texture handle in a float parameter (util/texture_channel_mask.h:16-23)."""
    mask = TEXTURED_PARAM_MASK | ((channel & 0x3) << 29) | (tex_id & 0x1FFFFFFF)
    return np.array([mask], dtype=np.uint32).view(np.float32)[0]


@dataclass
class Geometry:
    vertices: np.ndarray  # (N,3) float32
    indices: np.ndarray  # (M,3) uint32
    uvs: Optional[np.ndarray] = None  # (N,2) float32

    def num_tris(self) -> int:
        return int(self.indices.shape[0])


@dataclass
class Mesh:
    geometries: List[Geometry]

    def num_tris(self) -> int:
        return sum(g.num_tris() for g in self.geometries)


@dataclass
class ParameterizedMesh:
    mesh_id: int
    material_ids: List[int]


@dataclass
class Instance:
    transform: np.ndarray  # (16,) float32, column-major like glm::mat4
    parameterized_mesh_id: int


def disney_material(base_color=(0.9, 0.9, 0.9), metallic=0.0, specular=0.0, roughness=1.0,
                    specular_tint=0.0, anisotropy=0.0, sheen=0.0, sheen_tint=0.0, clearcoat=0.0,
                    clearcoat_gloss=0.0, ior=1.5, specular_transmission=0.0) -> np.ndarray:
    """16-float ``DisneyMaterial`` with the reference's defaults (util/material.h:29-46)."""
    m = np.zeros(16, dtype=np.float32)
    m[0:3] = base_color
    m[3:14] = [metallic, specular, roughness, specular_tint, anisotropy, sheen, sheen_tint,
               clearcoat, clearcoat_gloss, ior, specular_transmission]
    return m


@dataclass
class Image:
    width: int
    height: int
    channels: int
    img: np.ndarray  # (h, w, channels) uint8, row 0 first
    color_space: int = LINEAR
    name: str = ""


def _normalize(v):
    v = np.asarray(v, dtype=np.float32)
    return (v * (np.float32(1.0) / np.sqrt(np.dot(v, v), dtype=np.float32))).astype(np.float32)


def ortho_basis(n):
    """util/util.cpp:43-58."""
    n = np.asarray(n, dtype=np.float32)
    v_y = np.zeros(3, dtype=np.float32)
    if -0.6 < n[0] < 0.6:
        v_y[0] = 1
    elif -0.6 < n[1] < 0.6:
        v_y[1] = 1
    elif -0.6 < n[2] < 0.6:
        v_y[2] = 1
    else:
        v_y[0] = 1
    v_x = _normalize(np.cross(v_y, n))
    v_y = _normalize(np.cross(n, v_x))
    return v_x, v_y


def quad_light(emission, position, normal, v_x, v_y, width, height) -> np.ndarray:
    """20-float ``QuadLight`` (util/lights.h:6-18)."""
    l = np.zeros(20, dtype=np.float32)
    l[0:3] = emission[:3]
    l[3] = emission[3] if len(emission) > 3 else 0.0
    l[4:7] = position[:3]
    l[8:11] = normal[:3]
    l[12:15] = v_x
    l[15] = width
    l[16:19] = v_y
    l[19] = height
    return l


def obj_default_light() -> np.ndarray:
    """The light the OBJ loader generates (util/scene.cpp:218-227, quirk Q14)."""
    n = _normalize([0.5, -0.8, -0.5])
    pos = (np.float32(-10.0) * n).astype(np.float32)
    v_x, v_y = ortho_basis(n)
    l = quad_light([20.0, 20.0, 20.0, 20.0], pos, n, v_x, v_y, 5.0, 5.0)
    l[7] = np.float32(-0.0)  # `light.position = -10.f * light.normal` with normal.w = 0 (scene.cpp:222): minus zero
    return l


@dataclass
class Camera:
    position: np.ndarray
    center: np.ndarray
    up: np.ndarray
    fov_y: float


@dataclass
class Scene:
    meshes: List[Mesh] = field(default_factory=list)
    parameterized_meshes: List[ParameterizedMesh] = field(default_factory=list)
    instances: List[Instance] = field(default_factory=list)
    materials: List[np.ndarray] = field(default_factory=list)
    textures: List[Image] = field(default_factory=list)
    lights: List[np.ndarray] = field(default_factory=list)
    cameras: List[Camera] = field(default_factory=list)
    samples_per_pixel: int = 1
    name: str = ""

    def unique_tris(self) -> int:
        return sum(m.num_tris() for m in self.meshes)

    def total_tris(self) -> int:
        return sum(self.meshes[self.parameterized_meshes[i.parameterized_mesh_id].mesh_id].num_tris()
                   for i in self.instances)

    def white_diffuse(self) -> "Scene":
        """``-mat-mode white_diffuse`` (util/scene.cpp:126-130, 935-958): every geometry gets
        the default DisneyMaterial, no textures."""
        s = Scene(meshes=self.meshes, instances=self.instances, lights=self.lights,
                  cameras=self.cameras, samples_per_pixel=self.samples_per_pixel,
                  name=self.name + "+white_diffuse")
        s.parameterized_meshes = [ParameterizedMesh(p.mesh_id, [0] * len(p.material_ids))
                                  for p in self.parameterized_meshes]
        s.materials = [disney_material()]
        s.textures = []
        return s


# ---- ctypes mirror of include/crt_hip.h -----------------------------------------------------


class GeometryDesc(C.Structure):
    _fields_ = [("vertices", C.POINTER(C.c_float)), ("n_vertices", C.c_uint64),
                ("indices", C.POINTER(C.c_uint32)), ("n_triangles", C.c_uint64),
                ("uvs", C.POINTER(C.c_float))]


class MeshDesc(C.Structure):
    _fields_ = [("first_geometry", C.c_uint32), ("n_geometries", C.c_uint32)]


class ParameterizedMeshDesc(C.Structure):
    _fields_ = [("mesh_id", C.c_uint32), ("n_material_ids", C.c_uint32),
                ("material_ids", C.POINTER(C.c_uint32))]


class InstanceDesc(C.Structure):
    _fields_ = [("transform", C.c_float * 16), ("parameterized_mesh_id", C.c_uint32)]


class ImageDesc(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("channels", C.c_int32),
                ("color_space", C.c_int32), ("data", C.POINTER(C.c_uint8))]


class SceneDesc(C.Structure):
    _fields_ = [("geometries", C.POINTER(GeometryDesc)), ("n_geometries", C.c_uint32),
                ("meshes", C.POINTER(MeshDesc)), ("n_meshes", C.c_uint32),
                ("parameterized_meshes", C.POINTER(ParameterizedMeshDesc)),
                ("n_parameterized_meshes", C.c_uint32),
                ("instances", C.POINTER(InstanceDesc)), ("n_instances", C.c_uint32),
                ("materials", C.POINTER(C.c_float)), ("n_materials", C.c_uint32),
                ("textures", C.POINTER(ImageDesc)), ("n_textures", C.c_uint32),
                ("lights", C.POINTER(C.c_float)), ("n_lights", C.c_uint32),
                ("samples_per_pixel", C.c_uint32)]


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class PackedScene:
    """A ``crt_scene_desc`` plus the numpy arrays that back its pointers."""

    def __init__(self, scene: Scene):
        keep = []
        geoms = []
        mesh_descs = (MeshDesc * max(1, len(scene.meshes)))()
        for mi, m in enumerate(scene.meshes):
            mesh_descs[mi].first_geometry = len(geoms)
            mesh_descs[mi].n_geometries = len(m.geometries)
            geoms.extend(m.geometries)
        gdescs = (GeometryDesc * max(1, len(geoms)))()
        for gi, g in enumerate(geoms):
            v = np.ascontiguousarray(g.vertices, dtype=np.float32)
            idx = np.ascontiguousarray(g.indices, dtype=np.uint32)
            keep += [v, idx]
            gdescs[gi].vertices = _fptr(v)
            gdescs[gi].n_vertices = v.shape[0]
            gdescs[gi].indices = idx.ctypes.data_as(C.POINTER(C.c_uint32))
            gdescs[gi].n_triangles = idx.shape[0]
            if g.uvs is not None:
                uv = np.ascontiguousarray(g.uvs, dtype=np.float32)
                keep.append(uv)
                gdescs[gi].uvs = _fptr(uv)
        pm = (ParameterizedMeshDesc * max(1, len(scene.parameterized_meshes)))()
        for i, p in enumerate(scene.parameterized_meshes):
            ids = np.asarray(p.material_ids, dtype=np.uint32)
            keep.append(ids)
            pm[i].mesh_id = p.mesh_id
            pm[i].n_material_ids = ids.shape[0]
            pm[i].material_ids = ids.ctypes.data_as(C.POINTER(C.c_uint32))
        inst = (InstanceDesc * max(1, len(scene.instances)))()
        for i, it in enumerate(scene.instances):
            t = np.asarray(it.transform, dtype=np.float32).reshape(16)
            for k in range(16):
                inst[i].transform[k] = float(t[k])
            inst[i].parameterized_mesh_id = it.parameterized_mesh_id
        mats = np.ascontiguousarray(np.stack(scene.materials).astype(np.float32)) \
            if scene.materials else np.zeros((0, 16), np.float32)
        lights = np.ascontiguousarray(np.stack(scene.lights).astype(np.float32)) \
            if scene.lights else np.zeros((0, 20), np.float32)
        tex = (ImageDesc * max(1, len(scene.textures)))()
        for i, im in enumerate(scene.textures):
            data = np.ascontiguousarray(im.img, dtype=np.uint8)
            keep.append(data)
            tex[i].width, tex[i].height, tex[i].channels = im.width, im.height, im.channels
            tex[i].color_space = im.color_space
            tex[i].data = data.ctypes.data_as(C.POINTER(C.c_uint8))
        keep += [mats, lights, mesh_descs, gdescs, pm, inst, tex]
        d = SceneDesc()
        d.geometries, d.n_geometries = gdescs, len(geoms)
        d.meshes, d.n_meshes = mesh_descs, len(scene.meshes)
        d.parameterized_meshes, d.n_parameterized_meshes = pm, len(scene.parameterized_meshes)
        d.instances, d.n_instances = inst, len(scene.instances)
        d.materials, d.n_materials = _fptr(mats), mats.shape[0]
        d.textures, d.n_textures = tex, len(scene.textures)
        d.lights, d.n_lights = _fptr(lights), lights.shape[0]
        d.samples_per_pixel = scene.samples_per_pixel
        self.desc = d
        self._keep = keep

    def ptr(self):
        return C.byref(self.desc)
