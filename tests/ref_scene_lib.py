"""ctypes access to oracle/_ref/libref_scene.so: the REFERENCE's own scene importer
(util/scene.cpp, mesh.cpp, material.cpp, util.cpp, flatten_gltf.cpp, gltf_types.cpp, buffer_view.cpp,
file_mapping.cpp with their vendored parsers), compiled from /root/reference by `make -C oracle ref`
against the GLM stand-in of oracle/ref_shim_scene/. TEST INFRASTRUCTURE; exists in the development
container only -- elsewhere the tests use the golden dumps made from it (tests/golden/refscene_*.npz)."""
import ctypes as C
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(_ROOT, "oracle", "_ref", "libref_scene.so")
_LIB = None


def available():
    return os.path.exists(PATH)


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(PATH)
        vp, u64, u64p = C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)
        fp, u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        L.refscene_load.restype = vp
        L.refscene_load.argtypes = [C.c_char_p, C.c_int]
        L.refscene_error.restype = C.c_char_p
        L.refscene_free.argtypes = [vp]
        L.refscene_counts.argtypes = [vp, u64p]
        L.refscene_mesh_geometries.restype = u64
        L.refscene_mesh_geometries.argtypes = [vp, u64]
        L.refscene_geometry_sizes.argtypes = [vp, u64, u64, u64p]
        L.refscene_geometry_copy.argtypes = [vp, u64, u64, fp, fp, u32p]
        L.refscene_pmesh.restype = u64
        L.refscene_pmesh.argtypes = [vp, u64, u64p]
        L.refscene_pmesh_materials.argtypes = [vp, u64, u32p]
        L.refscene_instance.restype = u64
        L.refscene_instance.argtypes = [vp, u64, fp]
        L.refscene_material.argtypes = [vp, u64, fp]
        L.refscene_texture_info.argtypes = [vp, u64, C.POINTER(C.c_int32)]
        L.refscene_texture_copy.argtypes = [vp, u64, C.POINTER(C.c_uint8)]
        L.refscene_light.argtypes = [vp, u64, fp]
        L.refscene_camera.argtypes = [vp, u64, fp]
        _LIB = L
    return _LIB


def load(path, white_diffuse=False):
    """The reference's `Scene(path, material_mode)` as a flat dict of numpy arrays (npz-able)."""
    L = lib()
    h = L.refscene_load(os.path.abspath(path).encode(), int(white_diffuse))
    if not h:
        raise RuntimeError(L.refscene_error().decode())
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    try:
        n = (C.c_uint64 * 7)()
        L.refscene_counts(h, n)
        out = {"counts": np.array(list(n), np.uint64)}
        geoms_per_mesh = []
        for m in range(n[0]):
            ng = L.refscene_mesh_geometries(h, m)
            geoms_per_mesh.append(ng)
            for g in range(ng):
                sz = (C.c_uint64 * 4)()
                L.refscene_geometry_sizes(h, m, g, sz)
                v = np.zeros((sz[0], 3), np.float32)
                uv = np.zeros((sz[2], 2), np.float32)
                idx = np.zeros((sz[3], 3), np.uint32)
                L.refscene_geometry_copy(h, m, g, fp(v), fp(uv) if sz[2] else None, idx.ctypes.data_as(C.POINTER(C.c_uint32)))
                out[f"m{m}g{g}_vertices"], out[f"m{m}g{g}_uvs"], out[f"m{m}g{g}_indices"] = v, uv, idx
                out[f"m{m}g{g}_n_normals"] = np.array([sz[1]], np.uint64)
        out["geoms_per_mesh"] = np.array(geoms_per_mesh, np.uint64)
        for i in range(n[1]):
            mesh_id = C.c_uint64()
            k = L.refscene_pmesh(h, i, C.byref(mesh_id))
            ids = np.zeros(k, np.uint32)
            L.refscene_pmesh_materials(h, i, ids.ctypes.data_as(C.POINTER(C.c_uint32)))
            out[f"pm{i}_mesh"], out[f"pm{i}_materials"] = np.array([mesh_id.value], np.uint64), ids
        tr = np.zeros((n[2], 16), np.float32)
        pm = np.zeros(n[2], np.uint64)
        for i in range(n[2]):
            pm[i] = L.refscene_instance(h, i, fp(tr[i]))
        out["instance_transforms"], out["instance_pmesh"] = tr, pm
        mats = np.zeros((n[3], 16), np.float32)
        for i in range(n[3]):
            L.refscene_material(h, i, fp(mats[i]))
        out["materials"] = mats
        for i in range(n[4]):
            info = (C.c_int32 * 4)()
            L.refscene_texture_info(h, i, info)
            data = np.zeros(info[0] * info[1] * info[2], np.uint8)
            L.refscene_texture_copy(h, i, data.ctypes.data_as(C.POINTER(C.c_uint8)))
            out[f"tex{i}_info"], out[f"tex{i}_data"] = np.array(list(info), np.int32), data
        lights = np.zeros((n[5], 20), np.float32)
        for i in range(n[5]):
            L.refscene_light(h, i, fp(lights[i]))
        out["lights"] = lights
        cams = np.zeros((n[6], 10), np.float32)
        for i in range(n[6]):
            L.refscene_camera(h, i, fp(cams[i]))
        out["cameras"] = cams
        return out
    finally:
        L.refscene_free(h)


def flatten(sc):
    """A chameleonrt_amd.scene.Scene in the same flat form."""
    out = {"counts": np.array([len(sc.meshes), len(sc.parameterized_meshes), len(sc.instances), len(sc.materials),
                               len(sc.textures), len(sc.lights), len(sc.cameras)], np.uint64)}
    out["geoms_per_mesh"] = np.array([len(m.geometries) for m in sc.meshes], np.uint64)
    for m, mesh in enumerate(sc.meshes):
        for g, geom in enumerate(mesh.geometries):
            out[f"m{m}g{g}_vertices"] = np.asarray(geom.vertices, np.float32).reshape(-1, 3)
            out[f"m{m}g{g}_uvs"] = (np.zeros((0, 2), np.float32) if geom.uvs is None
                                   else np.asarray(geom.uvs, np.float32).reshape(-1, 2))
            out[f"m{m}g{g}_indices"] = np.asarray(geom.indices, np.uint32).reshape(-1, 3)
    for i, p in enumerate(sc.parameterized_meshes):
        out[f"pm{i}_mesh"] = np.array([p.mesh_id], np.uint64)
        out[f"pm{i}_materials"] = np.asarray(p.material_ids, np.uint32)
    out["instance_transforms"] = np.array([np.asarray(i.transform, np.float32).reshape(16) for i in sc.instances],
                                          np.float32).reshape(-1, 16)
    out["instance_pmesh"] = np.array([i.parameterized_mesh_id for i in sc.instances], np.uint64)
    out["materials"] = np.array([np.asarray(m, np.float32) for m in sc.materials], np.float32).reshape(-1, 16)
    for i, t in enumerate(sc.textures):
        out[f"tex{i}_info"] = np.array([t.width, t.height, t.channels, int(t.color_space)], np.int32)
        out[f"tex{i}_data"] = np.asarray(t.img, np.uint8).reshape(-1)
    out["lights"] = np.array([np.asarray(l, np.float32) for l in sc.lights], np.float32).reshape(-1, 20)
    out["cameras"] = np.array([[*c.position, *c.center, *c.up, c.fov_y] for c in sc.cameras], np.float32).reshape(-1, 10)
    return out
