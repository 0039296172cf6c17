"""OBJ/MTL scene ingest for the headless harness.

Follows what the reference's importer makes of an OBJ file (util/scene.cpp:94-228, which sits on
tinyobjloader): every `o`/`g` group becomes one Geometry of ONE Mesh; vertices are re-indexed on
unique (position, normal, uv) index triples in order of first use; the group's material is the
material of its first face; one ParameterizedMesh, one identity Instance; Disney parameters from
the MTL per scene.cpp:191-216 (quirk Q14: specular = clamp(Ns/500), roughness = 1 - specular,
transmission forced to 0, `map_Kd` becomes an sRGB base-colour texture loaded flipped and forced
to 4 channels, util/material.cpp:5-17, quirk Q13); faces without a material get the default
DisneyMaterial (validate_materials, scene.cpp:935-958); one generated quad light (scene.cpp:218-227).

`save_obj` writes a Scene of that shape back out, so the synthetic benchmark scenes can be fed to
a real ChameleonRT build for cross-checking.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
from typing import Dict, List

import numpy as np

from .scene import (SRGB, Geometry, Image, Instance, Mesh, ParameterizedMesh, Scene, disney_material,
                    obj_default_light, textured_param)


_SPACE = " \t"


def _mtl_real(rest: str, default: float):
    """tinyobjloader's parseReal (util/tiny_obj_loader.h:682-690): skip blanks, take the token up to the next blank, parse it with
    tryParseDouble's grammar; what does not parse -- or is not there -- reads as the default. (MTL has legitimate non-numeric
    forms, `Kd spectral file.rfl`; geometry numbers in the OBJ file are held to the grammar strictly, see _num.)"""
    rest = rest.lstrip(_SPACE)
    n = 0
    while n < len(rest) and rest[n] not in " \t\r":
        n += 1
    tok, rest = rest[:n], rest[n:]
    num = _number_prefix(tok)
    return (float(np.float32(float(num))) if num is not None else default), rest


def _mtl_texture_name(rest: str):
    """ParseTextureNameAndOption (util/tiny_obj_loader.h:906-976): options with their fixed numbers of arguments are skipped, the
    texture name is THE REST OF THE LINE from the first thing that is not an option (file names may contain blanks)."""
    one = ("-blendu", "-blendv", "-clamp", "-boost", "-bm", "-imfchan", "-colorspace")
    name = None
    while rest:
        rest = rest.lstrip(_SPACE)
        if not rest:
            break
        for opt, n_args in [(o, 1) for o in one] + [("-o", 3), ("-s", 3), ("-t", 3), ("-mm", 2)]:
            if rest.startswith(opt) and len(rest) > len(opt) and rest[len(opt)] in _SPACE:
                rest = rest[len(opt) + 1:]
                for _ in range(n_args):
                    _, rest = _mtl_real(rest, 0.0)  # (every argument is one blank-delimited token, numeric or not)
                break
        else:
            if rest.startswith("-type") and len(rest) > 5 and rest[5] in _SPACE:
                _, rest = _mtl_real(rest[5:], 0.0)
                continue
            name, rest = rest, ""
    return name


def _first_mtllib_that_opens(base_dir: str, rest: str):
    """`mtllib a.mtl b.mtl`: tinyobjloader splits the rest of the line at single blanks (SplitString = std::getline with ' ',
    util/tiny_obj_loader.h:1343-1351: two blanks in a row name "" in between, a trailing blank names nothing), tries the names in
    turn and stops at the first that OPENS (:2031-2049). MaterialFileReader (:1741-1749) only WARNS about a file it cannot open; if
    none opens, LoadObj warns "Failed to load material file(s). Use default material." and goes on -- the reference's importer
    throws only on !ret || !err.empty() (util/scene.cpp:110), which a missing .mtl does not set: no material is defined, every
    `usemtl` resolves to -1 and those geometries get the importer's default material. Returns the name, or None (with a warning)."""
    names = rest.split(" ")
    if names and names[-1] == "":
        names.pop()  # (std::getline yields no empty item after a trailing delimiter)
    for name in names:
        path = os.path.join(base_dir, name)
        # what std::ifstream opens: a readable file -- or a directory (`mtllib  a.mtl` names "" first: the OBJ's own directory
        # opens as a stream, nothing can be read from it, and tinyobjloader is content with that: see _parse_mtl)
        if os.path.isdir(path) or (os.path.isfile(path) and os.access(path, os.R_OK)):
            return name
    import warnings
    warnings.warn(f"no material library of `mtllib {rest}` could be opened in {base_dir}: the default material is used")
    return None


def _parse_mtl(path: str) -> List[dict]:
    """tinyobjloader's LoadMtl (util/tiny_obj_loader.h:1353-1725) for the three things the reference's importer reads from a
    material (util/scene.cpp:191-216): Kd, Ns, map_Kd. Its defaults are ZERO diffuse and shininess 1 (InitMaterial); missing
    components of `Kd` read as 0; a material is flushed by the next `newmtl` only if it has a name, and the last one always --
    so a file without any `newmtl` still yields one (unnamed) material; the name is everything after `newmtl` and ONE blank."""
    def fresh(name=""):
        return {"name": name, "Kd": (0.0, 0.0, 0.0), "Ns": 1.0, "map_Kd": ""}
    mats: List[dict] = []
    cur = fresh()
    if os.path.isdir(path):
        # `mtllib  a.mtl` (two blanks) names "" first: the reference then opens the OBJ's directory as a stream, reads nothing from
        # it and is content -- one unnamed material, and a.mtl is never looked at
        text = ""
    else:
        with open(path, newline="") as f:
            text = f.read()
    for line in re.split(r"\r\n|\n|\r", text):
        line = line.rstrip(_SPACE)
        tok = line.lstrip(_SPACE)
        if not tok or tok[0] == "#":
            continue
        def key(k):
            return tok.startswith(k) and len(tok) > len(k) and tok[len(k)] in _SPACE
        if key("newmtl"):
            if cur["name"] != "":
                mats.append(cur)
            cur = fresh(tok[7:])
        elif key("Kd"):
            r, rest = _mtl_real(tok[2:], 0.0)
            g, rest = _mtl_real(rest, 0.0)
            b, rest = _mtl_real(rest, 0.0)
            cur["Kd"] = (r, g, b)
        elif key("Ns"):
            cur["Ns"], _ = _mtl_real(tok[2:], 0.0)
        elif key("map_Kd"):
            name = _mtl_texture_name(tok[7:])
            if name is not None:
                cur["map_Kd"] = name
    mats.append(cur)
    return mats


def _load_texture(path: str, name: str) -> Image:
    from .image_io import decode_rgba
    with open(path, "rb") as f:
        a = decode_rgba(f.read())[::-1].copy()  # forced to 4 channels; stbi_set_flip_vertically_on_load(1)
    return Image(a.shape[1], a.shape[0], 4, a, SRGB, name)


_warned_native = False


def _read_obj_native(path: str):
    """Geometry half of load_obj through the streaming C++ reader (csrc/obj_reader.cpp, include/crt_scene_io.h):
    [(vertices, indices, uvs or None, material name or None)], mtllib file names."""
    import ctypes as C
    from . import build
    L = C.CDLL(build.build_scene_io())
    L.crt_obj_parse.restype = C.c_void_p
    L.crt_obj_parse.argtypes = [C.c_char_p]
    L.crt_obj_error.restype = C.c_char_p
    L.crt_obj_error.argtypes = [C.c_void_p]
    L.crt_obj_free.argtypes = [C.c_void_p]
    L.crt_obj_mtllib.restype = C.c_char_p
    L.crt_obj_mtllib.argtypes = [C.c_void_p, C.c_int]
    L.crt_obj_num_mtllibs.argtypes = [C.c_void_p]
    L.crt_obj_num_shapes.argtypes = [C.c_void_p]
    L.crt_obj_shape_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.crt_obj_shape_material.restype = C.c_char_p
    L.crt_obj_shape_material.argtypes = [C.c_void_p, C.c_int]
    L.crt_obj_shape_material_libs.argtypes = [C.c_void_p, C.c_int]
    L.crt_obj_shape_copy.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    f = L.crt_obj_parse(os.fsencode(path))
    try:
        err = L.crt_obj_error(f)
        if err:
            raise ValueError(f"{path}: {err.decode()}")
        libs = [L.crt_obj_mtllib(f, i).decode() for i in range(L.crt_obj_num_mtllibs(f))]
        shapes = []
        for s in range(L.crt_obj_num_shapes(f)):
            nv, nt, has_uv, has_mat = C.c_uint64(), C.c_uint64(), C.c_int(), C.c_int()
            assert L.crt_obj_shape_info(f, s, C.byref(nv), C.byref(nt), C.byref(has_uv), C.byref(has_mat)) == 0
            if has_uv.value < 0:
                raise ValueError("OBJ group mixes vertices with and without texture coordinates")
            v = np.zeros((nv.value, 3), np.float32)
            ix = np.zeros((nt.value, 3), np.uint32)
            uv = np.zeros((nv.value, 2), np.float32) if has_uv.value else None
            assert L.crt_obj_shape_copy(f, s, v.ctypes.data_as(C.c_void_p), ix.ctypes.data_as(C.c_void_p),
                                        uv.ctypes.data_as(C.c_void_p) if uv is not None else None) == 0
            shapes.append((v, ix, uv, L.crt_obj_shape_material(f, s).decode() if has_mat.value else None, L.crt_obj_shape_material_libs(f, s)))
        return shapes, libs
    finally:
        L.crt_obj_free(f)


def load_obj(path: str, material_mode: str = "default", samples_per_pixel: int = 1, reader: str = "native") -> Scene:
    """reader="native": the streaming C++ reader (a 10 M-triangle OBJ in seconds); "python": the line-by-line twin this
    module started with, kept as the statement of the semantics -- both must produce identical arrays (tests/test_obj_io.py)."""
    if reader == "python":
        return _load_obj_python(path, material_mode, samples_per_pixel)
    base_dir = os.path.dirname(os.path.abspath(path))
    try:
        raw, libs = _read_obj_native(path)
    except (OSError, subprocess.CalledProcessError) as e:
        # no C++17 compiler with floating-point std::from_chars (libstdc++ >= 11) on this host, or the library does not
        # load: the line-by-line twin reads the same arrays, only slower (this is scene ingest, not the render path)
        global _warned_native
        if not _warned_native:
            _warned_native = True
            print(f"[chameleonrt_amd.obj_io] native OBJ reader unavailable ({e}); using the Python reader", file=sys.stderr)
        return _load_obj_python(path, material_mode, samples_per_pixel)
    if not raw:
        raise ValueError(f"no faces in {path}")
    obj_materials: List[dict] = []
    mat_index_after: List[Dict[str, int]] = [{}]  # name -> id with the first k mtllib files read
    for lib in libs:
        idx = dict(mat_index_after[-1])
        for m in _parse_mtl(os.path.join(base_dir, lib)):
            idx.setdefault(m["name"], len(obj_materials))  # (tinyobjloader: std::map::insert -- the FIRST material of a name is the one `usemtl` finds)
            obj_materials.append(m)
        mat_index_after.append(idx)
    geoms = [Geometry(v, ix, uv) for v, ix, uv, _, _ in raw]
    material_ids = [(mat_index_after[k].get(name, -1) if name is not None else -1) if material_mode == "default" else -1
                    for _, _, _, name, k in raw]
    return _assemble_obj_scene(path, base_dir, geoms, material_ids, obj_materials, material_mode, samples_per_pixel)


_NUMBER = re.compile(r"[+-]?[0-9]+(\.[0-9]*)?([eE][+-]?[0-9]+)?\Z")


def _number_prefix(tok: str):
    """What tryParseDouble (util/tiny_obj_loader.h:567-680) makes of a token that need not be a number to its end: the longest
    prefix sign? digits+ ('.' digits*)? ([eE] sign? digits+)? -- it stops, successfully, at the first character that does not
    continue the number ("1.5abc" is 1.5) but FAILS as a whole on an exponent marker without digits ("1e", "2.5E+"). None = fails."""
    i, n = 0, len(tok)
    if i < n and tok[i] in "+-":
        i += 1
    d0 = i
    while i < n and tok[i].isdigit() and tok[i].isascii():
        i += 1
    if i == d0:
        return None
    if i < n and tok[i] == ".":
        i += 1
        while i < n and tok[i].isdigit() and tok[i].isascii():
            i += 1
    if i < n and tok[i] in "eE":
        j = i + 1
        if j < n and tok[j] in "+-":
            j += 1
        e0 = j
        while j < n and tok[j].isdigit() and tok[j].isascii():
            j += 1
        if j == e0:
            return None
        i = j
    return tok[:i]


def _num(tok: str) -> float:
    """A coordinate. The grammar is the reference's (tinyobjloader's tryParseDouble, util/tiny_obj_loader.h:567-680): optional
    sign, at least one digit, optional fraction, optional exponent. What does not parse there (".5", "inf", "1_0") reads as
    0.0 in the reference; here the file is refused instead of being loaded as something else."""
    if not _NUMBER.match(tok):
        raise ValueError(f"not a number in the reference's OBJ grammar: {tok!r}")
    return float(tok)


def _triangulate(corners, positions):
    """A face's corners -> triangles the way the reference's importer gets them from tinyobjloader
    (util/tiny_obj_loader.h:1107-1310, triangulate = true at util/scene.cpp:117): ear clipping, not a fan. The polygon is
    projected on two axes chosen from its first non-degenerate corner; its signed area gives the winding; from corner
    `guess` on, three consecutive remaining corners (a, b, c) are an ear when the turn at b has the polygon's winding and no
    other remaining corner lies inside (a, b, c) (crossing-number test); the ear is emitted as (a, b, c), b leaves the ring;
    a search that has made as many fruitless steps as corners are left gives up (what was emitted stays). float32
    arithmetic in the reference's order. The native reader (csrc/obj_reader.cpp `triangulate`) is the twin of this."""
    n = len(corners)
    if n < 3:
        return []
    if n == 3:
        return [tuple(corners)]
    f32 = np.float32
    P = [[f32(x) for x in positions[c[0]]] for c in corners]
    eps = f32(np.finfo(np.float32).eps)
    ax0, ax1 = 1, 2
    for k in range(n):
        a, b, c = P[k], P[(k + 1) % n], P[(k + 2) % n]
        e0 = [b[i] - a[i] for i in range(3)]
        e1 = [c[i] - b[i] for i in range(3)]
        cx = abs(e0[1] * e1[2] - e0[2] * e1[1])
        cy = abs(e0[2] * e1[0] - e0[0] * e1[2])
        cz = abs(e0[0] * e1[1] - e0[1] * e1[0])
        if cx > eps or cy > eps or cz > eps:
            if not (cx > cy and cx > cz):
                ax0 = 0
                if cz > cx and cz > cy:
                    ax1 = 1
            break
    area = f32(0.0)
    for k in range(n):
        a, b = P[k], P[(k + 1) % n]
        area = area + (a[ax0] * b[ax1] - a[ax1] * b[ax0]) * f32(0.5)

    def inside(vx, vy, tx, ty):
        c = False
        j = 2
        for i in range(3):
            if (vy[i] > ty) != (vy[j] > ty) and tx < (vx[j] - vx[i]) * (ty - vy[i]) / (vy[j] - vy[i]) + vx[i]:
                c = not c
            j = i
        return c

    ring = list(range(n))  # indices into corners / P
    out = []
    guess, budget, last_size = 0, n, n
    while len(ring) > 3 and budget > 0:
        m = len(ring)
        if guess >= m:
            guess -= m
        if last_size != m:
            last_size, budget = m, m
        else:
            budget -= 1
        tri = [ring[(guess + k) % m] for k in range(3)]
        vx = [P[t][ax0] for t in tri]
        vy = [P[t][ax1] for t in tri]
        cross = (vx[1] - vx[0]) * (vy[2] - vy[1]) - (vy[1] - vy[0]) * (vx[2] - vx[1])
        if cross * area < f32(0.0):
            guess += 1
            continue
        if any(inside(vx, vy, P[ring[(guess + o) % m]][ax0], P[ring[(guess + o) % m]][ax1]) for o in range(3, m)):
            guess += 1
            continue
        out.append(tuple(corners[t] for t in tri))
        del ring[(guess + 1) % m]
    if len(ring) == 3:
        out.append(tuple(corners[t] for t in ring))
    return out


def _load_obj_python(path: str, material_mode: str = "default", samples_per_pixel: int = 1) -> Scene:
    base_dir = os.path.dirname(os.path.abspath(path))
    positions: List[List[float]] = []
    normals: List[List[float]] = []
    texcoords: List[List[float]] = []
    obj_materials: List[dict] = []
    mat_index: Dict[str, int] = {}
    shapes: List[dict] = []
    cur = None
    cur_mat = -1
    since_change = False  # an `f` statement since the last group statement or change of material

    def shape():
        nonlocal cur
        if cur is None:
            cur = {"faces": [], "mats": []}
            shapes.append(cur)
        return cur

    def resolve(i: int, n: int, key_only: bool = False) -> int:
        # OBJ indices are 1-based; negative = relative. 0, or (v, vt) beyond what has been read either way: an error. A normal
        # index is only a re-indexing key (never dereferenced; the reference's tinyobjloader checks it for 0 only).
        if i == 0 or (not key_only and (i > n or i < -n)) or abs(i) > 0x7fffffff:
            raise ValueError("face index out of range")
        return i - 1 if i > 0 else n + i

    with open(path, newline="") as f:
        text = f.read()
    for line in re.split(r"\r\n|\n|\r", text):
        tok = line.split()
        if not tok or tok[0].startswith("#"):
            continue
        k = tok[0]
        body = line.lstrip(" \t")
        # tinyobjloader recognises a statement by its keyword AND a blank right after it: a bare `o`, `g`, `usemtl`, `v` is skipped
        if not (len(body) > len(k) and body[len(k)] in " \t"):
            continue
        if k == "v":
            positions.append([_num(x) for x in tok[1:4]])
        elif k == "vn":
            normals.append(None)  # counted only: the hot path reads no normals (quirk Q7); the native reader does the same
        elif k == "vt":
            texcoords.append([_num(tok[1]), _num(tok[2]) if len(tok) > 2 else 0.0])
        elif k in ("o", "g"):
            # tinyobjloader hands the faces read so far to the current shape when the MATERIAL changes (usemtl), and its `o`
            # statement keeps the shape only if there are faces it has not handed over yet (util/tiny_obj_loader.h:2117-2123;
            # `g` and the end of the file look at the shape itself): an `o` right after a material-changing `usemtl` LOSES the
            # object before it. The reference's scenes are what that makes of a file, so the faces are dropped here, too.
            if k == "o" and cur is not None and cur["faces"] and not since_change:
                cur["faces"], cur["mats"] = [], []
            since_change = False
            if cur is not None and not cur["faces"]:
                continue  # tinyobj does not emit empty shapes
            cur = None
        elif k == "mtllib":
            lib = _first_mtllib_that_opens(base_dir, body[7:])
            for m in (_parse_mtl(os.path.join(base_dir, lib)) if lib is not None else []):
                mat_index.setdefault(m["name"], len(obj_materials))
                obj_materials.append(m)
        elif k == "usemtl":
            new_mat = mat_index.get(body[7:], -1)  # the name as it stands after the keyword and ONE blank
            if new_mat != cur_mat:
                since_change = False
                cur_mat = new_mat
        elif k == "f":
            since_change = True
            corners = []
            for c in tok[1:]:
                parts = c.split("/")
                vi = resolve(int(parts[0]), len(positions))
                ti = resolve(int(parts[1]), len(texcoords)) if len(parts) > 1 and parts[1] else -1
                ni = resolve(int(parts[2]), len(normals), True) if len(parts) > 2 and parts[2] else -1
                corners.append((vi, ni, ti))
            for tri in _triangulate(corners, positions):
                s = shape()
                s["faces"].append(tri)
                s["mats"].append(cur_mat)
    shapes = [s for s in shapes if s["faces"]]
    if not shapes:
        raise ValueError(f"no faces in {path}")
    pos = np.asarray(positions, np.float32)
    tc = np.asarray(texcoords, np.float32) if texcoords else np.zeros((0, 2), np.float32)

    geoms, material_ids = [], []
    for s in shapes:
        material_ids.append(s["mats"][0] if material_mode == "default" else -1)  # first face's material
        remap: Dict[tuple, int] = {}
        verts, uvs, tris = [], [], []
        has_uv = False
        for face in s["faces"]:
            tri = []
            for idx in face:
                v = remap.get(idx)
                if v is None:
                    v = len(verts)
                    remap[idx] = v
                    verts.append(pos[idx[0]])
                    if idx[2] >= 0:
                        uvs.append(tc[idx[2]])
                        has_uv = True
                tri.append(v)
            tris.append(tri)
        if has_uv and len(uvs) != len(verts):
            raise ValueError("OBJ group mixes vertices with and without texture coordinates")
        geoms.append(Geometry(np.asarray(verts, np.float32), np.asarray(tris, np.uint32),
                              np.asarray(uvs, np.float32) if has_uv else None))
    return _assemble_obj_scene(path, base_dir, geoms, material_ids, obj_materials, material_mode, samples_per_pixel)


def _assemble_obj_scene(path, base_dir, geoms, material_ids, obj_materials, material_mode, samples_per_pixel) -> Scene:
    """One Mesh, one ParameterizedMesh, one identity Instance, MTL -> Disney, the generated light (scene.cpp:184-227)."""
    sc = Scene(name=os.path.basename(path))
    sc.meshes = [Mesh(geoms)]
    sc.instances = [Instance(np.eye(4, dtype=np.float32).reshape(16), 0)]
    sc.samples_per_pixel = samples_per_pixel
    if material_mode == "default":
        tex_ids: Dict[str, int] = {}
        for m in obj_materials:
            specular = float(np.clip(np.float32(m["Ns"]) / np.float32(500.0), 0.0, 1.0))
            d = disney_material(base_color=m["Kd"], specular=specular,
                                roughness=float(np.clip(np.float32(1.0) - np.float32(specular), 0.0, 1.0)),
                                specular_transmission=0.0)
            if m["map_Kd"]:
                name = m["map_Kd"]
                if name not in tex_ids:
                    tex_ids[name] = len(sc.textures)
                    sc.textures.append(_load_texture(os.path.join(base_dir, name.replace("\\", "/")), name))
                d[0] = textured_param(tex_ids[name])
            sc.materials.append(d)
    # validate_materials: ids of -1 get one default material appended at the end
    if any(m == -1 for m in material_ids):
        default_id = len(sc.materials)
        sc.materials.append(disney_material())
        material_ids = [default_id if m == -1 else m for m in material_ids]
    sc.parameterized_meshes = [ParameterizedMesh(0, material_ids)]
    sc.lights = [obj_default_light()]
    return sc


def save_obj(scene: Scene, path: str) -> None:
    """Write a single-mesh / identity-instance Scene as OBJ + MTL (+ PNG textures)."""
    if len(scene.meshes) != 1 or len(scene.instances) != 1:
        raise ValueError("save_obj handles the OBJ-shaped scenes only (one mesh, one instance)")
    base_dir = os.path.dirname(os.path.abspath(path))
    stem = os.path.splitext(os.path.basename(path))[0]
    with open(os.path.join(base_dir, stem + ".mtl"), "w") as f:
        for i, m in enumerate(scene.materials):
            bits = int(np.asarray(m[0:1], np.float32).view(np.uint32)[0])
            f.write(f"newmtl m{i}\n")
            if bits & 0x80000000:
                tid = bits & 0x1FFFFFFF
                f.write("Kd 1 1 1\n" f"map_Kd {stem}_tex{tid}.png\n")
            else:
                f.write("Kd %.9g %.9g %.9g\n" % tuple(float(x) for x in m[0:3]))
            f.write("Ns %.9g\n" % (float(m[4]) * 500.0))
    if scene.textures:
        from PIL import Image as PILImage
        for t, im in enumerate(scene.textures):
            a = np.asarray(im.img, np.uint8).reshape(im.height, im.width, im.channels)[::-1]
            mode = {1: "L", 3: "RGB", 4: "RGBA"}[im.channels]
            PILImage.fromarray(a[..., 0] if im.channels == 1 else a, mode).save(
                os.path.join(base_dir, f"{stem}_tex{t}.png"))
    mat_ids = scene.parameterized_meshes[0].material_ids
    with open(path, "w") as f:
        f.write(f"mtllib {stem}.mtl\n")
        voff = toff = 1
        for gi, g in enumerate(scene.meshes[0].geometries):
            f.write(f"o shape{gi}\nusemtl m{mat_ids[gi]}\n")
            for v in g.vertices:
                f.write("v %.9g %.9g %.9g\n" % (float(v[0]), float(v[1]), float(v[2])))
            if g.uvs is not None:
                for uv in g.uvs:
                    f.write("vt %.9g %.9g\n" % (float(uv[0]), float(uv[1])))
            for tri in g.indices:
                if g.uvs is not None:
                    f.write("f %d/%d %d/%d %d/%d\n" % tuple(x for i in tri for x in (voff + int(i), toff + int(i))))
                else:
                    f.write("f %d %d %d\n" % tuple(voff + int(i) for i in tri))
            voff += len(g.vertices)
            if g.uvs is not None:
                toff += len(g.uvs)
