"""Golden Scene dumps from the REFERENCE's importer (development container only).

    make -C oracle ref            # builds oracle/_ref/libref_scene.so from /root/reference/util/*.cpp
    python tests/golden/make_scene_golden.py

Writes small scene files of every on-disk format the reference's `Scene::Scene` dispatches on
(util/scene.cpp:49-72) under tests/golden/scenes/ and, next to them, what the reference's own
load_obj / load_gltf / load_crts (compiled from where they lie, GLM replaced by the stand-in of
oracle/ref_shim_scene/) make of each: tests/golden/refscene_<name>.npz, in DEFAULT and WHITE_DIFFUSE
material mode. tests/test_importers_pinned.py compares chameleonrt_amd's importers to these dumps
wherever the suite runs, and to the live library where it exists.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
SCENES = os.path.join(HERE, "scenes")


def write_files():
    """Every test file, deterministic. Returns {name: path}."""
    from chameleonrt_amd import scenes
    from chameleonrt_amd.crts_io import save_crts
    from chameleonrt_amd.obj_io import save_obj
    from tests import test_crts_io, test_gltf_io
    os.makedirs(SCENES, exist_ok=True)
    out = {}
    p = os.path.join(SCENES, "cornell.obj")
    save_obj(scenes.cornell(), p)
    out["obj_cornell"] = p
    p = os.path.join(SCENES, "atrium.obj")  # textured (map_Kd -> flipped RGBA8 sRGB), UVs outside [0, 1], 24 materials
    save_obj(scenes.sponza_like(detail=0.004, tex_size=8), p)
    out["obj_atrium"] = p
    with open(os.path.join(SCENES, "quirks.mtl"), "w") as f:
        f.write("newmtl shiny\nKd 0.2 0.4 0.6\nNs 250\nnewmtl dull\nKd 1 0 0\nNs 0\nnewmtl over\nKd 0.3 0.3 0.3\nNs 900\n")
    p = os.path.join(SCENES, "quirks.obj")
    with open(p, "w") as f:  # polygon fan, negative indices, first-face material, material-less group, vn/vt triples
        f.write("mtllib quirks.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1\nvt 0 0\nvt 1 0\nvt 1 1\nvn 0 0 1\nvn 0 2 0\n"
                "o quad\nusemtl shiny\nf 1 2 3 4\n"
                "o mixed\nusemtl dull\nf 1 2 5\nusemtl shiny\nf -1 -2 -3\n"
                "o triples\nusemtl over\nf 1/1/1 2/2/1 3/3/2\nf 1/1/2 3/3/2 4/2/2\n"
                "g tail\nf 1 3 5\n")
    out["obj_quirks"] = p
    p = os.path.join(SCENES, "polygons.obj")
    with open(p, "w") as f:  # what tinyobjloader's ear clipping (tiny_obj_loader.h:1107-1310) makes of non-trivial polygons
        f.write(polygons_obj())
    out["obj_polygons"] = p
    from PIL import Image as _PIL
    for k, n in enumerate(["mq0.png", "mq my tex.png"]):
        _PIL.fromarray(np.random.default_rng(40 + k).integers(0, 256, (4, 4, 3), dtype=np.uint8)).save(os.path.join(SCENES, n))
    with open(os.path.join(SCENES, "mtlquirks.mtl"), "w", newline="") as f:  # what tinyobjloader's LoadMtl makes of the odd corners of MTL
        f.write("Kd 0.1 0.2 0.3\r\n"                                   # before any newmtl: the unnamed material, dropped by the first newmtl
                "newmtl nokd\r\nNs 250\r\n"                            # no Kd: diffuse is ZERO, not grey
                "newmtl one\r\n  Kd\t0.5\r\n"                          # missing components read as 0
                "newmtl dup\r\nKd 1 0 0\r\nnewmtl dup\r\nKd 0 1 0\r\n"   # usemtl finds the FIRST of a name
                "newmtl spectral\r\nKd spectral file.rfl 1\r\nNs 2e\r\n"  # what is not a number reads as the default
                "newmtl tex\r\nmap_Kd -s 2 2 1 -o 0.5 0 0 -clamp on -bm 2 -mm 0 1 -imfchan r mq0.png\r\n"
                "newmtl texsp  \r\nmap_Kd mq my tex.png\r\n"           # the name is the rest of the line; trailing blanks of a line are trimmed
                "newmtl\r\n"                                            # not a newmtl line at all
                "newmtl last\r\nKd .5 7. 1.5abc")                      # no newline at the end; `.5` is not a number there, `1.5abc` is 1.5
    p = os.path.join(SCENES, "mtlquirks.obj")
    with open(p, "w") as f:
        f.write("mtllib mtlquirks.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\n" +
                "".join(f"o g{i}\nusemtl {n}\nf 1/1 2/2 3/3\n" for i, n in enumerate(
                    ["nokd", "one", "dup", "spectral", "tex", "texsp", "last", "missing"])))
    out["obj_mtlquirks"] = p
    p = os.path.join(SCENES, "bare.obj")
    with open(p, "w") as f:
        f.write("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    out["obj_bare"] = p
    p = os.path.join(SCENES, "scene.gltf")
    test_gltf_io._scene_file(p, glb=False)
    out["gltf_scene"] = p
    p = os.path.join(SCENES, "scene.glb")
    test_gltf_io._scene_file(p, glb=True)
    out["glb_scene"] = p
    # a node tree with rotations: flatten_gltf.cpp
    b = test_gltf_io.Builder()
    a_pos = b.accessor(b.view(test_gltf_io.QUAD_POS.tobytes()), 5126, "VEC3", 4)
    a_idx = b.accessor(b.view(np.uint16(test_gltf_io.QUAD_IDX).tobytes()), 5123, "SCALAR", 6)
    b.doc["meshes"] = [{"primitives": [{"attributes": {"POSITION": a_pos}, "indices": a_idx}]}] * 2
    s = float(np.sqrt(0.5))
    b.doc["nodes"] = [{"translation": [10, 0, 0], "rotation": [0.0, s, 0.0, s], "children": [1, 3]},
                      {"mesh": 0, "scale": [2, 3, 0.5], "rotation": [0.1825742, 0.3651484, 0.5477226, 0.7302967], "children": [2]},
                      {"mesh": 1, "translation": [0.25, 1, -3]},
                      {"mesh": 1, "translation": [0, 0, 5], "scale": [1.5, 1.5, 1.5]},
                      {"mesh": 0, "matrix": [0.5, 0, 0, 0, 0, 0.5, 0, 0, 0, 0, 0.5, 0, 1, 2, 3, 1]}]
    b.doc["scenes"][0]["nodes"] = [0, 4]
    p = os.path.join(SCENES, "tree.gltf")
    b.finish(p)
    out["gltf_tree"] = p
    p = os.path.join(SCENES, "handmade.crts")
    test_crts_io._handmade(p, with_light=True)
    out["crts_handmade"] = p
    p = os.path.join(SCENES, "nolight.crts")
    test_crts_io._handmade(p, with_light=False)
    out["crts_nolight"] = p
    p = os.path.join(SCENES, "grove.crts")  # two-level scene with textures, lights and a camera, written by save_crts
    save_crts(scenes.instanced_grove(n_instances=12, leaves_per_tree=40, tex_size=8), p)
    out["crts_grove"] = p
    return out


def polygons_obj():
    """Polygons a fan gets wrong or that steer the ear search: a concave quad with the reflex corner at each of the four
    positions and in both windings, in each of the three projection planes; a concave pentagon, an L, a plus sign, a
    five-pointed star (10 corners), a spiral; a non-planar quad; collinear runs; a corner used twice; a polygon with all
    corners on one line (no axes found, no area: nothing is an ear); a bow tie (self-intersecting: the search gives up);
    a 12-gon with texture coordinates; tiny and huge coordinates. Every face is its own `o` group so that the vertex
    order the clipping implies shows in each Geometry."""
    L = ["mtllib quirks.mtl", "vn 0 0 1"]
    nv = [0]

    def face(name, pts, uv=False, order=None):
        L.append(f"o {name}")
        for q in pts:
            L.append("v %.9g %.9g %.9g" % tuple(q))
        n = len(pts)
        if uv:
            for q in pts:
                L.append("vt %.9g %.9g" % (q[0] * 0.25, q[1] * 0.25 + q[2]))
        idx = list(order) if order is not None else list(range(n))
        base = nv[0] + 1
        L.append("f " + " ".join(f"{base + i}/{base_uv[0] + i}" if uv else f"{base + i}" for i in idx))
        nv[0] += n
        if uv:
            base_uv[0] += n

    base_uv = [1]
    dart = [(0, 0), (2, 0.5), (4, 0), (2, 3)]  # concave at corner 1
    for plane, lift in (("xy", lambda a, b: (a, b, 0.0)), ("yz", lambda a, b: (0.5, a, b)), ("zx", lambda a, b: (b, -1.0, a))):
        for rot in range(4):
            for wind in (1, -1):
                pts = [lift(*dart[(rot + wind * k) % 4]) for k in range(4)]
                face(f"dart_{plane}_r{rot}_{'ccw' if wind > 0 else 'cw'}", pts)
    face("pentagon_concave", [(0, 0, 0), (2, 0, 0), (2, 2, 0), (1, 0.5, 0), (0, 2, 0)])
    face("ell", [(0, 0, 1), (3, 0, 1), (3, 1, 1), (1, 1, 1), (1, 3, 1), (0, 3, 1)], uv=True)
    face("plus", [(1, 0, 0), (2, 0, 0), (2, 1, 0), (3, 1, 0), (3, 2, 0), (2, 2, 0), (2, 3, 0), (1, 3, 0), (1, 2, 0), (0, 2, 0), (0, 1, 0), (1, 1, 0)])
    star = [((1.0 if k % 2 == 0 else 0.38) * np.cos(np.pi * k / 5 + 0.3), (1.0 if k % 2 == 0 else 0.38) * np.sin(np.pi * k / 5 + 0.3), 0.1 * k) for k in range(10)]
    face("star_tilted", star)
    face("star_cw", star[::-1], uv=True)
    spiral = [(0, 0), (5, 0), (5, 5), (1, 5), (1, 2), (3, 2), (3, 3), (2, 3), (2, 4), (4, 4), (4, 1), (0, 1)]
    face("spiral", [(a, 0.25 * a, b) for a, b in spiral])
    face("nonplanar_quad", [(0, 0, 0), (1, 0, 0.7), (1, 1, -0.4), (0, 1, 0.9)])
    face("nonplanar_concave", [(0, 0, 0), (0.4, 0.4, 2.0), (1, 0, 0.3), (0, 1, -0.5)])
    face("collinear_run", [(0, 0, 0), (1, 0, 0), (2, 0, 0), (3, 0, 0), (3, 2, 0), (0, 2, 0)])
    face("collinear_start", [(0, 0, 0), (1, 1, 1), (2, 2, 2), (3, 3, 3), (3, 0, 3), (0, -1, 0)])
    face("all_on_a_line", [(0, 0, 0), (1, 1, 0), (2, 2, 0), (3, 3, 0), (4, 4, 0)])
    face("corner_twice", [(0, 0, 0), (2, 0, 0), (2, 2, 0), (1, 1, 0), (0, 2, 0)], order=[0, 1, 2, 3, 4, 3])
    face("same_corner_thrice", [(0, 0, 0), (1, 0, 0), (0, 1, 0)], order=[0, 1, 1, 2, 0])
    face("bow_tie", [(0, 0, 0), (2, 2, 0), (2, 0, 0), (0, 2, 0)])
    face("bow_tie_hex", [(0, 0, 0), (3, 2, 0), (3, 0, 0), (0, 2, 0), (1.5, 3, 0), (1.5, -1, 0)])
    face("gon12_uv", [(np.cos(k * np.pi / 6) * (2 + (k % 3 == 0)), np.sin(k * np.pi / 6) * (2 + (k % 3 == 0)), 0.05 * k * k) for k in range(12)], uv=True)
    face("tiny", [(1e-6 * a, 1e-6 * b, 0) for a, b in dart])
    face("sub_epsilon", [(1e-4 * a, 1e-4 * b, 0) for a, b in dart])  # every corner's cross product below FLT_EPSILON: default axes (y, z)
    face("huge", [(1e5 * a + 3e6, 1e5 * b, 7e5) for a, b in dart])
    face("sliver", [(0, 0, 0), (1000, 1e-3, 0), (2000, 0, 0), (1000, 2e-3, 0)])
    return "\n".join(L) + "\n"


DECODER_IMAGES = ["a420.jpg", "b444.jpg", "cgray.jpg", "dprog.jpg", "e32rle.tga", "f24.tga", "g24top.tga", "hgray.tga",
                  "i422.jpg", "j420prog.jpg", "k420rst.jpg", "l1x1.jpg", "m444q100.jpg", "ncmyk.jpg", "ograyprog.jpg", "p422rst.jpg"]


def write_decoder_files():
    """tests/golden/scenes/decoders/: an OBJ whose eight materials name JPEG (4:2:0, 4:4:4, greyscale, progressive) and TGA
    (32-bit RLE, 24-bit bottom-up and top-down, 8-bit grey) textures -- what real Sponza / San Miguel assets ship. The reference
    decodes them with its vendored stb_image (util/material.cpp:5-17); tests/test_texture_decoders.py holds our loader to it."""
    from PIL import Image
    d = os.path.join(SCENES, "decoders")
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(3)

    def pic(w, h, ch):
        y, x = np.mgrid[0:h, 0:w]
        a = np.stack([(np.sin(x / 7.0 + k) + np.cos(y / 5.0 * (k + 1))) * 60 + 128 + rng.normal(0, 12, (h, w)) for k in range(ch)], -1)
        return np.clip(a, 0, 255).astype(np.uint8)

    Image.fromarray(pic(67, 45, 3)).save(os.path.join(d, "a420.jpg"), quality=85, subsampling=2)
    Image.fromarray(pic(64, 64, 3)).save(os.path.join(d, "b444.jpg"), quality=92, subsampling=0)
    Image.fromarray(pic(50, 33, 1)[..., 0]).save(os.path.join(d, "cgray.jpg"), quality=80)
    Image.fromarray(pic(40, 24, 3)).save(os.path.join(d, "dprog.jpg"), quality=85, progressive=True)
    Image.fromarray(pic(37, 29, 4)).save(os.path.join(d, "e32rle.tga"), compression="tga_rle")
    Image.fromarray(pic(31, 18, 3)).save(os.path.join(d, "f24.tga"))
    Image.fromarray(pic(31, 18, 3)).save(os.path.join(d, "g24top.tga"), orientation=1)
    Image.fromarray(pic(21, 11, 1)[..., 0]).save(os.path.join(d, "hgray.tga"))
    # more JPEG shapes: 4:2:2, progressive 4:2:0 with optimised tables, restart intervals, one pixel, quality 100 (all-ones
    # quantisation: the largest coefficients), Adobe CMYK, progressive grey, 4:2:2 with a restart every row of MCUs
    Image.fromarray(pic(33, 17, 3)).save(os.path.join(d, "i422.jpg"), quality=70, subsampling=1)
    Image.fromarray(pic(50, 47, 3)).save(os.path.join(d, "j420prog.jpg"), quality=60, subsampling=2, progressive=True, optimize=True)
    Image.fromarray(pic(64, 48, 3)).save(os.path.join(d, "k420rst.jpg"), quality=80, subsampling=2, restart_marker_blocks=3)
    Image.fromarray(pic(1, 1, 3)).save(os.path.join(d, "l1x1.jpg"), quality=90)
    Image.fromarray(pic(24, 24, 3)).save(os.path.join(d, "m444q100.jpg"), quality=100, subsampling=0)
    Image.fromarray(pic(16, 12, 4), "CMYK").save(os.path.join(d, "ncmyk.jpg"), quality=85)
    Image.fromarray(pic(40, 31, 1)[..., 0]).save(os.path.join(d, "ograyprog.jpg"), quality=75, progressive=True)
    Image.fromarray(pic(47, 40, 3)).save(os.path.join(d, "p422rst.jpg"), quality=88, subsampling=1, restart_marker_rows=1)
    with open(os.path.join(d, "t.mtl"), "w") as f:
        for i, n in enumerate(DECODER_IMAGES):
            f.write(f"newmtl m{i}\nKd 1 1 1\nmap_Kd {n}\n")
    with open(os.path.join(d, "t.obj"), "w") as f:
        f.write("mtllib t.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\n")
        for i in range(len(DECODER_IMAGES)):
            f.write(f"o s{i}\nusemtl m{i}\nf 1/1 2/2 3/3\n")
    return os.path.join(d, "t.obj")


def main():
    from tests import ref_scene_lib as R
    if not R.available():
        raise SystemExit("oracle/_ref/libref_scene.so is missing: `make -C oracle ref` (needs /root/reference)")
    if "--decoders" in sys.argv:  # (the image files are committed: regenerating them needs the same Pillow / libjpeg build)
        d = R.load(write_decoder_files())
        np.savez_compressed(os.path.join(HERE, "refdecoders_obj.npz"), **{k: v for k, v in d.items() if k.startswith("tex") or k == "counts"})
        print("decoders", dict(zip("mesh pmesh inst mat tex light cam".split(), d["counts"])))
        return
    only = [a[len("--only="):] for a in sys.argv if a.startswith("--only=")]  # (npz bytes carry time stamps: add a file without touching the others)
    for name, path in write_files().items():
        if only and name not in only:
            continue
        for wd in (False, True):
            d = R.load(path, white_diffuse=wd)
            np.savez_compressed(os.path.join(HERE, f"refscene_{name}{'_wd' if wd else ''}.npz"), **d)
            print(name, "white_diffuse" if wd else "default", dict(zip("mesh pmesh inst mat tex light cam".split(), d["counts"])))


if __name__ == "__main__":
    main()
