"""CPU: OBJ/MTL ingest (util/scene.cpp:94-228 semantics) -- round trip of synthetic scenes, vertex
re-indexing on (position, normal, uv) triples, first-face material, MTL -> Disney mapping (quirk
Q14), texture flip + 4 channels (quirk Q13), default material for material-less groups."""
import contextlib
import os

import numpy as np
import pytest

from chameleonrt_amd import scenes
from chameleonrt_amd.obj_io import load_obj, save_obj


def _tri_soup(g):
    return g.vertices[g.indices.reshape(-1)].reshape(-1, 3, 3)


def test_round_trip_cornell(tmp_path):
    sc = scenes.cornell()
    p = os.path.join(tmp_path, "cornell.obj")
    save_obj(sc, p)
    ld = load_obj(p)
    assert len(ld.meshes) == 1 and len(ld.instances) == 1 and len(ld.parameterized_meshes) == 1
    assert ld.total_tris() == 34 and len(ld.meshes[0].geometries) == 7
    assert ld.parameterized_meshes[0].material_ids == sc.parameterized_meshes[0].material_ids
    for a, b in zip(sc.meshes[0].geometries, ld.meshes[0].geometries):
        assert np.array_equal(_tri_soup(a), _tri_soup(b))
    for a, b in zip(sc.materials, ld.materials):
        assert np.allclose(a, b, atol=1e-7)
    assert np.array_equal(ld.lights[0], sc.lights[0])  # the generated light of scene.cpp:218-227


def test_textured_round_trip_and_uv_dedup(tmp_path):
    sc = scenes.sponza_like(detail=0.01, tex_size=16)
    p = os.path.join(tmp_path, "s.obj")
    save_obj(sc, p)
    ld = load_obj(p)
    assert ld.total_tris() == sc.total_tris()
    assert len(ld.textures) <= len(sc.textures) and all(t.channels == 4 and t.color_space == 1 for t in ld.textures)
    for a, b in zip(sc.meshes[0].geometries[:10], ld.meshes[0].geometries[:10]):
        assert np.array_equal(_tri_soup(a), _tri_soup(b))
        assert np.array_equal(a.uvs[a.indices.reshape(-1)], b.uvs[b.indices.reshape(-1)])
        assert len(b.vertices) <= len(a.vertices)
    # texture survives: PNG written flipped, loaded flipped back
    m0 = ld.materials[0]
    bits = int(np.asarray(m0[0:1]).view(np.uint32)[0])
    assert bits & 0x80000000
    t_loaded = ld.textures[bits & 0x1FFFFFFF]
    t_orig = sc.textures[int(np.asarray(sc.materials[0][0:1]).view(np.uint32)[0]) & 0x1FFFFFFF]
    assert np.array_equal(np.asarray(t_loaded.img).reshape(-1), np.asarray(t_orig.img).reshape(-1))


def test_hand_written_obj_semantics(tmp_path):
    open(os.path.join(tmp_path, "m.mtl"), "w").write(
        "newmtl shiny\nKd 0.2 0.4 0.6\nNs 250\n" "newmtl dull\nKd 1 0 0\nNs 0\n")
    open(os.path.join(tmp_path, "t.obj"), "w").write(
        "mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1\n"
        "o quad\nusemtl shiny\nf 1 2 3 4\n"            # convex quad -> 2 triangles (ear clipping: the same as a fan), 4 shared vertices
        "o mixed\nusemtl dull\nf 1 2 5\nusemtl shiny\nf -1 -2 -3\n"  # first face's material wins; negative indices
        "g nomat_group\n" "g tail\nf 1 3 5\n")
    sc = load_obj(os.path.join(tmp_path, "t.obj"))
    g = sc.meshes[0].geometries
    assert [x.num_tris() for x in g] == [2, 2, 1]
    assert len(g[0].vertices) == 4 and np.array_equal(g[0].indices, [[0, 1, 2], [0, 2, 3]])
    assert sc.parameterized_meshes[0].material_ids == [0, 1, 0]  # 'tail' inherits the current usemtl (shiny)
    shiny = sc.materials[0]
    assert np.allclose(shiny[0:3], [0.2, 0.4, 0.6]) and np.isclose(shiny[4], 0.5) and np.isclose(shiny[5], 0.5)
    assert sc.materials[1][4] == 0 and sc.materials[1][5] == 1 and shiny[13] == 0
    # f -1 -2 -3 = v5 v4 v3; v5 was already used by the first face of the group (re-indexing)
    assert np.array_equal(g[1].vertices[2:5], [[0, 0, 1], [0, 1, 0], [1, 1, 0]])
    assert np.array_equal(g[1].indices, [[0, 1, 2], [2, 3, 4]])


def test_groups_without_material_get_the_default(tmp_path):
    open(os.path.join(tmp_path, "n.obj"), "w").write("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    sc = load_obj(os.path.join(tmp_path, "n.obj"))
    assert sc.parameterized_meshes[0].material_ids == [0] and np.allclose(sc.materials[0][0:3], 0.9)
    wd = load_obj(os.path.join(tmp_path, "n.obj"), material_mode="white_diffuse")
    assert len(wd.materials) == 1 and wd.materials[0][5] == 1.0


def _same_scene(a, b):
    assert len(a.meshes[0].geometries) == len(b.meshes[0].geometries)
    for ga, gb in zip(a.meshes[0].geometries, b.meshes[0].geometries):
        assert np.array_equal(np.asarray(ga.vertices).view(np.uint32), np.asarray(gb.vertices).view(np.uint32))
        assert np.array_equal(ga.indices, gb.indices)
        assert (ga.uvs is None) == (gb.uvs is None)
        if ga.uvs is not None:
            assert np.array_equal(np.asarray(ga.uvs).view(np.uint32), np.asarray(gb.uvs).view(np.uint32))
    assert a.parameterized_meshes[0].material_ids == b.parameterized_meshes[0].material_ids
    assert len(a.materials) == len(b.materials) and all(np.array_equal(x, y) for x, y in zip(a.materials, b.materials))
    assert len(a.textures) == len(b.textures)


def test_native_reader_equals_the_python_twin(tmp_path):
    """The streaming C++ reader (csrc/obj_reader.cpp) and the line-by-line Python loader it replaces produce identical
    arrays -- vertices after re-indexing bit for bit, indices, uvs, material ids -- on the hand-written semantics file
    (quads, negative indices, usemtl before / after mtllib, empty groups, `v//n` and `v/t/n` corners, a lone `vt u`),
    on the reference-pinned golden scenes and on a written-out synthetic scene."""
    open(os.path.join(tmp_path, "m.mtl"), "w").write("newmtl shiny\nKd 0.2 0.4 0.6\nNs 250\nnewmtl dull stuff\nKd 1 0 0\nNs 0\n")
    open(os.path.join(tmp_path, "t.obj"), "w").write(
        "# comment\nusemtl shiny\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1\nv -0.5 +2.5e-1 1.\nvn 0 0 1\nvt 0.25 0.75\nvt 0.5\n"
        "o early\nf 1 2 3\n"                      # usemtl before any mtllib: resolves to nothing
        "mtllib m.mtl\n"
        "o quad\nusemtl shiny\nf 1 2 3 4\n"
        "o mixed\nusemtl dull stuff\nf 1//1 2//1 5//1\nusemtl shiny\nf -1//1 -2//1 -3//1\n"
        "o textured\nf -1/1/1 -2/2/1 -3/1/1\nf 1/1 2/2 3/-1\n"
        "g nomat_group\n" "g tail\nf 1 3 5 6 2\n"   # pentagon: three triangles
        "\n   \nf 2 3 5\n")
    for mode in ("default", "white_diffuse"):
        _same_scene(load_obj(os.path.join(tmp_path, "t.obj"), mode), load_obj(os.path.join(tmp_path, "t.obj"), mode, reader="python"))
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenes")
    for name in sorted(os.listdir(golden)):
        if name.endswith(".obj"):
            _same_scene(load_obj(os.path.join(golden, name)), load_obj(os.path.join(golden, name), reader="python"))
    sc = scenes.sponza_like(detail=0.02, tex_size=16)
    p = os.path.join(tmp_path, "s.obj")
    save_obj(sc, p)
    _same_scene(load_obj(p), load_obj(p, reader="python"))


def _random_obj(rng, crlf):
    """A syntactically valid OBJ in the dialect the importers agree on: positions / texture coordinates in varied number
    formats, triangles, quads and n-gons with positive and negative indices, `v`, `v/t`, `v//n`, `v/t/n` corners (one
    kind per object, like real exporters), objects, groups, materials before and after their library, comments, blank
    lines, tabs and trailing blanks."""
    def num():
        x = float(rng.normal()) * 10 ** int(rng.integers(-3, 3))
        # (no ".5": a number without integer digits reads as 0.0 in the reference and is refused here, see the test below)
        return rng.choice([f"{x:.6g}", f"{x:.3e}", f"{x:+.4f}", f"{x:.5f}".rstrip("0") if 0 < x < 1 else f"{x:.2f}"])
    # (one normal up front: corners of the `v//n` kinds name normal 1, which the reference's importer dereferences)
    lines, n_v, n_vt = ["# fuzz", "vn 0 0 1"], 0, 0
    sep = lambda: rng.choice([" ", "  ", "\t"])
    if rng.random() < 0.7:
        lines.append("mtllib m.mtl")
    for o in range(int(rng.integers(1, 5))):
        for _ in range(int(rng.integers(3, 12))):
            lines.append("v" + sep() + sep().join(num() for _ in range(3)) + rng.choice(["", " ", "\t "]))
            n_v += 1
        for _ in range(int(rng.integers(0, 6))):
            lines.append("vt" + sep() + sep().join(num() for _ in range(int(rng.integers(1, 4)))))
            n_vt += 1
        if rng.random() < 0.3:
            lines.append("vn 0 1 0")
        lines.append(rng.choice(["o", "g"]) + f" part{o}" + rng.choice(["", " extra words"]))
        if rng.random() < 0.8:
            lines.append("usemtl " + rng.choice(["shiny", "dull stuff", "missing"]))
        kind = int(rng.integers(0, 4)) if n_vt else int(rng.choice([0, 2]))
        for _ in range(int(rng.integers(1, 8))):
            if rng.random() < 0.15:
                lines.append(rng.choice(["", "   ", "# mid comment", "s off", "usemtl shiny"]))
            corners = []
            for _ in range(int(rng.choice([3, 3, 3, 4, 4, 5, 6]))):
                v = int(rng.integers(1, n_v + 1))
                v = v if rng.random() < 0.7 else v - n_v - 1
                t = int(rng.integers(1, n_vt + 1)) if n_vt else 1
                t = t if rng.random() < 0.7 else t - n_vt - 1
                corners.append([f"{v}", f"{v}/{t}", f"{v}//1", f"{v}/{t}/1"][kind])
            lines.append("f" + sep() + sep().join(corners))
    return ("\r\n" if crlf else "\n").join(lines) + ("" if rng.random() < 0.3 else ("\r\n" if crlf else "\n"))


def test_native_reader_equals_the_python_twin_on_generated_files(tmp_path):
    """60 seeded random OBJ files (Unix and DOS line ends; triangles, quads and n-gons over RANDOM vertices, i.e. concave,
    self-intersecting, degenerate polygons with repeated corners): the two readers agree on every array, or refuse together --
    and where the reference's own importer exists (oracle/_ref/libref_scene.so = util/scene.cpp on tinyobjloader, compiled in
    place), every file both readers accept is loaded by it as well and compared array for array, bit for bit: the ear
    clipping of tiny_obj_loader.h:1107-1310 and the vertex order it implies are pinned to the reference, not to a twin."""
    from tests import ref_scene_lib as R
    open(os.path.join(tmp_path, "m.mtl"), "w").write("newmtl shiny\nKd 0.2 0.4 0.6\nNs 250\nnewmtl dull stuff\nKd 1 0 0\nNs 0\n")
    p = os.path.join(tmp_path, "f.obj")
    pinned = polygons = 0
    for seed in range(60):
        rng = np.random.default_rng(1000 + seed)
        with open(p, "w", newline="") as f:
            f.write(_random_obj(rng, crlf=seed % 3 == 0))
        res = []
        for reader in ("native", "python"):
            try:
                res.append(load_obj(p, reader=reader))
            except (ValueError, IndexError) as ex:
                res.append(type(ex))
        if isinstance(res[0], type) or isinstance(res[1], type):
            assert isinstance(res[0], type) and isinstance(res[1], type), (seed, res)
            continue
        _same_scene(res[0], res[1])
        if R.available():
            ref, mine = R.load(p), R.flatten(res[0])
            for k, v in ref.items():
                if k.endswith("_n_normals"):
                    continue  # (normals are not carried: the hot path does not read them, quirk Q7)
                assert k in mine and v.shape == mine[k].shape and v.tobytes() == mine[k].astype(v.dtype).tobytes(), (seed, k)
            pinned += 1
            polygons += sum(len(l.split()) > 4 for l in open(p).read().splitlines() if l.startswith("f"))
    if R.available():
        assert pinned >= 40 and polygons >= 200, (pinned, polygons)


def test_numbers_outside_the_reference_grammar_are_refused(tmp_path):
    """tinyobjloader's tryParseDouble (util/tiny_obj_loader.h:567-680) wants a digit after the optional sign: `.5`, `-.5`,
    `inf`, `nan` fail there and the reference silently reads 0.0. Both readers refuse such a file instead of loading
    something the reference would not."""
    p = os.path.join(tmp_path, "n.obj")
    for tok in (".5", "-.5", "+.25", "inf", "nan", "1_0", "0x10", "1.5abc", "1e", "e5"):
        open(p, "w").write(f"v 0 0 0\nv 1 0 {tok}\nv 0 1 0\nf 1 2 3\n")
        for reader in ("native", "python"):
            with pytest.raises(ValueError):
                load_obj(p, reader=reader)
    for tok, val in (("1.", 1.0), ("+2.50", 2.5), ("-3e-1", -0.3), ("4.E+1", 40.0), ("007", 7.0)):
        open(p, "w").write(f"v 0 0 0\nv 1 0 {tok}\nv 0 1 0\nf 1 2 3\n")
        for reader in ("native", "python"):
            assert load_obj(p, reader=reader).meshes[0].geometries[0].vertices[1][2] == np.float32(val)


def test_native_reader_refuses_garbage_without_crashing(tmp_path):
    """Random bytes, a truncated file, absurd indices: an error (ValueError) or an empty result, never a crash."""
    rng = np.random.default_rng(7)
    p = os.path.join(tmp_path, "g.obj")
    blobs = [rng.integers(0, 256, size=4096, dtype=np.uint8).tobytes(),
             b"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 99999999999\n",
             b"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 -7\n",
             b"f 1 2 3\n",
             b"v 1 2\nv\nf\nf 1\nf 1 2\nvt\nusemtl\nmtllib\n",
             b"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1/ 2/ 3/\nf 1/2/3/4 2 3\n",
             b"v 1e400 -1e400 nan\nv inf 0 0\nv 0 0 0\nf 1 2 3\n" + b"f 1 2 3" * 1]
    for blob in blobs:
        open(p, "wb").write(blob)
        for reader in ("native", "python"):
            try:
                sc = load_obj(p, reader=reader)
                assert sc.total_tris() >= 0
            except (ValueError, IndexError, UnicodeDecodeError, OverflowError):
                pass


@pytest.mark.parametrize("face, ok", [
    ("f 1/1/1 2/2/1 3/3/1", True), ("f -3/-3/-1 -2/-2/-1 -1/-1/-1", True),
    ("f 0 1 2", False),                     # 0 is not an index
    ("f 1 2 4", False), ("f 1 2 -4", False),  # position beyond what has been read, either way
    ("f 1/4 2/1 3/2", False), ("f 1/-5 2/1 3/2", False),  # texcoord: a negative index below the first one is an error, not "no vt"
    ("f 1//2 2//1 3//1", True), ("f 1//-2 2//1 3//1", True),  # a normal index is only a re-indexing key (tinyobj checks it for 0 only)
    ("f 1//0 2//1 3//1", False),
    ("f 4294967297 2 3", False),            # must not wrap to index 0 through a 32-bit narrowing
])
def test_native_reader_checks_every_face_index_before_narrowing(tmp_path, face, ok):
    """0 anywhere, a v / vt index beyond the arrays read so far (positive or negative), or one too large for 32 bits ->
    'face index out of range', from the native reader and from its Python twin alike (round-3 advisor finding). Three
    positions, three texcoords, one normal are defined."""
    path = tmp_path / "idx.obj"
    path.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nvn 0 0 1\n" + face + "\n")
    if ok:
        sc = load_obj(str(path))
        assert sum(len(g.indices) for g in sc.meshes[0].geometries) == 1
        assert sum(len(g.indices) for g in load_obj(str(path), reader="python").meshes[0].geometries) == 1
    else:
        for reader in ("native", "python"):
            with pytest.raises(ValueError, match="face index out of range"):
                load_obj(str(path), reader=reader)


def test_random_mtl_files_against_the_live_reference_importer(tmp_path):
    """120 seeded MTL files exercising tinyobjloader's LoadMtl (util/tiny_obj_loader.h:1353-1725) where it is least obvious --
    defaults (zero diffuse), `Kd` with 0 ... 4 components, tokens that are not numbers (`spectral`, `.5`, `2e`, `1.5abc`),
    duplicate names, statements before the first `newmtl`, a bare `newmtl`, `map_Kd` with every option and file names with
    blanks, CR LF, no final newline, an empty file -- each with an OBJ that uses the materials, loaded by the reference's own
    importer and by load_obj: materials, material ids and textures bit for bit."""
    from PIL import Image as PILImage
    from tests import ref_scene_lib as R
    if not R.available():
        pytest.skip("oracle/_ref/libref_scene.so is built where /root/reference exists")
    d = str(tmp_path)
    for k, n in enumerate(["t0.png", "t1.png", "my tex.png"]):
        PILImage.fromarray(np.random.default_rng(k).integers(0, 256, (4, 4, 3), dtype=np.uint8)).save(os.path.join(d, n))
    for seed in range(120):
        rng = np.random.default_rng(7000 + seed)

        def num(lo=0.0, hi=1.0):
            x = float(rng.uniform(lo, hi))
            return rng.choice([f"{x:.4f}", f"{x:.3e}", f"{x:.6g}", f"{x:+.3f}", f"{x:.3f}".replace("0.", ".", 1), "spectral", "1.5abc", "2e", "7."])
        lines, names = [], []
        if rng.random() < 0.3:
            lines.append("Kd 0.1 0.2 0.3")
        for m in range(int(rng.integers(0, 5))):
            nm = str(rng.choice(["a", "b c", f"mat_{m}", "a"]))
            names.append(nm)
            lines.append(str(rng.choice(["newmtl ", "newmtl\t", "  newmtl "])) + nm + str(rng.choice(["", " ", "\t"])))
            for _ in range(int(rng.integers(0, 6))):
                k = rng.choice(["Kd", "Ns", "map_Kd", "junk", "#c", "opt", "sp", "after", "Kdx", "bare"])
                lines.append({"Kd": str(rng.choice(["Kd ", "Kd\t", "  Kd  "])) + " ".join(num() for _ in range(int(rng.choice([3, 3, 1, 2, 0, 4])))),
                              "Ns": "Ns " + num(0, 900), "map_Kd": f"map_Kd t{rng.integers(0, 2)}.png", "junk": "foo bar 1 2", "#c": " # comment",
                              "opt": f"map_Kd -s 2 2 1 -o 0.5 0 0 -clamp on -bm 2 -mm 0 1 -imfchan r t{rng.integers(0, 2)}.png",
                              "sp": "map_Kd my tex.png", "after": "map_Kd -blendu off t0.png", "Kdx": "Kdx 1 1 1", "bare": "newmtl"}[str(k)])
        eol = str(rng.choice(["\n", "\r\n"]))
        with open(os.path.join(d, "m.mtl"), "w", newline="") as f:
            f.write(eol.join(lines) + (eol if rng.random() < 0.7 else ""))
        obj = ["mtllib m.mtl", "v 0 0 0", "v 1 0 0", "v 0 1 0", "vt 0 0", "vt 1 0", "vt 0 1"]
        for g in range(int(rng.integers(1, 4))):
            obj += [f"o g{g}", "usemtl " + str(rng.choice(names + ["nope"])), "f 1/1 2/2 3/3"]
        p = os.path.join(d, "f.obj")
        with open(p, "w") as f:
            f.write("\n".join(obj) + "\n")
        ref, mine = R.load(p), R.flatten(load_obj(p))
        for k, v in ref.items():
            if not k.endswith("_n_normals"):
                assert k in mine and v.shape == mine[k].shape and v.tobytes() == np.asarray(mine[k]).astype(v.dtype).tobytes(), (seed, k)


def test_missing_material_library_is_a_warning_not_an_error(tmp_path):
    """An OBJ whose `mtllib` names a file that is not there (common in the wild): tinyobjloader warns, tries the next name on
    the line, and with none left uses no materials -- the reference renders the file with its default material
    (util/tiny_obj_loader.h:1741-1749, 2031-2057; util/scene.cpp:110 throws on err only). Both readers do the same, and where
    the reference's importer is built here, produce its arrays bit for bit."""
    from tests import ref_scene_lib as R
    d = str(tmp_path)
    open(os.path.join(d, "real.mtl"), "w").write("newmtl red\nKd 1 0 0\nNs 10\n")
    body = "v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\ng a\nusemtl red\nf 1 2 3\ng b\nusemtl blue\nf 2 4 3\n"
    for k, first in enumerate(("mtllib nothere.mtl", "mtllib nothere.mtl real.mtl", "mtllib my materials.mtl", "mtllib nothere.mtl\nmtllib real.mtl",
                               "mtllib real.mtl nothere.mtl")):
        p = os.path.join(d, f"f{k}.obj")
        open(p, "w").write(first + "\n" + body)
        with pytest.warns(UserWarning) if k in (0, 2) else contextlib.nullcontext():
            twin = load_obj(p, reader="python")
        native = load_obj(p)
        _same_scene(native, twin)
        ids = [int(x) for pm in native.parameterized_meshes for x in pm.material_ids]
        if k in (0, 2):  # no library at all: nothing but the importer's default material
            assert len(native.materials) == 1 and set(ids) == {0}
        else:            # `red` is found in real.mtl, `blue` nowhere
            assert len(set(ids)) == 2
        if R.available():
            ref, mine = R.load(p), R.flatten(native)
            for key, v in ref.items():
                if not key.endswith("_n_normals"):
                    assert key in mine and v.shape == mine[key].shape and v.tobytes() == np.asarray(mine[key]).astype(v.dtype).tobytes(), (k, key)


def test_statement_level_quirks_against_the_live_reference_importer(tmp_path):
    """80 seeded OBJ files about STATEMENTS rather than numbers: bare `o` / `g` / `usemtl` (not statements for tinyobjloader: a
    keyword needs a blank after it), `usemtl` names with a second blank before or a blank after them (they name another
    material, i.e. none), `mtllib` with two names, with two blanks (the first name is then empty and the OBJ's directory is
    "read"), a second `mtllib` further down that redefines a name (the first definition stays), `g` with several names, `s`,
    `vp`, vertices with 4 and 6 values, texture coordinates with 1 and 3. Native reader == Python twin == reference, every array."""
    from tests import ref_scene_lib as R
    if not R.available():
        pytest.skip("oracle/_ref/libref_scene.so is built where /root/reference exists")
    d = str(tmp_path)
    open(os.path.join(d, "m.mtl"), "w").write("newmtl shiny\nKd 0.2 0.4 0.6\nNs 250\nnewmtl dull stuff\nKd 1 0 0\nNs 0\n")
    open(os.path.join(d, "n.mtl"), "w").write("newmtl other\nKd 0 0 1\nnewmtl shiny\nKd 1 1 1\n")
    p = os.path.join(d, "f.obj")
    for seed in range(80):
        rng = np.random.default_rng(3000 + seed)
        num = lambda: f"{float(rng.normal()):.5g}"
        lines, nv, nvt = ["vn 0 0 1"], 0, 0
        if rng.random() < 0.8:
            lines.append(str(rng.choice(["mtllib m.mtl", "mtllib m.mtl n.mtl", "mtllib  m.mtl", "mtllib n.mtl", "mtllib nothere.mtl",
                                         "mtllib nothere.mtl n.mtl", "mtllib nothere.mtl alsonot.mtl", "mtllib m.mtl "])))
        for _ in range(int(rng.integers(1, 4))):
            for _ in range(int(rng.integers(3, 8))):
                lines.append("v " + " ".join(num() for _ in range(int(rng.choice([3, 3, 3, 4, 6])))))
                nv += 1
            for _ in range(int(rng.integers(1, 4))):
                lines.append("vt " + " ".join(num() for _ in range(int(rng.choice([2, 2, 3, 1])))))
                nvt += 1
            lines.append(str(rng.choice(["o a", "g a b", "g", "o", "g a", "s 1", "s off", "o  spaced  name "])))
            if rng.random() < 0.3:
                lines.append("mtllib n.mtl")
            if rng.random() < 0.8:
                lines.append(str(rng.choice(["usemtl shiny", "usemtl  shiny", "usemtl shiny ", "usemtl dull stuff", "usemtl other", "usemtl\tshiny", "usemtl"])))
            for _ in range(int(rng.integers(1, 5))):
                lines.append("f " + " ".join(f"{int(rng.integers(1, nv + 1))}/{int(rng.integers(1, nvt + 1))}" for _ in range(int(rng.choice([3, 3, 4])))))
                if rng.random() < 0.2:
                    lines.append(str(rng.choice(["s 2", "g", "usemtl dull stuff", "# c", "vp 0 0"])))
        with open(p, "w") as f:
            f.write("\n".join(lines) + "\n")
        native, twin = load_obj(p), load_obj(p, reader="python")
        _same_scene(native, twin)
        ref, mine = R.load(p), R.flatten(native)
        for k, v in ref.items():
            if not k.endswith("_n_normals"):
                assert k in mine and v.shape == mine[k].shape and v.tobytes() == np.asarray(mine[k]).astype(v.dtype).tobytes(), (seed, k)
