"""CPU: chameleonrt_amd's scene importers (obj_io, gltf_io, crts_io) pinned to the REFERENCE's own
importer -- `Scene::Scene(fname, material_mode)`, util/scene.cpp:49-624 with util/flatten_gltf.cpp,
mesh.cpp, material.cpp, util.cpp and the vendored tinyobjloader / tinygltf / stb_image / json.

tests/golden/refscene_*.npz are dumps of what that code, compiled from the reference tree against a
GLM stand-in (oracle/Makefile target `ref`, tests/golden/make_scene_golden.py), makes of the files
under tests/golden/scenes/. Where oracle/_ref/libref_scene.so exists (the development container)
the live library is compared too. The bar: every array bit for bit -- vertices after the
(position, normal, uv) re-indexing, index buffers, material ids, the 16-float materials with the
texture handles in their bits, 8-bit texels after the flip / 4-channel rules, the generated or
loaded lights, cameras, and the instance transforms, including those that went through matrix
products (glTF TRS nodes, flattened node trees: gltf_io._mat4_mul multiplies in GLM's order).
"""
import glob
import os

import numpy as np
import pytest

from chameleonrt_amd.crts_io import load_crts
from chameleonrt_amd.gltf_io import load_gltf
from chameleonrt_amd.obj_io import load_obj
from tests import ref_scene_lib as R

HERE = os.path.dirname(os.path.abspath(__file__))
SCENES = os.path.join(HERE, "golden", "scenes")
FILES = {"obj_cornell": "cornell.obj", "obj_atrium": "atrium.obj", "obj_quirks": "quirks.obj", "obj_bare": "bare.obj", "obj_polygons": "polygons.obj",
         "gltf_scene": "scene.gltf", "glb_scene": "scene.glb", "gltf_tree": "tree.gltf", "crts_handmade": "handmade.crts",
         "crts_nolight": "nolight.crts", "crts_grove": "grove.crts"}


def _ours(path, white_diffuse):
    mode = "white_diffuse" if white_diffuse else "default"
    ext = path.rsplit(".", 1)[1]
    load = {"obj": load_obj, "gltf": load_gltf, "glb": load_gltf, "crts": load_crts}[ext]
    return R.flatten(load(path, material_mode=mode))


def _compare(ref, mine, what):
    assert list(ref["counts"]) == list(mine["counts"]), f"{what}: counts (mesh pmesh inst mat tex light cam)"
    for k in ref:
        if k.endswith("_n_normals"):
            continue  # the hot path never reads normals (quirk Q7); our Geometry does not carry them
        assert k in mine, f"{what}: {k} missing"
        a, b = np.asarray(ref[k]), np.asarray(mine[k])
        assert a.shape == b.shape, f"{what}: {k} shape {a.shape} vs {b.shape}"
        if a.dtype == np.float32:
            same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
            assert same, f"{what}: {k} differs (max abs {np.nanmax(np.abs(a.astype(np.float64) - b.astype(np.float64)))})"
        else:
            assert np.array_equal(a, b), f"{what}: {k}"


@pytest.mark.parametrize("white_diffuse", [False, True], ids=["default", "white_diffuse"])
@pytest.mark.parametrize("name", list(FILES))
def test_importer_equals_reference_dump(name, white_diffuse):
    ref = dict(np.load(os.path.join(HERE, "golden", f"refscene_{name}{'_wd' if white_diffuse else ''}.npz")))
    _compare(ref, _ours(os.path.join(SCENES, FILES[name]), white_diffuse), name)


@pytest.mark.skipif(not R.available(), reason="oracle/_ref/libref_scene.so is built where /root/reference exists")
@pytest.mark.parametrize("name", list(FILES))
def test_dumps_are_what_the_reference_importer_produces_now(name):
    """The committed dumps are not stale: the live reference importer still produces them."""
    path = os.path.join(SCENES, FILES[name])
    for wd in (False, True):
        live = R.load(path, white_diffuse=wd)
        gold = dict(np.load(os.path.join(HERE, "golden", f"refscene_{name}{'_wd' if wd else ''}.npz")))
        assert set(live) == set(gold)
        for k in live:
            assert np.array_equal(np.asarray(live[k]).view(np.uint8), np.asarray(gold[k]).view(np.uint8)), (name, k)


def test_every_golden_scene_file_is_covered():
    dumps = {os.path.basename(p)[len("refscene_"):-4] for p in glob.glob(os.path.join(HERE, "golden", "refscene_*.npz"))}
    assert dumps == {n + s for n in FILES for s in ("", "_wd")}
