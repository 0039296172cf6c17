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
FILES = {"obj_cornell": "cornell.obj", "obj_atrium": "atrium.obj", "obj_quirks": "quirks.obj", "obj_bare": "bare.obj", "obj_polygons": "polygons.obj", "obj_mtlquirks": "mtlquirks.obj",
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


@pytest.mark.skipif(not R.available(), reason="oracle/_ref/libref_scene.so is built where /root/reference exists")
def test_random_gltf_node_trees_against_the_live_reference_importer(tmp_path):
    """40 seeded glTF files with random node graphs -- TRS nodes (random unit and NON-unit quaternions, negative and zero-ish
    scales), explicit matrices, up to four levels of nesting, nodes without meshes in between, single-level and multi-level
    scenes (util/flatten_gltf.cpp only flattens the latter) -- loaded by the reference's own importer and by gltf_io:
    every instance transform bit for bit (read_node_transform's GLM products, flatten_gltf_node's parent * child), same
    instance order, same mesh assignment."""
    from tests.test_gltf_io import QUAD_IDX, QUAD_POS, Builder
    n_instances = 0
    for seed in range(40):
        rng = np.random.default_rng(500 + seed)
        b = Builder()
        a_pos = b.accessor(b.view(QUAD_POS.tobytes()), 5126, "VEC3", 4)
        a_idx = b.accessor(b.view(np.uint16(QUAD_IDX).tobytes()), 5123, "SCALAR", 6)
        n_meshes = int(rng.integers(1, 4))
        b.doc["meshes"] = [{"primitives": [{"attributes": {"POSITION": a_pos}, "indices": a_idx}]} for _ in range(n_meshes)]
        nodes = []

        def num(scale=1.0):
            x = float(rng.normal()) * scale
            return float(np.float32(x)) if rng.random() < 0.5 else x  # doubles that are not floats, too

        def make(depth):
            n = {}
            if rng.random() < 0.8 or depth >= 3:
                n["mesh"] = int(rng.integers(0, n_meshes))
            kind = rng.random()
            if kind < 0.2:
                m = np.eye(4) + rng.normal(size=(4, 4)) * 0.5
                m[3] = [0, 0, 0, 1]
                n["matrix"] = [float(x) for x in m.T.reshape(16)]  # column-major
            else:
                if rng.random() < 0.7:
                    n["translation"] = [num(5), num(5), num(5)]
                if rng.random() < 0.7:
                    q = rng.normal(size=4)
                    if rng.random() < 0.7:
                        q /= np.linalg.norm(q)
                    n["rotation"] = [float(x) for x in q]
                if rng.random() < 0.6:
                    n["scale"] = [num(2) if rng.random() < 0.9 else 1e-3 * num() for _ in range(3)]
            idx = len(nodes)
            nodes.append(n)
            if depth < 3 and rng.random() < (0.0 if single_level else 0.6):
                n["children"] = [make(depth + 1) for _ in range(int(rng.integers(1, 4)))]
            return idx

        single_level = seed % 4 == 0
        roots = [make(0) for _ in range(int(rng.integers(1, 4)))]
        b.doc["nodes"] = nodes
        b.doc["scenes"][0]["nodes"] = roots
        path = os.path.join(tmp_path, f"n{seed}." + ("glb" if seed % 2 else "gltf"))
        b.finish(path, glb=bool(seed % 2))
        ref, mine = R.load(path), _ours(path, False)
        for k in ("instance_transforms", "instance_pmesh", "counts"):
            assert ref[k].shape == mine[k].shape and ref[k].tobytes() == np.asarray(mine[k]).astype(ref[k].dtype).tobytes(), (seed, k)
        n_instances += len(ref["instance_pmesh"])
    assert n_instances > 150


@pytest.mark.skipif(not R.available(), reason="oracle/_ref/libref_scene.so is built where /root/reference exists")
def test_random_crts_scenes_against_the_live_reference_importer(tmp_path):
    """24 seeded instanced scenes (random instance counts, transforms, materials, textures, lights, with and without a camera /
    lights of their own) written by save_crts and loaded by the reference's load_crts (util/scene.cpp:417-624) and by crts_io:
    every array bit for bit, in both material modes."""
    from chameleonrt_amd import scenes
    from chameleonrt_amd.crts_io import save_crts
    compared = 0
    for seed in range(24):
        rng = np.random.default_rng(900 + seed)
        sc = scenes.instanced_grove(n_instances=int(rng.integers(1, 9)), leaves_per_tree=int(rng.integers(4, 30)),
                                    tex_size=int(rng.choice([2, 4, 8])), seed=seed + 1)
        for inst in sc.instances:  # arbitrary (also non-rigid, mirrored) transforms
            m = np.eye(4) + rng.normal(size=(4, 4)) * 0.3
            m[3] = [0, 0, 0, 1]
            inst.transform = np.float32(m.T.reshape(16))
        if seed % 3 == 0:
            sc.lights = []      # load_crts then generates its own (scene.cpp:605-623)
        if seed % 4 == 0:
            sc.cameras = []
        path = os.path.join(tmp_path, f"s{seed}.crts")
        save_crts(sc, path)
        for wd in (False, True):
            _compare(R.load(path, white_diffuse=wd), _ours(path, wd), f"crts fuzz seed {seed} wd {wd}")
            compared += 1
    assert compared == 48


@pytest.mark.skipif(not R.available(), reason="oracle/_ref/libref_scene.so is built where /root/reference exists")
def test_random_gltf_contents_against_the_live_reference_importer(tmp_path):
    """60 seeded glTF / GLB files about CONTENT (util/scene.cpp:230-415): 0 ... 3 embedded or data-URI PNG images of 3 and 4
    channels, textures, materials with any subset of baseColorFactor / metallicFactor / roughnessFactor / baseColorTexture /
    metallicRoughnessTexture (or none at all), meshes of 1 ... 3 primitives with interleaved or separate POSITION / TEXCOORD_0,
    16- and 32-bit indices whose count need not be a multiple of three, primitives with and without a material, nodes with and
    without meshes, a missing default scene -- in both material modes: every array of the reference's Scene bit for bit, or
    both importers refuse."""
    import base64
    from chameleonrt_amd.gltf_io import load_gltf
    from tests.test_gltf_io import Builder, _png
    d, bad = str(tmp_path), 0
    for seed in range(60):
        rng=np.random.default_rng(seed)
        b=Builder()
        n_img=int(rng.integers(0,3))
        b.doc["images"]=[]; b.doc["textures"]=[]
        for k in range(n_img):
            ch=int(rng.choice([4,4,3]))
            arr=rng.integers(0,256,(int(rng.integers(1,5)),int(rng.integers(1,5)),4),dtype=np.uint8)
            if ch==3: arr[...,3]=255
            png=_png(arr)
            if rng.random()<0.5:
                b.doc["images"].append({"bufferView": b.view(png), "mimeType":"image/png", "name": f"img{k}"})
            else:
                            b.doc["images"].append({"uri":"data:image/png;base64,"+base64.b64encode(png).decode()})
        for k in range(int(rng.integers(0,4)) if n_img else 0):
            b.doc["textures"].append({"source": int(rng.integers(0,n_img))})
        mats=[]
        for m in range(int(rng.integers(0,4))):
            pbr={}
            if rng.random()<0.7: pbr["baseColorFactor"]=[float(x) for x in rng.random(4)]
            if rng.random()<0.6: pbr["metallicFactor"]=float(rng.random())
            if rng.random()<0.6: pbr["roughnessFactor"]=float(rng.random())
            if b.doc["textures"] and rng.random()<0.5: pbr["baseColorTexture"]={"index": int(rng.integers(0,len(b.doc["textures"])))}
            if b.doc["textures"] and rng.random()<0.4: pbr["metallicRoughnessTexture"]={"index": int(rng.integers(0,len(b.doc["textures"])))}
            mat={"pbrMetallicRoughness": pbr} if (pbr or rng.random()<0.5) else {}
            if rng.random()<0.3: mat["name"]="m%d"%m
            mats.append(mat)
        if mats: b.doc["materials"]=mats
        meshes=[]
        for me in range(int(rng.integers(1,4))):
            prims=[]
            for p in range(int(rng.integers(1,4))):
                nv=int(rng.integers(3,9)); nt=int(rng.integers(1,5))
                pos=rng.normal(size=(nv,3)).astype(np.float32)
                att={}
                if rng.random()<0.5:
                    uv=rng.random((nv,2)).astype(np.float32)
                    inter=np.concatenate([pos,uv],1).astype(np.float32)
                    vi=b.view(inter.tobytes(), stride=20)
                    att["POSITION"]=b.accessor(vi,5126,"VEC3",nv); att["TEXCOORD_0"]=b.accessor(vi,5126,"VEC2",nv,offset=12)
                else:
                    att["POSITION"]=b.accessor(b.view(pos.tobytes()),5126,"VEC3",nv)
                    if rng.random()<0.4: att["TEXCOORD_0"]=b.accessor(b.view(rng.random((nv,2)).astype(np.float32).tobytes()),5126,"VEC2",nv)
                idx=rng.integers(0,nv,nt*3+int(rng.choice([0,0,1,2])))
                if rng.random()<0.5: ai=b.accessor(b.view(np.uint16(idx).tobytes()),5123,"SCALAR",len(idx))
                else: ai=b.accessor(b.view(np.uint32(idx).tobytes()),5125,"SCALAR",len(idx))
                pr={"attributes":att,"indices":ai}
                if mats and rng.random()<0.7: pr["material"]=int(rng.integers(0,len(mats)))
                if rng.random()<0.3: pr["mode"]=4
                prims.append(pr)
            meshes.append({"primitives":prims})
        b.doc["meshes"]=meshes
        nodes=[]
        for n in range(int(rng.integers(1,5))):
            nd={}
            if rng.random()<0.85: nd["mesh"]=int(rng.integers(0,len(meshes)))
            if rng.random()<0.5: nd["translation"]=[float(x) for x in rng.normal(size=3)]
            nodes.append(nd)
        b.doc["nodes"]=nodes; b.doc["scenes"][0]["nodes"]=list(range(len(nodes)))
        if rng.random()<0.3: del b.doc["scene"]
        glb=bool(seed%2)
        path=os.path.join(d,"f."+("glb" if glb else "gltf")); b.finish(path,glb=glb)
        for wd in (False,True):
            try: ref=R.load(path,white_diffuse=wd)
            except Exception as e: ref=e
            try: mine=R.flatten(load_gltf(path, material_mode="white_diffuse" if wd else "default"))
            except Exception as e: mine=e
            if isinstance(ref,Exception) or isinstance(mine,Exception):
                if not (isinstance(ref,Exception) and isinstance(mine,Exception)):
                    bad+=1; print(seed,wd,'RAISE ref',repr(ref)[:100] if isinstance(ref,Exception) else 'ok','| mine',repr(mine)[:100] if isinstance(mine,Exception) else 'ok')
                continue
            diff=[k for k in ref if not k.endswith('n_normals') and (k not in mine or ref[k].shape!=np.asarray(mine[k]).shape or ref[k].tobytes()!=np.asarray(mine[k]).astype(ref[k].dtype).tobytes())]
            if diff: bad+=1; print(seed,wd,'DIFF',diff[:5])
    assert bad == 0
