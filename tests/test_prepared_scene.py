"""CPU: the host half of set_scene (crt_hip_prepare_scene: SAH build, 4-wide collapse, 16-bit
quantisation, leaf-order triangle records, instance records) checked WITHOUT a device.

The oracle walks the product's host-built arrays with the product's visit rule
(orc_walk_foreign_bvh) and must find exactly what the oracle finds by testing every triangle with
no BVH at all -- so builder, quantiser and the documented rule are pinned on the CPU, and the GPU
tests only have to show that the kernels walk these arrays the same way (same hits, same visit
counts: tests/test_gpu_traversal.py, tests/test_gpu_scale.py).
"""
import ctypes as C
import os

import numpy as np
import pytest

from chameleonrt_amd import core, scenes
from chameleonrt_amd.render_hip import PreparedScene
from chameleonrt_amd.scene import PackedScene
from tests.parity import awkward_instances, node_refs, node_used, probe_rays, slot_triangles

SCENES = {
    "cornell": lambda: scenes.cornell(),
    "grove_two_level": lambda: scenes.instanced_grove(),
    "sponza_small": lambda: scenes.sponza_like(detail=0.05, tex_size=32),
    "rungholt_small": lambda: scenes.rungholt_like(n=160),
    "sanmiguel_small_instanced": lambda: scenes.sanmiguel_like(detail=0.01, tex_size=16, n_trees=60, leaves_per_tree=200,
                                                               n_instanced=40, glass=True),
}


@pytest.fixture(scope="module", autouse=True)
def _two_level_structure():
    """This module checks the top-level tree over instances (and its grafting); the world tree the library builds by
    default for instanced scenes that fit its memory budget has its own tests below, which override this."""
    old = os.environ.get("CRT_HIP_LEVELS")
    os.environ["CRT_HIP_LEVELS"] = "two"
    yield
    if old is None:
        os.environ.pop("CRT_HIP_LEVELS", None)
    else:
        os.environ["CRT_HIP_LEVELS"] = old


@pytest.fixture(scope="module", params=list(SCENES))
def prepared(request, oracle):
    sc = SCENES[request.param]()
    ps = PreparedScene(sc)
    yield sc, ps.bvh(), oracle.OracleScene(sc), ps
    ps.close()


def test_walk_of_host_built_bvh_equals_brute_force(prepared, oracle):
    sc, bvh, o, _ = prepared
    org, dirs = probe_rays(sc, 8000, seed=21)
    w = oracle.walk_product_bvh(bvh, org, dirs, 0.0, 1e20, closest=True)
    c = o.trace(org, dirs, 0.0, 1e20, closest=True, brute_force=True)
    for k in ("inst", "geom", "prim"):
        assert np.array_equal(w[k], c[k]), k
    hit = c["inst"] >= 0
    assert hit.sum() > 50
    assert np.array_equal(w["t"][hit].view(np.uint32), c["t"][hit].view(np.uint32))
    assert w["max_stack"] <= bvh["stack_need"]
    # occlusion rays with finite segments, tnear = EPSILON
    tmax = np.random.default_rng(22).random(len(org)).astype(np.float32) * 10
    w = oracle.walk_product_bvh(bvh, org, dirs, 1e-4, tmax, closest=False)
    c = o.trace(org, dirs, 1e-4, tmax, closest=False, brute_force=True)
    assert np.array_equal(w["t"], c["t"])


def test_instanced_scene_layout(prepared):
    sc, bvh, _, _ = prepared
    assert bvh["n_instances"] == len(sc.instances)
    assert bvh["two_level"] == (len(sc.instances) > 1)
    assert slot_triangles(bvh).sum() == sum(m.num_tris() for m in sc.meshes)  # one BLAS per Mesh, shared by its instances
    assert bvh["child_order"] in (0, 1)


def test_world_tree_is_the_default_within_its_memory_budget(monkeypatch):
    sc = SCENES["grove_two_level"]()
    monkeypatch.delenv("CRT_HIP_LEVELS")
    ps = PreparedScene(sc)
    assert ps.levels() == 2
    ps.close()
    monkeypatch.setenv("CRT_HIP_WORLD_TREE_MAX_TRIS", str(sc.total_tris() - 1))
    ps = PreparedScene(sc)
    assert ps.levels() == 1
    ps.close()
    ps = PreparedScene(SCENES["cornell"]())
    assert ps.levels() == 0
    ps.close()


@pytest.mark.parametrize("name", ["grove_two_level", "sanmiguel_small_instanced"])
def test_world_tree_of_an_instanced_scene(name, oracle, monkeypatch):
    """CRT_HIP_LEVELS=world: ONE tree in world space over per-instance copies of the triangle records (crt_types.h
    LEVELS_WORLD_TREE). Every (instance, triangle) pair has exactly one record, tagged (instance << 1) | identity;
    the walk -- world-space ray for the boxes, the triangle's instance space for the test -- finds what brute force
    over the instances finds, bit for bit (ids, t, occlusion), and so what the two-level walk finds, in fewer node
    visits; leaves hold at most two triangles (the kernels' world-tree leaf step handles no more)."""
    sc = SCENES[name]()
    monkeypatch.setenv("CRT_HIP_LEVELS", "two")
    ps = PreparedScene(sc)
    two = ps.bvh()
    ps.close()
    monkeypatch.setenv("CRT_HIP_LEVELS", "world")
    monkeypatch.setenv("CRT_BVH_MAX_LEAF", "4")  # read once per process; harmless if a test before this one fixed it at 2
    ps = PreparedScene(sc)
    bvh = ps.bvh()
    ps.close()
    assert two["levels"] == 1 and bvh["levels"] == 2 and not bvh["two_level"] and bvh["world_inst"] == -1
    assert bvh["n_instances"] == len(sc.instances) and bvh["root"] == 0
    tris_of = lambda i: sc.meshes[sc.parameterized_meshes[sc.instances[i].parameterized_mesh_id].mesh_id].num_tris()
    assert slot_triangles(bvh).sum() == sum(tris_of(i) for i in range(len(sc.instances)))
    tag = bvh["tris"][:, 15].view(np.uint32)
    per_inst = np.bincount(tag >> 1, weights=slot_triangles(bvh), minlength=len(sc.instances)).astype(np.int64)
    assert np.array_equal(per_inst, [tris_of(i) for i in range(len(sc.instances))])
    ident = np.array([np.array_equal(np.asarray(it.transform, np.float32).reshape(4, 4), np.eye(4, dtype=np.float32))
                      for it in sc.instances])
    assert np.array_equal((tag & 1).astype(bool), ident[tag >> 1])
    refs = node_refs(bvh["nodes"])
    leaves = refs[refs < 0]
    assert ((~leaves & 7) <= 3).all(), "leaves of at most CRT_BVH_MAX_LEAF slots"
    o = oracle.OracleScene(sc)
    org, dirs = probe_rays(sc, 8000, seed=33)
    w = oracle.walk_product_bvh(bvh, org, dirs, 0.0, 1e20, closest=True)
    t2 = oracle.walk_product_bvh(two, org, dirs, 0.0, 1e20, closest=True)
    c = o.trace(org, dirs, 0.0, 1e20, closest=True, brute_force=True)
    for k in ("inst", "geom", "prim"):
        assert np.array_equal(w[k], c[k]), k
        assert np.array_equal(w[k], t2[k]), k
    hit = c["inst"] >= 0
    assert (c["inst"][hit] == 0).any() and (c["inst"][hit] > 0).any()
    assert np.array_equal(w["t"][hit].view(np.uint32), c["t"][hit].view(np.uint32))
    assert w["max_stack"] <= bvh["stack_need"]
    assert w["nodes"] < 1.02 * t2["nodes"], "no second descent per instance"
    tmax = np.random.default_rng(34).random(len(org)).astype(np.float32) * 10
    w = oracle.walk_product_bvh(bvh, org, dirs, 1e-4, tmax, closest=False)
    c = o.trace(org, dirs, 1e-4, tmax, closest=False, brute_force=True)
    assert np.array_equal(w["t"], c["t"])


def test_world_tree_edge_cases(oracle, monkeypatch, tmp_path):
    """A world tree over awkward instances: a MIRRORED one (negative determinant: the object-space test sees a
    left-handed ray frame), a strongly non-uniform scale, two instances at exactly the same place (exact ties in t
    go to the lower instance id), an identity instance of a mesh that transformed instances share, and one mesh under
    two ParameterizedMeshes. Hits equal brute force bit for bit; the prepared scene survives save / load."""
    sc = awkward_instances()
    monkeypatch.setenv("CRT_HIP_LEVELS", "world")
    ps = PreparedScene(sc)
    bvh = ps.bvh()
    assert bvh["levels"] == 2 and slot_triangles(bvh).sum() == 2 + 5 * 400
    o = oracle.OracleScene(sc)
    org, dirs = probe_rays(sc, 12000, seed=3, spread=0.5)
    w = oracle.walk_product_bvh(bvh, org, dirs, 0.0, 1e20, closest=True)
    c = o.trace(org, dirs, 0.0, 1e20, closest=True, brute_force=True)
    for k in ("inst", "geom", "prim"):
        assert np.array_equal(w[k], c[k]), k
    hit = c["inst"] >= 0
    assert set(np.unique(c["inst"][hit])) == {0, 1, 2, 3, 4}, "every instance is hit, the coinciding pair only through the lower id"
    assert np.array_equal(w["t"][hit].view(np.uint32), c["t"][hit].view(np.uint32))
    tmax = np.random.default_rng(4).random(len(org)).astype(np.float32) * 12
    w = oracle.walk_product_bvh(bvh, org, dirs, 1e-4, tmax, closest=False)
    c = o.trace(org, dirs, 1e-4, tmax, closest=False, brute_force=True)
    assert np.array_equal(w["t"], c["t"])
    path = str(tmp_path / "world.bin")
    ps.save(path)
    back = PreparedScene(path=path)
    b2 = back.bvh()
    assert back.levels() == 2
    for k in ("nodes", "tris", "instances", "frame"):
        assert np.array_equal(bvh[k], b2[k]), k
    back.close()
    ps.close()


@pytest.mark.parametrize("levels", ["world", "two"])
def test_badly_conditioned_instances_lose_no_hits(levels, oracle, monkeypatch):
    """Instances stretched 2000 : 1 (and squeezed 1 : 2000) far from the origin: what lies under a world-space box is
    intersected with a transformed, i.e. rounded, ray, so the boxes must be padded by the transform's conditioning
    (scene_prepare.cpp instance_pad) on top of the outward quantisation. Rays aimed AT the surfaces: every brute-force
    hit must be found through the tree, whichever structure."""
    from chameleonrt_amd.scene import Camera, Geometry, Instance, Mesh, ParameterizedMesh, Scene, disney_material, obj_default_light
    n, stretch, shift = 40, 2000.0, 800.0
    u, v = np.meshgrid(np.linspace(-1, 1, n), np.linspace(-1, 1, n), indexing="ij")
    verts = np.stack([u.ravel(), 0.05 * np.sin(3 * u.ravel()) * np.cos(2 * v.ravel()), v.ravel()], 1).astype(np.float32)
    idx = [[i * n + j + d for d in tri] for i in range(n - 1) for j in range(n - 1) for tri in ((0, 1, n), (1, n + 1, n))]
    patch = Mesh([Geometry(verts, np.array(idx, np.uint32), None)])

    def trs(t, s, ry):
        m = np.eye(4, dtype=np.float32)
        c, si = np.cos(ry), np.sin(ry)
        m[:3, :3] = np.array([[c, 0, si], [0, 1, 0], [-si, 0, c]], np.float32) @ np.diag(np.asarray(s, np.float32))
        m[:3, 3] = t
        return m.T.reshape(16).astype(np.float32)

    insts = [Instance(trs([shift, 0, shift], [stretch, 1.0, 1.0], 0.6), 0),
             Instance(trs([shift, 3, shift + 5], [1.0, 1.0, 1.0 / stretch], 1.2), 0),
             Instance(trs([shift - 3, -2, shift], [30.0, 30.0, 30.0], 0.1), 0)]
    sc = Scene(meshes=[patch], parameterized_meshes=[ParameterizedMesh(0, [0])], instances=insts, materials=[disney_material()],
               lights=[obj_default_light()],
               cameras=[Camera(np.array([shift, 40, shift + 60], np.float32), np.array([shift, 0, shift], np.float32),
                               np.array([0, 1, 0], np.float32), 50.0)])
    rng = np.random.default_rng(1)
    per = 12000
    tgt = []
    for it in sc.instances:
        m = np.asarray(it.transform, np.float32).reshape(4, 4).T
        p = np.stack([rng.uniform(-1, 1, per), np.zeros(per), rng.uniform(-1, 1, per)], 1).astype(np.float32)
        tgt.append((p @ m[:3, :3].T + m[:3, 3]).astype(np.float32))
    tgt = np.concatenate(tgt)
    org = (np.asarray(sc.cameras[0].position, np.float32) + rng.normal(size=tgt.shape).astype(np.float32) * 5).astype(np.float32)
    dirs = tgt - org
    dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
    c = oracle.OracleScene(sc).trace(org, dirs, 0.0, 1e20, closest=True, brute_force=True)
    monkeypatch.setenv("CRT_HIP_LEVELS", levels)
    ps = PreparedScene(sc)
    bvh = ps.bvh()
    ps.close()
    w = oracle.walk_product_bvh(bvh, org, dirs, 0.0, 1e20, closest=True)
    hit = c["inst"] >= 0
    assert hit.mean() > 0.9
    for k in ("inst", "geom", "prim"):
        assert np.array_equal(w[k], c[k]), k
    assert np.array_equal(w["t"][hit].view(np.uint32), c["t"][hit].view(np.uint32))


@pytest.mark.parametrize("name", ["grove_two_level", "sanmiguel_small_instanced"])
def test_static_instance_is_grafted_into_the_top_level_tree(name, oracle, monkeypatch):
    """An identity instance whose mesh nothing else uses (the ground of the grove, the courtyard of the
    San-Miguel-like scene) is not entered like an instance: its BLAS is cut open and the subtrees hang in the
    top-level tree next to the other instances (scene_prepare.cpp, crt_types.h). Top-level leaves are
    instances iff their count field is 7, none of them names the grafted instance; the hits are those of the
    ungrafted build (CRT_HIP_NO_GRAFT), which the other tests of this file compare with brute force, bit for bit;
    rays need fewer instance entries, seen here as fewer node visits for rays that start on surfaces."""
    sc = SCENES[name]()
    ps = PreparedScene(sc)
    bvh = ps.bvh()
    ps.close()
    monkeypatch.setenv("CRT_HIP_NO_GRAFT", "1")
    ps = PreparedScene(sc)
    plain = ps.bvh()
    ps.close()
    monkeypatch.delenv("CRT_HIP_NO_GRAFT")
    # a mesh built on the device arrives quantised in its own frame; the hook hands the host-built mesh over in
    # that form, so the read-back of its boxes and the re-quantisation into the top-level frame run here too
    monkeypatch.setenv("CRT_HIP_GRAFT_QNODES", "1")
    ps = PreparedScene(sc)
    via_q = ps.bvh()
    ps.close()
    monkeypatch.delenv("CRT_HIP_GRAFT_QNODES")
    # (the top-level tree over read-back boxes, which are a quantum wider, may decide a split differently)
    assert via_q["world_inst"] == 0 and abs(via_q["nodes"].shape[0] - bvh["nodes"].shape[0]) <= 0.01 * bvh["nodes"].shape[0]
    assert plain["world_inst"] == -1 and bvh["world_inst"] == 0 and bvh["two_level"]
    refs = node_refs(bvh["nodes"])
    used = node_used(bvh["nodes"])
    # walk the top level: everything reachable from the root without passing an instance leaf
    seen_inst, stack, top_tri_leaves, visited = set(), [bvh["root"]], 0, set()
    while stack:
        n = stack.pop()
        if n in visited:
            continue
        visited.add(n)
        for k in range(4):
            if not used[n, k]:
                continue
            r = int(refs[n, k])
            if r >= 0:
                stack.append(r)
            elif (~r & 7) == 7:
                seen_inst.add((~r & 0xFFFFFFFF) >> 3)
            else:
                top_tri_leaves += 1
    assert seen_inst == set(range(1, len(sc.instances))), "every other instance is a leaf of the top-level tree, the grafted one is not"
    assert top_tri_leaves > 0
    org, dirs = probe_rays(sc, 6000, seed=5)
    a = oracle.walk_product_bvh(bvh, org, dirs, 0.0, 1e20, closest=True)
    b = oracle.walk_product_bvh(plain, org, dirs, 0.0, 1e20, closest=True)
    for k in ("inst", "geom", "prim"):
        assert np.array_equal(a[k], b[k]), k
    hit = a["inst"] >= 0
    assert (a["inst"][hit] == 0).any() and (a["inst"][hit] > 0).any()
    assert np.array_equal(a["t"][hit].view(np.uint32), b["t"][hit].view(np.uint32))
    q = oracle.walk_product_bvh(via_q, org, dirs, 0.0, 1e20, closest=True)
    for k in ("inst", "geom", "prim"):
        assert np.array_equal(q[k], b[k]), k
    assert np.array_equal(q["t"][hit].view(np.uint32), b["t"][hit].view(np.uint32))
    assert a["nodes"] <= q["nodes"] <= 1.05 * a["nodes"]  # boxes rounded outward a second time: a few more visits
    # rays leaving surfaces (what four of five bounces and every occlusion ray are)
    p = (org[hit] + dirs[hit] * a["t"][hit, None]).astype(np.float32)
    d2 = np.random.default_rng(8).normal(size=p.shape).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    a2 = oracle.walk_product_bvh(bvh, p, d2, 1e-3, 1e20, closest=True)
    b2 = oracle.walk_product_bvh(plain, p, d2, 1e-3, 1e20, closest=True)
    for k in ("inst", "geom", "prim"):
        assert np.array_equal(a2[k], b2[k]), k
    # (the grove's grafted ground is two triangles: nothing to gain there, nothing to lose either)
    assert a2["nodes"] <= 1.02 * b2["nodes"]
    if name == "sanmiguel_small_instanced":
        assert a2["nodes"] < b2["nodes"]


def test_save_load_round_trip(prepared, tmp_path):
    _, bvh, _, ps = prepared
    path = str(tmp_path / "prepared.bin")
    ps.save(path)
    back = PreparedScene(path=path)
    b2 = back.bvh()
    for k in ("nodes", "tris", "instances", "frame"):
        assert np.array_equal(bvh[k], b2[k]), k
    for k in ("root", "n_top_nodes", "stack_need", "n_instances", "two_level", "world_inst"):
        assert bvh[k] == b2[k], k
    back.close()
    # damaged files are refused with a message, never a crash or an exception across the C ABI (round-2 advisor finding):
    # a truncated file, a header whose counts promise more than the file holds, ids that point outside the arrays
    good = open(path, "rb").read()

    def refused(data, what):
        with open(path, "wb") as f:
            f.write(data)
        with pytest.raises(core.CoreError, match="prepared scene"):
            PreparedScene(path=path)

    refused(b"\0" * 8 + good[8:], "magic")
    refused(good[:len(good) // 2], "truncated")
    refused(good + b"\0" * 64, "trailing bytes")
    huge = bytearray(good)
    huge[16:24] = (2 ** 62).to_bytes(8, "little")  # n_nodes
    refused(bytes(huge), "absurd count")
    bad_root = bytearray(good)
    off_root = 16 + 8 * 8 + 24  # magic, abi, eight counts, the root frame
    bad_root[off_root:off_root + 4] = (2 ** 30).to_bytes(4, "little")
    refused(bytes(bad_root), "root outside the node array")
    with open(path, "wb") as f:
        f.write(good)
    PreparedScene(path=path).close()


def test_malformed_scenes_are_refused_not_crashed():
    """The header promises CRT_HIP_EINVAL for a malformed scene and no crash across the C ABI: NULL
    arrays with non-zero counts, mesh ranges that wrap in 32 bits, out-of-range ids."""
    L = core.load()
    sc = scenes.cornell()

    def refused(mutate):
        packed = PackedScene(sc)
        mutate(packed.desc)
        h = L.crt_hip_prepare_scene(C.byref(packed.desc), 1)
        assert not h, "malformed scene accepted"
        assert L.crt_hip_last_error(None)

    def null_instances(d):
        d.instances = None

    def null_materials(d):
        d.materials = None

    def null_lights(d):
        d.lights = None

    def null_vertices(d):
        d.geometries[0].vertices = None

    def null_indices(d):
        d.geometries[0].indices = None

    def null_material_ids(d):
        d.parameterized_meshes[0].material_ids = None

    def wrapped_mesh_range(d):
        d.meshes[0].first_geometry = 0xFFFFFFFF
        d.meshes[0].n_geometries = 2

    def bad_instance(d):
        d.instances[0].parameterized_mesh_id = 7

    for m in (null_instances, null_materials, null_lights, null_vertices, null_indices, null_material_ids,
              wrapped_mesh_range, bad_instance):
        refused(m)


@pytest.mark.parametrize("builder", ["lbvh", "ploc"])
@pytest.mark.parametrize("name", ["grove_two_level", "sponza_small", "rungholt_small"])
def test_linear_bvh_build_on_the_host(name, builder, oracle, monkeypatch):
    """The device builder's algorithm (lbvh.h: Morton keys with two normalisations, Karras' radix tree,
    bottom-up boxes, area-greedy collapse to 4-wide nodes) run serially on the host: a correct tree
    (walk == brute force) of bounded extra cost against the SAH tree. `ploc`: the same pipeline with the binary tree built
    bottom-up by locally-ordered clustering instead (Meister & Bittner 2018; round 6, priced and not adopted: DESIGN.md section 7)."""
    sc = SCENES[name]()
    sah = PreparedScene(sc).bvh()
    monkeypatch.setenv("CRT_BVH_BUILDER", builder)
    lin = PreparedScene(sc).bvh()
    assert lin["tris"].shape == sah["tris"].shape
    org, dirs = probe_rays(sc, 6000, seed=23)
    c = oracle.OracleScene(sc).trace(org, dirs, 0.0, 1e20, closest=True, brute_force=True)
    w = oracle.walk_product_bvh(lin, org, dirs, 0.0, 1e20, closest=True)
    for k in ("inst", "geom", "prim"):
        assert np.array_equal(w[k], c[k]), k
    hit = c["inst"] >= 0
    assert np.array_equal(w["t"][hit].view(np.uint32), c["t"][hit].view(np.uint32))
    assert w["max_stack"] <= lin["stack_need"]
    w_sah = oracle.walk_product_bvh(sah, org, dirs, 0.0, 1e20, closest=True)
    assert w["nodes"] <= 1.5 * w_sah["nodes"], "linear BVH much worse than expected against SAH"


def _scene_with_empty_meshes(all_empty):
    from chameleonrt_amd.scene import Camera, Geometry, Instance, Mesh, ParameterizedMesh, Scene, disney_material, obj_default_light
    none = Mesh([Geometry(np.zeros((3, 3), np.float32), np.zeros((0, 3), np.uint32), None)])
    tri = Mesh([Geometry(np.array([[-1, 0, 0], [1, 0, 0], [0, 1.5, 0]], np.float32), np.array([[0, 1, 2]], np.uint32), None)])
    eye = np.eye(4, dtype=np.float32)
    moved = eye.copy()
    moved[:3, 3] = [0.5, 0.2, -1.0]
    meshes = [none] if all_empty else [none, tri]
    insts = [Instance(eye.T.reshape(16), 0), Instance(moved.T.reshape(16), 0)]
    pms = [ParameterizedMesh(0, [0])]
    if not all_empty:
        insts += [Instance(eye.T.reshape(16), 1), Instance(moved.T.reshape(16), 1)]
        pms.append(ParameterizedMesh(1, [0]))
    return Scene(meshes=meshes, parameterized_meshes=pms, instances=insts, materials=[disney_material()], lights=[obj_default_light()],
                 cameras=[Camera(np.array([0, 0.5, 3], np.float32), np.zeros(3, np.float32), np.array([0, 1, 0], np.float32), 50.0)])


@pytest.mark.parametrize("levels", ["two", "world"])
def test_meshes_and_scenes_without_triangles(levels, oracle, monkeypatch):
    """A mesh without triangles is legal input (the reference commits an empty Embree geometry) and is handled alike by
    both structures: its instances hit nothing; a scene none of whose instances has a triangle renders the miss shader.
    (Round-2 advisor finding: the two-level path used to refuse what the world tree accepted.)"""
    monkeypatch.setenv("CRT_HIP_LEVELS", levels)
    sc = _scene_with_empty_meshes(all_empty=False)
    ps = PreparedScene(sc)
    bvh = ps.bvh()
    ps.close()
    assert bvh["levels"] == (1 if levels == "two" else 2) and slot_triangles(bvh).sum() == (1 if levels == "two" else 2)  # one BLAS / one record per instance
    org, dirs = probe_rays(sc, 4000, seed=61)
    w = oracle.walk_product_bvh(bvh, org, dirs, 0.0, 1e20, closest=True)
    c = oracle.OracleScene(sc).trace(org, dirs, 0.0, 1e20, closest=True, brute_force=True)
    for k in ("inst", "geom", "prim"):
        assert np.array_equal(w[k], c[k]), k
    hit = c["inst"] >= 0
    assert hit.any() and set(np.unique(c["inst"][hit])) <= {2, 3}
    assert np.array_equal(w["t"][hit].view(np.uint32), c["t"][hit].view(np.uint32))
    empty = _scene_with_empty_meshes(all_empty=True)
    ps = PreparedScene(empty)
    bvh = ps.bvh()
    ps.close()
    assert bvh["levels"] == 0 and bvh["nodes"].shape[0] == 1 and bvh["tris"].shape[0] == 0
    w = oracle.walk_product_bvh(bvh, org, dirs, 0.0, 1e20, closest=True)
    assert (w["inst"] == -1).all() and w["tris"] == 0
    w = oracle.walk_product_bvh(bvh, org, dirs, 1e-4, np.full(len(org), 5.0, np.float32), closest=False)
    assert (w["t"] == 1.0).all()


def test_empty_scene_survives_save_and_load(tmp_path):
    """The scene without a triangle (a one-node tree, no leaf slots) goes through crt_hip_save / load_prepared_scene like any
    other: a multi-GPU job's rank 0 saves what the other ranks load, and bench.py caches it (round-3 advisor finding: the
    loader's field check refused n_tris == 0)."""
    ps = PreparedScene(_scene_with_empty_meshes(all_empty=True))
    path = str(tmp_path / "empty.bin")
    ps.save(path)
    a = ps.bvh()
    ps.close()
    back = PreparedScene(path=path)
    b = back.bvh()
    back.close()
    assert b["nodes"].shape[0] == 1 and b["tris"].shape[0] == 0 and b["levels"] == a["levels"]
    assert np.array_equal(a["nodes"], b["nodes"])


def test_leaf_slots_hold_every_triangle_once_in_its_own_vertex_order():
    """The 64-byte leaf slots (crt_types.h LeafSlot, leaf_slots.h): triangle A is (v0, v1, v2), triangle B is picked from the
    four vertices by the 2-bit selectors next to the geomID. Every triangle of a single-instance scene must sit in exactly
    one slot WITH ITS OWN VERTICES IN ITS OWN ORDER (the kernels form e1 = v0 - v1 and e2 = v2 - v0 from them, and t / u / v
    keep the reference's bits only for that order); the two triangles of a slot belong to one geometry and share an edge."""
    sc = scenes.sponza_like(detail=0.05, tex_size=8)
    assert len(sc.instances) == 1
    ps = PreparedScene(sc)
    slots = ps.bvh()["tris"]
    ps.close()
    v = slots[:, 0:12].view(np.float32).reshape(-1, 4, 3)
    geom = slots[:, 12] & ((1 << 26) - 1)
    sel = slots[:, 12] >> 26
    prim0, prim1 = slots[:, 13], slots[:, 14]
    two = prim1 != 0xffffffff
    assert two.mean() > 0.5, "a tessellated scene is mostly quads"
    pick = lambda s: v[np.arange(len(v)), s & 3]  # noqa: E731
    tri_a = np.stack([v[:, 0], v[:, 1], v[:, 2]], axis=1)
    tri_b = np.stack([pick(sel), pick(sel >> 2), pick(sel >> 4)], axis=1)
    mesh = sc.meshes[sc.parameterized_meshes[sc.instances[0].parameterized_mesh_id].mesh_id]
    seen = [np.zeros(len(g.indices), np.int32) for g in mesh.geometries]
    for g_id, g in enumerate(mesh.geometries):
        src = np.asarray(g.vertices, np.float32)[np.asarray(g.indices, np.int64)]  # (n, 3, 3): the triangles as the scene has them
        for prims, tris, mask in ((prim0, tri_a, geom == g_id), (prim1, tri_b, (geom == g_id) & two)):
            p = prims[mask].astype(np.int64)
            assert (p < len(src)).all()
            assert np.array_equal(tris[mask].view(np.uint32), src[p].view(np.uint32)), "vertices or their order differ from the scene's"
            np.add.at(seen[g_id], p, 1)
    assert all((s == 1).all() for s in seen), "a triangle is missing from the slots or sits in two"
    # the pair shares an edge: two of B's three vertices are vertices of A (bit for bit)
    shared = (tri_b[two][:, :, None, :].view(np.uint32) == tri_a[two][:, None, :, :].view(np.uint32)).all(axis=3).any(axis=2).sum(axis=1)
    assert (shared >= 2).all()
