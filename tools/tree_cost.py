"""Price a BVH build WITHOUT a GPU: node / triangle visits per ray of the product's traversal rule on
the product's own arrays (the oracle's walker of them, which the instrumented kernels must equal --
tests/test_gpu_traversal.py), for camera rays, diffuse-bounce rays and occlusion rays of a workload.

    python tools/tree_cost.py C4 [n_rays] [key=value scene overrides ...]

Builder knobs are read from the environment by the library (CRT_BVH_*, CRT_HIP_NO_GRAFT, ...), so an A/B is
two runs of this script. The ray sets are cached under /tmp (they depend on the scene only).
Development tool: the oracle only provides hit points and the walker.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chameleonrt_amd import scenes  # noqa: E402
from chameleonrt_amd.camera import look_at  # noqa: E402
from chameleonrt_amd.render_hip import PreparedScene  # noqa: E402
from tests import oracle_lib  # noqa: E402


def ray_sets(sc, n, cache):
    if os.path.exists(cache):
        z = np.load(cache)
        return {k: z[k] for k in z.files}
    cam = sc.cameras[0]
    e, d, u = look_at(cam.position, cam.center, cam.up)
    rng = np.random.default_rng(1)
    w, h = 1920, 1080
    px, py = rng.random(n) * w, rng.random(n) * h
    plane_y = 2 * np.tan(np.radians(0.5 * cam.fov_y))
    plane_x = plane_y * w / h
    du = np.cross(d, u); du /= np.linalg.norm(du); du *= plane_x
    dv = np.cross(du, d); dv /= np.linalg.norm(dv); dv *= -plane_y
    tl = d - 0.5 * du - 0.5 * dv
    dirs = (px[:, None] / w) * du + (py[:, None] / h) * dv + tl
    dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
    org = np.tile(np.asarray(e, np.float32), (n, 1))
    osc = oracle_lib.OracleScene(sc)
    out = dict(p_org=org, p_dir=dirs)
    o, dd = org, dirs
    for bounce in (1, 2):  # two generations of uniformly scattered rays from the hit points
        hit = osc.trace(o, dd, 0.0 if bounce == 1 else 1e-4, 1e20, closest=True)
        ok = hit["inst"] >= 0
        p = (o + hit["t"][:, None] * dd)[ok]
        b = rng.normal(size=p.shape).astype(np.float32)
        b /= np.linalg.norm(b, axis=1, keepdims=True)
        o = (p - 1e-3 * dd[ok]).astype(np.float32)
        dd = b
        out[f"b{bounce}_org"], out[f"b{bounce}_dir"] = o, dd
        if bounce == 1:
            l0 = sc.lights[0]
            lp = (np.asarray(l0.position, np.float32)[:3] if hasattr(l0, "position")
                  else np.asarray(l0, np.float32).reshape(-1)[4:7])
            to_l = lp[None, :] - o
            dist = np.linalg.norm(to_l, axis=1).astype(np.float32)
            out["s_org"], out["s_dir"], out["s_tmax"] = o, (to_l / dist[:, None]).astype(np.float32), dist
    np.savez(cache, **out)
    return out


def flattened(sc):
    """The same triangles with every instance baked into ONE mesh (materials dropped): what a tree over the
    whole scene costs, i.e. the floor for any way of arranging the two levels."""
    from chameleonrt_amd.scene import Geometry, Instance, Mesh, ParameterizedMesh, Scene, disney_material
    geoms = []
    for it in sc.instances:
        m = np.asarray(it.transform, np.float32).reshape(4, 4).T  # column-major in the record
        for g in sc.meshes[sc.parameterized_meshes[it.parameterized_mesh_id].mesh_id].geometries:
            v = np.asarray(g.vertices, np.float32)
            w = (v @ m[:3, :3].T + m[:3, 3]).astype(np.float32)
            geoms.append(Geometry(w, np.asarray(g.indices, np.uint32), None))
    out = Scene(meshes=[Mesh(geoms)], parameterized_meshes=[ParameterizedMesh(0, [0] * len(geoms))],
                instances=[Instance(np.eye(4, dtype=np.float32).reshape(16), 0)], materials=[disney_material()],
                lights=sc.lights, cameras=sc.cameras, samples_per_pixel=sc.samples_per_pixel, name=sc.name + "+flat")
    return out


def main():
    name = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 and "=" not in sys.argv[2] else 100_000
    over = {}
    for a in sys.argv[2:]:
        if "=" in a:
            k, v = a.split("=")
            over[k] = float(v) if "." in v else int(v)
    if name in ("C4", "C4F", "C5", "C2"):
        over.setdefault("tex_size", 8)
    flat = over.pop("flatten", 0)
    sc, w, h, spp = scenes.make_workload(name, **over)
    tag = name + "".join(f"_{k}{v}" for k, v in sorted(over.items()))
    rays = ray_sets(sc, n, f"/tmp/treecost_{tag}_{n}.npz")
    tag += "_flattened" if flat else ""
    if flat:
        sc = flattened(sc)
    t0 = time.time()
    ps = PreparedScene(sc)
    bvh = ps.bvh()
    print(f"{tag}: prepare {time.time() - t0:.2f} s, {bvh['nodes'].shape[0]} nodes, {bvh['tris'].shape[0]} leaf slots, "
          f"{bvh['n_instances']} instances, stack_need {bvh['stack_need']}")
    total = entries = 0.0
    for key, closest, tmin in (("p", True, 0.0), ("b1", True, 1e-4), ("b2", True, 1e-4), ("s", False, 1e-4)):
        o, dd = rays[key + "_org"], rays[key + "_dir"]
        tmax = rays["s_tmax"] if key == "s" else 1e20
        r = oracle_lib.walk_product_bvh(bvh, o, dd, tmin, tmax, closest=closest)
        m = len(o)
        # a dependent "line visit" = one 64-byte node or one 64-byte leaf slot (1-2 triangles) fetched
        print(f"  {key:3s} {m:7d} rays: nodes/ray {r['nodes'] / m:7.2f}  leaf slots/ray {r['slots'] / m:6.2f}  tris/ray {r['tris'] / m:6.2f}  "
              f"lines/ray {(r['nodes'] + r['slots']) / m:7.2f}  instance entries/ray {r['inst_entries'] / m:5.2f}  "
              f"max stack {r['max_stack']}")
        total += (r["nodes"] + r["slots"]) / m
        entries += r["inst_entries"] / m
    print(f"  sum over the four sets: lines/ray {total:.2f}, instance entries/ray {entries:.2f}")


if __name__ == "__main__":
    main()
