"""One-shot GPU check of the world tree (CRT_HIP_LEVELS=world) without pytest / torch start-up cost: the assertions of
tests/test_gpu_world_tree.py, then frame times of a workload built both ways. Writes as it goes.

    python tools/gpu_world_tree_check.py [C4:tex_size=256] [frames]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chameleonrt_amd import core, scenes
from chameleonrt_amd.render_hip import RenderHIP
from tests import oracle_lib as oracle
from tests.parity import awkward_instances, camera_of, probe_rays, slot_triangles

OUT = open(os.path.join("gpurun_out", "world_tree_check.txt"), "a") if os.path.isdir("gpurun_out") else sys.stdout


def say(*a):
    line = " ".join(str(x) for x in a)
    print(line, flush=True)
    if OUT is not sys.stdout:
        OUT.write(line + "\n"); OUT.flush()


def renderer(sc, levels, w, h, flags=core.FLAG_COUNTERS):
    os.environ["CRT_HIP_LEVELS"] = levels
    r = RenderHIP(flags=flags); r.initialize(w, h)
    t = time.time(); r.set_scene(sc); dt = time.time() - t
    del os.environ["CRT_HIP_LEVELS"]
    return r, dt


def check(name, sc, w, h):
    ok = True
    def expect(cond, what):
        nonlocal ok
        ok = ok and bool(cond)
        say(f"  [{'ok' if cond else 'FAIL'}] {name}: {what}")
    r, _ = renderer(sc, "world", w, h)
    bvh = r.bvh()
    expect(bvh["levels"] == 2 and slot_triangles(bvh).sum() == sc.total_tris(), "world tree built, every (instance, triangle) in exactly one leaf slot")
    o = oracle.OracleScene(sc)
    org, dirs = probe_rays(sc, 30000, seed=41)
    g = r.trace(org, dirs, 0.0, 1e20, closest=True)
    c = o.trace(org, dirs, 0.0, 1e20, closest=True, brute_force=True)
    hit = c["inst"] >= 0
    expect(all(np.array_equal(g[k], c[k]) for k in ("inst", "geom", "prim")), f"closest-hit ids == brute force ({int(hit.sum())} hits)")
    expect(all(np.array_equal(g[k][hit].view(np.uint32), c[k][hit].view(np.uint32)) for k in ("t", "u", "v")), "t, u, v bit-identical")
    wk = oracle.walk_product_bvh(bvh, org, dirs, 0.0, 1e20, closest=True)
    expect((g["stats"].closest_nodes, g["stats"].closest_tris) == (wk["nodes"], wk["tris"]),
           f"closest-hit counters == oracle walk ({g['stats'].closest_nodes} / {wk['nodes']} nodes, {g['stats'].closest_tris} / {wk['tris']} tris)")
    tmax = np.random.default_rng(42).random(len(org)).astype(np.float32) * 10
    g = r.trace(org, dirs, 1e-4, tmax, closest=False)
    c = o.trace(org, dirs, 1e-4, tmax, closest=False, brute_force=True)
    expect(np.array_equal(g["t"], c["t"]), "occlusion == brute force")
    wk = oracle.walk_product_bvh(bvh, org, dirs, 1e-4, tmax, closest=False)
    expect((g["stats"].shadow_nodes, g["stats"].shadow_tris) == (wk["nodes"], wk["tris"]), "occlusion counters == oracle walk")
    e, d, u, fov = camera_of(sc)
    frames = {}
    for levels, rr in (("world", r), ("two", renderer(sc, "two", w, h)[0])):
        for f in range(2):
            rr.render(e, d, u, fov, f == 0, True)
        frames[levels] = (rr.accum().copy(), rr.ray_counts().copy(), rr.img.copy())
        rr.close()
    a, b = frames["world"], frames["two"]
    expect(np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), "accumulated radiance of 2 frames: world tree == two-level, bit for bit")
    expect(np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[1].sum() > 0, "ray counts and RGBA8 equal")
    return ok


def main():
    say("== world tree check", time.strftime("%H:%M:%S"))
    ok = check("grove", scenes.instanced_grove(), 320, 200)
    ok = check("sanmiguel_small_instanced", scenes.sanmiguel_like(detail=0.02, tex_size=64, n_trees=100, leaves_per_tree=300,
                                                                  n_instanced=64, glass=True, spp=2), 320, 180) and ok
    ok = check("awkward_instances", awkward_instances(), 256, 160) and ok
    say("PARITY", "GREEN" if ok else "RED")
    which = sys.argv[1] if len(sys.argv) > 1 else "C4:tex_size=256"
    nframes = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    over = {}
    if ":" in which:
        which, _, kv = which.partition(":")
        over = {k: int(v) for k, v in (x.split("=") for x in kv.split(","))}
    t = time.time(); sc, w, h, spp = scenes.make_workload(which, **over); say("scene gen", round(time.time() - t, 1), "s", sc.total_tris(), "tris", over)
    cam = sc.cameras[0]
    e, d, u, fov = camera_of(sc)
    for levels in ("two", "world"):
        r, dt = renderer(sc, levels, w, h, flags=core.FLAG_TIMING)
        ms = []
        for f in range(nframes):
            st = r.render(e, d, u, fov, f == 0, False)
            ms.append(st.render_time_ms)
            last = st
        say(f"{which} levels={levels}: set_scene {dt:.2f} s, frames {' '.join(f'{x:.1f}' for x in ms)} ms, best {min(ms):.2f} ms, "
            f"closest {last.closest_ms:.2f} shadow {last.shadow_ms:.2f} shade {last.shade_ms:.2f}, {last.rays} rays, {last.rays / min(ms) / 1e3:.0f} MRay/s")
        r.close()


if __name__ == "__main__":
    main()
