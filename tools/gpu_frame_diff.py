"""Where do frames of one scene rendered from two structures (world tree / two-level) differ?  GPU debugging aid.

    python tools/gpu_frame_diff.py [awkward|grove]

Renders frame 0 (and 1) from both structures and with the oracle, reports differing pixels; repeats the world-tree frame
to see whether it is deterministic; then runs the production kernels on probe rays, rays leaving surfaces and occlusion
segments in both structures against brute force and prints the first mismatches.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chameleonrt_amd.render_hip import RenderHIP
from tests import oracle_lib as oracle
from tests.parity import awkward_instances, camera_of, probe_rays


def renderer(sc, levels, w, h):
    os.environ["CRT_HIP_LEVELS"] = levels
    r = RenderHIP(); r.initialize(w, h); r.set_scene(sc)
    del os.environ["CRT_HIP_LEVELS"]
    return r


def main():
    sc = awkward_instances()
    w, h = 256, 160
    e, d, u, fov = camera_of(sc)
    o = oracle.OracleScene(sc)
    rs = {lv: renderer(sc, lv, w, h) for lv in ("world", "two")}
    acc = {}
    for lv, r in rs.items():
        for f in range(2):
            r.render(e, d, u, fov, f == 0, True)
            acc[(lv, f)] = (r.accum().copy(), r.ray_counts().copy())
    for f in range(2):
        a, b = acc[("world", f)], acc[("two", f)]
        bad = (a[0].view(np.uint32) != b[0].view(np.uint32)).any(axis=-1)
        print(f"frame {f}: {int(bad.sum())} pixels differ between world tree and two-level; ray counts differ on {int((a[1] != b[1]).sum())}")
        for y, x in list(zip(*np.nonzero(bad)))[:8]:
            print(f"   ({x},{y}) world {a[0][y, x]} two {b[0][y, x]} rays {a[1][y, x]} / {b[1][y, x]}")
    # determinism of either structure
    for lv, r in rs.items():
        r.render(e, d, u, fov, True, True)
        again = r.accum().copy()
        print(f"{lv}: frame 0 rendered again differs on {int((again.view(np.uint32) != acc[(lv, 0)][0].view(np.uint32)).any(axis=-1).sum())} pixels")
    # against the oracle (tolerance of tests/parity.py)
    orr = oracle.OracleRenderer(sc, w, h)
    orr.render(e, d, u, fov, True)
    ref, ref_rays = orr.accum(), orr.ray_counts()
    for lv in rs:
        g = acc[(lv, 0)]
        err = (np.abs(g[0] - ref) > 1e-4 + 1e-3 * np.abs(ref)).any(axis=-1)
        print(f"{lv} vs oracle, frame 0: {int(err.sum())} pixels out of tolerance, ray counts differ on {int((g[1] != ref_rays).sum())}")
        for y, x in list(zip(*np.nonzero(err)))[:8]:
            print(f"   ({x},{y}) {lv} {g[0][y, x]} oracle {ref[y, x]} rays {g[1][y, x]} / {ref_rays[y, x]}")
    # production kernels on rays
    org, dirs = probe_rays(sc, 60000, seed=41)
    c = o.trace(org, dirs, 0.0, 1e20, closest=True, brute_force=True)
    hit = c["inst"] >= 0
    p = org[hit] + c["t"][hit, None] * dirs[hit]
    d2 = np.random.default_rng(17).normal(size=p.shape).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    c2 = o.trace(p, d2, 1e-4, 1e20, closest=True, brute_force=True)
    tmax = np.random.default_rng(18).random(len(p)).astype(np.float32) * 10
    c3 = o.trace(p, d2, 1e-4, tmax, closest=False, brute_force=True)

    def cmp(tag, g, cc):
        hh = cc["inst"] >= 0
        bad = np.zeros(len(hh), bool)
        for k in ("inst", "geom", "prim"):
            if k in g:
                bad |= g[k] != cc[k]
        for k in ("t", "u", "v"):
            bad |= hh & (g[k].view(np.uint32) != cc[k].view(np.uint32))
        print(f"{tag}: {int(bad.sum())} of {len(hh)} rays differ from brute force")
        for i in np.nonzero(bad)[0][:6]:
            print("   ray", i, {k: (g[k][i], cc[k][i]) for k in ("t", "u", "v", "inst", "geom", "prim") if k in g})

    for lv, r in rs.items():
        for prod in (False, True):
            cmp(f"{lv} production={prod} primary", r.trace(org, dirs, 0.0, 1e20, closest=True, production=prod), c)
            cmp(f"{lv} production={prod} secondary", r.trace(p, d2, 1e-4, 1e20, closest=True, production=prod), c2)
            g = r.trace(p, d2, 1e-4, tmax, closest=False, production=prod)
            print(f"{lv} production={prod} occlusion: {int((g['t'] != c3['t']).sum())} of {len(p)} differ")
    for r in rs.values():
        r.close()


if __name__ == "__main__":
    main()
