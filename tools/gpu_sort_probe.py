"""What would ray REORDERING buy? (GPU box; development aid, not product.)

Renders one frame of a workload, copies the real incoherent rays the frame left in its queues (bounce-3 closest-hit rays and
the bounce-3 occlusion rays, pixel order as the frame had them) and re-traces them through the PRODUCTION traversal
kernels (crt_hip_trace_rays + CRT_HIP_TRACE_PRODUCTION) in other orders: shuffled, and sorted by keys of increasing
resolution (direction octant, Morton cell of the origin). Reports kernel ms per order -- the upper bound of what a sort
between bounces could gain, before a single line of the sort is written.

    python tools/gpu_sort_probe.py C4 [max_rays]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C

import numpy as np

from chameleonrt_amd import core, scenes
from chameleonrt_amd.camera import look_at
from chameleonrt_amd.render_hip import RenderHIP


def part1by2(x):
    x = x.astype(np.uint64) & np.uint64(0x1fffff)
    x = (x | (x << np.uint64(32))) & np.uint64(0x1f00000000ffff)
    x = (x | (x << np.uint64(16))) & np.uint64(0x1f0000ff0000ff)
    x = (x | (x << np.uint64(8))) & np.uint64(0x100f00f00f00f00f)
    x = (x | (x << np.uint64(4))) & np.uint64(0x10c30c30c30c30c3)
    x = (x | (x << np.uint64(2))) & np.uint64(0x1249249249249249)
    return x


def morton(o, lo, hi, bits):
    q = np.clip(((o - lo) / (hi - lo) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    return part1by2(q[:, 0]) | (part1by2(q[:, 1]) << np.uint64(1)) | (part1by2(q[:, 2]) << np.uint64(2))


def octant(d):
    return ((d[:, 0] < 0).astype(np.uint64) | ((d[:, 1] < 0).astype(np.uint64) << np.uint64(1)) |
            ((d[:, 2] < 0).astype(np.uint64) << np.uint64(2)))


def dir_cell(d, bits):
    """octahedral map of the direction, Morton-interleaved 2 x bits"""
    a = np.abs(d).sum(axis=1, keepdims=True)
    p = d[:, :2] / a
    neg = d[:, 2] < 0
    px = np.where(neg, (1 - np.abs(p[:, 1])) * np.sign(p[:, 0] + 1e-30), p[:, 0])
    py = np.where(neg, (1 - np.abs(p[:, 0])) * np.sign(p[:, 1] + 1e-30), p[:, 1])
    qx = np.clip(((px * 0.5 + 0.5) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    qy = np.clip(((py * 0.5 + 0.5) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    out = np.zeros(len(d), np.uint64)
    for b in range(bits):
        out |= ((qx >> b) & 1).astype(np.uint64) << np.uint64(2 * b)
        out |= ((qy >> b) & 1).astype(np.uint64) << np.uint64(2 * b + 1)
    return out


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "C4"
    max_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 6_000_000
    t = time.time()
    sc, w, h, spp = scenes.make_workload(which, tex_size=64) if which.startswith("C4") else scenes.make_workload(which)
    print("scene gen", round(time.time() - t, 1), "s", sc.total_tris(), "tris", w, h, spp, flush=True)
    r = RenderHIP(flags=core.FLAG_TIMING)
    r.initialize(w, h)
    r.set_scene(sc)
    cam = sc.cameras[0]
    e, d, u = look_at(cam.position, cam.center, cam.up)
    for f in range(2):
        st = r.render(e, d, u, cam.fov_y, f == 0, False)
    nb = list(st.closest_rays_bounce)
    ns = list(st.shadow_rays_bounce)
    print("frame", round(st.render_time_ms, 2), "ms; closest rays per bounce", nb, "ms", [round(x, 3) for x in st.closest_ms_bounce],
          "shadow", ns, [round(x, 3) for x in st.shadow_ms_bounce], flush=True)
    lib = core.load()

    def copy(which_q, first, n, nf):
        out = np.zeros((n, nf), np.float32)
        core.check(r._ctx, lib.crt_hip_debug_copy_queue(r._ctx, which_q, first, n, core.fptr(out)), "debug_copy_queue")
        return out

    n3 = int(nb[3])
    take = min(max_rays, n3)
    first = (n3 - take) // 2
    rays = copy(1, first, take, 6)  # bounce 3's closest-hit rays: PathQueue buffer 1
    n4s, n3s = int(ns[4]), int(ns[3])
    take_s = min(max_rays, n3s - n4s)
    first_s = n4s + (n3s - n4s - take_s) // 2
    srays = copy(2, first_s, take_s, 7)  # bounce 3's occlusion rays sit behind bounce 4's in ShadowQueueA
    rng = np.random.default_rng(7)

    def orders(o, dd):
        lo, hi = o.min(axis=0), o.max(axis=0)
        hi = np.maximum(hi, lo + 1e-6)
        n = len(o)
        oc = octant(dd)
        res = [("as the frame has them (pixel order)", np.arange(n)), ("shuffled", rng.permutation(n)),
               ("octant only (stable)", np.argsort(oc, kind="stable"))]
        for ob in (3, 4, 5, 7, 10):
            key = (oc << np.uint64(3 * ob)) | morton(o, lo, hi, ob)
            res.append((f"octant | origin morton {ob} bits/axis ({3 + 3 * ob} bit key, stable)", np.argsort(key, kind="stable")))
        key = (oc << np.uint64(9)) | morton(o, lo, hi, 3)
        sh = rng.permutation(n)
        res.append(("octant | morton 3 bits/axis, random order inside a bin", sh[np.argsort(key[sh], kind="stable")]))
        key = (oc << np.uint64(12)) | morton(o, lo, hi, 4)
        res.append(("octant | morton 4 bits/axis, random order inside a bin", sh[np.argsort(key[sh], kind="stable")]))
        for ob in (4, 7, 10):
            key = (morton(o, lo, hi, ob) << np.uint64(3)) | oc
            res.append((f"origin morton {ob} bits/axis | octant (stable)", np.argsort(key, kind="stable")))
        key = morton(o, lo, hi, 10)
        res.append(("origin morton 10 bits/axis, no direction", np.argsort(key, kind="stable")))
        for ob, db in ((5, 2), (7, 3)):
            key = (dir_cell(dd, db) << np.uint64(3 * ob)) | morton(o, lo, hi, ob)
            res.append((f"direction cell {4 ** db} | origin morton {ob} bits/axis", np.argsort(key, kind="stable")))
            key = (morton(o, lo, hi, ob) << np.uint64(2 * db)) | dir_cell(dd, db)
            res.append((f"origin morton {ob} bits/axis | direction cell {4 ** db}", np.argsort(key, kind="stable")))
        return res

    print(f"\nclosest-hit rays of bounce 3: {take} of {n3}")
    base = None
    for name, perm in orders(rays[:, :3], rays[:, 3:6]):
        o, dd = rays[perm, :3], rays[perm, 3:6]
        ms = []
        for _ in range(2):
            g = r.trace(o, dd, 1e-4, 1e20, closest=True, production=True)
            ms.append(g["stats"].render_time_ms)
        base = base or min(ms)
        print(f"  {min(ms):8.3f} ms ({min(ms) / base:5.2f}x)  {take / min(ms) / 1e3:7.1f} MRay/s  {name}", flush=True)
    print(f"\nocclusion rays of bounce 3: {take_s} of {n3s - n4s}")
    base = None
    for name, perm in orders(srays[:, :3], srays[:, 3:6]):
        o, dd, tm = srays[perm, :3], srays[perm, 3:6], srays[perm, 6]
        ms = []
        for _ in range(2):
            g = r.trace(o, dd, 1e-4, tm, closest=False, production=True)
            ms.append(g["stats"].render_time_ms)
        base = base or min(ms)
        print(f"  {min(ms):8.3f} ms ({min(ms) / base:5.2f}x)  {take_s / min(ms) / 1e3:7.1f} MRay/s  {name}", flush=True)
    r.close()


if __name__ == "__main__":
    main()
