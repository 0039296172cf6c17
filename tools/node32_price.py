"""Price a 32-byte, two-request BVH4 node WITHOUT building it (round-5 review item 3, step (i)).

    python tools/node32_price.py C4 [n_rays]

Today's packed node (crt_types.h PNode) spends 32 bytes on its frame and planes -- origin 3 x 16 bits, one 4-bit shift per axis,
4 children x 6 planes x 8 bits -- and 16 more on four child references: three 16-byte requests per visit. A one-sector node has to
hold the references in those 32 bytes as ONE base (children contiguous) + a 4-bit leaf mask, which takes 36 bits out of the planes:
7-bit planes (-24 bits) and ONE shift shared by the three axes (-8 bits), the 4 spare bits of today's frame for the mask.

This script re-quantises the product's own packed tree to candidate plane formats IN today's record layout (bytes <= 127, the three
shift codes equal, ...), so that the oracle's walker of the product's arrays -- the rule the instrumented kernels must equal --
counts node and leaf-slot visits per ray on it, for camera rays, two generations of bounce rays and occlusion rays of the workload
(the ray sets of tools/tree_cost.py). Boxes only grow (lo down, hi up to multiples of the new step), so hits are unchanged.
Development tool: the oracle only provides hit points and the walker.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chameleonrt_amd import scenes  # noqa: E402
from chameleonrt_amd.render_hip import PreparedScene  # noqa: E402
from tests import oracle_lib  # noqa: E402
from tools.tree_cost import ray_sets  # noqa: E402


def requantise(nodes, bits, shared):
    """nodes: (n, 16) uint32 PNodes. Returns a copy whose planes have `bits` bits (as bytes 0 .. 2^bits - 1) and, if `shared`, one
    shift for the three axes."""
    n = nodes.astype(np.uint32).copy()
    f0, f1 = n[:, 0], n[:, 1]
    origin = np.stack([f0 & 0xffff, f0 >> 16, f1 & 0xffff], 1).astype(np.int64)          # (n, 3) grid units
    code = np.stack([(f1 >> 16) & 31, (f1 >> 21) & 31, (f1 >> 26) & 31], 1).astype(np.int64)
    assert (code & 1).sum() == 0, "half steps are off in the product"
    e = code >> 1
    lo = np.stack([n[:, 2], n[:, 4], n[:, 6]], 1)  # lo_x, lo_y, lo_z
    hi = np.stack([n[:, 3], n[:, 5], n[:, 7]], 1)
    sh = np.arange(4, dtype=np.uint32) * 8
    lo_b = ((lo[:, :, None] >> sh) & 0xff).astype(np.int64)   # (n, 3, 4)
    hi_b = ((hi[:, :, None] >> sh) & 0xff).astype(np.int64)
    used = lo_b[:, 0, :] <= hi_b[:, 0, :]                      # (n, 4): an unused slot is inverted on every axis
    lo_g = lo_b << e[:, :, None]                               # offsets from the origin in grid units
    hi_g = hi_b << e[:, :, None]
    top = (1 << bits) - 1
    ext = np.where(used[:, None, :], hi_g, 0).max(axis=2)     # (n, 3) extent per axis (origin = smallest lo -> offset 0)
    e2 = np.zeros_like(e)
    for _ in range(16):
        grow = ((ext + (1 << e2) - 1) >> e2) > top
        if not grow.any():
            break
        e2 = e2 + grow
    if shared:
        e2 = np.repeat(e2.max(axis=1, keepdims=True), 3, axis=1)
    lo2 = lo_g >> e2[:, :, None]
    hi2 = (hi_g + (1 << e2[:, :, None]) - 1) >> e2[:, :, None]
    assert (hi2[np.broadcast_to(used[:, None, :], hi2.shape)] <= top).all()
    lo2 = np.where(used[:, None, :], lo2, 255)  # (the walker rejects lo > hi whatever the values)
    hi2 = np.where(used[:, None, :], hi2, 0)
    pack = lambda b: (b.astype(np.uint32) << sh).sum(axis=2, dtype=np.uint32)
    lo_w, hi_w = pack(lo2), pack(hi2)
    n[:, 2], n[:, 4], n[:, 6] = lo_w[:, 0], lo_w[:, 1], lo_w[:, 2]
    n[:, 3], n[:, 5], n[:, 7] = hi_w[:, 0], hi_w[:, 1], hi_w[:, 2]
    c2 = (e2 << 1).astype(np.uint32)
    n[:, 1] = (f1 & 0xffff) | (c2[:, 0] << 16) | (c2[:, 1] << 21) | (c2[:, 2] << 26)
    return n


def main():
    name = sys.argv[1]
    n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
    over = {"tex_size": 8} if name in ("C4", "C4F", "C5", "C2") else {}
    sc, w, h, spp = scenes.make_workload(name, **over)
    tag = name + "".join(f"_{k}{v}" for k, v in sorted(over.items()))
    rays = ray_sets(sc, n_rays, f"/tmp/treecost_{tag}_{n_rays}.npz")
    t0 = time.time()
    ps = PreparedScene(sc)
    bvh = ps.bvh()
    print(f"{name}: prepare {time.time() - t0:.1f} s, {bvh['nodes'].shape[0]} nodes, {bvh['tris'].shape[0]} leaf slots", flush=True)
    base_nodes = bvh["nodes"]
    sets = (("p", True, 0.0), ("b1", True, 1e-4), ("b2", True, 1e-4), ("s", False, 1e-4))
    ref = None
    for label, fmt in (("8-bit planes, shift per axis (today, 48 B)", None), ("8-bit, ONE shift", (8, True)), ("7-bit, shift per axis", (7, False)),
                       ("7-bit, ONE shift (the 32-byte candidate)", (7, True)), ("6-bit, ONE shift", (6, True))):
        bvh["nodes"] = base_nodes if fmt is None else requantise(base_nodes, *fmt)
        row, tot_n, tot_s = [], 0.0, 0.0
        for key, closest, tmin in sets:
            o, dd = rays[key + "_org"], rays[key + "_dir"]
            tmax = rays["s_tmax"] if key == "s" else 1e20
            r = oracle_lib.walk_product_bvh(bvh, o, dd, tmin, tmax, closest=closest)
            row.append((r["nodes"] / len(o), r["slots"] / len(o)))
            tot_n += r["nodes"] / len(o)
            tot_s += r["slots"] / len(o)
        if ref is None:
            ref = (row, tot_n, tot_s)
        rel = lambda a, b: f"{(a / b - 1) * 100:+5.1f} %"
        print(f"  {label:44s} nodes/ray " + " ".join(f"{k} {x[0]:6.2f}" for (k, _, _), x in zip(sets, row)) +
              " | slots/ray " + " ".join(f"{x[1]:5.2f}" for x in row) +
              f" | sum nodes {tot_n:7.2f} ({rel(tot_n, ref[1])}) slots {tot_s:6.2f} ({rel(tot_s, ref[2])}) lines {rel(tot_n + tot_s, ref[1] + ref[2])}"
              f" | closest (p+b1+b2) nodes {rel(sum(x[0] for x in row[:3]), sum(x[0] for x in ref[0][:3]))}, occlusion nodes {rel(row[3][0], ref[0][3][0])}", flush=True)


if __name__ == "__main__":
    main()
