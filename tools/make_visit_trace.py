"""Record the visit sequences of a workload's real secondary rays for tools/node_replay_microbench.hip (no GPU needed).

    python tools/make_visit_trace.py C4 [n_rays] -> build/trace_C4.bin

The oracle's walker of the product's own packed tree (the rule the instrumented kernels must equal) walks the bounce-1 and
bounce-2 closest-hit rays and the occlusion rays of tools/tree_cost.py's ray sets with ORC_WALK_TRACE set; the three traces are
concatenated (closest b1, closest b2, occlusion). The file also says how many nodes and leaf slots the tree has, so that the
microbenchmark sizes its buffers like the real arrays. Development tool."""
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chameleonrt_amd import scenes  # noqa: E402
from chameleonrt_amd.render_hip import PreparedScene  # noqa: E402
from tests import oracle_lib  # noqa: E402
from tools.tree_cost import ray_sets  # noqa: E402


def main():
    name = sys.argv[1]
    n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
    over = {"tex_size": 8} if name in ("C4", "C4F", "C5", "C2") else {}
    sc, w, h, spp = scenes.make_workload(name, **over)
    tag = name + "".join(f"_{k}{v}" for k, v in sorted(over.items()))
    rays = ray_sets(sc, n_rays, f"/tmp/treecost_{tag}_{n_rays}.npz")
    bvh = PreparedScene(sc).bvh()
    offs, visits = [np.zeros(1, np.uint32)], []
    tmp = "/tmp/orc_walk_trace.bin"
    for key, closest, tmin in (("b1", True, 1e-4), ("b2", True, 1e-4), ("s", False, 1e-4)):
        o, dd = rays[key + "_org"], rays[key + "_dir"]
        tmax = rays["s_tmax"] if key == "s" else 1e20
        os.environ["ORC_WALK_TRACE"] = tmp
        r = oracle_lib.walk_product_bvh(bvh, o, dd, tmin, tmax, closest=closest)
        del os.environ["ORC_WALK_TRACE"]
        raw = np.fromfile(tmp, np.uint32)
        n, nv = int(raw[0]), int(raw[1])
        off, vis = raw[2:3 + n], raw[3 + n:3 + n + nv]
        assert nv == r["nodes"] + r["slots"] and len(vis) == nv
        base = sum(len(v) for v in visits)
        offs.append((off[1:].astype(np.uint64) + base).astype(np.uint32))
        visits.append(vis)
        print(f"{name} {key}: {n} rays, {r['nodes'] / n:.2f} nodes + {r['slots'] / n:.2f} leaf slots per ray")
    offs, visits = np.concatenate(offs), np.concatenate(visits)
    os.makedirs("build", exist_ok=True)
    out = f"build/trace_{name}.bin"
    with open(out, "wb") as f:
        f.write(struct.pack("<4I", len(offs) - 1, len(visits), bvh["nodes"].shape[0], bvh["tris"].shape[0]))
        offs.tofile(f)
        visits.tofile(f)
    print(f"{out}: {len(offs) - 1} rays, {len(visits)} visits, tree of {bvh['nodes'].shape[0]} nodes + {bvh['tris'].shape[0]} leaf slots")


if __name__ == "__main__":
    main()
