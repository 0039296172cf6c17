"""Dump a scene's triangles and realistic ray sets for tools/traverse_sim.cpp:
primary rays, diffuse-bounce rays from the primary hit points (closest-hit, incoherent) and
occlusion rays from those points towards the first light (any-hit).

    python tools/make_sim_rays.py C2 /tmp/sim      -> /tmp/sim_tris.bin, _primary.bin, _bounce.bin, _shadow.bin
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chameleonrt_amd import scenes  # noqa: E402
from chameleonrt_amd.camera import look_at  # noqa: E402
from tests import oracle_lib  # noqa: E402  (development tool: the oracle only provides hit points)


def main():
    name, out = sys.argv[1], sys.argv[2]
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 200_000
    sc = {"C2": lambda: scenes.sponza_like(tex_size=8), "C3": lambda: scenes.rungholt_like(),
          "C4": lambda: scenes.sanmiguel_like(tex_size=8)}[name]()
    assert len(sc.instances) == 1
    tris = []
    for m in sc.meshes:
        for g in m.geometries:
            v = np.asarray(g.vertices, np.float32)
            tris.append(v[np.asarray(g.indices, np.int64)].reshape(-1, 9))
    tris = np.concatenate(tris).astype(np.float32)
    tris.tofile(out + "_tris.bin")
    cam = sc.cameras[0]
    e, d, u = look_at(cam.position, cam.center, cam.up)
    rng = np.random.default_rng(1)
    w, h = 1280, 720
    px = rng.random(n) * w
    py = rng.random(n) * h
    plane_y = 2 * np.tan(np.radians(0.5 * cam.fov_y))
    plane_x = plane_y * w / h
    du = np.cross(d, u); du /= np.linalg.norm(du); du *= plane_x
    dv = np.cross(du, d); dv /= np.linalg.norm(dv); dv *= -plane_y
    tl = d - 0.5 * du - 0.5 * dv
    # Morton-ish coherence is irrelevant for the scalar numbers; keep scanline order for the wave model
    order = np.lexsort((px.astype(int) // 8, py.astype(int) // 8))
    px, py = px[order], py[order]
    dirs = (px[:, None] / w) * du + (py[:, None] / h) * dv + tl
    dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
    org = np.tile(np.asarray(e, np.float32), (n, 1))

    def dump(fname, o, dd, tmin, tmax):
        rec = np.concatenate([o, dd, np.full((len(o), 1), tmin, np.float32), np.asarray(tmax, np.float32).reshape(-1, 1)], axis=1)
        rec.astype(np.float32).tofile(fname)
        print(fname, len(rec))

    dump(out + "_primary.bin", org, dirs, 0.0, np.full(n, 1e20))
    osc = oracle_lib.OracleScene(sc)
    hit = osc.trace(org, dirs, 0.0, 1e20, closest=True)
    ok = hit["inst"] >= 0
    p = (org + hit["t"][:, None] * dirs)[ok]
    # random hemisphere-ish directions around the reversed view direction's reflection: plain uniform sphere is
    # a fair stand-in for diffuse bounces of an interior (the normal is not needed for traversal cost)
    b = rng.normal(size=p.shape).astype(np.float32)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    p_b = (p - 1e-3 * dirs[ok]).astype(np.float32)
    dump(out + "_bounce.bin", p_b, b, 1e-4, np.full(len(p), 1e20))
    light = np.asarray(sc.lights[0].position, np.float32)[:3] if hasattr(sc.lights[0], "position") else None
    if light is None:
        lp = np.asarray(sc.lights[0], np.float32).reshape(-1)[4:7]
    else:
        lp = light
    to_l = lp[None, :] - p_b
    dist = np.linalg.norm(to_l, axis=1)
    dump(out + "_shadow.bin", p_b, (to_l / dist[:, None]).astype(np.float32), 1e-4, dist)


if __name__ == "__main__":
    main()
