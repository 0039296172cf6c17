"""First-light diagnostic: HIP core vs CPU oracle on small scenes (run through gpurun)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chameleonrt_amd import scenes, core
from chameleonrt_amd.camera import look_at
from chameleonrt_amd.render_hip import RenderHIP
from tests.oracle_lib import OracleRenderer, OracleScene

def rays_for(scene, n, seed=0):
    rng = np.random.default_rng(seed)
    cam = scene.cameras[0]
    eye, d, u = look_at(cam.position, cam.center, cam.up)
    org = np.tile(eye, (n, 1)).astype(np.float32)
    dirs = rng.normal(size=(n, 3)).astype(np.float32)
    dirs[: n // 2] = d + 0.6 * rng.normal(size=(n // 2, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    return org, dirs.astype(np.float32)

for name, sc, (w, h) in (("cornell", scenes.cornell(spp=2), (256, 256)),
                         ("grove", scenes.instanced_grove(), (320, 200))):
    print("==", name, sc.total_tris(), "tris")
    r = RenderHIP(flags=core.FLAG_TIMING | core.FLAG_COUNTERS)
    print(r.name())
    r.initialize(w, h); r.set_scene(sc)
    osc = OracleScene(sc)
    org, dirs = rays_for(sc, 20000)
    g = r.trace(org, dirs, 0.0, 1e20, closest=True)
    o = osc.trace(org, dirs, 0.0, 1e20, closest=True, brute_force=True)
    for k in ("inst", "geom", "prim"):
        print(k, "mismatch", int((g[k] != o[k]).sum()))
    hit = o["inst"] >= 0
    print("hits", int(hit.sum()), "t exact", int((g["t"][hit] != o["t"][hit]).sum()), "u", int((g["u"][hit] != o["u"][hit]).sum()))
    ga = r.trace(org, dirs, 1e-4, 5.0, closest=False); oa = osc.trace(org, dirs, 1e-4, 5.0, closest=False, brute_force=True)
    print("anyhit mismatch", int((ga["t"] != oa["t"]).sum()), "nodes/ray", g["stats"].closest_nodes / len(org), "tris/ray", g["stats"].closest_tris / len(org))
    cam = sc.cameras[0]; e, d, u = look_at(cam.position, cam.center, cam.up)
    orr = OracleRenderer(sc, w, h)
    for f in range(3):
        st = r.render(e, d, u, cam.fov_y, f == 0, True)
        ost = orr.render(e, d, u, cam.fov_y, f == 0)
        a, b = r.accum(), orr.accum()
        err = np.abs(a - b); tol = 1e-4 + 1e-3 * np.abs(b)
        print(f"frame {f}: gpu {st.render_time_ms:.2f} ms {st.rays} rays ({st.rays_per_second/1e6:.1f} MRay/s) oracle {ost.rays} rays;"
              f" bad px {(err > tol).any(axis=2).mean():.5f} max {err.max():.3g} mean rel {err.mean()/b.mean():.3g} nan {np.isnan(a).sum()}"
              f" closest_ms {st.closest_ms:.3f} shadow_ms {st.shadow_ms:.3f} shade_ms {st.shade_ms:.3f}")
        print("   raycount mismatch px", int((r.ray_counts() != orr.ray_counts()).sum()), "img diff>1", int((np.abs(r.img.view(np.uint8).astype(int) - orr.framebuffer().view(np.uint8).astype(int)) > 1).sum()))
    r.close()
print("done")
