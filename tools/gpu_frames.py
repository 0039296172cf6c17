import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chameleonrt_amd import scenes, core
from chameleonrt_amd.camera import look_at
from chameleonrt_amd.render_hip import RenderHIP
which = sys.argv[1] if len(sys.argv) > 1 else "grove"
flags = int(sys.argv[2]) if len(sys.argv) > 2 else core.FLAG_TIMING
nframes = int(sys.argv[3]) if len(sys.argv) > 3 else 8
if which == "grove":
    sc, w, h = scenes.instanced_grove(), 320, 200
elif which == "cornell":
    sc, w, h = scenes.cornell(spp=2), 256, 256
else:
    wd = which.endswith("wd")
    if wd:
        which = which[:-2]
    over = {}
    if ":" in which:  # e.g. C4:tex_size=512,n_tex=16
        which, _, kv = which.partition(":")
        over = {k: int(v) for k, v in (x.split("=") for x in kv.split(","))}
    t = time.time(); sc, w, h, spp = scenes.make_workload(which, **over); print("scene gen", time.time() - t, "s", sc.total_tris(), "tris")
    if wd:
        sc = sc.white_diffuse()
part = os.environ.get("CRT_PART")  # "rank/world": render only that share of the tiles (predicts strong scaling)
rank, world = (int(x) for x in part.split("/")) if part else (0, 1)
r = RenderHIP(flags=flags, rank=rank, world=world); r.initialize(w, h)
t = time.time(); r.set_scene(sc); print("set_scene", time.time() - t, "s")
cam = sc.cameras[0]; e, d, u = look_at(cam.position, cam.center, cam.up)
f, swapped_at = -1, None
while f + 1 < nframes:
    f += 1
    t = time.time()
    st = r.render(e, d, u, cam.fov_y, f == 0, False)
    wall = (time.time() - t) * 1e3
    print(f"frame {f}: wall {wall:.2f} ms render_time {st.render_time_ms:.2f} ms rays {st.rays} ({st.rays_per_second/1e6:.1f} MRay/s) closest {st.closest_ms:.3f} shadow {st.shadow_ms:.3f} shade {st.shade_ms:.3f}"
          f" | nodes/ray c {st.closest_nodes/max(1,st.closest_rays):.1f} s {st.shadow_nodes/max(1,st.shadow_rays):.1f} tris/ray c {st.closest_tris/max(1,st.closest_rays):.1f} s {st.shadow_tris/max(1,st.shadow_rays):.1f}"
          + (f" | refine state {r.refine_state()}" if flags & core.FLAG_REFINE_IN_BACKGROUND else ""))
    if flags & core.FLAG_REFINE_IN_BACKGROUND:  # keep rendering until the refined tree is in use, then three frames more
        state = r.refine_state()[0]
        if state in (1, 2) and f == nframes - 1 and nframes < 4000:
            nframes += 1
        elif state == 3 and swapped_at is None:
            swapped_at = f
            nframes = max(nframes, f + 4)
a = r.accum(); print("nan px", int(np.isnan(a).any(axis=2).sum()), "mean", np.nanmean(a))
