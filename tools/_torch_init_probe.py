import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
mode = sys.argv[1]
import torch
if mode == "init":
    torch.cuda.set_device(0); s = torch.cuda.Stream(); torch.cuda.synchronize()
from chameleonrt_amd import scenes, core
from chameleonrt_amd.render_hip import RenderHIP
from tests.parity import camera_of
sc, w, h, spp = scenes.make_workload("C2")
r = RenderHIP(flags=core.FLAG_TIMING); r.initialize(w, h); r.set_scene(sc)
e, d, u, f = camera_of(sc)
for k in range(3): r.render(e, d, u, f, k == 0, False)
t = time.perf_counter(); ks = 0
for k in range(16):
    st = r.render(e, d, u, f, False, False); ks += st.closest_ms + st.shadow_ms + st.shade_ms
el = (time.perf_counter() - t) / 16 * 1e3
print(mode, "ms/step", round(el, 3), "kernel sum", round(ks / 16, 3))
