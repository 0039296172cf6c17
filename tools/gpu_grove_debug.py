import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chameleonrt_amd import scenes
from chameleonrt_amd.render_hip import RenderHIP
from tests.oracle_lib import OracleRenderer
from chameleonrt_amd.camera import camera_of
sc = scenes.instanced_grove(); w, h = 320, 200
r = RenderHIP(); r.initialize(w, h); r.set_scene(sc); o = OracleRenderer(sc, w, h)
e, d, u, fovy = camera_of(sc)
r.render(e, d, u, fovy, True, True); o.render(e, d, u, fovy, True)
a, b = r.accum(), o.accum()
nfa, nfb = ~np.isfinite(a).all(axis=2), ~np.isfinite(b).all(axis=2)
print("nonfinite gpu", nfa.sum(), "cpu", nfb.sum(), "both", (nfa & nfb).sum(), "only gpu", (nfa & ~nfb).sum(), "only cpu", (~nfa & nfb).sum())
with np.errstate(invalid="ignore"):
    err = np.abs(a - b); bad = (err > 1e-4 + 1e-3 * np.abs(b)).any(axis=2) | (nfa != nfb)
ys, xs = np.where(bad)
gc, cc = r.ray_counts(), o.ray_counts()
for y, x in list(zip(ys, xs))[:25]:
    print((x, y), "gpu", a[y, x], "cpu", b[y, x], "rays", gc[y, x], cc[y, x])
print("n bad", bad.sum(), "of", w * h)
