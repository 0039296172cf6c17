import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chameleonrt_amd import scenes
from chameleonrt_amd.render_hip import RenderHIP
from tests import kat_inputs as K
from tests.oracle_lib import OracleScene
sc = scenes.instanced_grove()
r = RenderHIP(); r.initialize(64, 64); r.set_scene(sc); o = OracleScene(sc)
for name, fn, rec in (("eval", 1, K.disney_eval_records(20000)), ("sample", 2, K.disney_sample_records(20000)),
                      ("light", 3, K.light_records(5000)), ("tex", 4, K.texture_records(20000, len(sc.textures))),
                      ("miss", 5, K.dir_records(20000)), ("unpack", 9, K.unpack_records(5000, len(sc.materials)))):
    g, c = r.kat(fn, rec, K.N_OUT[fn]), o.kat(fn, rec, K.N_OUT[fn])
    gn, cn = np.isnan(g).any(axis=1), np.isnan(c).any(axis=1)
    d = np.abs(g - c) / (1e-6 + np.abs(c))
    print(name, "gpu nan rows", gn.sum(), "cpu nan rows", cn.sum(), "nan mismatch", (gn != cn).sum(), "max rel", np.nanmax(d), "bitexact rows", (g.view(np.uint32) == c.view(np.uint32)).all(axis=1).mean())
    bad = np.where(gn != cn)[0][:3]
    for i in bad:
        print("  rec", rec[i].tolist()); print("  gpu", g[i].tolist()); print("  cpu", c[i].tolist())
