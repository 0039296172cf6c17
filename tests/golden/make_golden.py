"""Generate tests/golden/*.npz from the CPU oracle (the reference ships no golden vectors and
cannot be built here, so these are regression anchors for the restatement, not reference data).

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from chameleonrt_amd import scenes  # noqa: E402
from tests import kat_inputs as K  # noqa: E402
from tests import oracle_lib  # noqa: E402
from tests.parity import camera_of  # noqa: E402


def main():
    manifest = {"kats": []}
    sets = [("disney_eval", K.KAT_DISNEY_EVAL, K.disney_eval_records(256, seed=101), False),
            ("disney_sample", K.KAT_DISNEY_SAMPLE, K.disney_sample_records(256, seed=102), False),
            ("light", K.KAT_LIGHT, K.light_records(128, seed=103), True),
            ("miss", K.KAT_MISS, K.dir_records(128, seed=104), False),
            ("ortho", K.KAT_ORTHO_BASIS, K.dir_records(128, seed=105), True),
            ("rng", K.KAT_RNG, K.rng_records(), True)]
    for name, fn, rec, exact in sets:
        out = oracle_lib.kat(fn, rec, K.N_OUT[fn])
        fname = f"kat_{name}.npz"
        np.savez_compressed(os.path.join(HERE, fname), input=rec, output=out)
        manifest["kats"].append({"file": fname, "fn": fn, "exact": exact})
    w, h, spp, frames = 64, 48, 2, 2
    sc = scenes.cornell(spp=spp)
    e, d, u, fovy = camera_of(sc)
    r = oracle_lib.OracleRenderer(sc, w, h)
    for f in range(frames):
        r.render(e, d, u, fovy, f == 0)
    np.savez_compressed(os.path.join(HERE, "frame_cornell.npz"), accum=r.accum(), ray_counts=r.ray_counts(),
                        rgba8=r.framebuffer())
    manifest["frame"] = {"file": "frame_cornell.npz", "width": w, "height": h, "spp": spp, "frames": frames}
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)


if __name__ == "__main__":
    main()
