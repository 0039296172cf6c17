"""GPU: the wave-packet traversal kernels (chameleonrt_amd/csrc/packet.h) -- one traversal stack per wave, node and leaf
slot fetched once per wave by the scalar unit, every lane testing its own ray -- against brute force and against the
per-lane kernels.

The closest hit is the lexicographic minimum of (t, inst, geom, prim) over all valid hits and an occlusion query asks
whether any exists, so WHICH nodes a wave visits on behalf of its other lanes cannot change a lane's answer: explicit
rays through the packet kernels equal brute force bit for bit -- also incoherent probe rays, the worst case for a packet,
which a frame never hands them -- and frames rendered with packets for bounce 0 (the default), for every bounce, and for
none are identical to the last bit (accumulated radiance, REPORT_RAY_STATS counts, RGBA8).
"""
import os

import numpy as np
import pytest

from chameleonrt_amd import scenes
from chameleonrt_amd.render_hip import RenderHIP
from tests.parity import awkward_instances, camera_of, probe_rays

pytestmark = pytest.mark.gpu

SCENES = {
    "cornell": (lambda: scenes.cornell(spp=2), 128, 96),                          # one identity instance
    "sponza_small": (lambda: scenes.sponza_like(spp=2, detail=0.02, tex_size=32), 160, 96),
    "grove_world_tree": (lambda: scenes.instanced_grove(), 160, 100),             # several instances: a world tree
    "awkward_instances": (awkward_instances, 96, 64),                             # mirrored / coincident instances: exact ties
}


def _renderer(sc, w, h, packet_bounces, monkeypatch):
    monkeypatch.setenv("CRT_HIP_PACKET_BOUNCES", str(packet_bounces))
    r = RenderHIP()
    r.initialize(w, h)
    r.set_scene(sc)
    return r


@pytest.mark.parametrize("name", list(SCENES))
def test_packet_kernels_match_brute_force(name, oracle, hip_lib, monkeypatch):
    gen, w, h = SCENES[name]
    sc = gen()
    o = oracle.OracleScene(sc)
    org, dirs = probe_rays(sc, 20000, seed=51)
    c = o.trace(org, dirs, 0.0, 1e20, closest=True, brute_force=True)
    hit = c["inst"] >= 0
    p = org[hit] + c["t"][hit, None] * dirs[hit]
    d2 = np.random.default_rng(52).normal(size=p.shape).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    c2 = o.trace(p, d2, 1e-4, 1e20, closest=True, brute_force=True)
    tmax = np.random.default_rng(53).random(len(p)).astype(np.float32) * 10
    c3 = o.trace(p, d2, 1e-4, tmax, closest=False, brute_force=True)
    for bounces in (5, 0):  # every production launch through the packet kernels; none (the per-lane production kernels)
        r = _renderer(sc, w, h, bounces, monkeypatch)
        try:
            for g, ref in ((r.trace(org, dirs, 0.0, 1e20, closest=True, production=True), c),
                           (r.trace(p, d2, 1e-4, 1e20, closest=True, production=True), c2)):
                for k in ("inst", "geom", "prim"):
                    assert np.array_equal(g[k], ref[k]), (bounces, k)
                hh = ref["inst"] >= 0
                for k in ("t", "u", "v"):
                    assert np.array_equal(g[k][hh].view(np.uint32), ref[k][hh].view(np.uint32)), (bounces, k)
            assert np.array_equal(r.trace(p, d2, 1e-4, tmax, closest=False, production=True)["t"], c3["t"]), bounces
        finally:
            r.close()


@pytest.mark.parametrize("name", list(SCENES))
def test_frames_with_and_without_packets_are_bit_identical(name, hip_lib, monkeypatch):
    gen, w, h = SCENES[name]
    sc = gen()
    e, d, u, fovy = camera_of(sc)
    frames = {}
    for bounces in (0, 1, 5):
        r = _renderer(sc, w, h, bounces, monkeypatch)
        for f in range(2):
            st = r.render(e, d, u, fovy, f == 0, True)
        frames[bounces] = (r.accum().copy(), r.ray_counts().copy(), r.img.copy(), int(st.rays))
        r.close()
    for bounces in (1, 5):
        a, b = frames[0], frames[bounces]
        assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), f"accumulated radiance, packets for {bounces} bounce(s)"
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3]
    assert frames[0][3] > 0 and np.isfinite(frames[0][0]).any()
