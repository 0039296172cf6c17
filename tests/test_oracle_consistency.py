"""CPU checks of the oracle itself (the reference ships no tests or golden vectors):
mathematical properties every correct restatement must satisfy, the oracle's BVH against its
own brute force, and the committed golden vectors (tests/golden/, made by
tests/golden/make_golden.py from the oracle) as a regression anchor."""
import json
import os

import numpy as np
import pytest

from chameleonrt_amd import scenes
from tests import kat_inputs as K
from tests.parity import camera_of, probe_rays

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_triangle_hit_is_on_the_ray_and_on_the_triangle(oracle):
    sc = scenes.cornell()
    o = oracle.OracleScene(sc)
    org, dirs = probe_rays(sc, 5000, seed=1)
    h = o.trace(org, dirs, 0.0, 1e20, closest=True, brute_force=True)
    hit = h["inst"] >= 0
    assert hit.sum() > 1000, hit.sum()
    geoms = sc.meshes[0].geometries
    for i in np.where(hit)[0][:500]:
        g = geoms[h["geom"][i]]
        v0, v1, v2 = g.vertices[g.indices[h["prim"][i]]]
        p_bary = (1 - h["u"][i] - h["v"][i]) * v0 + h["u"][i] * v1 + h["v"][i] * v2
        p_ray = org[i] + h["t"][i] * dirs[i]
        assert np.allclose(p_bary, p_ray, atol=2e-5)
        assert 0 <= h["u"][i] <= 1 and 0 <= h["v"][i] <= 1 and h["u"][i] + h["v"][i] <= 1 + 1e-6


@pytest.mark.parametrize("name", ["cornell", "grove"])
def test_oracle_bvh_equals_brute_force(oracle, name):
    sc = scenes.cornell() if name == "cornell" else scenes.instanced_grove(n_instances=12, leaves_per_tree=60)
    o = oracle.OracleScene(sc)
    org, dirs = probe_rays(sc, 4000, seed=2)
    a = o.trace(org, dirs, 1e-4, 1e20, closest=True, brute_force=False)
    b = o.trace(org, dirs, 1e-4, 1e20, closest=True, brute_force=True)
    for k in ("inst", "geom", "prim"):
        assert np.array_equal(a[k], b[k])
    assert np.array_equal(a["t"].view(np.uint32), b["t"].view(np.uint32))
    sa = o.trace(org, dirs, 1e-4, 6.0, closest=False, brute_force=False)
    sb = o.trace(org, dirs, 1e-4, 6.0, closest=False, brute_force=True)
    assert np.array_equal(sa["t"], sb["t"])


def test_sample_returns_its_own_pdf_and_value(oracle):
    """sample_disney_brdf returns (disney_brdf, disney_pdf) evaluated at the direction it chose
    (disney_bsdf.ih:427-428)."""
    rec = K.disney_sample_records(4000)
    s = oracle.kat(K.KAT_DISNEY_SAMPLE, rec, 8)
    ok = (s[:, 6] > 0) & np.isfinite(s).all(axis=1)
    ev = np.concatenate([rec[:, :20], s[:, 3:6], rec[:, 20:26]], axis=1)
    e = oracle.kat(K.KAT_DISNEY_EVAL, ev, 4)
    assert ok.sum() > 2000
    assert np.array_equal(e[ok][:, :3], s[ok][:, :3]) and np.array_equal(e[ok][:, 3], s[ok][:, 6])


def test_lambert_material_integrates_to_albedo(oracle):
    """White-furnace style check: for the default diffuse material E[f * cos / pdf] over the
    sampler is close to the albedo times the (roughness-1) Disney retro-reflection factor."""
    n = 40000
    rng = np.random.default_rng(5)
    rec = K.disney_sample_records(n, seed=21)
    rec[:, 0:14] = [0.8, 0.8, 0.8, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1.5, 0]
    nrm = rec[:, 14:17]
    rec[:, 17:20] = nrm  # view along the normal
    s = oracle.kat(K.KAT_DISNEY_SAMPLE, rec, 8)
    ok = s[:, 6] > 0
    cos = np.abs(np.einsum("ij,ij->i", s[:, 3:6], nrm))
    est = np.where(ok, s[:, 0] * cos / np.where(ok, s[:, 6], 1), 0).mean()
    assert 0.6 < est < 0.95


def test_frame_is_deterministic_and_threads_do_not_matter(oracle):
    sc = scenes.cornell(spp=2)
    e, d, u, fovy = camera_of(sc)
    imgs = []
    for nt in (1, 4):
        r = oracle.OracleRenderer(sc, 96, 80, nt)
        r.render(e, d, u, fovy, True)
        r.render(e, d, u, fovy, False)
        imgs.append((r.accum(), r.ray_counts(), r.framebuffer()))
    for a, b in zip(imgs[0], imgs[1]):
        assert np.array_equal(a, b)


def test_golden_vectors(oracle):
    """Regression anchor: oracle outputs recorded in tests/golden/*.npz by make_golden.py.
    Integer outputs must be identical; float outputs are allowed 2e-6 relative so the fixture
    survives a libm update."""
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        manifest = json.load(f)
    for entry in manifest["kats"]:
        z = np.load(os.path.join(GOLDEN, entry["file"]))
        out = oracle.kat(entry["fn"], z["input"], z["output"].shape[1])
        if entry["exact"]:
            assert np.array_equal(out.view(np.uint32), z["output"].view(np.uint32)), entry["file"]
        else:
            assert np.allclose(out, z["output"], rtol=2e-6, atol=1e-7, equal_nan=True), entry["file"]
    z = np.load(os.path.join(GOLDEN, manifest["frame"]["file"]))
    sc = scenes.cornell(spp=manifest["frame"]["spp"])
    e, d, u, fovy = camera_of(sc)
    r = oracle.OracleRenderer(sc, manifest["frame"]["width"], manifest["frame"]["height"])
    for f in range(manifest["frame"]["frames"]):
        st = r.render(e, d, u, fovy, f == 0)
    assert np.array_equal(r.ray_counts(), z["ray_counts"])
    assert np.allclose(r.accum(), z["accum"], rtol=1e-5, atol=1e-6)
    assert int(st.rays) == int(z["ray_counts"].sum())
