"""Seeded input records for the shading-function KATs (layouts: include/crt_kat.h)."""
import numpy as np

from chameleonrt_amd.scene import ortho_basis, obj_default_light

KAT_DISNEY_EVAL, KAT_DISNEY_SAMPLE, KAT_LIGHT, KAT_TEXTURE, KAT_MISS = 1, 2, 3, 4, 5
KAT_ORTHO_BASIS, KAT_SRGB8, KAT_RNG, KAT_UNPACK_MATERIAL, KAT_NEE, KAT_ROULETTE = 6, 7, 8, 9, 10, 11
N_OUT = {1: 4, 2: 8, 3: 9, 4: 5, 5: 3, 6: 6, 7: 1, 8: 17, 9: 14, 10: 17}


def _unit(v):
    return (v / np.linalg.norm(v, axis=-1, keepdims=True)).astype(np.float32)


def random_materials(rng, n):
    """14-float MaterialParams rows covering every Disney lobe."""
    m = np.zeros((n, 14), np.float32)
    m[:, 0:3] = rng.random((n, 3))
    m[:, 3] = rng.random(n) * (rng.random(n) < 0.5)          # metallic
    m[:, 4] = rng.random(n)                                   # specular
    m[:, 5] = np.clip(rng.random(n) * 1.1 - 0.05, 0, 1)       # roughness incl. 0 and 1
    m[:, 6] = rng.random(n)                                   # specular_tint
    m[:, 7] = rng.random(n) * (rng.random(n) < 0.4)           # anisotropy (0 for most)
    m[:, 8] = rng.random(n)                                   # sheen
    m[:, 9] = rng.random(n)                                   # sheen_tint
    m[:, 10] = rng.random(n) * (rng.random(n) < 0.5)          # clearcoat
    m[:, 11] = rng.random(n)                                  # clearcoat_gloss
    m[:, 12] = 1.0 + rng.random(n)                            # ior
    m[:, 13] = rng.random(n) * (rng.random(n) < 0.3)          # specular_transmission
    m[: n // 50, 0:3] = 0.0                                   # black base colour (lum == 0 branch)
    return m


def shading_frames(rng, n):
    nrm = _unit(rng.normal(size=(n, 3)))
    vx = np.zeros((n, 3), np.float32)
    vy = np.zeros((n, 3), np.float32)
    for i in range(n):
        vx[i], vy[i] = ortho_basis(nrm[i])
    return nrm, vx, vy


def disney_eval_records(n, seed=11):
    rng = np.random.default_rng(seed)
    mat = random_materials(rng, n)
    nrm, vx, vy = shading_frames(rng, n)
    w_o = _unit(rng.normal(size=(n, 3)))
    w_i = _unit(rng.normal(size=(n, 3)))
    # most w_o on the upper side, as after the reference's normal flip
    flip = (np.einsum("ij,ij->i", w_o, nrm) < 0) & (rng.random(n) < 0.8)
    w_o[flip] *= -1
    return np.concatenate([mat, nrm, w_o, w_i, vx, vy], axis=1).astype(np.float32)


def disney_sample_records(n, seed=12):
    rng = np.random.default_rng(seed)
    mat = random_materials(rng, n)
    nrm, vx, vy = shading_frames(rng, n)
    w_o = _unit(rng.normal(size=(n, 3)))
    flip = (np.einsum("ij,ij->i", w_o, nrm) < 0) & (rng.random(n) < 0.8)
    w_o[flip] *= -1
    state = rng.integers(0, 2**32, size=(n, 1), dtype=np.uint64).astype(np.uint32).view(np.float32)
    return np.concatenate([mat, nrm, w_o, vx, vy, state], axis=1).astype(np.float32)


def nee_records(n, light, seed=19):
    """Surface points around the scene's quad light (20 floats: emission, position, normal, v_x, v_y, each padded to 4; the
    half-extents ride in v_x.w / v_y.w) at distances from a fraction of its size to 30 times it, most of them facing it --
    so that light samples have useful pdfs and a share of the BSDF samples lands on the quad (the second ray)."""
    rng = np.random.default_rng(seed)
    mat = random_materials(rng, n)
    mat[: n // 4, 5] = np.clip(mat[: n // 4, 5], 0.5, 1.0)  # a share of rough surfaces: wide lobes find the light
    light = np.asarray(light, np.float32).reshape(-1)[:20]
    centre, l_n = light[4:7], light[8:11]
    size = max(float(abs(light[15])), float(abs(light[19])), 1e-3)
    d = _unit(rng.normal(size=(n, 3)))
    front = np.einsum("ij,j->i", d, l_n) < 0
    d[front & (rng.random(n) < 0.9)] *= -1  # most points on the emitting side
    dist = (size * np.exp(rng.uniform(np.log(0.3), np.log(30.0), size=n))).astype(np.float32)
    hit_p = (centre[None, :] + d * dist[:, None]).astype(np.float32)
    to_l = _unit(centre[None, :] - hit_p)
    nrm = _unit(to_l + rng.normal(size=(n, 3)).astype(np.float32) * rng.random((n, 1)).astype(np.float32) * 1.5)
    vx = np.zeros((n, 3), np.float32)
    vy = np.zeros((n, 3), np.float32)
    for i in range(n):
        vx[i], vy[i] = ortho_basis(nrm[i])
    w_o = _unit(nrm + rng.normal(size=(n, 3)).astype(np.float32))
    state = rng.integers(0, 2**32, size=(n, 1), dtype=np.uint64).astype(np.uint32).view(np.float32)
    return np.concatenate([mat, nrm, w_o, vx, vy, hit_p, state], axis=1).astype(np.float32)


def light_records(n, seed=13):
    rng = np.random.default_rng(seed)
    light = np.tile(obj_default_light(), (n, 1))
    orig = (rng.normal(size=(n, 3)) * 3).astype(np.float32)
    target = light[:, 4:7] + (rng.normal(size=(n, 3)) * 4).astype(np.float32)
    d = _unit(target - orig)
    s = rng.random((n, 2)).astype(np.float32)
    return np.concatenate([light, orig, d, s], axis=1).astype(np.float32)


def texture_records(n, n_tex, seed=14):
    rng = np.random.default_rng(seed)
    tid = rng.integers(0, n_tex, size=(n, 1)).astype(np.uint32).view(np.float32)
    uv = (rng.random((n, 2)) * 5 - 2).astype(np.float32)  # [-2, 3]: wrap + negative coords (quirk Q12)
    uv[: n // 20] = np.round(uv[: n // 20] * 8) / 8       # texel-boundary coordinates
    ch = rng.integers(0, 3, size=(n, 1)).astype(np.uint32).view(np.float32)
    return np.concatenate([tid, uv, ch], axis=1).astype(np.float32)


def dir_records(n, seed=15):
    rng = np.random.default_rng(seed)
    d = _unit(rng.normal(size=(n, 3)))
    d[:6] = np.array([[0, 1, 0], [0, -1, 0], [1, 0, 0], [0, 0, 1], [0, 0, -1], [-1, 0, 0]], np.float32)
    return d.astype(np.float32)


def srgb_records(n, seed=16):
    rng = np.random.default_rng(seed)
    x = np.concatenate([rng.random(n - 8) ** 3 * 1.5, [0, 0.0031308, 0.0031309, 1.0, 2.0, -0.5, 1e-8, 0.5]])
    return x.reshape(-1, 1).astype(np.float32)


def rng_records():
    pix = np.array([0, 1, 262143, 12345, 1920 * 1080 - 1, 0xFFFFFFFF, 77, 3840 * 2160 - 1], np.uint32)
    frm = np.array([1, 1, 1, 17, 5, 0xFFFFFFFF, 4 * 9 + 1 + 3, 64 * 1000 + 64], np.uint32)
    return np.stack([pix.view(np.float32), frm.view(np.float32)], axis=1)


def unpack_records(n, n_mat, seed=17):
    rng = np.random.default_rng(seed)
    mid = rng.integers(0, n_mat, size=(n, 1)).astype(np.uint32).view(np.float32)
    uv = (rng.random((n, 2)) * 5 - 2).astype(np.float32)
    return np.concatenate([mid, uv], axis=1).astype(np.float32)


def roulette_records(n, seed=23):
    """Throughputs for render_embree.ispc:327-335 over every regime of q = max(0.05, 1 - max(tp)): dim paths (q near 1), bright ones
    (max(tp) >= 1 -> q = 0.05 exactly), throughputs around the 0.95 switch-over, zeros, negatives, and the non-finite values glass
    produces (inf, NaN in any subset of the channels -- the reference's max is a select, `a < b ? b : a`, which treats a NaN
    differently by operand position), each with a random RNG state."""
    rng = np.random.default_rng(seed)
    tp = np.empty((n, 3), np.float32)
    kind = rng.integers(0, 8, n)
    tp[:] = rng.random((n, 3)).astype(np.float32)                                   # 0: ordinary
    m = kind == 1; tp[m] *= np.float32(1e-3)                                        # dim
    m = kind == 2; tp[m] = (rng.random((m.sum(), 3)) * 4).astype(np.float32)        # bright: often >= 1
    m = kind == 3; tp[m] = (0.95 + (rng.random((m.sum(), 3)) - 0.5) * 1e-6).astype(np.float32)  # at the q = 0.05 switch-over
    m = kind == 4; tp[m] = 0.0
    m = kind == 5; tp[m] = -tp[m]
    special = np.array([np.nan, np.inf, -np.inf, 0.0, 1.0, 0.5], np.float32)
    m = kind >= 6                                                                   # non-finite values in random channels
    pick = rng.integers(0, len(special), (m.sum(), 3))
    keep = rng.random((m.sum(), 3)) < 0.4
    tp[m] = np.where(keep, tp[m], special[pick])
    rec = np.zeros((n, 4), np.float32)
    rec[:, :3] = tp
    rec[:, 3] = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    # the position-dependent NaN cases, explicitly: NaN first with finite others, and the other way round
    fixed = np.array([[np.nan, 0.5, 0.2], [0.5, np.nan, 0.2], [0.5, 0.2, np.nan], [np.nan, np.nan, 0.3], [np.inf, 0.1, 0.1],
                      [0.1, -np.inf, 0.1], [1.0, 1.0, 1.0], [0.95, 0.0, 0.0], [0.0, 0.0, 0.0]], np.float32)
    rec[:len(fixed), :3] = fixed
    return rec
