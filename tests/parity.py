"""Shared parity helpers for the GPU tests (HIP core vs CPU oracle)."""
import numpy as np

from chameleonrt_amd.camera import camera_of, look_at  # noqa: F401  (camera_of re-exported for the tests)

# Image tolerance (SURVEY §8c / north star "within a stated float tolerance"): the two
# implementations evaluate the same expressions in the same order and differ only in libm
# transcendentals (a few ulp) and in the order the per-sample radiance sums are added, so a
# pixel must agree to  |gpu - cpu| <= 1e-4 + 1e-3 * |cpu|.  A path whose ulp-level difference
# flips a discrete decision (shadow edge, Russian roulette, checker boundary) diverges
# completely; such pixels are counted and must stay below 0.1 % of the image.
ABS_TOL, REL_TOL, MAX_DIVERGED = 1e-4, 1e-3, 1e-3


def slot_triangles(bvh):
    """Triangles per leaf slot of a product BVH (RenderHIP.bvh() / PreparedScene.bvh()): bvh["tris"] holds the 64-byte leaf
    slots -- four vertices, geomID | selectors, primID of triangle A, primID of triangle B or 0xffffffff, tag (crt_types.h)."""
    return 1 + (bvh["tris"][:, 14].view(np.uint32) != 0xffffffff).astype(np.int64)


def node_refs(nodes):
    """(n, 4) child references of the packed 4-wide nodes (crt_types.h PNode: dwords 8..11 of the 64-byte record)."""
    return nodes[:, 8:12].astype(np.uint32).view(np.int32)


def node_boxes(nodes):
    """(n, 4, 3, 2) child boxes in units of the BVH's 16-bit grid: PNode dwords 0..1 hold the per-axis origin and scale
    (code e << 1 | m: 2^e or 1.5 * 2^e), dwords 2..7 the lo / hi plane bytes (byte c = child c); an unused slot is
    inverted (lo = 255 > hi = 0)."""
    n = nodes.shape[0]
    f0, f1 = nodes[:, 0].astype(np.int64), nodes[:, 1].astype(np.int64)
    origin = np.stack([f0 & 0xffff, f0 >> 16, f1 & 0xffff], axis=1).astype(np.float64)
    code = np.stack([(f1 >> 16) & 31, (f1 >> 21) & 31, f1 >> 26], axis=1)
    scale = np.where(code & 1, 1.5, 1.0) * 2.0 ** (code >> 1)
    out = np.zeros((n, 4, 3, 2), np.float64)
    for a in range(3):
        for side in range(2):
            w = nodes[:, 2 + 2 * a + side].astype(np.int64)
            for c in range(4):
                out[:, c, a, side] = origin[:, a] + ((w >> (8 * c)) & 255) * scale[:, a]
    return out


def node_used(nodes):
    """(n, 4) which child slots are in use"""
    lo, hi = nodes[:, 2].astype(np.int64), nodes[:, 3].astype(np.int64)
    return np.stack([((lo >> (8 * c)) & 255) <= ((hi >> (8 * c)) & 255) for c in range(4)], axis=1)


def probe_rays(scene, n, seed=0, spread=0.3):
    """Half camera-cone rays, half uniformly random directions from points around the scene."""
    rng = np.random.default_rng(seed)
    e, d, u, _ = camera_of(scene)
    org = np.tile(e, (n, 1)).astype(np.float32)
    dirs = rng.normal(size=(n, 3)).astype(np.float32)
    dirs[: n // 2] = d + spread * rng.normal(size=(n // 2, 3)).astype(np.float32)
    org[n // 2:] += (rng.normal(size=(n - n // 2, 3)) * 0.5).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    return org, dirs.astype(np.float32)


def compare_images(gpu, cpu):
    """Returns (fraction of diverged pixels, mean relative error over agreeing pixels)."""
    # the reference's arithmetic produces inf/NaN radiance in a few corner cases (quirks Q2, Q6:
    # zero-length normalisations, 1/(1-1)); a pixel that is non-finite in BOTH images agrees
    nf_g, nf_c = ~np.isfinite(gpu).all(axis=2), ~np.isfinite(cpu).all(axis=2)
    both_nf = nf_g & nf_c
    with np.errstate(invalid="ignore"):
        err = np.abs(gpu - cpu)
        tol = ABS_TOL + REL_TOL * np.abs(cpu)
        bad = ((err > tol).any(axis=2) | (nf_g != nf_c)) & ~both_nf
    good = ~bad & ~both_nf
    mean_rel = float(err[good].mean() / max(1e-12, np.abs(cpu[good]).mean()))
    return float(bad.mean()), mean_rel


def awkward_instances():
    """A small scene of awkward instances: a MIRRORED one (negative determinant), a strongly non-uniform scale, two
    instances at exactly the same place (exact ties in t go to the lower instance id), an identity instance of a mesh
    that transformed instances share, and one mesh under two ParameterizedMeshes with different material tables."""
    from chameleonrt_amd.scene import Camera, Geometry, Instance, Mesh, ParameterizedMesh, Scene, disney_material, obj_default_light
    rng = np.random.default_rng(5)
    v = rng.normal(size=(300, 3)).astype(np.float32)
    idx = rng.integers(0, 300, size=(400, 3)).astype(np.uint32)
    blob = Mesh([Geometry(v, idx[:250], None), Geometry(v * np.float32(0.5), idx[250:], None)])
    ground = Mesh([Geometry(np.array([[-9, -2, -9], [9, -2, -9], [9, -2, 9], [-9, -2, 9]], np.float32),
                            np.array([[0, 1, 2], [0, 2, 3]], np.uint32), None)])

    def trs(t, s, ry=0.0):
        m = np.eye(4, dtype=np.float32)
        c, si = np.cos(ry), np.sin(ry)
        m[:3, :3] = np.array([[c, 0, si], [0, 1, 0], [-si, 0, c]], np.float32) @ np.diag(np.asarray(s, np.float32))
        m[:3, 3] = t
        return m.T.reshape(16).astype(np.float32)  # column-major

    insts = [Instance(np.eye(4, dtype=np.float32).reshape(16), 0),        # ground, identity
             Instance(np.eye(4, dtype=np.float32).reshape(16), 1),        # the blob itself, identity, mesh shared below
             Instance(trs([4, 0, 0], [-1, 1, 1], 0.3), 1),                # mirrored
             Instance(trs([-4, 0.5, 1], [0.2, 3.0, 0.7], 1.1), 2),        # non-uniform scale, the other material table
             Instance(trs([0, 0, 5], [1, 1, 1], 0.7), 1),
             Instance(trs([0, 0, 5], [1, 1, 1], 0.7), 2)]                 # coincides with the one before it
    sc = Scene(meshes=[ground, blob], parameterized_meshes=[ParameterizedMesh(0, [0]), ParameterizedMesh(1, [0, 1]), ParameterizedMesh(1, [1, 0])],
               instances=insts, materials=[disney_material(), disney_material()], lights=[obj_default_light()],
               cameras=[Camera(np.array([0, 2, 14], np.float32), np.zeros(3, np.float32), np.array([0, 1, 0], np.float32), 50.0)])
    return sc


def random_material_grove(seed):
    """(scene, width, height): the instanced grove with EVERY material replaced by random Disney parameters over their whole ranges
    -- metallic, specular, roughness down to 0, anisotropy, sheen, clearcoat, ior in [1, 2.5], specular transmission on a third of
    them (the reference's glass: negative pdfs, non-finite throughputs, NaN pixels) --, 1 ... 3 quad lights of random size,
    place and direction, an odd framebuffer size, 1 ... 3 spp. (Parameters are >= 0: a sign bit is a texture handle.)"""
    from chameleonrt_amd import scenes
    from chameleonrt_amd.scene import ortho_basis, quad_light
    rng = np.random.default_rng(4242 + seed)
    sc = scenes.instanced_grove(spp=int(rng.integers(1, 4)), n_instances=int(rng.integers(2, 10)), leaves_per_tree=int(rng.integers(10, 60)),
                                tex_size=8, seed=seed + 11)
    for m in sc.materials:
        textured = np.asarray(m[0:1], np.float32).view(np.uint32)[0] & 0x80000000
        r = rng.random(16).astype(np.float32)
        if not (textured and rng.random() < 0.5):
            m[0:3] = r[0:3]                      # base colour (half of the textured ones keep their texture handle)
        m[3] = r[3] if rng.random() < 0.7 else np.float32(rng.choice([0.0, 1.0]))     # metallic
        m[4] = r[4]                              # specular
        m[5] = r[5] if rng.random() < 0.8 else np.float32(rng.choice([0.0, 1.0]))     # roughness
        m[6] = r[6]                              # specular tint
        m[7] = r[7] if rng.random() < 0.5 else np.float32(0)                           # anisotropy
        m[8], m[9] = (r[8], r[9]) if rng.random() < 0.5 else (np.float32(0), np.float32(0))    # sheen, sheen tint
        m[10], m[11] = (r[10], r[11]) if rng.random() < 0.5 else (np.float32(0), np.float32(0))  # clearcoat, gloss
        m[12] = np.float32(1.0 + 1.5 * r[12])    # ior
        m[13] = r[13] if rng.random() < 0.33 else np.float32(0)                         # specular transmission
    lights = []
    for _ in range(int(rng.integers(1, 4))):
        n = rng.normal(size=3).astype(np.float32)
        n /= np.linalg.norm(n)
        pos = (rng.normal(size=3) * 6).astype(np.float32)
        v_x, v_y = ortho_basis(n)
        e = float(rng.uniform(2, 40))
        lights.append(quad_light([e, e, e, e], pos, n, v_x, v_y, float(rng.uniform(0.2, 8)), float(rng.uniform(0.2, 8))))
    sc.lights = lights
    return sc, int(rng.integers(33, 97)), int(rng.integers(17, 65))
