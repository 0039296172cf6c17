// lbvh.h — the pieces of the linear-BVH build that are the same on the host and on the device.
//
// SURVEY 8f-1 (the step before the hot path: rtcCommitScene, reference embree_utils.cpp:63-76,121-129):
// the SAH builder of bvh_builder.cpp runs on the host cores; this is the MI355X-side alternative --
// Morton codes of the triangle centroids, one radix sort, the binary radix tree of Karras 2012
// ("Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees"), a bottom-up box
// pass and a level-by-level collapse to the 4-wide quantised nodes the traversal kernels read.
// Every function here is __host__ __device__: bvh_device.hip calls them from its kernels, and
// build_lbvh_host() (bvh_builder.cpp) runs the same code serially so the CPU tests can check the
// algorithm (tests/test_prepared_scene.py with CRT_BVH_BUILDER=lbvh) where there is no GPU.
#pragma once
#include <stdint.h>

#include "bvh_builder.h"
#include "crt_types.h"

#if defined(__HIPCC__)
#define CRT_HD __host__ __device__ inline
#else
#define CRT_HD inline
#endif

namespace crt {

CRT_HD uint64_t lbvh_spread21(uint32_t x) // 21 bits -> every third bit of 63
{
    uint64_t v = x & 0x1fffffu;
    v = (v | (v << 32)) & 0x1f00000000ffffull;
    v = (v | (v << 16)) & 0x1f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}
// 63-bit Morton code of a point given in [0, 1]^3
CRT_HD uint64_t lbvh_morton63(float x, float y, float z)
{
    auto q = [](float f) -> uint32_t {
        const float s = f * 2097152.f; // 2^21
        return s <= 0.f ? 0u : (s >= 2097151.f ? 2097151u : (uint32_t)s);
    };
    return (lbvh_spread21(q(x)) << 2) | (lbvh_spread21(q(y)) << 1) | lbvh_spread21(q(z));
}

// Sort key of an item: Morton code of its box centre. Two normalisations, neither better everywhere:
//   mode 0  each axis scaled by its own extent -- the cells have the proportions of the scene, which suits
//           architecture whose structure repeats at binary fractions of its bounds (+9 % node visits over SAH
//           on the Sponza-like atrium, +31 % with mode 1);
//   mode 1  one scale, the largest extent, for all three axes -- the cells are cubes in world space, so a
//           flat scene is never cut along its thin axis near the root (mode 0 doubled the node visits of
//           primary rays on the Rungholt-like city).
// The builders make the binary tree both ways and keep the one with the smaller summed surface area of
// its internal nodes (the SAH estimate of node visits).
constexpr int LBVH_KEY_MODES = 2;
CRT_HD uint64_t lbvh_key(const Aabb &box, const Aabb &bounds, int mode)
{
    float ext[3], emax = 0.f;
    for (int a = 0; a < 3; ++a) {
        ext[a] = bounds.hi[a] - bounds.lo[a];
        emax = ext[a] > emax ? ext[a] : emax;
    }
    float c[3];
    for (int a = 0; a < 3; ++a) {
        const float e = mode == 0 ? ext[a] : emax;
        c[a] = e > 0.f ? (0.5f * (box.lo[a] + box.hi[a]) - bounds.lo[a]) / e : 0.f;
    }
    return lbvh_morton63(c[0], c[1], c[2]);
}

CRT_HD int lbvh_clz64(uint64_t x)
{
    return x == 0 ? 64 : __builtin_clzll(x);
}

// Length of the common prefix of keys i and j (sorted keys; duplicates are told apart by their
// position), -1 if j is outside [0, n).
CRT_HD int lbvh_delta(const uint64_t *keys, int n, int i, int j)
{
    if (j < 0 || j >= n) {
        return -1;
    }
    const uint64_t a = keys[i], b = keys[j];
    if (a != b) {
        return lbvh_clz64(a ^ b);
    }
    return 64 + lbvh_clz64((uint64_t)(uint32_t)i ^ (uint64_t)(uint32_t)j);
}

// Internal node i of the binary radix tree over n sorted keys (n >= 2, 0 <= i <= n - 2): the
// range [lo, hi] of keys it covers and its two children. A child reference c is an internal
// node index if c >= 0 and the sorted position ~c of a single key otherwise. Node 0 is the root.
CRT_HD void lbvh_node(const uint64_t *keys, int n, int i, int32_t &left, int32_t &right, int &lo, int &hi)
{
    const int d = lbvh_delta(keys, n, i, i + 1) - lbvh_delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int delta_min = lbvh_delta(keys, n, i, i - d);
    int lmax = 2;
    while (lbvh_delta(keys, n, i, i + lmax * d) > delta_min) {
        lmax *= 2;
    }
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2) {
        if (lbvh_delta(keys, n, i, i + (l + t) * d) > delta_min) {
            l += t;
        }
    }
    const int j = i + l * d;
    const int delta_node = lbvh_delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (lbvh_delta(keys, n, i, i + (s + t) * d) > delta_node) {
            s += t;
        }
        if (t == 1) {
            break;
        }
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    lo = i < j ? i : j;
    hi = i < j ? j : i;
    left = lo == gamma ? ~(int32_t)gamma : (int32_t)gamma;
    right = hi == gamma + 1 ? ~(int32_t)(gamma + 1) : (int32_t)(gamma + 1);
}

CRT_HD float lbvh_half_area(const Aabb &b)
{
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// ---- PLOC: parallel locally-ordered clustering (Meister & Bittner, "Parallel Locally-Ordered Clustering for Bounding Volume
// Hierarchy Construction", IEEE TVCG 2018) -- the other way to turn the Morton-sorted items into a binary tree (round 6). The
// Karras tree splits where the Morton codes say; PLOC builds BOTTOM-UP by surface area: the clusters -- at first the items in
// Morton order -- each look for the neighbour, among the `radius` clusters on either side of them in the array, whose union with
// them has the smallest surface area; two clusters that choose EACH OTHER merge into a node that takes the place of the first,
// the array is compacted, and so on until one cluster is left. Everything an iteration decides is a function of the cluster
// array alone, so the serial host twin (bvh_builder.cpp build_ploc_host) and the kernels (bvh_device.hip) build the SAME tree.
constexpr uint32_t PLOC_DEFAULT_RADIUS = 16; // (CRT_PLOC_RADIUS in the environment overrides it: bvh_builder.cpp ploc_radius())
CRT_HD float ploc_union_half_area(const Aabb &a, const Aabb &b)
{
    const float lx = a.lo[0] < b.lo[0] ? a.lo[0] : b.lo[0], ly = a.lo[1] < b.lo[1] ? a.lo[1] : b.lo[1], lz = a.lo[2] < b.lo[2] ? a.lo[2] : b.lo[2];
    const float hx = a.hi[0] > b.hi[0] ? a.hi[0] : b.hi[0], hy = a.hi[1] > b.hi[1] ? a.hi[1] : b.hi[1], hz = a.hi[2] > b.hi[2] ? a.hi[2] : b.hi[2];
    const float dx = hx - lx, dy = hy - ly, dz = hz - lz;
    return dx * dy + dy * dz + dz * dx;
}
CRT_HD Aabb ploc_union(const Aabb &a, const Aabb &b)
{
    Aabb u;
    for (int k = 0; k < 3; ++k) {
        u.lo[k] = a.lo[k] < b.lo[k] ? a.lo[k] : b.lo[k];
        u.hi[k] = a.hi[k] > b.hi[k] ? a.hi[k] : b.hi[k];
    }
    return u;
}
// the neighbour cluster i chooses among [i - radius, i + radius] of m clusters: smallest union area, the lower index on a tie
CRT_HD uint32_t ploc_nearest(const Aabb *cbox, uint32_t m, uint32_t i, uint32_t radius)
{
    const uint32_t lo = i > radius ? i - radius : 0u, hi = i + radius < m - 1u ? i + radius : m - 1u;
    const Aabb me = cbox[i];
    uint32_t best = i;
    float best_area = 3.4e38f;
    for (uint32_t j = lo; j <= hi; ++j) {
        if (j != i) {
            const float a = ploc_union_half_area(me, cbox[j]);
            if (a < best_area) {
                best_area = a;
                best = j;
            }
        }
    }
    return best;
}

// The binary tree as the collapse sees it.
struct LbvhTree {
    const int32_t *left, *right; // per internal node
    const int32_t *lo, *hi;      // per internal node: range of sorted positions it covers
    const Aabb *ibox;            // per internal node
    const Aabb *pbox;            // per sorted position
};
CRT_HD uint32_t lbvh_count(const LbvhTree &t, int32_t sub) { return sub >= 0 ? (uint32_t)(t.hi[sub] - t.lo[sub] + 1) : 1u; }
CRT_HD const Aabb &lbvh_box(const LbvhTree &t, int32_t sub) { return sub >= 0 ? t.ibox[sub] : t.pbox[~sub]; }

// The up to four subtrees that become the children of the wide node rooted at binary node k:
// k's two children, then repeatedly the child with the largest surface area that is not yet a
// leaf of the wide tree (more than max_leaf items) is replaced by ITS two children.
CRT_HD int lbvh_wide_children(const LbvhTree &t, int32_t k, uint32_t max_leaf, int32_t sub[BVH_WIDTH])
{
    int n = 2;
    sub[0] = t.left[k];
    sub[1] = t.right[k];
    while (n < BVH_WIDTH) {
        int best = -1;
        float best_area = -1.f;
        for (int c = 0; c < n; ++c) {
            if (sub[c] >= 0 && lbvh_count(t, sub[c]) > max_leaf) {
                const float a = lbvh_half_area(t.ibox[sub[c]]);
                if (a > best_area) {
                    best_area = a;
                    best = c;
                }
            }
        }
        if (best < 0) {
            break;
        }
        const int32_t b = sub[best];
        sub[best] = t.left[b];
        sub[n++] = t.right[b];
    }
    return n;
}

// One child quarter of a quantised node: the outward-rounded 16-bit box (bvh_builder.cpp quantise():
// lo one quantum further down than floor(), hi one further up than ceil()) -- same arithmetic, so
// a host-built and a device-built node of the same tree are the same bytes.
CRT_HD void lbvh_quantise_child(QChild &q, const Aabb &b, int32_t ref, const QFrame &f)
{
    for (int a = 0; a < 3; ++a) {
        double x = __builtin_floor(((double)b.lo[a] - (double)f.base[a]) / (double)f.step[a]) - 1.0;
        q.q[a][0] = (uint16_t)(x < 0.0 ? 0.0 : (x > 65535.0 ? 65535.0 : x));
        x = __builtin_ceil(((double)b.hi[a] - (double)f.base[a]) / (double)f.step[a]) + 1.0;
        q.q[a][1] = (uint16_t)(x < 0.0 ? 0.0 : (x > 65535.0 ? 65535.0 : x));
    }
    q.ref = ref;
}
CRT_HD void lbvh_unused_child(QChild &q, int32_t slot0_ref) // inverted box + a copy of slot 0's reference (crt_types.h)
{
    for (int a = 0; a < 3; ++a) {
        q.q[a][0] = 65535;
        q.q[a][1] = 0;
    }
    q.ref = slot0_ref;
}

// leaf reference of the traversal kernels: x = ~ref, first = x >> 3, count = (x & 7) + 1
CRT_HD int32_t lbvh_leaf_ref(uint32_t first, uint32_t count) { return (int32_t)~((first << 3) | (count - 1u)); }

} // namespace crt
