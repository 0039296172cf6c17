// slab.h — the ray / box test of the traversal kernels against a packed 4-wide node (crt_types.h PNode), written so
// that the SAME source compiles for the device (hipcc) and for the host (g++: tests/native/slab_check.cpp checks it
// against the plain min/max formulation; the oracle's BVH walker mirrors the visit rule).
//
// The ray in the fixed-point frame of a BVH (QFrame: coordinate = base + q * step) is t(q) = fma(q, qa, qb) with
// qa = step/d, qb = (base - o)/d (|d| clamped to >= 1e-18, sign kept). t is monotone in q with the sign of qa, so the
// entry plane of an axis is `lo` if d > 0 and `hi` if d < 0: instead of computing both parameters and taking min / max,
// the near and far plane words of an axis are SELECTED once per node from the sign of qa (the values are bit-identical
// to the min / max form). A box stored inverted (lo > hi) has near > far for every direction, so unused child slots
// need no separate test.
#pragma once
#include <stdint.h>

#include "crt_types.h" // PNode and its CRT_PNODE_HALF_STEPS

#if defined(__HIPCC__)
#define CRT_SLAB_FN __host__ __device__ inline
#else
#define CRT_SLAB_FN inline
#endif

namespace crt {

struct SlabRay {
    float qa[3], qb[3];  // t(q) = fma(q, qa, qb) per axis
};

// bit pattern of a float / sign test without <cmath> or device headers
CRT_SLAB_FN uint32_t slab_bits(float x) { return __builtin_bit_cast(uint32_t, x); }

// ---- the packed node (crt_types.h PNode): byte planes on a per-node origin and shift ----------------------------------
// A plane byte b of an axis stands for grid coordinate origin + b * scale (scale = 2^e or 1.5 * 2^e, a float with one
// mantissa bit), whose ray parameter fma(origin + b*scale, qa, qb) is evaluated in steps that cost one conversion and one
// fma per plane: a = fma(origin, qa, qb) and s = qa * scale once per node and axis, then t(b) = fma(b, s, a). Against the
// one-step form this rounds three times; the extra roundings are at most half an ulp of a, with |a| <= |t| + 255 |s|,
// and 255 half ulps of s: together below 2^-14 of a grid unit's parameter span on top of the half ulp of t every box
// test has -- the builders' one-unit outward rounding (crt_types.h QNode) covers it.
// The near planes of an axis are the `lo` bytes if qa >= 0 and the `hi` bytes otherwise (one select per node and axis
// instead of a rotate per child); an unused slot (lo = 255, hi = 0) is inverted for both signs.
struct SlabAxis {
    float a, s;
    uint32_t near, far; // byte c = child c
};

// scale: the five-bit code e << 1 | m of crt_types.h PNode
CRT_SLAB_FN SlabAxis slab_axis(uint32_t origin, uint32_t scale, uint32_t lo, uint32_t hi, float qa, float qb)
{
    SlabAxis x;
    x.a = __builtin_fmaf((float)origin, qa, qb);
#if !CRT_PNODE_HALF_STEPS
    x.s = __builtin_ldexpf(qa, (int)(scale >> 1)); // (the packer made even codes only)
#else
    x.s = qa * __builtin_bit_cast(float, (scale << 22) + (127u << 23));
#endif
    const bool neg = (slab_bits(qa) >> 31) != 0u;
    x.near = neg ? hi : lo;
    x.far = neg ? lo : hi;
    return x;
}

// bits [at, at + n) of w (the device's one-instruction bit-field extract: the step is bound by its instruction count)
CRT_SLAB_FN uint32_t slab_field(uint32_t w, uint32_t at, uint32_t n)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ubfe(w, at, n);
#else
    return (w >> at) & ((1u << n) - 1u);
#endif
}

struct SlabNode {
    SlabAxis x, y, z;
};

// w[0..7]: the first eight dwords of a PNode
CRT_SLAB_FN SlabNode slab_node(uint32_t f0, uint32_t f1, uint32_t lo_x, uint32_t hi_x, uint32_t lo_y, uint32_t hi_y, uint32_t lo_z,
                               uint32_t hi_z, const SlabRay &r)
{
    SlabNode n;
    n.x = slab_axis(f0 & 0xffffu, slab_field(f1, 16, 5), lo_x, hi_x, r.qa[0], r.qb[0]);
    n.y = slab_axis(f0 >> 16, slab_field(f1, 21, 5), lo_y, hi_y, r.qa[1], r.qb[1]);
    n.z = slab_axis(f1 & 0xffffu, f1 >> 26, lo_z, hi_z, r.qa[2], r.qb[2]);
    return n;
}

template <int C> CRT_SLAB_FN float slab_byte(uint32_t w) { return (float)((w >> (8 * C)) & 0xffu); }

// Entry distance of child C clamped to tmin (tn) and exit distance clamped to tmax (tf). (Plain fmas: gfx950's packed
// v_pk_fma_f32 -- one instruction for the near and the far plane of an axis -- was built and measured slower, 61.8 against
// 60.3 ms on C4, profiles/r04_issue_bound_ab.txt.)
template <int C> CRT_SLAB_FN void slab_packed_span(const SlabNode &n, float tmin, float tmax, float &tn, float &tf)
{
    const float nx = __builtin_fmaf(slab_byte<C>(n.x.near), n.x.s, n.x.a), fx = __builtin_fmaf(slab_byte<C>(n.x.far), n.x.s, n.x.a);
    const float ny = __builtin_fmaf(slab_byte<C>(n.y.near), n.y.s, n.y.a), fy = __builtin_fmaf(slab_byte<C>(n.y.far), n.y.s, n.y.a);
    const float nz = __builtin_fmaf(slab_byte<C>(n.z.near), n.z.s, n.z.a), fz = __builtin_fmaf(slab_byte<C>(n.z.far), n.z.s, n.z.a);
    tn = __builtin_fmaxf(__builtin_fmaxf(nx, ny), __builtin_fmaxf(nz, tmin));
    tf = __builtin_fminf(__builtin_fminf(fx, fy), __builtin_fminf(fz, tmax));
}

// Sort keys of the four children (inner-node phase of trace_wavefront): the entry distance (exit widened by 2 ulp, like
// every box test of this path tracer) as bits, with the slot in the two lowest; all-ones if the ray does not enter the
// box within [tmin, tmax].
CRT_SLAB_FN void slab_packed_keys(const SlabNode &n, float tmin, float tmax, uint32_t key[4])
{
    float tn[4], tf[4];
    slab_packed_span<0>(n, tmin, tmax, tn[0], tf[0]);
    slab_packed_span<1>(n, tmin, tmax, tn[1], tf[1]);
    slab_packed_span<2>(n, tmin, tmax, tn[2], tf[2]);
    slab_packed_span<3>(n, tmin, tmax, tn[3], tf[3]);
    const float wide[4] = {tf[0] * 1.0000004f, tf[1] * 1.0000004f, tf[2] * 1.0000004f, tf[3] * 1.0000004f};
    for (int c = 0; c < 4; ++c) {
        key[c] = tn[c] <= wide[c] ? ((slab_bits(tn[c]) & 0x7ffffffcu) | (uint32_t)c) : 0xffffffffu;
    }
}

template <int C> CRT_SLAB_FN uint32_t slab_packed_key(const SlabNode &n, float tmin, float tmax)
{
    uint32_t key[4];
    slab_packed_keys(n, tmin, tmax, key);
    return key[C];
}

} // namespace crt
