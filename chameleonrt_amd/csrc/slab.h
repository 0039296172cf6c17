// slab.h — the ray / quantised-box test of the traversal kernels, written so that the SAME source
// compiles for the device (hipcc) and for the host (g++: tests/native/slab_check.cpp checks it
// against the plain min/max formulation; the oracle's BVH walker mirrors the visit rule).
//
// A child's box is three dwords, one per axis: lo | hi << 16 in quanta of the BVH's QFrame. A plane
// at coordinate q has ray parameter t(q) = fma(q, qa, qb) with qa = step/d, qb = (base - o)/d
// (|d| clamped to >= 1e-18, sign kept). t(q) is monotone in q with the sign of qa, so the entry
// plane of an axis is `lo` if d > 0 and `hi` if d < 0: instead of computing both parameters and
// taking min / max (6 + 6 instructions per child), the dword is ROTATED by 0 or 16 bits (per ray
// and axis, from the sign of d) so that the entry plane sits in the low half, and near / far are
// read from fixed halves: 3 rotates per child, and the values are bit-identical to the min / max
// form. A box stored inverted (lo > hi) has near > far for every direction, so unused child slots
// need no separate test.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define CRT_SLAB_FN __host__ __device__ inline
#else
#define CRT_SLAB_FN inline
#endif

namespace crt {

struct SlabRay {
    float qa[3], qb[3];  // t(q) = fma(q, qa, qb) per axis
    // (the rotate count of an axis -- 0 if qa >= 0: entry plane = lo; 16 if qa < 0: entry plane = hi -- is derived from
    // qa's sign where it is used: three registers less to keep alive across the leaf steps of the traversal kernels)
};

CRT_SLAB_FN uint32_t slab_rotr(uint32_t w, uint32_t sh) { return (w >> sh) | (w << ((32u - sh) & 31u)); }

// bit pattern of a float / sign test without <cmath> or device headers
CRT_SLAB_FN uint32_t slab_bits(float x) { return __builtin_bit_cast(uint32_t, x); }

// rot for an axis whose plane parameter scales with qa
CRT_SLAB_FN uint32_t slab_rot_of(float qa) { return (slab_bits(qa) >> 31) << 4; }

// Entry distance of the ray into the child box {wx, wy, wz} clamped to tmin, and whether the ray
// enters it within [tmin, tmax] (exit widened by 2 ulp, like every box test of this path tracer).
CRT_SLAB_FN bool slab_enter(uint32_t wx, uint32_t wy, uint32_t wz, const SlabRay &r, float tmin, float tmax, float &tn)
{
    const uint32_t rx = slab_rotr(wx, slab_rot_of(r.qa[0])), ry = slab_rotr(wy, slab_rot_of(r.qa[1])), rz = slab_rotr(wz, slab_rot_of(r.qa[2]));
    const float nx = __builtin_fmaf((float)(rx & 0xffffu), r.qa[0], r.qb[0]), fx = __builtin_fmaf((float)(rx >> 16), r.qa[0], r.qb[0]);
    const float ny = __builtin_fmaf((float)(ry & 0xffffu), r.qa[1], r.qb[1]), fy = __builtin_fmaf((float)(ry >> 16), r.qa[1], r.qb[1]);
    const float nz = __builtin_fmaf((float)(rz & 0xffffu), r.qa[2], r.qb[2]), fz = __builtin_fmaf((float)(rz >> 16), r.qa[2], r.qb[2]);
    tn = __builtin_fmaxf(__builtin_fmaxf(nx, ny), __builtin_fmaxf(nz, tmin));
    const float tf = __builtin_fminf(__builtin_fminf(fx, fy), __builtin_fminf(fz, tmax));
    return tn <= tf * 1.0000004f;
}

// Sort key of a child (inner-node phase of trace_wavefront): entry distance bits with the slot in
// the two lowest, or all-ones if the ray does not enter the box.
CRT_SLAB_FN uint32_t slab_child_key(uint32_t wx, uint32_t wy, uint32_t wz, uint32_t slot, const SlabRay &r, float tmin,
                                    float tmax)
{
    float tn;
    return slab_enter(wx, wy, wz, r, tmin, tmax, tn) ? ((slab_bits(tn) & 0x7ffffffcu) | slot) : 0xffffffffu;
}

} // namespace crt
