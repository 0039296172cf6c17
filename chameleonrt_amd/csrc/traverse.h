// traverse.h — BVH2 traversal + ray/triangle intersection for gfx950 (the Embree stand-in).
//
// Replaces what the reference gets from rtcIntersectV / rtcOccludedV (call sites
// backends/embree/render_embree.ispc:245, :144, :170). Semantics (DESIGN.md
// "Traversal rule"), identical to what the parity tests check against:
//   * triangle record (v0, e1 = v0 - v1, e2 = v2 - v0), Ng = cross(e2, e1)
//   * valid hit: den != 0, U >= 0, V >= 0, U + V <= |den|, |den|*tnear < T <= |den|*tfar
//   * t = T/|den|, u = U/|den|, v = V/|den|
//   * closest hit = lexicographic min of (t, inst, geom, prim) -> independent of visit order
//   * occluded = any valid hit
// Boxes are tested with a conservative slab test (exit widened by 2 ulp, NaN-ignoring
// min/max), children visited nearest-entry first, far child pushed on a per-lane stack whose
// first LDS_STACK entries live in LDS ([depth][lane] so a wave's accesses are conflict-free)
// and the rest in scratch.
#pragma once
#include "pt_device.h"

namespace crt {

#ifndef CRT_LDS_STACK
#define CRT_LDS_STACK 8
#endif
constexpr int LDS_STACK = CRT_LDS_STACK; // per-lane stack entries kept in LDS
constexpr int SCRATCH_STACK = 56;  // overflow entries in private memory (LDS_STACK + this = 64)
constexpr int32_t STACK_SENTINEL = (int32_t)0x80000000; // marks "leave instance" (two-level)

struct RayHit {
    float t, u, v;
    int32_t tri;  // global index into SceneView::tris, -1 = miss
    int32_t inst; // instance index
};

struct TraversalStack {
    int32_t *lds; // this lane's column: entry k at lds[k * stride]
    int stride;
    int32_t spill[SCRATCH_STACK];
    int sp;
    CRT_DEV void push(int32_t x)
    {
        if (sp < LDS_STACK) {
            lds[sp * stride] = x;
        } else {
            spill[sp - LDS_STACK] = x;
        }
        ++sp;
    }
    CRT_DEV int32_t pop()
    {
        --sp;
        return sp < LDS_STACK ? lds[sp * stride] : spill[sp - LDS_STACK];
    }
};

CRT_DEV V3 xfm_point(const float *m, V3 p) // column-major affine, rows evaluated left to right
{
    return v3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
              m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
CRT_DEV V3 xfm_vector(const float *m, V3 v)
{
    return v3(m[0] * v.x + m[4] * v.y + m[8] * v.z, m[1] * v.x + m[5] * v.y + m[9] * v.z,
              m[2] * v.x + m[6] * v.y + m[10] * v.z);
}

// Slab test of one child box; returns entry distance in tn.
CRT_DEV bool slab(float lox, float loy, float loz, float hix, float hiy, float hiz, V3 o, V3 inv, float tmin,
                  float tmax, float &tn)
{
    const float t0x = (lox - o.x) * inv.x, t1x = (hix - o.x) * inv.x;
    const float t0y = (loy - o.y) * inv.y, t1y = (hiy - o.y) * inv.y;
    const float t0z = (loz - o.z) * inv.z, t1z = (hiz - o.z) * inv.z;
    tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), tmin));
    const float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax));
    return tn <= tf * 1.0000004f;
}

CRT_DEV bool tri_test(const float4 a, const float4 b, const float4 c, V3 O, V3 D, float tnear, float tfar,
                      float &t, float &u, float &v)
{
    const V3 v0 = v3(a.x, a.y, a.z), e1 = v3(a.w, b.x, b.y), e2 = v3(b.z, b.w, c.x);
    const V3 Ng = cross3(e2, e1);
    const V3 C = v0 - O;
    const V3 R = cross3(C, D);
    const float den = dot3(Ng, D);
    const float abs_den = fabsf(den);
    const uint32_t sgn = __float_as_uint(den) & 0x80000000u;
    const float U = __uint_as_float(__float_as_uint(dot3(R, e2)) ^ sgn);
    const float V = __uint_as_float(__float_as_uint(dot3(R, e1)) ^ sgn);
    const float T = __uint_as_float(__float_as_uint(dot3(Ng, C)) ^ sgn);
    if (den == 0.f) {
        return false;
    }
    if (!(U >= 0.f && V >= 0.f && U + V <= abs_den)) {
        return false;
    }
    if (!(T > abs_den * tnear && T <= abs_den * tfar)) {
        return false;
    }
    t = T / abs_den;
    u = U / abs_den;
    v = V / abs_den;
    return true;
}

// top: LDS copy of nodes [sc.root, sc.root + sc.n_top_nodes) or nullptr.
template <bool ANY_HIT, bool TWO_LEVEL, bool COUNTERS>
CRT_DEV void traverse(const SceneView &sc, const BvhNode *top, V3 org, V3 dir, float tnear, float tfar,
                      RayHit &hit, TraversalStack &st, uint32_t &n_nodes, uint32_t &n_tris)
{
    hit.t = tfar;
    hit.u = hit.v = 0.f;
    hit.tri = -1;
    hit.inst = -1;
    uint32_t best_geom = 0, best_prim = 0;
    st.sp = 0;

    V3 o = org, d = dir;
    int32_t cur_inst = 0;
    bool in_blas = !TWO_LEVEL;
    if (!TWO_LEVEL) {
        const InstanceRec &in = sc.instances[0];
        if (!in.identity) {
            o = xfm_point(in.w2o, org);
            d = xfm_vector(in.w2o, dir);
        }
    }
    V3 inv = v3(1.f / d.x, 1.f / d.y, 1.f / d.z);
    int32_t cur = sc.root;
    const int32_t top_lo = sc.root, top_hi = sc.root + (int32_t)sc.n_top_nodes;

    for (;;) {
        if (cur >= 0) {
            float4 q0, q1, q2, q3;
            if (top != nullptr && cur >= top_lo && cur < top_hi) {
                const float4 *p = reinterpret_cast<const float4 *>(top + (cur - top_lo));
                q0 = p[0];
                q1 = p[1];
                q2 = p[2];
                q3 = p[3];
            } else {
                const float4 *p = reinterpret_cast<const float4 *>(sc.nodes + cur);
                q0 = p[0];
                q1 = p[1];
                q2 = p[2];
                q3 = p[3];
            }
            if (COUNTERS) {
                ++n_nodes;
            }
            float t0, t1;
            const bool h0 = slab(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, o, inv, tnear, hit.t, t0);
            const bool h1 = slab(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, o, inv, tnear, hit.t, t1);
            const int32_t c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y);
            if (h0 && h1) {
                const bool first0 = t0 <= t1;
                st.push(first0 ? c1 : c0);
                cur = first0 ? c0 : c1;
                continue;
            }
            if (h0) {
                cur = c0;
                continue;
            }
            if (h1) {
                cur = c1;
                continue;
            }
        } else {
            const uint32_t x = ~(uint32_t)cur;
            const uint32_t first = x >> 3;
            if (TWO_LEVEL && !in_blas) {
                // TLAS leaf: enter the instance (Embree transforms the ray, keeps t)
                const InstanceRec &in = sc.instances[first];
                cur_inst = (int32_t)first;
                if (!in.identity) {
                    o = xfm_point(in.w2o, org);
                    d = xfm_vector(in.w2o, dir);
                    inv = v3(1.f / d.x, 1.f / d.y, 1.f / d.z);
                }
                in_blas = true;
                st.push(STACK_SENTINEL);
                cur = in.blas_root;
                continue;
            }
            const uint32_t count = (x & 7u) + 1u;
            bool occluded = false;
            for (uint32_t k = first; k < first + count; ++k) {
                const float4 *p = reinterpret_cast<const float4 *>(sc.tris + k);
                const float4 a = p[0], b = p[1], c = p[2];
                if (COUNTERS) {
                    ++n_tris;
                }
                float t, u, v;
                if (tri_test(a, b, c, o, d, tnear, tfar, t, u, v)) {
                    if (ANY_HIT) {
                        occluded = true;
                        break;
                    }
                    const uint32_t geom = __float_as_uint(c.y), prim = __float_as_uint(c.z);
                    bool take = t < hit.t;
                    if (t == hit.t && hit.tri >= 0) { // tie: (inst, geom, prim) decides
                        take = cur_inst != hit.inst ? cur_inst < hit.inst
                                                    : (geom != best_geom ? geom < best_geom : prim < best_prim);
                    } else if (t == hit.t) {
                        take = true; // first hit exactly at tfar
                    }
                    if (take) {
                        hit.t = t;
                        hit.u = u;
                        hit.v = v;
                        hit.tri = (int32_t)k;
                        hit.inst = cur_inst;
                        best_geom = geom;
                        best_prim = prim;
                    }
                }
            }
            if (ANY_HIT && occluded) {
                hit.tri = 0;
                hit.inst = cur_inst;
                hit.t = 0.f;
                return;
            }
        }
        // pop
        for (;;) {
            if (st.sp == 0) {
                return;
            }
            cur = st.pop();
            if (TWO_LEVEL && cur == STACK_SENTINEL) {
                o = org;
                d = dir;
                inv = v3(1.f / d.x, 1.f / d.y, 1.f / d.z);
                in_blas = false;
                continue;
            }
            break;
        }
    }
}

} // namespace crt
