// slab_check.cpp — host-side check of chameleonrt_amd/csrc/slab.h, the ray / packed-node box test the traversal
// kernels compile (same source, same operations; built with g++ -ffp-contract=off by tests/test_slab.py). It must agree
// BIT FOR BIT with the plain formulation the oracle's BVH walker uses (both plane parameters per axis, min / max, unused
// slots skipped explicitly), on nodes made by the product's own packer (crt_types.h pack_node) from random 16-bit boxes:
//   * same enter / miss decision and the same sort key for every child, including rays with exactly zero direction
//     components (1/d clamped, sign kept: +0 and -0), origins inside, on and far outside the frame, tmin / tmax that cut
//     the box, flat boxes, nodes of every scale from one grid unit to the whole frame;
//   * every ray that enters a child's 16-bit box as the builders made it (plain one-step test) also enters the packed box
//     -- packing may only widen;
//   * an unused slot (inverted box) is never entered by a ray whose two plane parameters differ on at least one axis.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

#include "crt_types.h"
#include "slab.h"

using namespace crt;

static float box_dir(float x) { return std::fabs(x) < 1e-18f ? std::copysign(1e-18f, x) : x; }

// the oracle's form: planes origin + byte * scale, a = fma(origin, qa, qb), s = qa * scale, t = fma(byte, s, a), min / max
static bool reference(const PNode &p, int c, const float qa[3], const float qb[3], float tmin, float tmax, float &tn)
{
    const uint32_t origin[3] = {p.frame[0] & 0xffffu, p.frame[0] >> 16, p.frame[1] & 0xffffu};
    const uint32_t code[3] = {(p.frame[1] >> 16) & 31u, (p.frame[1] >> 21) & 31u, p.frame[1] >> 26};
    const uint32_t lo_w[3] = {p.lo_x, p.lo_y, p.lo_z}, hi_w[3] = {p.hi_x, p.hi_y, p.hi_z};
    float n[3], f[3];
    for (int k = 0; k < 3; ++k) {
        const float a = std::fma((float)origin[k], qa[k], qb[k]);
        const float s = qa[k] * std::ldexp((code[k] & 1u) ? 1.5f : 1.f, (int)(code[k] >> 1));
        const float t0 = std::fma((float)((lo_w[k] >> (8 * c)) & 255u), s, a), t1 = std::fma((float)((hi_w[k] >> (8 * c)) & 255u), s, a);
        n[k] = std::fmin(t0, t1);
        f[k] = std::fmax(t0, t1);
    }
    tn = std::fmax(std::fmax(n[0], n[1]), std::fmax(n[2], tmin));
    const float tf = std::fmin(std::fmin(f[0], f[1]), std::fmin(f[2], tmax));
    return tn <= tf * 1.0000004f;
}

// the 16-bit box as the builders deliver it, one-step planes
static bool wide_box(const QChild &q, const float qa[3], const float qb[3], float tmin, float tmax)
{
    float n[3], f[3];
    for (int k = 0; k < 3; ++k) {
        const float t0 = std::fma((float)q.q[k][0], qa[k], qb[k]), t1 = std::fma((float)q.q[k][1], qa[k], qb[k]);
        n[k] = std::fmin(t0, t1);
        f[k] = std::fmax(t0, t1);
    }
    const float tn = std::fmax(std::fmax(n[0], n[1]), std::fmax(n[2], tmin));
    const float tf = std::fmin(std::fmin(f[0], f[1]), std::fmin(f[2], tmax));
    return tn <= tf;
}

template <int C> static void check_child(const PNode &p, const SlabNode &sn, const QNode &q, bool used, const float qa[3], const float qb[3],
                                         float tmin, float tmax, long &errors, long &entered, long &inverted_entered, long &narrowed)
{
    const uint32_t key = slab_packed_key<C>(sn, tmin, tmax);
    float tn_ref;
    const bool h_ref = reference(p, C, qa, qb, tmin, tmax, tn_ref);
    uint32_t b_ref;
    std::memcpy(&b_ref, &tn_ref, 4);
    if (used) {
        if (key != (h_ref ? ((b_ref & 0x7ffffffcu) | (uint32_t)C) : 0xffffffffu)) {
            if (++errors <= 5) {
                std::fprintf(stderr, "child %d: key %08x, min/max form %d %.9g\n", C, key, (int)h_ref, tn_ref);
            }
        }
        entered += h_ref;
        if (wide_box(q.child[C], qa, qb, tmin, tmax) && key == 0xffffffffu) {
            ++narrowed; // the packed box must contain the builders' box
        }
    } else if (key != 0xffffffffu) {
        // allowed only if the two planes are indistinguishable on every axis (precision collapse)
        bool collapsed = true;
        const SlabAxis ax[3] = {sn.x, sn.y, sn.z};
        for (int k = 0; k < 3; ++k) {
            const float t0 = std::fma(255.f, ax[k].s, ax[k].a), t1 = ax[k].a;
            collapsed &= std::fabs(t0 - t1) <= 4e-7f * std::fmax(std::fabs(t0), std::fabs(t1));
        }
        ++inverted_entered;
        errors += !collapsed;
    }
}

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 500000;
    std::mt19937 rng(12345);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    long errors = 0, entered = 0, inverted_entered = 0, zero_dirs = 0, narrowed = 0, scales[32] = {0};
    for (long i = 0; i < n; ++i) {
        // frame and ray
        const float step[3] = {std::ldexp(1.f + 0.5f * U(rng), -10 - (int)(rng() % 8)), std::ldexp(1.f, -12), 3e-4f};
        const float base[3] = {10.f * U(rng), 10.f * U(rng), 1000.f * U(rng)};
        const float scale = (i % 7 == 0) ? 1e6f : ((i % 11 == 0) ? 1e-3f : 10.f); // far away / inside
        float o[3] = {scale * U(rng), scale * U(rng), scale * U(rng)}, d[3] = {U(rng), U(rng), U(rng)};
        if (i % 5 == 0) {
            d[rng() % 3] = (rng() & 1) ? 0.f : -0.f; // exactly zero components, both signs
            ++zero_dirs;
        }
        if (i % 13 == 0) {
            d[rng() % 3] = 1e-30f * U(rng); // below the clamp
        }
        // a node: 1..4 children inside a parent box of a random size (one grid unit .. the whole frame)
        QNode q;
        const int n_used = 1 + (int)(rng() % 4);
        uint32_t plo[3], pext[3];
        for (int k = 0; k < 3; ++k) {
            pext[k] = 1u << (rng() % 17); // 1 .. 65536
            pext[k] = std::min(pext[k] + (uint32_t)(rng() % pext[k]), 65535u);
            plo[k] = (uint32_t)(rng() % (65536u - pext[k]));
        }
        for (int c = 0; c < 4; ++c) {
            for (int k = 0; k < 3; ++k) {
                if (c < n_used) {
                    const uint32_t a = plo[k] + (uint32_t)(rng() % (pext[k] + 1u)), b = plo[k] + (uint32_t)(rng() % (pext[k] + 1u));
                    q.child[c].q[k][0] = (uint16_t)std::min(a, b);
                    q.child[c].q[k][1] = (uint16_t)((i % 17 == 0) ? std::min(a, b) : std::max(a, b)); // sometimes flat
                } else {
                    q.child[c].q[k][0] = 65535;
                    q.child[c].q[k][1] = 0;
                }
            }
            q.child[c].ref = c < n_used ? 100 + c : 100;
        }
        const PNode p = pack_node(q);
        ++scales[(p.frame[1] >> 16) & 31u];
        if (i % 5 != 0 && i % 3 != 0) { // aim most rays at (or just past the edge of) a child so that many enter it
            const int c = (int)(rng() % (unsigned)n_used);
            for (int k = 0; k < 3; ++k) {
                const float s = 0.5f + 0.55f * U(rng); // [-0.05, 1.05] across the box
                const float target = base[k] + ((float)q.child[c].q[k][0] + s * (float)(q.child[c].q[k][1] - q.child[c].q[k][0])) * step[k];
                d[k] = target - o[k];
            }
        }
        SlabRay r;
        float qa[3], qb[3];
        for (int k = 0; k < 3; ++k) {
            const float inv = 1.f / box_dir(d[k]);
            qa[k] = r.qa[k] = step[k] * inv;
            qb[k] = r.qb[k] = (base[k] - o[k]) * inv;
        }
        const float tmin = (i & 1) ? 0.f : 1e-4f;
        const float tmax = (i % 4 != 1) ? 1e20f : std::fabs(2.f * U(rng)); // d is unnormalised when aimed: t ~ 1 at the box
        const SlabNode sn = slab_node(p.frame[0], p.frame[1], p.lo_x, p.hi_x, p.lo_y, p.hi_y, p.lo_z, p.hi_z, r);
        check_child<0>(p, sn, q, 0 < n_used, qa, qb, tmin, tmax, errors, entered, inverted_entered, narrowed);
        check_child<1>(p, sn, q, 1 < n_used, qa, qb, tmin, tmax, errors, entered, inverted_entered, narrowed);
        check_child<2>(p, sn, q, 2 < n_used, qa, qb, tmin, tmax, errors, entered, inverted_entered, narrowed);
        check_child<3>(p, sn, q, 3 < n_used, qa, qb, tmin, tmax, errors, entered, inverted_entered, narrowed);
    }
    int n_scales = 0;
    for (long s : scales) {
        n_scales += s > 0;
    }
    std::printf("nodes %ld children entered %ld zero-direction rays %ld scales seen %d inverted boxes entered %ld narrowed %ld errors %ld\n", n,
                entered, zero_dirs, n_scales, inverted_entered, narrowed, errors);
    return errors == 0 && narrowed == 0 && entered > n / 100 && n_scales >= (CRT_PNODE_HALF_STEPS ? 16 : 9) ? 0 : 1;
}
