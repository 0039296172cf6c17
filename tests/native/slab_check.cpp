// slab_check.cpp — host-side check of chameleonrt_amd/csrc/slab.h, the ray / quantised-box test the
// traversal kernels compile (same source, same operations; built with g++ -ffp-contract=off by
// tests/test_slab.py). It must agree BIT FOR BIT with the plain formulation the oracle's BVH walker
// uses (both plane parameters per axis, min / max, unused slots skipped explicitly):
//   * same enter / miss decision and the same entry distance for every ordinary box and ray,
//     including rays with exactly zero direction components (1/d clamped, sign kept: +0 and -0),
//     origins inside, on and far outside the frame, and tmin / tmax that cut the box;
//   * an inverted box (an unused child slot) is never entered by a ray whose two plane parameters
//     differ on at least one axis.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

#include "slab.h"

using namespace crt;

static float box_dir(float x) { return std::fabs(x) < 1e-18f ? std::copysign(1e-18f, x) : x; }

static bool reference(const uint16_t q[3][2], const float qa[3], const float qb[3], float tmin, float tmax, float &tn)
{
    float n[3], f[3];
    for (int k = 0; k < 3; ++k) {
        const float t0 = std::fma((float)q[k][0], qa[k], qb[k]), t1 = std::fma((float)q[k][1], qa[k], qb[k]);
        n[k] = std::fmin(t0, t1);
        f[k] = std::fmax(t0, t1);
    }
    tn = std::fmax(std::fmax(n[0], n[1]), std::fmax(n[2], tmin));
    const float tf = std::fmin(std::fmin(f[0], f[1]), std::fmin(f[2], tmax));
    return tn <= tf * 1.0000004f;
}

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 2000000;
    std::mt19937 rng(12345);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::uniform_int_distribution<int> Q(0, 65535);
    long errors = 0, entered = 0, inverted_entered = 0, zero_dirs = 0;
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
        uint16_t q[3][2];
        for (int k = 0; k < 3; ++k) {
            const int a = Q(rng), b = Q(rng);
            q[k][0] = (uint16_t)std::min(a, b);
            q[k][1] = (uint16_t)std::max(a, b);
            if (i % 17 == 0) {
                q[k][1] = q[k][0]; // flat box
            }
        }
        if (i % 5 != 0 && i % 3 != 0) { // aim most rays at (or just past the edge of) the box so that many enter it
            for (int k = 0; k < 3; ++k) {
                const float s = 0.5f + 0.55f * U(rng); // [-0.05, 1.05] across the box
                const float target = base[k] + ((float)q[k][0] + s * (float)(q[k][1] - q[k][0])) * step[k];
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
        uint32_t w[3];
        for (int k = 0; k < 3; ++k) {
            w[k] = (uint32_t)q[k][0] | ((uint32_t)q[k][1] << 16);
        }
        float tn_ref, tn_new;
        const bool h_ref = reference(q, qa, qb, tmin, tmax, tn_ref);
        const bool h_new = slab_enter(w[0], w[1], w[2], r, tmin, tmax, tn_new);
        uint32_t b_ref, b_new;
        std::memcpy(&b_ref, &tn_ref, 4);
        std::memcpy(&b_new, &tn_new, 4);
        if (h_ref != h_new || b_ref != b_new) {
            if (++errors <= 5) {
                std::fprintf(stderr, "mismatch at %ld: ref %d %.9g new %d %.9g\n", i, (int)h_ref, tn_ref, (int)h_new, tn_new);
            }
        }
        const uint32_t key = slab_child_key(w[0], w[1], w[2], 2u, r, tmin, tmax);
        if (key != (h_ref ? ((b_ref & 0x7ffffffcu) | 2u) : 0xffffffffu)) {
            ++errors;
        }
        entered += h_ref;
        // the unused-slot box: inverted on every axis
        const uint32_t inv_w = 65535u; // lo = 65535, hi = 0
        float tn_inv;
        if (slab_enter(inv_w, inv_w, inv_w, r, tmin, tmax, tn_inv)) {
            // allowed only if the two planes are indistinguishable on every axis (precision collapse)
            bool collapsed = true;
            for (int k = 0; k < 3; ++k) {
                const float t0 = std::fma(65535.f, qa[k], qb[k]), t1 = std::fma(0.f, qa[k], qb[k]);
                collapsed &= std::fabs(t0 - t1) <= 4e-7f * std::fmax(std::fabs(t0), std::fabs(t1));
            }
            inverted_entered += 1;
            errors += !collapsed;
        }
    }
    std::printf("boxes %ld entered %ld zero-direction rays %ld inverted boxes entered %ld errors %ld\n", n, entered, zero_dirs,
                inverted_entered, errors);
    return errors == 0 && entered > n / 100 ? 0 : 1;
}
