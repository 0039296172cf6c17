// bvh_check.cpp — host-only check of the product's BVH builder and node quantiser
// (chameleonrt_amd/csrc/bvh_builder.{h,cpp}); built with g++ by tests/test_bvh_builder.py.
//
//   bvh_check <n_items> <threads> <seed> <mode>      mode: 0 scattered boxes, 1 axis-aligned flat
//                                                    quads far from the origin, 2 duplicates
// Verifies: every item is in exactly one leaf; every child box contains its subtree's items;
// leaves hold <= max_leaf items; depth fits the traversal stack; and for every node the
// DEQUANTISED child boxes -- base + q*step evaluated in fp32 exactly like the kernels' planes --
// contain the full-precision boxes (quantisation may only grow a box).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "bvh_builder.h"

using namespace crt;

int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 100000;
    const int threads = argc > 2 ? atoi(argv[2]) : 4;
    const unsigned seed = argc > 3 ? (unsigned)atoi(argv[3]) : 1;
    const int mode = argc > 4 ? atoi(argv[4]) : 0;
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    std::vector<Aabb> boxes(n);
    for (size_t i = 0; i < n; ++i) {
        float c[3] = {U(rng) * 40.f - 20.f, U(rng) * 6.f, U(rng) * 40.f - 20.f};
        float h[3] = {0.01f + 0.1f * U(rng), 0.01f + 0.1f * U(rng), 0.01f + 0.1f * U(rng)};
        if (mode == 1) { // voxel-city like: flat, axis-aligned, offset 1000 units from the origin
            for (int k = 0; k < 3; ++k) {
                c[k] = std::floor(c[k] * 4.f) * 0.25f + 1000.f;
                h[k] = 0.125f;
            }
            h[rng() % 3] = 0.f;
        } else if (mode == 2 && i % 3) { // many identical items
            boxes[i] = boxes[i - 1];
            continue;
        }
        for (int k = 0; k < 3; ++k) {
            boxes[i].lo[k] = c[k] - h[k];
            boxes[i].hi[k] = c[k] + h[k];
        }
    }
    const int max_leaf = 4;
    const BuiltBvh b = build_bvh(boxes.data(), n, max_leaf, 0, 0, false, 127, threads);
    const QFrame f = make_frame(b.bounds);
    std::vector<char> seen(n, 0);
    int errors = 0;
    size_t inflated = 0, planes = 0;
    double slack = 0.0;
    // subtree bounds by recursion over the emitted nodes
    struct Rec {
        static void subtree(const BuiltBvh &b, const std::vector<Aabb> &boxes, int32_t ref, Aabb &out, std::vector<char> &seen,
                            int &errors, int max_leaf)
        {
            for (int k = 0; k < 3; ++k) {
                out.lo[k] = INFINITY;
                out.hi[k] = -INFINITY;
            }
            if (ref < 0) {
                const uint32_t x = ~(uint32_t)ref, first = x >> 3, cnt = (x & 7u) + 1u;
                if ((int)cnt > max_leaf) {
                    ++errors;
                }
                for (uint32_t i = first; i < first + cnt; ++i) {
                    const uint32_t id = b.order[i];
                    if (seen[id]++) {
                        ++errors;
                    }
                    for (int k = 0; k < 3; ++k) {
                        out.lo[k] = std::fmin(out.lo[k], boxes[id].lo[k]);
                        out.hi[k] = std::fmax(out.hi[k], boxes[id].hi[k]);
                    }
                }
                return;
            }
            const BvhNode &nd = b.nodes[ref];
            Aabb c0, c1;
            subtree(b, boxes, nd.c0, c0, seen, errors, max_leaf);
            if (nd.c1 == nd.c0) { // a one-leaf BVH lists its leaf twice (the repeat loses every tie)
                c1 = c0;
            } else {
                subtree(b, boxes, nd.c1, c1, seen, errors, max_leaf);
            }
            for (int k = 0; k < 3; ++k) {
                if (c0.lo[k] < nd.lo0[k] || c0.hi[k] > nd.hi0[k] || c1.lo[k] < nd.lo1[k] || c1.hi[k] > nd.hi1[k]) {
                    ++errors; // a child box does not contain its subtree
                }
                out.lo[k] = std::fmin(c0.lo[k], c1.lo[k]);
                out.hi[k] = std::fmax(c0.hi[k], c1.hi[k]);
            }
        }
    };
    Aabb all;
    Rec::subtree(b, boxes, 0, all, seen, errors, max_leaf);
    for (size_t i = 0; i < n; ++i) {
        if (seen[i] != 1) {
            ++errors;
        }
    }
    if (b.max_depth > 52) {
        ++errors;
    }
    for (const BvhNode &nd : b.nodes) {
        const QNode q = quantise(nd, f);
        const float *lo[2] = {nd.lo0, nd.lo1}, *hi[2] = {nd.hi0, nd.hi1};
        const uint16_t *qlo[2] = {q.lo0, q.lo1}, *qhi[2] = {q.hi0, q.hi1};
        for (int c = 0; c < 2; ++c) {
            for (int k = 0; k < 3; ++k) {
                // the kernels place the plane at base + q*step (as fma(q, step/d, (base-o)/d)); in fp32:
                const float dl = f.base[k] + (float)qlo[c][k] * f.step[k];
                const float dh = f.base[k] + (float)qhi[c][k] * f.step[k];
                if (!(dl <= lo[c][k]) || !(dh >= hi[c][k])) {
                    ++errors;
                }
                slack += (lo[c][k] - dl) + (dh - hi[c][k]);
                planes += 2;
                inflated += (lo[c][k] - dl > 3.f * f.step[k]) + (dh - hi[c][k] > 3.f * f.step[k]);
            }
        }
        if (q.c0 != nd.c0 || q.c1 != nd.c1) {
            ++errors;
        }
    }
    std::printf("items %zu nodes %zu depth %u top %u errors %d mean_slack_quanta %.3f over_3_quanta %zu\n", n, b.nodes.size(),
                b.max_depth, b.n_top, errors, planes ? slack / planes / f.step[0] : 0.0, inflated);
    return errors == 0 && inflated == 0 ? 0 : 1;
}
