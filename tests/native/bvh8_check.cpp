// bvh8_check.cpp — host-only check of the 8-wide build (build_bvh width 8, quantise8; crt_types.h QNode8), built with g++ by
// tests/test_bvh_builder.py.   bvh8_check <n_items> <threads> <seed> <mode>    (modes as bvh_check.cpp: 0 scattered boxes,
//                                                                              1 flat axis-aligned quads far from the origin, 2 duplicates)
// Verifies: every item is the single item of exactly one leaf; inner children come first in a node and are consecutive node
// records, leaf children consecutive positions of the item order (quantise8 throws otherwise); every child box contains its
// subtree; the reported depth is the walked one; and for every node the DEQUANTISED child boxes -- origin + (byte << e) quanta
// of the frame, evaluated in fp32 -- contain the full-precision boxes (both quantisations may only grow a box); unused
// children are inverted (lo = 255, hi = 0).
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
        if (mode == 1) {
            for (int k = 0; k < 3; ++k) {
                c[k] = std::floor(c[k] * 4.f) * 0.25f + 1000.f;
                h[k] = 0.125f;
            }
            h[rng() % 3] = 0.f;
        } else if (mode == 2 && i % 3) {
            boxes[i] = boxes[i - 1];
            continue;
        }
        for (int k = 0; k < 3; ++k) {
            boxes[i].lo[k] = c[k] - h[k];
            boxes[i].hi[k] = c[k] + h[k];
        }
    }
    const BuiltBvh b = build_bvh(boxes.data(), n, 1, 0, 0, false, 85, threads, BVH8_WIDTH);
    const QFrame f = make_frame(b.bounds);
    int errors = 0;
    std::vector<char> seen(n, 0);
    std::vector<Aabb> sub(b.nodes8.size());
    std::vector<uint32_t> depth(b.nodes8.size(), 0);
    uint32_t walked_depth = 0;
    size_t used_total = 0, inflated = 0;
    // children have larger indices than their parent (blocks are allocated when the parent is laid out): bottom-up by index
    for (size_t i = b.nodes8.size(); i-- > 0;) {
        const BvhNode8 &nd = b.nodes8[i];
        Aabb all;
        for (int k = 0; k < 3; ++k) {
            all.lo[k] = INFINITY;
            all.hi[k] = -INFINITY;
        }
        int used = 0, inner = 0;
        uint32_t d = 1;
        for (int c = 0; c < BVH8_WIDTH; ++c) {
            if (nd.c[c] == EMPTY_CHILD) {
                continue;
            }
            errors += c != used; // used children come first
            ++used;
            Aabb cb;
            if (nd.c[c] >= 0) {
                errors += inner != c;                 // inner children before leaves
                errors += (size_t)nd.c[c] <= i;      // and behind their parent in the array
                ++inner;
                cb = sub[(size_t)nd.c[c]];
                d = std::max(d, depth[(size_t)nd.c[c]] + 1);
            } else {
                const uint32_t x = ~(uint32_t)nd.c[c];
                errors += (x & 7u) != 0u; // one item per leaf
                const uint32_t id = b.order[x >> 3];
                errors += seen[id]++ != 0;
                cb = boxes[id];
            }
            for (int k = 0; k < 3; ++k) {
                errors += cb.lo[k] < nd.lo[c][k] || cb.hi[k] > nd.hi[c][k]; // the child box contains its subtree
                all.lo[k] = std::fmin(all.lo[k], cb.lo[k]);
                all.hi[k] = std::fmax(all.hi[k], cb.hi[k]);
            }
        }
        errors += used == 0 || (used == 1 && b.nodes8.size() > 1);
        used_total += (size_t)used;
        sub[i] = all;
        depth[i] = d;
        if (i == 0) {
            walked_depth = d;
        }
        const QNode8 q = quantise8(nd, f); // throws if the children are not consecutive blocks
        errors += (int)(q.meta >> 12) != inner;
        for (int a = 0; a < 3; ++a) {
            const uint32_t e = (q.meta >> (4 * a)) & 15u;
            errors += e > 9u; // 65535 quanta over 255 cells
            for (int c = 0; c < BVH8_WIDTH; ++c) {
                if (c >= used) {
                    errors += !(q.lo[a][c] == 255 && q.hi[a][c] == 0);
                    continue;
                }
                const float dl = f.base[a] + (float)(q.org[a] + ((uint32_t)q.lo[a][c] << e)) * f.step[a];
                const float dh = f.base[a] + (float)(q.org[a] + ((uint32_t)q.hi[a][c] << e)) * f.step[a];
                errors += !(dl <= nd.lo[c][a]) || !(dh >= nd.hi[c][a]);
                inflated += (nd.lo[c][a] - dl > (float)(3u << e) * f.step[a]) + (dh - nd.hi[c][a] > (float)(3u << e) * f.step[a]);
            }
        }
    }
    errors += walked_depth != b.max_depth;
    for (size_t i = 0; i < n; ++i) {
        errors += seen[i] != 1;
    }
    std::printf("items %zu nodes %zu fill %.2f depth %u top %u errors %d over_3_cells %zu\n", n, b.nodes8.size(),
                b.nodes8.empty() ? 0.0 : (double)used_total / (BVH8_WIDTH * b.nodes8.size()), b.max_depth, b.n_top, errors, inflated);
    return errors == 0 && inflated == 0 ? 0 : 1;
}
