// traverse_sim.cpp — CPU what-if tool for the traversal schedule (development aid, not product).
//
// Builds the product's 4-wide BVH with the product's builder (chameleonrt_amd/csrc/bvh_builder.cpp),
// walks real rays through it with the product's visit rule (DESIGN.md "Traversal rule"), and
// answers two questions before any GPU time is spent on them:
//
//  1. scalar, per ray: what does POSTPONING leaves cost? Policy k: a ray that reaches a leaf stashes
//     it and keeps traversing until k leaves are pending (or its stack is empty), then tests them
//     all. k = 1 is today's behaviour. Closest-hit rays lose culling (hit.t shrinks later), any-hit
//     rays terminate later; the tool reports nodes and triangles per ray for each k and checks that
//     the hits do not change.
//  2. wave model: 64 lanes with the kernel's phase scheduler (inner-node phase while a fraction of
//     the live lanes wants it, then one leaf step, batched retire + refill), counting VALU issue
//     slots with per-step costs taken from the kernel's ISA. Reports slots per ray and lanes per
//     slot for each threshold and for R = 1 or 2 rays held per lane (R = 2: a lane joins a phase if
//     either of its rays wants it, switching assumed free: an upper bound).
//
//   traverse_sim <tris.bin: n x 9 f32> <rays.bin: m x 8 f32 (org, dir, tmin, tmax)> <closest 0|1> [max_leaf]
//
// Build: g++ -O2 -std=c++17 -pthread -ffp-contract=off -Ichameleonrt_amd/csrc tools/traverse_sim.cpp
//        chameleonrt_amd/csrc/bvh_builder.cpp -o build/traverse_sim
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bvh_builder.h"

using namespace crt;

namespace {

struct Tri {
    float v0[3], e1[3], e2[3];
};
struct Ray {
    float o[3], d[3], tmin, tmax;
};

std::vector<QNode> g_nodes;
std::vector<Tri> g_tris;
QFrame g_frame;
int g_order = 0; // child ordering rule, see Lane::inner_step (TRAVERSE_SIM_ORDER)

inline float box_dir(float x) { return std::fabs(x) < 1e-18f ? std::copysign(1e-18f, x) : x; }

struct RayCtx {
    float qa[3], qb[3];
    const Ray *r;
};

inline void setup(const Ray &r, RayCtx &c)
{
    c.r = &r;
    for (int k = 0; k < 3; ++k) {
        const float inv = 1.f / box_dir(r.d[k]);
        c.qa[k] = g_frame.step[k] * inv;
        c.qb[k] = (g_frame.base[k] - r.o[k]) * inv;
    }
}

inline bool slab(const QChild &ch, const RayCtx &c, float tmin, float tmax, float &tn)
{
    float t0[3], t1[3];
    for (int k = 0; k < 3; ++k) {
        t0[k] = std::fma((float)ch.q[k][0], c.qa[k], c.qb[k]);
        t1[k] = std::fma((float)ch.q[k][1], c.qa[k], c.qb[k]);
    }
    tn = std::fmax(std::fmax(std::fmin(t0[0], t1[0]), std::fmin(t0[1], t1[1])), std::fmax(std::fmin(t0[2], t1[2]), tmin));
    const float tf = std::fmin(std::fmin(std::fmax(t0[0], t1[0]), std::fmax(t0[1], t1[1])),
                               std::fmin(std::fmax(t0[2], t1[2]), tmax));
    return tn <= tf * 1.0000004f;
}

inline bool tri_test(const Tri &tr, const Ray &r, float tfar, float &t)
{
    const float *v0 = tr.v0, *e1 = tr.e1, *e2 = tr.e2;
    const float ng[3] = {e2[1] * e1[2] - e2[2] * e1[1], e2[2] * e1[0] - e2[0] * e1[2], e2[0] * e1[1] - e2[1] * e1[0]};
    const float c[3] = {v0[0] - r.o[0], v0[1] - r.o[1], v0[2] - r.o[2]};
    const float rr[3] = {c[1] * r.d[2] - c[2] * r.d[1], c[2] * r.d[0] - c[0] * r.d[2], c[0] * r.d[1] - c[1] * r.d[0]};
    const float den = ng[0] * r.d[0] + ng[1] * r.d[1] + ng[2] * r.d[2];
    const float ad = std::fabs(den);
    const float sg = den < 0.f ? -1.f : 1.f;
    const float U = sg * (rr[0] * e2[0] + rr[1] * e2[1] + rr[2] * e2[2]);
    const float V = sg * (rr[0] * e1[0] + rr[1] * e1[1] + rr[2] * e1[2]);
    const float T = sg * (ng[0] * c[0] + ng[1] * c[1] + ng[2] * c[2]);
    if (den == 0.f || !(U >= 0.f && V >= 0.f && U + V <= ad) || !(T > ad * r.tmin && T <= ad * tfar)) {
        return false;
    }
    t = T / ad;
    return true;
}

// One lane's traversal state machine, advanced one step at a time so that both the scalar and the
// wave model can drive it. `pend` holds postponed leaves (at most K).
struct Lane {
    RayCtx ctx;
    int32_t cur = 0;
    std::vector<int32_t> stack;
    int32_t pend[4];
    int n_pend = 0;
    float best = 0.f;
    int32_t best_tri = -1;
    bool occluded = false, done = true;
    uint64_t nodes = 0, tris = 0;

    void begin(const Ray &r)
    {
        setup(r, ctx);
        cur = 0;
        stack.clear();
        n_pend = 0;
        best = r.tmax;
        best_tri = -1;
        occluded = false;
        done = false;
    }
    // a lane may take inner-node steps while it has room for another postponed leaf
    bool at_inner(int K) const { return !done && cur >= 0 && n_pend < K; }

    void pop()
    {
        if (stack.empty()) {
            cur = INT32_MIN; // nothing left to traverse
        } else {
            cur = stack.back();
            stack.pop_back();
        }
    }
    // after cur changed: stash leaves while there is room; finished when nothing is left at all
    void settle(int K)
    {
        while (cur < 0 && cur != INT32_MIN && n_pend < K) {
            pend[n_pend++] = cur;
            pop();
        }
        if (cur == INT32_MIN && n_pend == 0) {
            done = true;
        }
    }

    void inner_step(bool closest, int K)
    {
        const QNode &nd = g_nodes[cur];
        ++nodes;
        uint32_t keys[4];
        int n = 0;
        for (uint32_t k = 0; k < 4; ++k) {
            float tn;
            if (nd.child[k].q[0][0] <= nd.child[k].q[0][1] && slab(nd.child[k], ctx, ctx.r->tmin, best, tn)) { // not an unused slot
                uint32_t b;
                std::memcpy(&b, &tn, 4);
                keys[n++] = (b & 0x7ffffffcu) | k;
            }
        }
        (void)closest;
        if (n == 0) {
            pop();
        } else if (g_order == 0) { // full sort: nearest first, the rest stacked farthest first (the kernel today)
            std::sort(keys, keys + n);
            for (int k = n - 1; k >= 1; --k) {
                stack.push_back(nd.child[keys[k] & 3u].ref);
            }
            cur = nd.child[keys[0] & 3u].ref;
        } else if (g_order == 1) { // nearest first, the rest stacked in slot order (no sort network)
            int best = 0;
            for (int k = 1; k < n; ++k) {
                if (keys[k] < keys[best]) {
                    best = k;
                }
            }
            for (int k = n - 1; k >= 0; --k) {
                if (k != best) {
                    stack.push_back(nd.child[keys[k] & 3u].ref);
                }
            }
            cur = nd.child[keys[best] & 3u].ref;
        } else { // slot order only
            for (int k = n - 1; k >= 1; --k) {
                stack.push_back(nd.child[keys[k] & 3u].ref);
            }
            cur = nd.child[keys[0] & 3u].ref;
        }
        settle(K);
    }
    // tests all pending leaves; returns the number of triangle-loop iterations this lane needs
    int leaf_step(bool closest, int K)
    {
        int iters = 0;
        for (int p = 0; p < n_pend && !occluded; ++p) {
            const uint32_t x = ~(uint32_t)pend[p];
            const uint32_t first = x >> 3, count = (x & 7u) + 1u;
            for (uint32_t k = first; k < first + count; ++k) {
                ++tris;
                ++iters;
                float t;
                if (tri_test(g_tris[k], *ctx.r, ctx.r->tmax, t)) {
                    if (!closest) {
                        occluded = true;
                        break;
                    }
                    if (t < best || (t == best && (best_tri < 0 || (int32_t)k < best_tri))) {
                        best = t;
                        best_tri = (int32_t)k;
                    }
                }
            }
        }
        n_pend = 0;
        if (occluded) {
            done = true;
            return iters;
        }
        settle(K);
        return iters;
    }
};

} // namespace

int main(int argc, char **argv)
{
    if (argc < 4) {
        std::fprintf(stderr, "usage: traverse_sim tris.bin rays.bin closest [max_leaf]\n");
        return 2;
    }
    const bool closest = std::atoi(argv[3]) != 0;
    g_order = std::getenv("TRAVERSE_SIM_ORDER") ? std::atoi(std::getenv("TRAVERSE_SIM_ORDER")) : 0;
    const int max_leaf = argc > 4 ? std::atoi(argv[4]) : 2;
    std::vector<float> tv, rv;
    for (int f = 0; f < 2; ++f) {
        FILE *fp = std::fopen(argv[1 + f], "rb");
        if (!fp) {
            std::perror(argv[1 + f]);
            return 2;
        }
        std::fseek(fp, 0, SEEK_END);
        const long sz = std::ftell(fp);
        std::fseek(fp, 0, SEEK_SET);
        std::vector<float> &v = f == 0 ? tv : rv;
        v.resize(sz / 4);
        if (std::fread(v.data(), 4, v.size(), fp) != v.size()) {
            return 2;
        }
        std::fclose(fp);
    }
    const size_t nt = tv.size() / 9, nr = rv.size() / 8;
    std::vector<Aabb> boxes(nt);
    for (size_t i = 0; i < nt; ++i) {
        for (int k = 0; k < 3; ++k) {
            const float a = tv[9 * i + k], b = tv[9 * i + 3 + k], c = tv[9 * i + 6 + k];
            boxes[i].lo[k] = std::fmin(a, std::fmin(b, c));
            boxes[i].hi[k] = std::fmax(a, std::fmax(b, c));
        }
    }
    const BuiltBvh bvh = build_bvh(boxes.data(), nt, max_leaf, 0, 0, false, 85, 8);
    g_frame = make_frame(bvh.bounds);
    g_nodes.resize(bvh.nodes.size());
    for (size_t i = 0; i < bvh.nodes.size(); ++i) {
        g_nodes[i] = quantise(bvh.nodes[i], g_frame);
    }
    g_tris.resize(nt);
    for (size_t i = 0; i < nt; ++i) {
        const float *p = &tv[9 * (size_t)bvh.order[i]];
        for (int k = 0; k < 3; ++k) {
            g_tris[i].v0[k] = p[k];
            g_tris[i].e1[k] = p[k] - p[3 + k];
            g_tris[i].e2[k] = p[6 + k] - p[k];
        }
    }
    std::vector<Ray> rays(nr);
    std::memcpy(rays.data(), rv.data(), nr * sizeof(Ray));
    std::printf("%zu triangles, %zu wide nodes, %zu rays (%s)\n", nt, g_nodes.size(), nr, closest ? "closest hit" : "any hit");

    // ---- 1. scalar: cost of postponing leaves ---------------------------------------------------
    std::vector<int32_t> ref_hit(nr);
    for (int K = 1; K <= 3; ++K) {
        uint64_t nodes = 0, tris = 0, mismatches = 0;
        Lane ln;
        for (size_t i = 0; i < nr; ++i) {
            ln.begin(rays[i]);
            ln.nodes = ln.tris = 0;
            ln.settle(K);
            while (!ln.done) {
                if (ln.at_inner(K)) {
                    ln.inner_step(closest, K);
                } else {
                    ln.leaf_step(closest, K);
                }
            }
            nodes += ln.nodes;
            tris += ln.tris;
            const int32_t h = closest ? ln.best_tri : (ln.occluded ? 1 : 0);
            if (K == 1) {
                ref_hit[i] = h;
            } else if (h != ref_hit[i]) {
                ++mismatches;
            }
        }
        std::printf("scalar K=%d: %.2f nodes/ray, %.2f tris/ray, hits differing from K=1: %llu\n", K, (double)nodes / nr,
                    (double)tris / nr, (unsigned long long)mismatches);
    }

    // ---- 2. wave model ----------------------------------------------------------------------------
    // VALU issue slots per step, from the ISA of k_trace_closest
    const double C_INNER = 175, C_LEAF_FIXED = 30, C_TRI = 75, C_RETIRE = 70, C_REFILL = 90, C_LOOP = 25;
    const int REFILL_MIN = 16;
    // R = ray slots per lane: with R = 2 a lane takes part in a phase if EITHER of its rays wants it
    // (switching between them assumed free: an upper bound on what multiple rays per lane can give).
    for (int R = 1; R <= 2; ++R) {
        for (int num : {1, 2, 3}) {
            const int K = 1;
            const int den = num == 1 ? 2 : (num == 2 ? 3 : 4); // 1/2, 2/3, 3/4
            const int NS = 64 * R;
            std::vector<Lane> lane(NS);
            std::vector<int> ray_of(NS, -1);
            size_t next = 0;
            double slots = 0, lane_slots = 0, inner_slots = 0, inner_lanes = 0, leaf_slots = 0, leaf_lanes = 0;
            uint64_t nodes = 0;
            for (auto &l : lane) {
                l.done = true;
            }
            auto live = [&](int i) { return ray_of[i] >= 0 && !lane[i].done; };
            for (;;) {
                // retire + refill in batches (per ray slot; cost charged per wave-level event)
                int n_done = 0, n_idle = 0;
                for (int i = 0; i < NS; ++i) {
                    n_done += ray_of[i] >= 0 && lane[i].done;
                    n_idle += ray_of[i] < 0;
                }
                const bool exhausted = next >= nr;
                const int n_wait = exhausted ? n_done : n_done + n_idle;
                if (n_done > 0 && (n_wait >= REFILL_MIN || n_done + n_idle == NS)) {
                    slots += C_RETIRE;
                    lane_slots += C_RETIRE * std::min(n_done, 64);
                    for (int i = 0; i < NS; ++i) {
                        if (ray_of[i] >= 0 && lane[i].done) {
                            nodes += lane[i].nodes;
                            ray_of[i] = -1;
                        }
                    }
                    n_idle += n_done;
                }
                if (!exhausted && n_idle >= REFILL_MIN) {
                    int filled = 0;
                    for (int i = 0; i < NS && next < nr; ++i) {
                        if (ray_of[i] < 0) {
                            ray_of[i] = (int)next;
                            lane[i].begin(rays[next++]);
                            lane[i].nodes = lane[i].tris = 0;
                            lane[i].settle(K);
                            ++filled;
                        }
                    }
                    slots += C_REFILL * ((filled + 63) / 64);
                    lane_slots += C_REFILL * filled;
                }
                int n_live = 0;
                for (int i = 0; i < NS; ++i) {
                    n_live += live(i);
                }
                if (n_live == 0) {
                    bool any = false;
                    for (int i = 0; i < NS; ++i) {
                        any |= ray_of[i] >= 0;
                    }
                    if (!any && next >= nr) {
                        break;
                    }
                    continue;
                }
                slots += C_LOOP;
                // inner phase: lane l = ray slots {l, l + 64, ...}; it steps the first of them that is at an inner node
                for (;;) {
                    int lanes_inner = 0, lanes_live = 0;
                    for (int l = 0; l < 64; ++l) {
                        bool any_live = false, any_inner = false;
                        for (int r = 0; r < R; ++r) {
                            any_live |= live(l + 64 * r);
                            any_inner |= ray_of[l + 64 * r] >= 0 && lane[l + 64 * r].at_inner(K);
                        }
                        lanes_live += any_live;
                        lanes_inner += any_inner;
                    }
                    if (lanes_inner == 0 || den * lanes_inner < num * lanes_live) {
                        break;
                    }
                    for (int l = 0; l < 64; ++l) {
                        for (int r = 0; r < R; ++r) {
                            const int i = l + 64 * r;
                            if (ray_of[i] >= 0 && lane[i].at_inner(K)) {
                                lane[i].inner_step(closest, K);
                                break;
                            }
                        }
                    }
                    slots += C_INNER;
                    lane_slots += C_INNER * lanes_inner;
                    inner_slots += C_INNER;
                    inner_lanes += C_INNER * lanes_inner;
                }
                // one leaf step: every lane tests the pending leaf of the first of its rays that has one
                int n_leaf = 0, max_iters = 0;
                for (int l = 0; l < 64; ++l) {
                    for (int r = 0; r < R; ++r) {
                        const int i = l + 64 * r;
                        if (live(i) && !lane[i].at_inner(K)) {
                            ++n_leaf;
                            max_iters = std::max(max_iters, lane[i].leaf_step(closest, K));
                            break;
                        }
                    }
                }
                if (n_leaf > 0) {
                    const double c = C_LEAF_FIXED + C_TRI * max_iters;
                    slots += c;
                    lane_slots += c * n_leaf; // upper bound: lanes with fewer triangles idle in the loop
                    leaf_slots += c;
                    leaf_lanes += c * n_leaf;
                }
            }
            std::printf("wave R=%d threshold %d/%d: %.1f slots/ray, %.1f lanes/slot (inner %.1f over %.0f%% of the slots, leaf "
                        "%.1f over %.0f%%), %.2f nodes/ray\n",
                        R, num, den, slots / nr, lane_slots / slots, inner_lanes / std::max(1.0, inner_slots),
                        100.0 * inner_slots / slots, leaf_lanes / std::max(1.0, leaf_slots), 100.0 * leaf_slots / slots,
                        (double)nodes / nr);
        }
    }
    return 0;
}
