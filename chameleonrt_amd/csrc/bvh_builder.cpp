// bvh_builder.cpp — parallel top-down binned-SAH build, collapse to 4-wide nodes, cache-aware node layout.
#include "bvh_builder.h"
#include "bvh_device.h"
#include "host_parallel.h"
#include "lbvh.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <limits>
#include <memory>
#include <queue>
#include <stdexcept>

namespace crt {
namespace {

// 64 bins per axis: with 16 a split plane of a large node can only fall every 1/16 of its centroid extent, which on the
// San-Miguel-like scenes is two rows of shrubs or half an arcade -- priced with the oracle's walker of the product's
// arrays (tools/tree_cost.py, cache-line visits per ray summed over camera, two bounce and occlusion ray sets):
// 16 -> 32 -> 64 bins: C4 (world tree) 134.0 -> 121.8 -> 116.8, C4F 124.2 -> 119.7 -> 118.7, C2 and C3 unchanged
// (+-0.5 %); 128 / 256 bins are no better (122.4 / 120.1: greedy top-down SAH is noisy at that level); build time +10 %.
#ifndef CRT_BVH_BINS
#define CRT_BVH_BINS 64
#endif
#ifndef CRT_BVH_SMALL_RANGE
#define CRT_BVH_SMALL_RANGE 12
#endif
constexpr int MAX_BINS = CRT_BVH_BINS; // per axis; a builder may use fewer (Builder::n_bins)
// SAH: cost of fetching+testing one node relative to one triangle (CRT_BVH_NODE_COST overrides, tuning)
static const float NODE_COST = std::getenv("CRT_BVH_NODE_COST") ? (float)std::atof(std::getenv("CRT_BVH_NODE_COST")) : 1.0f;
constexpr size_t PARALLEL_MIN = 1 << 15;
constexpr uint32_t SMALL_RANGE = CRT_BVH_SMALL_RANGE; // ranges up to this size get an exact sorted SAH sweep

inline void box_reset(Aabb &b)
{
    const float inf = std::numeric_limits<float>::infinity();
    for (int k = 0; k < 3; ++k) {
        b.lo[k] = inf;
        b.hi[k] = -inf;
    }
}
inline void box_grow(Aabb &b, const Aabb &o)
{
    for (int k = 0; k < 3; ++k) {
        b.lo[k] = std::min(b.lo[k], o.lo[k]);
        b.hi[k] = std::max(b.hi[k], o.hi[k]);
    }
}
inline float half_area(const Aabb &b)
{
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

struct TNode { // no default member initialisers: the node pool is raw storage until Builder::alloc() hands a node out
    Aabb box;
    int32_t left, right; // temp-node indices; -1 = leaf
    uint32_t first, count;
    uint32_t depth;
    // Collapse to 4-wide nodes (see build_bvh): cheapest summed surface area of the wide nodes below
    // this node if it may use 1 (= it is the root of a wide node), 2 or 3 child slots of its
    // parent's wide node. Filled when both children are complete, i.e. inside the parallel build.
    double slot_cost[3];
};

// One item as the builder moves it around: 32 B, partitioned IN PLACE so every pass streams
// through contiguous memory (an index indirection here made a 10 M-triangle build 18 s).
struct Prim {
    float lo[3], hi[3];
    uint32_t id;
    uint32_t pad;
};

struct Bins { // n_bins bins on each of the 3 axes, filled in one pass over the items
    Aabb box[3][MAX_BINS];
    uint32_t cnt[3][MAX_BINS];
    Aabb bounds, cbounds;
    int n_bins = MAX_BINS;
    void reset()
    {
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < n_bins; ++b) {
                box_reset(box[a][b]);
                cnt[a][b] = 0;
            }
        }
    }
    void merge(const Bins &o)
    {
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < n_bins; ++b) {
                box_grow(box[a][b], o.box[a][b]);
                cnt[a][b] += o.cnt[a][b];
            }
        }
    }
};

struct Builder {
    std::unique_ptr<Prim[]> prims; // raw storage, filled by several threads (build_bvh)
    // 2n nodes of raw storage: pages are first touched by whichever build thread allocates a node there (a
    // value-initialised vector made one thread write 1.4 GB before the 10 M-triangle build could start)
    std::unique_ptr<TNode[]> tn;
    std::atomic<int32_t> next{0};
    std::atomic<int> spare_threads{0};
    int max_leaf;
    int n_threads = 1;
    int n_bins = MAX_BINS;

    int32_t alloc()
    {
        const int32_t i = next.fetch_add(1);
        TNode &t = tn[i];
        t.left = t.right = -1;
        t.first = t.count = t.depth = 0;
        t.slot_cost[0] = t.slot_cost[1] = t.slot_cost[2] = 0.0;
        return i;
    }

    static inline float centroid(const Prim &p, int a) { return 0.5f * (p.lo[a] + p.hi[a]); }

    // bounds of the items and of their centroids over [first, first + count), in parallel for big ranges
    void range_bounds(uint32_t first, uint32_t count, Aabb &nb, Aabb &cb, int threads)
    {
        auto scan = [&](uint32_t lo, uint32_t hi, Aabb &b, Aabb &c) {
            box_reset(b);
            box_reset(c);
            for (uint32_t i = lo; i < hi; ++i) {
                const Prim &p = prims[i];
                for (int k = 0; k < 3; ++k) {
                    b.lo[k] = std::min(b.lo[k], p.lo[k]);
                    b.hi[k] = std::max(b.hi[k], p.hi[k]);
                    const float ck = centroid(p, k);
                    c.lo[k] = std::min(c.lo[k], ck);
                    c.hi[k] = std::max(c.hi[k], ck);
                }
            }
        };
        if (threads <= 1) {
            scan(first, first + count, nb, cb);
            return;
        }
        std::vector<Aabb> bs(threads), cs(threads);
        std::vector<std::future<void>> jobs;
        for (int t = 0; t < threads; ++t) {
            const uint32_t lo = first + (uint32_t)((uint64_t)count * t / threads);
            const uint32_t hi = first + (uint32_t)((uint64_t)count * (t + 1) / threads);
            jobs.push_back(std::async(std::launch::async, [&, lo, hi, t]() { scan(lo, hi, bs[t], cs[t]); }));
        }
        box_reset(nb);
        box_reset(cb);
        for (int t = 0; t < threads; ++t) {
            jobs[t].get();
            box_grow(nb, bs[t]);
            box_grow(cb, cs[t]);
        }
    }

    void fill_bins(uint32_t first, uint32_t count, const Aabb &cb, const float scale[3], Bins &bins, int threads)
    {
        const int N_BINS = n_bins;
        bins.n_bins = N_BINS;
        auto scan = [&](uint32_t lo, uint32_t hi, Bins &out) {
            out.n_bins = N_BINS;
            out.reset();
            for (uint32_t i = lo; i < hi; ++i) {
                const Prim &p = prims[i];
                Aabb pb;
                for (int k = 0; k < 3; ++k) {
                    pb.lo[k] = p.lo[k];
                    pb.hi[k] = p.hi[k];
                }
                for (int a = 0; a < 3; ++a) {
                    if (scale[a] > 0.f) {
                        int b = (int)((centroid(p, a) - cb.lo[a]) * scale[a]);
                        b = std::min(std::max(b, 0), N_BINS - 1);
                        box_grow(out.box[a][b], pb);
                        ++out.cnt[a][b];
                    }
                }
            }
        };
        if (threads <= 1) {
            scan(first, first + count, bins);
            return;
        }
        std::vector<Bins> part(threads);
        std::vector<std::future<void>> jobs;
        for (int t = 0; t < threads; ++t) {
            const uint32_t lo = first + (uint32_t)((uint64_t)count * t / threads);
            const uint32_t hi = first + (uint32_t)((uint64_t)count * (t + 1) / threads);
            jobs.push_back(std::async(std::launch::async, [&, lo, hi, t]() { scan(lo, hi, part[t]); }));
        }
        bins.reset();
        for (int t = 0; t < threads; ++t) {
            jobs[t].get();
            bins.merge(part[t]);
        }
    }

    int32_t build(uint32_t first, uint32_t count, uint32_t depth)
    {
        const int N_BINS = n_bins;
        const int32_t me = alloc();
        tn[me].first = first;
        tn[me].count = count;
        tn[me].depth = depth;
        // big ranges near the root are scanned by several threads; below that, subtrees run in parallel
        const int scan_threads = count >= (1u << 20) ? std::max(1, std::min(n_threads, spare_threads.load() + 1)) : 1;
        Aabb nb, cb;
        range_bounds(first, count, nb, cb, scan_threads);
        tn[me].box = nb;
        if (count == 1) {
            return me;
        }
        float best_cost = std::numeric_limits<float>::infinity();
        int best_axis = -1, best_bin = -1;
        uint32_t exact_mid = 0; // small ranges: split position found by the exact sweep
        float scale[3] = {0.f, 0.f, 0.f};
        if (count <= SMALL_RANGE) {
            // Exact SAH for small ranges (half of all nodes): sort the few items along each axis and
            // evaluate every split. Cheaper than resetting and sweeping 3 x N_BINS bins, and better.
            Prim tmp[SMALL_RANGE], best_order[SMALL_RANGE];
            for (int axis = 0; axis < 3; ++axis) {
                if (!(cb.hi[axis] - cb.lo[axis] > 0.f)) {
                    continue;
                }
                for (uint32_t i = 0; i < count; ++i) { // insertion sort by (centroid, id)
                    Prim p = prims[first + i];
                    uint32_t j = i;
                    while (j > 0 && (centroid(tmp[j - 1], axis) > centroid(p, axis) ||
                                     (centroid(tmp[j - 1], axis) == centroid(p, axis) && tmp[j - 1].id > p.id))) {
                        tmp[j] = tmp[j - 1];
                        --j;
                    }
                    tmp[j] = p;
                }
                float r_area[SMALL_RANGE];
                Aabb acc;
                box_reset(acc);
                for (uint32_t i = count - 1; i > 0; --i) {
                    Aabb pb;
                    for (int k = 0; k < 3; ++k) {
                        pb.lo[k] = tmp[i].lo[k];
                        pb.hi[k] = tmp[i].hi[k];
                    }
                    box_grow(acc, pb);
                    r_area[i] = half_area(acc);
                }
                box_reset(acc);
                bool improved = false;
                for (uint32_t i = 0; i + 1 < count; ++i) {
                    Aabb pb;
                    for (int k = 0; k < 3; ++k) {
                        pb.lo[k] = tmp[i].lo[k];
                        pb.hi[k] = tmp[i].hi[k];
                    }
                    box_grow(acc, pb);
                    const float cost = half_area(acc) * (float)(i + 1) + r_area[i + 1] * (float)(count - i - 1);
                    if (cost < best_cost) {
                        best_cost = cost;
                        best_axis = axis;
                        exact_mid = first + i + 1;
                        improved = true;
                    }
                }
                if (improved) {
                    for (uint32_t i = 0; i < count; ++i) {
                        best_order[i] = tmp[i];
                    }
                }
            }
            if (best_axis >= 0) {
                for (uint32_t i = 0; i < count; ++i) {
                    prims[first + i] = best_order[i];
                }
            }
        } else {
        // binned SAH over all three axes in one pass
        for (int a = 0; a < 3; ++a) {
            const float cext = cb.hi[a] - cb.lo[a];
            scale[a] = cext > 0.f ? N_BINS / cext : 0.f;
        }
        Bins bins_storage; // ~1.4 KB of stack per level (a heap allocation per node made the threads
        Bins *bins = &bins_storage; // fight over malloc)
        fill_bins(first, count, cb, scale, *bins, scan_threads);
        for (int axis = 0; axis < 3; ++axis) {
            if (!(scale[axis] > 0.f)) {
                continue;
            }
            float r_area[MAX_BINS];
            uint32_t r_cnt[MAX_BINS];
            Aabb acc;
            box_reset(acc);
            uint32_t cnt = 0;
            for (int b = N_BINS - 1; b > 0; --b) {
                box_grow(acc, bins->box[axis][b]);
                cnt += bins->cnt[axis][b];
                r_area[b] = cnt ? half_area(acc) : 0.f;
                r_cnt[b] = cnt;
            }
            box_reset(acc);
            cnt = 0;
            for (int b = 0; b < N_BINS - 1; ++b) {
                box_grow(acc, bins->box[axis][b]);
                cnt += bins->cnt[axis][b];
                if (cnt == 0 || r_cnt[b + 1] == 0) {
                    continue;
                }
                const float cost = half_area(acc) * cnt + r_area[b + 1] * r_cnt[b + 1];
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = axis;
                    best_bin = b;
                }
            }
        }
        }
        const float area = half_area(nb);
        if (count <= (uint32_t)max_leaf) {
            // SAH termination: leaf if testing all items is no dearer than one more level
            const float leaf_cost = (float)count * area;
            if (best_axis < 0 || leaf_cost <= NODE_COST * area + best_cost) {
                return me;
            }
        }
        uint32_t mid = first;
        if (best_axis >= 0 && count <= SMALL_RANGE) {
            mid = exact_mid; // items already reordered along the winning axis
        } else if (best_axis >= 0) {
            const int axis = best_axis;
            const float lo = cb.lo[axis], sc = scale[axis];
            auto it = std::partition(prims.get() + first, prims.get() + first + count, [&](const Prim &p) {
                int b = (int)((centroid(p, axis) - lo) * sc);
                b = std::min(std::max(b, 0), N_BINS - 1);
                return b <= best_bin;
            });
            mid = (uint32_t)(it - prims.get());
        }
        if (mid == first || mid == first + count) {
            // fall back to an object-median split along the widest box axis
            int axis = 0;
            float ext = nb.hi[0] - nb.lo[0];
            for (int k = 1; k < 3; ++k) {
                if (nb.hi[k] - nb.lo[k] > ext) {
                    ext = nb.hi[k] - nb.lo[k];
                    axis = k;
                }
            }
            mid = first + count / 2;
            std::nth_element(prims.get() + first, prims.get() + mid, prims.get() + first + count,
                             [&](const Prim &a, const Prim &b) {
                                 const float ca = centroid(a, axis), cb2 = centroid(b, axis);
                                 return ca != cb2 ? ca < cb2 : a.id < b.id;
                             });
        }
        const uint32_t lc = mid - first, rc = count - lc;
        int32_t l, r;
        if (count >= PARALLEL_MIN && spare_threads.fetch_sub(1) > 0) {
            auto fut = std::async(std::launch::async, [this, first, lc, depth]() {
                const int32_t res = build(first, lc, depth + 1);
                spare_threads.fetch_add(1);
                return res;
            });
            r = build(mid, rc, depth + 1);
            l = fut.get();
        } else {
            if (count >= PARALLEL_MIN) {
                spare_threads.fetch_add(1);
            }
            l = build(first, lc, depth + 1);
            r = build(mid, rc, depth + 1);
        }
        tn[me].left = l;
        tn[me].right = r;
        fill_slot_cost(me);
        return me;
    }
    void fill_slot_cost(int32_t me) // both children complete
    {
        const Aabb &x = tn[me].box;
        const double dx = (double)x.hi[0] - x.lo[0], dy = (double)x.hi[1] - x.lo[1], dz = (double)x.hi[2] - x.lo[2];
        int a;
        const double as_root = (dx * dy + dy * dz + dz * dx) + best_split(me, 4, a);
        tn[me].slot_cost[0] = as_root;
        tn[me].slot_cost[1] = std::min(as_root, best_split(me, 2, a));
        tn[me].slot_cost[2] = std::min(as_root, best_split(me, 3, a));
    }
    // slots(t, j): see TNode::slot_cost; a leaf costs nothing here (its triangles are tested either way)
    double slots(int32_t t, int j) const { return tn[t].left >= 0 ? tn[t].slot_cost[j - 1] : 0.0; }
    // the cheapest split of j >= 2 slots between the children of node t: cost, and the left child's share
    double best_split(int32_t t, int j, int &a_out) const
    {
        const int32_t l = tn[t].left, r = tn[t].right;
        double best = std::numeric_limits<double>::infinity();
        for (int a = 1; a < j; ++a) {
            const double c = slots(l, std::min(a, 3)) + slots(r, std::min(j - a, 3));
            if (c < best) {
                best = c;
                a_out = a;
            }
        }
        return best;
    }
};

// ---- insertion-based optimisation of the binary tree (CRT_BVH_REINSERT=<passes>, default 2) ----------------------------
// A top-down build decides every split with the items of one range in hand and never revisits it. Bittner, Hapala,
// Havran 2013 ("Fast insertion-based optimization of bounding volume hierarchies") repair that afterwards: take a subtree
// out (its parent goes with it, the sibling moves up), find the place where putting it back costs the least summed
// surface area -- the direct cost area(x U y) of the new parent plus the growth of every ancestor of y -- by a
// branch-and-bound search from the root, and put it there, reusing the parent node. Leaves keep their item ranges, so
// nothing below changes. Serial and deterministic (ties by node index). Role in the reference: the quality of
// rtcCommitScene's tree (embree_utils.cpp:63-76); priced like every builder change by tools/tree_cost.py.
struct ReinsertTree { // shared by the workers of a pass, which own disjoint sets of nodes
    TNode *tn;
    std::vector<int32_t> parent;
    std::vector<float> area;
    void index(int32_t n_nodes)
    {
        parent.assign(n_nodes, -1);
        area.assign(n_nodes, 0.f);
        for (int32_t t = 0; t < n_nodes; ++t) {
            area[t] = half_area(tn[t].box);
            if (tn[t].left >= 0) {
                parent[tn[t].left] = t;
                parent[tn[t].right] = t;
            }
        }
    }
};

// works on the subtree under `root` (whose parent link is -1 while it does): searches start there, refits end there
struct Reinserter {
    ReinsertTree &T;
    int32_t root;
    struct Cand {
        float induced;
        int32_t node;
    };
    std::vector<Cand> heap;
    Reinserter(ReinsertTree &tree, int32_t r) : T(tree), root(r) {}

    static Aabb merged(const Aabb &a, const Aabb &b)
    {
        Aabb o = a;
        box_grow(o, b);
        return o;
    }
    void refit_up(int32_t t)
    {
        TNode *tn = T.tn;
        for (; t >= 0; t = T.parent[t]) {
            const Aabb nb = merged(tn[tn[t].left].box, tn[tn[t].right].box);
            if (std::memcmp(&nb, &tn[t].box, sizeof(Aabb)) == 0) {
                break;
            }
            tn[t].box = nb;
            T.area[t] = half_area(nb);
        }
    }
    static bool worse(const Cand &a, const Cand &b) { return a.induced != b.induced ? a.induced > b.induced : a.node > b.node; }
    int32_t best_place(int32_t x)
    {
        const TNode *tn = T.tn;
        const Aabb &xb = tn[x].box;
        const float ax = T.area[x];
        float best_cost = std::numeric_limits<float>::infinity();
        int32_t best = root;
        heap.clear();
        heap.push_back(Cand{0.f, root});
        while (!heap.empty()) {
            std::pop_heap(heap.begin(), heap.end(), worse);
            const Cand c = heap.back();
            heap.pop_back();
            if (c.induced + ax >= best_cost) {
                break; // every remaining candidate has at least this much induced cost
            }
            const float direct = half_area(merged(tn[c.node].box, xb));
            const float total = c.induced + direct;
            if (total < best_cost || (total == best_cost && c.node < best)) {
                best_cost = total;
                best = c.node;
            }
            const float below = total - T.area[c.node]; // what every place under this node pays for growing it
            if (tn[c.node].left >= 0 && below + ax < best_cost) {
                heap.push_back(Cand{below, tn[c.node].left});
                std::push_heap(heap.begin(), heap.end(), worse);
                heap.push_back(Cand{below, tn[c.node].right});
                std::push_heap(heap.begin(), heap.end(), worse);
            }
        }
        return best;
    }
    // returns whether the subtree moved
    bool reinsert(int32_t x)
    {
        TNode *tn = T.tn;
        std::vector<int32_t> &parent = T.parent;
        const int32_t p = parent[x];
        if (p < 0 || parent[p] < 0) {
            return false; // the root and its children stay
        }
        const int32_t g = parent[p];
        const int32_t s = tn[p].left == x ? tn[p].right : tn[p].left;
        (tn[g].left == p ? tn[g].left : tn[g].right) = s;
        parent[s] = g;
        refit_up(g);
        const int32_t y = best_place(x);
        const int32_t py = parent[y];
        tn[p].left = y;
        tn[p].right = x;
        parent[y] = p;
        parent[x] = p;
        parent[p] = py;
        tn[p].box = merged(tn[y].box, tn[x].box);
        T.area[p] = half_area(tn[p].box);
        if (py < 0) {
            root = p;
        } else {
            (tn[py].left == y ? tn[py].left : tn[py].right) = p;
            refit_up(py);
        }
        return y != s;
    }
};

// `passes` sweeps over all subtrees, largest first; then the collapse table (TNode::slot_cost) is rebuilt bottom-up.
// A pass is cut into independent pieces: the tree is split at the highest nodes that have at most REINSERT_PIECE
// nodes below them (a property of the tree, not of the thread count, so the result does not depend on it), every piece
// (2^17) is swept by one worker with searches confined to the piece, and the few nodes above the cut -- and the pieces' own
// roots -- are then swept serially with searches over the whole tree. 10 M nodes: 60 s per pass serially, ~10 s on 8 cores.
int32_t optimise_by_reinsertion(Builder &b, int32_t root, int32_t n_nodes, int passes, bool verbose)
{
    // (CRT_BVH_REINSERT_PIECE: the tests cut small trees into many pieces with it)
    const int32_t REINSERT_PIECE = std::getenv("CRT_BVH_REINSERT_PIECE") ? std::max(2, std::atoi(std::getenv("CRT_BVH_REINSERT_PIECE"))) : 1 << 17;
    ReinsertTree T;
    T.tn = b.tn.get();
    T.index(n_nodes);
    TNode *tn = T.tn;
    auto inner_area = [&]() {
        double sum = 0.0;
        for (int32_t t = 0; t < n_nodes; ++t) {
            sum += tn[t].left >= 0 ? (double)T.area[t] : 0.0;
        }
        return sum;
    };
    auto by_area = [&](int32_t x, int32_t y) { return T.area[x] != T.area[y] ? T.area[x] > T.area[y] : x < y; };
    std::vector<int32_t> size(n_nodes), order;
    for (int pass = 0; pass < passes; ++pass) {
        const double before = verbose ? inner_area() : 0.0;
        // nodes under every node (children after their parents in `order`, so backwards is bottom-up)
        order.clear();
        order.push_back(root);
        for (size_t i = 0; i < order.size(); ++i) {
            const int32_t t = order[i];
            if (tn[t].left >= 0) {
                order.push_back(tn[t].left);
                order.push_back(tn[t].right);
            }
        }
        for (size_t i = order.size(); i-- > 0;) {
            const int32_t t = order[i];
            size[t] = tn[t].left >= 0 ? 1 + size[tn[t].left] + size[tn[t].right] : 1;
        }
        std::vector<int32_t> pieces, above;
        for (int32_t t : order) {
            const int32_t p = T.parent[t];
            if (size[t] > REINSERT_PIECE) {
                above.push_back(t);
            } else if (p < 0 || size[p] > REINSERT_PIECE) {
                pieces.push_back(t);
            }
        }
        std::vector<int32_t> piece_parent(pieces.size()), piece_root(pieces.size());
        std::vector<size_t> piece_moved(pieces.size(), 0);
        for (size_t k = 0; k < pieces.size(); ++k) {
            piece_parent[k] = T.parent[pieces[k]];
            T.parent[pieces[k]] = -1;
        }
        std::atomic<size_t> next_piece{0};
        auto worker = [&]() {
            std::vector<int32_t> mine;
            for (size_t k = next_piece.fetch_add(1); k < pieces.size(); k = next_piece.fetch_add(1)) {
                Reinserter r(T, pieces[k]);
                mine.clear();
                mine.push_back(pieces[k]);
                for (size_t i = 0; i < mine.size(); ++i) {
                    const int32_t t = mine[i];
                    if (tn[t].left >= 0) {
                        mine.push_back(tn[t].left);
                        mine.push_back(tn[t].right);
                    }
                }
                std::sort(mine.begin(), mine.end(), by_area);
                for (int32_t x : mine) {
                    piece_moved[k] += r.reinsert(x) ? 1 : 0;
                }
                piece_root[k] = r.root;
            }
        };
        {
            std::vector<std::future<void>> jobs;
            for (int t = 1; t < std::min<int>(b.n_threads, (int)pieces.size()); ++t) {
                jobs.push_back(std::async(std::launch::async, worker));
            }
            worker();
            for (auto &j : jobs) {
                j.get();
            }
        }
        size_t moved = 0;
        Reinserter top(T, root);
        for (size_t k = 0; k < pieces.size(); ++k) { // hang the pieces back in (their roots may have changed)
            moved += piece_moved[k];
            const int32_t p = piece_parent[k];
            T.parent[piece_root[k]] = p;
            if (p < 0) {
                top.root = piece_root[k];
            } else {
                (tn[p].left == pieces[k] ? tn[p].left : tn[p].right) = piece_root[k];
            }
        }
        for (size_t k = 0; k < pieces.size(); ++k) {
            if (piece_parent[k] >= 0) {
                top.refit_up(piece_parent[k]);
            }
        }
        // the nodes above the cut and the pieces' roots, over the whole tree
        above.insert(above.end(), piece_root.begin(), piece_root.end());
        std::sort(above.begin(), above.end(), by_area);
        for (int32_t x : above) {
            moved += top.reinsert(x) ? 1 : 0;
        }
        root = top.root;
        if (verbose) {
            const double after = inner_area();
            std::fprintf(stderr, "[crt_hip]   reinsertion pass %d: %zu of %d subtrees moved (%zu pieces), summed inner area %.4g -> %.4g (%.1f %%)\n",
                         pass, moved, n_nodes, pieces.size(), before, after, 100.0 * (after / before - 1.0));
        }
        if (moved == 0) {
            break;
        }
    }
    std::vector<int32_t> stack{root}, post;
    post.reserve(n_nodes / 2 + 1);
    while (!stack.empty()) {
        const int32_t t = stack.back();
        stack.pop_back();
        if (b.tn[t].left >= 0) {
            post.push_back(t);
            stack.push_back(b.tn[t].left);
            stack.push_back(b.tn[t].right);
        }
    }
    for (size_t i = post.size(); i-- > 0;) { // parents come before their children in `post`
        b.fill_slot_cost(post[i]);
    }
    return root;
}

inline int32_t leaf_ref(uint32_t first, uint32_t count) { return (int32_t)~((first << 3) | (count - 1u)); }

} // namespace

BuiltBvh build_bvh(const Aabb *boxes, size_t n, int max_leaf, int32_t node_base, uint32_t item_base,
                   bool leaf_holds_item_id, uint32_t max_top_nodes, int n_threads, int reinsert_passes)
{
    if (n == 0) {
        throw std::runtime_error("build_bvh: no items");
    }
    if (n >= (1u << 28)) {
        throw std::runtime_error("build_bvh: too many items for the 28-bit leaf reference");
    }
    if (max_leaf < 1 || max_leaf > 8 || (leaf_holds_item_id && max_leaf != 1)) {
        throw std::runtime_error("build_bvh: bad max_leaf");
    }
    const bool dbg = std::getenv("CRT_HIP_DEBUG") != nullptr;
    auto t_prev = std::chrono::high_resolution_clock::now();
    auto phase = [&](const char *what) {
        const auto now = std::chrono::high_resolution_clock::now();
        if (dbg && n > 100000) {
            std::fprintf(stderr, "[crt_hip]   build_bvh %-18s %8.1f ms\n", what,
                         std::chrono::duration<double, std::milli>(now - t_prev).count());
        }
        t_prev = now;
    };
    Builder b;
    b.max_leaf = max_leaf;
    b.n_threads = std::max(1, n_threads);
    // (the top-level tree over instance boxes keeps 16 bins: with 64 the two-level C4 walks 13 % MORE lines per ray and
    // renders 8 % slower -- greedy SAH over a few thousand overlapping boxes is that fickle; the finer bins pay for
    // triangles, see CRT_BVH_BINS)
    b.n_bins = leaf_holds_item_id ? std::min(16, MAX_BINS) : MAX_BINS;
    b.prims.reset(new Prim[n]);
    parallel_for(n, b.n_threads, 1u << 16, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            Prim &p = b.prims[i];
            for (int k = 0; k < 3; ++k) {
                p.lo[k] = boxes[i].lo[k];
                p.hi[k] = boxes[i].hi[k];
            }
            p.id = (uint32_t)i;
            p.pad = 0;
        }
    });
    b.tn.reset(new TNode[2 * n]);
    b.spare_threads = std::max(0, n_threads - 1);
    phase("setup");
    int32_t root = b.build(0, (uint32_t)n, 0);
    phase("recursive build");
    {
        // two passes by default (round 5; C4 -8 % line visits per ray, 54.6 -> 53.0 ms); CRT_BVH_REINSERT=0 switches it off
        const char *e = std::getenv("CRT_BVH_REINSERT");
        const int passes = reinsert_passes >= 0 ? reinsert_passes : e != nullptr ? std::atoi(e) : 2;
        const int32_t n_nodes = b.next.load();
        if (passes > 0 && n_nodes > 7) {
            root = optimise_by_reinsertion(b, root, n_nodes, passes, dbg && n > 100000);
            phase("reinsertion");
        }
    }

    BuiltBvh out;
    out.bounds = b.tn[root].box;
    out.order.resize(n);
    parallel_for(n, b.n_threads, 1u << 16, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            out.order[i] = b.prims[i].id;
        }
    });
    const int32_t n_tn = b.next.load();
    auto is_inner = [&](int32_t t) { return b.tn[t].left >= 0; };
    auto half_area = [&](int32_t t) {
        const Aabb &x = b.tn[t].box;
        const double dx = (double)x.hi[0] - x.lo[0], dy = (double)x.hi[1] - x.lo[1], dz = (double)x.hi[2] - x.lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    // Collapse the binary tree to BVH_WIDTH-wide nodes. A traversal step costs the same whatever the
    // number of used child slots, so the expected cost of a ray is proportional to the summed
    // surface area of the binary nodes that become roots of wide nodes; every other inner node is
    // absorbed into its parent's wide node. That sum is minimised exactly by dynamic programming
    // over the binary tree (Ylitie, Karras, Laine 2017, section 3.1, for width 4):
    //   root_cost[n] = area(n) + min over a + b = 4 of slots(left, a) + slots(right, b)
    //   slots(n, j)  = cheapest way to hand subtree n at most j child slots of its parent's node:
    //                  a leaf costs 0; an inner node either takes one slot as the root of its own
    //                  wide node (root_cost[n]) or, if j >= 2, is absorbed and splits its j slots
    //                  between its children.
    // CRT_BVH_COLLAPSE=greedy selects the earlier heuristic (expand the inner child of largest
    // surface area until the node is full) for comparison.
    struct Wide {
        int32_t kid[BVH_WIDTH];
        int n;
    };
    static const bool greedy_collapse = [] {
        const char *e = std::getenv("CRT_BVH_COLLAPSE");
        return e != nullptr && std::strcmp(e, "greedy") == 0;
    }();
    static_assert(BVH_WIDTH == 4, "the collapse below is written for 4-wide nodes");
    // The table slots(n, j) is TNode::slot_cost, filled bottom-up inside the (parallel) build.
    auto slots = [&](int32_t t, int j) { return b.slots(t, j); };
    auto best_split = [&](int32_t t, int j, int &a_out) { return b.best_split(t, j, a_out); };
    if (is_inner(root)) {
        out.collapse_cost = (float)(b.tn[root].slot_cost[0] / std::max(half_area(root), 1e-300));
        if (dbg && n > 100000) {
            double leaf_area = 0.0; // expected triangle tests: summed area of the leaves x their triangle count
            for (int32_t t = 0; t < n_tn; ++t) {
                if (!is_inner(t)) {
                    leaf_area += half_area(t) * (double)b.tn[t].count;
                }
            }
            std::fprintf(stderr, "[crt_hip]   build_bvh SAH: %.2f node visits + %.2f triangle tests per ray that enters the root box (%d bins)\n",
                         (double)out.collapse_cost, leaf_area / std::max(half_area(root), 1e-300), b.n_bins);
        }
    }
    auto wide_children = [&](int32_t t) {
        Wide w;
        w.n = 0;
        if (greedy_collapse) {
            w.kid[0] = b.tn[t].left;
            w.kid[1] = b.tn[t].right;
            w.n = 2;
            while (w.n < BVH_WIDTH) {
                int best = -1;
                double best_area = -1.0;
                for (int k = 0; k < w.n; ++k) {
                    if (is_inner(w.kid[k])) {
                        const double a = half_area(w.kid[k]);
                        if (a > best_area) {
                            best_area = a;
                            best = k;
                        }
                    }
                }
                if (best < 0) {
                    break;
                }
                const int32_t x = w.kid[best];
                w.kid[best] = b.tn[x].left; // keeps the expanded child's position, sibling goes last
                w.kid[w.n++] = b.tn[x].right;
            }
            return w;
        }
        // hand `j` slots to subtree x: explicit stack instead of recursion (at most 3 absorbed nodes)
        struct Item {
            int32_t x;
            int j;
        };
        Item work[8];
        int nw = 0;
        int a = 2;
        best_split(t, 4, a);
        work[nw++] = Item{b.tn[t].right, 4 - a};
        work[nw++] = Item{b.tn[t].left, a};
        while (nw > 0) {
            const Item it = work[--nw];
            bool own_slot = !is_inner(it.x) || it.j == 1;
            int sa = 1;
            if (!own_slot) {
                const double split = best_split(it.x, it.j, sa);
                own_slot = slots(it.x, 1) <= split; // on a tie the subtree keeps its own node (fewer, fuller parents)
            }
            if (own_slot) {
                w.kid[w.n++] = it.x;
            } else {
                work[nw++] = Item{b.tn[it.x].right, it.j - sa};
                work[nw++] = Item{b.tn[it.x].left, sa};
            }
        }
        return w;
    };
    // final index of every wide node (keyed by the temp node it is rooted at): BFS for the first
    // max_top_nodes, then DFS pre-order per remaining subtree (children of a node end up close to
    // it in memory)
    std::vector<int32_t> final_idx(n_tn, -1);
    std::vector<int32_t> order;  // root temp node of every wide node, in final order
    std::vector<Wide> wides;     // and its children (worked out once, here; the emit pass below reuses them)
    order.reserve(n_tn / 3 + 1);
    wides.reserve(n_tn / 3 + 1);
    uint32_t max_depth = 1;
    if (is_inner(root)) {
        std::queue<std::pair<int32_t, uint32_t>> q;
        q.push({root, 1u});
        while (!q.empty() && order.size() < max_top_nodes) {
            const auto [t, dep] = q.front();
            q.pop();
            final_idx[t] = (int32_t)order.size();
            order.push_back(t);
            max_depth = std::max(max_depth, dep);
            const Wide w = wide_children(t);
            wides.push_back(w);
            for (int k = 0; k < w.n; ++k) {
                if (is_inner(w.kid[k])) {
                    q.push({w.kid[k], dep + 1});
                }
            }
        }
        out.n_top = (uint32_t)order.size();
        std::vector<std::pair<int32_t, uint32_t>> stack;
        while (!q.empty()) {
            stack.push_back(q.front());
            q.pop();
            while (!stack.empty()) {
                const auto [t, dep] = stack.back();
                stack.pop_back();
                final_idx[t] = (int32_t)order.size();
                order.push_back(t);
                max_depth = std::max(max_depth, dep);
                const Wide w = wide_children(t);
                wides.push_back(w);
                for (int k = w.n - 1; k >= 0; --k) {
                    if (is_inner(w.kid[k])) {
                        stack.push_back({w.kid[k], dep + 1});
                    }
                }
            }
        }
    }
    auto child_ref = [&](int32_t t) -> int32_t {
        const TNode &c = b.tn[t];
        if (c.left >= 0) {
            return final_idx[t] + node_base;
        }
        if (leaf_holds_item_id) {
            return leaf_ref(b.prims[c.first].id + item_base, 1);
        }
        return leaf_ref(c.first + item_base, c.count);
    };
    auto set_child = [&](BvhNode &nd, int k, int32_t t) {
        for (int a = 0; a < 3; ++a) {
            nd.lo[k][a] = b.tn[t].box.lo[a];
            nd.hi[k][a] = b.tn[t].box.hi[a];
        }
        nd.c[k] = child_ref(t);
    };
    auto empty_node = [] {
        BvhNode nd;
        std::memset(&nd, 0, sizeof(nd));
        for (int k = 0; k < BVH_WIDTH; ++k) {
            nd.c[k] = EMPTY_CHILD;
        }
        return nd;
    };
    out.max_depth = max_depth; // of the WIDE tree: a traversal stack holds < BVH_WIDTH entries per level
    phase("node order");
    if (!is_inner(root)) {
        // a single leaf: one node with one used slot
        BvhNode nd = empty_node();
        set_child(nd, 0, root);
        out.nodes.push_back(nd);
        out.n_top = 1;
        return out;
    }
    out.nodes.resize(order.size());
    parallel_for(order.size(), b.n_threads, 1u << 14, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            const Wide &w = wides[i];
            BvhNode nd = empty_node();
            for (int k = 0; k < w.n; ++k) {
                set_child(nd, k, w.kid[k]);
            }
            out.nodes[i] = nd;
        }
    });
    phase("emit nodes");
    return out;
}

// The level-by-level collapse of a binary tree (root = node 0) to 4-wide nodes in BFS order, as bvh_device.hip's k_collapse does it.
// first_of: per internal node the first sorted position it covers (leaves of several items; nullptr with max_leaf = 1).
static void collapse_binary_tree(const LbvhTree &tree, const int32_t *first_of, int max_leaf, uint32_t max_top_nodes, BuiltBvh &out)
{
    std::vector<int32_t> frontier{0}, next;
    uint32_t level_base = 0, depth = 0;
    while (!frontier.empty()) {
        next.clear();
        const uint32_t n_in = (uint32_t)frontier.size();
        for (uint32_t i = 0; i < n_in; ++i) {
            int32_t sub[BVH_WIDTH];
            const int nc = lbvh_wide_children(tree, frontier[i], (uint32_t)max_leaf, sub);
            BvhNode nd;
            std::memset(&nd, 0, sizeof(nd));
            for (int c = 0; c < BVH_WIDTH; ++c) {
                nd.c[c] = EMPTY_CHILD;
            }
            for (int c = 0; c < nc; ++c) {
                const uint32_t count = lbvh_count(tree, sub[c]);
                if (count <= (uint32_t)max_leaf) {
                    nd.c[c] = lbvh_leaf_ref(sub[c] >= 0 ? (uint32_t)first_of[sub[c]] : (uint32_t)~sub[c], count);
                } else {
                    nd.c[c] = (int32_t)(level_base + n_in + next.size());
                    next.push_back(sub[c]);
                }
                const Aabb &b = lbvh_box(tree, sub[c]);
                for (int a = 0; a < 3; ++a) {
                    nd.lo[c][a] = b.lo[a];
                    nd.hi[c][a] = b.hi[a];
                }
            }
            out.nodes.push_back(nd);
        }
        level_base += n_in;
        frontier.swap(next);
        ++depth;
    }
    out.max_depth = depth;
    out.n_top = std::min<uint32_t>((uint32_t)out.nodes.size(), max_top_nodes);
}

// ---- linear BVH on the host: the device builder's algorithm (lbvh.h), run serially ---------------------
BuiltBvh build_lbvh_host(const Aabb *boxes, size_t n, int max_leaf, uint32_t max_top_nodes)
{
    BuiltBvh out;
    box_reset(out.bounds);
    for (size_t i = 0; i < n; ++i) {
        box_grow(out.bounds, boxes[i]);
    }
    if (n < 3) { // degenerate sizes: the SAH builder's single-node forms
        return build_bvh(boxes, n, max_leaf, 0, 0, false, max_top_nodes, 1);
    }
    std::vector<uint64_t> keys(n);
    std::vector<Aabb> pbox(n), ibox(n);
    std::vector<int32_t> left(n), right(n), lo(n), hi(n);
    out.order.resize(n);
    // the binary tree with either key normalisation; returns the summed half-area of its internal nodes
    auto build_binary = [&](int mode) -> double {
        std::vector<std::pair<uint64_t, uint32_t>> ki(n);
        for (size_t i = 0; i < n; ++i) {
            ki[i] = {lbvh_key(boxes[i], out.bounds, mode), (uint32_t)i};
        }
        std::stable_sort(ki.begin(), ki.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
        for (size_t i = 0; i < n; ++i) {
            keys[i] = ki[i].first;
            out.order[i] = ki[i].second;
            pbox[i] = boxes[ki[i].second];
        }
        for (size_t i = 0; i + 1 < n; ++i) {
            int a, b;
            lbvh_node(keys.data(), (int)n, (int)i, left[i], right[i], a, b);
            lo[i] = a;
            hi[i] = b;
        }
        // boxes bottom-up: a node covers a narrower key range than its parent, so ascending range size works
        std::vector<uint32_t> by_size(n - 1);
        for (uint32_t i = 0; i + 1 < n; ++i) {
            by_size[i] = i;
        }
        std::sort(by_size.begin(), by_size.end(), [&](uint32_t a, uint32_t b) { return hi[a] - lo[a] < hi[b] - lo[b]; });
        double cost = 0.0;
        for (uint32_t k : by_size) {
            box_reset(ibox[k]);
            box_grow(ibox[k], left[k] >= 0 ? ibox[left[k]] : pbox[~left[k]]);
            box_grow(ibox[k], right[k] >= 0 ? ibox[right[k]] : pbox[~right[k]]);
            cost += (double)lbvh_half_area(ibox[k]);
        }
        return cost;
    };
    const double cost0 = build_binary(0), cost1 = build_binary(1);
    if (cost0 <= cost1) {
        build_binary(0);
    }
    if (std::getenv("CRT_BVH_REPORT")) {
        std::fprintf(stderr, "[crt_hip] Karras tree over %zu items: summed half-area of the binary tree %.6g (keys per axis) / %.6g (cubic keys)\n", n, cost0, cost1);
    }
    const LbvhTree tree{left.data(), right.data(), lo.data(), hi.data(), ibox.data(), pbox.data()};
    collapse_binary_tree(tree, lo.data(), max_leaf, max_top_nodes, out);
    return out;
}

// ---- PLOC on the host: the device builder's other binary tree (lbvh.h "PLOC"), run serially -----------------------------------
// Leaves of one item only (the default; a leaf of several slots needs a contiguous range of sorted positions, which clustering
// by area does not give): with CRT_BVH_MAX_LEAF > 1 the Karras tree is built instead.
uint32_t ploc_radius()
{
    if (const char *e = std::getenv("CRT_PLOC_RADIUS")) {
        const long r = std::atol(e);
        if (r >= 1 && r <= 1024) {
            return (uint32_t)r;
        }
    }
    return PLOC_DEFAULT_RADIUS;
}

BuiltBvh build_ploc_host(const Aabb *boxes, size_t n, int max_leaf, uint32_t max_top_nodes)
{
    const uint32_t radius = ploc_radius();
    if (n < 3 || max_leaf != 1) {
        return build_lbvh_host(boxes, n, max_leaf, max_top_nodes);
    }
    BuiltBvh out;
    box_reset(out.bounds);
    for (size_t i = 0; i < n; ++i) {
        box_grow(out.bounds, boxes[i]);
    }
    std::vector<Aabb> pbox(n), ibox(n);
    std::vector<int32_t> left(n), right(n), lo(n, 0), hi(n, 1); // (lo / hi: every internal node "holds two items", i.e. more than max_leaf = 1)
    std::vector<uint32_t> best_order;
    // clusters in Morton order of either key normalisation; returns the summed half-area of the internal nodes
    auto build_binary = [&](int mode) -> double {
        std::vector<std::pair<uint64_t, uint32_t>> ki(n);
        for (size_t i = 0; i < n; ++i) {
            ki[i] = {lbvh_key(boxes[i], out.bounds, mode), (uint32_t)i};
        }
        std::stable_sort(ki.begin(), ki.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
        out.order.resize(n);
        std::vector<int32_t> cid(n), cid2;
        std::vector<Aabb> cbox(n), cbox2;
        for (size_t i = 0; i < n; ++i) {
            out.order[i] = ki[i].second;
            pbox[i] = boxes[ki[i].second];
            cid[i] = ~(int32_t)i;
            cbox[i] = pbox[i];
        }
        std::vector<uint32_t> nn(n);
        uint32_t m = (uint32_t)n;
        int32_t next_id = (int32_t)n - 1; // ids are handed out downwards: the last merge makes node 0, the root
        double cost = 0.0;
        while (m > 1) {
            for (uint32_t i = 0; i < m; ++i) {
                nn[i] = ploc_nearest(cbox.data(), m, i, radius);
            }
            uint32_t merges = 0;
            for (uint32_t i = 0; i < m; ++i) {
                merges += nn[nn[i]] == i && i < nn[i];
            }
            if (merges == 0) { // (cannot happen with finite boxes -- the pair of smallest area chooses each other; NaN boxes compare false everywhere)
                nn[0] = 1;
                nn[1] = 0;
                merges = 1;
            }
            cid2.clear();
            cbox2.clear();
            uint32_t k = 0;
            for (uint32_t i = 0; i < m; ++i) {
                const uint32_t j = nn[i];
                if (nn[j] == i) {
                    if (i < j) {
                        const int32_t id = next_id - 1 - (int32_t)k; // the k-th merge of this iteration, in array order
                        ++k;
                        left[id] = cid[i];
                        right[id] = cid[j];
                        ibox[id] = ploc_union(cbox[i], cbox[j]);
                        cost += (double)lbvh_half_area(ibox[id]);
                        cid2.push_back(id);
                        cbox2.push_back(ibox[id]);
                    }
                } else {
                    cid2.push_back(cid[i]);
                    cbox2.push_back(cbox[i]);
                }
            }
            next_id -= (int32_t)merges;
            cid.swap(cid2);
            cbox.swap(cbox2);
            m = (uint32_t)cid.size();
        }
        return cost;
    };
    const double cost0 = build_binary(0), cost1 = build_binary(1);
    if (cost0 <= cost1) {
        build_binary(0);
    }
    if (std::getenv("CRT_BVH_REPORT")) {
        std::fprintf(stderr, "[crt_hip] PLOC tree (radius %d) over %zu items: summed half-area of the binary tree %.6g (keys per axis) / %.6g (cubic keys)\n",
                     (int)radius, n, cost0, cost1);
    }
    const LbvhTree tree{left.data(), right.data(), lo.data(), hi.data(), ibox.data(), pbox.data()};
    collapse_binary_tree(tree, nullptr, max_leaf, max_top_nodes, out);
    return out;
}

// Fixed-point frame of a BVH with bounds b: 65531 quanta span the box, 2 quanta of margin on
// either side absorb the outward rounding below.
QFrame make_frame(const Aabb &b)
{
    QFrame f;
    for (int a = 0; a < 3; ++a) {
        const double ext = (double)b.hi[a] - (double)b.lo[a];
        const double floor_step = (std::fabs((double)b.lo[a]) + std::fabs((double)b.hi[a])) * 1e-7 + 1e-30;
        f.step[a] = (float)std::max(ext / 65531.0, floor_step);
        f.base[a] = (float)((double)b.lo[a] - 2.0 * (double)f.step[a]);
    }
    return f;
}

// Outward-rounded 16-bit box: lo one quantum further down than floor(), hi one further up than
// ceil(), so base + q*step (evaluated in fp32 on the device) still brackets the true box.
QNode quantise(const BvhNode &n, const QFrame &f)
{
    QNode q;
    auto lo = [&](float v, int a) {
        const double x = std::floor(((double)v - (double)f.base[a]) / (double)f.step[a]) - 1.0;
        return (uint16_t)std::min(std::max(x, 0.0), 65535.0);
    };
    auto hi = [&](float v, int a) {
        const double x = std::ceil(((double)v - (double)f.base[a]) / (double)f.step[a]) + 1.0;
        return (uint16_t)std::min(std::max(x, 0.0), 65535.0);
    };
    for (int k = 0; k < BVH_WIDTH; ++k) {
        const bool used = n.c[k] != EMPTY_CHILD;
        for (int a = 0; a < 3; ++a) {
            q.child[k].q[a][0] = used ? lo(n.lo[k][a], a) : 65535; // unused: inverted box (crt_types.h)
            q.child[k].q[a][1] = used ? hi(n.hi[k][a], a) : 0;
        }
        q.child[k].ref = used ? n.c[k] : n.c[0]; // slot 0 is always used (used slots come first)
    }
    return q;
}

} // namespace crt
