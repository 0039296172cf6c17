// bvh_builder.cpp — parallel top-down binned-SAH BVH2 build + cache-aware node layout.
#include "bvh_builder.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <future>
#include <limits>
#include <queue>
#include <stdexcept>

namespace crt {
namespace {

constexpr int N_BINS = 16;
// SAH: cost of fetching+testing one node relative to one triangle (CRT_BVH_NODE_COST overrides, tuning)
static const float NODE_COST = std::getenv("CRT_BVH_NODE_COST") ? (float)std::atof(std::getenv("CRT_BVH_NODE_COST")) : 1.0f;
constexpr size_t PARALLEL_MIN = 1 << 15;

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

struct TNode {
    Aabb box;
    int32_t left = -1, right = -1; // temp-node indices; -1 = leaf
    uint32_t first = 0, count = 0;
    uint32_t depth = 0;
};

struct Builder {
    const Aabb *boxes;
    std::vector<float> cent; // 3 per item
    std::vector<uint32_t> ids;
    std::vector<TNode> tn;
    std::atomic<int32_t> next{0};
    std::atomic<int> spare_threads{0};
    int max_leaf;

    int32_t alloc() { return next.fetch_add(1); }

    int32_t build(uint32_t first, uint32_t count, uint32_t depth)
    {
        const int32_t me = alloc();
        TNode &node = tn[me];
        node.first = first;
        node.count = count;
        node.depth = depth;
        Aabb nb, cb;
        box_reset(nb);
        box_reset(cb);
        for (uint32_t i = first; i < first + count; ++i) {
            const uint32_t id = ids[i];
            box_grow(nb, boxes[id]);
            for (int k = 0; k < 3; ++k) {
                cb.lo[k] = std::min(cb.lo[k], cent[3 * (size_t)id + k]);
                cb.hi[k] = std::max(cb.hi[k], cent[3 * (size_t)id + k]);
            }
        }
        node.box = nb;
        if (count == 1) {
            return me;
        }
        // pick the split: binned SAH over the widest centroid axis first, then the others
        float best_cost = std::numeric_limits<float>::infinity();
        int best_axis = -1, best_bin = -1;
        float best_lo = 0.f, best_scale = 0.f;
        for (int axis = 0; axis < 3; ++axis) {
            const float cmin = cb.lo[axis], cext = cb.hi[axis] - cb.lo[axis];
            if (!(cext > 0.f)) {
                continue;
            }
            const float scale = N_BINS / cext;
            Aabb bb[N_BINS];
            uint32_t bc[N_BINS];
            for (int b = 0; b < N_BINS; ++b) {
                box_reset(bb[b]);
                bc[b] = 0;
            }
            for (uint32_t i = first; i < first + count; ++i) {
                const uint32_t id = ids[i];
                int b = (int)((cent[3 * (size_t)id + axis] - cmin) * scale);
                b = std::min(std::max(b, 0), N_BINS - 1);
                box_grow(bb[b], boxes[id]);
                ++bc[b];
            }
            float r_area[N_BINS];
            uint32_t r_cnt[N_BINS];
            Aabb acc;
            box_reset(acc);
            uint32_t cnt = 0;
            for (int b = N_BINS - 1; b > 0; --b) {
                box_grow(acc, bb[b]);
                cnt += bc[b];
                r_area[b] = cnt ? half_area(acc) : 0.f;
                r_cnt[b] = cnt;
            }
            box_reset(acc);
            cnt = 0;
            for (int b = 0; b < N_BINS - 1; ++b) {
                box_grow(acc, bb[b]);
                cnt += bc[b];
                if (cnt == 0 || r_cnt[b + 1] == 0) {
                    continue;
                }
                const float cost = half_area(acc) * cnt + r_area[b + 1] * r_cnt[b + 1];
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = axis;
                    best_bin = b;
                    best_lo = cmin;
                    best_scale = scale;
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
        uint32_t mid;
        if (best_axis >= 0) {
            const int axis = best_axis;
            auto it = std::partition(ids.begin() + first, ids.begin() + first + count, [&](uint32_t id) {
                int b = (int)((cent[3 * (size_t)id + axis] - best_lo) * best_scale);
                b = std::min(std::max(b, 0), N_BINS - 1);
                return b <= best_bin;
            });
            mid = (uint32_t)(it - ids.begin());
        } else {
            mid = first; // all centroids coincide
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
            std::nth_element(ids.begin() + first, ids.begin() + mid, ids.begin() + first + count,
                             [&](uint32_t a, uint32_t b) {
                                 const float ca = cent[3 * (size_t)a + axis], cb2 = cent[3 * (size_t)b + axis];
                                 return ca != cb2 ? ca < cb2 : a < b;
                             });
        }
        const uint32_t lc = mid - first, rc = count - lc;
        int32_t l, r;
        if (count >= PARALLEL_MIN && spare_threads.fetch_sub(1) > 0) {
            auto fut = std::async(std::launch::async, [&, first, lc, depth]() {
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
        return me;
    }
};

inline int32_t leaf_ref(uint32_t first, uint32_t count) { return (int32_t)~((first << 3) | (count - 1u)); }

} // namespace

BuiltBvh build_bvh(const Aabb *boxes, size_t n, int max_leaf, int32_t node_base, uint32_t item_base,
                   bool leaf_holds_item_id, uint32_t max_top_nodes, int n_threads)
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
    Builder b;
    b.boxes = boxes;
    b.max_leaf = max_leaf;
    b.cent.resize(3 * n);
    b.ids.resize(n);
    for (size_t i = 0; i < n; ++i) {
        b.ids[i] = (uint32_t)i;
        for (int k = 0; k < 3; ++k) {
            b.cent[3 * i + k] = 0.5f * (boxes[i].lo[k] + boxes[i].hi[k]);
        }
    }
    b.tn.resize(2 * n);
    b.spare_threads = std::max(0, n_threads - 1);
    const int32_t root = b.build(0, (uint32_t)n, 0);

    BuiltBvh out;
    out.bounds = b.tn[root].box;
    out.order = b.ids;
    const int32_t n_tn = b.next.load();
    // final index of every inner temp node: BFS for the first max_top_nodes, then DFS pre-order
    // per remaining subtree (children of a node end up close to it in memory)
    std::vector<int32_t> final_idx(n_tn, -1);
    std::vector<int32_t> order; // temp node ids in final order
    order.reserve(n_tn / 2 + 1);
    auto is_inner = [&](int32_t t) { return b.tn[t].left >= 0; };
    std::vector<int32_t> pending;
    if (is_inner(root)) {
        std::queue<int32_t> q;
        q.push(root);
        while (!q.empty() && order.size() < max_top_nodes) {
            const int32_t t = q.front();
            q.pop();
            final_idx[t] = (int32_t)order.size();
            order.push_back(t);
            if (is_inner(b.tn[t].left)) {
                q.push(b.tn[t].left);
            }
            if (is_inner(b.tn[t].right)) {
                q.push(b.tn[t].right);
            }
        }
        out.n_top = (uint32_t)order.size();
        while (!q.empty()) {
            pending.push_back(q.front());
            q.pop();
        }
        std::vector<int32_t> stack;
        for (int32_t sub : pending) {
            stack.push_back(sub);
            while (!stack.empty()) {
                const int32_t t = stack.back();
                stack.pop_back();
                final_idx[t] = (int32_t)order.size();
                order.push_back(t);
                if (is_inner(b.tn[t].right)) {
                    stack.push_back(b.tn[t].right);
                }
                if (is_inner(b.tn[t].left)) {
                    stack.push_back(b.tn[t].left);
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
            return leaf_ref(b.ids[c.first] + item_base, 1);
        }
        return leaf_ref(c.first + item_base, c.count);
    };
    auto set_box = [](float *lo, float *hi, const Aabb &bx) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = bx.lo[k];
            hi[k] = bx.hi[k];
        }
    };
    uint32_t max_depth = 0;
    for (int32_t t = 0; t < n_tn; ++t) {
        max_depth = std::max(max_depth, b.tn[t].depth);
    }
    out.max_depth = max_depth;
    if (!is_inner(root)) {
        // a single leaf: wrap it in one node listing it twice (the repeat loses every tie, so
        // results are unchanged; far-away dummy boxes would not survive box quantisation)
        BvhNode nd;
        std::memset(&nd, 0, sizeof(nd));
        set_box(nd.lo0, nd.hi0, b.tn[root].box);
        set_box(nd.lo1, nd.hi1, b.tn[root].box);
        nd.c0 = nd.c1 = child_ref(root);
        out.nodes.push_back(nd);
        out.n_top = 1;
        return out;
    }
    out.nodes.resize(order.size());
    for (size_t i = 0; i < order.size(); ++i) {
        const TNode &t = b.tn[order[i]];
        BvhNode nd;
        std::memset(&nd, 0, sizeof(nd));
        set_box(nd.lo0, nd.hi0, b.tn[t.left].box);
        set_box(nd.lo1, nd.hi1, b.tn[t.right].box);
        nd.c0 = child_ref(t.left);
        nd.c1 = child_ref(t.right);
        out.nodes[i] = nd;
    }
    return out;
}

} // namespace crt
