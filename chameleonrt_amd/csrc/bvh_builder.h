// bvh_builder.h — host-side binned-SAH builder: a binary SAH tree collapsed to the 4-wide nodes the
// traversal kernels read (crt_types.h). Stands in for Embree's rtcCommitScene
// (reference backends/embree/embree_utils.cpp:63-76, 121-129); static scenes only, so the build
// runs once per set_scene on the host cores and the result is uploaded to HBM.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "crt_types.h"

namespace crt {

struct Aabb {
    float lo[3], hi[3];
};

struct BuiltBvh {
    std::vector<BvhNode> nodes;  // nodes[0] is the root; the first n_top nodes are the top levels in BFS order
    std::vector<uint32_t> order; // item ids in leaf order (leaf `first` indexes this array)
    uint32_t n_top = 0;
    Aabb bounds;
    uint32_t max_depth = 0;      // levels of the wide tree (a single node = 1)
    float collapse_cost = 0.f;   // summed surface area of the wide nodes / area of the root (expected node visits
                                 // of a random ray that hits the root box); 0 if not computed
};

// boxes: one per item. Leaves hold at most max_leaf (<= 8) items.
// node_base / item_base are added to the inner-node indices / leaf `first` values so several
// BVHs can be concatenated into one array. If leaf_holds_item_id is set (TLAS), max_leaf must
// be 1 and a leaf's `first` is the item id itself instead of its position in `order`.
// reinsert_passes: passes of insertion-based re-optimisation after the top-down build (Bittner et al. 2013); < 0 = CRT_BVH_REINSERT
// from the environment, default 2 (0 is what the quick tree of a background-refined scene asks for, crt_hip.h
// CRT_HIP_FLAG_REFINE_IN_BACKGROUND).
BuiltBvh build_bvh(const Aabb *boxes, size_t n, int max_leaf, int32_t node_base, uint32_t item_base,
                   bool leaf_holds_item_id, uint32_t max_top_nodes, int n_threads, int reinsert_passes = -1);

// Fixed-point frame of a BVH with bounds b, and the outward-rounded 64-byte form of a node in it
// (crt_types.h QFrame / QNode): what the traversal kernels actually read.
QFrame make_frame(const Aabb &b);
QNode quantise(const BvhNode &n, const QFrame &f);

} // namespace crt
