// bvh_device.h — BLAS construction on the device (bvh_device.hip); see lbvh.h for the algorithm.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/crt_hip.h"
#include "bvh_builder.h"
#include "crt_types.h"

namespace crt {

// One mesh's BLAS as the device built it, copied back to the host: quantised 4-wide nodes in BFS
// order (inner references are indices into `nodes`, leaf references index `slots`), the leaf slots and
// their triangles' vertex UVs (TRI_UV_STRIDE floats per triangle index 2 * slot + which) in leaf order.
struct DeviceBuiltMesh {
    std::vector<QNode> nodes;
    std::vector<LeafSlot> slots;
    std::vector<float> tri_uvs;
    Aabb bounds;
    QFrame frame;
    uint32_t max_depth = 0; // levels of the wide tree
    uint32_t n_top = 0;
};

// Builds the BLAS of the mesh made of geoms[0 .. n_geoms) on HIP device `device`, over the leaf slots geom_slots[g] of
// each geometry (leaf_slots.h: which triangles share a slot, decided on the host). Returns false (and leaves `out`
// alone) for meshes too small to be worth it -- the caller then uses the host builder. Throws std::runtime_error on
// HIP errors. The caller's current device is left as it was.
struct SlotTris;
bool device_build_mesh(int device, const crt_geometry_desc *geoms, uint32_t n_geoms, const std::vector<SlotTris> *geom_slots,
                       uint32_t max_leaf, uint32_t max_top_nodes, DeviceBuiltMesh &out);

// The WORLD TREE of a scene with several instances (crt_types.h LEVELS_WORLD_TREE) built the same way: one item per
// (instance, leaf slot of its mesh), the slot record in the mesh's object space with tag = (instance << 1) | identity, the
// box around the transformed vertices pushed out by inst_pad[instance] (scene_prepare.cpp instance_pad; 0 for an identity
// instance) -- the host loop of build_world_tree, per item on the device. The reference's role: rtcCommitScene of the
// top-level scene (embree_utils.cpp:121-129).
bool device_build_world(int device, const crt_scene_desc *scene, const std::vector<SlotTris> *geom_slots, const uint32_t *inst_identity,
                        const float *inst_pad, uint32_t max_leaf, uint32_t max_top_nodes, DeviceBuiltMesh &out);

// The same algorithm run serially on the host (shares lbvh.h with the kernels): what the CPU tests
// check, and the reference the device result is compared against (CRT_BVH_BUILDER=lbvh).
BuiltBvh build_lbvh_host(const Aabb *boxes, size_t n, int max_leaf, uint32_t max_top_nodes);
// ... and of the device builder's PLOC tree (lbvh.h "PLOC"; CRT_BVH_BUILDER=ploc)
BuiltBvh build_ploc_host(const Aabb *boxes, size_t n, int max_leaf, uint32_t max_top_nodes);
uint32_t ploc_radius(); // the search radius both use: lbvh.h PLOC_DEFAULT_RADIUS, or CRT_PLOC_RADIUS from the environment

} // namespace crt
