// scene_prepare.h — the host half of set_scene and the prepared scene it produces (scene_prepare.cpp).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/crt_hip.h"
#include "crt_types.h"

// What crt_hip_prepare_scene hands out (include/crt_hip.h): the arrays the kernels read, ready for upload.
struct crt_hip_prepared_scene {
    std::vector<crt::PNode> nodes;    // packed 4-wide nodes (crt_types.h PNode)
    std::vector<crt::LeafSlot> slots; // the leaves: one or two triangles each (crt_types.h)
    std::vector<float> tri_uvs;      // TRI_UV_STRIDE per triangle index 2 * slot + which
    std::vector<crt::InstanceRec> insts;
    std::vector<uint32_t> material_ids;
    std::vector<float> materials, lights;
    std::vector<crt::TexRec> tex;
    std::vector<uint8_t> texels;
    crt::QFrame root_frame{};
    int32_t root = 0;
    uint32_t two_level = 0, n_top = 0, n_lights = 0, n_instances = 0, spp = 1, stack_need = 0;
    int32_t world_inst = -1; // instance grafted into the top-level tree (prepare_scene), or -1
    double build_ms = 0.0;
};

namespace crt {

// Validates the scene (throws std::runtime_error on a malformed one: nothing may crash across the C ABI) and fills
// `prepared`. build_device >= 0: meshes large enough to be worth it get their BLAS from the device builder
// (bvh_device.hip) on that HIP device; -1: the host SAH builder for everything.
// reinsert_passes: see build_bvh (< 0: the environment's / default). tree_only: the scene was validated and its textures prepared by
// an earlier call -- only what depends on the TREE is filled (nodes, leaf slots, their uv records, instance records, root, frame,
// stack need); `scene->textures` is not read (crt_core.cpp: the background refinement of CRT_HIP_FLAG_REFINE_IN_BACKGROUND).
void prepare_scene(const crt_scene_desc *scene, crt_hip_prepared_scene *prepared, int n_threads, int build_device = -1, int reinsert_passes = -1,
                   bool tree_only = false);

// Host cores this process may use: affinity mask, capped by the cgroup CPU quota, overridable with CRT_HIP_BUILD_THREADS.
int host_threads();

// The error text crt_hip_last_error(NULL) returns (calls that have no context to keep it in), per thread.
void set_global_error(const std::string &msg);
const std::string &global_error();

} // namespace crt
