// scene_prepare.cpp — the host half of set_scene: everything that does not need a device.
//
// RenderEmbree::set_scene (render_embree.cpp:58-133, embree_utils.cpp:9-136): one BLAS per Mesh, the TLAS over
// the instances, sRGB -> linear textures in 8 bits, material and light tables -- as the flat arrays the kernels
// read (crt_types.h). Split from the upload (crt_core.cpp) so that the GPUs of one node share ONE build:
// RenderHIP::set_scene prepares once and uploads to every context; bench.py's rank 0 prepares, saves to /dev/shm
// and the other ranks load. Nothing here touches a device, except that a BLAS may be built on one
// (bvh_device.hip) when the caller asks for it.
#include "scene_prepare.h"

#include <hip/hip_runtime.h>

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <queue>
#include <stdexcept>
#include <string>
#include <thread>

#include "bvh_builder.h"
#include "bvh_device.h"
#include "host_parallel.h"
#include "kernels.h"
#include "lbvh.h"
#include "leaf_slots.h"
#include "presplit.h"

namespace crt {
namespace {

constexpr int MAX_TOP_NODES_HOST = CRT_MAX_TOP_NODES; // kernels.h
// Scenes with several instances whose triangles fit the budget get a world tree unless CRT_HIP_LEVELS says otherwise.
#ifndef CRT_WORLD_TREE_DEFAULT
#define CRT_WORLD_TREE_DEFAULT 1
#endif

// util/util.cpp:102-108 (std::pow(float, double): evaluated in double)
inline float srgb_to_linear(float x)
{
    if (x <= 0.04045f) {
        return x / 12.92f;
    }
    return (float)std::pow((double)((x + 0.055f) / 1.055f), 2.4);
}

// glm::inverse (embree_utils.cpp:97: world_to_object = inverse(instance.transform)), column-major 4x4.
bool invert4x4(const float m_[16], float out[16])
{
    // GLM 0.9.9.8 (the version the reference's cmake/glm.cmake pins), glm/detail/func_matrix.inl, compute_inverse<4, 4, T, Q,
    // Aligned>, scalar path -- third party, absent from the reference tree and from this image, restated from its published
    // source: eighteen 2x2 sub-determinants, the adjugate's columns as vec4 expressions evaluated left to right
    // ((a * b - c * d) + e * f), the sign pattern, the determinant as dot(column 0 of m, row 0 of the adjugate) summed
    // (x + y) + (z + w) like GLM's compute_dot<vec4>, and one reciprocal multiplied through. m[c][r] = m_[4 * c + r].
    // GLM does not look at the determinant; a zero one is reported here (the reference would go on with infinities).
    #define M(c, r) m_[4 * (c) + (r)]
    const float Coef00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3);
    const float Coef02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3);
    const float Coef03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3);
    const float Coef04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3);
    const float Coef06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3);
    const float Coef07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3);
    const float Coef08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2);
    const float Coef10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2);
    const float Coef11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2);
    const float Coef12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3);
    const float Coef14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3);
    const float Coef15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3);
    const float Coef16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2);
    const float Coef18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2);
    const float Coef19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2);
    const float Coef20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1);
    const float Coef22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1);
    const float Coef23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);
    const float Fac0[4] = {Coef00, Coef00, Coef02, Coef03}, Fac1[4] = {Coef04, Coef04, Coef06, Coef07};
    const float Fac2[4] = {Coef08, Coef08, Coef10, Coef11}, Fac3[4] = {Coef12, Coef12, Coef14, Coef15};
    const float Fac4[4] = {Coef16, Coef16, Coef18, Coef19}, Fac5[4] = {Coef20, Coef20, Coef22, Coef23};
    const float Vec0[4] = {M(1, 0), M(0, 0), M(0, 0), M(0, 0)}, Vec1[4] = {M(1, 1), M(0, 1), M(0, 1), M(0, 1)};
    const float Vec2[4] = {M(1, 2), M(0, 2), M(0, 2), M(0, 2)}, Vec3[4] = {M(1, 3), M(0, 3), M(0, 3), M(0, 3)};
    #undef M
    const float SignA[4] = {+1.f, -1.f, +1.f, -1.f}, SignB[4] = {-1.f, +1.f, -1.f, +1.f};
    float inv[16]; // the adjugate, column c at inv[4 * c ..]
    for (int i = 0; i < 4; ++i) {
        inv[0 + i] = (Vec1[i] * Fac0[i] - Vec2[i] * Fac1[i] + Vec3[i] * Fac2[i]) * SignA[i];
        inv[4 + i] = (Vec0[i] * Fac0[i] - Vec2[i] * Fac3[i] + Vec3[i] * Fac4[i]) * SignB[i];
        inv[8 + i] = (Vec0[i] * Fac1[i] - Vec1[i] * Fac3[i] + Vec3[i] * Fac5[i]) * SignA[i];
        inv[12 + i] = (Vec0[i] * Fac2[i] - Vec1[i] * Fac4[i] + Vec2[i] * Fac5[i]) * SignB[i];
    }
    const float Dot0[4] = {m_[0] * inv[0], m_[1] * inv[4], m_[2] * inv[8], m_[3] * inv[12]};
    const float Dot1 = (Dot0[0] + Dot0[1]) + (Dot0[2] + Dot0[3]);
    if (Dot1 == 0.f) {
        return false;
    }
    const float OneOverDeterminant = 1.f / Dot1;
    for (int i = 0; i < 16; ++i) {
        out[i] = inv[i] * OneOverDeterminant;
    }
    return true;
}

bool is_identity(const float m[16])
{
    static const float id[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    return std::memcmp(m, id, sizeof(id)) == 0;
}

// How far the world box of an instance's transformed vertices (two-level) or of one of its triangles (world tree) is
// pushed out. What lies under the box is intersected with a TRANSFORMED ray, o' = W o, d' = W d in float: a hit found
// in object space at parameter t sits, in world space, at M (o' + t d') = o + t d + M e with |M e| of the order
// eps * cond(M) * (size of the coordinates involved) -- and the walk only gets there if the world-space ray enters the
// box. 1e-5 of the instance's extent covers rotations and moderate scales; the second term keeps a badly conditioned
// transform (an instance stretched 1000 : 1) from losing hits. m: object_to_world, column-major 4x4; w2o: its inverse's
// affine part as InstanceRec stores it; scene_mag: largest |coordinate| of the scene's world bounds.
float instance_pad(const float m[16], const float w2o[12], float ext, float scene_mag)
{
    float nm = 0.f, nw = 0.f; // infinity norms of the linear parts
    for (int r = 0; r < 3; ++r) {
        nm = std::max(nm, std::fabs(m[r]) + std::fabs(m[4 + r]) + std::fabs(m[8 + r]));
        nw = std::max(nw, std::fabs(w2o[r]) + std::fabs(w2o[3 + r]) + std::fabs(w2o[6 + r]));
    }
    const float pad = std::max(1e-5f * ext, 8.f * 1.1920929e-7f * nm * nw * scene_mag);
    return pad >= 0.f && std::isfinite(pad) ? pad : 0.f;
}

void check_scene(const crt_scene_desc *s)
{
    if (!s) {
        throw std::runtime_error("scene is null");
    }
    if (s->n_instances == 0 || s->n_meshes == 0 || s->n_geometries == 0 || s->n_parameterized_meshes == 0) {
        throw std::runtime_error("scene has no instances/meshes/geometries");
    }
    if (s->n_lights == 0) {
        throw std::runtime_error("scene has no lights (the reference divides by num_lights)");
    }
    if (s->n_materials == 0) {
        throw std::runtime_error("scene has no materials");
    }
    // every array with a non-zero count must be there: nothing may crash across the C ABI
    if (!s->instances || !s->meshes || !s->geometries || !s->parameterized_meshes || !s->materials || !s->lights ||
        (s->n_textures != 0 && !s->textures)) {
        throw std::runtime_error("scene has a NULL array with a non-zero count");
    }
    for (uint32_t i = 0; i < s->n_instances; ++i) {
        const uint32_t pm = s->instances[i].parameterized_mesh_id;
        if (pm >= s->n_parameterized_meshes) {
            throw std::runtime_error("instance references a missing parameterized mesh");
        }
        const crt_parameterized_mesh_desc &p = s->parameterized_meshes[pm];
        if (p.mesh_id >= s->n_meshes || p.n_material_ids < s->meshes[p.mesh_id].n_geometries) {
            throw std::runtime_error("parameterized mesh / material id count mismatch");
        }
        if (p.n_material_ids != 0 && !p.material_ids) {
            throw std::runtime_error("parameterized mesh without its material id array");
        }
        for (uint32_t k = 0; k < p.n_material_ids; ++k) {
            if (p.material_ids[k] >= s->n_materials) {
                throw std::runtime_error("material id out of range");
            }
        }
    }
    for (uint32_t m = 0; m < s->n_meshes; ++m) {
        if ((uint64_t)s->meshes[m].first_geometry + (uint64_t)s->meshes[m].n_geometries > (uint64_t)s->n_geometries) {
            throw std::runtime_error("mesh geometry range out of bounds");
        }
        if (s->meshes[m].n_geometries > SLOT_GEOM_MASK) {
            throw std::runtime_error("mesh with more than 2^26 geometries: the geomID shares its leaf-slot word with six selector bits");
        }
    }
    // limits of the encodings, whatever acceleration structure the scene ends up with: an instance leaf holds the instance
    // in 28 bits (and 0x80000000, the traversal stack's sentinel, would be instance 2^28 - 1); a world tree's leaf slot holds
    // (instance << 1) | identity; bit 31 of a material id says that the material reads a texture (MATERIAL_TEXTURED)
    if (s->n_instances >= (1u << 28) - 1u) {
        throw std::runtime_error("too many instances for the 28-bit leaf reference");
    }
    if (s->n_materials >= MATERIAL_TEXTURED) {
        throw std::runtime_error("too many materials: bit 31 of a material id is the textured flag");
    }
    for (uint32_t g = 0; g < s->n_geometries; ++g) {
        const crt_geometry_desc &gd = s->geometries[g];
        if ((gd.n_vertices != 0 && !gd.vertices) || (gd.n_triangles != 0 && !gd.indices)) {
            throw std::runtime_error("geometry without its vertex / index array");
        }
        if (gd.n_triangles >= (1ull << 32) || gd.n_vertices >= (1ull << 32)) {
            throw std::runtime_error("geometry too large for 32-bit indices");
        }
        for (uint64_t t = 0; t < 3 * gd.n_triangles; ++t) {
            if (gd.indices[t] >= gd.n_vertices) {
                throw std::runtime_error("triangle index out of range");
            }
        }
    }
    for (uint32_t m = 0; m < s->n_materials; ++m) {
        for (int k = 0; k < 14; ++k) {
            if (k == 1 || k == 2) {
                continue; // base_color.g/.b are never handles
            }
            uint32_t bits;
            std::memcpy(&bits, &s->materials[16 * (size_t)m + k], 4);
            if ((bits & 0x80000000u) && (bits & 0x1fffffffu) >= s->n_textures) {
                throw std::runtime_error("material references a missing texture");
            }
        }
    }
}

// World tree or two levels for a scene with several instances (crt_types.h LEVELS_WORLD_TREE)? Read per call, so a
// process can prepare scenes both ways (tests).
bool world_tree_wanted(uint64_t instanced_tris)
{
    const char *levels = std::getenv("CRT_HIP_LEVELS");
    if (levels != nullptr && std::strcmp(levels, "two") == 0) {
        return false;
    }
    if (instanced_tris >= (1u << 28)) { // the leaf reference has 28 bits
        return false;
    }
    if (levels != nullptr && std::strcmp(levels, "world") == 0) {
        return true;
    }
    // ~100 bytes per triangle in HBM (its half of a 64-byte slot, two uv records, its share of the nodes): 2^26 triangles
    // are under 7 GB of a 288 GB part. What bounds the budget is the HOST: the SAH build over the slots needs ~300 bytes per
    // triangle while it runs (records, boxes, the builder's items and temporary nodes) -- 20 GB for 2^26 -- so the default
    // is what a third of the memory the host has AVAILABLE right now covers, at most 2^26 (the two-level structure, which
    // costs O(instances) like the reference's rtcCommitScene, takes over beyond that).
    uint64_t budget = 1ull << 26;
    if (const char *cap = std::getenv("CRT_HIP_WORLD_TREE_MAX_TRIS")) {
        budget = std::strtoull(cap, nullptr, 10);
    } else if (FILE *f = std::fopen("/proc/meminfo", "r")) {
        char line[128];
        while (std::fgets(line, sizeof(line), f)) {
            unsigned long long kb = 0;
            if (std::sscanf(line, "MemAvailable: %llu kB", &kb) == 1) {
                budget = std::min<uint64_t>(budget, kb * 1024ull / 3ull / 300ull);
                break;
            }
        }
        std::fclose(f);
    }
    return CRT_WORLD_TREE_DEFAULT && instanced_tris <= budget;
}


// A leaf is ONE 64-byte slot (one triangle, or two that share an edge: leaf_slots.h): with 4-wide nodes a leaf is one of
// four boxes tested per node fetch, so small leaves are cheap to reach, and a leaf visit is then exactly one cache line
// and one dependent step. CRT_BVH_MAX_LEAF (tuning) counts slots.
int max_leaf_setting()
{
    static const int max_leaf = std::getenv("CRT_BVH_MAX_LEAF") ? std::atoi(std::getenv("CRT_BVH_MAX_LEAF")) : 1;
    return max_leaf;
}

// One prepare_scene call: the scene, the arrays being filled, and what the steps hand each other.
struct ScenePreparer {
    const crt_scene_desc *s;
    crt_hip_prepared_scene *ps;
    const int n_threads;
    const int build_device; // >= 0: meshes large enough to be worth it get their BLAS from the device builder on that HIP device
    const int max_leaf = max_leaf_setting();
    int host_lbvh = 0;       // CRT_BVH_BUILDER=lbvh / ploc: one of the device builder's algorithms (1 Karras tree, 2 PLOC), run serially on the host
    bool world_tree = false; // several instances, one tree in world space (crt_types.h LEVELS_WORLD_TREE)
    bool two_level = false;  // several instances, a top-level tree over them
    bool empty_scene = false; // no instance has a triangle: one node without children, every ray misses (like the reference's empty Embree scene)
    std::vector<uint8_t> mesh_empty; // two-level: meshes without triangles have no BLAS, their instances are not in the top-level tree
    uint64_t instanced_tris = 0;
    // per mesh: its BLAS as built (host: full-precision nodes; device: quantised already), and where it ends up
    std::vector<BuiltBvh> built;
    std::vector<std::vector<QNode>> built_q;
    std::vector<QFrame> blas_frame;
    std::vector<int32_t> blas_root; // first the mesh's triangle base, after append_mesh_trees() its root node
    std::vector<Aabb> blas_bounds;
    std::vector<uint32_t> blas_top;
    uint32_t blas_depth = 0, tlas_depth = 0; // levels of the wide trees
    int32_t world_inst = -1;                 // the instance grafted into the top-level tree, and its mesh
    uint32_t world_mesh = 0xffffffffu;
    std::vector<Aabb> inst_boxes;            // two-level: world box of every instance
    std::vector<std::vector<SlotTris>> geom_slots; // per geometry of the scene: which triangles share a leaf slot (leaf_slots.h)
    uint32_t n_top = 0;
    int32_t root = 0;
    QFrame root_frame{};
    std::vector<QNode> qnodes; // the tree as the builders deliver it; packed into ps->nodes at the end (pack_nodes)
    const bool dbg = std::getenv("CRT_HIP_DEBUG") != nullptr;
    std::chrono::high_resolution_clock::time_point t_phase = std::chrono::high_resolution_clock::now();

    const int reinsert_passes; // build_bvh's: < 0 = the environment's / default
    const bool tree_only;      // prepare_scene: skip validation and textures (done by an earlier call on the same scene)
    ScenePreparer(const crt_scene_desc *scene, crt_hip_prepared_scene *prepared, int threads, int device, int reinsert, bool only_tree)
        : s(scene), ps(prepared), n_threads(threads), build_device(device), reinsert_passes(reinsert), tree_only(only_tree)
    {
    }

    void phase(const char *what)
    {
        const auto now = std::chrono::high_resolution_clock::now();
        if (dbg) {
            std::fprintf(stderr, "[crt_hip] set_scene %-22s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_phase).count());
        }
        t_phase = now;
    }

    void run()
    {
        const auto t_begin = t_phase;
        if (!tree_only) {
            check_scene(s);
            phase("validate");
        }
        ps->spp = s->samples_per_pixel ? s->samples_per_pixel : 1;
        choose_structure();
        pair_all_geometries();
        phase("triangle pairs");
        if (empty_scene) {
            make_instance_records();
            make_empty_tree();
        } else {
            build_mesh_trees();
            phase("leaf-order triangles");
            make_instance_records();
            build_world_tree();
            build_top_level_tree();
            append_mesh_trees();
        }
        finish_references();
        pack_nodes();
        phase("TLAS + quantisation");
        if (!tree_only) {
            linearise_textures();
            phase("textures");
        }
        copy_tables();
        ps->root_frame = root_frame;
        ps->root = root;
        ps->two_level = world_tree ? LEVELS_WORLD_TREE : two_level ? 1u : 0u;
        ps->n_top = n_top;
        ps->n_lights = s->n_lights;
        ps->n_instances = s->n_instances;
        ps->build_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t_begin).count();
    }

    // Several instances: either a top-level tree over instances (two-level traversal, what Embree does:
    // embree_utils.cpp:90-129), or -- when the instanced triangles fit a memory budget, which on a 288 GB part is
    // nearly always -- ONE tree in world space over per-instance copies of the triangle records (crt_types.h
    // LEVELS_WORLD_TREE). CRT_HIP_LEVELS=two|world overrides the choice.
    void choose_structure()
    {
        for (uint32_t i = 0; i < s->n_instances; ++i) {
            const crt_mesh_desc &md = s->meshes[s->parameterized_meshes[s->instances[i].parameterized_mesh_id].mesh_id];
            for (uint32_t k = 0; k < md.n_geometries; ++k) {
                instanced_tris += s->geometries[md.first_geometry + k].n_triangles;
            }
        }
        empty_scene = instanced_tris == 0; // (meshes nobody instances do not count: they are never seen)
        world_tree = !empty_scene && s->n_instances > 1 && world_tree_wanted(instanced_tris);
        two_level = !empty_scene && s->n_instances > 1 && !world_tree;
        mesh_empty.assign(s->n_meshes, 0);
        blas_frame.resize(s->n_meshes);
        blas_root.resize(s->n_meshes);
        blas_bounds.resize(s->n_meshes);
        blas_top.resize(s->n_meshes);
        built.resize(s->n_meshes);
        built_q.resize(s->n_meshes);
        const char *builder_env = std::getenv("CRT_BVH_BUILDER");
        host_lbvh = builder_env && std::strcmp(builder_env, "lbvh") == 0 ? 1 : builder_env && std::strcmp(builder_env, "ploc") == 0 ? 2 : 0;
        // The static part of an instanced scene -- an identity instance whose mesh nothing else uses: the building
        // of a San-Miguel-like scene with its instanced plants -- is not entered like an instance. Its BLAS is
        // opened from the root down to a CUT of subtrees about as large as the other instances, the top-level tree is
        // built over those subtrees AND the other instances' boxes, and a cut subtree is referenced by a plain
        // child reference (its nodes are quantised in the top-level frame; an identity instance is traversed with
        // the world-space ray anyway, so the hits are the same bit for bit). A ray then no longer walks a TLAS down
        // to an instance box that covers the whole scene, enters it and starts again at the BLAS root: it walks
        // one tree in which the plants sit where they stand, and the entry / exit steps of the big instance are gone.
        // A mesh that was built on the device arrives quantised in its own frame: its boxes are read back from the
        // 16-bit form for the cut, and its nodes re-quantised (outward again) into the top-level frame.
        if (two_level && max_leaf <= 7 && !std::getenv("CRT_HIP_NO_GRAFT")) {
            std::vector<uint32_t> mesh_refs(s->n_meshes, 0);
            for (uint32_t i = 0; i < s->n_instances; ++i) {
                ++mesh_refs[s->parameterized_meshes[s->instances[i].parameterized_mesh_id].mesh_id];
            }
            uint64_t most = 0;
            for (uint32_t i = 0; i < s->n_instances; ++i) {
                const uint32_t m = s->parameterized_meshes[s->instances[i].parameterized_mesh_id].mesh_id;
                if (mesh_refs[m] != 1 || !is_identity(s->instances[i].transform)) {
                    continue;
                }
                uint64_t n_tris_m = 0;
                for (uint32_t k = 0; k < s->meshes[m].n_geometries; ++k) {
                    n_tris_m += s->geometries[s->meshes[m].first_geometry + k].n_triangles;
                }
                if (n_tris_m > most) {
                    most = n_tris_m;
                    world_inst = (int32_t)i;
                    world_mesh = m;
                }
            }
        }
    }

    // Which triangles of each geometry share a leaf slot: once per geometry, whatever the number of instances of its mesh.
    void pair_all_geometries()
    {
        geom_slots.assign(s->n_geometries, {});
        const float max_ratio = pair_max_ratio();
        // (a scene is a handful of big geometries or thousands of small ones: one geometry per task either way)
        std::atomic<uint32_t> next{0};
        auto work = [&]() {
            for (uint32_t g = next.fetch_add(1); g < s->n_geometries; g = next.fetch_add(1)) {
                const crt_geometry_desc &gd = s->geometries[g];
                geom_slots[g] = pair_triangles(gd.vertices, gd.indices, gd.n_triangles, max_ratio);
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < std::min<int>(n_threads, (int)s->n_geometries); ++t) {
            pool.emplace_back(work);
        }
        work();
        for (std::thread &th : pool) {
            th.join();
        }
        if (dbg) {
            uint64_t n_slots = 0, n_tris_all = 0;
            for (uint32_t g = 0; g < s->n_geometries; ++g) {
                n_slots += geom_slots[g].size();
                n_tris_all += s->geometries[g].n_triangles;
            }
            std::fprintf(stderr, "[crt_hip] set_scene: %llu triangles in %llu leaf slots (%.1f %% of the triangles paired)\n",
                         (unsigned long long)n_tris_all, (unsigned long long)n_slots,
                         n_tris_all ? 200.0 * (double)(n_tris_all - n_slots) / (double)n_tris_all : 0.0);
        }
    }

    // leaf slot + the uvs of its triangles' vertices (uv_buf[indices.x|y|z], render_embree.ispc:278-283) at position `at`
    void place_slot(const crt_mesh_desc &md, const LeafSlot &r, size_t at)
    {
        ps->slots[at] = r;
        const crt_geometry_desc &gd = s->geometries[md.first_geometry + (r.geom_sel & SLOT_GEOM_MASK)];
        if (gd.uvs) {
            for (int which = 0; which < 2; ++which) {
                const uint32_t prim = which == 0 ? r.prim0 : r.prim1;
                if (prim == SLOT_NO_SECOND) {
                    continue;
                }
                float *uv = ps->tri_uvs.data() + (size_t)TRI_UV_STRIDE * (2 * at + (size_t)which);
                for (int c = 0; c < 3; ++c) {
                    const uint32_t vi = gd.indices[3 * (size_t)prim + c];
                    uv[2 * c] = gd.uvs[2 * (size_t)vi];
                    uv[2 * c + 1] = gd.uvs[2 * (size_t)vi + 1];
                }
            }
        }
    }

    static void report_presplit(const PresplitStats &st)
    {
        if (std::getenv("CRT_HIP_DEBUG") || std::getenv("CRT_BVH_SPLITS_REPORT")) {
            std::fprintf(stderr, "[crt_hip] presplit: %llu -> %llu leaf items (+%.1f %%), summed box area %.4g -> %.4g (%.1f %%)\n",
                         (unsigned long long)st.items_in, (unsigned long long)st.items_out,
                         100.0 * (double)(st.items_out - st.items_in) / (double)std::max<uint64_t>(1, st.items_in), st.area_in, st.area_out,
                         100.0 * st.area_out / std::max(1e-30, st.area_in));
        }
    }

    // world-space (or, m == nullptr, object-space) box of a slot's triangles, pushed out by `pad`
    static Aabb slot_box(const crt_geometry_desc &gd, SlotTris st, const float *m, float pad)
    {
        Aabb b;
        for (int a = 0; a < 3; ++a) {
            b.lo[a] = INFINITY;
            b.hi[a] = -INFINITY;
        }
        for (int which = 0; which < (st.b == SLOT_NO_SECOND ? 1 : 2); ++which) {
            const uint32_t prim = which == 0 ? st.a : st.b;
            for (int c = 0; c < 3; ++c) {
                const float *p = gd.vertices + 3 * (size_t)gd.indices[3 * (size_t)prim + c];
                for (int a = 0; a < 3; ++a) {
                    const float w = m ? m[a] * p[0] + m[4 + a] * p[1] + m[8 + a] * p[2] + m[12 + a] : p[a];
                    b.lo[a] = std::min(b.lo[a], w);
                    b.hi[a] = std::max(b.hi[a], w);
                }
            }
        }
        for (int a = 0; a < 3; ++a) {
            b.lo[a] -= pad;
            b.hi[a] += pad;
        }
        return b;
    }

    // one BLAS per Mesh (embree_utils.cpp:63-76); a world tree has none
    void build_mesh_trees()
    {
        std::vector<LeafSlot> &slots = ps->slots;
        std::vector<float> &tri_uvs = ps->tri_uvs;
        for (uint32_t m = 0; m < s->n_meshes && !world_tree; ++m) {
            const crt_mesh_desc &md = s->meshes[m];
            if (build_device >= 0) {
                DeviceBuiltMesh db;
                bool built_on_device = false;
                try {
                    built_on_device = device_build_mesh(build_device, s->geometries + md.first_geometry, md.n_geometries,
                                                        geom_slots.data() + md.first_geometry, (uint32_t)max_leaf,
                                                        two_level ? 0 : MAX_TOP_NODES_HOST, db);
                } catch (const std::exception &e) { // e.g. out of device memory: the host builder still can
                    std::fprintf(stderr, "[crt_hip] %s -- building mesh %u on the host instead\n", e.what(), m);
                    (void)hipGetLastError();
                }
                if (built_on_device) {
                    const size_t slot_base = slots.size();
                    slots.insert(slots.end(), db.slots.begin(), db.slots.end());
                    tri_uvs.insert(tri_uvs.end(), db.tri_uvs.begin(), db.tri_uvs.end());
                    built_q[m] = std::move(db.nodes);
                    built[m].n_top = db.n_top;
                    built[m].max_depth = db.max_depth;
                    built[m].bounds = db.bounds;
                    blas_depth = std::max(blas_depth, db.max_depth);
                    blas_bounds[m] = db.bounds;
                    blas_frame[m] = db.frame;
                    blas_root[m] = (int32_t)slot_base;
                    continue;
                }
            }
            uint64_t n_mesh_slots = 0;
            std::vector<uint64_t> first_slot(md.n_geometries + 1, 0);
            for (uint32_t k = 0; k < md.n_geometries; ++k) {
                n_mesh_slots += geom_slots[md.first_geometry + k].size();
                first_slot[k + 1] = n_mesh_slots;
            }
            std::vector<LeafSlot> recs(n_mesh_slots);
            std::vector<Aabb> boxes(n_mesh_slots);
            for (uint32_t k = 0; k < md.n_geometries; ++k) {
                const crt_geometry_desc &gd = s->geometries[md.first_geometry + k];
                const std::vector<SlotTris> &gs = geom_slots[md.first_geometry + k];
                const uint64_t at0 = first_slot[k];
                parallel_for(gs.size(), n_threads, 1u << 15, [&](size_t lo, size_t hi) {
                    for (size_t i = lo; i < hi; ++i) {
                        recs[at0 + i] = make_leaf_slot(gd.vertices, gd.indices, k, gs[i], 0u);
                        boxes[at0 + i] = slot_box(gd, gs[i], nullptr, 0.f);
                    }
                });
            }
            if (recs.empty()) { // a mesh without triangles: nothing to build, nothing to hit (its instances stay out of the top-level tree)
                mesh_empty[m] = 1;
                blas_root[m] = (int32_t)slots.size();
                blas_frame[m] = make_frame(Aabb{{0.f, 0.f, 0.f}, {1.f, 1.f, 1.f}});
                blas_bounds[m] = Aabb{{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
                continue;
            }
            if (presplit_fraction() > 0.0) { // opt-in (CRT_BVH_SPLITS): loosely boxed slots get several leaves (presplit.h)
                report_presplit(presplit_items(recs, boxes, [](const LeafSlot &, const float *&mm, float &pad) { mm = nullptr; pad = 0.f; },
                                               presplit_fraction()));
            }
            built[m] = host_lbvh == 2 ? build_ploc_host(boxes.data(), boxes.size(), max_leaf, two_level ? 0 : MAX_TOP_NODES_HOST)
                       : host_lbvh ? build_lbvh_host(boxes.data(), boxes.size(), max_leaf, two_level ? 0 : MAX_TOP_NODES_HOST)
                                 : build_bvh(boxes.data(), boxes.size(), max_leaf, 0, 0, false,
                                             two_level ? 0 : MAX_TOP_NODES_HOST, n_threads, reinsert_passes);
            blas_depth = std::max(blas_depth, built[m].max_depth);
            // slots in leaf order
            const size_t slot_base = slots.size();
            slots.resize(slot_base + recs.size());
            tri_uvs.resize((size_t)TRI_UV_STRIDE * 2 * slots.size(), 0.f);
            parallel_for(recs.size(), n_threads, 1u << 15, [&](size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; ++i) {
                    place_slot(md, recs[built[m].order[i]], slot_base + i);
                }
            });
            blas_bounds[m] = built[m].bounds;
            blas_frame[m] = make_frame(built[m].bounds);
            // re-base the node / leaf references later, once the TLAS size is known
            blas_root[m] = (int32_t)slot_base; // temporarily: slot base
        }
    }

    void make_instance_records()
    {
        // instances + TLAS (embree_utils.cpp:90-104, 121-129)
        std::vector<InstanceRec> &insts = ps->insts;
        insts.assign(s->n_instances, InstanceRec{});
        std::vector<uint32_t> &material_ids = ps->material_ids;
        inst_boxes.assign(s->n_instances, Aabb{});
        for (uint32_t i = 0; i < s->n_instances; ++i) {
            const crt_instance_desc &id = s->instances[i];
            const crt_parameterized_mesh_desc &pm = s->parameterized_meshes[id.parameterized_mesh_id];
            InstanceRec r;
            std::memset(&r, 0, sizeof(r));
            float inv[16];
            if (!invert4x4(id.transform, inv)) {
                throw std::runtime_error("singular instance transform");
            }
            for (int c = 0; c < 4; ++c) { // keep the affine 3x4 part (the last row of an instance transform is 0 0 0 1)
                for (int rr = 0; rr < 3; ++rr) {
                    r.w2o[c * 3 + rr] = inv[c * 4 + rr];
                }
            }
            r.identity = is_identity(id.transform) ? 1u : 0u;
            r.geom_base = s->meshes[pm.mesh_id].first_geometry;
            r.mat_base = (uint32_t)material_ids.size();
            r.blas_root = (int32_t)pm.mesh_id; // temporarily: mesh id
            r.frame = blas_frame[pm.mesh_id];
            for (uint32_t k = 0; k < pm.n_material_ids; ++k) {
                // bit 31: some parameter of the material is a texture handle (render_embree.ispc:66-103 tests the same
                // sign bit per parameter) -- k_shade fetches the hit's uv record only then
                const uint32_t id = pm.material_ids[k];
                bool textured = false;
                for (int f = 0; f < 14; ++f) {
                    uint32_t bits;
                    std::memcpy(&bits, s->materials + 16 * (size_t)id + f, 4);
                    textured = textured || (bits & 0x80000000u) != 0u;
                }
                material_ids.push_back(id | (textured ? MATERIAL_TEXTURED : 0u));
            }
            insts[i] = r;
            if (world_tree || empty_scene) {
                continue; // no instance boxes: the tree is built over the triangles (below), or there is nothing to bound
            }
            const Aabb &mb = blas_bounds[pm.mesh_id];
            Aabb wb;
            for (int a = 0; a < 3; ++a) {
                wb.lo[a] = INFINITY;
                wb.hi[a] = -INFINITY;
            }
            const float *m = id.transform;
            const crt_mesh_desc &imd = s->meshes[pm.mesh_id];
            uint64_t mesh_verts = 0;
            for (uint32_t k = 0; k < imd.n_geometries; ++k) {
                mesh_verts += s->geometries[imd.first_geometry + k].n_vertices;
            }
            if (!r.identity && mesh_verts <= (1u << 20)) {
                // The world box of the transformed VERTICES, not of the transformed corners of the object-space box:
                // for a rotated instance the latter is up to 40 % wider on each axis, and every ray that enters an
                // instance box pays a transform, a frame change and a walk from the BLAS root. (Instanced meshes are
                // small; a mesh of more than a million vertices keeps the corner box.)
                for (uint32_t k = 0; k < imd.n_geometries; ++k) {
                    const crt_geometry_desc &gd = s->geometries[imd.first_geometry + k];
                    for (uint64_t v = 0; v < gd.n_vertices; ++v) {
                        const float *p = gd.vertices + 3 * v;
                        for (int a = 0; a < 3; ++a) {
                            const float w = m[a] * p[0] + m[4 + a] * p[1] + m[8 + a] * p[2] + m[12 + a];
                            wb.lo[a] = std::min(wb.lo[a], w);
                            wb.hi[a] = std::max(wb.hi[a], w);
                        }
                    }
                }
            } else {
                for (int c = 0; c < 8; ++c) {
                    const float p[3] = {(c & 1) ? mb.hi[0] : mb.lo[0], (c & 2) ? mb.hi[1] : mb.lo[1],
                                        (c & 4) ? mb.hi[2] : mb.lo[2]};
                    for (int a = 0; a < 3; ++a) {
                        const float w = r.identity ? p[a] : m[a] * p[0] + m[4 + a] * p[1] + m[8 + a] * p[2] + m[12 + a];
                        wb.lo[a] = std::min(wb.lo[a], w);
                        wb.hi[a] = std::max(wb.hi[a], w);
                    }
                }
            }
            inst_boxes[i] = wb;
        }
        // pad: the BLAS is walked with a transformed (rounded) ray (instance_pad)
        float scene_mag = 0.f;
        for (const Aabb &b : inst_boxes) {
            for (int a = 0; a < 3; ++a) {
                scene_mag = std::max(scene_mag, std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])));
            }
        }
        for (uint32_t i = 0; i < s->n_instances && !world_tree; ++i) {
            Aabb &wb = inst_boxes[i];
            const float ext = std::max(wb.hi[0] - wb.lo[0], std::max(wb.hi[1] - wb.lo[1], wb.hi[2] - wb.lo[2]));
            const float pad = insts[i].identity ? 1e-5f * ext : instance_pad(s->instances[i].transform, insts[i].w2o, ext, scene_mag);
            for (int a = 0; a < 3; ++a) {
                wb.lo[a] -= pad;
                wb.hi[a] += pad;
            }
        }
    }

    void build_world_tree()
    {
        std::vector<QNode> &nodes = qnodes;
        std::vector<LeafSlot> &slots = ps->slots;
        std::vector<float> &tri_uvs = ps->tri_uvs;
        std::vector<InstanceRec> &insts = ps->insts;
        if (world_tree) {
            // One leaf slot + one world-space box per (instance, slot of its mesh). The record is the mesh's own (object
            // space: the triangle test runs there, with the ray transformed like the reference transforms it, so t / u / v
            // come out bit for bit as in the two-level walk); the box bounds the transformed vertices, padded like an
            // instance box (the test ray is a rounded transform of the world ray) -- and quantisation rounds outward by
            // >= 1 quantum of the scene's extent on top of that.
            // where each instance's slots start, so that the instances can be filled side by side
            std::vector<uint64_t> first_rec(s->n_instances + 1, 0);
            for (uint32_t i = 0; i < s->n_instances; ++i) {
                const crt_mesh_desc &md = s->meshes[s->parameterized_meshes[s->instances[i].parameterized_mesh_id].mesh_id];
                uint64_t n_inst_slots = 0;
                for (uint32_t k = 0; k < md.n_geometries; ++k) {
                    n_inst_slots += geom_slots[md.first_geometry + k].size();
                }
                first_rec[i + 1] = first_rec[i] + n_inst_slots;
            }
            const uint64_t instanced_slots = first_rec[s->n_instances];
            std::vector<LeafSlot> recs; // (sized below, once it is clear that the host builds the tree)
            std::vector<Aabb> boxes;
            // world box of every instance's vertices: its extent and the scene's magnitude set the padding (instance_pad)
            std::vector<Aabb> inst_world(s->n_instances);
            parallel_for(s->n_instances, n_threads, 1, [&](size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; ++i) {
                    const crt_instance_desc &id = s->instances[i];
                    const crt_mesh_desc &md = s->meshes[s->parameterized_meshes[id.parameterized_mesh_id].mesh_id];
                    const float *m = id.transform;
                    const bool ident = insts[i].identity != 0u;
                    Aabb wb;
                    for (int a = 0; a < 3; ++a) {
                        wb.lo[a] = INFINITY;
                        wb.hi[a] = -INFINITY;
                    }
                    for (uint32_t k = 0; k < md.n_geometries; ++k) {
                        const crt_geometry_desc &gd = s->geometries[md.first_geometry + k];
                        for (uint64_t v = 0; v < gd.n_vertices; ++v) {
                            const float *p = gd.vertices + 3 * v;
                            for (int a = 0; a < 3; ++a) {
                                const float w = ident ? p[a] : m[a] * p[0] + m[4 + a] * p[1] + m[8 + a] * p[2] + m[12 + a];
                                wb.lo[a] = std::min(wb.lo[a], w);
                                wb.hi[a] = std::max(wb.hi[a], w);
                            }
                        }
                    }
                    inst_world[i] = wb;
                }
            });
            float scene_mag = 0.f;
            for (const Aabb &b : inst_world) {
                for (int a = 0; a < 3; ++a) {
                    if (b.lo[a] <= b.hi[a]) {
                        scene_mag = std::max(scene_mag, std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])));
                    }
                }
            }
            auto pad_of = [&](uint32_t i) -> float { // an identity instance's triangles are tested with the world ray itself
                if (insts[i].identity != 0u) {
                    return 0.f;
                }
                const Aabb &wb = inst_world[i];
                return instance_pad(s->instances[i].transform, insts[i].w2o,
                                    std::max(wb.hi[0] - wb.lo[0], std::max(wb.hi[1] - wb.lo[1], wb.hi[2] - wb.lo[2])), scene_mag);
            };
            if (build_device >= 0 && !host_lbvh) {
                // The same tree on the device (bvh_device.hip device_build_world): slot records and world boxes per
                // (instance, slot), Morton sort, radix tree, refit, collapse -- set_scene of the instanced C4 without the
                // host SAH build over 5 M slots. Falls through to the host builder if the device cannot (memory).
                std::vector<uint32_t> ident(s->n_instances);
                std::vector<float> pads(s->n_instances);
                for (uint32_t i = 0; i < s->n_instances; ++i) {
                    ident[i] = insts[i].identity;
                    pads[i] = pad_of(i);
                }
                DeviceBuiltMesh db;
                bool built_on_device = false;
                try {
                    built_on_device = device_build_world(build_device, s, geom_slots.data(), ident.data(), pads.data(), (uint32_t)max_leaf,
                                                         MAX_TOP_NODES_HOST, db);
                } catch (const std::exception &e) {
                    std::fprintf(stderr, "[crt_hip] %s -- building the world tree on the host instead\n", e.what());
                    (void)hipGetLastError();
                }
                if (built_on_device) {
                    nodes = std::move(db.nodes);
                    slots = std::move(db.slots);
                    tri_uvs = std::move(db.tri_uvs);
                    blas_depth = db.max_depth;
                    root_frame = db.frame;
                    n_top = db.n_top;
                    root = 0;
                    for (InstanceRec &r : insts) {
                        r.frame = root_frame;
                    }
                    return;
                }
            }
            recs.resize(instanced_slots);
            boxes.resize(instanced_slots);
            // (a scene is one big static instance plus many small ones, or many alike: instances are dealt out one at a
            // time, and a big one is cut by geometry ranges inside fill_instance)
            auto fill_instance = [&](uint32_t i, int threads) {
                const crt_instance_desc &id = s->instances[i];
                const crt_mesh_desc &md = s->meshes[s->parameterized_meshes[id.parameterized_mesh_id].mesh_id];
                const float *m = id.transform;
                const bool ident = insts[i].identity != 0u;
                const float pad = pad_of(i);
                uint64_t at0 = first_rec[i];
                for (uint32_t k = 0; k < md.n_geometries; ++k) {
                    const crt_geometry_desc &gd = s->geometries[md.first_geometry + k];
                    const std::vector<SlotTris> &gs = geom_slots[md.first_geometry + k];
                    parallel_for(gs.size(), threads, 1u << 15, [&](size_t lo, size_t hi) {
                        for (size_t t = lo; t < hi; ++t) {
                            recs[at0 + t] = make_leaf_slot(gd.vertices, gd.indices, k, gs[t], (i << 1) | (ident ? 1u : 0u));
                            boxes[at0 + t] = slot_box(gd, gs[t], ident ? nullptr : m, pad);
                        }
                    });
                    at0 += gs.size();
                }
            };
            {
                // big instances one after the other with all threads inside, the small ones dealt out to the threads
                const uint64_t big = std::max<uint64_t>(1u << 16, instanced_slots / (uint64_t)std::max(1, n_threads));
                std::vector<uint32_t> small;
                for (uint32_t i = 0; i < s->n_instances; ++i) {
                    if (first_rec[i + 1] - first_rec[i] >= big) {
                        fill_instance(i, n_threads);
                    } else {
                        small.push_back(i);
                    }
                }
                parallel_for(small.size(), n_threads, 4, [&](size_t lo, size_t hi) {
                    for (size_t j = lo; j < hi; ++j) {
                        fill_instance(small[j], 1);
                    }
                });
            }
            if (presplit_fraction() > 0.0) {
                std::vector<float> pads(s->n_instances);
                for (uint32_t i = 0; i < s->n_instances; ++i) {
                    pads[i] = pad_of(i);
                }
                report_presplit(presplit_items(recs, boxes, [&](const LeafSlot &r, const float *&mm, float &pad) {
                    mm = (r.tag & 1u) != 0u ? nullptr : s->instances[r.tag >> 1].transform;
                    pad = pads[r.tag >> 1];
                }, presplit_fraction()));
            }
            BuiltBvh tree = host_lbvh == 2 ? build_ploc_host(boxes.data(), boxes.size(), max_leaf, MAX_TOP_NODES_HOST)
                            : host_lbvh ? build_lbvh_host(boxes.data(), boxes.size(), max_leaf, MAX_TOP_NODES_HOST)
                                      : build_bvh(boxes.data(), boxes.size(), max_leaf, 0, 0, false, MAX_TOP_NODES_HOST, n_threads, reinsert_passes);
            boxes = std::vector<Aabb>();
            blas_depth = tree.max_depth;
            slots.resize(recs.size());
            tri_uvs.resize((size_t)TRI_UV_STRIDE * 2 * slots.size(), 0.f);
            parallel_for(recs.size(), n_threads, 1u << 15, [&](size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; ++i) {
                    const LeafSlot &r = recs[tree.order[i]];
                    const crt_instance_desc &id = s->instances[r.tag >> 1];
                    place_slot(s->meshes[s->parameterized_meshes[id.parameterized_mesh_id].mesh_id], r, i);
                }
            });
            root_frame = make_frame(tree.bounds);
            nodes.resize(tree.nodes.size());
            parallel_for(tree.nodes.size(), n_threads, 1u << 14, [&](size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; ++i) {
                    nodes[i] = quantise(tree.nodes[i], root_frame);
                }
            });
            n_top = tree.n_top;
            root = 0;
            for (InstanceRec &r : insts) {
                r.frame = root_frame; // not read by the traversal of a world tree; blas_root stays a mesh id until the loop below
            }
        }
    }

    // the top-level tree of a two-level scene (embree_utils.cpp:121-129), with the static instance grafted into it
    void build_top_level_tree()
    {
        std::vector<QNode> &nodes = qnodes;
        if (two_level) {
            // items of the top-level tree: the cut through the grafted mesh's BLAS (if any), then one box per other instance
            std::vector<Aabb> items;
            std::vector<int32_t> item_ref; // what the leaf of an item becomes: a reference local to the grafted BLAS, or an instance leaf
            std::vector<uint8_t> item_is_cut;
            if (world_inst >= 0 && std::getenv("CRT_HIP_GRAFT_QNODES") && built_q[world_mesh].empty()) {
                // (test hook: hand the host-built mesh over in the quantised form a device build delivers, so the CPU
                // tests reach the read-back / re-quantise path below without a GPU)
                for (const BvhNode &nd : built[world_mesh].nodes) {
                    built_q[world_mesh].push_back(quantise(nd, blas_frame[world_mesh]));
                }
                built[world_mesh].nodes.clear();
            }
            const QFrame world_frame_in = world_inst >= 0 ? blas_frame[world_mesh] : QFrame{}; // the frame its quantised nodes are in
            auto dequantised = [&](const QChild &c) {
                Aabb b;
                for (int a = 0; a < 3; ++a) {
                    b.lo[a] = world_frame_in.base[a] + (float)c.q[a][0] * world_frame_in.step[a];
                    b.hi[a] = world_frame_in.base[a] + (float)c.q[a][1] * world_frame_in.step[a];
                }
                return b;
            };
            if (world_inst >= 0) {
                const std::vector<BvhNode> &wn = built[world_mesh].nodes;
                const std::vector<QNode> &wq = built_q[world_mesh];
                double inst_area = 0.0; // mean half surface area of the other instances' boxes
                for (uint32_t i = 0; i < s->n_instances; ++i) {
                    if ((int32_t)i != world_inst) {
                        const Aabb &b = inst_boxes[i];
                        const double dx = (double)b.hi[0] - b.lo[0], dy = (double)b.hi[1] - b.lo[1], dz = (double)b.hi[2] - b.lo[2];
                        inst_area += (dx * dy + dy * dz + dz * dx) / (double)(s->n_instances - 1);
                    }
                }
                struct CutEntry {
                    double area;
                    Aabb box;
                    int32_t ref;
                    bool operator<(const CutEntry &o) const { return area < o.area; }
                };
                std::priority_queue<CutEntry> open; // inner nodes that may still be opened, largest first
                std::vector<CutEntry> cut;
                auto add_children = [&](int32_t node) {
                    for (int k = 0; k < BVH_WIDTH; ++k) {
                        CutEntry e;
                        if (!wq.empty()) {
                            const QChild &c = wq[(size_t)node].child[k];
                            if (c.q[0][0] > c.q[0][1]) { // unused slot
                                continue;
                            }
                            e.box = dequantised(c);
                            e.ref = c.ref;
                        } else {
                            const BvhNode &nd = wn[(size_t)node];
                            if (nd.c[k] == EMPTY_CHILD) {
                                continue;
                            }
                            for (int a = 0; a < 3; ++a) {
                                e.box.lo[a] = nd.lo[k][a];
                                e.box.hi[a] = nd.hi[k][a];
                            }
                            e.ref = nd.c[k];
                        }
                        const double dx = (double)e.box.hi[0] - e.box.lo[0], dy = (double)e.box.hi[1] - e.box.lo[1],
                                     dz = (double)e.box.hi[2] - e.box.lo[2];
                        e.area = dx * dy + dy * dz + dz * dx;
                        if (e.ref >= 0) {
                            open.push(e);
                        } else {
                            cut.push_back(e); // a triangle leaf directly under an opened node
                        }
                    }
                };
                add_children(0);
                // (how far to open: a cut of 1x, 4x, 16x, 64x the instance count was priced with the oracle's walker on a
                // reduced C4 -- 4x is the flat optimum for camera, bounce and occlusion rays alike)
                const size_t cap = 4 * (size_t)s->n_instances + 64;
                while (!open.empty() && open.top().area > inst_area && cut.size() + open.size() + BVH_WIDTH <= cap) {
                    const CutEntry e = open.top();
                    open.pop();
                    add_children(e.ref);
                }
                for (; !open.empty(); open.pop()) {
                    cut.push_back(open.top());
                }
                for (const CutEntry &e : cut) {
                    items.push_back(e.box);
                    item_ref.push_back(e.ref);
                    item_is_cut.push_back(1);
                }
            }
            for (uint32_t i = 0; i < s->n_instances; ++i) {
                if ((int32_t)i != world_inst && !mesh_empty[s->parameterized_meshes[s->instances[i].parameterized_mesh_id].mesh_id]) {
                    items.push_back(inst_boxes[i]);
                    item_ref.push_back(instance_leaf_ref(i));
                    item_is_cut.push_back(0);
                }
            }
            BuiltBvh tlas = build_bvh(items.data(), items.size(), 1, 0, 0, true, CRT_MAX_TOP_NODES_TWO_LEVEL, 1);
            tlas_depth = tlas.max_depth;
            root_frame = make_frame(tlas.bounds);
            // where the grafted BLAS will lie: the per-mesh loop below appends the meshes in order behind the top-level nodes
            int32_t world_node_base = (int32_t)tlas.nodes.size();
            for (uint32_t m = 0; world_inst >= 0 && m < world_mesh; ++m) {
                world_node_base += (int32_t)(built[m].nodes.size() + built_q[m].size());
            }
            const uint32_t world_tri_base = world_inst >= 0 ? (uint32_t)blas_root[world_mesh] : 0u;
            for (BvhNode nd : tlas.nodes) {
                for (int k = 0; k < BVH_WIDTH; ++k) {
                    const int32_t c = nd.c[k];
                    if (c >= 0 || c == EMPTY_CHILD) {
                        continue;
                    }
                    const uint32_t id = (~(uint32_t)c) >> 3;
                    int32_t ref = item_ref[id];
                    if (item_is_cut[id]) { // local to the grafted BLAS -> global
                        if (ref >= 0) {
                            ref += world_node_base;
                        } else {
                            const uint32_t x = ~(uint32_t)ref;
                            ref = (int32_t)~((((x >> 3) + world_tri_base) << 3) | (x & 7u));
                        }
                    }
                    nd.c[k] = ref;
                }
                if (!std::getenv("CRT_HIP_NO_SLOT_ORDER")) {
                    // Slot order is the order occlusion rays try the children in, and the order in which the children a
                    // closest-hit ray does not take first are stacked: subtrees and triangles of the static mesh before
                    // instances, so that a ray has found what the cheap part of the node holds (an occluder; a nearer hit
                    // that culls the instance's box) before it pays for entering an instance.
                    BvhNode ord = nd;
                    int at = 0;
                    for (int pass = 0; pass < 3; ++pass) {
                        for (int k = 0; k < BVH_WIDTH; ++k) {
                            const int kind = nd.c[k] == EMPTY_CHILD ? 2 : (is_instance_leaf(nd.c[k]) ? 1 : 0);
                            if (kind == pass) {
                                for (int a = 0; a < 3; ++a) {
                                    ord.lo[at][a] = nd.lo[k][a];
                                    ord.hi[at][a] = nd.hi[k][a];
                                }
                                ord.c[at++] = nd.c[k];
                            }
                        }
                    }
                    nd = ord;
                }
                nodes.push_back(quantise(nd, root_frame));
            }
            if (world_inst >= 0) {
                blas_frame[world_mesh] = root_frame; // its nodes are reached from the top-level tree without a frame change
                for (QNode &q : built_q[world_mesh]) { // device-built: from its own frame into that one, outward again
                    for (int k = 0; k < BVH_WIDTH; ++k) {
                        QChild &c = q.child[k];
                        if (c.q[0][0] <= c.q[0][1]) {
                            lbvh_quantise_child(c, dequantised(c), c.ref, root_frame);
                        }
                    }
                }
            }
            n_top = tlas.n_top;
        }
    }

    // A scene none of whose instances has a triangle: one node whose four slots are unused (inverted boxes, which the slab test
    // rejects by itself), traversed as a single-level tree -- every ray misses, every occlusion ray is unoccluded, the frame
    // is the miss shader's checkerboard, as from the reference's empty Embree scene.
    void make_empty_tree()
    {
        BvhNode nd;
        std::memset(&nd, 0, sizeof(nd));
        for (int k = 0; k < BVH_WIDTH; ++k) {
            nd.c[k] = EMPTY_CHILD;
        }
        root_frame = make_frame(Aabb{{0.f, 0.f, 0.f}, {1.f, 1.f, 1.f}});
        QNode q = quantise(nd, root_frame);
        for (int k = 0; k < BVH_WIDTH; ++k) {
            q.child[k].ref = 0; // (never followed: no ray enters an inverted box)
        }
        qnodes.assign(1, q);
        root = 0;
        n_top = 1;
        blas_depth = 1;
        for (InstanceRec &r : ps->insts) {
            r.frame = root_frame;
        }
    }

    // the meshes' trees behind the top-level nodes: references local to a mesh become global
    void append_mesh_trees()
    {
        std::vector<QNode> &nodes = qnodes;
        for (uint32_t m = 0; m < s->n_meshes; ++m) {
            const int32_t node_base = (int32_t)nodes.size();
            const uint32_t tri_base = (uint32_t)blas_root[m];
            auto rebase = [&](int32_t c) -> int32_t {
                if (c >= 0) {
                    return c + node_base;
                }
                const uint32_t x = ~(uint32_t)c;
                return (int32_t)~((((x >> 3) + tri_base) << 3) | (x & 7u));
            };
            for (QNode q : built_q[m]) { // device-built: quantised already, references local to the mesh
                for (int k = 0; k < BVH_WIDTH; ++k) {
                    q.child[k].ref = rebase(q.child[k].ref);
                }
                nodes.push_back(q);
            }
            const size_t host_base = nodes.size();
            nodes.resize(host_base + built[m].nodes.size());
            parallel_for(built[m].nodes.size(), n_threads, 1u << 14, [&](size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; ++i) {
                    BvhNode nd = built[m].nodes[i];
                    for (int k = 0; k < BVH_WIDTH; ++k) {
                        if (nd.c[k] != EMPTY_CHILD) {
                            nd.c[k] = rebase(nd.c[k]);
                        }
                    }
                    nodes[host_base + i] = quantise(nd, blas_frame[m]);
                }
            });
            blas_root[m] = node_base;
            blas_top[m] = built[m].n_top;
            built[m] = BuiltBvh();
            built_q[m] = std::vector<QNode>();
        }
    }

    // the kernels' 48-byte form of every node (crt_types.h PNode)
    void pack_nodes()
    {
        ps->nodes.resize(qnodes.size());
        parallel_for(qnodes.size(), n_threads, 1u << 14, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) {
                ps->nodes[i] = pack_node(qnodes[i]);
            }
        });
        qnodes = std::vector<QNode>();
    }

    void finish_references()
    {
        std::vector<InstanceRec> &insts = ps->insts;
        // a ray's stack holds at most BVH_WIDTH-1 pending siblings per level of the path it is on,
        // plus the instance-exit sentinel. The LDS part of the stack is fixed; the HBM slab behind it is
        // sized from this number at upload, so no tree is "too deep" (the reference renders any scene)
        ps->stack_need = (BVH_WIDTH - 1) * (blas_depth + tlas_depth) + 2;
        if (ps->slots.size() >= (1u << 28)) {
            throw std::runtime_error("too many leaf slots for the 28-bit leaf reference");
        }
        for (InstanceRec &r : insts) {
            r.blas_root = world_tree || empty_scene || mesh_empty[(size_t)r.blas_root] ? 0 : blas_root[r.blas_root];
        }
        if (world_inst >= 0) {
            insts[(size_t)world_inst].blas_root = 0; // never entered: its subtrees hang in the top-level tree (node 0 = its root)
        }
        ps->world_inst = world_inst;
        if (!two_level && !world_tree && !empty_scene) {
            const uint32_t mesh0 = s->parameterized_meshes[s->instances[0].parameterized_mesh_id].mesh_id;
            root = insts[0].blas_root;
            n_top = blas_top[mesh0];
            root_frame = blas_frame[mesh0];
        }
    }

    void linearise_textures()
    {
        // textures: sRGB -> linear in 8 bits, on the host, like the reference (render_embree.cpp:90-104)
        std::vector<TexRec> &tex = ps->tex;
        tex.assign(s->n_textures, TexRec{});
        std::vector<uint8_t> &texels = ps->texels;
        {
            size_t total = 0;
            for (uint32_t t = 0; t < s->n_textures; ++t) {
                const crt_image_desc &im = s->textures[t];
                if (im.width <= 0 || im.height <= 0 || im.channels < 1 || im.channels > 4 || !im.data) {
                    throw std::runtime_error("bad texture");
                }
                total = (total + 15) / 16 * 16 + (size_t)tex_tiled_texels(im.width, im.height) * im.channels;
            }
            texels.reserve(total + 16);
        }
        uint8_t lut[256];
        for (int v = 0; v < 256; ++v) {
            const float x = srgb_to_linear(v / 255.f);
            lut[v] = (uint8_t)std::min(std::max(x * 255.f, 0.f), 255.f);
        }
        // lay the textures out first, then copy + linearise them in parallel (1 GB of texels on a San-Miguel-class
        // scene: 0.35 s on one core, and the only second-scale host phase left once the BVH comes from the device)
        {
            size_t total = 0;
            for (uint32_t t = 0; t < s->n_textures; ++t) {
                const crt_image_desc &im = s->textures[t];
                total = (total + 15) / 16 * 16;
                TexRec r;
                std::memset(&r, 0, sizeof(r));
                r.width = im.width;
                r.height = im.height;
                r.channels = im.channels;
                if (total / 16 > 0xffffffffull) {
                    throw std::runtime_error("more than 64 GB of texels");
                }
                r.offset16 = (uint32_t)(total / 16);
                tex[t] = r;
                if (tex_tiled_texels(im.width, im.height) > 0xffffffffull) {
                    throw std::runtime_error("texture too large");
                }
                total += (size_t)tex_tiled_texels(im.width, im.height) * im.channels;
            }
            texels.assign(total, 0);
            std::atomic<uint32_t> next_tex{0};
            auto work = [&]() {
                for (uint32_t t = next_tex.fetch_add(1); t < s->n_textures; t = next_tex.fetch_add(1)) {
                    const crt_image_desc &im = s->textures[t];
                    // rows of texels -> 8 x 4 tiles (crt_types.h tex_slot), linearising the colour channels on the way
                    uint8_t *p = texels.data() + (size_t)tex[t].offset16 * 16;
                    const uint8_t *src = static_cast<const uint8_t *>(im.data);
                    const int convert_channels = im.color_space == CRT_COLORSPACE_SRGB ? std::min(3, im.channels) : 0;
                    const uint32_t tiles_x = tex_tiles_x(im.width);
                    for (int32_t y = 0; y < im.height; ++y) {
                        const uint32_t row = tex_row_part(tiles_x, y);
                        const uint8_t *s_row = src + (size_t)y * im.width * im.channels;
                        for (int32_t x = 0; x < im.width; ++x) {
                            uint8_t *d = p + (size_t)(row + tex_col_part(x)) * im.channels;
                            for (int c = 0; c < im.channels; ++c) {
                                const uint8_t v = s_row[(size_t)x * im.channels + c];
                                d[c] = c < convert_channels ? lut[v] : v;
                            }
                        }
                    }
                }
            };
            std::vector<std::thread> pool;
            const int n_workers = std::max(1, std::min<int>(n_threads, (int)s->n_textures));
            for (int w = 1; w < n_workers; ++w) {
                pool.emplace_back(work);
            }
            work();
            for (std::thread &th : pool) {
                th.join();
            }
        }
    }

    void copy_tables()
    {
        std::vector<float> &materials = ps->materials;
        materials.assign((size_t)s->n_materials * 16, 0.f);
        std::memcpy(materials.data(), s->materials, materials.size() * sizeof(float));
        std::vector<float> &lights = ps->lights;
        lights.assign((size_t)s->n_lights * 20, 0.f);
        std::memcpy(lights.data(), s->lights, lights.size() * sizeof(float));
    }
};

} // namespace

// Host cores this process may use: affinity mask, capped by the cgroup CPU quota (a container
// with 16 of 128 cores must not start 128 build threads), overridable with CRT_HIP_BUILD_THREADS.
int host_threads()
{
    if (const char *e = std::getenv("CRT_HIP_BUILD_THREADS")) {
        const int v = std::atoi(e);
        if (v > 0) {
            return v;
        }
    }
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
        n = std::max(1, std::min(n, CPU_COUNT(&set)));
    }
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32];
        long long period = 0;
        if (std::fscanf(f, "%31s %lld", quota, &period) == 2 && std::strcmp(quota, "max") != 0 && period > 0) {
            n = std::max(1, std::min(n, (int)(std::atoll(quota) / period)));
        }
        std::fclose(f);
    }
    return n;
}

void prepare_scene(const crt_scene_desc *s, crt_hip_prepared_scene *ps, int n_threads, int build_device, int reinsert_passes, bool tree_only)
{
    ScenePreparer(s, ps, n_threads, build_device, reinsert_passes, tree_only).run();
}

} // namespace crt

namespace crt {
namespace {
thread_local std::string g_global_error;
}
void set_global_error(const std::string &msg) { g_global_error = msg; }
const std::string &global_error() { return g_global_error; }
} // namespace crt

using namespace crt;

namespace {
int fail_global(int code, const std::string &msg)
{
    set_global_error(msg);
    return code;
}
} // namespace

// Flat serialisation of a prepared scene: header, then the arrays back to back. Meant for a tmpfs
// path (/dev/shm) shared by the ranks of one node; same build, same machine -- not an exchange format.
namespace {
constexpr uint64_t PREP_MAGIC = 0x3630505250545243ull; // "CRTPRP06" (02: tiled texels; 03: grafted world instance; 04: textured flag on material ids; 05: 64-byte leaf slots; 06: packed nodes)
struct PrepHeader {
    uint64_t magic, abi;
    uint64_t n_nodes, n_tris, n_insts, n_matids, n_materials, n_lights_f, n_tex, n_texels;
    QFrame root_frame;
    int32_t root;
    uint32_t two_level, n_top, n_lights, n_instances, spp, stack_need;
    int32_t world_inst;
};
template <typename T> bool prep_put(FILE *f, const std::vector<T> &v) { return v.empty() || std::fwrite(v.data(), sizeof(T), v.size(), f) == v.size(); }
template <typename T> bool prep_get(FILE *f, std::vector<T> &v, uint64_t n)
{
    v.resize(n);
    return n == 0 || std::fread(v.data(), sizeof(T), n, f) == n;
}
} // namespace

extern "C" {

crt_hip_prepared_scene *crt_hip_prepare_scene_on(const crt_scene_desc *scene, int n_threads, int build_device)
{
    std::unique_ptr<crt_hip_prepared_scene> ps(new crt_hip_prepared_scene);
    try {
        if (build_device >= crt_hip_device_count()) {
            throw std::runtime_error("prepare_scene: no such HIP device to build on");
        }
        prepare_scene(scene, ps.get(), n_threads > 0 ? n_threads : host_threads(), build_device);
    } catch (const std::exception &e) {
        set_global_error(e.what());
        return nullptr;
    }
    return ps.release();
}

crt_hip_prepared_scene *crt_hip_prepare_scene(const crt_scene_desc *scene, int n_threads)
{
    std::unique_ptr<crt_hip_prepared_scene> ps(new crt_hip_prepared_scene);
    try {
        prepare_scene(scene, ps.get(), n_threads > 0 ? n_threads : host_threads());
    } catch (const std::exception &e) {
        set_global_error(e.what());
        return nullptr;
    }
    return ps.release();
}

void crt_hip_free_prepared_scene(crt_hip_prepared_scene *ps) { delete ps; }

int32_t crt_hip_prepared_scene_world_instance(const crt_hip_prepared_scene *ps) { return ps ? ps->world_inst : -1; }
int crt_hip_prepared_scene_info(const crt_hip_prepared_scene *ps, uint64_t *n_nodes, uint64_t *n_tris,
                                uint64_t *n_instances, int32_t *two_level, float *root_frame, int32_t *root,
                                uint32_t *n_top_nodes, uint32_t *stack_need, double *build_ms)
{
    if (!ps) {
        return fail_global(CRT_HIP_EINVAL, "prepared scene is null");
    }
    if (n_nodes) {
        *n_nodes = ps->nodes.size();
    }
    if (n_tris) {
        *n_tris = ps->slots.size();
    }
    if (n_instances) {
        *n_instances = ps->insts.size();
    }
    if (two_level) {
        *two_level = (int32_t)ps->two_level;
    }
    if (root_frame) {
        std::memcpy(root_frame, &ps->root_frame, sizeof(QFrame));
    }
    if (root) {
        *root = ps->root;
    }
    if (n_top_nodes) {
        *n_top_nodes = ps->n_top;
    }
    if (stack_need) {
        *stack_need = ps->stack_need;
    }
    if (build_ms) {
        *build_ms = ps->build_ms;
    }
    return CRT_HIP_OK;
}

int crt_hip_prepared_scene_copy(const crt_hip_prepared_scene *ps, void *nodes, void *tris, void *instances)
{
    if (!ps) {
        return fail_global(CRT_HIP_EINVAL, "prepared scene is null");
    }
    if (nodes) {
        std::memcpy(nodes, ps->nodes.data(), ps->nodes.size() * sizeof(PNode));
    }
    if (tris) {
        std::memcpy(tris, ps->slots.data(), ps->slots.size() * sizeof(LeafSlot));
    }
    if (instances) {
        std::memcpy(instances, ps->insts.data(), ps->insts.size() * sizeof(InstanceRec));
    }
    return CRT_HIP_OK;
}

int crt_hip_prepared_scene_set_spp(crt_hip_prepared_scene *ps, uint32_t samples_per_pixel)
{
    if (!ps || samples_per_pixel == 0) {
        return fail_global(CRT_HIP_EINVAL, "prepared_scene_set_spp: bad arguments");
    }
    ps->spp = samples_per_pixel;
    return CRT_HIP_OK;
}


int crt_hip_save_prepared_scene(const crt_hip_prepared_scene *ps, const char *path)
{
    if (!ps || !path) {
        return fail_global(CRT_HIP_EINVAL, "save_prepared_scene: bad arguments");
    }
    FILE *f = std::fopen(path, "wb");
    if (!f) {
        return fail_global(CRT_HIP_EINVAL, std::string("cannot write ") + path);
    }
    PrepHeader h{};
    h.magic = PREP_MAGIC;
    h.abi = CRT_HIP_ABI_VERSION;
    h.n_nodes = ps->nodes.size();
    h.n_tris = ps->slots.size();
    h.n_insts = ps->insts.size();
    h.n_matids = ps->material_ids.size();
    h.n_materials = ps->materials.size();
    h.n_lights_f = ps->lights.size();
    h.n_tex = ps->tex.size();
    h.n_texels = ps->texels.size();
    h.root_frame = ps->root_frame;
    h.root = ps->root;
    h.two_level = ps->two_level;
    h.n_top = ps->n_top;
    h.n_lights = ps->n_lights;
    h.n_instances = ps->n_instances;
    h.spp = ps->spp;
    h.stack_need = ps->stack_need;
    h.world_inst = ps->world_inst;
    const bool ok = std::fwrite(&h, sizeof(h), 1, f) == 1 && prep_put(f, ps->nodes) && prep_put(f, ps->slots) && prep_put(f, ps->tri_uvs) &&
                    prep_put(f, ps->insts) && prep_put(f, ps->material_ids) && prep_put(f, ps->materials) && prep_put(f, ps->lights) &&
                    prep_put(f, ps->tex) && prep_put(f, ps->texels);
    const bool closed = std::fclose(f) == 0;
    return ok && closed ? CRT_HIP_OK : fail_global(CRT_HIP_EINVAL, std::string("short write to ") + path);
}

crt_hip_prepared_scene *crt_hip_load_prepared_scene(const char *path)
{
    // Nothing may crash or throw across the C ABI: a truncated or foreign file is refused with a message. The counts of
    // the header are checked against the file's size BEFORE anything is allocated from them, and against each other.
    FILE *f = path ? std::fopen(path, "rb") : nullptr;
    if (!f) {
        set_global_error(std::string("cannot read ") + (path ? path : "(null)"));
        return nullptr;
    }
    auto refuse = [&](const char *why) -> crt_hip_prepared_scene * {
        std::fclose(f);
        set_global_error(std::string("not a prepared scene of this build (") + why + "): " + path);
        return nullptr;
    };
    try {
        std::unique_ptr<crt_hip_prepared_scene> ps(new crt_hip_prepared_scene);
        PrepHeader h{};
        if (std::fread(&h, sizeof(h), 1, f) != 1 || h.magic != PREP_MAGIC || h.abi != CRT_HIP_ABI_VERSION) {
            return refuse("header");
        }
        std::fseek(f, 0, SEEK_END);
        const uint64_t file_size = (uint64_t)std::ftell(f);
        std::fseek(f, (long)sizeof(h), SEEK_SET);
        const uint64_t limit = file_size; // no array can hold more elements than the file has bytes
        for (uint64_t c : {h.n_nodes, h.n_tris, h.n_insts, h.n_matids, h.n_materials, h.n_lights_f, h.n_tex, h.n_texels}) {
            if (c > limit) {
                return refuse("counts");
            }
        }
        const uint64_t expect = sizeof(h) + h.n_nodes * sizeof(PNode) + h.n_tris * sizeof(LeafSlot) + h.n_tris * 2 * TRI_UV_STRIDE * sizeof(float) +
                                h.n_insts * sizeof(InstanceRec) + h.n_matids * 4 + h.n_materials * 4 + h.n_lights_f * 4 +
                                h.n_tex * sizeof(TexRec) + h.n_texels;
        if (expect != file_size) {
            return refuse("size");
        }
        if (h.n_insts != h.n_instances || h.n_instances == 0 || h.two_level > LEVELS_WORLD_TREE || h.n_nodes == 0 ||
            (h.n_tris == 0 && h.n_nodes != 1) /* only the empty scene's one-node tree has no leaf slots (make_empty_tree) */ ||
            h.root < 0 || (uint64_t)h.root >= h.n_nodes || (uint64_t)h.root + h.n_top > h.n_nodes || h.world_inst < -1 ||
            (h.world_inst >= 0 && (uint32_t)h.world_inst >= h.n_instances) || h.n_lights == 0 || h.n_lights_f != 20ull * h.n_lights ||
            h.n_materials == 0 || h.n_materials % 16 != 0 || h.spp == 0 || h.n_tris >= (1ull << 28)) {
            return refuse("fields");
        }
        bool ok = prep_get(f, ps->nodes, h.n_nodes) && prep_get(f, ps->slots, h.n_tris) && prep_get(f, ps->tri_uvs, (uint64_t)TRI_UV_STRIDE * 2 * h.n_tris) &&
                  prep_get(f, ps->insts, h.n_insts) && prep_get(f, ps->material_ids, h.n_matids) && prep_get(f, ps->materials, h.n_materials) &&
                  prep_get(f, ps->lights, h.n_lights_f) && prep_get(f, ps->tex, h.n_tex) && prep_get(f, ps->texels, h.n_texels);
        if (!ok) {
            return refuse("short read");
        }
        // what the kernels index with: material ids, texture extents
        const uint64_t n_mat = h.n_materials / 16;
        for (uint32_t id : ps->material_ids) {
            if ((id & ~MATERIAL_TEXTURED) >= n_mat) {
                return refuse("material id");
            }
        }
        for (const TexRec &t : ps->tex) {
            if (t.width <= 0 || t.height <= 0 || t.channels < 1 || t.channels > 4 ||
                (uint64_t)t.offset16 * 16 + tex_tiled_texels(t.width, t.height) * (uint64_t)t.channels > h.n_texels) {
                return refuse("texture");
            }
        }
        for (const InstanceRec &r : ps->insts) {
            if (r.blas_root < 0 || (uint64_t)r.blas_root >= h.n_nodes || (uint64_t)r.mat_base >= h.n_matids) {
                return refuse("instance");
            }
        }
        std::fclose(f);
        ps->root_frame = h.root_frame;
        ps->root = h.root;
        ps->two_level = h.two_level;
        ps->n_top = h.n_top;
        ps->n_lights = h.n_lights;
        ps->n_instances = h.n_instances;
        ps->spp = h.spp;
        ps->stack_need = h.stack_need;
        ps->world_inst = h.world_inst;
        return ps.release();
    } catch (const std::exception &e) { // bad_alloc and friends
        std::fclose(f);
        set_global_error(std::string("load_prepared_scene: ") + e.what());
        return nullptr;
    }
}

} // extern "C"
