// crt_core.cpp — the C-ABI of include/crt_hip.h: context, scene upload, frame loop.
//
// Host half of the MI355X backend, mirroring what RenderEmbree does around its kernels
// (reference backends/embree/render_embree.cpp:19-216): initialize -> framebuffer + accumulation
// state, set_scene -> geometry/BVH/textures/materials/lights resident in HBM, render -> view
// parameters, the wavefront launch sequence, timing and REPORT_RAY_STATS accounting.
// There is no CPU fallback anywhere in this file: no device, no context.

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
#include <string>
#include <thread>
#include <queue>
#include <vector>

#include "../../include/crt_hip.h"
#include "bvh_builder.h"
#include "bvh_device.h"
#include "crt_types.h"
#include "kernels.h"
#include "lbvh.h"
#include "wavefront.h"

using namespace crt;

namespace {

thread_local std::string g_create_error;

struct HipError {
    std::string msg;
};

#define HIP_CHECK(expr)                                                                                   \
    do {                                                                                                  \
        hipError_t err__ = (expr);                                                                        \
        if (err__ != hipSuccess) {                                                                        \
            throw HipError{std::string(#expr) + ": " + hipGetErrorString(err__)};                          \
        }                                                                                                 \
    } while (0)

struct DeviceBuffer {
    void *ptr = nullptr;
    size_t bytes = 0;
    void alloc(size_t n)
    {
        release();
        if (n == 0) {
            n = 16;
        }
        HIP_CHECK(hipMalloc(&ptr, n));
        bytes = n;
    }
    void release()
    {
        if (ptr) {
            (void)hipFree(ptr);
            ptr = nullptr;
            bytes = 0;
        }
    }
    template <typename T> T *as() const { return static_cast<T *>(ptr); }
    ~DeviceBuffer() { release(); }
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
};

template <typename T> void upload(DeviceBuffer &buf, const std::vector<T> &v, hipStream_t s)
{
    buf.alloc(v.size() * sizeof(T));
    if (!v.empty()) {
        HIP_CHECK(hipMemcpyAsync(buf.ptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipStreamSynchronize(s));
    }
}

// util/util.cpp:102-108 (std::pow(float, double): evaluated in double)
inline float srgb_to_linear(float x)
{
    if (x <= 0.04045f) {
        return x / 12.92f;
    }
    return (float)std::pow((double)((x + 0.055f) / 1.055f), 2.4);
}

// 4x4 inverse by cofactor expansion: stands in for glm::inverse (embree_utils.cpp:97).
bool invert4x4(const float m[16], float out[16])
{
    float inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] +
             m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] -
             m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] +
             m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] -
              m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] -
             m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] +
             m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] -
             m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] +
              m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] +
             m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] -
             m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] +
              m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] -
              m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] -
             m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] +
             m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] -
              m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] +
              m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0.f) {
        return false;
    }
    const float r = 1.f / det;
    for (int i = 0; i < 16; ++i) {
        out[i] = inv[i] * r;
    }
    return true;
}

bool is_identity(const float m[16])
{
    static const float id[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    return std::memcmp(m, id, sizeof(id)) == 0;
}

struct Vec3 {
    float x, y, z;
};
inline Vec3 sub(Vec3 a, Vec3 b) { return Vec3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 mul(Vec3 a, float s) { return Vec3{a.x * s, a.y * s, a.z * s}; }
inline Vec3 cross(Vec3 a, Vec3 b) { return Vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline Vec3 normalize(Vec3 v) // glm::normalize = v * inversesqrt(dot(v, v))
{
    const float c = 1.f / std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    return Vec3{v.x * c, v.y * c, v.z * c};
}

constexpr int MAX_TOP_NODES_HOST = CRT_MAX_TOP_NODES; // kernels.h
// Scenes with several instances whose triangles fit the budget get a world tree unless CRT_HIP_LEVELS says otherwise.
#ifndef CRT_WORLD_TREE_DEFAULT
#define CRT_WORLD_TREE_DEFAULT 1
#endif

} // namespace

struct crt_hip_ctx {
    int device = 0;
    uint32_t flags = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // Occlusion rays of bounce b and closest-hit rays of bounce b+1 are independent: with overlap on (the default),
    // the occlusion launch goes to aux_stream so that its waves fill the CUs the other launch's tail
    // leaves idle (and vice versa). It needs its own traversal-stack spill slab.
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool overlap = false;
    int n_cus = 256;
    std::string name, err;
    int rank = 0, world = 1;

    // framebuffer state (RenderEmbree::initialize)
    int width = 0, height = 0, ntx = 0, nty = 0;
    std::vector<uint32_t> tile_ids; // tiles this rank renders
    uint32_t n_local_tiles = 0, n_tiles_padded = 0;
    DeviceBuffer d_tile_ids, d_accum, d_tile_fb[2], d_img, d_ray_counts;
    // The compact RGBA8 tile buffer is double-buffered by frame parity: the gather of frame f (RCCL, on
    // another stream) may still read its buffer while frame f+1 is traced and accumulated into the other.
    int tile_fb_last = 0; // buffer the last rendered frame wrote
    std::vector<uint32_t> img;
    uint32_t frame_id = 0;

    // scene (RenderEmbree::set_scene)
    bool has_scene = false;
    uint32_t spp = 1;
    SceneView sv{};
    DeviceBuffer d_spill, d_tri_uvs;
    DeviceBuffer d_nodes, d_tris, d_instances, d_material_ids, d_materials, d_textures, d_texels, d_lights;
    uint64_t n_nodes = 0, n_tris = 0;
    uint32_t stack_need = 0; // traversal-stack entries the deepest path of this scene's BVH can need

    // wavefront state
    uint64_t capacity = 0; // paths per pass
    DeviceBuffer d_queue_mem, d_pc;
    PathQueue q[2]{};
    HitBuf hits{};
    ShadowQueueA sa{};
    ShadowQueueB sb{};
    float4 *radiance = nullptr;
    PassCounters *h_pc = nullptr; // pinned, one per pass
    uint32_t h_pc_slots = 0;
    std::vector<hipEvent_t> events;

    ~crt_hip_ctx()
    {
        for (hipEvent_t e : events) {
            (void)hipEventDestroy(e);
        }
        if (h_pc) {
            (void)hipHostFree(h_pc);
        }
        if (aux_stream) {
            (void)hipStreamDestroy(aux_stream);
            (void)hipEventDestroy(ev_fork);
            (void)hipEventDestroy(ev_join);
        }
        if (own_stream) {
            (void)hipStreamDestroy(own_stream);
        }
    }
    LaunchCfg cfg() const { return LaunchCfg{stream, n_cus, (flags & CRT_HIP_FLAG_COUNTERS) != 0}; }
};

namespace {

int fail(crt_hip_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) {
        ctx->err = msg;
    } else {
        g_create_error = msg;
    }
    return code;
}

template <typename F> int guarded(crt_hip_ctx *ctx, F &&f)
{
    if (!ctx) {
        return fail(nullptr, CRT_HIP_EINVAL, "null context");
    }
    try {
        HIP_CHECK(hipSetDevice(ctx->device));
        return f();
    } catch (const HipError &e) {
        return fail(ctx, CRT_HIP_EDEVICE, e.msg);
    } catch (const std::exception &e) {
        return fail(ctx, CRT_HIP_EINVAL, e.what());
    }
}

uint64_t default_capacity()
{
    if (const char *s = std::getenv("CRT_HIP_MAX_PATHS")) {
        const long long v = std::atoll(s);
        if (v > 0) {
            return (uint64_t)v;
        }
    }
    return 32ull << 20; // 32 Mi paths ~ 7.7 GiB of queue state, a sliver of 288 GB
}

// carve the SoA queues out of one allocation
void setup_queues(crt_hip_ctx *c)
{
    const uint64_t total_slots = (uint64_t)c->n_local_tiles * TILE_PIXELS;
    uint64_t cap = std::min<uint64_t>(default_capacity(), total_slots * c->spp);
    cap = std::min<uint64_t>(cap, 1ull << PATH_ID_BITS); // a path's index shares its queue word with its ray count (crt_types.h)
    const uint64_t slots_per_pass = std::max<uint64_t>(64, (cap / c->spp) / 64 * 64);
    cap = slots_per_pass * c->spp;
    if (cap > (1ull << PATH_ID_BITS)) {
        throw std::runtime_error("samples_per_pixel too large: 64 pixels x spp paths must fit one pass of 2^27 paths");
    }
    c->capacity = cap;
    const size_t n_fields = 2 * 11 + 9 + 12 + 18 + 4;
    c->d_queue_mem.alloc(n_fields * cap * sizeof(float));
    uint32_t *base = c->d_queue_mem.as<uint32_t>();
    size_t k = 0;
    auto f32 = [&]() { return reinterpret_cast<float *>(base + (k++) * cap); };
    auto u32 = [&]() { return base + (k++) * cap; };
    auto i32 = [&]() { return reinterpret_cast<int32_t *>(base + (k++) * cap); };
    for (int qi = 0; qi < 2; ++qi) {
        for (int a = 0; a < 3; ++a) {
            c->q[qi].o[a] = f32();
        }
        for (int a = 0; a < 3; ++a) {
            c->q[qi].d[a] = f32();
        }
        c->q[qi].path = u32();
        c->q[qi].rng = u32();
        for (int a = 0; a < 3; ++a) {
            c->q[qi].tp[a] = f32();
        }
    }
    c->hits.t = f32();
    c->hits.u = f32();
    c->hits.v = f32();
    c->hits.tri = i32();
    c->hits.inst = i32();
    for (int a = 0; a < 3; ++a) {
        c->hits.ng[a] = f32();
    }
    c->hits.mat = u32();
    for (int a = 0; a < 3; ++a) {
        c->sa.o[a] = f32();
    }
    for (int a = 0; a < 3; ++a) {
        c->sa.d[a] = f32();
    }
    c->sa.tmax = f32();
    for (int a = 0; a < 3; ++a) {
        c->sa.c[a] = f32();
    }
    c->sa.path = u32();
    c->sa.bslot = i32();
    for (int a = 0; a < 3; ++a) {
        c->sb.o[a] = f32();
    }
    for (int a = 0; a < 3; ++a) {
        c->sb.d[a] = f32();
    }
    c->sb.tmax = f32();
    for (int a = 0; a < 3; ++a) {
        c->sb.ca[a] = f32();
    }
    for (int a = 0; a < 3; ++a) {
        c->sb.cb[a] = f32();
    }
    for (int a = 0; a < 3; ++a) {
        c->sb.tp[a] = f32();
    }
    c->sb.path = u32();
    c->sb.reserved = i32();
    c->radiance = reinterpret_cast<float4 *>(base + k * cap);
    c->d_pc.alloc(sizeof(PassCounters));
    const uint64_t total_paths = total_slots * c->spp;
    const uint32_t n_pass = (uint32_t)((total_paths + cap - 1) / cap);
    if (c->h_pc) {
        (void)hipHostFree(c->h_pc);
        c->h_pc = nullptr;
    }
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&c->h_pc), sizeof(PassCounters) * n_pass));
    c->h_pc_slots = n_pass;
}

hipEvent_t get_event(crt_hip_ctx *c, size_t i)
{
    while (c->events.size() <= i) {
        hipEvent_t e;
        HIP_CHECK(hipEventCreate(&e));
        c->events.push_back(e);
    }
    return c->events[i];
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

} // namespace

extern "C" {

int crt_hip_abi_version(void) { return CRT_HIP_ABI_VERSION; }

int crt_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        return 0;
    }
    return n;
}

crt_hip_ctx *crt_hip_create(int device_id, uint32_t flags)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_create_error = "no HIP device available (this backend has no CPU fallback)";
        return nullptr;
    }
    if (device_id < 0 || device_id >= n) {
        g_create_error = "device id out of range";
        return nullptr;
    }
    std::unique_ptr<crt_hip_ctx> c(new crt_hip_ctx);
    try {
        c->device = device_id;
        c->flags = flags;
        HIP_CHECK(hipSetDevice(device_id));
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
        c->n_cus = prop.multiProcessorCount;
        c->name = std::string("HIP wavefront path tracer (") + prop.name + ", " + prop.gcnArchName + ")";
        HIP_CHECK(hipStreamCreate(&c->own_stream));
        c->overlap = true; // CRT_HIP_OVERLAP=0: strictly serial launches
        if (const char *e = std::getenv("CRT_HIP_OVERLAP")) {
            c->overlap = std::atoi(e) != 0;
        }
        if (c->overlap) {
            HIP_CHECK(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
            HIP_CHECK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
        }
        c->stream = c->own_stream;
    } catch (const HipError &err) {
        g_create_error = err.msg;
        return nullptr;
    }
    return c.release();
}

void crt_hip_destroy(crt_hip_ctx *ctx)
{
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        delete ctx;
    }
}

const char *crt_hip_last_error(const crt_hip_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }
const char *crt_hip_name(const crt_hip_ctx *ctx) { return ctx ? ctx->name.c_str() : ""; }
uint32_t crt_hip_frame_id(const crt_hip_ctx *ctx) { return ctx ? ctx->frame_id : 0; }

int crt_hip_set_stream(crt_hip_ctx *ctx, void *hip_stream)
{
    return guarded(ctx, [&]() -> int {
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
        return CRT_HIP_OK;
    });
}

int crt_hip_set_partition(crt_hip_ctx *ctx, int rank, int world)
{
    return guarded(ctx, [&]() -> int {
        if (world < 1 || rank < 0 || rank >= world) {
            return fail(ctx, CRT_HIP_EINVAL, "bad rank/world");
        }
        if (ctx->width != 0) {
            return fail(ctx, CRT_HIP_ESTATE, "set_partition must precede initialize");
        }
        ctx->rank = rank;
        ctx->world = world;
        return CRT_HIP_OK;
    });
}

// RenderEmbree::initialize (render_embree.cpp:38-56)
int crt_hip_initialize(crt_hip_ctx *ctx, int fb_width, int fb_height)
{
    return guarded(ctx, [&]() -> int {
        if (fb_width <= 0 || fb_height <= 0) {
            return fail(ctx, CRT_HIP_EINVAL, "bad framebuffer size");
        }
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        ctx->frame_id = 0;
        ctx->width = fb_width;
        ctx->height = fb_height;
        ctx->ntx = fb_width / TILE + (fb_width % TILE != 0 ? 1 : 0);
        ctx->nty = fb_height / TILE + (fb_height % TILE != 0 ? 1 : 0);
        const uint32_t ntiles = (uint32_t)(ctx->ntx * ctx->nty);
        ctx->tile_ids.clear();
        for (uint32_t t = (uint32_t)ctx->rank; t < ntiles; t += (uint32_t)ctx->world) {
            ctx->tile_ids.push_back(t);
        }
        ctx->n_local_tiles = (uint32_t)ctx->tile_ids.size();
        ctx->n_tiles_padded = (ntiles + ctx->world - 1) / ctx->world;
        ctx->img.assign((size_t)fb_width * fb_height, 0u);
        std::vector<uint32_t> ids = ctx->tile_ids;
        if (ids.empty()) {
            ids.push_back(0);
        }
        upload(ctx->d_tile_ids, ids, ctx->stream);
        const size_t slots = (size_t)std::max(1u, ctx->n_local_tiles) * TILE_PIXELS;
        ctx->d_accum.alloc(slots * sizeof(float4));
        ctx->d_ray_counts.alloc(slots * sizeof(uint32_t));
        for (DeviceBuffer &b : ctx->d_tile_fb) {
            b.alloc((size_t)std::max(1u, ctx->n_tiles_padded) * TILE_PIXELS * sizeof(uint32_t));
        }
        ctx->d_img.alloc((size_t)fb_width * fb_height * sizeof(uint32_t));
        HIP_CHECK(hipMemsetAsync(ctx->d_accum.ptr, 0, ctx->d_accum.bytes, ctx->stream));
        HIP_CHECK(hipMemsetAsync(ctx->d_ray_counts.ptr, 0, ctx->d_ray_counts.bytes, ctx->stream));
        for (DeviceBuffer &b : ctx->d_tile_fb) {
            HIP_CHECK(hipMemsetAsync(b.ptr, 0, b.bytes, ctx->stream));
        }
        HIP_CHECK(hipMemsetAsync(ctx->d_img.ptr, 0, ctx->d_img.bytes, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        ctx->capacity = 0; // queues are (re)sized lazily: they depend on spp too
        return CRT_HIP_OK;
    });
}

} // extern "C"

// ---- set_scene, host half: everything that does not need a device -------------------------------
// RenderEmbree::set_scene (render_embree.cpp:58-133, embree_utils.cpp:9-136): one BLAS per Mesh, the
// TLAS over the instances, sRGB -> linear textures in 8 bits, material and light tables -- as the flat
// arrays the kernels read. Split from the upload so that the GPUs of one node share ONE build
// (RenderHIP::set_scene prepares once and uploads to every context; bench.py's rank 0 prepares, saves
// to /dev/shm and the other ranks load).
struct crt_hip_prepared_scene {
    std::vector<QNode> nodes;
    std::vector<TriRec> tris;
    std::vector<float> tri_uvs; // TRI_UV_STRIDE per TriRec
    std::vector<InstanceRec> insts;
    std::vector<uint32_t> material_ids;
    std::vector<float> materials, lights;
    std::vector<TexRec> tex;
    std::vector<uint8_t> texels;
    QFrame root_frame{};
    int32_t root = 0;
    uint32_t two_level = 0, n_top = 0, n_lights = 0, n_instances = 0, spp = 1, stack_need = 0;
    int32_t world_inst = -1; // instance grafted into the top-level tree (prepare_scene), or -1
    double build_ms = 0.0;
};

namespace {

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
    // ~95 bytes per triangle (record, uv record, its share of the nodes): 2^27 triangles are 12.7 GB of a 288 GB part
    const char *cap = std::getenv("CRT_HIP_WORLD_TREE_MAX_TRIS");
    const uint64_t budget = cap != nullptr ? std::strtoull(cap, nullptr, 10) : (1ull << 27);
    return CRT_WORLD_TREE_DEFAULT && instanced_tris <= budget;
}

// build_device >= 0: meshes large enough to be worth it get their BLAS from the device builder
// (bvh_device.hip) on that HIP device; -1: the host SAH builder for everything.
void prepare_scene(const crt_scene_desc *s, crt_hip_prepared_scene *ps, int n_threads, int build_device = -1)
{
        const bool dbg = std::getenv("CRT_HIP_DEBUG") != nullptr;
        auto t_phase = std::chrono::high_resolution_clock::now();
        const auto t_begin = t_phase;
        auto phase = [&](const char *what) {
            const auto now = std::chrono::high_resolution_clock::now();
            if (dbg) {
                std::fprintf(stderr, "[crt_hip] set_scene %-22s %8.1f ms\n", what,
                             std::chrono::duration<double, std::milli>(now - t_phase).count());
            }
            t_phase = now;
        };
        check_scene(s);
        phase("validate");
        ps->spp = s->samples_per_pixel ? s->samples_per_pixel : 1;
        std::vector<QNode> &nodes = ps->nodes;
        std::vector<TriRec> &tris = ps->tris;
        std::vector<float> &tri_uvs = ps->tri_uvs;
        // one BLAS per Mesh (embree_utils.cpp:63-76)
        // Several instances: either a top-level tree over instances (two-level traversal, what Embree does:
        // embree_utils.cpp:90-129), or -- when the instanced triangles fit a memory budget, which on a 288 GB part is
        // nearly always -- ONE tree in world space over per-instance copies of the triangle records (crt_types.h
        // LEVELS_WORLD_TREE). CRT_HIP_LEVELS=two|world overrides the choice.
        uint64_t instanced_tris = 0;
        for (uint32_t i = 0; i < s->n_instances; ++i) {
            const crt_mesh_desc &md = s->meshes[s->parameterized_meshes[s->instances[i].parameterized_mesh_id].mesh_id];
            for (uint32_t k = 0; k < md.n_geometries; ++k) {
                instanced_tris += s->geometries[md.first_geometry + k].n_triangles;
            }
        }
        const bool world_tree = s->n_instances > 1 && world_tree_wanted(instanced_tris);
        const bool two_level = s->n_instances > 1 && !world_tree;
        std::vector<QFrame> blas_frame(s->n_meshes);
        std::vector<int32_t> blas_root(s->n_meshes);
        std::vector<Aabb> blas_bounds(s->n_meshes);
        std::vector<uint32_t> blas_top(s->n_meshes);
        // node 0.. are reserved for the TLAS when two_level so that the staged top levels are the TLAS's
        std::vector<BuiltBvh> built(s->n_meshes);
        std::vector<std::vector<QNode>> built_q(s->n_meshes); // device-built meshes: already quantised
        uint32_t blas_depth = 0, tlas_depth = 0; // levels of the wide trees
        // leaves of at most 2 triangles: with 4-wide nodes a leaf is one of four boxes tested per node
        // fetch, so small leaves are cheap to reach, and every triangle test saved is 3 lane requests
        static const int max_leaf = std::getenv("CRT_BVH_MAX_LEAF") ? std::atoi(std::getenv("CRT_BVH_MAX_LEAF")) : 2;
        const char *builder_env = std::getenv("CRT_BVH_BUILDER"); // "lbvh": the device algorithm, run on the host
        const bool host_lbvh = builder_env && std::strcmp(builder_env, "lbvh") == 0;
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
        int32_t world_inst = -1;
        uint32_t world_mesh = 0xffffffffu;
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
        // triangle record + the uvs of its three vertices (uv_buf[indices.x|y|z], render_embree.ispc:278-283) at position `at`
        auto place_tri = [&](const crt_mesh_desc &md, const TriRec &r, size_t at) {
            tris[at] = r;
            const crt_geometry_desc &gd = s->geometries[md.first_geometry + r.geom];
            if (gd.uvs) {
                for (int c = 0; c < 3; ++c) {
                    const uint32_t vi = gd.indices[3 * (size_t)r.prim + c];
                    tri_uvs[(size_t)TRI_UV_STRIDE * at + 2 * c] = gd.uvs[2 * (size_t)vi];
                    tri_uvs[(size_t)TRI_UV_STRIDE * at + 2 * c + 1] = gd.uvs[2 * (size_t)vi + 1];
                }
            }
        };
        for (uint32_t m = 0; m < s->n_meshes && !world_tree; ++m) {
            const crt_mesh_desc &md = s->meshes[m];
            if (build_device >= 0) {
                DeviceBuiltMesh db;
                bool built_on_device = false;
                try {
                    built_on_device = device_build_mesh(build_device, s->geometries + md.first_geometry, md.n_geometries,
                                                        (uint32_t)max_leaf, two_level ? 0 : MAX_TOP_NODES_HOST, db);
                } catch (const std::exception &e) { // e.g. out of device memory: the host builder still can
                    std::fprintf(stderr, "[crt_hip] %s -- building mesh %u on the host instead\n", e.what(), m);
                    (void)hipGetLastError();
                }
                if (built_on_device) {
                    const size_t tri_base = tris.size();
                    tris.insert(tris.end(), db.tris.begin(), db.tris.end());
                    tri_uvs.insert(tri_uvs.end(), db.tri_uvs.begin(), db.tri_uvs.end());
                    built_q[m] = std::move(db.nodes);
                    built[m].n_top = db.n_top;
                    built[m].max_depth = db.max_depth;
                    built[m].bounds = db.bounds;
                    blas_depth = std::max(blas_depth, db.max_depth);
                    blas_bounds[m] = db.bounds;
                    blas_frame[m] = db.frame;
                    blas_root[m] = (int32_t)tri_base;
                    continue;
                }
            }
            std::vector<TriRec> recs;
            std::vector<Aabb> boxes;
            {
                uint64_t n_mesh_tris = 0; // reserve ONCE: growing per geometry re-copies everything each time
                for (uint32_t k = 0; k < md.n_geometries; ++k) {
                    n_mesh_tris += s->geometries[md.first_geometry + k].n_triangles;
                }
                recs.reserve(n_mesh_tris);
                boxes.reserve(n_mesh_tris);
            }
            for (uint32_t k = 0; k < md.n_geometries; ++k) {
                const crt_geometry_desc &gd = s->geometries[md.first_geometry + k];
                for (uint64_t t = 0; t < gd.n_triangles; ++t) {
                    const float *v0 = gd.vertices + 3 * (size_t)gd.indices[3 * t];
                    const float *v1 = gd.vertices + 3 * (size_t)gd.indices[3 * t + 1];
                    const float *v2 = gd.vertices + 3 * (size_t)gd.indices[3 * t + 2];
                    TriRec r;
                    Aabb b;
                    for (int a = 0; a < 3; ++a) {
                        r.v0[a] = v0[a];
                        r.e1[a] = v0[a] - v1[a];
                        r.e2[a] = v2[a] - v0[a];
                        b.lo[a] = std::min(v0[a], std::min(v1[a], v2[a]));
                        b.hi[a] = std::max(v0[a], std::max(v1[a], v2[a]));
                    }
                    r.geom = k;
                    r.prim = (uint32_t)t;
                    r.pad = 0;
                    recs.push_back(r);
                    boxes.push_back(b);
                }
            }
            if (recs.empty()) {
                throw std::runtime_error("mesh without triangles");
            }
            built[m] = host_lbvh ? build_lbvh_host(boxes.data(), boxes.size(), max_leaf, two_level ? 0 : MAX_TOP_NODES_HOST)
                                 : build_bvh(boxes.data(), boxes.size(), max_leaf, 0, 0, false,
                                             two_level ? 0 : MAX_TOP_NODES_HOST, n_threads);
            blas_depth = std::max(blas_depth, built[m].max_depth);
            // triangles in leaf order
            const size_t tri_base = tris.size();
            tris.resize(tri_base + recs.size());
            tri_uvs.resize((size_t)TRI_UV_STRIDE * tris.size(), 0.f);
            for (size_t i = 0; i < recs.size(); ++i) {
                place_tri(md, recs[built[m].order[i]], tri_base + i);
            }
            blas_bounds[m] = built[m].bounds;
            blas_frame[m] = make_frame(built[m].bounds);
            // re-base the node / leaf references later, once the TLAS size is known
            blas_root[m] = (int32_t)tri_base; // temporarily: triangle base
        }

        phase("leaf-order triangles");
        // instances + TLAS (embree_utils.cpp:90-104, 121-129)
        std::vector<InstanceRec> &insts = ps->insts;
        insts.assign(s->n_instances, InstanceRec{});
        std::vector<uint32_t> &material_ids = ps->material_ids;
        std::vector<Aabb> inst_boxes(s->n_instances);
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
            if (world_tree) {
                continue; // no instance boxes: the tree is built over the triangles (below)
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
            // pad: the BLAS is walked with a transformed (rounded) ray
            const float ext = std::max(wb.hi[0] - wb.lo[0], std::max(wb.hi[1] - wb.lo[1], wb.hi[2] - wb.lo[2]));
            for (int a = 0; a < 3; ++a) {
                wb.lo[a] -= 1e-5f * ext;
                wb.hi[a] += 1e-5f * ext;
            }
            inst_boxes[i] = wb;
        }
        uint32_t n_top = 0;
        int32_t root = 0;
        QFrame root_frame{};
        if (world_tree) {
            // One record + one world-space box per (instance, triangle). The record is the mesh's own (object space: the
            // triangle test runs there, with the ray transformed like the reference transforms it, so t / u / v come out
            // bit for bit as in the two-level walk); the box bounds the transformed vertices, padded like an instance box
            // (the test ray is a rounded transform of the world ray) -- and quantisation rounds outward by >= 1 quantum
            // of the scene's extent on top of that.
            std::vector<TriRec> recs;
            std::vector<Aabb> boxes;
            recs.reserve(instanced_tris);
            boxes.reserve(instanced_tris);
            for (uint32_t i = 0; i < s->n_instances; ++i) {
                const crt_instance_desc &id = s->instances[i];
                const crt_mesh_desc &md = s->meshes[s->parameterized_meshes[id.parameterized_mesh_id].mesh_id];
                const float *m = id.transform;
                const bool ident = insts[i].identity != 0u;
                auto to_world = [&](const float *p, int a) {
                    return ident ? p[a] : m[a] * p[0] + m[4 + a] * p[1] + m[8 + a] * p[2] + m[12 + a];
                };
                float pad = 0.f;
                if (!ident) {
                    Aabb wb;
                    for (int a = 0; a < 3; ++a) {
                        wb.lo[a] = INFINITY;
                        wb.hi[a] = -INFINITY;
                    }
                    for (uint32_t k = 0; k < md.n_geometries; ++k) {
                        const crt_geometry_desc &gd = s->geometries[md.first_geometry + k];
                        for (uint64_t v = 0; v < gd.n_vertices; ++v) {
                            for (int a = 0; a < 3; ++a) {
                                const float w = to_world(gd.vertices + 3 * v, a);
                                wb.lo[a] = std::min(wb.lo[a], w);
                                wb.hi[a] = std::max(wb.hi[a], w);
                            }
                        }
                    }
                    pad = 1e-5f * std::max(wb.hi[0] - wb.lo[0], std::max(wb.hi[1] - wb.lo[1], wb.hi[2] - wb.lo[2]));
                    if (!(pad >= 0.f)) { // an instance without vertices that any triangle uses
                        pad = 0.f;
                    }
                }
                for (uint32_t k = 0; k < md.n_geometries; ++k) {
                    const crt_geometry_desc &gd = s->geometries[md.first_geometry + k];
                    for (uint64_t t = 0; t < gd.n_triangles; ++t) {
                        const float *v[3] = {gd.vertices + 3 * (size_t)gd.indices[3 * t], gd.vertices + 3 * (size_t)gd.indices[3 * t + 1],
                                             gd.vertices + 3 * (size_t)gd.indices[3 * t + 2]};
                        TriRec r;
                        Aabb b;
                        for (int a = 0; a < 3; ++a) {
                            r.v0[a] = v[0][a];
                            r.e1[a] = v[0][a] - v[1][a];
                            r.e2[a] = v[2][a] - v[0][a];
                            const float w0 = to_world(v[0], a), w1 = to_world(v[1], a), w2 = to_world(v[2], a);
                            b.lo[a] = std::min(w0, std::min(w1, w2)) - pad;
                            b.hi[a] = std::max(w0, std::max(w1, w2)) + pad;
                        }
                        r.geom = k;
                        r.prim = (uint32_t)t;
                        r.pad = (i << 1) | (ident ? 1u : 0u);
                        recs.push_back(r);
                        boxes.push_back(b);
                    }
                }
            }
            if (recs.empty()) {
                throw std::runtime_error("scene without triangles");
            }
            const int wt_leaf = std::min(max_leaf, 2); // the kernels' world-tree leaf step handles one or two triangles
            BuiltBvh tree = host_lbvh ? build_lbvh_host(boxes.data(), boxes.size(), wt_leaf, MAX_TOP_NODES_HOST)
                                      : build_bvh(boxes.data(), boxes.size(), wt_leaf, 0, 0, false, MAX_TOP_NODES_HOST, n_threads);
            boxes = std::vector<Aabb>();
            blas_depth = tree.max_depth;
            tris.resize(recs.size());
            tri_uvs.resize((size_t)TRI_UV_STRIDE * tris.size(), 0.f);
            for (size_t i = 0; i < recs.size(); ++i) {
                const TriRec &r = recs[tree.order[i]];
                const crt_instance_desc &id = s->instances[r.pad >> 1];
                place_tri(s->meshes[s->parameterized_meshes[id.parameterized_mesh_id].mesh_id], r, i);
            }
            root_frame = make_frame(tree.bounds);
            nodes.reserve(tree.nodes.size());
            for (const BvhNode &nd : tree.nodes) {
                nodes.push_back(quantise(nd, root_frame));
            }
            n_top = tree.n_top;
            root = 0;
            for (InstanceRec &r : insts) {
                r.frame = root_frame; // not read by the traversal of a world tree; blas_root stays a mesh id until the loop below
            }
        }
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
                if ((int32_t)i != world_inst) {
                    items.push_back(inst_boxes[i]);
                    item_ref.push_back(instance_leaf_ref(i));
                    item_is_cut.push_back(0);
                }
            }
            if (s->n_instances >= (1u << 28) - 1u) {
                throw std::runtime_error("too many instances for the 28-bit leaf reference");
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
            for (BvhNode nd : built[m].nodes) {
                for (int k = 0; k < BVH_WIDTH; ++k) {
                    if (nd.c[k] != EMPTY_CHILD) {
                        nd.c[k] = rebase(nd.c[k]);
                    }
                }
                nodes.push_back(quantise(nd, blas_frame[m]));
            }
            blas_root[m] = node_base;
            blas_top[m] = built[m].n_top;
            built[m] = BuiltBvh();
            built_q[m] = std::vector<QNode>();
        }
        // a ray's stack holds at most BVH_WIDTH-1 pending siblings per level of the path it is on,
        // plus the instance-exit sentinel. The LDS part of the stack is fixed; the HBM slab behind it is
        // sized from this number at upload, so no tree is "too deep" (the reference renders any scene)
        ps->stack_need = (BVH_WIDTH - 1) * (blas_depth + tlas_depth) + 2;
        if (tris.size() >= (1u << 28)) {
            throw std::runtime_error("too many triangles for the 28-bit leaf reference");
        }
        for (InstanceRec &r : insts) {
            r.blas_root = world_tree ? 0 : blas_root[r.blas_root];
        }
        if (world_inst >= 0) {
            insts[(size_t)world_inst].blas_root = 0; // never entered: its subtrees hang in the top-level tree (node 0 = its root)
        }
        ps->world_inst = world_inst;
        if (!two_level && !world_tree) {
            const uint32_t mesh0 = s->parameterized_meshes[s->instances[0].parameterized_mesh_id].mesh_id;
            root = insts[0].blas_root;
            n_top = blas_top[mesh0];
            root_frame = blas_frame[mesh0];
        }

        phase("TLAS + quantisation");
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
        std::vector<float> &materials = ps->materials;
        materials.assign((size_t)s->n_materials * 16, 0.f);
        std::memcpy(materials.data(), s->materials, materials.size() * sizeof(float));
        std::vector<float> &lights = ps->lights;
        lights.assign((size_t)s->n_lights * 20, 0.f);
        std::memcpy(lights.data(), s->lights, lights.size() * sizeof(float));

        phase("textures");
        ps->root_frame = root_frame;
        ps->root = root;
        ps->two_level = world_tree ? LEVELS_WORLD_TREE : two_level ? 1u : 0u;
        ps->n_top = n_top;
        ps->n_lights = s->n_lights;
        ps->n_instances = s->n_instances;
        ps->build_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t_begin).count();
}

// device half of set_scene: the prepared arrays -> HBM, SceneView, traversal-stack slab
void upload_scene(crt_hip_ctx *ctx, const crt_hip_prepared_scene &ps)
{
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->frame_id = 0;
    ctx->has_scene = false;
    ctx->spp = ps.spp;
    ctx->capacity = 0;
    upload(ctx->d_nodes, ps.nodes, ctx->stream);
    upload(ctx->d_tris, ps.tris, ctx->stream);
    upload(ctx->d_tri_uvs, ps.tri_uvs, ctx->stream);
    upload(ctx->d_instances, ps.insts, ctx->stream);
    upload(ctx->d_material_ids, ps.material_ids, ctx->stream);
    upload(ctx->d_materials, ps.materials, ctx->stream);
    upload(ctx->d_textures, ps.tex, ctx->stream);
    upload(ctx->d_texels, ps.texels, ctx->stream);
    upload(ctx->d_lights, ps.lights, ctx->stream);
    ctx->n_nodes = ps.nodes.size();
    ctx->n_tris = ps.tris.size();
    ctx->stack_need = ps.stack_need;

    SceneView &sv = ctx->sv;
    sv.nodes = ctx->d_nodes.as<QNode>();
    sv.root_frame = ps.root_frame;
    sv.tris = ctx->d_tris.as<TriRec>();
    sv.tri_uvs = ctx->d_tri_uvs.as<float>();
    sv.instances = ctx->d_instances.as<InstanceRec>();
    sv.material_ids = ctx->d_material_ids.as<uint32_t>();
    sv.materials = ctx->d_materials.as<float>();
    sv.textures = ctx->d_textures.as<TexRec>();
    sv.texels = ctx->d_texels.as<uint8_t>();
    sv.lights = ctx->d_lights.as<float>();
    sv.n_lights = ps.n_lights;
    sv.n_instances = ps.n_instances;
    // HBM part of the per-lane traversal stack: what the deepest path can need beyond the LDS part,
    // [wave of the persistent grid][depth][lane]
    const uint32_t lds_stack = traversal_lds_stack(ps.two_level);
    sv.spill_depth = std::max<uint32_t>(8u, ps.stack_need > lds_stack ? ps.stack_need - lds_stack : 0u);
    sv.spill_stride = traversal_grid_threads(ctx->n_cus);
    const size_t spill_words = (size_t)sv.spill_stride * sv.spill_depth;
    ctx->d_spill.alloc((ctx->overlap ? 2 : 1) * spill_words * sizeof(int32_t));
    sv.stack_spill = ctx->d_spill.as<int32_t>();
    sv.root = ps.root;
    sv.two_level = ps.two_level;
    sv.world_inst = ps.world_inst;
    sv.n_top_nodes = ps.n_top;
    ctx->has_scene = true;
}

} // namespace

// Flat serialisation of a prepared scene: header, then the arrays back to back. Meant for a tmpfs
// path (/dev/shm) shared by the ranks of one node; same build, same machine -- not an exchange format.
namespace {
constexpr uint64_t PREP_MAGIC = 0x3430505250545243ull; // "CRTPRP04" (02: tiled texels; 03: grafted world instance; 04: textured flag on material ids)
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
        g_create_error = e.what();
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
        g_create_error = e.what();
        return nullptr;
    }
    return ps.release();
}

void crt_hip_free_prepared_scene(crt_hip_prepared_scene *ps) { delete ps; }

int crt_hip_set_prepared_scene(crt_hip_ctx *ctx, const crt_hip_prepared_scene *ps)
{
    return guarded(ctx, [&]() -> int {
        if (!ps) {
            return fail(ctx, CRT_HIP_EINVAL, "prepared scene is null");
        }
        upload_scene(ctx, *ps);
        return CRT_HIP_OK;
    });
}

// RenderBackend::set_scene = prepare + upload
int crt_hip_set_scene(crt_hip_ctx *ctx, const crt_scene_desc *s)
{
    return guarded(ctx, [&]() -> int {
        crt_hip_prepared_scene ps;
        const char *where = std::getenv("CRT_HIP_BUILD"); // "device": BLAS of large meshes built on this context's GPU
        prepare_scene(s, &ps, host_threads(), where && std::strcmp(where, "device") == 0 ? ctx->device : -1);
        const auto t0 = std::chrono::high_resolution_clock::now();
        upload_scene(ctx, ps);
        if (std::getenv("CRT_HIP_DEBUG")) {
            std::fprintf(stderr, "[crt_hip] set_scene %-22s %8.1f ms\n", "upload",
                         std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count());
        }
        return CRT_HIP_OK;
    });
}

int32_t crt_hip_prepared_scene_world_instance(const crt_hip_prepared_scene *ps) { return ps ? ps->world_inst : -1; }
int32_t crt_hip_world_instance(crt_hip_ctx *ctx) { return ctx && ctx->has_scene ? ctx->sv.world_inst : -1; }

int crt_hip_prepared_scene_info(const crt_hip_prepared_scene *ps, uint64_t *n_nodes, uint64_t *n_tris,
                                uint64_t *n_instances, int32_t *two_level, float *root_frame, int32_t *root,
                                uint32_t *n_top_nodes, uint32_t *stack_need, double *build_ms)
{
    if (!ps) {
        return fail(nullptr, CRT_HIP_EINVAL, "prepared scene is null");
    }
    if (n_nodes) {
        *n_nodes = ps->nodes.size();
    }
    if (n_tris) {
        *n_tris = ps->tris.size();
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
        return fail(nullptr, CRT_HIP_EINVAL, "prepared scene is null");
    }
    if (nodes) {
        std::memcpy(nodes, ps->nodes.data(), ps->nodes.size() * sizeof(QNode));
    }
    if (tris) {
        std::memcpy(tris, ps->tris.data(), ps->tris.size() * sizeof(TriRec));
    }
    if (instances) {
        std::memcpy(instances, ps->insts.data(), ps->insts.size() * sizeof(InstanceRec));
    }
    return CRT_HIP_OK;
}

int crt_hip_prepared_scene_set_spp(crt_hip_prepared_scene *ps, uint32_t samples_per_pixel)
{
    if (!ps || samples_per_pixel == 0) {
        return fail(nullptr, CRT_HIP_EINVAL, "prepared_scene_set_spp: bad arguments");
    }
    ps->spp = samples_per_pixel;
    return CRT_HIP_OK;
}

int crt_hip_child_order(void) { return traversal_child_order(); }
uint32_t crt_hip_lds_stack_entries(int two_level) { return traversal_lds_stack((uint32_t)two_level); }

int crt_hip_save_prepared_scene(const crt_hip_prepared_scene *ps, const char *path)
{
    if (!ps || !path) {
        return fail(nullptr, CRT_HIP_EINVAL, "save_prepared_scene: bad arguments");
    }
    FILE *f = std::fopen(path, "wb");
    if (!f) {
        return fail(nullptr, CRT_HIP_EINVAL, std::string("cannot write ") + path);
    }
    PrepHeader h{};
    h.magic = PREP_MAGIC;
    h.abi = CRT_HIP_ABI_VERSION;
    h.n_nodes = ps->nodes.size();
    h.n_tris = ps->tris.size();
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
    const bool ok = std::fwrite(&h, sizeof(h), 1, f) == 1 && prep_put(f, ps->nodes) && prep_put(f, ps->tris) && prep_put(f, ps->tri_uvs) &&
                    prep_put(f, ps->insts) && prep_put(f, ps->material_ids) && prep_put(f, ps->materials) && prep_put(f, ps->lights) &&
                    prep_put(f, ps->tex) && prep_put(f, ps->texels);
    const bool closed = std::fclose(f) == 0;
    return ok && closed ? CRT_HIP_OK : fail(nullptr, CRT_HIP_EINVAL, std::string("short write to ") + path);
}

crt_hip_prepared_scene *crt_hip_load_prepared_scene(const char *path)
{
    FILE *f = path ? std::fopen(path, "rb") : nullptr;
    if (!f) {
        g_create_error = std::string("cannot read ") + (path ? path : "(null)");
        return nullptr;
    }
    std::unique_ptr<crt_hip_prepared_scene> ps(new crt_hip_prepared_scene);
    PrepHeader h{};
    bool ok = std::fread(&h, sizeof(h), 1, f) == 1 && h.magic == PREP_MAGIC && h.abi == CRT_HIP_ABI_VERSION;
    ok = ok && prep_get(f, ps->nodes, h.n_nodes) && prep_get(f, ps->tris, h.n_tris) && prep_get(f, ps->tri_uvs, (uint64_t)TRI_UV_STRIDE * h.n_tris) &&
         prep_get(f, ps->insts, h.n_insts) && prep_get(f, ps->material_ids, h.n_matids) && prep_get(f, ps->materials, h.n_materials) &&
         prep_get(f, ps->lights, h.n_lights_f) && prep_get(f, ps->tex, h.n_tex) && prep_get(f, ps->texels, h.n_texels);
    std::fclose(f);
    if (!ok) {
        g_create_error = std::string("not a prepared scene of this build: ") + path;
        return nullptr;
    }
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
}

// RenderEmbree::render (render_embree.cpp:135-216)
int crt_hip_render(crt_hip_ctx *ctx, const float pos[3], const float dir_[3], const float up_[3], float fovy,
                   int camera_changed, int readback, crt_render_stats *stats)
{
    return guarded(ctx, [&]() -> int {
        if (!ctx->has_scene || ctx->width == 0) {
            return fail(ctx, CRT_HIP_ESTATE, "render before initialize + set_scene");
        }
        if (ctx->capacity == 0) {
            setup_queues(ctx);
        }
        if (camera_changed) {
            ctx->frame_id = 0;
        }
        // render_embree.cpp:149-159; glm::radians(x) = x * 0.0174532925...
        const Vec3 dir{dir_[0], dir_[1], dir_[2]}, up{up_[0], up_[1], up_[2]};
        const float plane_y = 2.f * std::tan(0.5f * fovy * 0.01745329251994329576923690768489f);
        const float plane_x = plane_y * (float)ctx->width / (float)ctx->height;
        const Vec3 du = mul(normalize(cross(dir, up)), plane_x);
        const Vec3 ndv = normalize(cross(du, dir));
        const Vec3 dv = mul(Vec3{-ndv.x, -ndv.y, -ndv.z}, plane_y);
        const Vec3 tl = sub(sub(dir, mul(du, 0.5f)), mul(dv, 0.5f));
        ViewParams vp;
        vp.pos[0] = pos[0];
        vp.pos[1] = pos[1];
        vp.pos[2] = pos[2];
        vp.dir_du[0] = du.x;
        vp.dir_du[1] = du.y;
        vp.dir_du[2] = du.z;
        vp.dir_dv[0] = dv.x;
        vp.dir_dv[1] = dv.y;
        vp.dir_dv[2] = dv.z;
        vp.dir_top_left[0] = tl.x;
        vp.dir_top_left[1] = tl.y;
        vp.dir_top_left[2] = tl.z;
        vp.frame_id = ctx->frame_id;
        vp.fb_width = (uint32_t)ctx->width;
        vp.fb_height = (uint32_t)ctx->height;
        vp.spp = ctx->spp;
        vp.n_tiles_x = (uint32_t)ctx->ntx;

        const LaunchCfg cfg = ctx->cfg();
        const bool timing = (ctx->flags & CRT_HIP_FLAG_TIMING) != 0;
        const uint32_t *d_tiles = ctx->d_tile_ids.as<uint32_t>();
        const uint64_t total_slots = (uint64_t)ctx->n_local_tiles * TILE_PIXELS;
        const uint64_t slots_per_pass = ctx->capacity / ctx->spp;
        size_t ev = 0;
        struct Span {
            size_t a, b;
            int kind;
        };
        std::vector<Span> spans;
        // event pairs around each launch, recorded on the stream the launch goes to: with the overlapped
        // schedule the occlusion launches' spans live on the auxiliary stream and run concurrently with the
        // closest-hit spans of the next bounce (so the per-kind sums may add up to more than the frame time)
        auto mark = [&](int kind, hipStream_t on) {
            if (timing) {
                hipEvent_t e0 = get_event(ctx, ev), e1 = get_event(ctx, ev + 1);
                (void)e1;
                HIP_CHECK(hipEventRecord(e0, on));
                spans.push_back(Span{ev, ev + 1, kind});
                ev += 2;
            }
        };
        auto mark_end = [&](hipStream_t on) {
            if (timing) {
                HIP_CHECK(hipEventRecord(get_event(ctx, spans.back().b), on));
            }
        };

        const bool overlap = ctx->overlap;
        LaunchCfg aux_cfg = cfg;
        aux_cfg.stream = ctx->aux_stream;
        SceneView aux_sv = ctx->sv;
        if (overlap) {
            aux_sv.stack_spill += (size_t)aux_sv.spill_stride * aux_sv.spill_depth;
        }
        const auto t0 = std::chrono::high_resolution_clock::now();
        uint32_t pass = 0;
        for (uint64_t slot0 = 0; slot0 < total_slots; slot0 += slots_per_pass, ++pass) {
            const uint32_t n_slots = (uint32_t)std::min<uint64_t>(slots_per_pass, total_slots - slot0);
            const uint32_t n_paths = n_slots * ctx->spp;
            PassCounters *d_pc = ctx->d_pc.as<PassCounters>();
            HIP_CHECK(hipMemsetAsync(d_pc, 0, sizeof(PassCounters), ctx->stream));
            if (cfg.counters) { // atomicMin targets start at all-ones
                HIP_CHECK(hipMemsetAsync(d_pc->t_start, 0xff, 2 * MAX_PATH_DEPTH * sizeof(unsigned long long), ctx->stream));
            }
            mark(2, ctx->stream);
            launch_raygen(cfg, vp, d_tiles, (uint32_t)slot0, n_paths, ctx->q[0], ctx->radiance, d_pc);
            mark_end(ctx->stream);
            for (int b = 0; b < MAX_PATH_DEPTH; ++b) {
                if (!overlap || b == 0) {
                    mark(0, ctx->stream);
                    launch_trace_closest(cfg, ctx->sv, ctx->q[b & 1], ctx->hits, d_pc, b);
                    mark_end(ctx->stream);
                }
                mark(2, ctx->stream);
                launch_shade(cfg, ctx->sv, ctx->q[b & 1], ctx->hits, ctx->q[(b + 1) & 1], ctx->sa, ctx->sb,
                             ctx->radiance, d_pc, b);
                mark_end(ctx->stream);
                if (overlap) {
                    // shade(b) -> { shadow(b) on aux  ||  closest(b+1) on the main stream } -> shade(b+1)
                    HIP_CHECK(hipEventRecord(ctx->ev_fork, ctx->stream));
                    HIP_CHECK(hipStreamWaitEvent(ctx->aux_stream, ctx->ev_fork, 0));
                    mark(1, ctx->aux_stream);
                    launch_trace_shadow(aux_cfg, aux_sv, ctx->sa, ctx->sb, ctx->radiance, d_pc, b);
                    mark_end(ctx->aux_stream);
                    HIP_CHECK(hipEventRecord(ctx->ev_join, ctx->aux_stream));
                    if (b + 1 < MAX_PATH_DEPTH) {
                        mark(0, ctx->stream);
                        launch_trace_closest(cfg, ctx->sv, ctx->q[(b + 1) & 1], ctx->hits, d_pc, b + 1);
                        mark_end(ctx->stream);
                    }
                    HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
                } else {
                    mark(1, ctx->stream);
                    launch_trace_shadow(cfg, ctx->sv, ctx->sa, ctx->sb, ctx->radiance, d_pc, b);
                    mark_end(ctx->stream);
                }
            }
            mark(2, ctx->stream);
            launch_accumulate(cfg, vp, d_tiles, (uint32_t)slot0, n_slots, ctx->radiance, ctx->d_accum.as<float4>(),
                              ctx->d_tile_fb[ctx->frame_id & 1u].as<uint32_t>(), ctx->world == 1 ? ctx->d_img.as<uint32_t>() : nullptr,
                              ctx->d_ray_counts.as<uint32_t>());
            mark_end(ctx->stream);
            HIP_CHECK(hipMemcpyAsync(&ctx->h_pc[pass], d_pc, sizeof(PassCounters), hipMemcpyDeviceToHost,
                                     ctx->stream));
        }
        HIP_CHECK(hipGetLastError());
        if (readback && ctx->world == 1) {
            HIP_CHECK(hipMemcpyAsync(ctx->img.data(), ctx->d_img.ptr, ctx->img.size() * sizeof(uint32_t),
                                     hipMemcpyDeviceToHost, ctx->stream));
        }
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        const auto t1 = std::chrono::high_resolution_clock::now();

        crt_render_stats st;
        std::memset(&st, 0, sizeof(st));
        for (uint32_t p = 0; p < pass; ++p) {
            const PassCounters &pc = ctx->h_pc[p];
            for (int b = 0; b < MAX_PATH_DEPTH; ++b) {
                st.closest_rays += pc.n_queue[b];
                st.shadow_rays += (uint64_t)pc.n_shadow_a[b] + pc.n_shadow_b[b];
            }
            st.closest_nodes += pc.nodes_closest;
            st.closest_tris += pc.tris_closest;
            st.shadow_nodes += pc.nodes_shadow;
            st.shadow_tris += pc.tris_shadow;
        }
        st.rays = st.closest_rays + st.shadow_rays;
        if (std::getenv("CRT_HIP_DEBUG")) { // per-bounce queue sizes of the first pass
            const PassCounters &pc = ctx->h_pc[0];
            std::fprintf(stderr, "[crt_hip] frame %u worst closest ray: %u nodes, o (%.9g %.9g %.9g) d (%.9g %.9g %.9g) t %.9g\n",
                         ctx->frame_id, pc.max_ray_nodes, pc.worst_ray[0], pc.worst_ray[1], pc.worst_ray[2], pc.worst_ray[3],
                         pc.worst_ray[4], pc.worst_ray[5], pc.worst_ray[6]);
            for (int b = 0; b < MAX_PATH_DEPTH && (ctx->flags & CRT_HIP_FLAG_COUNTERS); ++b) {
                std::fprintf(stderr, "[crt_hip] frame %u closest launch %d: %.1f us total, queue drained after %.1f us (tail %.0f%%)\n",
                             ctx->frame_id, b, (pc.t_end[b] - pc.t_start[b]) / 100.0, (pc.t_drained[b] - pc.t_start[b]) / 100.0,
                             100.0 * (double)(pc.t_end[b] - pc.t_drained[b]) / (double)(pc.t_end[b] - pc.t_start[b]));
            }
            for (int k = 0; k < 2 && (ctx->flags & CRT_HIP_FLAG_COUNTERS); ++k) {
                static const char *names[4] = {"refill", "inner", "leaf", "retire"};
                double total = 0.0;
                for (int ph = 0; ph < 4; ++ph) {
                    total += (double)pc.prof_cycles[k][ph];
                }
                for (int ph = 0; ph < 4; ++ph) {
                    const double it = (double)std::max<unsigned long long>(1ull, pc.prof_iters[k][ph]);
                    std::fprintf(stderr, "[crt_hip] frame %u %s waves, %-6s: %5.1f%% of wave cycles, %.0f cycles/iteration, %.1f lanes/iteration\n",
                                 ctx->frame_id, k == 0 ? "closest" : "shadow ", names[ph],
                                 100.0 * (double)pc.prof_cycles[k][ph] / std::max(1.0, total),
                                 (double)pc.prof_cycles[k][ph] / it, (double)pc.prof_lanes[k][ph] / it);
                }
            }
            for (int b = 0; b < MAX_PATH_DEPTH; ++b) {
                std::fprintf(stderr, "[crt_hip] frame %u bounce %d: closest %u shadow_a %u shadow_b %u\n", ctx->frame_id,
                             b, pc.n_queue[b], pc.n_shadow_a[b], pc.n_shadow_b[b]);
            }
        }
        st.render_time_ms = (float)std::chrono::duration<double, std::milli>(t1 - t0).count();
        st.rays_per_second = (float)(st.rays / (st.render_time_ms * 1.0e-3));
        if (timing) {
            const bool dbg = std::getenv("CRT_HIP_DEBUG") != nullptr;
            for (const Span &sp : spans) {
                float ms = 0.f;
                HIP_CHECK(hipEventElapsedTime(&ms, ctx->events[sp.a], ctx->events[sp.b]));
                (sp.kind == 0 ? st.closest_ms : (sp.kind == 1 ? st.shadow_ms : st.shade_ms)) += ms;
                if (dbg) {
                    std::fprintf(stderr, "[crt_hip] frame %u span kind %d: %.3f ms\n", ctx->frame_id, sp.kind, ms);
                }
            }
        }
        if (stats) {
            *stats = st;
        }
        ctx->tile_fb_last = (int)(ctx->frame_id & 1u);
        ++ctx->frame_id;
        return CRT_HIP_OK;
    });
}

const uint32_t *crt_hip_framebuffer(const crt_hip_ctx *ctx) { return ctx ? ctx->img.data() : nullptr; }

int crt_hip_device_framebuffer(crt_hip_ctx *ctx, void **device_ptr, size_t *pitch_bytes)
{
    return guarded(ctx, [&]() -> int {
        if (ctx->width == 0 || !device_ptr) {
            return fail(ctx, CRT_HIP_ESTATE, "device_framebuffer before initialize");
        }
        *device_ptr = ctx->d_img.ptr;
        if (pitch_bytes) {
            *pitch_bytes = (size_t)ctx->width * sizeof(uint32_t);
        }
        return CRT_HIP_OK;
    });
}

int crt_hip_read_accum(crt_hip_ctx *ctx, float *rgb)
{
    return guarded(ctx, [&]() -> int {
        if (!rgb || ctx->width == 0) {
            return fail(ctx, CRT_HIP_EINVAL, "read_accum: bad arguments");
        }
        const size_t slots = (size_t)ctx->n_local_tiles * TILE_PIXELS;
        std::vector<float4> a(slots);
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        HIP_CHECK(hipMemcpy(a.data(), ctx->d_accum.ptr, slots * sizeof(float4), hipMemcpyDeviceToHost));
        std::memset(rgb, 0, sizeof(float) * 3 * (size_t)ctx->width * ctx->height);
        for (uint32_t lt = 0; lt < ctx->n_local_tiles; ++lt) {
            const uint32_t tile = ctx->tile_ids[lt];
            const uint32_t tx = (tile % ctx->ntx) * TILE, ty = (tile / ctx->ntx) * TILE;
            for (uint32_t m = 0; m < (uint32_t)TILE_PIXELS; ++m) {
                uint32_t ix = 0, iy = 0;
                for (int b = 0; b < 6; ++b) {
                    ix |= ((m >> (2 * b)) & 1u) << b;
                    iy |= ((m >> (2 * b + 1)) & 1u) << b;
                }
                const uint32_t x = tx + ix, y = ty + iy;
                if (x < (uint32_t)ctx->width && y < (uint32_t)ctx->height) {
                    const float4 v = a[(size_t)lt * TILE_PIXELS + m];
                    float *dst = rgb + 3 * ((size_t)y * ctx->width + x);
                    dst[0] = v.x;
                    dst[1] = v.y;
                    dst[2] = v.z;
                }
            }
        }
        return CRT_HIP_OK;
    });
}

int crt_hip_read_ray_counts(crt_hip_ctx *ctx, uint32_t *counts)
{
    return guarded(ctx, [&]() -> int {
        if (!counts || ctx->width == 0) {
            return fail(ctx, CRT_HIP_EINVAL, "read_ray_counts: bad arguments");
        }
        const size_t slots = (size_t)ctx->n_local_tiles * TILE_PIXELS;
        std::vector<uint32_t> a(slots);
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        HIP_CHECK(hipMemcpy(a.data(), ctx->d_ray_counts.ptr, slots * sizeof(uint32_t), hipMemcpyDeviceToHost));
        std::memset(counts, 0, sizeof(uint32_t) * (size_t)ctx->width * ctx->height);
        for (uint32_t lt = 0; lt < ctx->n_local_tiles; ++lt) {
            const uint32_t tile = ctx->tile_ids[lt];
            const uint32_t tx = (tile % ctx->ntx) * TILE, ty = (tile / ctx->ntx) * TILE;
            for (uint32_t m = 0; m < (uint32_t)TILE_PIXELS; ++m) {
                uint32_t ix = 0, iy = 0;
                for (int b = 0; b < 6; ++b) {
                    ix |= ((m >> (2 * b)) & 1u) << b;
                    iy |= ((m >> (2 * b + 1)) & 1u) << b;
                }
                const uint32_t x = tx + ix, y = ty + iy;
                if (x < (uint32_t)ctx->width && y < (uint32_t)ctx->height) {
                    counts[(size_t)y * ctx->width + x] = a[(size_t)lt * TILE_PIXELS + m];
                }
            }
        }
        return CRT_HIP_OK;
    });
}

int crt_hip_tile_buffer(crt_hip_ctx *ctx, void **device_ptr, size_t *n_bytes)
{
    return guarded(ctx, [&]() -> int {
        if (ctx->width == 0) {
            return fail(ctx, CRT_HIP_ESTATE, "tile_buffer before initialize");
        }
        if (device_ptr) {
            *device_ptr = ctx->d_tile_fb[ctx->tile_fb_last].ptr;
        }
        if (n_bytes) {
            *n_bytes = (size_t)ctx->n_tiles_padded * TILE_PIXELS * sizeof(uint32_t);
        }
        return CRT_HIP_OK;
    });
}

int crt_hip_assemble_tiles(crt_hip_ctx *ctx, const void *gathered, int world, int readback)
{
    return guarded(ctx, [&]() -> int {
        if (!gathered || world != ctx->world || ctx->width == 0) {
            return fail(ctx, CRT_HIP_EINVAL, "assemble_tiles: bad arguments");
        }
        launch_assemble(ctx->cfg(), static_cast<const uint32_t *>(gathered), ctx->n_tiles_padded * TILE_PIXELS, world,
                        (uint32_t)ctx->width, (uint32_t)ctx->height, ctx->d_img.as<uint32_t>());
        HIP_CHECK(hipGetLastError());
        if (readback) {
            HIP_CHECK(hipMemcpyAsync(ctx->img.data(), ctx->d_img.ptr, ctx->img.size() * sizeof(uint32_t),
                                     hipMemcpyDeviceToHost, ctx->stream));
        }
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        return CRT_HIP_OK;
    });
}

int crt_hip_trace_rays(crt_hip_ctx *ctx, uint64_t n, const float *org, const float *dir, const float *tmin,
                       const float *tmax, int closest, float *out_t, float *out_u, float *out_v, int32_t *out_inst,
                       int32_t *out_geom, int32_t *out_prim, crt_render_stats *stats)
{
    return guarded(ctx, [&]() -> int {
        if (!ctx->has_scene) {
            return fail(ctx, CRT_HIP_ESTATE, "trace_rays before set_scene");
        }
        if (!org || !dir || !tmin || !tmax || !out_t || n == 0 || n > 0x7fffffffull ||
            (closest && (!out_u || !out_v || !out_inst || !out_geom || !out_prim))) {
            return fail(ctx, CRT_HIP_EINVAL, "trace_rays: bad arguments");
        }
        for (uint64_t i = 1; i < n; ++i) {
            if (tmin[i] != tmin[0]) {
                return fail(ctx, CRT_HIP_EINVAL, "trace_rays: tmin must be the same for every ray of a batch");
            }
        }
        DeviceBuffer d_org, d_dir, d_tmin, d_tmax, d_t, d_u, d_v, d_inst, d_geom, d_prim, d_ctr;
        d_org.alloc(n * 12);
        d_dir.alloc(n * 12);
        d_tmin.alloc(n * 4);
        d_tmax.alloc(n * 4);
        d_t.alloc(n * 4);
        d_u.alloc(n * 4);
        d_v.alloc(n * 4);
        d_inst.alloc(n * 4);
        d_geom.alloc(n * 4);
        d_prim.alloc(n * 4);
        d_ctr.alloc(24);
        hipStream_t s = ctx->stream;
        HIP_CHECK(hipMemcpyAsync(d_org.ptr, org, n * 12, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(d_dir.ptr, dir, n * 12, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(d_tmin.ptr, tmin, n * 4, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(d_tmax.ptr, tmax, n * 4, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemsetAsync(d_ctr.ptr, 0, 24, s));
        hipEvent_t e0 = get_event(ctx, 0), e1 = get_event(ctx, 1);
        HIP_CHECK(hipEventRecord(e0, s));
        launch_trace_diag(ctx->cfg(), ctx->sv, (uint32_t)n, d_org.as<float>(), d_dir.as<float>(), tmin[0],
                          d_tmax.as<float>(), closest != 0, d_t.as<float>(), d_u.as<float>(), d_v.as<float>(),
                          d_inst.as<int32_t>(), d_geom.as<int32_t>(), d_prim.as<int32_t>(),
                          d_ctr.as<unsigned long long>());
        HIP_CHECK(hipEventRecord(e1, s));
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(out_t, d_t.ptr, n * 4, hipMemcpyDeviceToHost, s));
        if (closest) {
            HIP_CHECK(hipMemcpyAsync(out_u, d_u.ptr, n * 4, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipMemcpyAsync(out_v, d_v.ptr, n * 4, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipMemcpyAsync(out_inst, d_inst.ptr, n * 4, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipMemcpyAsync(out_geom, d_geom.ptr, n * 4, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipMemcpyAsync(out_prim, d_prim.ptr, n * 4, hipMemcpyDeviceToHost, s));
        }
        unsigned long long ctr[2] = {0, 0};
        HIP_CHECK(hipMemcpyAsync(ctr, d_ctr.ptr, 16, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        if (stats) {
            std::memset(stats, 0, sizeof(*stats));
            float ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            stats->rays = n;
            stats->render_time_ms = ms;
            stats->rays_per_second = (float)(n / (ms * 1e-3));
            if (closest) {
                stats->closest_rays = n;
                stats->closest_ms = ms;
                stats->closest_nodes = ctr[0];
                stats->closest_tris = ctr[1];
            } else {
                stats->shadow_rays = n;
                stats->shadow_ms = ms;
                stats->shadow_nodes = ctr[0];
                stats->shadow_tris = ctr[1];
            }
        }
        return CRT_HIP_OK;
    });
}

int crt_hip_kat(crt_hip_ctx *ctx, int fn, uint64_t n, const float *in, int in_stride, float *out, int out_stride)
{
    return guarded(ctx, [&]() -> int {
        if (!in || !out || n == 0 || n > 0x7fffffffull || in_stride <= 0 || out_stride <= 0) {
            return fail(ctx, CRT_HIP_EINVAL, "kat: bad arguments");
        }
        DeviceBuffer d_in, d_out;
        d_in.alloc(n * in_stride * 4);
        d_out.alloc(n * out_stride * 4);
        hipStream_t s = ctx->stream;
        HIP_CHECK(hipMemcpyAsync(d_in.ptr, in, n * in_stride * 4, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemsetAsync(d_out.ptr, 0, n * out_stride * 4, s));
        if (launch_kat(ctx->cfg(), ctx->sv, fn, (uint32_t)n, d_in.as<float>(), in_stride, d_out.as<float>(),
                       out_stride) != 0) {
            return fail(ctx, CRT_HIP_EINVAL, "kat: unknown function id");
        }
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(out, d_out.ptr, n * out_stride * 4, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        return CRT_HIP_OK;
    });
}

int crt_hip_bvh_info(crt_hip_ctx *ctx, uint64_t *n_nodes, uint64_t *n_tris, uint64_t *n_instances,
                     int32_t *two_level, float *root_frame)
{
    return guarded(ctx, [&]() -> int {
        if (!ctx->has_scene) {
            return fail(ctx, CRT_HIP_ESTATE, "bvh_info before set_scene");
        }
        if (n_nodes) {
            *n_nodes = ctx->n_nodes;
        }
        if (n_tris) {
            *n_tris = ctx->n_tris;
        }
        if (n_instances) {
            *n_instances = ctx->sv.n_instances;
        }
        if (two_level) {
            *two_level = (int32_t)ctx->sv.two_level;
        }
        if (root_frame) {
            std::memcpy(root_frame, &ctx->sv.root_frame, sizeof(QFrame));
        }
        return CRT_HIP_OK;
    });
}

int crt_hip_bvh_copy(crt_hip_ctx *ctx, void *nodes, void *tris)
{
    return guarded(ctx, [&]() -> int {
        if (!ctx->has_scene) {
            return fail(ctx, CRT_HIP_ESTATE, "bvh_copy before set_scene");
        }
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (nodes) {
            HIP_CHECK(hipMemcpy(nodes, ctx->d_nodes.ptr, ctx->n_nodes * sizeof(QNode), hipMemcpyDeviceToHost));
        }
        if (tris) {
            HIP_CHECK(hipMemcpy(tris, ctx->d_tris.ptr, ctx->n_tris * sizeof(TriRec), hipMemcpyDeviceToHost));
        }
        return CRT_HIP_OK;
    });
}

int crt_hip_bvh_layout(crt_hip_ctx *ctx, int32_t *root, uint32_t *n_top_nodes, uint32_t *stack_need,
                       uint32_t *lds_stack, int32_t *child_order)
{
    return guarded(ctx, [&]() -> int {
        if (!ctx->has_scene) {
            return fail(ctx, CRT_HIP_ESTATE, "bvh_layout before set_scene");
        }
        if (root) {
            *root = ctx->sv.root;
        }
        if (n_top_nodes) {
            *n_top_nodes = ctx->sv.n_top_nodes;
        }
        if (stack_need) {
            *stack_need = ctx->stack_need;
        }
        if (lds_stack) {
            *lds_stack = traversal_lds_stack(ctx->sv.two_level);
        }
        if (child_order) {
            *child_order = traversal_child_order();
        }
        return CRT_HIP_OK;
    });
}

int crt_hip_bvh_copy_instances(crt_hip_ctx *ctx, void *instances)
{
    return guarded(ctx, [&]() -> int {
        if (!ctx->has_scene || !instances) {
            return fail(ctx, CRT_HIP_ESTATE, "bvh_copy_instances: no scene / null buffer");
        }
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        HIP_CHECK(hipMemcpy(instances, ctx->d_instances.ptr, (size_t)ctx->sv.n_instances * sizeof(InstanceRec),
                            hipMemcpyDeviceToHost));
        return CRT_HIP_OK;
    });
}

} // extern "C"
