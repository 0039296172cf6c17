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
#include <mutex>
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
#include "scene_prepare.h"
#include "wavefront.h"

using namespace crt;

namespace {


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


} // namespace

// A deep copy of what the TREE of a scene is built from (crt_scene_desc borrows the caller's arrays for the duration of
// crt_hip_set_scene only): geometry, meshes, parameterised meshes with their material ids, instances, materials (the textured flag
// of a material id is read from them), lights. Texture images are not copied -- the refinement does not read them
// (scene_prepare.h prepare_scene tree_only).
struct SceneCopy {
    std::vector<std::vector<float>> verts, uvs;
    std::vector<std::vector<uint32_t>> indices, mat_ids;
    std::vector<crt_geometry_desc> geoms;
    std::vector<crt_mesh_desc> meshes;
    std::vector<crt_parameterized_mesh_desc> pmeshes;
    std::vector<crt_instance_desc> instances;
    std::vector<float> materials, lights;
    crt_scene_desc desc{};
    explicit SceneCopy(const crt_scene_desc &s)
    {
        geoms.resize(s.n_geometries);
        verts.resize(s.n_geometries);
        uvs.resize(s.n_geometries);
        indices.resize(s.n_geometries);
        for (uint32_t g = 0; g < s.n_geometries; ++g) {
            const crt_geometry_desc &in = s.geometries[g];
            verts[g].assign(in.vertices, in.vertices + 3 * in.n_vertices);
            indices[g].assign(in.indices, in.indices + 3 * in.n_triangles);
            if (in.uvs) {
                uvs[g].assign(in.uvs, in.uvs + 2 * in.n_vertices);
            }
            geoms[g] = crt_geometry_desc{verts[g].data(), in.n_vertices, indices[g].data(), in.n_triangles, in.uvs ? uvs[g].data() : nullptr};
        }
        meshes.assign(s.meshes, s.meshes + s.n_meshes);
        pmeshes.assign(s.parameterized_meshes, s.parameterized_meshes + s.n_parameterized_meshes);
        mat_ids.resize(s.n_parameterized_meshes);
        for (uint32_t k = 0; k < s.n_parameterized_meshes; ++k) {
            mat_ids[k].assign(pmeshes[k].material_ids, pmeshes[k].material_ids + pmeshes[k].n_material_ids);
            pmeshes[k].material_ids = mat_ids[k].data();
        }
        instances.assign(s.instances, s.instances + s.n_instances);
        materials.assign(s.materials, s.materials + (size_t)16 * s.n_materials);
        lights.assign(s.lights, s.lights + (size_t)20 * s.n_lights);
        desc = s;
        desc.geometries = geoms.data();
        desc.meshes = meshes.data();
        desc.parameterized_meshes = pmeshes.data();
        desc.instances = instances.data();
        desc.materials = materials.data();
        desc.lights = lights.data();
        desc.textures = nullptr; // (never read by a tree-only preparation)
    }
};

// The better tree of CRT_HIP_FLAG_REFINE_IN_BACKGROUND, built and uploaded by the context's refinement thread, waiting to be
// swapped in between two frames (swap_in_refined_tree). Only what depends on the tree: everything else of the scene stays.
struct RefinedTree {
    DeviceBuffer nodes, slots, tri_uvs, instances;
    uint64_t n_nodes = 0, n_slots = 0;
    uint32_t stack_need = 0, two_level = 0, n_top = 0;
    int32_t root = 0, world_inst = -1;
    QFrame root_frame{};
    double build_ms = 0.0;
};

struct crt_hip_ctx {
    int device = 0;
    uint32_t flags = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // Occlusion rays of bounce b and closest-hit rays of bounce b+1 are independent: with overlap on (the default),
    // the occlusion launch goes to the lane's aux stream so that its waves fill the CUs the other launch's tail
    // leaves idle (and vice versa). It needs its own traversal-stack spill slab.
    bool overlap = false;
    // PASS LANES. A frame is a chain of 17 dependent launches per pass, and every traversal launch ends with a few hundred
    // microseconds in which its longest rays finish on an otherwise idle chip -- a fixed cost per launch that does not
    // shrink with the ray count (a GPU's share of a frame at N = 8 is mostly that). Passes are independent (disjoint
    // pixel slots), so with the overlapped schedule a frame is cut into at least n_lanes passes and pass p runs on lane
    // p % n_lanes: its own queues, counters, streams and spill slabs. The launches of one lane fill the tails of the other's.
    // Results do not depend on how a frame is cut into passes (tests/test_gpu_edge_cases.py, test_gpu_scale.py).
    // Measured (profiles/r03_pass_lanes_ab.txt): C2 (3.7 M paths per frame) 8.60 -> 7.55 ms with two lanes; C4 (33 M) 68.0 ->
    // 69.6 ms and C3 (16.6 M) 9.9 -> 10.2 ms -- two passes' kernels side by side also share the caches; and one eighth of
    // C4 (4.1 M paths of expensive rays) 10.1 -> 11.2 ms. Path count alone does not tell, so frames of at most
    // LANES_MAX_PATHS paths are TRIED both ways after a (re)configuration: frame 0 warms up, frames 1-2 run with one lane,
    // frame 3 warms the re-carved queues up with two, frames 4-5 run with two; the smaller of each setting's two frame
    // times decides (one noisy frame cannot lock the slower setting in), the choice is reported in crt_render_stats::
    // pass_lanes and images are bit-identical either way. CRT_HIP_LANES=n cuts every frame (ranks of a multi-GPU job that
    // must agree set it).
    static constexpr uint64_t LANES_MAX_PATHS = 8ull << 20;
    static constexpr int LANE_TUNE_FRAMES = 6;
    int lane_choice = 1;   // lanes the next setup_queues cuts a tunable frame for
    int lane_tune = 0;     // frames rendered since the last (re)configuration, up to LANE_TUNE_FRAMES (tuned)
    float lane_t1 = 0.f, lane_t2 = 0.f; // best frame time with one lane / with two
    int lanes_in_use = 1;  // what setup_queues carved queues for
    struct PassLane {
        DeviceBuffer queue_mem, pc;
        PathQueue q[2]{};
        HitBuf hits{};
        ShadowQueueA sa{};
        ShadowQueueB sb{};
        float4 *radiance = nullptr;
        hipStream_t main = nullptr; // lane 0: the context's stream (not owned)
        hipStream_t aux = nullptr;
        hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_done = nullptr;
        bool owns_main = false;
    };
    static constexpr int MAX_LANES = 4;
    PassLane lanes[MAX_LANES];
    int n_lanes = 1;         // lanes with streams
    // With several lanes in use each lane's launches go strictly in order on ONE stream per lane: the other lane is what
    // fills the tails, and four streams (two lanes x main + aux) are as many as a process has hardware queues by default --
    // in a process that holds further streams (torch, a second context) they alias and the frame got 28 % SLOWER
    // (profiles/r03_pass_lanes_ab.txt: C2 7.45 ms with two streams either way; 7.49 / 9.58 ms with four). CRT_HIP_LANE_AUX=1
    // restores the per-lane occlusion stream.
    bool lane_aux = false;
    bool lanes_forced = false; // CRT_HIP_LANES given: cut every frame that is large enough; else small frames by trial
    hipEvent_t ev_begin = nullptr;
    int n_cus = 256;
    std::string name, err;
    int rank = 0, world = 1;

    // framebuffer state (RenderEmbree::initialize)
    int width = 0, height = 0, ntx = 0, nty = 0;
    std::vector<uint32_t> tile_ids; // tiles this rank renders
    uint32_t n_local_tiles = 0, n_tiles_padded = 0;
    DeviceBuffer d_tile_ids, d_accum, d_tile_fb[2], d_img, d_ray_counts;
    // The compact RGBA8 tile buffer is double-buffered: the gather of frame f (RCCL, on another stream) may still
    // read its buffer while frame f+1 is traced and accumulated into the other. The buffers alternate with every
    // rendered frame, independently of frame_id (which a moving camera resets every frame).
    int tile_fb_last = 1; // buffer the last rendered frame wrote (the first frame writes buffer 0)
    std::vector<uint32_t> img;
    uint32_t frame_id = 0;

    // scene (RenderEmbree::set_scene)
    bool has_scene = false;
    uint32_t spp = 1;
    SceneView sv{};
    DeviceBuffer d_spill, d_tri_uvs;
    DeviceBuffer d_nodes, d_slots, d_instances, d_material_ids, d_materials, d_textures, d_texels, d_lights;
    uint64_t n_nodes = 0, n_tris = 0;
    uint32_t stack_need = 0; // traversal-stack entries the deepest path of this scene's BVH can need
    // CRT_HIP_FLAG_REFINE_IN_BACKGROUND: set_scene uploads a quickly built tree and this thread builds the full-quality one
    // (host SAH + re-insertion), uploads it on a stream of its own and parks it in `refined`; the next render_begin swaps it in.
    // refine_state: 0 none, 1 building, 2 ready to be swapped in, 3 swapped in, -1 failed (refine_error says why; the quick tree stays).
    std::thread refine_thread;
    std::atomic<int> refine_state{0};
    std::unique_ptr<RefinedTree> refined;
    std::string refine_error;
    double refine_quick_ms = 0.0, refine_full_ms = 0.0;

    // wavefront state
    uint64_t capacity = 0; // paths per pass (every lane's queues hold that many)
    std::vector<hipEvent_t> events; // timing events of the diagnostic entry points
    // A frame is ENQUEUED (crt_hip_render_begin: every launch of it, no host wait) and later COLLECTED (crt_hip_render_end:
    // wait for its last event, read its counters and timings); crt_hip_render does both. Two slots, so that a caller may
    // enqueue frame f+1 before it collects frame f and the GPU never waits for the host between frames.
    struct Span {
        size_t a, b;
        int kind;   // 0 closest-hit traversal, 1 any-hit traversal, 2 raygen / shade / accumulate
        int bounce; // path-loop iteration of the launch; -1 raygen, -2 accumulate
    };
    struct FrameSlot {
        bool pending = false;
        std::vector<Span> spans;
        uint32_t passes = 0, used_lanes = 1, frame_id = 0;
        uint64_t total_paths = 0;
        std::chrono::high_resolution_clock::time_point t0;
        hipEvent_t begin = nullptr, done = nullptr;
        PassCounters *h_pc = nullptr; // pinned host copy of every pass's counters; grown by the frame that needs more
        uint32_t h_pc_cap = 0;
        std::vector<hipEvent_t> events; // event pairs around the frame's launches (CRT_HIP_FLAG_TIMING)
    };
    FrameSlot slots[2];
    int next_slot = 0;    // where the next frame is enqueued
    int oldest_slot = 0;  // the pending frame that is collected next

    ~crt_hip_ctx()
    {
        if (refine_thread.joinable()) {
            refine_thread.join();
        }
        for (hipEvent_t e : events) {
            (void)hipEventDestroy(e);
        }
        for (FrameSlot &f : slots) {
            if (f.begin) {
                (void)hipEventDestroy(f.begin);
                (void)hipEventDestroy(f.done);
            }
            for (hipEvent_t e : f.events) {
                (void)hipEventDestroy(e);
            }
            if (f.h_pc) {
                (void)hipHostFree(f.h_pc);
            }
        }
        for (PassLane &l : lanes) {
            if (l.aux) {
                (void)hipStreamDestroy(l.aux);
                (void)hipEventDestroy(l.ev_fork);
                (void)hipEventDestroy(l.ev_join);
            }
            if (l.ev_done) {
                (void)hipEventDestroy(l.ev_done);
            }
            if (l.owns_main && l.main) {
                (void)hipStreamDestroy(l.main);
            }
        }
        if (ev_begin) {
            (void)hipEventDestroy(ev_begin);
        }
        if (own_stream) {
            (void)hipStreamDestroy(own_stream);
        }
    }
    LaunchCfg cfg() const { return LaunchCfg{stream, n_cus, (flags & CRT_HIP_FLAG_COUNTERS) != 0, (flags & CRT_HIP_FLAG_ELIDE_UNUSED_SHADOW_RAYS) != 0}; }
};

namespace {

int fail(crt_hip_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) {
        ctx->err = msg;
    } else {
        set_global_error(msg);
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

// carve a lane's SoA queues out of one allocation
void carve_queues(crt_hip_ctx::PassLane &l, uint64_t cap)
{
    const size_t n_fields = 2 * 11 + 8 + 12 + 18 + 4; // PathQueue x 2, HitBuf records, ShadowQueueA, ShadowQueueB, radiance
    l.queue_mem.alloc(n_fields * cap * sizeof(float));
    uint32_t *base = l.queue_mem.as<uint32_t>();
    size_t k = 0;
    auto f32 = [&]() { return reinterpret_cast<float *>(base + (k++) * cap); };
    auto u32 = [&]() { return base + (k++) * cap; };
    auto i32 = [&]() { return reinterpret_cast<int32_t *>(base + (k++) * cap); };
    for (int qi = 0; qi < 2; ++qi) {
        for (int a = 0; a < 3; ++a) {
            l.q[qi].o[a] = f32();
        }
        for (int a = 0; a < 3; ++a) {
            l.q[qi].d[a] = f32();
        }
        l.q[qi].path = u32();
        l.q[qi].rng = u32();
        for (int a = 0; a < 3; ++a) {
            l.q[qi].tp[a] = f32();
        }
    }
    l.hits.rec = reinterpret_cast<float4 *>(base + k * cap); // 8 dwords per ray (cap is a multiple of 64: 16-byte aligned)
    k += 8;
    l.hits.inst_debug = nullptr;
    for (int a = 0; a < 3; ++a) {
        l.sa.o[a] = f32();
    }
    for (int a = 0; a < 3; ++a) {
        l.sa.d[a] = f32();
    }
    l.sa.tmax = f32();
    l.sa.cp = reinterpret_cast<float4 *>(base + k * cap); // 4 dwords per item (cap is a multiple of 64: 16-byte aligned)
    k += 4;
    l.sa.bslot = i32();
    for (int a = 0; a < 3; ++a) {
        l.sb.o[a] = f32();
    }
    for (int a = 0; a < 3; ++a) {
        l.sb.d[a] = f32();
    }
    l.sb.tmax = f32();
    for (int a = 0; a < 3; ++a) {
        l.sb.ca[a] = f32();
    }
    for (int a = 0; a < 3; ++a) {
        l.sb.cb[a] = f32();
    }
    for (int a = 0; a < 3; ++a) {
        l.sb.tp[a] = f32();
    }
    l.sb.path = u32();
    l.sb.reserved = i32();
    l.radiance = reinterpret_cast<float4 *>(base + k * cap);
    l.pc.alloc(sizeof(PassCounters));
}

// May frames of this many paths be cut for the lanes by trial (crt_hip_ctx::PassLane)?
bool lanes_tunable(const crt_hip_ctx *c, uint64_t total_paths)
{
    return c->n_lanes >= 2 && !c->lanes_forced && total_paths >= (2ull << 18) && total_paths <= crt_hip_ctx::LANES_MAX_PATHS;
}

int lanes_for(const crt_hip_ctx *c, uint64_t total_paths)
{
    if (c->lanes_forced) {
        return c->n_lanes > 1 && total_paths >= ((uint64_t)c->n_lanes << 18) ? c->n_lanes : 1;
    }
    return lanes_tunable(c, total_paths) ? c->lane_choice : 1;
}

// Paths per pass, and the queues of every lane. One pass if the frame fits and there is one lane; with several lanes
// the frame is cut into at least that many passes (when it is large enough for the cut to be worth a launch sequence).
void setup_queues(crt_hip_ctx *c)
{
    const uint64_t total_slots = (uint64_t)c->n_local_tiles * TILE_PIXELS;
    const uint64_t total_paths = total_slots * c->spp;
    uint64_t cap = std::min<uint64_t>(default_capacity(), total_paths);
    const int lanes = lanes_for(c, total_paths);
    if (lanes > 1) {
        cap = std::min<uint64_t>(cap, (total_paths + (uint64_t)lanes - 1) / (uint64_t)lanes + 64ull * c->spp);
    }
    cap = std::min<uint64_t>(cap, 1ull << PATH_ID_BITS); // a path's index shares its queue word with its ray count (crt_types.h)
    const uint64_t slots_per_pass = std::max<uint64_t>(64, (cap / c->spp) / 64 * 64);
    cap = slots_per_pass * c->spp;
    if (cap > (1ull << PATH_ID_BITS)) {
        throw std::runtime_error("samples_per_pixel too large: 64 pixels x spp paths must fit one pass of 2^27 paths");
    }
    c->capacity = cap;
    const uint32_t n_pass = (uint32_t)((total_paths + cap - 1) / cap);
    const int used_lanes = (int)std::min<uint32_t>((uint32_t)lanes, std::max<uint32_t>(1u, n_pass));
    c->lanes_in_use = used_lanes;
    for (int i = 0; i < crt_hip_ctx::MAX_LANES; ++i) {
        if (i < used_lanes) {
            carve_queues(c->lanes[i], cap);
        } else {
            c->lanes[i].queue_mem.release();
            c->lanes[i].pc.release();
        }
    }
    (void)n_pass;
}

hipEvent_t get_event(std::vector<hipEvent_t> &pool, size_t i)
{
    while (pool.size() <= i) {
        hipEvent_t e;
        HIP_CHECK(hipEventCreate(&e));
        pool.push_back(e);
    }
    return pool[i];
}
hipEvent_t get_event(crt_hip_ctx *c, size_t i) { return get_event(c->events, i); }
bool frames_in_flight(const crt_hip_ctx *c) { return c->slots[0].pending || c->slots[1].pending; }


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
        set_global_error("no HIP device available (this backend has no CPU fallback)");
        return nullptr;
    }
    if (device_id < 0 || device_id >= n) {
        set_global_error("device id out of range");
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
#if defined(CRT_SPEED_MODE)
        c->name += " [speed mode: fast-math build]";
#endif
        HIP_CHECK(hipStreamCreate(&c->own_stream));
        c->overlap = true; // CRT_HIP_OVERLAP=0: strictly serial launches
        if (const char *e = std::getenv("CRT_HIP_OVERLAP")) {
            c->overlap = std::atoi(e) != 0;
        }
        c->n_lanes = c->overlap ? 2 : 1; // CRT_HIP_LANES: passes in flight (1 .. 4)
        if (const char *e = std::getenv("CRT_HIP_LANES")) {
            c->n_lanes = std::min(std::max(std::atoi(e), 1), (int)crt_hip_ctx::MAX_LANES);
            c->lanes_forced = true;
        }
        if (const char *e = std::getenv("CRT_HIP_LANE_AUX")) {
            c->lane_aux = std::atoi(e) != 0;
        }
        HIP_CHECK(hipEventCreateWithFlags(&c->ev_begin, hipEventDisableTiming));
        for (int i = 0; i < c->n_lanes; ++i) {
            crt_hip_ctx::PassLane &l = c->lanes[i];
            if (i > 0) {
                HIP_CHECK(hipStreamCreateWithFlags(&l.main, hipStreamNonBlocking));
                l.owns_main = true;
                HIP_CHECK(hipEventCreateWithFlags(&l.ev_done, hipEventDisableTiming));
            }
            if (c->overlap && (i == 0 || c->lane_aux)) { // the occlusion stream: of the single lane, or of every lane on request
                HIP_CHECK(hipStreamCreateWithFlags(&l.aux, hipStreamNonBlocking));
                HIP_CHECK(hipEventCreateWithFlags(&l.ev_fork, hipEventDisableTiming));
                HIP_CHECK(hipEventCreateWithFlags(&l.ev_join, hipEventDisableTiming));
            }
        }
        c->stream = c->own_stream;
    } catch (const HipError &err) {
        set_global_error(err.msg);
        return nullptr;
    }
    return c.release();
}

void crt_hip_destroy(crt_hip_ctx *ctx)
{
    if (ctx) {
        if (ctx->refine_thread.joinable()) {
            ctx->refine_thread.join();
        }
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        delete ctx;
    }
}

const char *crt_hip_last_error(const crt_hip_ctx *ctx) { return ctx ? ctx->err.c_str() : global_error().c_str(); }
const char *crt_hip_name(const crt_hip_ctx *ctx) { return ctx ? ctx->name.c_str() : ""; }
uint32_t crt_hip_frame_id(const crt_hip_ctx *ctx) { return ctx ? ctx->frame_id : 0; }

int crt_hip_set_stream(crt_hip_ctx *ctx, void *hip_stream)
{
    return guarded(ctx, [&]() -> int {
        if (frames_in_flight(ctx)) {
            return fail(ctx, CRT_HIP_ESTATE, "a frame enqueued with crt_hip_render_begin is still in flight: collect it with crt_hip_render_end first");
        }
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
        return CRT_HIP_OK;
    });
}

int crt_hip_set_partition(crt_hip_ctx *ctx, int rank, int world)
{
    return guarded(ctx, [&]() -> int {
        if (frames_in_flight(ctx)) {
            return fail(ctx, CRT_HIP_ESTATE, "a frame enqueued with crt_hip_render_begin is still in flight: collect it with crt_hip_render_end first");
        }
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
        if (frames_in_flight(ctx)) {
            return fail(ctx, CRT_HIP_ESTATE, "a frame enqueued with crt_hip_render_begin is still in flight: collect it with crt_hip_render_end first");
        }
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
        ctx->lane_tune = 0;
        ctx->lane_choice = 1;
        return CRT_HIP_OK;
    });
}

} // extern "C"

namespace {

// device half of set_scene (the host half: scene_prepare.cpp): the prepared arrays -> HBM, SceneView, traversal-stack slab
void upload_scene(crt_hip_ctx *ctx, const crt_hip_prepared_scene &ps)
{
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->frame_id = 0;
    ctx->has_scene = false;
    ctx->spp = ps.spp;
    ctx->capacity = 0;
    ctx->lane_tune = 0;
    ctx->lane_choice = 1;
    upload(ctx->d_nodes, ps.nodes, ctx->stream);
    upload(ctx->d_slots, ps.slots, ctx->stream);
    upload(ctx->d_tri_uvs, ps.tri_uvs, ctx->stream);
    upload(ctx->d_instances, ps.insts, ctx->stream);
    upload(ctx->d_material_ids, ps.material_ids, ctx->stream);
    upload(ctx->d_materials, ps.materials, ctx->stream);
    upload(ctx->d_textures, ps.tex, ctx->stream);
    upload(ctx->d_texels, ps.texels, ctx->stream);
    upload(ctx->d_lights, ps.lights, ctx->stream);
    ctx->n_nodes = ps.nodes.size();
    ctx->n_tris = ps.slots.size(); // leaf slots (one or two triangles each)
    ctx->stack_need = ps.stack_need;

    SceneView &sv = ctx->sv;
    sv.nodes = ctx->d_nodes.as<PNode>();
    sv.root_frame = ps.root_frame;
    sv.slots = ctx->d_slots.as<LeafSlot>();
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
    ctx->d_spill.alloc((size_t)(ctx->overlap ? 2 : 1) * (size_t)ctx->n_lanes * spill_words * sizeof(int32_t)); // a slab per stream
    sv.stack_spill = ctx->d_spill.as<int32_t>();
    sv.root = ps.root;
    sv.two_level = ps.two_level;
    sv.world_inst = ps.world_inst;
    // (the kernels stage at most this many of the BFS-ordered top levels in LDS and test `cur < root + n_top_nodes`
    // for "is it there": a prepared scene that claims more is clamped, not trusted)
    sv.n_top_nodes = std::min<uint32_t>(ps.n_top, ps.two_level == 1u ? (uint32_t)CRT_MAX_TOP_NODES_TWO_LEVEL : (uint32_t)CRT_MAX_TOP_NODES);
    ctx->has_scene = true;
}

// ---- CRT_HIP_FLAG_REFINE_IN_BACKGROUND ---------------------------------------------------------------------------------------
// A frame's image does not depend on the tree (closest hit = lexicographic minimum, occlusion = boolean: any correct tree gives
// the same bits), so a better tree may replace the one in use BETWEEN two frames of an accumulation without anybody noticing
// anything but the frame time. set_scene returns as soon as a quickly built tree is resident -- the host SAH tree without
// re-insertion passes, or the device builder's linear tree with CRT_HIP_BUILD=device -- and this thread builds the tree the
// default path would have made the caller wait for (embree_utils.cpp:63-76,121-129: the reference's rtcCommitScene blocks).
void wait_for_refinement(crt_hip_ctx *ctx)
{
    if (ctx->refine_thread.joinable()) {
        ctx->refine_thread.join();
    }
}

void refinement_thread(crt_hip_ctx *ctx, std::shared_ptr<SceneCopy> scene, int n_threads)
{
    try {
        const auto t0 = std::chrono::high_resolution_clock::now();
        crt_hip_prepared_scene ps;
        prepare_scene(&scene->desc, &ps, n_threads, -1, -1, true); // host SAH + the environment's / default re-insertion passes; tree only
        scene.reset();
        std::unique_ptr<RefinedTree> r(new RefinedTree);
        HIP_CHECK(hipSetDevice(ctx->device));
        hipStream_t up = nullptr;
        HIP_CHECK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
        try {
            upload(r->nodes, ps.nodes, up);
            upload(r->slots, ps.slots, up);
            upload(r->tri_uvs, ps.tri_uvs, up);
            upload(r->instances, ps.insts, up);
        } catch (...) {
            (void)hipStreamDestroy(up);
            throw;
        }
        (void)hipStreamDestroy(up);
        r->n_nodes = ps.nodes.size();
        r->n_slots = ps.slots.size();
        r->stack_need = ps.stack_need;
        r->two_level = ps.two_level;
        r->n_top = ps.n_top;
        r->root = ps.root;
        r->world_inst = ps.world_inst;
        r->root_frame = ps.root_frame;
        r->build_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
        ctx->refine_full_ms = r->build_ms;
        ctx->refined = std::move(r);
        ctx->refine_state.store(2, std::memory_order_release);
    } catch (const HipError &e) {
        ctx->refine_error = e.msg;
        ctx->refine_state.store(-1, std::memory_order_release);
    } catch (const std::exception &e) {
        ctx->refine_error = e.what();
        ctx->refine_state.store(-1, std::memory_order_release);
    }
}

// render_begin: if the refined tree is ready, let the frames in flight finish and swap it in (pointers only: it is resident already)
void swap_in_refined_tree(crt_hip_ctx *ctx)
{
    if (ctx->refine_state.load(std::memory_order_acquire) != 2) {
        return;
    }
    wait_for_refinement(ctx);
    for (crt_hip_ctx::FrameSlot &f : ctx->slots) {
        if (f.pending) {
            HIP_CHECK(hipEventSynchronize(f.done)); // (its statistics are still collected by render_end)
        }
    }
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    RefinedTree &r = *ctx->refined;
    SceneView &sv = ctx->sv;
    if (r.two_level != sv.two_level || r.n_slots != ctx->n_tris) {
        ctx->refine_error = "the refined tree is of another kind than the quick one";
        ctx->refined.reset();
        ctx->refine_state.store(-1, std::memory_order_release);
        return;
    }
    std::swap(ctx->d_nodes.ptr, r.nodes.ptr);
    std::swap(ctx->d_nodes.bytes, r.nodes.bytes);
    std::swap(ctx->d_slots.ptr, r.slots.ptr);
    std::swap(ctx->d_slots.bytes, r.slots.bytes);
    std::swap(ctx->d_tri_uvs.ptr, r.tri_uvs.ptr);
    std::swap(ctx->d_tri_uvs.bytes, r.tri_uvs.bytes);
    std::swap(ctx->d_instances.ptr, r.instances.ptr);
    std::swap(ctx->d_instances.bytes, r.instances.bytes);
    ctx->n_nodes = r.n_nodes;
    sv.nodes = ctx->d_nodes.as<PNode>();
    sv.slots = ctx->d_slots.as<LeafSlot>();
    sv.tri_uvs = ctx->d_tri_uvs.as<float>();
    sv.instances = ctx->d_instances.as<InstanceRec>();
    sv.root = r.root;
    sv.root_frame = r.root_frame;
    sv.world_inst = r.world_inst;
    sv.n_top_nodes = std::min<uint32_t>(r.n_top, r.two_level == 1u ? (uint32_t)CRT_MAX_TOP_NODES_TWO_LEVEL : (uint32_t)CRT_MAX_TOP_NODES);
    if (r.stack_need > ctx->stack_need) { // a deeper tree: the HBM part of the traversal stack grows with it
        const uint32_t lds_stack = traversal_lds_stack(r.two_level);
        sv.spill_depth = std::max<uint32_t>(8u, r.stack_need > lds_stack ? r.stack_need - lds_stack : 0u);
        const size_t spill_words = (size_t)sv.spill_stride * sv.spill_depth;
        ctx->d_spill.alloc((size_t)(ctx->overlap ? 2 : 1) * (size_t)ctx->n_lanes * spill_words * sizeof(int32_t));
        sv.stack_spill = ctx->d_spill.as<int32_t>();
        ctx->capacity = 0; // (the pass lanes' spill slabs are carved from it: setup_queues again)
    }
    ctx->stack_need = std::max(ctx->stack_need, r.stack_need);
    ctx->refined.reset(); // frees the quick tree's arrays
    ctx->refine_state.store(3, std::memory_order_release);
    if (std::getenv("CRT_HIP_DEBUG")) {
        std::fprintf(stderr, "[crt_hip] refined tree swapped in: %llu nodes, built + uploaded in the background in %.1f ms (set_scene returned after %.1f ms)\n",
                     (unsigned long long)ctx->n_nodes, ctx->refine_full_ms, ctx->refine_quick_ms);
    }
}

} // namespace


namespace {

// crt_hip_trace_rays(CRT_HIP_TRACE_PRODUCTION): explicit rays through the kernels a FRAME launches -- k_trace_closest /
// k_trace_shadow without counters, with ClosestSource / ShadowSource -- fed and read back through the frame's own queue
// records, so that the bit-exact traversal tests cover the production instantiations (register allocation, retire path)
// and not only the instrumented diagnostic kernel.
int trace_rays_production(crt_hip_ctx *ctx, uint64_t n, const float *org, const float *dir, const float *tmin, const float *tmax,
                          bool closest, float *out_t, float *out_u, float *out_v, int32_t *out_inst, int32_t *out_geom,
                          int32_t *out_prim, crt_render_stats *stats)
{
    if (!org || !dir || !tmin || !tmax || !out_t || n == 0 || n > (1ull << PATH_ID_BITS) ||
        (closest && (!out_u || !out_v || !out_inst || !out_geom || !out_prim))) {
        return fail(ctx, CRT_HIP_EINVAL, "trace_rays: bad arguments");
    }
    if (!(tmin[0] == 0.f || tmin[0] == RAY_EPS)) {
        return fail(ctx, CRT_HIP_EINVAL, "trace_rays (production kernels): tmin must be 0 or EPSILON, as inside a frame");
    }
    for (uint64_t i = 0; i < n; ++i) {
        if (tmin[i] != tmin[0] || (closest && tmax[i] != RAY_TFAR)) {
            return fail(ctx, CRT_HIP_EINVAL, "trace_rays (production kernels): one tmin per batch, and tmax = 1e20 for closest hits");
        }
    }
    if (!closest && tmin[0] != RAY_EPS) {
        return fail(ctx, CRT_HIP_EINVAL, "trace_rays (production kernels): occlusion rays start at EPSILON");
    }
    const int bounce = tmin[0] == 0.f ? 0 : 1; // k_trace_closest: tnear = bounce == 0 ? 0 : EPSILON
    hipStream_t s = ctx->stream;
    LaunchCfg cfg = ctx->cfg();
    cfg.counters = false;
    // SoA ray fields, filled on the host
    std::vector<float> soa(7 * n);
    for (uint64_t i = 0; i < n; ++i) {
        for (int a = 0; a < 3; ++a) {
            soa[(size_t)a * n + i] = org[3 * i + a];
            soa[(size_t)(3 + a) * n + i] = dir[3 * i + a];
        }
        soa[6 * n + i] = tmax[i];
    }
    DeviceBuffer d_rays, d_pc, d_out, d_aux;
    d_rays.alloc(7 * n * sizeof(float));
    d_pc.alloc(sizeof(PassCounters));
    HIP_CHECK(hipMemcpyAsync(d_rays.ptr, soa.data(), 7 * n * sizeof(float), hipMemcpyHostToDevice, s));
    PassCounters pc;
    std::memset(&pc, 0, sizeof(pc));
    hipEvent_t e0 = get_event(ctx, 0), e1 = get_event(ctx, 1);
    float *base = d_rays.as<float>();
    if (closest) {
        pc.n_queue[bounce].v = (uint32_t)n;
        HIP_CHECK(hipMemcpyAsync(d_pc.ptr, &pc, sizeof(pc), hipMemcpyHostToDevice, s));
        PathQueue q{};
        for (int a = 0; a < 3; ++a) {
            q.o[a] = base + (size_t)a * n;
            q.d[a] = base + (size_t)(3 + a) * n;
        }
        d_out.alloc(2 * n * sizeof(float4));
        d_aux.alloc(n * sizeof(int32_t));
        HIP_CHECK(hipMemsetAsync(d_out.ptr, 0xff, d_out.bytes, s));
        HitBuf hb{d_out.as<float4>(), d_aux.as<int32_t>()};
        HIP_CHECK(hipEventRecord(e0, s));
        launch_trace_closest(cfg, ctx->sv, q, hb, d_pc.as<PassCounters>(), bounce);
        HIP_CHECK(hipEventRecord(e1, s));
        HIP_CHECK(hipGetLastError());
        std::vector<float4> rec(2 * n);
        std::vector<LeafSlot> slots(ctx->n_tris);
        HIP_CHECK(hipMemcpyAsync(rec.data(), d_out.ptr, 2 * n * sizeof(float4), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(out_inst, d_aux.ptr, n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(slots.data(), ctx->d_slots.ptr, ctx->n_tris * sizeof(LeafSlot), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        for (uint64_t i = 0; i < n; ++i) {
            int32_t tri;
            std::memcpy(&tri, &rec[2 * i].w, 4);
            out_t[i] = rec[2 * i].x;
            out_u[i] = rec[2 * i].y;
            out_v[i] = rec[2 * i].z;
            const LeafSlot &sl = slots[tri < 0 ? 0 : (size_t)((uint32_t)tri >> 1)];
            out_geom[i] = tri < 0 ? -1 : (int32_t)(sl.geom_sel & SLOT_GEOM_MASK);
            out_prim[i] = tri < 0 ? -1 : (int32_t)((tri & 1) != 0 ? sl.prim1 : sl.prim0);
        }
    } else {
        pc.n_shadow_a[bounce].v = (uint32_t)n;
        HIP_CHECK(hipMemcpyAsync(d_pc.ptr, &pc, sizeof(pc), hipMemcpyHostToDevice, s));
        // one ShadowQueueA item per ray, contribution (1, 0, 0), no second ray: a visible ray leaves radiance.x = 1
        std::vector<float4> extra(n);
        for (uint64_t i = 0; i < n; ++i) {
            uint32_t path = (uint32_t)i;
            float pw;
            std::memcpy(&pw, &path, 4);
            extra[i] = make_float4(1.f, 0.f, 0.f, pw);
        }
        d_aux.alloc(n * sizeof(float4));
        HIP_CHECK(hipMemcpyAsync(d_aux.ptr, extra.data(), n * sizeof(float4), hipMemcpyHostToDevice, s));
        ShadowQueueA sa{};
        for (int a = 0; a < 3; ++a) {
            sa.o[a] = base + (size_t)a * n;
            sa.d[a] = base + (size_t)(3 + a) * n;
        }
        sa.tmax = base + 6 * n;
        sa.cp = d_aux.as<float4>();
        sa.bslot = nullptr; // (no item has a second ray)
        ShadowQueueB sb{};
        d_out.alloc(n * sizeof(float4));
        HIP_CHECK(hipMemsetAsync(d_out.ptr, 0, d_out.bytes, s));
        HIP_CHECK(hipEventRecord(e0, s));
        launch_trace_shadow(cfg, ctx->sv, sa, sb, d_out.as<float4>(), d_pc.as<PassCounters>(), bounce);
        HIP_CHECK(hipEventRecord(e1, s));
        HIP_CHECK(hipGetLastError());
        std::vector<float4> rad(n);
        HIP_CHECK(hipMemcpyAsync(rad.data(), d_out.ptr, n * sizeof(float4), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        for (uint64_t i = 0; i < n; ++i) {
            out_t[i] = rad[i].x; // 1 = the segment is unoccluded
        }
    }
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        stats->rays = n;
        stats->render_time_ms = ms;
        stats->rays_per_second = (float)(n / (ms * 1e-3));
        (closest ? stats->closest_rays : stats->shadow_rays) = n;
        (closest ? stats->closest_ms : stats->shadow_ms) = ms;
    }
    return CRT_HIP_OK;
}

} // namespace

extern "C" {

int crt_hip_set_prepared_scene(crt_hip_ctx *ctx, const crt_hip_prepared_scene *ps)
{
    return guarded(ctx, [&]() -> int {
        if (frames_in_flight(ctx)) {
            return fail(ctx, CRT_HIP_ESTATE, "a frame enqueued with crt_hip_render_begin is still in flight: collect it with crt_hip_render_end first");
        }
        if (!ps) {
            return fail(ctx, CRT_HIP_EINVAL, "prepared scene is null");
        }
        wait_for_refinement(ctx); // (of an earlier scene)
        ctx->refined.reset();
        ctx->refine_state.store(0);
        upload_scene(ctx, *ps);
        return CRT_HIP_OK;
    });
}

// RenderBackend::set_scene = prepare + upload
int crt_hip_set_scene(crt_hip_ctx *ctx, const crt_scene_desc *s)
{
    return guarded(ctx, [&]() -> int {
        if (frames_in_flight(ctx)) {
            return fail(ctx, CRT_HIP_ESTATE, "a frame enqueued with crt_hip_render_begin is still in flight: collect it with crt_hip_render_end first");
        }
        wait_for_refinement(ctx); // (of an earlier scene)
        ctx->refined.reset();
        ctx->refine_state.store(0);
        crt_hip_prepared_scene ps;
        const char *where = std::getenv("CRT_HIP_BUILD"); // "device": BLAS of large meshes built on this context's GPU
        // (a scene of a few thousand triangles builds in milliseconds either way: it gets the full-quality tree at once, not a
        // quick one that nothing would ever replace)
        uint64_t instanced_tris = 0;
        if (s && s->instances && s->parameterized_meshes && s->meshes && s->geometries) {
            for (uint32_t i = 0; i < s->n_instances; ++i) {
                const uint32_t pm = s->instances[i].parameterized_mesh_id;
                if (pm < s->n_parameterized_meshes && s->parameterized_meshes[pm].mesh_id < s->n_meshes) {
                    const crt_mesh_desc &m = s->meshes[s->parameterized_meshes[pm].mesh_id];
                    for (uint32_t g = m.first_geometry; g < m.first_geometry + m.n_geometries && g < s->n_geometries; ++g) {
                        instanced_tris += s->geometries[g].n_triangles;
                    }
                }
            }
        }
        const bool refine = (ctx->flags & CRT_HIP_FLAG_REFINE_IN_BACKGROUND) != 0 && instanced_tris >= 16384;
        const auto t_prep = std::chrono::high_resolution_clock::now();
        // (refine: the quick tree -- no re-insertion passes on the host; the full-quality tree follows in the background)
        prepare_scene(s, &ps, host_threads(), where && std::strcmp(where, "device") == 0 ? ctx->device : -1, refine ? 0 : -1);
        const auto t0 = std::chrono::high_resolution_clock::now();
        upload_scene(ctx, ps);
        if (refine && !ps.slots.empty()) {
            ctx->refine_quick_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t_prep).count();
            ctx->refine_state.store(1);
            ctx->refine_thread = std::thread(refinement_thread, ctx, std::make_shared<SceneCopy>(*s), host_threads());
        }
        if (std::getenv("CRT_HIP_DEBUG")) {
            std::fprintf(stderr, "[crt_hip] set_scene %-22s %8.1f ms\n", "upload",
                         std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count());
        }
        return CRT_HIP_OK;
    });
}

int crt_hip_refine_state(crt_hip_ctx *ctx, double *quick_ms, double *full_ms)
{
    if (!ctx) {
        return 0;
    }
    const int st = ctx->refine_state.load(std::memory_order_acquire);
    if (quick_ms) {
        *quick_ms = ctx->refine_quick_ms;
    }
    if (full_ms) {
        *full_ms = st >= 2 ? ctx->refine_full_ms : 0.0;
    }
    if (st < 0) {
        ctx->err = "background refinement failed: " + ctx->refine_error;
    }
    return st;
}

int32_t crt_hip_world_instance(crt_hip_ctx *ctx) { return ctx && ctx->has_scene ? ctx->sv.world_inst : -1; }
int crt_hip_child_order(void) { return traversal_child_order(); }
uint32_t crt_hip_lds_stack_entries(int two_level) { return traversal_lds_stack((uint32_t)two_level); }

// RenderEmbree::render (render_embree.cpp:135-216), first half: every launch of the frame goes to the stream(s); nothing waits
int crt_hip_render_begin(crt_hip_ctx *ctx, const float pos[3], const float dir_[3], const float up_[3], float fovy,
                         int camera_changed, int readback)
{
    return guarded(ctx, [&]() -> int {
        if (!ctx->has_scene || ctx->width == 0) {
            return fail(ctx, CRT_HIP_ESTATE, "render before initialize + set_scene");
        }
        crt_hip_ctx::FrameSlot &fs = ctx->slots[ctx->next_slot];
        if (fs.pending) {
            return fail(ctx, CRT_HIP_ESTATE, "render_begin: two frames are in flight already; collect one with crt_hip_render_end");
        }
        swap_in_refined_tree(ctx); // CRT_HIP_FLAG_REFINE_IN_BACKGROUND: the better tree, if it has arrived
        if (readback && frames_in_flight(ctx)) {
            // there is ONE host image (RenderBackend::img): a second frame's copy would overwrite the first one's before its
            // render_end hands it out. Pipelined callers read tiles / the device framebuffer; a host image wants one frame at a time
            return fail(ctx, CRT_HIP_ESTATE, "render_begin: readback = 1 while another frame is in flight (one host image): collect it first");
        }
        if (ctx->capacity == 0) {
            // the queues are (re)carved: nothing of an earlier frame may still be using them
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
            setup_queues(ctx);
        }
        if (!fs.begin) {
            HIP_CHECK(hipEventCreate(&fs.begin));
            HIP_CHECK(hipEventCreate(&fs.done));
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
        using Span = crt_hip_ctx::Span;
        std::vector<Span> &spans = fs.spans;
        spans.clear();
        PassCounters *h_pc = nullptr; // set below, once the number of passes is known
        // event pairs around each launch, recorded on the stream the launch goes to: with the overlapped
        // schedule the occlusion launches' spans live on the auxiliary stream and run concurrently with the
        // closest-hit spans of the next bounce (so the per-kind sums may add up to more than the frame time)
        auto mark = [&](int kind, hipStream_t on, int bounce) {
            if (timing) {
                hipEvent_t e0 = get_event(fs.events, ev), e1 = get_event(fs.events, ev + 1);
                (void)e1;
                HIP_CHECK(hipEventRecord(e0, on));
                spans.push_back(Span{ev, ev + 1, kind, bounce});
                ev += 2;
            }
        };
        auto mark_end = [&](hipStream_t on) {
            if (timing) {
                HIP_CHECK(hipEventRecord(get_event(fs.events, spans.back().b), on));
            }
        };

        bool overlap = ctx->overlap;
        // The compact tile buffer alternates with every RENDERED frame, whatever frame_id does (a moving camera resets
        // frame_id to 0 every frame): the asynchronous gather of the previous frame may still be reading the other one.
        const int tile_buf = ctx->tile_fb_last ^ 1;
        fs.t0 = std::chrono::high_resolution_clock::now();
        HIP_CHECK(hipEventRecord(fs.begin, ctx->stream));
        const uint32_t n_pass_frame = (uint32_t)((total_slots + slots_per_pass - 1) / slots_per_pass);
        const int used_lanes = (int)std::min<uint32_t>((uint32_t)ctx->lanes_in_use, std::max<uint32_t>(1u, n_pass_frame));
        if (fs.h_pc_cap < n_pass_frame) {
            if (fs.h_pc) {
                HIP_CHECK(hipHostFree(fs.h_pc));
                fs.h_pc = nullptr;
            }
            HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&fs.h_pc), sizeof(PassCounters) * n_pass_frame));
            fs.h_pc_cap = n_pass_frame;
        }
        h_pc = fs.h_pc;
        ctx->lanes[0].main = ctx->stream;
        if (used_lanes > 1 && !ctx->lane_aux) {
            overlap = false;
        }
        if (used_lanes > 1) { // the other lanes start where the caller's stream stands
            HIP_CHECK(hipEventRecord(ctx->ev_begin, ctx->stream));
            for (int i = 1; i < used_lanes; ++i) {
                HIP_CHECK(hipStreamWaitEvent(ctx->lanes[i].main, ctx->ev_begin, 0));
            }
        }
        const size_t slab = (size_t)ctx->sv.spill_stride * ctx->sv.spill_depth; // traversal-stack spill slab of one stream
        uint32_t pass = 0;
        for (uint64_t slot0 = 0; slot0 < total_slots; slot0 += slots_per_pass, ++pass) {
            const int li = (int)(pass % (uint32_t)used_lanes);
            crt_hip_ctx::PassLane &ln = ctx->lanes[li];
            LaunchCfg lcfg = cfg, aux_cfg = cfg;
            lcfg.stream = ln.main;
            aux_cfg.stream = ln.aux;
            SceneView sv = ctx->sv, aux_sv = ctx->sv;
            sv.stack_spill += slab * (size_t)((overlap ? 2 : 1) * li);
            aux_sv.stack_spill = sv.stack_spill + (overlap ? slab : 0);
            const uint32_t n_slots = (uint32_t)std::min<uint64_t>(slots_per_pass, total_slots - slot0);
            const uint32_t n_paths = n_slots * ctx->spp;
            PassCounters *d_pc = ln.pc.as<PassCounters>();
            HIP_CHECK(hipMemsetAsync(d_pc, 0, sizeof(PassCounters), ln.main));
            if (cfg.counters) { // atomicMin targets start at all-ones
                HIP_CHECK(hipMemsetAsync(d_pc->t_start, 0xff, 2 * MAX_PATH_DEPTH * sizeof(unsigned long long), ln.main));
            }
            mark(2, ln.main, -1);
            launch_raygen(lcfg, vp, d_tiles, (uint32_t)slot0, n_paths, ln.q[0], ln.radiance, d_pc);
            mark_end(ln.main);
            for (int b = 0; b < MAX_PATH_DEPTH; ++b) {
                if (!overlap || b == 0) {
                    mark(0, ln.main, b);
                    launch_trace_closest(lcfg, sv, ln.q[b & 1], ln.hits, d_pc, b);
                    mark_end(ln.main);
                }
                mark(2, ln.main, b);
                launch_shade(lcfg, sv, ln.q[b & 1], ln.hits, ln.q[(b + 1) & 1], ln.sa, ln.sb, ln.radiance, d_pc, b, n_paths);
                mark_end(ln.main);
                if (overlap) {
                    // shade(b) -> { shadow(b) on aux  ||  closest(b+1) on the lane's main stream } -> shade(b+1)
                    HIP_CHECK(hipEventRecord(ln.ev_fork, ln.main));
                    HIP_CHECK(hipStreamWaitEvent(ln.aux, ln.ev_fork, 0));
                    mark(1, ln.aux, b);
                    launch_trace_shadow(aux_cfg, aux_sv, ln.sa, ln.sb, ln.radiance, d_pc, b);
                    mark_end(ln.aux);
                    HIP_CHECK(hipEventRecord(ln.ev_join, ln.aux));
                    if (b + 1 < MAX_PATH_DEPTH) {
                        mark(0, ln.main, b + 1);
                        launch_trace_closest(lcfg, sv, ln.q[(b + 1) & 1], ln.hits, d_pc, b + 1);
                        mark_end(ln.main);
                    }
                    HIP_CHECK(hipStreamWaitEvent(ln.main, ln.ev_join, 0));
                } else {
                    mark(1, ln.main, b);
                    launch_trace_shadow(lcfg, sv, ln.sa, ln.sb, ln.radiance, d_pc, b);
                    mark_end(ln.main);
                }
            }
            mark(2, ln.main, -2);
            launch_accumulate(lcfg, vp, d_tiles, (uint32_t)slot0, n_slots, ln.radiance, ctx->d_accum.as<float4>(),
                              ctx->d_tile_fb[tile_buf].as<uint32_t>(), ctx->world == 1 ? ctx->d_img.as<uint32_t>() : nullptr,
                              ctx->d_ray_counts.as<uint32_t>());
            mark_end(ln.main);
            HIP_CHECK(hipMemcpyAsync(&h_pc[pass], d_pc, sizeof(PassCounters), hipMemcpyDeviceToHost, ln.main));
        }
        for (int i = 1; i < used_lanes; ++i) { // the caller's stream continues when every lane is done
            HIP_CHECK(hipEventRecord(ctx->lanes[i].ev_done, ctx->lanes[i].main));
            HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->lanes[i].ev_done, 0));
        }
        HIP_CHECK(hipGetLastError());
        if (readback && ctx->world == 1) {
            HIP_CHECK(hipMemcpyAsync(ctx->img.data(), ctx->d_img.ptr, ctx->img.size() * sizeof(uint32_t),
                                     hipMemcpyDeviceToHost, ctx->stream));
        }
        HIP_CHECK(hipEventRecord(fs.done, ctx->stream));
        fs.pending = true;
        fs.passes = pass;
        fs.used_lanes = (uint32_t)used_lanes;
        fs.frame_id = ctx->frame_id;
        fs.total_paths = total_slots * ctx->spp;
        ctx->tile_fb_last = tile_buf;
        ++ctx->frame_id;
        ctx->next_slot ^= 1;
        // while the library is still trying this frame size with one pass lane and with two, the queues may be carved
        // again after the frame: it is completed here (collected by crt_hip_render_end as usual)
        if (ctx->lane_tune < crt_hip_ctx::LANE_TUNE_FRAMES && lanes_tunable(ctx, fs.total_paths)) {
            HIP_CHECK(hipEventSynchronize(fs.done));
        }
        return CRT_HIP_OK;
    });
}

// ... second half: wait for the OLDEST frame in flight, read its counters and timings. back_to_back: called by crt_hip_render
// right after render_begin -- the frame time is then the host's wall clock around both, as the reference times its
// render() (render_embree.cpp:177-211); otherwise the GPU's time between the frame's first and last event.
static int render_end(crt_hip_ctx *ctx, crt_render_stats *stats, bool back_to_back)
{
    return guarded(ctx, [&]() -> int {
        crt_hip_ctx::FrameSlot &fs = ctx->slots[ctx->oldest_slot];
        if (!fs.pending) {
            return fail(ctx, CRT_HIP_ESTATE, "render_end without a frame in flight");
        }
        using Span = crt_hip_ctx::Span;
        const std::vector<Span> &spans = fs.spans;
        const PassCounters *h_pc = fs.h_pc;
        const uint32_t pass = fs.passes;
        const bool timing = (ctx->flags & CRT_HIP_FLAG_TIMING) != 0;
        HIP_CHECK(hipEventSynchronize(fs.done));
        const auto t1 = std::chrono::high_resolution_clock::now();
        fs.pending = false;
        ctx->oldest_slot ^= 1;
        const uint32_t frame_id = fs.frame_id;
        crt_render_stats st;
        std::memset(&st, 0, sizeof(st));
        for (uint32_t p = 0; p < pass; ++p) {
            const PassCounters &pc = h_pc[p];
            for (int b = 0; b < MAX_PATH_DEPTH; ++b) {
                st.closest_rays += pc.n_queue[b].v;
                st.shadow_rays += (uint64_t)pc.n_shadow_a[b].v + pc.n_shadow_b[b].v;
                st.closest_rays_bounce[b] += pc.n_queue[b].v;
                st.shadow_rays_bounce[b] += (uint64_t)pc.n_shadow_a[b].v + pc.n_shadow_b[b].v;
                st.shadow_rays_elided += pc.n_shadow_elided[b].v;
            }
            st.closest_nodes += pc.nodes_closest;
            st.closest_tris += pc.tris_closest;
            st.shadow_nodes += pc.nodes_shadow;
            st.shadow_tris += pc.tris_shadow;
            st.closest_slots += pc.slots_closest;
            st.shadow_slots += pc.slots_shadow;
        }
        st.rays = st.closest_rays + st.shadow_rays + st.shadow_rays_elided; // REPORT_RAY_STATS semantics: every ray the reference issues
        if (std::getenv("CRT_HIP_DEBUG")) { // per-bounce queue sizes of the first pass
            const PassCounters &pc = h_pc[0];
            std::fprintf(stderr, "[crt_hip] frame %u worst closest ray: %u nodes, o (%.9g %.9g %.9g) d (%.9g %.9g %.9g) t %.9g\n",
                         frame_id, pc.max_ray_nodes, pc.worst_ray[0], pc.worst_ray[1], pc.worst_ray[2], pc.worst_ray[3],
                         pc.worst_ray[4], pc.worst_ray[5], pc.worst_ray[6]);
            for (int b = 0; b < MAX_PATH_DEPTH && (ctx->flags & CRT_HIP_FLAG_COUNTERS); ++b) {
                std::fprintf(stderr, "[crt_hip] frame %u closest launch %d: %.1f us total, queue drained after %.1f us (tail %.0f%%)\n",
                             frame_id, b, (pc.t_end[b] - pc.t_start[b]) / 100.0, (pc.t_drained[b] - pc.t_start[b]) / 100.0,
                             100.0 * (double)(pc.t_end[b] - pc.t_drained[b]) / (double)(pc.t_end[b] - pc.t_start[b]));
            }
            for (int k = 0; k < 2 && ((ctx->flags & CRT_HIP_FLAG_COUNTERS) || pc.prof_cycles[k][1] != 0); ++k) { // (a CRT_PHASE_PROFILE build fills them in every frame)
                static const char *names[4] = {"refill", "inner", "leaf", "retire"};
                double total = 0.0;
                for (int ph = 0; ph < 4; ++ph) {
                    total += (double)pc.prof_cycles[k][ph];
                }
                for (int ph = 0; ph < 4; ++ph) {
                    const double it = (double)std::max<unsigned long long>(1ull, pc.prof_iters[k][ph]);
                    std::fprintf(stderr, "[crt_hip] frame %u %s waves, %-6s: %5.1f%% of wave cycles, %.0f cycles/iteration, %.1f lanes/iteration\n",
                                 frame_id, k == 0 ? "closest" : "shadow ", names[ph],
                                 100.0 * (double)pc.prof_cycles[k][ph] / std::max(1.0, total),
                                 (double)pc.prof_cycles[k][ph] / it, (double)pc.prof_lanes[k][ph] / it);
                }
            }
            for (int b = 0; b < MAX_PATH_DEPTH; ++b) {
                std::fprintf(stderr, "[crt_hip] frame %u bounce %d: closest %u shadow_a %u shadow_b %u\n", frame_id,
                             b, pc.n_queue[b].v, pc.n_shadow_a[b].v, pc.n_shadow_b[b].v);
            }
        }
        if (back_to_back) {
            st.render_time_ms = (float)std::chrono::duration<double, std::milli>(t1 - fs.t0).count();
        } else {
            HIP_CHECK(hipEventElapsedTime(&st.render_time_ms, fs.begin, fs.done));
        }
        // one lane or two for frames of this size? (crt_hip_ctx::PassLane)
        st.passes = pass;
        st.pass_lanes = fs.used_lanes;
        if (ctx->lane_tune < crt_hip_ctx::LANE_TUNE_FRAMES && lanes_tunable(ctx, fs.total_paths)) {
            const int f = ctx->lane_tune;
            const float t = st.render_time_ms;
            if (f == 1 || f == 2) {
                ctx->lane_t1 = f == 1 ? t : std::min(ctx->lane_t1, t);
            } else if (f == 4 || f == 5) {
                ctx->lane_t2 = f == 4 ? t : std::min(ctx->lane_t2, t);
            }
            if (f == 2) {
                ctx->lane_choice = 2;
                ctx->capacity = 0; // the queues are carved again before the next frame
            } else if (f == 5) {
                if (!(ctx->lane_t2 < ctx->lane_t1)) {
                    ctx->lane_choice = 1;
                    ctx->capacity = 0;
                }
                if (std::getenv("CRT_HIP_DEBUG")) {
                    std::fprintf(stderr, "[crt_hip] pass lanes: %.3f ms with one, %.3f ms with two -> %d\n", ctx->lane_t1, ctx->lane_t2,
                                 ctx->lane_choice);
                }
            }
            ++ctx->lane_tune;
        }
        st.rays_per_second = (float)((st.rays - st.shadow_rays_elided) / (st.render_time_ms * 1.0e-3)); // rays really traced
        if (timing) {
            const bool dbg = std::getenv("CRT_HIP_DEBUG") != nullptr;
            for (const Span &sp : spans) {
                float ms = 0.f;
                HIP_CHECK(hipEventElapsedTime(&ms, fs.events[sp.a], fs.events[sp.b]));
                (sp.kind == 0 ? st.closest_ms : (sp.kind == 1 ? st.shadow_ms : st.shade_ms)) += ms;
                if (sp.bounce >= 0) {
                    (sp.kind == 0 ? st.closest_ms_bounce : (sp.kind == 1 ? st.shadow_ms_bounce : st.shade_ms_bounce))[sp.bounce] += ms;
                } else {
                    (sp.bounce == -1 ? st.raygen_ms : st.accumulate_ms) += ms;
                }
                if (dbg) {
                    std::fprintf(stderr, "[crt_hip] frame %u span kind %d: %.3f ms\n", frame_id, sp.kind, ms);
                }
            }
        }
        if (stats) {
            *stats = st;
        }
        return CRT_HIP_OK;
    });
}

int crt_hip_render_end(crt_hip_ctx *ctx, crt_render_stats *stats) { return render_end(ctx, stats, false); }

// RenderEmbree::render (render_embree.cpp:135-216): enqueue, wait, collect
int crt_hip_render(crt_hip_ctx *ctx, const float pos[3], const float dir[3], const float up[3], float fovy, int camera_changed,
                   int readback, crt_render_stats *stats)
{
    if (ctx && (ctx->slots[0].pending || ctx->slots[1].pending)) {
        return fail(ctx, CRT_HIP_ESTATE, "render: a frame enqueued with crt_hip_render_begin has not been collected");
    }
    const int rc = crt_hip_render_begin(ctx, pos, dir, up, fovy, camera_changed, readback);
    return rc != CRT_HIP_OK ? rc : render_end(ctx, stats, true);
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
            HIP_CHECK(hipStreamSynchronize(ctx->stream)); // the host image is the caller's as soon as this returns
        }
        // (without a read-back nothing waits: the assembled image stays in HBM -- crt_hip_device_framebuffer -- and the
        // kernel is ordered on the context's stream like every later launch; a frame loop that enqueues the next frame
        // before it collects this one keeps the GPU busy across the gather)
        return CRT_HIP_OK;
    });
}

int crt_hip_trace_rays(crt_hip_ctx *ctx, uint64_t n, const float *org, const float *dir, const float *tmin,
                       const float *tmax, int closest_and_flags, float *out_t, float *out_u, float *out_v, int32_t *out_inst,
                       int32_t *out_geom, int32_t *out_prim, crt_render_stats *stats)
{
    const int closest = closest_and_flags & 1;
    const bool production = (closest_and_flags & CRT_HIP_TRACE_PRODUCTION) != 0;
    return guarded(ctx, [&]() -> int {
        if (!ctx->has_scene) {
            return fail(ctx, CRT_HIP_ESTATE, "trace_rays before set_scene");
        }
        if (frames_in_flight(ctx)) {
            return fail(ctx, CRT_HIP_ESTATE, "trace_rays while a frame is in flight");
        }
        if (production) {
            return trace_rays_production(ctx, n, org, dir, tmin, tmax, closest != 0, out_t, out_u, out_v, out_inst, out_geom,
                                         out_prim, stats);
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
        d_ctr.alloc(32);
        hipStream_t s = ctx->stream;
        HIP_CHECK(hipMemcpyAsync(d_org.ptr, org, n * 12, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(d_dir.ptr, dir, n * 12, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(d_tmin.ptr, tmin, n * 4, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(d_tmax.ptr, tmax, n * 4, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemsetAsync(d_ctr.ptr, 0, 32, s));
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
        unsigned long long ctr[4] = {0, 0, 0, 0};
        HIP_CHECK(hipMemcpyAsync(ctr, d_ctr.ptr, 32, hipMemcpyDeviceToHost, s));
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
                stats->closest_slots = ctr[3];
            } else {
                stats->shadow_rays = n;
                stats->shadow_ms = ms;
                stats->shadow_nodes = ctr[0];
                stats->shadow_tris = ctr[1];
                stats->shadow_slots = ctr[3];
            }
        }
        return CRT_HIP_OK;
    });
}

// Diagnostics (include/crt_hip.h): the rays a frame left behind in lane 0's queues, AoS on the host side.
int crt_hip_debug_copy_queue(crt_hip_ctx *ctx, int which, uint64_t first, uint64_t n, float *out)
{
    return guarded(ctx, [&]() -> int {
        if (!out || ctx->capacity == 0 || which < 0 || which > 2 || first + n > ctx->capacity) {
            return fail(ctx, CRT_HIP_EINVAL, "debug_copy_queue: bad arguments (or no frame rendered yet)");
        }
        if (frames_in_flight(ctx)) {
            return fail(ctx, CRT_HIP_ESTATE, "debug_copy_queue while a frame is in flight");
        }
        const crt_hip_ctx::PassLane &l = ctx->lanes[0];
        const int n_fields = which == 2 ? 7 : 6;
        const float *src[7];
        for (int a = 0; a < 3; ++a) {
            src[a] = which == 2 ? l.sa.o[a] : l.q[which].o[a];
            src[3 + a] = which == 2 ? l.sa.d[a] : l.q[which].d[a];
        }
        src[6] = l.sa.tmax;
        std::vector<float> field(n);
        for (int k = 0; k < n_fields; ++k) {
            HIP_CHECK(hipMemcpy(field.data(), src[k] + first, n * sizeof(float), hipMemcpyDeviceToHost));
            for (uint64_t i = 0; i < n; ++i) {
                out[i * n_fields + k] = field[i];
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
        if (frames_in_flight(ctx)) {
            return fail(ctx, CRT_HIP_ESTATE, "kat while a frame is in flight");
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
        if (frames_in_flight(ctx)) {
            return fail(ctx, CRT_HIP_ESTATE, "bvh_copy while a frame is in flight");
        }
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (nodes) {
            HIP_CHECK(hipMemcpy(nodes, ctx->d_nodes.ptr, ctx->n_nodes * sizeof(PNode), hipMemcpyDeviceToHost));
        }
        if (tris) {
            HIP_CHECK(hipMemcpy(tris, ctx->d_slots.ptr, ctx->n_tris * sizeof(LeafSlot), hipMemcpyDeviceToHost));
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
