/* crt_hip.h — C-ABI of the MI355X wavefront path-tracing core (libcrt_hip_core.so).
 *
 * This is the drop-in boundary for ChameleonRT's render hot path. Every entry point
 * replaces one piece of the reference's `RenderBackend` contract as implemented by the
 * Embree backend; citations are into the reference tree (Twinklebear/ChameleonRT):
 *
 *   crt_hip_create / crt_hip_destroy   RenderEmbree::RenderEmbree / ~RenderEmbree
 *                                      (backends/embree/render_embree.cpp:19-31)
 *   crt_hip_initialize                 RenderBackend::initialize (util/render_backend.h:20,
 *                                      backends/embree/render_embree.cpp:38-56)
 *   crt_hip_set_scene                  RenderBackend::set_scene (util/render_backend.h:23,
 *                                      backends/embree/render_embree.cpp:58-133)
 *   crt_hip_render                     RenderBackend::render (util/render_backend.h:26-31,
 *                                      backends/embree/render_embree.cpp:135-216)
 *   crt_hip_framebuffer                RenderBackend::img (util/render_backend.h:13)
 *   crt_hip_name                       RenderBackend::name (util/render_backend.h:18)
 *
 * The C++ plugin shim `backends/hip/render_hip.{h,cpp}` (struct RenderHIP : RenderBackend)
 * and the Python ctypes binding `chameleonrt_amd/core.py` both sit on exactly these
 * symbols. Plain pointers and sizes only; no C++ or torch types cross this boundary.
 *
 * Conventions: every function returning int returns 0 on success and a negative
 * CRT_HIP_E* code on failure; nothing throws across the boundary. `crt_hip_last_error`
 * gives the message. A context is single-threaded (the reference calls its backend from
 * the main thread only, main.cpp:231-380). There is no CPU fallback: if no gfx950 device
 * is usable, crt_hip_create fails.
 */
#ifndef CRT_HIP_H
#define CRT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRT_HIP_ABI_VERSION 4
#define CRT_HIP_MAX_PATH_DEPTH 5 /* MAX_PATH_DEPTH, backends/embree/util.ih:10 */

enum {
    CRT_HIP_OK = 0,
    CRT_HIP_EINVAL = -1,  /* bad argument / malformed scene */
    CRT_HIP_EDEVICE = -2, /* HIP runtime error (no device, OOM, launch failure) */
    CRT_HIP_ESTATE = -3   /* call order violated (render before initialize/set_scene) */
};

/* Image colour spaces, util/material.h:9 (enum ColorSpace { LINEAR, SRGB }). */
enum { CRT_COLORSPACE_LINEAR = 0, CRT_COLORSPACE_SRGB = 1 };

/* One `Geometry` (util/mesh.h:6-12). Per-vertex normals are not part of the hot path:
 * the reference ignores them (render_embree.ispc:269-270), so they are not passed. */
typedef struct crt_geometry_desc {
    const float *vertices;   /* n_vertices * 3 floats (glm::vec3) */
    uint64_t n_vertices;
    const uint32_t *indices; /* n_triangles * 3 (glm::uvec3) */
    uint64_t n_triangles;
    const float *uvs;        /* n_vertices * 2 floats (glm::vec2) or NULL */
} crt_geometry_desc;

/* One `Mesh` (util/mesh.h:14-22): a contiguous range of the geometry array. The position
 * of a geometry inside its mesh is Embree's geomID (embree_utils.cpp:63-76). */
typedef struct crt_mesh_desc {
    uint32_t first_geometry;
    uint32_t n_geometries;
} crt_mesh_desc;

/* One `ParameterizedMesh` (util/mesh.h:28-36): one material id per geometry. */
typedef struct crt_parameterized_mesh_desc {
    uint32_t mesh_id;
    uint32_t n_material_ids;
    const uint32_t *material_ids;
} crt_parameterized_mesh_desc;

/* One `Instance` (util/mesh.h:40-47): glm::mat4 is column-major, m[c*4+r]. */
typedef struct crt_instance_desc {
    float transform[16];
    uint32_t parameterized_mesh_id;
} crt_instance_desc;

/* One `Image` (util/material.h:11-27). */
typedef struct crt_image_desc {
    int32_t width, height, channels;
    int32_t color_space; /* CRT_COLORSPACE_* */
    const uint8_t *data; /* width*height*channels bytes, row 0 first */
} crt_image_desc;

/* `Scene` as handed to set_scene (util/scene.h:23-32). All pointers are borrowed for the
 * duration of crt_hip_set_scene only; the core copies what it needs (the reference
 * destroys its Scene right after set_scene, main.cpp:185-214).
 *
 * materials: n_materials * 16 floats, the 64-byte `DisneyMaterial` of util/material.h:29-46
 *   (base_color.rgb, metallic, specular, roughness, specular_tint, anisotropy, sheen,
 *   sheen_tint, clearcoat, clearcoat_gloss, ior, specular_transmission, pad.xy). Textured
 *   parameters are encoded in the float bits per util/texture_channel_mask.h:16-23.
 * lights: n_lights * 20 floats, the 80-byte `QuadLight` of util/lights.h:6-18
 *   (emission.xyzw, position.xyzw, normal.xyzw, v_x.xyz, width, v_y.xyz, height). */
typedef struct crt_scene_desc {
    const crt_geometry_desc *geometries;
    uint32_t n_geometries;
    const crt_mesh_desc *meshes;
    uint32_t n_meshes;
    const crt_parameterized_mesh_desc *parameterized_meshes;
    uint32_t n_parameterized_meshes;
    const crt_instance_desc *instances;
    uint32_t n_instances;
    const float *materials;
    uint32_t n_materials;
    const crt_image_desc *textures;
    uint32_t n_textures;
    const float *lights;
    uint32_t n_lights;
    uint32_t samples_per_pixel; /* Scene::samples_per_pixel, util/scene.h:31 */
} crt_scene_desc;

/* RenderStats (util/render_backend.h:7-10) plus what the roofline needs.
 * rays follow REPORT_RAY_STATS semantics (render_embree.ispc:145-147,171-173,246-248):
 * one per closest-hit trace and one per occlusion trace, misses included. Counted in
 * 64 bits (the reference's uint16/int accumulation overflows at 4K/64spp). */
typedef struct crt_render_stats {
    float render_time_ms;  /* all kernels of the frame incl. tonemap + stat reduction */
    float rays_per_second; /* rays really traced / (render_time_ms * 1e-3) (= rays / ... without the elision flag) */
    uint64_t rays;
    uint64_t closest_rays; /* rays traced by the closest-hit traversal kernel */
    uint64_t shadow_rays;  /* rays traced by the any-hit traversal kernel */
    float closest_ms;      /* summed duration of the closest-hit traversal launches */
    float shadow_ms;       /* summed duration of the any-hit traversal launches */
    float shade_ms;        /* raygen + shade + accumulate launches */
    /* only when CRT_HIP_FLAG_COUNTERS: nodes fetched / triangles tested by traversal */
    uint64_t closest_nodes, closest_tris, shadow_nodes, shadow_tris;
    /* (ABI 2) the same per path-loop iteration b = 0..4 (render_embree.ispc:243-336): rays entering the
     * closest-hit / any-hit launch of that bounce, and -- with CRT_HIP_FLAG_TIMING -- the duration of its
     * closest-hit, any-hit and shade launches, summed over the passes of the frame. */
    uint64_t closest_rays_bounce[CRT_HIP_MAX_PATH_DEPTH], shadow_rays_bounce[CRT_HIP_MAX_PATH_DEPTH];
    float closest_ms_bounce[CRT_HIP_MAX_PATH_DEPTH], shadow_ms_bounce[CRT_HIP_MAX_PATH_DEPTH],
        shade_ms_bounce[CRT_HIP_MAX_PATH_DEPTH];
    float raygen_ms, accumulate_ms;
    /* only when CRT_HIP_FLAG_COUNTERS: 64-byte leaf slots fetched (each holds one or two triangles) */
    uint64_t closest_slots, shadow_slots;
    /* (ABI 3) how the frame was cut: passes (each at most the path capacity) and the pass lanes they ran on -- 1, or 2
     * where the library's trial found two faster for frames of this size (bit-identical images either way) */
    uint32_t passes, pass_lanes;
    /* (ABI 4) CRT_HIP_FLAG_ELIDE_UNUSED_SHADOW_RAYS: occlusion rays the reference issues and counts but whose result it never
     * looks at, not traced here. `rays` keeps the reference's count (closest_rays + shadow_rays + shadow_rays_elided);
     * `shadow_rays` and `shadow_rays_bounce` are the rays the any-hit kernel really traced. 0 without the flag. */
    uint64_t shadow_rays_elided;
} crt_render_stats;

typedef struct crt_hip_ctx crt_hip_ctx;

enum {
    CRT_HIP_FLAG_NONE = 0,
    CRT_HIP_FLAG_COUNTERS = 1, /* count BVH nodes / triangles touched (instrumented kernels) */
    CRT_HIP_FLAG_TIMING = 2,   /* per-kernel-class HIP event timing in crt_render_stats */
    /* Opt-in, off by default: do not trace the next-event occlusion rays whose result cannot reach the image. The reference
     * (render_embree.ispc:131-153) traces the light-sample ray of EVERY hit, counts it, and then uses its answer only if
     * light_pdf >= EPSILON && bsdf_pdf >= EPSILON; where those fail -- the light's back side, the wrong hemisphere of an
     * opaque surface -- or the BSDF evaluates to exactly zero, the ray's contribution is an exact zero whatever it hits.
     * With this flag such a ray (and only a hit's single ray: a hit with a second, BSDF-sampled occlusion ray keeps both) is
     * dropped in the shading kernel: accumulated radiance, RGBA8 and the per-pixel ray counts are bit-identical to the
     * default path's (tests/test_gpu_elide.py), the any-hit kernel has less to do. Ray statistics keep the reference's
     * semantics (crt_render_stats::rays); the rays not traced are reported apart (shadow_rays_elided). The default traces
     * every ray the reference traces. */
    CRT_HIP_FLAG_ELIDE_UNUSED_SHADOW_RAYS = 4,
    /* Opt-in (the C++ plugin sets it unless CRT_HIP_REFINE=0): crt_hip_set_scene returns as soon as a QUICKLY built tree is resident
     * -- the host SAH tree without its re-insertion passes, or the device builder's linear tree with CRT_HIP_BUILD=device -- and a
     * thread of the context builds the full-quality tree (what set_scene otherwise makes the caller wait for: rtcCommitScene in
     * the reference, embree_utils.cpp:63-76,121-129), uploads it next to the quick one, and the first crt_hip_render_begin after
     * that swaps it in between two frames. Images do not depend on the tree (closest hit = lexicographic minimum of (t, inst, geom,
     * prim), occlusion = boolean), so the accumulation goes on undisturbed: frames with the flag are bit-identical to frames
     * without it (tests/test_gpu_refine.py); only set_scene's latency and the first seconds' frame times differ.
     * crt_hip_set_prepared_scene (multi-GPU: one preparation per node) is not affected. */
    CRT_HIP_FLAG_REFINE_IN_BACKGROUND = 8
};

int crt_hip_abi_version(void);
int crt_hip_device_count(void);

/* device_id: HIP device ordinal. NULL on failure (message via crt_hip_last_error(NULL)). */
crt_hip_ctx *crt_hip_create(int device_id, uint32_t flags);
void crt_hip_destroy(crt_hip_ctx *ctx);
const char *crt_hip_last_error(const crt_hip_ctx *ctx);
const char *crt_hip_name(const crt_hip_ctx *ctx);

/* Run all work of this context on an existing HIP stream (hipStream_t passed as void*),
 * e.g. torch.cuda.current_stream().cuda_stream. NULL = the context's own stream. */
int crt_hip_set_stream(crt_hip_ctx *ctx, void *hip_stream);

/* Image-tile partition for multi-GPU rendering (new functionality): the
 * framebuffer is cut into the reference's 64x64 tiles (render_embree.h:25) and this
 * context renders the tiles with tile_id % world == rank. Must precede initialize. */
int crt_hip_set_partition(crt_hip_ctx *ctx, int rank, int world);

int crt_hip_initialize(crt_hip_ctx *ctx, int fb_width, int fb_height);
int crt_hip_set_scene(crt_hip_ctx *ctx, const crt_scene_desc *scene);
/* CRT_HIP_FLAG_REFINE_IN_BACKGROUND: 0 = no refinement (flag off, or a scene too small to bother), 1 = the better tree is being
 * built, 2 = built and resident, waiting for the next crt_hip_render_begin, 3 = in use, -1 = failed (crt_hip_last_error(ctx); the
 * quick tree stays in use). quick_ms / full_ms (may be NULL): how long set_scene's own tree took, and the background one. */
int crt_hip_refine_state(crt_hip_ctx *ctx, double *quick_ms, double *full_ms);

/* set_scene in two halves, for the multi-GPU case (new functionality): the host half
 * (BLAS/TLAS build = the reference's rtcCommitScene, embree_utils.cpp:63-76,121-129; 8-bit texture
 * linearisation, render_embree.cpp:90-104; material and light tables) runs ONCE per node and its
 * result is uploaded to every GPU's context, instead of N identical builds oversubscribing the
 * host. crt_hip_set_scene(ctx, s) == prepare + set_prepared + free. n_threads 0 = the host cores this
 * process may use (affinity mask and cgroup quota honoured; CRT_HIP_BUILD_THREADS overrides).
 * prepare/load return NULL on failure (message via crt_hip_last_error(NULL)). save/load move a
 * prepared scene between the processes of one node through a file (meant for /dev/shm): rank 0
 * prepares and saves, the other ranks load -- same build, same machine, not an exchange format. */
typedef struct crt_hip_prepared_scene crt_hip_prepared_scene;
crt_hip_prepared_scene *crt_hip_prepare_scene(const crt_scene_desc *scene, int n_threads);
/* The same with the tree of every large mesh -- and the world tree of an instanced scene -- built ON HIP device
 * `build_device` (linear BVH -- Morton sort, binary radix tree, bottom-up boxes, collapse to the same
 * 4-wide nodes; bvh_device.hip)
 * instead of by the host SAH builder: set_scene of a 10 M-triangle scene in a fraction of the time, at
 * a lower tree quality. build_device < 0: host build. crt_hip_set_scene takes this
 * path when CRT_HIP_BUILD=device is set. */
crt_hip_prepared_scene *crt_hip_prepare_scene_on(const crt_scene_desc *scene, int n_threads, int build_device);
void crt_hip_free_prepared_scene(crt_hip_prepared_scene *prepared);
int crt_hip_set_prepared_scene(crt_hip_ctx *ctx, const crt_hip_prepared_scene *prepared);
int crt_hip_save_prepared_scene(const crt_hip_prepared_scene *prepared, const char *path);
/* Introspection of the host-built arrays (no device needed: the CPU tests check the builder and the
 * quantiser this way). Same records as crt_hip_bvh_info / _layout / _copy / _copy_instances below.
 * two_level: 0 = one instance, its BLAS traversed directly; 1 = a top-level tree over the instances (what Embree
 * builds, embree_utils.cpp:90-129); 2 = a "world tree": ONE tree in world space over per-instance copies of the
 * triangle records, whose last word is (instance << 1) | identity -- a ray is transformed into an instance's object
 * space only to test a triangle of it (same hits bit for bit; chosen when the instanced triangles fit a memory
 * budget, CRT_HIP_LEVELS=two|world overrides; chameleonrt_amd/csrc/crt_types.h LEVELS_WORLD_TREE). */
int crt_hip_prepared_scene_info(const crt_hip_prepared_scene *prepared, uint64_t *n_nodes, uint64_t *n_tris,
                                uint64_t *n_instances, int32_t *two_level, float *root_frame, int32_t *root,
                                uint32_t *n_top_nodes, uint32_t *stack_need, double *build_ms);
int crt_hip_prepared_scene_copy(const crt_hip_prepared_scene *prepared, void *nodes, void *tris, void *instances);
/* Scene::samples_per_pixel of a prepared scene (the one thing bench.py varies between its strong-
 * and weak-scaling legs; everything else in a prepared scene is immutable). */
int crt_hip_prepared_scene_set_spp(crt_hip_prepared_scene *prepared, uint32_t samples_per_pixel);
int crt_hip_child_order(void); /* the build's CRT_CHILD_ORDER (see crt_hip_bvh_layout) */
/* Scenes with more than one instance: the index of the instance whose triangles were made leaves of the
 * top-level tree itself (an identity instance whose mesh no other instance uses -- the static part of the
 * scene; DESIGN.md "Traversal"), or -1. In the top-level tree a leaf whose 3-bit count field is 7 is an
 * instance, any other leaf holds triangles of that instance. */
int32_t crt_hip_prepared_scene_world_instance(const crt_hip_prepared_scene *prepared);
int32_t crt_hip_world_instance(crt_hip_ctx *ctx); /* the same of the scene the context holds */
uint32_t crt_hip_lds_stack_entries(int two_level); /* per-lane traversal-stack entries kept in LDS by the kernels of that kind of scene (0 / 1 / 2 as above); deeper ones live in HBM */
crt_hip_prepared_scene *crt_hip_load_prepared_scene(const char *path);

/* One frame. fovy in degrees; camera_changed resets accumulation (frame_id = 0). When
 * readback != 0 the RGBA8 image is copied to the host buffer behind crt_hip_framebuffer
 * (for world > 1 only this rank's tiles are valid; see crt_hip_assemble_tiles). */
int crt_hip_render(crt_hip_ctx *ctx, const float pos[3], const float dir[3], const float up[3],
                   float fovy_deg, int camera_changed, int readback, crt_render_stats *stats);

/* The two halves of crt_hip_render for callers that must not stall the GPU between frames (the multi-GPU frame loop:
 * gather + assemble of frame f run while frame f+1 is traced). crt_hip_render_begin enqueues every launch of the frame on
 * the context's stream(s) and returns without waiting; crt_hip_render_end waits for the OLDEST frame in flight and fills
 * its statistics (render_time_ms is then the GPU's time from the frame's first to its last event). At most two frames may
 * be in flight; results are identical to crt_hip_render's (same launches, same order on the stream). While frames are in
 * flight only render_begin / render_end / tile_buffer / assemble_tiles / device_framebuffer may be called (the others
 * return CRT_HIP_ESTATE), and a frame with readback = 1 is refused while another is pending (there is one host image). The tile
 * buffer of a frame (crt_hip_tile_buffer right after its render_begin) is reused by the frame after the next one:
 * its gather must have been ordered before that frame is enqueued. No reference counterpart (the reference's render()
 * is synchronous, render_embree.cpp:135-216). */
int crt_hip_render_begin(crt_hip_ctx *ctx, const float pos[3], const float dir[3], const float up[3], float fovy_deg,
                         int camera_changed, int readback);
int crt_hip_render_end(crt_hip_ctx *ctx, crt_render_stats *stats);

/* W*H RGBA8 (8-bit sRGB, A=255), row 0 = top: RenderBackend::img. */
const uint32_t *crt_hip_framebuffer(const crt_hip_ctx *ctx);

/* Display interop (the reference's GLNativeRenderer path, util/display/gldisplay.h:35-37,
 * backends/optix/render_optix.cpp:104-121,410-426): the row-major RGBA8 image as it sits in HBM after
 * crt_hip_render (world == 1) or crt_hip_assemble_tiles (world > 1), so that a display can copy it
 * device-to-device into a registered GL texture (hipGraphicsGLRegisterImage + hipMemcpy2DToArray) and
 * call render() with readback = 0 -- no host round trip. The pointer stays valid until the next
 * crt_hip_initialize; the data is complete once the context's stream has been synchronised (render and
 * assemble_tiles return synchronised). */
int crt_hip_device_framebuffer(crt_hip_ctx *ctx, void **device_ptr, size_t *pitch_bytes);

/* Parity/diagnostic reads (synchronise the stream). Row-major W*H. */
int crt_hip_read_accum(crt_hip_ctx *ctx, float *rgb /* W*H*3 */);
int crt_hip_read_ray_counts(crt_hip_ctx *ctx, uint32_t *counts /* W*H, last frame */);
uint32_t crt_hip_frame_id(const crt_hip_ctx *ctx);

/* Multi-GPU assembly. Each rank exposes the compact tile-major RGBA8 buffer of the frame it rendered
 * LAST (n_local_tiles_padded * 64*64 uint32, same size on every rank) as a device pointer -- two
 * buffers alternate by frame parity, so the gather of frame f may overlap the tracing of frame f+1
 * and must have completed before frame f+2 is rendered; the
 * caller gathers them (RCCL gather/all_gather) into world consecutive slabs on the root,
 * which un-permutes them into its row-major image (kernel K8). */
int crt_hip_tile_buffer(crt_hip_ctx *ctx, void **device_ptr, size_t *n_bytes);
int crt_hip_assemble_tiles(crt_hip_ctx *ctx, const void *gathered_device_ptr, int world,
                           int readback);

/* ---- Diagnostic entry points used by the parity tests and the roofline bench ---- */

/* Trace n arbitrary world-space rays through the scene with the traversal kernels. Host arrays; tmin
 * must hold one value for the whole batch (inside a frame it is 0 for primary rays and EPSILON for all
 * others, util.ih:8). closest: out_t/out_u/out_v/out_inst/out_geom/out_prim (inst = -1 on a miss).
 * any-hit (closest == 0): out_t[i] = 1 if the segment (tmin, tmax] is unoccluded else 0, other outputs
 * may be NULL.
 * `closest` bit 1 (CRT_HIP_TRACE_PRODUCTION) selects WHICH instantiation runs: clear = the instrumented
 * diagnostic kernel (counts nodes / triangles into stats); set = the very kernels a frame launches
 * (k_trace_closest / k_trace_shadow without counters, fed through a PathQueue / ShadowQueueA and read back
 * from the HitBuf / radiance buffer like k_shade reads them), which requires tmin = 0 or EPSILON and, for
 * closest hits, tmax = 1e20 (set_ray_hit, util.ih:118) for every ray; out_inst is then the hit's instance as K2
 * resolves it for its normal and material look-up (the frame's 32-byte hit record does not carry it: this call asks
 * K2's retire for a copy, HitBuf::inst_debug) and stats carry no counters. */
#define CRT_HIP_TRACE_PRODUCTION 2
int crt_hip_trace_rays(crt_hip_ctx *ctx, uint64_t n, const float *org /* n*3 */,
                       const float *dir /* n*3 */, const float *tmin, const float *tmax,
                       int closest, float *out_t, float *out_u, float *out_v,
                       int32_t *out_inst, int32_t *out_geom, int32_t *out_prim,
                       crt_render_stats *stats);

/* Diagnostics: copy rays out of the queues the last rendered frame left behind (lane 0 of the pass lanes), as
 * records of floats on the host: which = 0 / 1: PathQueue buffer 0 / 1 (bounce b's closest-hit rays sit in buffer
 * b & 1), 6 floats per ray {o, d}; which = 2: ShadowQueueA (the last bounces' occlusion rays), 7 floats {o, d, tmax}.
 * Used by tools/gpu_sort_probe.py to re-trace a frame's real incoherent rays in other orders (crt_hip_trace_rays with
 * CRT_HIP_TRACE_PRODUCTION). No reference counterpart. */
int crt_hip_debug_copy_queue(crt_hip_ctx *ctx, int which, uint64_t first, uint64_t n, float *out);

/* Known-answer tests of the device shading functions: run device function `fn` on n
 * input records of in_stride floats, producing out_stride floats each (see
 * include/crt_kat.h for the record layouts). */
int crt_hip_kat(crt_hip_ctx *ctx, int fn, uint64_t n, const float *in, int in_stride,
                float *out, int out_stride);

/* BVH introspection for tests: copy out the traversal arrays the kernels use. Nodes are
 * 64-byte packed 4-wide records (48 bytes used: per-axis origin and scale on the BVH's 16-bit grid, one byte per
 * child plane, four references), leaves 64-byte slots of one or two triangles -- n_tris counts SLOTS
 * (DESIGN.md "Data layout in HBM");
 * root_frame receives the 6 floats {base.xyz, step.xyz} of the root BVH's fixed-point frame. */
int crt_hip_bvh_info(crt_hip_ctx *ctx, uint64_t *n_nodes, uint64_t *n_tris,
                     uint64_t *n_instances, int32_t *two_level, float *root_frame);
int crt_hip_bvh_copy(crt_hip_ctx *ctx, void *nodes, void *tris);
/* root: node index traversal starts at; n_top_nodes: BFS-ordered top levels staged in LDS;
 * stack_need: traversal-stack entries the deepest path of this BVH can need (the HBM slab behind the
 * lds_stack entries kept in LDS is sized from it); child_order: the build's visit rule (0 = entered
 * children fully sorted by entry distance, 1 = nearest first, rest in slot order). */
int crt_hip_bvh_layout(crt_hip_ctx *ctx, int32_t *root, uint32_t *n_top_nodes, uint32_t *stack_need,
                       uint32_t *lds_stack, int32_t *child_order);
/* n_instances 128-byte instance records (affine world_to_object[12], blas_root, identity, frame[6],
 * geom_base, mat_base, pad[10]) as the two-level traversal reads them. */
int crt_hip_bvh_copy_instances(crt_hip_ctx *ctx, void *instances);

#ifdef __cplusplus
}
#endif
#endif /* CRT_HIP_H */
