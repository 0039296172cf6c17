/* crt_oracle.h — C API of the CPU oracle (liborc.so).
 *
 * TEST INFRASTRUCTURE ONLY. The oracle is a scalar CPU restatement of the reference's
 * Embree-backend path tracer. Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it, and only as the checker. Nothing under chameleonrt_amd/
 * or backends/hip/ may include, link or call it.
 *
 * PARITY STATUS
 *  - PINNED, bit for bit, for everything the reference defines: RNG keying, camera rays, the path
 *    loop, material unpacking and textures, the Disney BSDF, quad lights, NEE + MIS, Russian
 *    roulette, the running mean, sRGB8 and the ray statistics. The pin is the reference's own
 *    kernel source (backends/embree_sycl/render_embree_kernel.inl and the headers it includes:
 *    the C++ twin of the ISPC files this restatement follows), compiled from /root/reference by
 *    `make -C oracle ref` and run on Cornell, a textured scene and a 64-instance scene with
 *    glass: accumulated radiance, RGBA8 and per-pixel ray counts are identical to the last bit
 *    (tests/test_oracle_pinned.py; tests/golden/ref_*.npz carry its frames to machines without
 *    the reference tree). The integer RNG is also pinned against an independent
 *    MurmurHash3/LCG implementation (tests/test_oracle_rng.py).
 *  - "parity unpinned" at the ray/triangle boundary only. The reference delegates BVH build,
 *    traversal and triangle intersection to Embree 4 (pinned 4.0.1 in
 *    .github/workflows/cmake.yml:12; call sites backends/embree/render_embree.ispc:144,170,245),
 *    which is not in /root/reference and not installable here; the pinning run above plugs the
 *    oracle's own intersector in where the kernel calls rtcIntersect1 / rtcOccluded1, and
 *    GLM's matrix inverse (third-party too: GLM 0.9.9.8 per cmake/glm.cmake, absent here) is restated from its published
 *    source (glm/detail/func_matrix.inl, compute_inverse<4, 4>), operation for operation. The ISPC build's `--opt=fast-math`
 *    and approximate transcendentals are not reproduced by any C++ build either.
 */
#ifndef CRT_ORACLE_H
#define CRT_ORACLE_H

#include "../include/crt_hip.h" /* scene POD types only */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;
typedef struct orc_renderer orc_renderer;

typedef struct orc_stats {
    double render_time_ms;
    double rays_per_second;
    uint64_t rays;
    uint64_t closest_rays, shadow_rays;
    uint64_t nodes_visited, tris_tested; /* of the oracle's own BVH */
} orc_stats;

orc_scene *orc_scene_create(const crt_scene_desc *desc);
void orc_scene_destroy(orc_scene *s);
uint64_t orc_scene_num_triangles(const orc_scene *s);

orc_renderer *orc_renderer_create(orc_scene *s, int fb_width, int fb_height, int n_threads);
void orc_renderer_destroy(orc_renderer *r);

/* RenderEmbree::render (backends/embree/render_embree.cpp:135-216). tile_begin/tile_end
 * restrict the frame to a range of the reference's 64x64 tiles (row-major tile ids) so
 * the CPU baseline can time a bounded sample; pass 0, -1 for the whole frame. */
int orc_render(orc_renderer *r, const float pos[3], const float dir[3], const float up[3],
               float fovy_deg, int camera_changed, int tile_begin, int tile_end,
               orc_stats *stats);
/* The same for an explicit LIST of tiles (a fixed sample spread over the image: bench.py's cpu_baseline and in-run
 * parity check, the full-configuration tile parity tests). Tiles not named keep what they held. */
int orc_render_tiles(orc_renderer *r, const float pos[3], const float dir[3], const float up[3],
                     float fovy_deg, int camera_changed, const int *tile_ids, int n_tiles, orc_stats *stats);
const uint32_t *orc_framebuffer(const orc_renderer *r);
int orc_read_accum(const orc_renderer *r, float *rgb /* W*H*3 row-major */);
/* Debug aid: re-trace one pixel of frame `frame_id` printing the per-bounce state to stderr. */
int orc_debug_pixel(orc_renderer *r, const float pos[3], const float dir[3], const float up[3], float fovy_deg,
                    uint32_t frame_id, int x, int y);
int orc_read_ray_counts(const orc_renderer *r, uint32_t *counts /* W*H */);
int orc_num_tiles(const orc_renderer *r);

/* rtcIntersectV / rtcOccludedV stand-in. brute_force != 0 tests every triangle of every
 * instance (no BVH): the definition of the right answer. Outputs as crt_hip_trace_rays. */
int orc_trace_rays(const orc_scene *s, uint64_t n, const float *org, const float *dir,
                   const float *tmin, const float *tmax, int closest, int brute_force,
                   float *out_t, float *out_u, float *out_v, int32_t *out_inst,
                   int32_t *out_geom, int32_t *out_prim, orc_stats *stats);

/* glm::inverse (GLM 0.9.9.8's compute_inverse<4,4>, restated) used for Instance::world_to_object (embree_utils.cpp:97). 1 = invertible. */
int orc_invert4x4(const float m[16], float out[16]);

/* The same stand-in for one ray, on the calling thread: 1 = hit / occluded, 0 = not. */
int orc_intersect1(const orc_scene *s, const float org[3], const float dir[3], float tnear, float tfar,
                   float *t, float *u, float *v, int32_t *inst, int32_t *geom, int32_t *prim);
int orc_occluded1(const orc_scene *s, const float org[3], const float dir[3], float tnear, float tfar);

/* Walk a FOREIGN BVH (the product's 64-byte packed 4-wide nodes + frames / 64-byte leaf slots of 1-2 triangles /
 * 128-byte instance records) with the product's documented visit rule (child_order =
 * the product's CRT_CHILD_ORDER build setting): counts nodes fetched / triangles tested, to
 * cross-check the HIP kernels' CRT_HIP_FLAG_COUNTERS numbers (the roofline input), reports the
 * deepest traversal stack any ray needed, and optionally returns the hits (closest: t / inst / geom /
 * prim, miss = -1 ids and t = tmax; occlusion: out_t = 1 visible, 0 occluded). instances == NULL or
 * n_instances <= 1: single-level walk from `root` (instance 0's transform applied if it is given and
 * not the identity). levels = the product's SceneView::two_level (0 one instance, 1 two-level, 2 one tree in world
 * space whose triangle records carry their instance; < 0: 0 or 1 by instance count). inst_entries: how many times a
 * ray was transformed into an instance's space. Outputs other than the two counters may be NULL. */
int orc_walk_foreign_bvh(const void *nodes, const void *tris, const void *instances, uint64_t n_instances,
                         int32_t world_inst, int32_t root, const float root_frame[6], int child_order, uint64_t n, const float *org,
                         const float *dir, const float *tmin, const float *tmax, int closest,
                         uint64_t *nodes_visited, uint64_t *tris_tested, uint32_t *max_stack, float *out_t,
                         int32_t *out_inst, int32_t *out_geom, int32_t *out_prim, uint64_t *inst_entries,
                         int levels, uint64_t *leaf_slots /* 64-byte leaf slots fetched (each holds 1-2 triangles), or NULL */);

/* Shading-function KATs, record layouts in include/crt_kat.h. scene may be NULL for the
 * functions that need none. */
int orc_kat(const orc_scene *s, int fn, uint64_t n, const float *in, int in_stride, float *out,
            int out_stride);

/* TEST switch (default 0 = the reference's pow(x, 5), disney_bsdf.ih:74-76): form schlick_weight's fifth power by three
 * multiplications, as the product does, so that a test can show the Disney KAT's tolerance is that one documented deviation
 * (tests/test_gpu_kat.py::test_disney_eval_is_bit_exact_once_the_oracle_forms_schlick_like_the_product). Process-wide. */
void orc_set_schlick_by_multiplication(int on);

#ifdef __cplusplus
}
#endif
#endif
