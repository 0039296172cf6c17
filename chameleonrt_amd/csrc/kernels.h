// kernels.h — host-callable launchers of the wavefront kernels (implemented in kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "crt_types.h"
#include "wavefront.h"

namespace crt {

// BFS-ordered top BVH levels the traversal kernels stage in LDS (48 B each). Round 4: 85 nodes (four full levels) -> 5 (the root
// and its children): the 4 KB are worth more as four more entries of every lane's stack -- C4 55.4 -> 54.6 ms; 0 nodes measured the
// same then (profiles/r04_issue_bound_ab.txt). Round 6: NONE -- every instruction of the inner step counts (profiles/
// r06_fast_push_ab.txt), and the LDS-or-HBM test of a node's address is five of them for a fetch that hits L2 anyway: C3 7.89 -> 7.82
// ms, C4 50.95 -> 50.85, C2 +-0 on top of the branch-free pushes. The builder is asked for this many nodes in BFS order.
#ifndef CRT_MAX_TOP_NODES
#define CRT_MAX_TOP_NODES 0
#endif
// two-level scenes: top levels of the TLAS staged in LDS: none. The LDS of the two-level kernels also holds the cold
// ray state and a traversal stack that runs much deeper than in a single tree (26 entries on the instanced C4), and
// LDS spent on stack entries pays more than LDS spent on node copies (85 nodes + 10 entries -> 5 nodes + 15 entries:
// C4 -2 % frame time; 5 -> 0 nodes, which also takes the LDS-or-HBM branch out of the inner loop: -0.8 %)
#ifndef CRT_MAX_TOP_NODES_TWO_LEVEL
#define CRT_MAX_TOP_NODES_TWO_LEVEL 0
#endif

struct LaunchCfg {
    hipStream_t stream;
    int n_cus;      // compute units of the device (grid sizing)
    bool counters;  // instrumented traversal (CRT_HIP_FLAG_COUNTERS)
    bool elide = false; // CRT_HIP_FLAG_ELIDE_UNUSED_SHADOW_RAYS (k_shade)
};

// Geometry of the persistent traversal grid (sizes the stack-overflow slab in SceneView).
uint32_t traversal_grid_threads(int n_cus);
uint32_t traversal_lds_stack(uint32_t levels); // per-lane stack entries kept in LDS; deeper ones go to the HBM slab
int traversal_child_order();    // the build's CRT_CHILD_ORDER (the oracle's BVH walker mirrors the rule)

// K1: primary rays for `n_paths` pixel-samples starting at local pixel slot `slot0`.
void launch_raygen(const LaunchCfg &cfg, const ViewParams &vp, const uint32_t *tile_ids, uint32_t slot0,
                   uint32_t n_paths, PathQueue q, float4 *radiance, PassCounters *pc);
// K2: closest-hit traversal of the rays of bounce `bounce` (persistent waves, dynamic fetch).
void launch_trace_closest(const LaunchCfg &cfg, const SceneView &sc, PathQueue q, HitBuf hits,
                          PassCounters *pc, int bounce);
// K3: hit shading: material unpack, NEE set-up, BSDF sampling, Russian roulette, compaction.
void launch_shade(const LaunchCfg &cfg, const SceneView &sc, PathQueue qin, HitBuf hits, PathQueue qout,
                  ShadowQueueA sa, ShadowQueueB sb, float4 *radiance, PassCounters *pc, int bounce,
                  uint32_t n_paths_max /* paths of the pass: the queue cannot be longer */);
// K4: any-hit traversal of the NEE occlusion rays (light sample, then the rare BSDF-sample ray of
// the same hit, by the same lane).
void launch_trace_shadow(const LaunchCfg &cfg, const SceneView &sc, ShadowQueueA sa, ShadowQueueB sb,
                         float4 *radiance, PassCounters *pc, int bounce);
// K5: per-pixel sample sum, running mean over frames, sRGB8, ray statistics.
void launch_accumulate(const LaunchCfg &cfg, const ViewParams &vp, const uint32_t *tile_ids, uint32_t slot0,
                       uint32_t n_slots, const float4 *radiance, float4 *accum, uint32_t *tile_fb,
                       uint32_t *img_rowmajor, uint32_t *ray_counts);
// K8: un-permute `world` gathered compact tile buffers into the row-major image.
void launch_assemble(const LaunchCfg &cfg, const uint32_t *gathered, uint32_t slab_pixels, int world,
                     uint32_t fb_width, uint32_t fb_height, uint32_t *img_rowmajor);

// Diagnostics (crt_hip_trace_rays / crt_hip_kat): same traversal code, explicit rays.
void launch_trace_diag(const LaunchCfg &cfg, const SceneView &sc, uint32_t n, const float *org, const float *dir,
                       float tmin, const float *tmax, bool closest, float *out_t, float *out_u,
                       float *out_v, int32_t *out_inst, int32_t *out_geom, int32_t *out_prim,
                       unsigned long long *counters /* nodes, tris, ray cursor */);
int launch_kat(const LaunchCfg &cfg, const SceneView &sc, int fn, uint32_t n, const float *in, int in_stride,
               float *out, int out_stride);

} // namespace crt
