// wavefront.h — SoA ray/hit queues and per-pass counters of the wavefront path tracer.
//
// One "pass" renders all samples of a contiguous range of this GPU's pixel slots. Its state
// lives in HBM as struct-of-arrays so that lane i of a wave touches element i of each field
// array (coalesced 256-B wave accesses):
//
//   PathQueue (x2, ping-pong)  closest-hit rays of one bounce: origin, direction, path id,
//                              RNG state, throughput                       11 dwords / ray
//   HitBuf                     t, u, v, triangle index | shading normal, material: ONE 32-byte record / ray
//   ShadowQueueA               light-sample occlusion rays (one per hit)   12 dwords / ray (7 SoA + one 16-byte record + 1)
//   ShadowQueueB               BSDF-sample-hits-light occlusion rays (rare on C4, 8 % of C2's hits) 18 dwords / ray; NOT dense:
//                              a record sits at its item's index in the shading kernel's input queue (no slot reservation)
//   radiance                   float4 per path: rgb = radiance so far, w = rays traced
//
// Queue sizes are produced on the device (ballot + LDS ranks, one atomic per block and queue) and never read
// by the host between bounces: every kernel takes its element count from PassCounters.
#pragma once
#include "crt_types.h"

namespace crt {

struct PathQueue {
    float *o[3];
    float *d[3];
    uint32_t *path; // path index inside the pass (= (pixel slot - first slot) * spp + s)
    uint32_t *rng;  // LCG state (lcg_rng.ih)
    float *tp[3];   // path_throughput (render_embree.ispc:241)
};

// What K2 hands K3 about a ray: one 32-byte record, two 16-byte halves written by the lane that retires the ray
// and read back coalesced (the nine per-field dword stores of the first version left K2 with 3.3x the HBM write traffic
// of its payload: a wave retires rays in batches of scattered indices, so every field array had a half-written line
// open per wave, evicted and re-written several times before it filled up):
//   rec[2i]     = {t, u, v, bits(tri)}         tri = index into SceneView::tris, -1 on a miss (the only half a miss writes)
//   rec[2i + 1] = {n.x, n.y, n.z, bits(mat)}   n = normalize(transpose(world_to_object) * normalize(hit.Ng)), i.e.
//                 render_embree.ispc:269-270, 288-290 evaluated where the triangle and the instance are at hand;
//                 mat = instance->material_ids[hit.geomID] (ispc:292-293) with MATERIAL_TEXTURED in bit 31
struct HitBuf {
    float4 *rec;
    int32_t *inst_debug; // NULL in a frame; crt_hip_trace_rays(CRT_HIP_TRACE_PRODUCTION) asks for the instance id here
};

// First NEE shadow ray of a hit (render_embree.ispc:131-153). c = throughput * contribution,
// added to the path's radiance if the ray is unoccluded and the hit has no B ray. If the hit
// also spawned a B ray (bslot >= 0), the same lane traces it next and resolves both.
struct ShadowQueueA {
    float *o[3];
    float *d[3];
    float *tmax;
    // What retiring the ray needs, as ONE 16-byte record {c.x, c.y, c.z, bits(path | has_b << 31)}: a lane retires
    // whatever item it happens to hold, so five SoA dwords (c, path, bslot) were five divergent line visits per
    // occlusion ray in a kernel bound by exactly those; the ray itself (o, d, tmax) stays SoA --
    // a refill takes consecutive items, which coalesce.
    float4 *cp;
    int32_t *bslot; // index into ShadowQueueB; written and read only for items whose has_b bit is set
};
constexpr uint32_t SHADOW_HAS_B = 0x80000000u; // in ShadowQueueA::cp[i].w, above the path index (PATH_ID_BITS = 27)

// Second NEE shadow ray (render_embree.ispc:156-179): keeps both contributions and the
// throughput unmultiplied so that illum += tp * (cA*visA + cB*visB) is evaluated in the
// reference's order.
struct ShadowQueueB {
    float *o[3];
    float *d[3];
    float *tmax;
    float *ca[3];
    float *cb[3];
    float *tp[3];
    uint32_t *path;
    int32_t *reserved;
};

// A counter that thousands of waves add to. Atomics on one address retire at ~88 per microsecond on this chip, and the unit
// that executes them is found by the address's cache line: with all of a pass's queue sizes and cursors in ONE 128-byte line
// (rounds 1-4) the appends of k_shade (two per 256 items), the chunk fetches of a closest-hit launch and those of the occlusion
// launch running next to it all queued up behind each other. Each hot counter now has a line pair of its own
// (CRT_COUNTER_ALIGN; 4 = the old, packed layout, for A/B).
#ifndef CRT_COUNTER_ALIGN
#define CRT_COUNTER_ALIGN 256
#endif
struct alignas(CRT_COUNTER_ALIGN) HotCounter {
    uint32_t v;
};

struct PassCounters {
    HotCounter n_queue[MAX_PATH_DEPTH + 1]; // closest-hit rays entering bounce b
    HotCounter n_shadow_a[MAX_PATH_DEPTH];
    HotCounter n_shadow_b[MAX_PATH_DEPTH];
    HotCounter n_shadow_elided[MAX_PATH_DEPTH]; // CRT_HIP_FLAG_ELIDE_UNUSED_SHADOW_RAYS: A rays counted but not enqueued
    // dynamic ray-fetch cursors (one per launch). They count CHUNKS of CRT_POOL_CHUNK rays, not rays, and only the chunks BEYOND the
    // grid's static first helping (wave w owns chunk w, traverse.h CRT_POOL_STATIC_FIRST): a value's meaning is tied to the grid size
    // of the launch that used it. Chunk indices are 32-bit: (cursor + waves of the grid) * CRT_POOL_CHUNK is compared with the queue
    // size in 64 bits (pool_take), and the grid itself is checked on the host (kernels.hip persistent_grid).
    HotCounter cur_closest[MAX_PATH_DEPTH];
    HotCounter cur_shadow_a[MAX_PATH_DEPTH];
    HotCounter cur_shadow_b[MAX_PATH_DEPTH];
    uint32_t max_ray_nodes; // CRT_HIP_FLAG_COUNTERS: most node fetches spent on one ray, and that ray
    unsigned long long nodes_closest, tris_closest, nodes_shadow, tris_shadow; // CRT_HIP_FLAG_COUNTERS
    unsigned long long slots_closest, slots_shadow;                            // leaf slots fetched (1-2 triangles each)
    float worst_ray[8];
    // CRT_HIP_FLAG_COUNTERS: wall-clock ticks (100 MHz) of the closest-hit launches: first wave start,
    // first wave that found the queue empty, last wave end -- how much of a launch is tail
    unsigned long long t_start[MAX_PATH_DEPTH], t_drained[MAX_PATH_DEPTH], t_end[MAX_PATH_DEPTH];
    // CRT_HIP_FLAG_COUNTERS: where the persistent traversal waves spend their time, summed over waves:
    // [phase] = refill, inner-node steps, leaf steps, retire; shader-clock cycles, loop iterations
    // that ran the phase, and lanes that took part in them. [0] closest-hit launches, [1] occlusion.
    unsigned long long prof_cycles[2][4], prof_iters[2][4], prof_lanes[2][4];
};

} // namespace crt
