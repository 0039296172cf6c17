// kernels.hip — wavefront path-tracing kernels for MI355X (gfx950, wave64).
//
// The reference's per-pixel loop (backends/embree/render_embree.ispc:198-355, one ISPC lane
// per pixel walking its whole path) is split into one kernel per stage so that a wave always
// executes ONE stage for 64 different paths:
//
//   K1 raygen         pixel-sample -> primary ray                       (ispc:213-232)
//   K2 trace_closest  BVH4 traversal + triangle test, closest hit       (rtcIntersectV, ispc:245)
//   K3 shade          hit -> material, NEE set-up, BSDF sample, RR      (ispc:251-335)
//   K4 trace_shadow   any-hit traversal of NEE rays                     (rtcOccludedV, ispc:144,170)
//   K5 accumulate     sample sum, running mean, sRGB8                   (ispc:339-353, 358-370)
//   K8 assemble       multi-GPU tile un-permute                         (new)
//
// Rays that survive a bounce are compacted into the next queue with __ballot + mbcnt, staged in
// LDS and flushed with one atomic per 128-256 entries. Traversal kernels are persistent: a fixed
// grid of waves pulls 128-ray chunks from the queue with an atomic cursor and refills lanes as
// their rays finish, so long rays do not stall a whole block.
// No MFMA anywhere: the path is divergent pointer chasing, bound by the per-CU vector-memory front end (one
// 128-byte L2 -> L1 line fill per lane and node visit), not a contraction.

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstddef>
#include <cstdlib>

#include "../../include/crt_kat.h"
#include "kernels.h"
#include "pt_device.h"
#include "traverse.h"

namespace crt {

#ifndef CRT_TRACE_BLOCK
#define CRT_TRACE_BLOCK 256
#endif
#ifndef CRT_TRACE_BLOCKS_PER_CU
#define CRT_TRACE_BLOCKS_PER_CU 7
#endif
// waves per SIMD the register allocator must leave room for in the traversal kernels (= blocks per CU
// for 256-thread blocks). 7 = 72 VGPRs: every production instantiation fits without scratch once the compiler no longer
// forms packed-fp32 instructions (build.py -fno-slp-vectorize: 65-72 VGPRs; with them the kernels needed 80 and
// spilled at 7 waves, which cost far more than the seventh wave buys). 22.6 KB of LDS per block (traverse.h).
// The instrumented (COUNTERS) instantiations spill ~110-140 B at this bound: diagnostics, not the frame.
#ifndef CRT_TRACE_MIN_WAVES
#define CRT_TRACE_MIN_WAVES 7
#endif
constexpr int TRACE_BLOCK = CRT_TRACE_BLOCK;     // threads per traversal block
constexpr int MAX_TOP_NODES = CRT_MAX_TOP_NODES; // BFS-ordered top BVH levels staged in LDS (48 B each)
#ifndef CRT_SHADE_BLOCK
#define CRT_SHADE_BLOCK 256 // threads per block of k_raygen / k_shade / k_accumulate (128 and 512 measured: profiles/r03_shade_grid_ab.txt)
#endif
constexpr int SHADE_BLOCK = CRT_SHADE_BLOCK;

// ---- wave-level helpers (wave64) -------------------------------------------------------------
CRT_DEV uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
CRT_DEV uint32_t lanes_below(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// Compacting append on an LDS counter: every lane of the wave must call this (wave-uniform control flow); lanes with pred get
// consecutive slots (undefined if !pred). No memory-side atomic: a block reserves its queue space once per step.
CRT_DEV uint32_t wave_append_lds(uint32_t *lds_counter, bool pred)
{
    const uint64_t mask = __ballot(pred);
    if (mask == 0) {
        return 0;
    }
    const uint32_t rank = lanes_below(mask);
    const int leader = __ffsll((unsigned long long)mask) - 1;
    uint32_t base = 0;
    if ((int)lane_id() == leader) {
        base = atomicAdd(lds_counter, (uint32_t)__popcll(mask));
    }
    base = __shfl(base, leader);
    return base + rank;
}

// pixel slot -> pixel. Slots are tile-major over this GPU's tiles; inside a 64x64 tile they
// follow a Morton curve so the 64 lanes of a wave cover a compact pixel block (coherent
// primary rays) for any spp.
CRT_DEV uint32_t compact_bits(uint32_t x)
{
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0f0f0f0fu;
    x = (x | (x >> 4)) & 0x00ff00ffu;
    return x;
}
CRT_DEV bool slot_to_pixel(const ViewParams &vp, const uint32_t *tile_ids, uint32_t slot, uint32_t &x,
                           uint32_t &y, uint32_t &ix, uint32_t &iy)
{
    const uint32_t tile = tile_ids[slot / TILE_PIXELS];
    const uint32_t m = slot % TILE_PIXELS;
    ix = compact_bits(m);
    iy = compact_bits(m >> 1);
    x = (tile % vp.n_tiles_x) * TILE + ix;
    y = (tile / vp.n_tiles_x) * TILE + iy;
    return x < vp.fb_width && y < vp.fb_height;
}

// ---- K1 raygen: render_embree.ispc:213-232 ------------------------------------------------------
// Each block owns one contiguous chunk of paths: it counts the chunk's on-image pixel-samples,
// reserves their queue slots with ONE atomic, then writes them in path order (edge tiles are
// clipped, render_embree.cpp:180-183, so not every slot of a 64x64 tile is a pixel).
__global__ __launch_bounds__(SHADE_BLOCK) void k_raygen(ViewParams vp, const uint32_t *tile_ids,
                                                        uint32_t slot0, uint32_t n_paths, uint32_t chunk,
                                                        PathQueue q, float4 *radiance, PassCounters *pc)
{
    __shared__ uint32_t s_wave[SHADE_BLOCK / 64];
    __shared__ uint32_t s_base;
    const uint32_t begin = blockIdx.x * chunk;
    const uint32_t end = min(begin + chunk, n_paths);
    const uint32_t wave = threadIdx.x / 64;
    uint32_t x, y, ix, iy;
    // phase 1: count
    uint32_t mine = 0;
    for (uint32_t p = begin + threadIdx.x; p < end; p += SHADE_BLOCK) {
        mine += slot_to_pixel(vp, tile_ids, slot0 + p / vp.spp, x, y, ix, iy) ? 1u : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) {
        mine += __shfl_down(mine, off);
    }
    if (lane_id() == 0) {
        s_wave[wave] = mine;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int w = 0; w < SHADE_BLOCK / 64; ++w) {
            total += s_wave[w];
        }
        s_base = total ? atomicAdd(&pc->n_queue[0].v, total) : 0u;
    }
    __syncthreads();
    uint32_t running = s_base;
    // phase 2: emit
    for (uint32_t base = begin; base < end; base += SHADE_BLOCK) {
        const uint32_t p = base + threadIdx.x;
        bool valid = p < end;
        uint32_t s = 0;
        x = y = 0;
        if (valid) {
            s = p % vp.spp;
            valid = slot_to_pixel(vp, tile_ids, slot0 + p / vp.spp, x, y, ix, iy);
            radiance[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const uint64_t mask = __ballot(valid);
        __syncthreads(); // s_wave is reused
        if (lane_id() == 0) {
            s_wave[wave] = (uint32_t)__popcll(mask);
        }
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (uint32_t w = 0; w < SHADE_BLOCK / 64; ++w) {
            before += w < wave ? s_wave[w] : 0u;
            total += s_wave[w];
        }
        const uint32_t slot = running + before + lanes_below(mask);
        running += total;
        if (valid) {
            // quirk Q1: the Embree backend keys the RNG with frame_id * spp + 1 + s
            uint32_t rng = rng_seed(x + y * vp.fb_width, vp.frame_id * vp.spp + 1u + s);
            const float px_x = (x + rng_nextf(rng)) / vp.fb_width;
            const float px_y = (y + rng_nextf(rng)) / vp.fb_height;
            const V3 dir = unit(v3(vp.dir_du[0] * px_x + vp.dir_dv[0] * px_y + vp.dir_top_left[0],
                                   vp.dir_du[1] * px_x + vp.dir_dv[1] * px_y + vp.dir_top_left[1],
                                   vp.dir_du[2] * px_x + vp.dir_dv[2] * px_y + vp.dir_top_left[2]));
            q.o[0][slot] = vp.pos[0];
            q.o[1][slot] = vp.pos[1];
            q.o[2][slot] = vp.pos[2];
            q.d[0][slot] = dir.x;
            q.d[1][slot] = dir.y;
            q.d[2][slot] = dir.z;
            q.path[slot] = p;
            q.rng[slot] = rng;
            q.tp[0][slot] = 1.f;
            q.tp[1][slot] = 1.f;
            q.tp[2][slot] = 1.f;
        }
    }
}

// ---- LDS layout shared by the traversal kernels ----------------------------------------------
// (CULL: the closest-hit kernels' stack rows carry a 16-bit entry distance per lane behind the references, traverse.h CRT_POP_CULL)
template <bool CULL> struct StackRow {
    int32_t ref[TRACE_BLOCK];
};
template <> struct StackRow<true> {
    int32_t ref[TRACE_BLOCK];
    uint16_t dist[TRACE_BLOCK];
};
template <bool TWO_LEVEL, bool INST_TRIS = false, bool CULL = false> struct TraceLds {
    PNodeHead top[(TWO_LEVEL ? CRT_MAX_TOP_NODES_TWO_LEVEL : MAX_TOP_NODES) + 1]; // 48 of a node's 64 bytes
    StackRow<CULL> stack[lds_stack_of(levels_of(TWO_LEVEL, INST_TRIS), CULL)];
    // two-level kernels: cold per-ray state of each lane (world-space ray, u / v / ids of the best hit; traverse.h)
    // world tree: the ray in the object space of the instance whose triangle the lane tested last
    float cold[lds_cold_of(levels_of(TWO_LEVEL, INST_TRIS))][TRACE_BLOCK];
};

template <typename Lds> CRT_DEV const PNodeHead *stage_top_nodes(const SceneView &sc, Lds &lds)
{
    // Cooperative copy of the BFS-ordered top levels into LDS, 16 B per lane per step: the three quarters of each
    // 64-byte record that the traversal reads.
    const uint32_t n = min(sc.n_top_nodes, (uint32_t)(sizeof(lds.top) / sizeof(PNodeHead) - 1));
    const float4 *src = reinterpret_cast<const float4 *>(sc.nodes + sc.root);
    float4 *dst = reinterpret_cast<float4 *>(lds.top);
    for (uint32_t i = threadIdx.x; i < n * 3; i += blockDim.x) {
        dst[i] = src[i + i / 3];
    }
    __syncthreads();
    return lds.top; // always the LDS array (so loads through it stay ds_read); holds min(n_top_nodes, MAX_TOP_NODES) nodes
}

template <bool CULL, typename Stack, typename Lds> CRT_DEV void init_traversal_stack(Stack &st, Lds &lds, const SceneView &sc)
{
    st.lds = (TV_LDS int32_t *)&lds.stack[0].ref[threadIdx.x];
    st.stride = TRACE_BLOCK;
    st.limit = (uint32_t)(uintptr_t)(TV_LDS int32_t *)&lds.stack[0].ref[0] + (uint32_t)(sizeof(lds.stack));
    st.dist_off = 0u;
    if constexpr (CULL) {
        st.dist_off = (uint32_t)(uintptr_t)(TV_LDS uint16_t *)&lds.stack[0].dist[threadIdx.x] - (uint32_t)(uintptr_t)st.lds;
    }
    st.cold = (TV_LDS float *)&lds.cold[0][threadIdx.x];
    st.spill = (TV_HBM int32_t *)(sc.stack_spill + (size_t)((blockIdx.x * TRACE_BLOCK + threadIdx.x) / 64) * (sc.spill_depth * 64u) +
                                  (threadIdx.x & 63));
}

// ---- K2 trace_closest ----------------------------------------------------------------------------
template <int LEVELS> struct ClosestSource { // LEVELS: SceneView::two_level of the scene the kernel is built for
    static constexpr bool CONST_TFAR = true, MULTI_RAY = false;
    PathQueue q;
    HitBuf hits;
    const LeafSlot *slots;
    const InstanceRec *instances;
    const uint32_t *material_ids;
    CRT_DEV void load(uint32_t i, V3 &o, V3 &d, float &tfar) const
    {
        o = v3(q.o[0][i], q.o[1][i], q.o[2][i]);
        d = v3(q.d[0][i], q.d[1][i], q.d[2][i]);
        tfar = RAY_TFAR;
    }
    // One 32-byte record per ray (wavefront.h HitBuf), two 16-byte stores by the retiring lane. The surface normal of
    // the reference -- normalize(hit.Ng), then normalize(transpose(world_to_object) * n), render_embree.ispc:269-270,
    // 288-290 -- is evaluated HERE, where the triangle and the instance are at hand (and still in cache), with exactly
    // the expressions k_shade used to run on the raw Ng: K3 then needs neither the instance id nor a matrix fetch, and
    // the record is 8 dwords instead of 9 scattered ones.
    CRT_DEV bool retire(uint32_t i, uint32_t &, const RayHit &h, V3 &, V3 &, float &, uint32_t &) const
    {
        hits.rec[2 * (size_t)i] = make_float4(h.t, h.u, h.v, __int_as_float(h.tri));

        if (h.tri >= 0) {
            // Embree's hit.Ng = cross(e2, e1) of the hit triangle, instance-local, unnormalised (ispc:269): triangle
            // h.tri & 1 of leaf slot h.tri >> 1, e1 = v0 - v1 and e2 = v2 - v0 formed like the traversal forms them
            const float4 *sl = reinterpret_cast<const float4 *>(slots + ((uint32_t)h.tri >> 1));
            const float4 q0 = sl[0], q1 = sl[1], q2 = sl[2], q3 = sl[3];
            const SlotVerts sv = slot_verts(q0, q1, q2);
            const uint32_t sel = (h.tri & 1) != 0 ? __float_as_uint(q3.x) >> SLOT_GEOM_BITS : 0x24u; // A = (v[0], v[1], v[2])
            const V3 va = slot_pick(sv, sel), vb = slot_pick(sv, sel >> 2), vc = slot_pick(sv, sel >> 4);
            V3 normal = unit(cross3(vc - va, va - vb));
            // the hit's instance: carried by the two-level walk, named by the slot's tag in a world tree, else the only one
            const int32_t inst = LEVELS == 1 ? h.inst : LEVELS == 2 ? (int32_t)(__float_as_uint(q3.w) >> 1) : 0;
            if (hits.inst_debug != nullptr) { // crt_hip_trace_rays(CRT_HIP_TRACE_PRODUCTION) only; NULL in a frame
                hits.inst_debug[i] = inst;
            }
            const InstanceRec &in = instances[inst];
            // normal = normalize(transpose(world_to_object) * normal), ispc:288-290. For the identity matrix the same
            // expression is evaluated on literal 1s and 0s -- bit for bit what the loaded matrix gives, signed zeros
            // and non-finite values included (no fast-math: x * 0 is not folded) -- which saves the three requests
            // for the matrix on every hit of an OBJ scene and most hits of C4.
            if (in.identity) {
                normal = unit(v3(1.f * normal.x + 0.f * normal.y + 0.f * normal.z,
                                 0.f * normal.x + 1.f * normal.y + 0.f * normal.z,
                                 0.f * normal.x + 0.f * normal.y + 1.f * normal.z));
            } else {
                const float *m = in.w2o; // 3x4: column c, row r at m[c*3 + r]
                normal = unit(v3(m[0] * normal.x + m[1] * normal.y + m[2] * normal.z,
                                 m[3] * normal.x + m[4] * normal.y + m[5] * normal.z,
                                 m[6] * normal.x + m[7] * normal.y + m[8] * normal.z));
            }
            // materials[instance->material_ids[geomID]] (ispc:292-293), MATERIAL_TEXTURED in bit 31
            const uint32_t mat = material_ids[in.mat_base + (__float_as_uint(q3.x) & SLOT_GEOM_MASK)];
            hits.rec[2 * (size_t)i + 1] = make_float4(normal.x, normal.y, normal.z, __uint_as_float(mat));
        } else if (hits.inst_debug != nullptr) {
            hits.inst_debug[i] = -1;
        }
        return false;
    }
};

template <bool TWO_LEVEL, bool COUNTERS, bool INST_TRIS = false>
__global__ __launch_bounds__(TRACE_BLOCK, CRT_TRACE_MIN_WAVES) void k_trace_closest(SceneView sc, PathQueue q, HitBuf hits,
                                                               PassCounters *pc, int bounce)
{
    if (pool_block_is_idle(pc->n_queue[bounce].v)) {
        return; // (the queue ends before this block's first chunk: traverse.h)
    }
    __shared__ TraceLds<TWO_LEVEL, INST_TRIS, stack_culls(false, TWO_LEVEL)> lds;
    const PNodeHead *top = stage_top_nodes(sc, lds);
    constexpr bool CULL = stack_culls(false, TWO_LEVEL);
    TraversalStack<lds_stack_of(levels_of(TWO_LEVEL, INST_TRIS), CULL), CULL> st;
    init_traversal_stack<CULL>(st, lds, sc);
    // primary rays start at tnear = 0, later rays at EPSILON (ispc:231, 323)
    const float tnear = bounce == 0 ? 0.f : RAY_EPS;
    uint32_t n_nodes = 0, n_tris = 0, n_slots = 0;
    const ClosestSource<levels_of(TWO_LEVEL, INST_TRIS)> src{q, hits, sc.slots, sc.instances, sc.material_ids};
    trace_wavefront<false, TWO_LEVEL, COUNTERS, ClosestSource<levels_of(TWO_LEVEL, INST_TRIS)>, INST_TRIS>(sc, top, st, pc->n_queue[bounce].v, &pc->cur_closest[bounce].v,
                                                                          tnear, src, n_nodes, n_tris, n_slots, &pc->max_ray_nodes,
                                                                          pc->worst_ray, &pc->t_start[bounce], &pc->prof_cycles[0][0]);
    if (COUNTERS) {
        atomicAdd(&pc->nodes_closest, (unsigned long long)n_nodes);
        atomicAdd(&pc->tris_closest, (unsigned long long)n_tris);
        atomicAdd(&pc->slots_closest, (unsigned long long)n_slots);
    }
}

// ---- K4 trace_shadow: the NEE occlusion rays of one bounce -------------------------------------
// One queue item per hit (ShadowQueueA). Its first ray goes to the sampled light point
// (ispc:131-153); the few hits whose BSDF sample also lands on the light carry a second ray in
// ShadowQueueB (ispc:156-179), traced by the same lane right after the first. The lane then
// evaluates  illum += tp * (cA*visA + cB*visB)  in the reference's order (ispc:117,151,175,301):
// one thread per path and stream order between kernels, so no atomics on the radiance.
struct ShadowSource {
    static constexpr bool CONST_TFAR = false, MULTI_RAY = true;
    ShadowQueueA sa;
    ShadowQueueB sb;
    float4 *radiance;
    CRT_DEV void load(uint32_t i, V3 &o, V3 &d, float &tfar) const
    {
        o = v3(sa.o[0][i], sa.o[1][i], sa.o[2][i]);
        d = v3(sa.d[0][i], sa.d[1][i], sa.d[2][i]);
        tfar = sa.tmax[i];
    }
    CRT_DEV bool retire(uint32_t i, uint32_t &stage, const RayHit &h, V3 &o, V3 &d, float &tfar, uint32_t &carry) const
    {
        const bool visible = h.tri < 0; // `shadow_ray.tfar > 0.f`, ispc:148,174
        if (stage == 0) {
            const float4 cp = sa.cp[i]; // {c, path | has_b}: one request for everything the common case needs
            const uint32_t pw = __float_as_uint(cp.w);
            if ((pw & SHADOW_HAS_B) == 0u) {
                if (visible) { // nee = cA
                    float4 L = radiance[pw];
                    L.x = L.x + cp.x;
                    L.y = L.y + cp.y;
                    L.z = L.z + cp.z;
                    radiance[pw] = L;
                }
                return false;
            }
            const int32_t b = sa.bslot[i];
            carry = visible ? 1u : 0u;
            o = v3(sb.o[0][b], sb.o[1][b], sb.o[2][b]); // same origin, BSDF-sampled direction
            d = v3(sb.d[0][b], sb.d[1][b], sb.d[2][b]);
            tfar = sb.tmax[b];
            stage = 1;
            return true;
        }
        // sample_direct_light's return value: illum = 0; [illum = cA;] [illum = illum + cB]
        const int32_t b = sa.bslot[i];
        V3 nee = v3(0.f);
        if (carry) {
            nee = v3(sb.ca[0][b], sb.ca[1][b], sb.ca[2][b]);
        }
        if (visible) {
            nee = nee + v3(sb.cb[0][b], sb.cb[1][b], sb.cb[2][b]);
        }
        const V3 add = v3(sb.tp[0][b], sb.tp[1][b], sb.tp[2][b]) * nee;
        const uint32_t p = sb.path[b];
        float4 L = radiance[p];
        L.x = L.x + add.x;
        L.y = L.y + add.y;
        L.z = L.z + add.z;
        radiance[p] = L;
        return false;
    }
};

template <bool TWO_LEVEL, bool COUNTERS, bool INST_TRIS = false>
__global__ __launch_bounds__(TRACE_BLOCK, CRT_TRACE_MIN_WAVES) void k_trace_shadow(SceneView sc, ShadowQueueA sa, ShadowQueueB sb,
                                                              float4 *radiance, PassCounters *pc, int bounce)
{
    if (pool_block_is_idle(pc->n_shadow_a[bounce].v)) {
        return; // (the queue ends before this block's first chunk: traverse.h)
    }
    __shared__ TraceLds<TWO_LEVEL, INST_TRIS> lds;
    const PNodeHead *top = stage_top_nodes(sc, lds);
    constexpr bool CULL = false;
    TraversalStack<lds_stack_of(levels_of(TWO_LEVEL, INST_TRIS), CULL), CULL> st;
    init_traversal_stack<CULL>(st, lds, sc);
    uint32_t n_nodes = 0, n_tris = 0, n_slots = 0;
    const ShadowSource src{sa, sb, radiance};
    trace_wavefront<true, TWO_LEVEL, COUNTERS, ShadowSource, INST_TRIS>(sc, top, st, pc->n_shadow_a[bounce].v, &pc->cur_shadow_a[bounce].v,
                                                                        RAY_EPS, src, n_nodes, n_tris, n_slots, nullptr, nullptr, nullptr,
                                                                        &pc->prof_cycles[1][0]);
    if (COUNTERS) {
        atomicAdd(&pc->nodes_shadow, (unsigned long long)n_nodes);
        atomicAdd(&pc->tris_shadow, (unsigned long long)n_tris);
        atomicAdd(&pc->slots_shadow, (unsigned long long)n_slots);
    }
}

// ---- sample_direct_light without its two occlusion queries (render_embree.ispc:105-181) -------------------------------
// What the reference computes around rtcOccluded: the light-sample ray (hit_p, light_dir, EPSILON, light_dist) with its
// contribution c_a if unoccluded (zero when a pdf is below EPSILON: the ray is traced and counted all the same,
// ispc:144-147), and -- when the BSDF sample lands on the light with both pdfs >= EPSILON -- the second ray
// (hit_p, w_i_b, EPSILON, light_dist_b) with c_b. k_trace_shadow resolves illum = [c_a] [+ c_b] from the visibilities
// (ShadowSource::retire). Shared by k_shade and the known-answer test CRT_KAT_NEE.
CRT_DEV void nee_setup(const SceneView &sc, const Surface &mat, V3 normal, V3 w_o, V3 v_x, V3 v_y, V3 hit_p, uint32_t &rng,
                       V3 &c_a, V3 &light_dir, float &light_dist, bool &has_b, V3 &c_b, V3 &w_i_b, float &light_dist_b)
{
    uint32_t light_id = (uint32_t)(rng_nextf(rng) * sc.n_lights);
    light_id = min(light_id, sc.n_lights - 1u);
    // every OBJ / glTF scene has exactly one light (scene.cpp:218-227, 406-414): its 20 floats then sit at a
    // wave-uniform address and are fetched once per wave by the scalar unit instead of 5 vector requests per lane
    const QuadLight light = sc.n_lights == 1u ? load_light(sc.lights) : load_light(sc.lights + 20 * (size_t)light_id);
    {
        V2 ls;
        ls.x = rng_nextf(rng);
        ls.y = rng_nextf(rng);
        const V3 light_pos = light_sample_position(light, ls);
        light_dir = light_pos - hit_p;
        light_dist = len3(light_dir);
        light_dir = unit(light_dir);
        const float l_pdf = light_pdf(light, light_pos, light_dir);
        const float b_pdf = disney_pdf(mat, normal, w_o, light_dir, v_x, v_y);
        if (l_pdf >= RAY_EPS && b_pdf >= RAY_EPS) {
            const V3 bsdf = disney_eval(mat, normal, w_o, light_dir, v_x, v_y);
            const float w = mis_power(1.f, l_pdf, 1.f, b_pdf);
            c_a = bsdf * light.emission * fabsf(dot3(light_dir, normal)) * w / l_pdf;
        }
    }
    {
        // ispc:156-179. The reference evaluates the BSDF first and tests the light quad
        // second; the quad test is the cheap and rarely-true one, so it goes first here
        // (pure functions: same value, ~1/3 of the shading ALU work saved).
        V3 light_pos;
        if (disney_sample_dir(mat, normal, w_o, v_x, v_y, rng, w_i_b) &&
            light_intersect(light, hit_p, w_i_b, light_dist_b, light_pos)) {
            const float b_pdf = disney_pdf(mat, normal, w_o, w_i_b, v_x, v_y);
            const V3 bsdf = disney_eval(mat, normal, w_o, w_i_b, v_x, v_y);
            if (!is_black(bsdf) && b_pdf >= RAY_EPS) {
                const float l_pdf = light_pdf(light, light_pos, w_i_b);
                if (l_pdf >= RAY_EPS) {
                    const float w = mis_power(1.f, b_pdf, 1.f, l_pdf);
                    c_b = bsdf * light.emission * fabsf(dot3(w_i_b, normal)) * w / b_pdf;
                    has_b = true;
                }
            }
        }
    }
}

// ---- K3 shade: render_embree.ispc:251-335 + sample_direct_light :105-181 -----------------------
// Output compaction. Survivors are appended to an LDS staging buffer (wave ballot + LDS atomic) and every step sends
// what it staged -- at most SHADE_BLOCK entries per queue -- to HBM with ONE memory-side atomic per queue and coalesced
// stores, three barriers per step. This keeps the queue counters (a single word each) far below their ~88 atomics/us
// ceiling. (Rounds 1-2 kept up to 128 entries back for a fuller flush and slid the rest down: seven barriers per step,
// 12 KB more LDS, 16 B more scratch; C4 shade 16.9 -> 16.1 ms without it, profiles/r03_shade_grid_ab.txt.)
#ifndef CRT_SHADE_GRID
#define CRT_SHADE_GRID (8 * 256 / CRT_SHADE_BLOCK) // blocks per CU in the grid-stride launch of k_shade
#endif
#ifndef CRT_SHADE_WAVES
#define CRT_SHADE_WAVES 4 // waves per SIMD the register allocator must leave room for
#endif
constexpr int STAGE_CAP = SHADE_BLOCK;
struct ShadeStage {
    uint32_t next[11][STAGE_CAP]; // PathQueue fields in declaration order
    uint32_t a[12][STAGE_CAP];    // ShadowQueueA: o, d, tmax (its seven SoA fields in declaration order), c.xyz, path | has_b, slot in ShadowQueueB
    uint32_t cnt_a[2], cnt_next[2]; // entries staged in this step; the counters alternate with the step's parity
    uint32_t cnt_b[2];              // B rays written in this step (counted for the ray statistics only)
    uint32_t cnt_e[2];              // A rays elided in this step (CRT_HIP_FLAG_ELIDE_UNUSED_SHADOW_RAYS; statistics only)
    uint32_t base, base_next;       // where the step's entries go in the global queues
};
static_assert(sizeof(PathQueue) == 11 * sizeof(void *) && sizeof(ShadowQueueA) == 9 * sizeof(void *) && offsetof(ShadowQueueA, cp) == 7 * sizeof(void *),
              "PathQueue is an array of field pointers, ShadowQueueA starts with seven");

// In-block regrouping of a step's items by material (round 6; CRT_SHADE_SORT=1, default 0 = items in queue order). k_shade runs at
// ~32 of 64 lanes per vector instruction; perfect per-wave material uniformity was priced at -1.4 ms of 12.5 on C4 (round 5,
// CRT_EXP_SHADE_ONE_MAT). Which lane evaluates which item of the block's 256 is free: per-pixel results do not depend on it and
// the output compaction re-orders the queues anyway. So before phase 1 the block sorts its items by a 6-bit key of the hit's
// material id (64: a miss; 65: beyond the queue) -- a STABLE counting sort (ranks by ballot inside a wave, a 4 x 66 histogram
// across the waves), so the order, and with it the order of the output queues, is deterministic.
#ifndef CRT_SHADE_SORT
#define CRT_SHADE_SORT 0
#endif
constexpr int SHADE_SORT_KEYS = 66;
struct ShadeSort {
    uint32_t hist[SHADE_BLOCK / 64][SHADE_SORT_KEYS]; // items of wave w with key k, then: first sorted position of those items
    uint16_t perm[SHADE_BLOCK];                       // sorted position -> item of the step
};
// every thread of the block must call this; returns the item (0 .. SHADE_BLOCK - 1) the calling thread evaluates
CRT_DEV uint32_t shade_sort_items(ShadeSort &ss, uint32_t key)
{
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    for (uint32_t k = threadIdx.x; k < (SHADE_BLOCK / 64) * SHADE_SORT_KEYS; k += SHADE_BLOCK) {
        (&ss.hist[0][0])[k] = 0u;
    }
    __syncthreads();
    // stable rank of the lane among the lanes of its wave with the same key, one round per distinct key of the wave
    uint32_t rank = 0;
    uint64_t todo = ~0ull;
    while (todo != 0ull) {
        const int leader = __ffsll((unsigned long long)todo) - 1;
        const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, leader);
        const uint64_t same = __ballot(key == k);
        if (key == k) {
            rank = lanes_below(same);
        }
        if ((int)lane == leader) {
            ss.hist[wave][k] = (uint32_t)__popcll(same);
        }
        todo &= ~same;
    }
    __syncthreads();
    // first sorted position of (key, wave): keys ascending, waves ascending inside a key. Wave 0 scans the 66 key totals.
    if (wave == 0) {
        uint32_t tot = 0, tot_hi = 0; // lane l: key l; lanes 0 and 1 also: keys 64 and 65
        for (int w = 0; w < SHADE_BLOCK / 64; ++w) {
            tot += ss.hist[w][lane];
            tot_hi += lane < 2u ? ss.hist[w][64 + lane] : 0u;
        }
        uint32_t incl = tot;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            incl += lane >= (uint32_t)d ? up : 0u;
        }
        const uint32_t all64 = __shfl(incl, 63);
        const uint32_t tot64 = __shfl(tot_hi, 0);
        uint32_t at = incl - tot, at_hi = all64 + (lane == 1u ? tot64 : 0u);
        for (int w = 0; w < SHADE_BLOCK / 64; ++w) {
            const uint32_t c = ss.hist[w][lane];
            ss.hist[w][lane] = at;
            at += c;
            if (lane < 2u) {
                const uint32_t ch = ss.hist[w][64 + lane];
                ss.hist[w][64 + lane] = at_hi;
                at_hi += ch;
            }
        }
    }
    __syncthreads();
    ss.perm[ss.hist[wave][key] + rank] = (uint16_t)threadIdx.x;
    __syncthreads();
    return ss.perm[threadIdx.x];
}

__global__ __launch_bounds__(SHADE_BLOCK, CRT_SHADE_WAVES) void k_shade(SceneView sc, PathQueue qin, HitBuf hits, PathQueue qout,
                                                       ShadowQueueA sa, ShadowQueueB sb, float4 *radiance,
                                                       PassCounters *pc, int bounce, int elide)
{
    __shared__ ShadeStage stage;
#if CRT_SHADE_SORT
    __shared__ ShadeSort sort_ws;
#endif
    // The grid is sized for the pass (the host does not know the queue's size): from the second bounce on a growing share of the
    // blocks has nothing to do -- half of them on C3's bounce 1, 99 % on any bounce 4 -- and leaves before it sets anything up.
    const uint32_t n = pc->n_queue[bounce].v;
    if (CRT_IDLE_EXIT && blockIdx.x * blockDim.x >= n) {
        return;
    }
    if (threadIdx.x == 0) {
        stage.cnt_a[0] = stage.cnt_a[1] = stage.cnt_next[0] = stage.cnt_next[1] = stage.cnt_b[0] = stage.cnt_b[1] = stage.cnt_e[0] = stage.cnt_e[1] = 0;
    }
    unorm8_init(); // (ends with the barrier that also publishes the counters)
    uint32_t parity = 0;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += stride) {
#if CRT_SHADE_SORT
        uint32_t i;
        {
            const uint32_t mine = base + threadIdx.x;
            uint32_t key = 65u;
            if (mine < n) {
                const int32_t tri0 = __float_as_int(hits.rec[2 * (size_t)mine].w);
                key = tri0 < 0 ? 64u : ((__float_as_uint(hits.rec[2 * (size_t)mine + 1].w) & ~MATERIAL_TEXTURED) & 63u);
            }
            i = base + shade_sort_items(sort_ws, key);
        }
#else
        const uint32_t i = base + threadIdx.x;
#endif
        const bool valid = i < n;
        // Phase 1 (per lane): hit -> surface, next-event estimation (sample_direct_light, ispc:105-181).
        // Its outputs are staged immediately (phase 2) so their registers are free again before the
        // BSDF is sampled for the continuation ray (phase 3).
        bool is_hit = false, has_b = false;
        V3 hit_p = v3(0.f), normal = v3(0.f), w_o = v3(0.f), tp_in = v3(0.f);
        V3 light_dir = v3(0.f), w_i_b = v3(0.f), c_a = v3(0.f), c_b = v3(0.f);
        float light_dist = 0.f, light_dist_b = 0.f;
        uint32_t path = 0, rng = 0, n_rays = 0;
        Surface mat;
        if (valid) {
            const V3 o = v3(qin.o[0][i], qin.o[1][i], qin.o[2][i]);
            const V3 d = v3(qin.d[0][i], qin.d[1][i], qin.d[2][i]);
            // The rays a path has traced so far (REPORT_RAY_STATS) ride in the top bits of its queue word and reach
            // radiance[path].w once, when the path ends -- not with a 16-byte read-modify-write per bounce.
            const uint32_t word = qin.path[i];
            path = word & PATH_ID_MASK;
            n_rays = (word >> PATH_ID_BITS) + 1u; // + the closest-hit ray (ispc:246-248)
            rng = qin.rng[i];
            tp_in = v3(qin.tp[0][i], qin.tp[1][i], qin.tp[2][i]);
            const float4 h0 = hits.rec[2 * (size_t)i]; // {t, u, v, tri}: the half every ray has
            const int32_t tri = __float_as_int(h0.w);
            if (tri < 0) {
                // ispc:258-262
                float4 L = radiance[path];
                L.w += (float)n_rays;
                const V3 m = tp_in * miss_color(d);
                L.x = L.x + m.x;
                L.y = L.y + m.y;
                L.z = L.z + m.z;
                radiance[path] = L;
            } else {
                is_hit = true;
                const float t = h0.x, bu = h0.y, bv = h0.z;
                const float4 h1 = hits.rec[2 * (size_t)i + 1]; // {normal, material}: K2 ran ispc:269-270, 288-293
#ifdef CRT_EXP_SHADE_ONE_MAT // TIMING EXPERIMENT ONLY (wrong image): every hit of a wave takes the material of the wave's first hit -- what material divergence costs
                const uint32_t mat_word = __builtin_amdgcn_readfirstlane(__float_as_uint(h1.w));
#else
                const uint32_t mat_word = __float_as_uint(h1.w);
#endif
                const uint32_t mat_id = mat_word & ~MATERIAL_TEXTURED;
                w_o = -d;
                hit_p = v3(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z); // ispc:264-267
                normal = v3(h1.x, h1.y, h1.z);
                V2 uv = v2(0.f, 0.f);
#ifdef CRT_EXP_SHADE_NO_UV // TIMING EXPERIMENT ONLY (wrong image): what the tri_uvs gather costs
                if (false) {
#else
                if (mat_word & MATERIAL_TEXTURED) { // (a material without textures never looks at uv: no record fetched)
#endif
                    // ispc:277-285. tri_uvs holds the hit triangle's three vertex UVs, gathered per BVH
                    // triangle at set_scene; all zeros for a geometry without UVs, which interpolates
                    // to the reference's uv = (0, 0)
                    const float4 *tu = reinterpret_cast<const float4 *>(sc.tri_uvs + TRI_UV_STRIDE * (size_t)tri);
                    const float4 ab = tu[0], cz = tu[1]; // two 16-byte requests (the record is padded to 32 bytes)
                    uv = (1.f - bu - bv) * v2(ab.x, ab.y) + bu * v2(ab.z, ab.w) + bv * v2(cz.x, cz.y);
                }
                unpack_material(sc, mat, sc.materials + 16 * (size_t)mat_id, uv);
                if (mat.specular_transmission == 0.f && dot3(w_o, normal) < 0.f) { // ispc:297-299
                    normal = -normal;
                }
                V3 v_x, v_y;
                ortho_basis(v_x, v_y, normal);

#ifdef CRT_EXP_SHADE_NO_NEE // TIMING EXPERIMENT ONLY (wrong image): what sample_direct_light's arithmetic costs
                light_dir = normal;
                light_dist = 1.f;
                rng_nextf(rng);
#else
                nee_setup(sc, mat, normal, w_o, v_x, v_y, hit_p, rng, c_a, light_dir, light_dist, has_b, c_b, w_i_b, light_dist_b);
#endif
                n_rays += has_b ? 2u : 1u; // the occlusion rays (ispc:145-147, 171-173)
                // `illum + path_throughput * nee` is evaluated even when nee == 0 (ispc:301): a
                // non-finite throughput (the reference's glass pdfs can be negative or overflow)
                // turns the pixel into NaN there, so it must here too. For a finite throughput the
                // term is a zero whose addition changes no bit of L (L is never -0), so only the
                // non-finite case touches the radiance here.
                const V3 poison = tp_in * 0.f;
                if (!(poison.x == 0.f && poison.y == 0.f && poison.z == 0.f)) {
                    float4 L = radiance[path];
                    L.x = L.x + poison.x;
                    L.y = L.y + poison.y;
                    L.z = L.z + poison.z;
                    radiance[path] = L;
                }
            }
        }

        // Phase 2 (wave-uniform): stage the occlusion rays.
        // B rays go straight to HBM, into the slot of ShadowQueueB that carries the item's own index in the input queue: the
        // queue is only ever reached through the slot an A record names, so it need not be dense, and no slot has to be
        // reserved. (Rounds 1-4 compacted it with one memory-side atomic per wave that had a B ray, "rare" being true of C4 --
        // one hit in 10^5 -- and false of C2, whose light is large and near: 8 % of the hits, i.e. practically every wave,
        // 58 k atomics on one word per 0.6 ms launch = the chip's limit for one address, and half of k_shade's time there.)
        // The count is kept for the ray statistics: per block, with the step's other appends.
        // C2 7.0 -> 6.0 ms with this and the counters' own lines, C3 / C4 +-0 (sessions r5s9, r5s13).
        const uint32_t slot_b = i;
        {
            const uint64_t b_mask = __ballot(has_b);
            if (b_mask != 0 && lane_id() == 0) {
                atomicAdd(&stage.cnt_b[parity], (uint32_t)__popcll(b_mask));
            }
        }
        if (has_b) {
            sb.o[0][slot_b] = hit_p.x;
            sb.o[1][slot_b] = hit_p.y;
            sb.o[2][slot_b] = hit_p.z;
            sb.d[0][slot_b] = w_i_b.x;
            sb.d[1][slot_b] = w_i_b.y;
            sb.d[2][slot_b] = w_i_b.z;
            sb.tmax[slot_b] = light_dist_b;
            sb.ca[0][slot_b] = c_a.x;
            sb.ca[1][slot_b] = c_a.y;
            sb.ca[2][slot_b] = c_a.z;
            sb.cb[0][slot_b] = c_b.x;
            sb.cb[1][slot_b] = c_b.y;
            sb.cb[2][slot_b] = c_b.z;
            sb.tp[0][slot_b] = tp_in.x;
            sb.tp[1][slot_b] = tp_in.y;
            sb.tp[2][slot_b] = tp_in.z;
            sb.path[slot_b] = path;
        }
        // The hit's light-sample ray. Its retire adds c = tp * cA to the path's radiance if nothing is in the way (ispc:148-151);
        // c is an exact zero where the reference's `light_pdf >= EPSILON && bsdf_pdf >= EPSILON` fails or the BSDF evaluates to
        // zero, and radiance + (+-0) is radiance (it is never -0: sums of a +0 start). With CRT_HIP_FLAG_ELIDE_UNUSED_SHADOW_RAYS
        // such a ray is counted (n_rays above) but not traced -- unless the hit has a second ray, whose lane resolves both.
        const V3 c = tp_in * c_a;
        const bool dead_a = elide != 0 && is_hit && !has_b && c.x == 0.f && c.y == 0.f && c.z == 0.f;
        if (elide != 0) {
            const uint64_t e_mask = __ballot(dead_a);
            if (e_mask != 0 && lane_id() == 0) {
                atomicAdd(&stage.cnt_e[parity], (uint32_t)__popcll(e_mask));
            }
        }
        const bool stage_a = is_hit && !dead_a;
        const uint32_t la = wave_append_lds(&stage.cnt_a[parity], stage_a);
        if (stage_a) {
            stage.a[0][la] = __float_as_uint(hit_p.x);
            stage.a[1][la] = __float_as_uint(hit_p.y);
            stage.a[2][la] = __float_as_uint(hit_p.z);
            stage.a[3][la] = __float_as_uint(light_dir.x);
            stage.a[4][la] = __float_as_uint(light_dir.y);
            stage.a[5][la] = __float_as_uint(light_dir.z);
            stage.a[6][la] = __float_as_uint(light_dist);
            stage.a[7][la] = __float_as_uint(c.x);
            stage.a[8][la] = __float_as_uint(c.y);
            stage.a[9][la] = __float_as_uint(c.z);
            stage.a[10][la] = path | (has_b ? SHADOW_HAS_B : 0u);
            stage.a[11][la] = slot_b;
        }

        // Phase 3 (per lane): continue the path, ispc:313-335. On the last iteration
        // (`while (bounce < MAX_PATH_DEPTH)`) the sampled direction, throughput and roulette
        // draw are never observed: skipped.
        bool alive = false;
        V3 w_i = v3(0.f), tp = tp_in;
#ifdef CRT_EXP_SHADE_NO_SAMPLE // TIMING EXPERIMENT ONLY (wrong image): what the continuation's BSDF sample costs
        if (is_hit && bounce + 1 < MAX_PATH_DEPTH) {
            w_i = normal;
            alive = (rng_next(rng) & 15u) != 0u;
        }
        if (false) {
#else
        if (is_hit && bounce + 1 < MAX_PATH_DEPTH) {
#endif
            V3 v_x, v_y;
            ortho_basis(v_x, v_y, normal); // recomputed rather than kept live across phase 2
            float pdf;
            const V3 bsdf = disney_sample(mat, normal, w_o, v_x, v_y, rng, w_i, pdf);
            alive = !(pdf == 0.f || is_black(bsdf));
            if (alive) {
                tp = tp * bsdf * fabsf(dot3(w_i, normal)) / pdf;
                if (bounce + 1 > 3 && russian_roulette(tp, rng)) { // ispc:327-335
                    alive = false;
                }
            }
        }

        if (is_hit && !alive) { // the path ends on this surface: hand in its ray count
            radiance[path].w += (float)n_rays;
        }

        // Phase 4 (wave-uniform): stage the continuation rays, flush full staging buffers.
        const uint32_t ln = wave_append_lds(&stage.cnt_next[parity], alive);
        if (alive) {
            stage.next[0][ln] = __float_as_uint(hit_p.x);
            stage.next[1][ln] = __float_as_uint(hit_p.y);
            stage.next[2][ln] = __float_as_uint(hit_p.z);
            stage.next[3][ln] = __float_as_uint(w_i.x);
            stage.next[4][ln] = __float_as_uint(w_i.y);
            stage.next[5][ln] = __float_as_uint(w_i.z);
            stage.next[6][ln] = path | (n_rays << PATH_ID_BITS);
            stage.next[7][ln] = rng;
            stage.next[8][ln] = __float_as_uint(tp.x);
            stage.next[9][ln] = __float_as_uint(tp.y);
            stage.next[10][ln] = __float_as_uint(tp.z);
        }
        __syncthreads();
        // everything staged in this step leaves now: one atomic per queue, coalesced stores, and the counters of the
        // OTHER parity (last read before the previous step's final barrier) are zeroed for the next step
        if (threadIdx.x == 0) {
            const uint32_t na = stage.cnt_a[parity], nn = stage.cnt_next[parity];
            stage.base = na ? atomicAdd(&pc->n_shadow_a[bounce].v, na) : 0u;
            stage.base_next = nn ? atomicAdd(&pc->n_queue[bounce + 1].v, nn) : 0u;
            if (stage.cnt_b[parity] != 0u) {
                atomicAdd(&pc->n_shadow_b[bounce].v, stage.cnt_b[parity]);
            }
            if (stage.cnt_e[parity] != 0u) {
                atomicAdd(&pc->n_shadow_elided[bounce].v, stage.cnt_e[parity]);
            }
            stage.cnt_a[parity ^ 1u] = 0;
            stage.cnt_next[parity ^ 1u] = 0;
            stage.cnt_b[parity ^ 1u] = 0;
            stage.cnt_e[parity ^ 1u] = 0;
        }
        __syncthreads();
        {
            const uint32_t t = threadIdx.x, na = stage.cnt_a[parity], nn = stage.cnt_next[parity];
            const uint32_t ba = stage.base, bn = stage.base_next;
            uint32_t *const *fa = reinterpret_cast<uint32_t *const *>(&sa);
            uint32_t *const *fn = reinterpret_cast<uint32_t *const *>(&qout);
            if (t < na) {
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    fa[k][ba + t] = stage.a[k][t];
                }
                sa.cp[ba + t] = make_float4(__uint_as_float(stage.a[7][t]), __uint_as_float(stage.a[8][t]), __uint_as_float(stage.a[9][t]),
                                            __uint_as_float(stage.a[10][t]));
                if ((stage.a[10][t] & SHADOW_HAS_B) != 0u) { // (one hit in 10^5)
                    sa.bslot[ba + t] = (int32_t)stage.a[11][t];
                }
            }
            if (t < nn) {
#pragma unroll
                for (int k = 0; k < 11; ++k) {
                    fn[k][bn + t] = stage.next[k][t];
                }
            }
        }
        __syncthreads(); // the buffers are free again
        parity ^= 1u;
    }
}

// ---- K5 accumulate: render_embree.ispc:339-353 + tile_to_uint8 :358-370 ------------------------
// A pixel's samples are summed in sample order (the reference's per-sample `illum = illum + ...`), so one thread sums one
// pixel -- but the spp float4 records of a pixel are contiguous, and 64 lanes each reading their own 16 x spp bytes touch 64
// different lines per load. For 2 <= spp <= ACC_TILE the block therefore streams its pixels' records through LDS: every
// thread loads consecutive records (one 1 KB run per wave and instruction), then the threads of the first pixels of the
// chunk sum their records from LDS in order. Same additions in the same order, coalesced reads: 0.49 -> ~0.2 ms on C4.
constexpr int ACC_TILE = 1024;  // float4 records per LDS chunk (16 KB + padding)
CRT_DEV uint32_t acc_pad(uint32_t e) { return e + (e >> 4); } // one spare record every 16: a pixel stride of 16 records would hit one bank group
__global__ __launch_bounds__(SHADE_BLOCK) void k_accumulate(ViewParams vp, const uint32_t *tile_ids,
                                                            uint32_t slot0, uint32_t n_slots, const float4 *radiance,
                                                            float4 *accum, uint32_t *tile_fb, uint32_t *img_rowmajor,
                                                            uint32_t *ray_counts)
{
    __shared__ float4 s_tile[ACC_TILE + ACC_TILE / 16 + 1];
    __shared__ float4 s_sum[SHADE_BLOCK];
    const uint32_t k0 = blockIdx.x * blockDim.x;
    const uint32_t k = k0 + threadIdx.x;
    const uint32_t spp = vp.spp;
    V3 illum = v3(0.f);
    float rays = 0.f;
    if (spp >= 2u && spp <= (uint32_t)ACC_TILE) {
        const uint32_t n_here = min((uint32_t)SHADE_BLOCK, n_slots - k0); // (k0 < n_slots: the grid is ceil(n_slots / block))
        const uint32_t per_chunk = (uint32_t)ACC_TILE / spp;              // whole pixels per chunk
        const float4 *src = radiance + (size_t)k0 * spp;
        for (uint32_t p0 = 0; p0 < n_here; p0 += per_chunk) {
            const uint32_t pixels = min(per_chunk, n_here - p0), records = pixels * spp;
            const float4 *chunk = src + (size_t)p0 * spp;
            for (uint32_t e = threadIdx.x; e < records; e += SHADE_BLOCK) {
                s_tile[acc_pad(e)] = chunk[e];
            }
            __syncthreads();
            if (threadIdx.x < pixels) {
                V3 sum = v3(0.f);
                float r = 0.f;
                for (uint32_t smp = 0; smp < spp; ++smp) {
                    const float4 L = s_tile[acc_pad(threadIdx.x * spp + smp)];
                    sum = sum + v3(L.x, L.y, L.z);
                    r += L.w;
                }
                s_sum[p0 + threadIdx.x] = make_float4(sum.x, sum.y, sum.z, r);
            }
            __syncthreads();
        }
        if (k < n_slots) {
            const float4 t = s_sum[threadIdx.x];
            illum = v3(t.x, t.y, t.z);
            rays = t.w;
        }
    } else if (k < n_slots) {
        for (uint32_t smp = 0; smp < spp; ++smp) {
            const float4 L = radiance[(size_t)k * spp + smp];
            illum = illum + v3(L.x, L.y, L.z);
            rays += L.w;
        }
    }
    if (k >= n_slots) {
        return;
    }
    const uint32_t slot = slot0 + k;
    uint32_t x, y, ix, iy;
    if (!slot_to_pixel(vp, tile_ids, slot, x, y, ix, iy)) {
        return;
    }
    illum = illum / (float)spp;
    const float4 a = accum[slot];
    illum = (illum + (float)vp.frame_id * v3(a.x, a.y, a.z)) / (float)(vp.frame_id + 1u);
    accum[slot] = make_float4(illum.x, illum.y, illum.z, 0.f);
    const uint32_t rgba = srgb8(illum.x) | (srgb8(illum.y) << 8) | (srgb8(illum.z) << 16) | 0xff000000u;
    tile_fb[(size_t)(slot / TILE_PIXELS) * TILE_PIXELS + iy * TILE + ix] = rgba;
    if (img_rowmajor) {
        img_rowmajor[(size_t)y * vp.fb_width + x] = rgba;
    }
    ray_counts[slot] = (uint32_t)rays;
}

// ---- K8 assemble: gathered[rank][local_tile][64*64] -> row-major image --------------------------
__global__ void k_assemble(const uint32_t *gathered, uint32_t slab_pixels, int world, uint32_t fb_width,
                           uint32_t fb_height, uint32_t *img)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= fb_width || y >= fb_height) {
        return;
    }
    const uint32_t ntx = (fb_width + TILE - 1) / TILE;
    const uint32_t tile = (y / TILE) * ntx + x / TILE;
    const uint32_t rank = tile % (uint32_t)world, local = tile / (uint32_t)world;
    img[(size_t)y * fb_width + x] =
        gathered[(size_t)rank * slab_pixels + (size_t)local * TILE_PIXELS + (y % TILE) * TILE + (x % TILE)];
}

// ---- diagnostics: explicit rays through the production traversal -------------------------------
template <bool ANY_HIT, int LEVELS> struct DiagSource {
    static constexpr bool CONST_TFAR = false, MULTI_RAY = false;
    SceneView sc;
    const float *org, *dir, *tmax;
    float *out_t, *out_u, *out_v;
    int32_t *out_inst, *out_geom, *out_prim;
    CRT_DEV void load(uint32_t i, V3 &o, V3 &d, float &tfar) const
    {
        o = v3(org[3 * i], org[3 * i + 1], org[3 * i + 2]);
        d = v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
        tfar = tmax[i];
    }
    CRT_DEV bool retire(uint32_t i, uint32_t &, const RayHit &h, V3 &, V3 &, float &, uint32_t &) const
    {
        if (ANY_HIT) {
            out_t[i] = h.tri < 0 ? 1.f : 0.f;
        } else {
            out_t[i] = h.t;
            out_u[i] = h.u;
            out_v[i] = h.v;
            const LeafSlot &sl = sc.slots[h.tri < 0 ? 0 : ((uint32_t)h.tri >> 1)];
            out_geom[i] = h.tri < 0 ? -1 : (int32_t)(sl.geom_sel & SLOT_GEOM_MASK);
            out_prim[i] = h.tri < 0 ? -1 : (int32_t)((h.tri & 1) != 0 ? sl.prim1 : sl.prim0);
            out_inst[i] = h.tri < 0 ? -1 : LEVELS == 1 ? h.inst : LEVELS == 2 ? (int32_t)(sl.tag >> 1) : 0;
        }
        return false;
    }
};

// counters: [0] nodes, [1] triangles, [2] low word = ray cursor, [3] leaf slots. tmin must be uniform over the
// batch (as it is inside a frame: 0 for primary rays, EPSILON afterwards).
template <bool ANY_HIT, bool TWO_LEVEL, bool INST_TRIS = false>
__global__ __launch_bounds__(TRACE_BLOCK) void k_trace_diag(SceneView sc, uint32_t n, const float *org,
                                                            const float *dir, float tmin, const float *tmax,
                                                            float *out_t, float *out_u, float *out_v,
                                                            int32_t *out_inst, int32_t *out_geom, int32_t *out_prim,
                                                            unsigned long long *counters)
{
    if (pool_block_is_idle(n)) {
        return; // (the queue ends before this block's first chunk: traverse.h)
    }
    __shared__ TraceLds<TWO_LEVEL, INST_TRIS, stack_culls(ANY_HIT, TWO_LEVEL)> lds;
    const PNodeHead *top = stage_top_nodes(sc, lds);
    constexpr bool CULL = stack_culls(ANY_HIT, TWO_LEVEL);
    TraversalStack<lds_stack_of(levels_of(TWO_LEVEL, INST_TRIS), CULL), CULL> st;
    init_traversal_stack<CULL>(st, lds, sc);
    uint32_t n_nodes = 0, n_tris = 0, n_slots = 0;
    const DiagSource<ANY_HIT, levels_of(TWO_LEVEL, INST_TRIS)> src{sc, org, dir, tmax, out_t, out_u, out_v, out_inst, out_geom, out_prim};
    trace_wavefront<ANY_HIT, TWO_LEVEL, true, DiagSource<ANY_HIT, levels_of(TWO_LEVEL, INST_TRIS)>, INST_TRIS>(sc, top, st, n, reinterpret_cast<uint32_t *>(&counters[2]),
                                                                              tmin, src, n_nodes, n_tris, n_slots);
    atomicAdd(&counters[0], (unsigned long long)n_nodes);
    atomicAdd(&counters[1], (unsigned long long)n_tris);
    atomicAdd(&counters[3], (unsigned long long)n_slots);
}

// ---- KATs of the device shading functions (record layouts: include/crt_kat.h) ------------------
__global__ void k_kat(SceneView sc, int fn, uint32_t n, const float *in, int in_stride, float *out, int out_stride)
{
    unorm8_init();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const float *a = in + (size_t)i * in_stride;
    float *o = out + (size_t)i * out_stride;
    auto st3 = [](float *p, V3 v) {
        p[0] = v.x;
        p[1] = v.y;
        p[2] = v.z;
    };
    auto load_surface = [](const float *p) {
        Surface m;
        m.base_color = v3(p[0], p[1], p[2]);
        m.metallic = p[3];
        m.specular = p[4];
        m.roughness = p[5];
        m.specular_tint = p[6];
        m.anisotropy = p[7];
        m.sheen = p[8];
        m.sheen_tint = p[9];
        m.clearcoat = p[10];
        m.clearcoat_gloss = p[11];
        m.ior = p[12];
        m.specular_transmission = p[13];
        return m;
    };
    switch (fn) {
    case CRT_KAT_DISNEY_EVAL: {
        const Surface m = load_surface(a);
        const V3 nn = ld3(a + 14), w_o = ld3(a + 17), w_i = ld3(a + 20), v_x = ld3(a + 23), v_y = ld3(a + 26);
        st3(o, disney_eval(m, nn, w_o, w_i, v_x, v_y));
        o[3] = disney_pdf(m, nn, w_o, w_i, v_x, v_y);
        break;
    }
    case CRT_KAT_DISNEY_SAMPLE: {
        const Surface m = load_surface(a);
        const V3 nn = ld3(a + 14), w_o = ld3(a + 17), v_x = ld3(a + 20), v_y = ld3(a + 23);
        uint32_t rng = __float_as_uint(a[26]);
        V3 w_i = v3(0.f);
        float pdf = 0.f;
        const V3 f = disney_sample(m, nn, w_o, v_x, v_y, rng, w_i, pdf);
        st3(o, f);
        st3(o + 3, w_i);
        o[6] = pdf;
        o[7] = __uint_as_float(rng);
        break;
    }
    case CRT_KAT_NEE: {
        const Surface m = load_surface(a);
        const V3 nn = ld3(a + 14), w_o = ld3(a + 17), v_x = ld3(a + 20), v_y = ld3(a + 23), hit_p = ld3(a + 26);
        uint32_t rng = __float_as_uint(a[29]);
        V3 c_a = v3(0.f), c_b = v3(0.f), light_dir = v3(0.f), w_i_b = v3(0.f);
        float light_dist = 0.f, light_dist_b = 0.f;
        bool has_b = false;
        nee_setup(sc, m, nn, w_o, v_x, v_y, hit_p, rng, c_a, light_dir, light_dist, has_b, c_b, w_i_b, light_dist_b);
        st3(o, c_a);
        st3(o + 3, light_dir);
        o[6] = light_dist;
        o[7] = has_b ? 1.f : 0.f;
        st3(o + 8, has_b ? c_b : v3(0.f));
        st3(o + 11, has_b ? w_i_b : v3(0.f));
        o[14] = has_b ? light_dist_b : 0.f;
        o[15] = __uint_as_float(rng);
        o[16] = has_b ? 2.f : 1.f;
        break;
    }
    case CRT_KAT_ROULETTE: {
        V3 tp = ld3(a);
        uint32_t rng = __float_as_uint(a[3]);
        float q = 0.f;
        const bool ended = russian_roulette(tp, rng, &q);
        o[0] = ended ? 1.f : 0.f;
        st3(o + 1, tp);
        o[4] = __uint_as_float(rng);
        o[5] = q;
        break;
    }
    case CRT_KAT_LIGHT: {
        const QuadLight l = load_light(a);
        const V3 orig = ld3(a + 20), dir = ld3(a + 23);
        const V3 p = light_sample_position(l, v2(a[26], a[27]));
        st3(o, p);
        o[3] = light_pdf(l, p, dir);
        float t = 0.f;
        V3 lp = v3(0.f);
        const bool hit = light_intersect(l, orig, dir, t, lp);
        o[4] = hit ? 1.f : 0.f;
        o[5] = hit ? t : 0.f;
        st3(o + 6, hit ? lp : v3(0.f));
        break;
    }
    case CRT_KAT_TEXTURE: {
        const TexRec &t = sc.textures[__float_as_uint(a[0])];
        const V4 c = sample_rgba(sc, t, v2(a[1], a[2]));
        o[0] = c.x;
        o[1] = c.y;
        o[2] = c.z;
        o[3] = c.w;
        const int ch = (int)__float_as_uint(a[3]);
        o[4] = ch < t.channels ? sample_channel(sc, t, v2(a[1], a[2]), ch) : 0.f;
        break;
    }
    case CRT_KAT_MISS:
        st3(o, miss_color(ld3(a)));
        break;
    case CRT_KAT_ORTHO_BASIS: {
        V3 v_x, v_y;
        ortho_basis(v_x, v_y, ld3(a));
        st3(o, v_x);
        st3(o + 3, v_y);
        break;
    }
    case CRT_KAT_SRGB8:
        o[0] = (float)srgb8(a[0]);
        break;
    case CRT_KAT_RNG: {
        uint32_t rng = rng_seed(__float_as_uint(a[0]), __float_as_uint(a[1]));
        o[0] = __uint_as_float(rng);
        for (int k = 0; k < 8; ++k) {
            uint32_t copy = rng;
            const float f = rng_nextf(copy);
            const uint32_t r = rng_next(rng);
            o[1 + 2 * k] = __uint_as_float(r);
            o[2 + 2 * k] = f;
        }
        break;
    }
    case CRT_KAT_UNPACK_MATERIAL: {
        Surface m;
        unpack_material(sc, m, sc.materials + 16 * (size_t)__float_as_uint(a[0]), v2(a[1], a[2]));
        st3(o, m.base_color);
        o[3] = m.metallic;
        o[4] = m.specular;
        o[5] = m.roughness;
        o[6] = m.specular_tint;
        o[7] = m.anisotropy;
        o[8] = m.sheen;
        o[9] = m.sheen_tint;
        o[10] = m.clearcoat;
        o[11] = m.clearcoat_gloss;
        o[12] = m.ior;
        o[13] = m.specular_transmission;
        break;
    }
    default:
        break;
    }
}

// ---- launchers ---------------------------------------------------------------------------------
uint32_t traversal_grid_threads(int n_cus) { return (uint32_t)n_cus * CRT_TRACE_BLOCKS_PER_CU * TRACE_BLOCK; }
// (the fewest LDS entries any kernel of that structure keeps: the closest-hit kernels' when their stack carries distances --
// the HBM slab is sized for the deepest path beyond it)
uint32_t traversal_lds_stack(uint32_t levels) { return (uint32_t)lds_stack_of((int)levels, stack_culls(false, levels == 1u)); }
int traversal_child_order() { return CRT_CHILD_ORDER; }

static inline int persistent_grid(const LaunchCfg &cfg, int blocks_per_cu)
{
    // the ray pool's chunk indices are 32-bit (traverse.h pool_take: wave w owns chunk w, the cursor hands out the chunks after the
    // grid's first helping): every wave of the grid must have a first chunk whose first ray index still fits 32 bits. 256 CUs x 7
    // blocks x 4 waves x 128 rays = 917 504; the bound is 2^32 / CRT_POOL_CHUNK waves.
    const long long waves = (long long)cfg.n_cus * blocks_per_cu * (TRACE_BLOCK / 64);
    if (waves * (long long)CRT_POOL_CHUNK >= (1ll << 32)) {
        std::fprintf(stderr, "[crt_hip] traversal grid of %lld waves x %d-ray chunks overflows the ray pool's 32-bit indices\n", waves, (int)CRT_POOL_CHUNK);
        std::abort();
    }
    return cfg.n_cus * blocks_per_cu;
}
static inline int capped_grid(const LaunchCfg &cfg, uint32_t n, int block)
{
    const uint32_t want = (n + block - 1) / block;
    const uint32_t cap = (uint32_t)cfg.n_cus * 8u;
    return (int)(want < 1 ? 1 : (want < cap ? want : cap));
}

void launch_raygen(const LaunchCfg &cfg, const ViewParams &vp, const uint32_t *tile_ids, uint32_t slot0,
                   uint32_t n_paths, PathQueue q, float4 *radiance, PassCounters *pc)
{
    // one contiguous chunk per block, a multiple of the block size
    const uint32_t blocks = (uint32_t)capped_grid(cfg, n_paths, SHADE_BLOCK);
    uint32_t chunk = (n_paths + blocks - 1) / blocks;
    chunk = (chunk + SHADE_BLOCK - 1) / SHADE_BLOCK * SHADE_BLOCK;
    const uint32_t grid = (n_paths + chunk - 1) / chunk;
    k_raygen<<<grid, SHADE_BLOCK, 0, cfg.stream>>>(vp, tile_ids, slot0, n_paths, chunk, q, radiance, pc);
}

// SceneView::two_level: 0 = one instance, 1 = two-level traversal, 2 = one tree in world space whose triangles carry
// their instance (LEVELS_WORLD_TREE, crt_types.h)
template <typename... Args> static void launch6(uint32_t levels, bool counters, void (*k00)(Args...), void (*k01)(Args...),
                                                void (*k10)(Args...), void (*k11)(Args...), void (*k20)(Args...),
                                                void (*k21)(Args...), int grid, hipStream_t stream, Args... args)
{
    auto k = levels == LEVELS_WORLD_TREE ? (counters ? k21 : k20) : levels != 0 ? (counters ? k11 : k10) : (counters ? k01 : k00);
    k<<<grid, TRACE_BLOCK, 0, stream>>>(args...);
}

void launch_trace_closest(const LaunchCfg &cfg, const SceneView &sc, PathQueue q, HitBuf hits, PassCounters *pc,
                          int bounce)
{

    launch6(sc.two_level, cfg.counters, k_trace_closest<false, false>, k_trace_closest<false, true>, k_trace_closest<true, false>,
            k_trace_closest<true, true>, k_trace_closest<false, false, true>, k_trace_closest<false, true, true>,
            persistent_grid(cfg, CRT_TRACE_BLOCKS_PER_CU), cfg.stream, sc, q, hits, pc, bounce);
}

void launch_trace_shadow(const LaunchCfg &cfg, const SceneView &sc, ShadowQueueA sa, ShadowQueueB sb,
                         float4 *radiance, PassCounters *pc, int bounce)
{

    launch6(sc.two_level, cfg.counters, k_trace_shadow<false, false>, k_trace_shadow<false, true>, k_trace_shadow<true, false>,
            k_trace_shadow<true, true>, k_trace_shadow<false, false, true>, k_trace_shadow<false, true, true>,
            persistent_grid(cfg, CRT_TRACE_BLOCKS_PER_CU), cfg.stream, sc, sa, sb, radiance, pc, bounce);
}

// The grid of k_shade's grid-stride loop. The queue is in pixel order, so items that are near each other in it touch the
// same materials, texture tiles and uv records; blocks are dispatched in index order, so the larger the grid, the closer
// together in the queue the ~1 000 resident blocks work at any time (and the finer the balance between cheap and
// expensive items). Measured (profiles/r03_shade_grid_ab.txt, blocks per CU): C4 shade 19.5 ms at 8, 18.6 at 16, 18.2 at
// 32, 18.0 at 128; C3 3.40 -> 3.15 ms; fewer than 8 is slower (20.4 at 4). A block should still have a few iterations to
// amortise its start and its final partial flush (C2, 1.8 M paths per pass: 8 per CU is best), hence: one block per
// CRT_SHADE_MIN_ITERS x 256 paths of the pass, between CRT_SHADE_GRID and CRT_SHADE_GRID_MAX blocks per CU (256 with 2
// iterations: C4 shade a further -0.3 ms against 128 with 4; round 4, after the kernel lost a fifth of its arithmetic: 1024 with 1
// iteration -- one block per 256 paths, no loop at all for a full-size pass -- 14.37 -> 13.66 ms on C4, C4F 13.96 -> 13.69, C2 / C3 /
// C4's eighth +-1 %, profiles/r04_shade_alu_ab.txt). (Handing the steps
// out in queue order by an atomic cursor instead: C4 shade 17.4 ms with 8 blocks per CU, but C3 +3 % and C2 +25 %: not kept. Runs of 2 .. 64
// consecutive steps per XCD (block b runs on XCD b % 8) so that neighbours meet in one L2: no difference on any workload.)
#ifndef CRT_SHADE_GRID_MAX
#define CRT_SHADE_GRID_MAX 1024
#endif
#ifndef CRT_SHADE_MIN_ITERS
#define CRT_SHADE_MIN_ITERS 1
#endif
void launch_shade(const LaunchCfg &cfg, const SceneView &sc, PathQueue qin, HitBuf hits, PathQueue qout,
                  ShadowQueueA sa, ShadowQueueB sb, float4 *radiance, PassCounters *pc, int bounce, uint32_t n_paths_max)
{
    const long want = (long)(n_paths_max / (uint32_t)(SHADE_BLOCK * CRT_SHADE_MIN_ITERS));
    const long lo = persistent_grid(cfg, CRT_SHADE_GRID), hi = persistent_grid(cfg, CRT_SHADE_GRID_MAX);
    const int grid = (int)(want < lo ? lo : want > hi ? hi : want);
    k_shade<<<grid, SHADE_BLOCK, 0, cfg.stream>>>(sc, qin, hits, qout, sa, sb, radiance, pc, bounce, cfg.elide ? 1 : 0);
}

void launch_accumulate(const LaunchCfg &cfg, const ViewParams &vp, const uint32_t *tile_ids, uint32_t slot0,
                       uint32_t n_slots, const float4 *radiance, float4 *accum, uint32_t *tile_fb,
                       uint32_t *img_rowmajor, uint32_t *ray_counts)
{
    const int grid = (int)((n_slots + SHADE_BLOCK - 1) / SHADE_BLOCK);
    k_accumulate<<<grid, SHADE_BLOCK, 0, cfg.stream>>>(vp, tile_ids, slot0, n_slots, radiance, accum, tile_fb,
                                                       img_rowmajor, ray_counts);
}

void launch_assemble(const LaunchCfg &cfg, const uint32_t *gathered, uint32_t slab_pixels, int world,
                     uint32_t fb_width, uint32_t fb_height, uint32_t *img_rowmajor)
{
    const dim3 block(64, 4);
    const dim3 grid((fb_width + 63) / 64, (fb_height + 3) / 4);
    k_assemble<<<grid, block, 0, cfg.stream>>>(gathered, slab_pixels, world, fb_width, fb_height, img_rowmajor);
}

void launch_trace_diag(const LaunchCfg &cfg, const SceneView &sc, uint32_t n, const float *org, const float *dir,
                       float tmin, const float *tmax, bool closest, float *out_t, float *out_u, float *out_v,
                       int32_t *out_inst, int32_t *out_geom, int32_t *out_prim, unsigned long long *counters)
{
    const int grid = persistent_grid(cfg, CRT_TRACE_BLOCKS_PER_CU);
    auto k = closest ? (sc.two_level == LEVELS_WORLD_TREE ? k_trace_diag<false, false, true>
                        : sc.two_level               ? k_trace_diag<false, true>
                                                     : k_trace_diag<false, false>)
                     : (sc.two_level == LEVELS_WORLD_TREE ? k_trace_diag<true, false, true>
                        : sc.two_level               ? k_trace_diag<true, true>
                                                     : k_trace_diag<true, false>);
    k<<<grid, TRACE_BLOCK, 0, cfg.stream>>>(sc, n, org, dir, tmin, tmax, out_t, out_u, out_v, out_inst, out_geom, out_prim, counters);
}

int launch_kat(const LaunchCfg &cfg, const SceneView &sc, int fn, uint32_t n, const float *in, int in_stride,
               float *out, int out_stride)
{
    if (fn < CRT_KAT_DISNEY_EVAL || fn > CRT_KAT_ROULETTE) {
        return -1;
    }
    k_kat<<<(n + 63) / 64, 64, 0, cfg.stream>>>(sc, fn, n, in, in_stride, out, out_stride);
    return 0;
}

} // namespace crt
