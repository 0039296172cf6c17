// traverse.h — BVH4 traversal + ray/triangle intersection for gfx950 (the Embree stand-in).
//
// Replaces what the reference gets from rtcIntersectV / rtcOccludedV (call sites
// backends/embree/render_embree.ispc:245, :144, :170). Semantics (DESIGN.md
// "Traversal rule"), identical to what the parity tests check against:
//   * triangle record (v0, e1 = v0 - v1, e2 = v2 - v0), Ng = cross(e2, e1)
//   * valid hit: den != 0, U >= 0, V >= 0, U + V <= |den|, |den|*tnear < T <= |den|*tfar
//   * t = T/|den|, u = U/|den|, v = V/|den|
//   * closest hit = lexicographic min of (t, inst, geom, prim) -> independent of visit order
//   * occluded = any valid hit
// Boxes are tested with a conservative slab test (exit widened by 2 ulp, NaN-ignoring
// min/max), the children of a 4-wide node visited nearest-entry first, the others pushed
// farthest first on a per-lane stack whose first LDS_STACK entries live in LDS ([depth][lane] so
// a wave's accesses are conflict-free) and the rest in an HBM slab.
#pragma once
#include "pt_device.h"
#include "slab.h"

namespace crt {

// Per-lane stack entries kept in LDS. Every entry beyond them is a 4-byte lane request to HBM, so the LDS part is as
// deep as the LDS budget of 7 blocks per CU allows (160 KB / 7 = 22.8 KB per block): 21 entries for the single-level
// kernels, 13 for the two-level ones, which also keep 9 dwords of cold ray state per lane there, 16 for a world tree's
// (6 cold dwords). History: 8 -> 12 entries: C4F -3.7 % frame time; 16 -> 19: C4F -2.2 %, C3 -2 %; the seventh wave per
// SIMD is worth more than the entries it costs (C4 58.1 -> 56.9 ms), and entries are worth more than LDS copies of the
// tree's top levels (kernels.h CRT_MAX_TOP_NODES; profiles/r04_issue_bound_ab.txt).
#ifndef CRT_LDS_STACK
#define CRT_LDS_STACK 21
#endif
#ifndef CRT_LDS_STACK_TWO_LEVEL
#define CRT_LDS_STACK_TWO_LEVEL 13
#endif
// the kernels of a world tree (INST_TRIS below) keep the world-space ray in LDS next to the stack, like the two-level
// ones: six dwords per lane, paid for with five stack entries
#ifndef CRT_LDS_STACK_WORLD_TREE
#define CRT_LDS_STACK_WORLD_TREE 16
#endif
// Round 6 experiment, OFF (CRT_POP_CULL=1 builds it): the stack of the CLOSEST-HIT kernels of single trees and world trees carries each
// entry's ENTRY DISTANCE, and an entry whose box the ray enters beyond what has meanwhile become the best hit is dropped at the pop
// without its node being fetched. The product stacks references only and finds out by fetching the node and failing its four
// children: 24 % of the line visits of C4's camera rays, 18 % of its bounce rays', 10 % on C3 and C2 (CPU pricing with the oracle's
// walker, ORC_WALK_POP_CULL). The distance is kept as the upper 16 bits of the child's sort key (a bfloat16 rounded towards zero: a
// lower bound, so nothing that could still matter is dropped) in a second LDS array, 2 bytes per entry next to the reference's 4;
// the LDS budget of 7 blocks per CU then holds 14 / 11 entries instead of 21 / 16; entries in the HBM slab have no distance and are
// never dropped. MEASURED (profiles/r06_pop_cull.txt): node visits per closest-hit ray of C4 29.9 -> 24.6, hits bit-identical, and
// the kernel SLOWER -- 23.4 -> 26.7 ms with one pop per inner step, 28.7 with up to three, 30.3 with up to six: a dropped entry
// either costs its lane the inner step the fetch would have cost (nothing gained but the request), or another LDS round trip in
// front of the step's node fetch, which the whole wave waits for -- and each of those costs more than the L2-resident visit it saves.
#ifndef CRT_POP_CULL
#define CRT_POP_CULL 0
#endif
#ifndef CRT_LDS_STACK_CULL
#define CRT_LDS_STACK_CULL 14
#endif
#ifndef CRT_LDS_STACK_WORLD_TREE_CULL
#define CRT_LDS_STACK_WORLD_TREE_CULL 11
#endif
// levels: SceneView::two_level (0 one instance, 1 two-level, 2 = LEVELS_WORLD_TREE); cull: the stack carries entry distances
constexpr bool stack_culls(bool any_hit, bool two_level) { return CRT_POP_CULL != 0 && !any_hit && !two_level; }
constexpr int lds_stack_of(int levels, bool cull = false)
{
    return levels == 1 ? CRT_LDS_STACK_TWO_LEVEL : levels == 2 ? (cull ? CRT_LDS_STACK_WORLD_TREE_CULL : CRT_LDS_STACK_WORLD_TREE) : (cull ? CRT_LDS_STACK_CULL : CRT_LDS_STACK);
}
constexpr int lds_cold_of(int levels) { return levels == 1 ? 9 : levels == 2 ? 6 : 1; } // dwords of cold per-ray state per lane in LDS
constexpr int levels_of(bool two_level, bool inst_tris) { return two_level ? 1 : inst_tris ? 2 : 0; }
// Deeper entries go to an explicit HBM slab laid out [wave][depth][lane]: coalesced across a wave
// and compact per wave, so deep traversals stay within a few pages. Its depth is a property of the
// scene (SceneView::spill_depth, sized at set_scene from the BVH's depth), not a compile-time limit.
// Not a private array: scratch-backed kernels get their wave occupancy throttled by the
// runtime's scratch ring, which cost this kernel most of its latency hiding.
constexpr int32_t STACK_SENTINEL = (int32_t)0x80000000; // marks "leave instance" (two-level)

struct RayHit {
    float t, u, v;
    int32_t tri;  // global index into SceneView::tris, -1 = miss
    int32_t inst; // instance index
};

// Pointers into LDS and HBM carry their address space in the type: through generic pointers the
// compiler merges the two arms of "LDS or HBM?" into one flat_load, which is slower than either
// ds_read or global_load and sends even LDS-resident data through the vector-memory front end.
#define TV_LDS __attribute__((address_space(3)))
#define TV_HBM __attribute__((address_space(1)))

// The stack pointer is kept as the LDS ADDRESS of the next free entry (`top`), not as a count: a push is a store at
// `top` and one add, a pop one subtract and a load -- no index -> address arithmetic in the step, which is bound by the
// vector instructions it issues. Entries of one lane are stride * 4 = 1024 bytes apart and a lane's column starts less
// than that into the array, so "the entry lies in the LDS part" is one compare of `top` against `limit`, the same
// constant for every lane (the address LDS_STACK rows into the block's stack array).
template <int LDS_STACK, bool DIST = false> struct TraversalStack {
    static constexpr bool HAS_DIST = DIST;
    TV_LDS int32_t *lds;   // this lane's column of the LDS part: entry k at lds[k * stride] (DIST: rows of [references][distances], 6 bytes per lane)
    int stride;
    TV_LDS float *cold;    // this lane's column of the cold per-ray state kept in LDS (two-level: world-space ray), same stride
    TV_HBM int32_t *spill; // this lane's column of its wave's HBM slab: entry k at spill[k * 64]
    uint32_t limit;        // LDS address of row LDS_STACK of the block's stack array
    uint32_t top;
    uint32_t dist_off;     // DIST: LDS address of an entry's distance = the address of its reference + dist_off (a constant of the lane)
    CRT_DEV uint32_t base() const { return (uint32_t)(uintptr_t)lds; }
    CRT_DEV uint32_t step() const { return (uint32_t)stride * (DIST ? 6u : 4u); }
    CRT_DEV void clear() { top = base(); }
    CRT_DEV bool empty() const { return top == base(); }
    CRT_DEV int depth() const { return (int)((top - base()) / step()); }
    CRT_DEV TV_HBM int32_t *spilled(uint32_t at) const { return spill + (size_t)((at - base()) / step() - (uint32_t)LDS_STACK) * 64u; }
    CRT_DEV void push(int32_t x)
    {
        if (top < limit) {
            *(TV_LDS int32_t *)(uintptr_t)top = x;
        } else {
            *spilled(top) = x;
        }
        top += step();
    }
    // DIST: key = the child's sort key, whose upper 16 bits are its entry distance as a bfloat16 rounded towards zero
    CRT_DEV void push(int32_t x, uint32_t key)
    {
        if (top < limit) {
            *(TV_LDS int32_t *)(uintptr_t)top = x;
            if (DIST) {
                *(TV_LDS uint16_t *)(uintptr_t)(top + dist_off) = (uint16_t)(key >> 16);
            }
        } else {
            *spilled(top) = x;
        }
        top += step();
    }
    CRT_DEV int32_t pop()
    {
        top -= step();
        return top < limit ? *(TV_LDS int32_t *)(uintptr_t)top : *spilled(top);
    }
    CRT_DEV int32_t peek() const // the top entry (not empty), left on the stack
    {
        const uint32_t at = top - step();
        return at < limit ? *(TV_LDS int32_t *)(uintptr_t)at : *spilled(at);
    }
    // DIST: the top entry and a lower bound of the distance at which the ray enters its box (0 for an entry of the HBM slab, which has none)
    CRT_DEV int32_t peek(float &entry_dist) const
    {
        const uint32_t at = top - step();
        if (at < limit) {
            entry_dist = __uint_as_float((uint32_t) * (TV_LDS uint16_t *)(uintptr_t)(at + dist_off) << 16);
            return *(TV_LDS int32_t *)(uintptr_t)at;
        }
        entry_dist = 0.f;
        return *spilled(at);
    }
    CRT_DEV void drop() { top -= step(); } // consume the entry peek() returned
};

// m: InstanceRec::w2o, the 3x4 affine part of the column-major matrix (column c, row r at m[c*3 + r]);
// rows evaluated left to right, as mat4.ih / the oracle evaluate the 4x4 product
CRT_DEV V3 xfm_point(const float *m, V3 p)
{
    return v3(m[0] * p.x + m[3] * p.y + m[6] * p.z + m[9], m[1] * p.x + m[4] * p.y + m[7] * p.z + m[10],
              m[2] * p.x + m[5] * p.y + m[8] * p.z + m[11]);
}
CRT_DEV V3 xfm_vector(const float *m, V3 v)
{
    return v3(m[0] * v.x + m[3] * v.y + m[6] * v.z, m[1] * v.x + m[4] * v.y + m[7] * v.z,
              m[2] * v.x + m[5] * v.y + m[8] * v.z);
}

CRT_DEV float box_dir(float x) { return fabsf(x) < 1e-18f ? copysignf(1e-18f, x) : x; }

// The ray / quantised-box test lives in slab.h (shared with the host-side check).
typedef uint32_t tv_u4 __attribute__((ext_vector_type(4))); // plain vector: loadable from any address space

// (a, b, c): the triangle's vertices in its index order. e1 = v0 - v1, e2 = v2 - v0 are formed here from
// the leaf slot's vertices -- the IEEE subtractions the host made for the 48-byte records of rounds 1-2, so every bit of
// t / u / v is what it was.
// tri_test_raw leaves the barycentrics undivided -- (U, V, |den|): only t's quotient is needed while walking (it is what hits are
// compared by); u = U / |den| and v = V / |den| of the BEST hit are formed once, when the ray retires -- the same IEEE divisions on
// the same operands, so the same bits, but one division per accepted hit instead of three (a division is ~10 instructions;
// CRT_DEFER_UV). tri_test divides at once (occlusion rays never look at u, v; the two-level kernels keep them in LDS).
CRT_DEV bool tri_test_raw(const V3 a, const V3 b, const V3 c, V3 O, V3 D, float tnear, float tfar, float &t, float &U_out, float &V_out, float &den_out)
{
    const V3 v0 = a, e1 = a - b, e2 = c - a;
    const V3 Ng = cross3(e2, e1);
    const V3 C = v0 - O;
    const V3 R = cross3(C, D);
    const float den = dot3(Ng, D);
    const float abs_den = fabsf(den);
    const uint32_t sgn = __float_as_uint(den) & 0x80000000u;
    const float U = __uint_as_float(__float_as_uint(dot3(R, e2)) ^ sgn);
    const float V = __uint_as_float(__float_as_uint(dot3(R, e1)) ^ sgn);
    const float T = __uint_as_float(__float_as_uint(dot3(Ng, C)) ^ sgn);
    // ONE branch for the three rejections (they are compares of values all lanes have anyway; three exec-mask nests cost a dozen
    // scalar instructions per triangle, and a step is priced by what it issues). `&` on purpose: no short-circuit control flow.
    // (merging the three rejections into ONE branch was measured: +-0 on C4 / C3 / C2, session r6s19)
    if (den == 0.f) {
        return false;
    }
    if (!(U >= 0.f && V >= 0.f && U + V <= abs_den)) {
        return false;
    }
    if (!(T > abs_den * tnear && T <= abs_den * tfar)) {
        return false;
    }
    t = T / abs_den;
    U_out = U;
    V_out = V;
    den_out = abs_den;
    return true;
}
CRT_DEV bool tri_test(const V3 a, const V3 b, const V3 c, V3 O, V3 D, float tnear, float tfar, float &t, float &u, float &v)
{
    float U, V, abs_den;
    if (!tri_test_raw(a, b, c, O, D, tnear, tfar, t, U, V, abs_den)) {
        return false;
    }
    u = U / abs_den;
    v = V / abs_den;
    return true;
}

// The four quarters of a leaf slot (crt_types.h LeafSlot) -> its vertices, and triangle B's vertex k (two selector bits).
struct SlotVerts {
    V3 p0, p1, p2, p3;
};
CRT_DEV SlotVerts slot_verts(const float4 q0, const float4 q1, const float4 q2)
{
    SlotVerts s;
    s.p0 = v3(q0.x, q0.y, q0.z);
    s.p1 = v3(q0.w, q1.x, q1.y);
    s.p2 = v3(q1.z, q1.w, q2.x);
    s.p3 = v3(q2.y, q2.z, q2.w);
    return s;
}
CRT_DEV V3 slot_pick(const SlotVerts &s, uint32_t sel)
{
    sel &= 3u;
    const V3 lo = sel == 0u ? s.p0 : s.p1, hi = sel == 2u ? s.p2 : s.p3;
    return sel < 2u ? lo : hi;
}

// ------------------------------------------------------------------------------------------
// Wavefront traversal: what the production kernels run.
//
// A persistent wave keeps 64 rays in flight. Three things keep its lanes busy on incoherent
// rays (PMC on the first version: 28 % of VALU lanes active, 71 % of wave cycles waiting):
//   * a lane whose ray is finished is refilled from the queue as soon as REFILL_MIN lanes are
//     idle (the wave owns a private pool of POOL_CHUNK consecutive ray indices, so the global
//     cursor sees one atomic per POOL_CHUNK rays);
//   * inner-node steps and leaf (triangle / instance-entry) steps run in separate phases, each
//     entered only when enough lanes want it, instead of serialising both bodies every step;
//   * results are written per ray when it retires.
// Semantics are exactly those of traverse() above (same slab and triangle arithmetic, same
// visit rule per ray), so hits are bit-identical; only the schedule differs.
// ------------------------------------------------------------------------------------------
#ifndef CRT_REFILL_MIN
#define CRT_REFILL_MIN 16
#endif
// the inner-node phase goes on while n_inner / n_active >= CRT_INNER_NUM / CRT_INNER_DEN
// (measured with the 4-wide tree: 1/3 is 4 % slower than 1/2 on C4, 2/3 is 3 % faster on C2 and
// 0.4 % on C4: an inner step is the expensive one, so it should run with most lanes on board)
// (re-measured with the all-loads-first leaf step: 1/2 is 2.3 % faster than 2/3 on the instanced C4 and equal on
// C4F; 3/4 is 2-4 % slower on both)
#ifndef CRT_INNER_NUM
#define CRT_INNER_NUM 1
#endif
#ifndef CRT_INNER_DEN
#define CRT_INNER_DEN 2
#endif
// Child visit order. 0: entered children fully sorted by entry distance. 1 (default): nearest first, the
// rest stacked in slot order -- 0-3 % more node visits (tools/traverse_sim.cpp), no sorting network and no
// slot -> reference selects: C4 -1.5 %, C4F -0.7 % frame time. The oracle's walker of the product's arrays
// takes the rule as a parameter (crt_hip_bvh_layout reports the build's), so the counters tests hold for both.
#ifndef CRT_CHILD_ORDER
#define CRT_CHILD_ORDER 1
#endif
// occlusion rays visit children nearest first too: unsorted (lowest slot first) is 9 % faster on C2
// but 14 % slower on C4, where the nearer child is much more often the occluder
#ifndef CRT_ANYHIT_SORT
#define CRT_ANYHIT_SORT 1
#endif
#ifndef CRT_DEFER_RETIRE
#define CRT_DEFER_RETIRE 1
#endif
#ifndef CRT_POOL_CHUNK
#define CRT_POOL_CHUNK 128
#endif
// How a wave gets its chunks of the queue (CRT_POOL_STATIC_FIRST, round 5). Wave w of the grid OWNS chunk w from the start -- no
// atomic -- and only the chunks beyond the grid's first helping are handed out by the cursor (which counts chunks, starting
// after those). Before, a launch BEGAN with every wave of the grid (7 168) asking one word for its first chunk at the same
// instant; same-address atomics retire at ~88 per microsecond on this chip (MI355X_MICROARCH.md "dequeue"), so the last wave
// started tracing ~80 us into the launch, and a queue smaller than the grid's first helping -- the late bounces of small
// frames, a rank's share at N = 8 -- paid for all of them all the same. C2 7.55 -> 6.85 ms, C3 9.21 -> 8.54, C4's eighth
// 9.79 -> 8.65, C4 -0.2 % (sessions r5s5, r5s6; profiles/r05_static_first_chunk_ab.txt). Chunks are dealt in the same order as
// before (consecutive chunks to consecutive waves); any dealing gives the same image.
// Tried on top of it and dropped: reading the cursor (agent scope) before asking near the end of the queue, to spare the
// launch the failing atomic every wave ENDS with -- the load costs more than the atomic (C2 6.96 -> 8.1 ms, C3 8.49 -> 9.0);
// dealing a queue smaller than the first helping out evenly (chunks of ceil(n / waves) >= 64 rays: one generation of rays
// instead of two over half the waves) -- no difference on C2, C3, C4 or C4's eighth.
#ifndef CRT_POOL_STATIC_FIRST
#define CRT_POOL_STATIC_FIRST 1
#endif
// closest-hit kernels of single trees and world trees: divide the best hit's barycentrics at retire (tri_test_raw): C4
// closest-hit 24.37 -> 23.90 ms, C2 / C3 +-0 (sessions r5s3, r5s4); 0 = three divisions per accepted hit, as in rounds 1-4
#ifndef CRT_DEFER_UV
#define CRT_DEFER_UV 1
#endif
// two-level scenes: 0 = instances are entered in the leaf phase; 1 = in the inner-node phase (see there) -- measured
// 16 % SLOWER on the instanced C4 (123.2 vs 105.9 ms): the entry's transform and frame change then sit in the hot
// loop that most iterations run, for the few lanes that need them
#ifndef CRT_ENTRY_IN_INNER
#define CRT_ENTRY_IN_INNER 0
#endif
// Postponed leaves ("speculative while-while", Aila & Laine 2009), single trees and world trees. A lane that reaches a leaf does
// not drop out of the inner-node phase: it keeps the leaf's reference (`post`), pops its next reference and walks on; the
// leaf phase tests postponed leaves. More lanes per inner step, denser leaf steps -- for node visits whose children are
// tested against a hit distance the postponed leaf has not shortened yet (closest hit), or that an occluder in the postponed
// leaf makes useless (occlusion). Results do not depend on it: closest hit = lexicographic minimum, occlusion = boolean.
//   0  off: a lane on a leaf waits for the leaf phase (rounds 1-5)
//   1  opportunistic: the leaf phase tests every postponed leaf, wherever its lane has got to meanwhile
//   2  deterministic: a postponed leaf is tested only once its lane has reached its NEXT leaf (or emptied its stack) -- the
//      per-ray visit sequence then does not depend on the other lanes of the wave
#ifndef CRT_SPECULATE_CLOSEST
#define CRT_SPECULATE_CLOSEST 0
#endif
#ifndef CRT_SPECULATE_ANYHIT
#define CRT_SPECULATE_ANYHIT 0
#endif
constexpr int32_t CUR_DONE = (int32_t)0x80000001; // not a node, not a leaf, not the sentinel
constexpr int32_t CUR_EXIT = (int32_t)0x80000003; // two level: the lane popped the sentinel and has to leave its instance
constexpr int32_t CUR_POP = (int32_t)0x80000005;  // CRT_POP_CULL: the entry the lane popped was dropped (its box lies beyond the best hit): pop again

// The wave's mask of a predicate, straight from the compare that formed it. HIP's __ballot(int) first MATERIALISES the predicate as
// 0 / 1 in a vector register and compares that with zero again: two vector instructions per ballot, and the inner step -- priced by the
// instructions it issues, ~0.2 ms of C4 each -- has two.
CRT_DEV uint64_t tv_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
CRT_DEV uint32_t tv_lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
CRT_DEV uint32_t tv_lanes_below(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Has the block that holds waves [blockIdx.x * waves per block, ...) anything to do at all? With a static first helping, a
// queue that ends before the block's first wave's own chunk never reaches it (there are no dynamic chunks then either): such
// a block leaves before it stages anything (the late bounces of every frame: most blocks of the grid).
#ifndef CRT_IDLE_EXIT
#define CRT_IDLE_EXIT 1 // (0: every block sets itself up, as before -- the A/B switch of this and of k_shade's early exit)
#endif
CRT_DEV bool pool_block_is_idle(uint32_t n)
{
    return CRT_IDLE_EXIT && CRT_POOL_STATIC_FIRST && (uint64_t)(blockIdx.x * (blockDim.x >> 6)) * (uint32_t)CRT_POOL_CHUNK >= n;
}

// Source: struct with
//   static constexpr bool CONST_TFAR;                          every ray of the source ends at RAY_TFAR (load's tfar is ignored)
//   static constexpr bool MULTI_RAY;                           retire() may hand the lane a follow-up ray (uses stage / carry)
//   void load(uint32_t i, V3 &o, V3 &d, float &tfar) const;   first ray of queue item i
//   bool retire(uint32_t i, uint32_t &stage, const RayHit &h, V3 &o, V3 &d, float &tfar, uint32_t &carry) const;
//        consume a finished ray's result (h.tri < 0: miss / unoccluded; h.inst is maintained by the two-level kernels
//        only: with one instance it is 0, in a world tree the hit's leaf slot names it). Returning true hands the
//        lane a follow-up ray of the same item (o, d, tfar, stage updated): the two NEE occlusion
//        rays of one hit are traced back to back by one lane, which costs nothing in a wave whose
//        lanes are refilled independently.
//
// INST_TRIS (with TWO_LEVEL = false): the scene has several instances but ONE tree over all of them in world space
// (scene_prepare.cpp "world tree": every instance's triangles have records of their own, boxes around the transformed
// vertices). Boxes are walked with the world-space ray and never change frame; a triangle record's last word says
// whose it is, (instance << 1) | identity, and the triangle is tested -- like the reference's Embree instance -- with
// the ray transformed into that instance's object space (same expressions as the two-level entry, so the same bits),
// which the lane keeps until a triangle of another instance comes along.
template <bool ANY_HIT, bool TWO_LEVEL, bool COUNTERS, typename Source, bool INST_TRIS = false>
CRT_DEV void trace_wavefront(const SceneView &sc, const PNodeHead *top,
                             TraversalStack<lds_stack_of(levels_of(TWO_LEVEL, INST_TRIS), stack_culls(ANY_HIT, TWO_LEVEL)), stack_culls(ANY_HIT, TWO_LEVEL)> &st, uint32_t n,
                             uint32_t *cursor, float tnear, const Source &src, uint32_t &n_nodes, uint32_t &n_tris, uint32_t &n_slots,
                             uint32_t *max_ray_nodes = nullptr, float *worst_ray = nullptr,
                             unsigned long long *t_marks = nullptr /* [start, drained, end] strided by MAX_PATH_DEPTH */,
                             unsigned long long *prof = nullptr /* PassCounters::prof_cycles[kind], iters, lanes 8 and 16 on */)
{
    // Phase clocks (refill / inner / leaf / retire): wave-uniform, scalar. Normally part of the instrumented (COUNTERS)
    // instantiations -- whose visit counters cost VGPRs the closest-hit kernels do not have (they spill: 116 B of scratch), which
    // distorts exactly what the clocks are to measure. -DCRT_PHASE_PROFILE=1 (a tools/variants.py build) puts the clocks
    // alone into the PRODUCTION instantiations, whose register allocation they leave alone.
#ifndef CRT_PHASE_PROFILE
#define CRT_PHASE_PROFILE 0
#endif
    constexpr bool PROF = COUNTERS || CRT_PHASE_PROFILE != 0;
    // (32-bit cycle sums -- a launch is far shorter than 2^32 cycles per phase -- and only the counts that differ: nine SGPRs)
    uint32_t pf_cyc[4] = {0, 0, 0, 0}; // (32-bit sums per wave: they wrap after 2^32 shader clocks, ~1.8 s of one wave's life -- far beyond a launch)
    uint32_t pf_inner_steps = 0, pf_outer = 0, pf_inner_lanes = 0, pf_leaf_lanes_sum = 0;
    uint32_t pf_t = PROF ? (uint32_t)clock64() : 0u;
    auto pf_mark = [&](int phase, uint32_t lanes) {
        if (PROF) {
            const uint32_t now = (uint32_t)clock64();
            pf_cyc[phase] += now - pf_t;
            pf_t = now;
            if (phase == 1) {
                pf_inner_steps += 1u;
                pf_inner_lanes += lanes;
            } else if (phase == 2) {
                pf_outer += 1u;
                pf_leaf_lanes_sum += lanes;
            }
        }
    };
    if (COUNTERS && t_marks != nullptr && tv_lane_id() == 0) {
        atomicMin(&t_marks[0], (unsigned long long)wall_clock64());
    }
    uint32_t ray_nodes = 0;
    // per-lane ray state
    int32_t ray = -1;
    int32_t cur = CUR_DONE;
    // World-space ray. Single level: registers (dead once the ray has started). Two level: needed again
    // only when an instance is entered or left and when the ray retires, so it lives in LDS -- six
    // VGPRs fewer is the difference between 5 and 6 resident waves per SIMD for the two-level kernels.
    V3 org_r = v3(0.f), dir_r = v3(0.f);
    auto world_org = [&]() -> V3 {
        return (TWO_LEVEL || INST_TRIS) ? v3(st.cold[0], st.cold[st.stride], st.cold[2 * st.stride]) : org_r;
    };
    auto world_dir = [&]() -> V3 {
        return (TWO_LEVEL || INST_TRIS) ? v3(st.cold[3 * st.stride], st.cold[4 * st.stride], st.cold[5 * st.stride]) : dir_r;
    };
    auto set_world = [&](V3 wo, V3 wd) {
        if (TWO_LEVEL || INST_TRIS) {
            st.cold[0] = wo.x;
            st.cold[st.stride] = wo.y;
            st.cold[2 * st.stride] = wo.z;
            st.cold[3 * st.stride] = wd.x;
            st.cold[4 * st.stride] = wd.y;
            st.cold[5 * st.stride] = wd.z;
        } else {
            org_r = wo;
            dir_r = wd;
        }
    };
    V3 o = v3(0.f), d = v3(0.f);                 // ray in the space being traversed
    SlabRay sr;                                  // that ray in the fixed-point frame of the current BVH
    sr.qa[0] = sr.qa[1] = sr.qa[2] = sr.qb[0] = sr.qb[1] = sr.qb[2] = 0.f;
    float tfar_var = 0.f;
    RayHit hit;
    hit.t = 0.f;
    hit.u = hit.v = 0.f;
    hit.tri = hit.inst = -1;
    // Two level: the barycentrics of the best hit so far are written when a hit is accepted and read when the ray
    // retires: cold state, kept in LDS next to the world-space ray (slots 6, 7). Its geomID / primID are only ever
    // needed on an EXACT tie in t between two hits of one instance -- two coincident triangles -- and are then read back
    // from the best hit's leaf slot rather than carried in two registers through every step of every ray.
    auto store_hit_cold = [&](float u, float v) {
        if (TWO_LEVEL) {
            st.cold[6 * st.stride] = u;
            st.cold[7 * st.stride] = v;
        } else {
            hit.u = u;
            hit.v = v;
        }
    };
    // (inst, geom, prim) < those of the best hit so far? The best hit's ids are not carried in registers: its leaf slot
    // names geomID / primID and, in a world tree, the instance (tag); with one instance there is nothing to compare.
    auto tie_break = [&](int32_t inst, uint32_t geom, uint32_t prim) -> bool {
        const LeafSlot &b = sc.slots[(uint32_t)hit.tri >> 1];
        const int32_t bi = TWO_LEVEL ? hit.inst : INST_TRIS ? (int32_t)(b.tag >> 1) : inst;
        if (inst != bi) {
            return inst < bi;
        }
        const uint32_t bg = b.geom_sel & SLOT_GEOM_MASK, bp = (hit.tri & 1) != 0 ? b.prim1 : b.prim0;
        return geom != bg ? geom < bg : prim < bp;
    };
    constexpr bool CULL = stack_culls(ANY_HIT, TWO_LEVEL); // the stack carries entry distances: entries beyond the best hit are dropped at the pop
    constexpr int SPEC = (TWO_LEVEL || CULL) ? 0 : ANY_HIT ? CRT_SPECULATE_ANYHIT : CRT_SPECULATE_CLOSEST;
    int32_t post = 0; // SPEC: the postponed leaf reference (negative), 0 = none
    int32_t cur_inst = TWO_LEVEL ? sc.world_inst : 0;
    bool in_blas = !TWO_LEVEL;
    static_assert(!(INST_TRIS && TWO_LEVEL), "per-triangle instances belong to the single tree in world space");
    // INST_TRIS: like the two-level kernels, the lane keeps the world-space ray in its cold LDS slots 0..5 and (o, d) is
    // the ray in the space of the triangle it tested last: xf_space = 1 for world space (every identity instance), else
    // the tag (instance << 1) of a transformed instance. (Both rays in registers cost the kernels scratch spills.)
    uint32_t xf_space = 1u;
    st.clear();
    // multi-ray items (Source::retire): which ray of the item the lane is on (bit 0) and what it carries over (bit 1) --
    // read and written only when a ray retires, so one dword, and for the sources that use it (Source::MULTI_RAY)
    // a cold LDS slot of the lane where the kernel has LDS to spare (one instance, two-level), not a register
    constexpr int ITEM_STATE_SLOT = TWO_LEVEL ? 8 : 0;
    constexpr bool ITEM_STATE_IN_LDS = Source::MULTI_RAY && !INST_TRIS;
    uint32_t item_state_r = 0;
    auto item_state = [&]() -> uint32_t { return ITEM_STATE_IN_LDS ? __float_as_uint(st.cold[ITEM_STATE_SLOT * st.stride]) : item_state_r; };
    auto set_item_state = [&](uint32_t x) {
        if (ITEM_STATE_IN_LDS) {
            st.cold[ITEM_STATE_SLOT * st.stride] = __uint_as_float(x);
        } else {
            item_state_r = x;
        }
    };
    // wave-uniform pool of ray indices
    uint32_t pool_next = 0, pool_end = 0;
    bool exhausted = false;
    constexpr bool DEFER_UV = CRT_DEFER_UV != 0 && !ANY_HIT && !TWO_LEVEL;
    float hit_den = 1.f; // DEFER_UV: |den| of the best hit, whose hit.u / hit.v then hold U and V undivided
    const int32_t top_lo = sc.root, top_hi = sc.root + (int32_t)sc.n_top_nodes; // host: n_top_nodes <= CRT_MAX_TOP_NODES

    // Box tests use 1/d with |d| clamped to >= 1e-18 (sign kept): with an exactly zero component
    // q*inf + (-inf) would be NaN for every plane of that axis and switch its culling off (one such
    // ray then walks ~10^4 nodes). The clamp moves the ray by < 1e-15 inside any scene, far less
    // than the boxes' one-quantum margin, so the test stays conservative. Triangles use the true d.
    auto set_frame = [&](const QFrame &f) {
        const V3 inv = v3(1.f / box_dir(d.x), 1.f / box_dir(d.y), 1.f / box_dir(d.z));
        sr.qa[0] = f.step[0] * inv.x;
        sr.qa[1] = f.step[1] * inv.y;
        sr.qa[2] = f.step[2] * inv.z;
        sr.qb[0] = (f.base[0] - o.x) * inv.x;
        sr.qb[1] = (f.base[1] - o.y) * inv.y;
        sr.qb[2] = (f.base[2] - o.z) * inv.z;
    };

    // start traversing the world-space ray (org, dir, tfar)
    auto begin_ray = [&]() {
        const V3 org = world_org(), dir = world_dir();
        o = org;
        d = dir;
        cur_inst = TWO_LEVEL ? sc.world_inst : 0; // triangles of the top-level tree belong to the grafted instance
        in_blas = !TWO_LEVEL;
        if (!TWO_LEVEL && !INST_TRIS) {
            const InstanceRec &in = sc.instances[0];
            if (!in.identity) {
                o = xfm_point(in.w2o, org);
                d = xfm_vector(in.w2o, dir);
            }
        }
        if (INST_TRIS) {
            xf_space = 1u;
        }
        set_frame(sc.root_frame);
        hit.t = Source::CONST_TFAR ? RAY_TFAR : tfar_var;
        hit.u = hit.v = 0.f;
        hit.tri = -1;
        hit.inst = -1;
        st.clear();
        cur = sc.root;
        if (SPEC) {
            post = 0;
        }
    };

    // Take the next reference off the stack (or finish). Popping the instance-exit sentinel only MARKS the lane
    // (CUR_EXIT): restoring the world-space ray and its frame (three reciprocals, a dozen multiplies, six LDS reads)
    // is done by the leaf phase, like entering an instance, so that the inner-node loop -- which most iterations
    // run -- carries no code for the few lanes that change level.
    auto pop_next = [&]() {
        if (st.empty()) {
            cur = CUR_DONE;
            return;
        }
        if (CULL) {
            float entry_dist;
            cur = st.peek(entry_dist);
            st.drop();
            if (entry_dist > hit.t) {
                cur = CUR_POP; // dropped unfetched; the lane pops again at its next inner step
            }
            return;
        }
        cur = st.pop();
        if (TWO_LEVEL && cur == STACK_SENTINEL) {
            cur = CUR_EXIT;
        }
    };

    // BRANCH-FREE POP (round 6, CRT_FAST_POP; single trees and world trees). pop_next() above is a nest of branches -- empty? LDS part
    // or HBM slab? -- executed whenever any lane of the wave pops. If no popping lane's top entry lies in the HBM slab (nearly always),
    // every lane simply READS the word below its top -- for an empty stack that is the word below its column, whatever it holds --
    // and selects: the entry and top - step, or CUR_DONE and top. One LDS read, two selects, no branch.
#ifndef CRT_FAST_POP
#define CRT_FAST_POP 0
#endif
    constexpr bool FAST_POP = CRT_FAST_POP != 0 && !TWO_LEVEL && !CULL;
    auto pop_next_fast = [&]() {
        if (FAST_POP && tv_ballot(!(st.top - st.step() < st.limit)) == 0ull) { // (an empty stack passes: its top is its base, below the limit)
            const bool has = !st.empty();
            const uint32_t below = st.top - (has ? st.step() : 0u); // (an empty stack reads its own first word: never an address outside the LDS allocation)
            const int32_t entry = *(TV_LDS int32_t *)(uintptr_t)below;
            cur = has ? entry : CUR_DONE;
            st.top = below;
        } else {
            pop_next();
        }
    };

    // up to `want` ray indices [pool_next, pool_next + take) of the wave's pool, which is topped up from the queue cursor
    // (one atomic per CRT_POOL_CHUNK rays) when it is empty; the caller advances pool_next by what it uses
    // (guided, shrinking chunks -- round 4, a measured loss: profiles/r04_guided_chunks_ab.txt -- are gone)
    const uint32_t pool_waves = gridDim.x * (blockDim.x >> 6); // waves of the grid
    bool pool_first = CRT_POOL_STATIC_FIRST != 0;               // the wave's own chunk is still to come
    // (the wave's index in the grid, formed NOW into a scalar register: formed where it is used, it keeps threadIdx.x alive -- a vector
    // register, or a scratch slot in the kernels at their 72-VGPR limit -- through the whole main loop)
    const uint32_t wave_in_grid = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    auto pool_take = [&](uint32_t want) -> uint32_t {
        if (pool_next == pool_end && !exhausted) {
            constexpr uint32_t chunk = (uint32_t)CRT_POOL_CHUNK;
            uint32_t c; // chunk index
            if (pool_first) {
                pool_first = false;
                c = wave_in_grid;
            } else {
                uint32_t ticket = 0;
                // (the lane id is formed HERE, by an asm the optimiser cannot hoist: computed once before the main loop it is a
                // register held -- or, in the world-tree closest-hit kernel at its 72-VGPR limit, a scratch slot -- for a value
                // needed once per 128 rays)
                uint32_t lane_here;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_here));
                if (lane_here == 0) {
                    ticket = atomicAdd(cursor, 1u);
                }
                c = (CRT_POOL_STATIC_FIRST ? pool_waves : 0u) + __builtin_amdgcn_readfirstlane(ticket);
            }
            if ((uint64_t)c * chunk >= n) {
                if (COUNTERS && t_marks != nullptr && tv_lane_id() == 0) {
                    atomicMin(&t_marks[MAX_PATH_DEPTH], (unsigned long long)wall_clock64());
                }
                exhausted = true;
                pool_next = pool_end = 0;
            } else {
                pool_next = c * chunk;
                pool_end = min(pool_next + chunk, n);
            }
        }
        return min(want, pool_end - pool_next);
    };
    for (;;) {
        // ---- refill idle lanes --------------------------------------------------------------
        {
            const bool idle = ray < 0;
            const uint64_t idle_mask = tv_ballot(idle);
            const uint32_t n_idle = (uint32_t)__popcll(idle_mask);
            if (n_idle >= CRT_REFILL_MIN && !exhausted) {
                const uint32_t take = pool_take(n_idle);
                if (idle) {
                    const uint32_t rank = tv_lanes_below(idle_mask);
                    if (rank < take) {
                        ray = (int32_t)(pool_next + rank);
                        V3 wo, wd;
                        src.load((uint32_t)ray, wo, wd, tfar_var);
                        set_world(wo, wd);
                        if (Source::MULTI_RAY) {
                            set_item_state(0u);
                        }
                        begin_ray();
                    }
                }
                pool_next += take;
            }
        }
        const uint64_t active_mask = tv_ballot(ray >= 0);
        pf_mark(0, 0u);
        if (active_mask == 0) {
            if (exhausted) {
                if (COUNTERS && t_marks != nullptr && tv_lane_id() == 0) {
                    atomicMax(&t_marks[2 * MAX_PATH_DEPTH], (unsigned long long)wall_clock64());
                }
                break;
            }
            continue;
        }
        // lanes with a ray still being traversed (a finished ray may wait for its batch to retire)
        const uint32_t n_active = (uint32_t)__popcll(tv_ballot(ray >= 0 && (cur != CUR_DONE || (SPEC && post < 0))));

        // ---- inner-node phase: step while at least 2/3 of the active lanes are on an inner node
        for (;;) {
            // (two level, CRT_ENTRY_IN_INNER = 1, a measured loss) a lane whose next reference is a TLAS leaf enters its
            // instance in THIS phase, sharing the wave's wait with the other lanes' node fetches
            if (SPEC && ray >= 0 && cur < 0 && cur != CUR_DONE && post >= 0) { // postpone the leaf, walk on
                post = cur;
                pop_next();
            }
            if (CULL) {
                // a dropped entry must not cost its lane an inner step of its own (the ray's walk would be no shorter than with the
                // node fetched and its children failed): pop again, up to CRT_POP_CULL_ROUNDS times, while any lane of the wave has to
#ifndef CRT_POP_CULL_ROUNDS
#define CRT_POP_CULL_ROUNDS 3
#endif
#pragma unroll 1
                for (int round = 0; round < CRT_POP_CULL_ROUNDS; ++round) {
                    const bool again = ray >= 0 && cur == CUR_POP;
                    if (tv_ballot(again) == 0ull) {
                        break;
                    }
                    if (again) {
                        pop_next();
                    }
                }
            }
            const bool enter = CRT_ENTRY_IN_INNER && TWO_LEVEL && ray >= 0 && cur != CUR_DONE && !in_blas && is_instance_leaf(cur);
            // (an idle lane -- ray < 0 -- always has cur == CUR_DONE: it was retired in that state and begin_ray is the only place that
            // changes it; so "on an inner node" is cur >= 0 alone: one compare whose result IS the wave mask the phase rule counts)
            const bool inner = cur >= 0 || enter;
            // (a lane that still has to pop again belongs to this phase: it is about to reach a node, a leaf or the end of its stack)
            const uint32_t n_inner = (uint32_t)__popcll(tv_ballot(inner || (CULL && ray >= 0 && cur == CUR_POP)));
            if (n_inner == 0 || CRT_INNER_DEN * n_inner < CRT_INNER_NUM * n_active) {
                break;
            }
            if (enter) {
                const uint32_t first = (~(uint32_t)cur) >> 3;
                const InstanceRec &in = sc.instances[first];
                cur_inst = (int32_t)first;
                if (!in.identity) {
                    o = xfm_point(in.w2o, world_org());
                    d = xfm_vector(in.w2o, world_dir());
                }
                set_frame(in.frame);
                in_blas = true;
                st.push(STACK_SENTINEL);
                cur = in.blas_root;
            } else if (inner) {
                // (CRT_INNER_PEEK) the entry the lane will continue with if the ray misses all four children is read NOW, together with
                // the node: a step that ends in a pop then pays no LDS round trip of its own in front of the next step's fetch
#ifndef CRT_INNER_PEEK
#define CRT_INNER_PEEK 0
#endif
                constexpr bool PEEK = CRT_INNER_PEEK != 0 && !CULL && !SPEC;
                const bool peek_has = PEEK && !st.empty();
                const int32_t peek_ref = peek_has ? st.peek() : CUR_DONE;
                auto pop_peeked = [&]() {
                    if (peek_has) {
                        st.drop();
                    }
                    cur = TWO_LEVEL && peek_ref == STACK_SENTINEL ? CUR_EXIT : peek_ref;
                };
                // three of the record's four 16-byte quarters (crt_types.h PNode): frame + x planes, y and z planes, references
                tv_u4 k0, k1, k2;
                if ((TWO_LEVEL ? CRT_MAX_TOP_NODES_TWO_LEVEL : CRT_MAX_TOP_NODES) > 0 && cur >= top_lo && cur < top_hi) { // LDS-resident top levels: ds_read_b128
                    const TV_LDS tv_u4 *p = (const TV_LDS tv_u4 *)(top + (cur - top_lo));
                    k0 = p[0];
                    k1 = p[1];
                    k2 = p[2];
                } else {
                    const TV_HBM tv_u4 *p = (const TV_HBM tv_u4 *)(sc.nodes + cur);
                    k0 = p[0];
                    k1 = p[1];
                    k2 = p[2];
                }
                if (COUNTERS) {
                    ++n_nodes;
                    ++ray_nodes;
                }
#if defined(CRT_EXP_NODE_EXTRA_REQUEST) // timing experiment (same image): a FOURTH 16-byte request per node visit -- the record's unused quarter, all zero -- to price
                                        // one request more on the production kernels, hence one request less (profiles/r06_node32_pricing.txt)
                if (!(cur >= top_lo && cur < top_hi)) {
                    const tv_u4 k3 = ((const TV_HBM tv_u4 *)(sc.nodes + cur))[3];
                    k2.x |= k3.x;
                }
#endif
#if defined(CRT_EXP_INNER_PAD) // timing experiment: N extra VALU instructions per inner step (is the step issue-bound?)
#pragma unroll
                for (int pad_i = 0; pad_i < CRT_EXP_INNER_PAD; ++pad_i) {
                    asm volatile("v_add_u32 %0, %0, 1" : "+v"(k0.x));
                }
                k0.x -= (uint32_t)CRT_EXP_INNER_PAD;
#endif
                const SlabNode sn = slab_node(k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w, sr);
                // Visit order: children whose box the ray enters, nearest entry first. The key is the entry distance with
                // its two lowest mantissa bits replaced by the child slot (distances are >= tnear >= 0, so their bit
                // patterns order like the values; the slot makes keys distinct and breaks ties towards the lower slot);
                // all-ones for a child that is missed or unused. (Finding the nearest child by comparing the distances
                // themselves saves the four slot inserts and was measured slower: 56.7 against 55.9 ms on C4,
                // profiles/r04_issue_bound_ab.txt.)
                uint32_t child_keys[4];
                slab_packed_keys(sn, tnear, hit.t, child_keys);
                const uint32_t s0 = child_keys[0], s1 = child_keys[1], s2 = child_keys[2], s3 = child_keys[3];
                // 5-comparator sorting network
                const uint32_t a0 = min(s0, s1), a1 = max(s0, s1), a2 = min(s2, s3), a3 = max(s2, s3);
                const uint32_t b0 = min(a0, a2), b2 = max(a0, a2), b1 = min(a1, a3), b3 = max(a1, a3);
                const uint32_t c1 = min(b1, b2), c2 = max(b1, b2);
                auto ref_of = [&](uint32_t key) -> int32_t {
                    const uint32_t slot = key & 3u;
                    return (int32_t)(slot == 0u ? k2.x : slot == 1u ? k2.y : slot == 2u ? k2.z : k2.w);
                };
                if (ANY_HIT && !CRT_ANYHIT_SORT) {
                    // occlusion rays: any order finds the same answer; lowest used slot first
                    const bool h0 = s0 != 0xffffffffu, h1 = s1 != 0xffffffffu, h2 = s2 != 0xffffffffu, h3 = s3 != 0xffffffffu;
                    if (!(h0 || h1 || h2 || h3)) {
                        if (PEEK) {
                            pop_peeked();
                        } else {
                            pop_next();
                        }
                    } else {
                        const int first = h0 ? 0 : h1 ? 1 : h2 ? 2 : 3;
                        if (h3 && first < 3) {
                            st.push((int32_t)k2.w);
                        }
                        if (h2 && first < 2) {
                            st.push((int32_t)k2.z);
                        }
                        if (h1 && first < 1) {
                            st.push((int32_t)k2.y);
                        }
                        cur = (int32_t)(first == 0 ? k2.x : first == 1 ? k2.y : first == 2 ? k2.z : k2.w);
                    }
                } else if (CRT_CHILD_ORDER == 1) {
                    // nearest child first, the other entered children stacked in slot order: no sort network
                    const uint32_t nearest = min(min(s0, s1), min(s2, s3));
                    if (nearest == 0xffffffffu) {
                        if (CULL) {
                            cur = CUR_POP; // (ONE pop site per iteration: the top of the inner step, where dropped entries are retried too)
                        } else if (PEEK) {
                            pop_peeked();
                        } else {
                            pop_next_fast();
                        }
                    } else {
                        // BRANCH-FREE PUSHES (round 6, CRT_FAST_PUSH). Four conditional pushes are four nests of exec-mask branches --
                        // compare, save / restore exec, "does the entry still fit the LDS part?", store, advance: ~16 instructions each
                        // whenever ANY lane of the wave pushes at that site, about as many as the whole box test -- and a step is priced
                        // by the instructions it issues. If every lane of the wave that pushes at all has room for four more entries
                        // in its LDS part (nearly always: the HBM slab is for the deepest few paths), each candidate is simply WRITTEN
                        // at the lane's current top and the top advanced only if the child is really stacked: a store, a select and an
                        // add per site, no branch. (What is written above the top is free space: the next real push overwrites it.)
#ifndef CRT_FAST_PUSH
#define CRT_FAST_PUSH 1
#endif
                        const bool p3 = s3 != 0xffffffffu && s3 != nearest, p2 = s2 != 0xffffffffu && s2 != nearest;
                        const bool p1 = s1 != 0xffffffffu && s1 != nearest, p0 = s0 != 0xffffffffu && s0 != nearest;
                        if (CRT_FAST_PUSH && !CULL && tv_ballot(!(st.top + 3u * st.step() < st.limit)) == 0ull) {
                            *(TV_LDS int32_t *)(uintptr_t)st.top = (int32_t)k2.w;
                            st.top += p3 ? st.step() : 0u;
                            *(TV_LDS int32_t *)(uintptr_t)st.top = (int32_t)k2.z;
                            st.top += p2 ? st.step() : 0u;
                            *(TV_LDS int32_t *)(uintptr_t)st.top = (int32_t)k2.y;
                            st.top += p1 ? st.step() : 0u;
                            *(TV_LDS int32_t *)(uintptr_t)st.top = (int32_t)k2.x;
                            st.top += p0 ? st.step() : 0u;
                        } else {
                        if (p3) {
                            st.push((int32_t)k2.w, s3);
                        }
                        if (p2) {
                            st.push((int32_t)k2.z, s2);
                        }
                        if (p1) {
                            st.push((int32_t)k2.y, s1);
                        }
                        if (p0) {
                            st.push((int32_t)k2.x, s0);
                        }
                        }
                        // (keys are distinct. Three selects in a row, not a nested conditional: the compiler turned that one into two nests
                        // of exec-mask branches, fifteen instructions for what three v_cndmask do)
#if defined(CRT_CUR_SELECT_NESTED) // (the A/B switch of the remark above)
                        cur = (int32_t)(s3 == nearest ? k2.w : s2 == nearest ? k2.z : s1 == nearest ? k2.y : k2.x);
#else
                        uint32_t next_cur = k2.x;
                        next_cur = s1 == nearest ? k2.y : next_cur;
                        next_cur = s2 == nearest ? k2.z : next_cur;
                        next_cur = s3 == nearest ? k2.w : next_cur;
                        cur = (int32_t)next_cur;
#endif
                    }
                } else if (b0 == 0xffffffffu) {
                    if (CULL) {
                        cur = CUR_POP;
                    } else if (PEEK) {
                        pop_peeked();
                    } else {
                        pop_next();
                    }
                } else {
                    if (b3 != 0xffffffffu) {
                        st.push(ref_of(b3), b3);
                    }
                    if (c2 != 0xffffffffu) {
                        st.push(ref_of(c2), c2);
                    }
                    if (c1 != 0xffffffffu) {
                        st.push(ref_of(c1), c1);
                    }
                    cur = ref_of(b0);
                }
            }
            pf_mark(1, n_inner);
        }

        // ---- leaf phase: triangles, or entering an instance -----------------------------------
        const bool leaf_lane = SPEC ? (ray >= 0 && post < 0 && (SPEC == 1 || cur < 0)) : (cur < 0 && cur != CUR_DONE && !(CULL && cur == CUR_POP)); // (idle lanes: cur == CUR_DONE, see the inner phase)
        const uint32_t pf_leaf_lanes = PROF ? (uint32_t)__popcll(tv_ballot(leaf_lane)) : 0u;
        if (leaf_lane) {
            bool entered = false;
            if (TWO_LEVEL && cur == CUR_EXIT) {
                // Leave the instance. If the next reference is another instance (rays through a layer of instanced
                // shrubs go from one straight into the next), it is entered below in the same step and the TLAS's
                // frame is never needed; otherwise back to the world-space ray in that frame.
                in_blas = false;
                cur_inst = sc.world_inst;
                pop_next();
                if (!(cur != CUR_DONE && is_instance_leaf(cur))) {
                    o = world_org();
                    d = world_dir();
                    set_frame(sc.root_frame);
                }
                entered = true; // this lane's step is used up unless it enters an instance now
            }
            const uint32_t x = ~(uint32_t)(SPEC ? post : cur);
            const uint32_t first = x >> 3;
            // top level: a leaf is an instance (count field 7) or, in a scene whose static mesh was grafted into the
            // top-level tree (scene_prepare.cpp), triangles of that mesh, tested right here with the world-space ray
            if (TWO_LEVEL && !in_blas && cur != CUR_DONE && is_instance_leaf(cur)) {
                const InstanceRec &in = sc.instances[first];
                cur_inst = (int32_t)first;
                const V3 wo = world_org(), wd = world_dir();
                o = in.identity ? wo : xfm_point(in.w2o, wo);
                d = in.identity ? wd : xfm_vector(in.w2o, wd);
                set_frame(in.frame);
                in_blas = true;
                st.push(STACK_SENTINEL);
                cur = in.blas_root;
            } else if (SPEC || (!entered && cur < 0 && cur != CUR_DONE)) {
                const uint32_t count = (x & 7u) + 1u; // leaf slots (the builders make leaves of one)
                bool occluded = false;
                // ALL of the step's loads are issued before anything is tested: the slot -- one 64-byte line holding the
                // leaf's one or two triangles -- and the stack entry the lane will continue with, so a leaf step pays ONE
                // memory latency (plus one for the instance's matrix when a world tree's triangle belongs to another
                // instance than the last one tested).
                const float4 *p = reinterpret_cast<const float4 *>(sc.slots + first);
                float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
                // (SPEC: the lane's next reference was popped when the leaf was postponed; nothing to fetch from the stack)
                // (CULL: the lane's next entry is taken at the top of the next inner step, where its entry distance is compared with
                // the hit distance this very step may shrink: nothing of the stack is held in registers across the tests)
                const bool have_next = !SPEC && !CULL && !st.empty();
                int32_t next_ref;
#ifndef CRT_FAST_PEEK
#define CRT_FAST_PEEK CRT_FAST_POP
#endif
                if (CRT_FAST_PEEK && FAST_POP && !SPEC && tv_ballot(!(st.top - st.step() < st.limit)) == 0ull) { // (as pop_next_fast: one read, one select, no branch)
                    const int32_t entry = *(TV_LDS int32_t *)(uintptr_t)(st.top - st.step());
                    next_ref = have_next ? entry : CUR_DONE;
                } else {
                    next_ref = have_next ? st.peek() : CUR_DONE;
                }
                // closest-hit rays of a frame all end at RAY_TFAR (set_ray_hit, util.ih:118): a constant, not a register
                const float tfar = Source::CONST_TFAR ? RAY_TFAR : tfar_var;
                auto test_slot = [&](uint32_t slot) {
                    if (COUNTERS) {
                        ++n_slots;
                    }
                    const uint32_t geom = __float_as_uint(q3.x) & SLOT_GEOM_MASK, sel = __float_as_uint(q3.x) >> SLOT_GEOM_BITS;
                    const uint32_t prim0 = __float_as_uint(q3.y), prim1 = __float_as_uint(q3.z);
                    if (INST_TRIS) {
                        const uint32_t tag = __float_as_uint(q3.w); // LeafSlot::tag: (instance << 1) | identity
                        cur_inst = (int32_t)(tag >> 1);
                        const uint32_t space = (tag & 1u) != 0u ? 1u : tag;
                        if (space != xf_space) {
                            const V3 wo = world_org(), wd = world_dir();
                            if (space == 1u) {
                                o = wo;
                                d = wd;
                            } else { // the two-level entry's expressions: same bits as entering the instance
                                const InstanceRec &in = sc.instances[tag >> 1];
                                o = xfm_point(in.w2o, wo);
                                d = xfm_vector(in.w2o, wd);
                            }
                            xf_space = space;
                        }
                    }
                    const SlotVerts sv = slot_verts(q0, q1, q2);
                    auto accept = [&](float t, float u, float v, uint32_t prim, uint32_t which, float den = 1.f) {
                        bool take = t < hit.t;
                        if (t == hit.t && hit.tri >= 0) { // exact tie with the best hit so far: (inst, geom, prim) decides
                            take = tie_break(cur_inst, geom, prim);
                        } else if (t == hit.t) {
                            take = true; // first hit exactly at tfar
                        }
                        if (take) {
                            hit.t = t;
                            hit.tri = (int32_t)(2u * slot + which);
                            if (TWO_LEVEL) { // (one instance: 0; world tree: the slot's tag names it -- see tie_break)
                                hit.inst = cur_inst;
                            }
                            store_hit_cold(u, v);
                            if (DEFER_UV) {
                                hit_den = den;
                            }
                        }
                    };
                    float t, u, v;
                    if (COUNTERS) {
                        ++n_tris;
                    }
                    float den_abs = 1.f;
                    if (DEFER_UV ? tri_test_raw(sv.p0, sv.p1, sv.p2, o, d, tnear, tfar, t, u, v, den_abs) : tri_test(sv.p0, sv.p1, sv.p2, o, d, tnear, tfar, t, u, v)) {
                        if (ANY_HIT) {
                            occluded = true;
                            return;
                        }
                        accept(t, u, v, prim0, 0u, den_abs);
                    }
                    if (prim1 != SLOT_NO_SECOND) {
                        if (COUNTERS) {
                            ++n_tris;
                        }
                        if (DEFER_UV ? tri_test_raw(slot_pick(sv, sel), slot_pick(sv, sel >> 2), slot_pick(sv, sel >> 4), o, d, tnear, tfar, t, u, v, den_abs)
                                     : tri_test(slot_pick(sv, sel), slot_pick(sv, sel >> 2), slot_pick(sv, sel >> 4), o, d, tnear, tfar, t, u, v)) {
                            if (ANY_HIT) {
                                occluded = true;
                                return;
                            }
                            accept(t, u, v, prim1, 1u, den_abs);
                        }
                    }
                };
                test_slot(first);
                for (uint32_t k = first + 1u; k < first + count && !(ANY_HIT && occluded); ++k) { // leaves of several slots (CRT_BVH_MAX_LEAF)
                    const float4 *pk = reinterpret_cast<const float4 *>(sc.slots + k);
                    q0 = pk[0];
                    q1 = pk[1];
                    q2 = pk[2];
                    q3 = pk[3];
                    test_slot(k);
                }
                if (SPEC) {
                    post = 0;
                }
                if (ANY_HIT && occluded) {
                    hit.tri = 0;
                    hit.t = 0.f;
                    cur = CUR_DONE;
                } else if (SPEC) {
                    // cur is what the lane walked on to
                } else if (CULL) {
                    cur = CUR_POP;
                } else if (!have_next) {
                    cur = CUR_DONE;
                } else {
                    st.drop(); // consume the entry read above
                    cur = TWO_LEVEL && next_ref == STACK_SENTINEL ? CUR_EXIT : next_ref;
                }
            }
        }

        if (PROF) {
            pf_mark(2, pf_leaf_lanes);
        }
        // ---- retire finished rays -------------------------------------------------------------
        // Retiring costs a few dozen instructions and a chain of dependent loads whatever the number
        // of lanes that take part, and a finished lane has nothing to do before the next refill
        // anyway: so rays are retired in batches, when finished + idle lanes reach the refill
        // threshold (or nothing else is left to do in this wave), not one or two per iteration.
        bool do_retire = true;
        if (CRT_DEFER_RETIRE) {
            const uint32_t n_done = (uint32_t)__popcll(tv_ballot(ray >= 0 && cur == CUR_DONE && !(SPEC && post < 0)));
            const uint32_t n_idle = (uint32_t)__popcll(tv_ballot(ray < 0));
            const uint32_t n_wait = exhausted ? n_done : n_done + n_idle;
            do_retire = n_wait >= CRT_REFILL_MIN || n_done + n_idle == 64u;
        }
        if (do_retire && ray >= 0 && cur == CUR_DONE && !(SPEC && post < 0)) {
            if (COUNTERS && max_ray_nodes != nullptr && ray_nodes > 2000u) {
                if (atomicMax(max_ray_nodes, ray_nodes) < ray_nodes) {
                    const V3 org = world_org(), dir = world_dir();
                    worst_ray[0] = org.x;
                    worst_ray[1] = org.y;
                    worst_ray[2] = org.z;
                    worst_ray[3] = dir.x;
                    worst_ray[4] = dir.y;
                    worst_ray[5] = dir.z;
                    worst_ray[6] = hit.t;
                    worst_ray[7] = (float)ray_nodes;
                }
            }
            ray_nodes = 0;
            if (TWO_LEVEL) {
                hit.u = st.cold[6 * st.stride];
                hit.v = st.cold[7 * st.stride];
            }
            if (DEFER_UV && hit.tri >= 0) { // the best hit's barycentrics: the divisions tri_test would have made when it was found
                hit.u = hit.u / hit_den;
                hit.v = hit.v / hit_den;
            }
            V3 wo = world_org(), wd = world_dir();
            uint32_t stage = 0, carry = 0;
            if (Source::MULTI_RAY) {
                const uint32_t is = item_state();
                stage = is & 1u;
                carry = is >> 1;
            }
            const bool again = src.retire((uint32_t)ray, stage, hit, wo, wd, tfar_var, carry);
            if (Source::MULTI_RAY && again) {
                set_item_state(stage | (carry << 1));
            }
            if (again) {
                set_world(wo, wd);
                begin_ray();
            } else {
                ray = -1;
            }
        }
        if (PROF) {
            pf_mark(3, 0u);
        }
    }
    if (PROF && prof != nullptr && tv_lane_id() == 0) {
        for (int k = 0; k < 4; ++k) {
            atomicAdd(&prof[k], (unsigned long long)pf_cyc[k]);
            atomicAdd(&prof[8 + k], (unsigned long long)(k == 1 ? pf_inner_steps : pf_outer));
            atomicAdd(&prof[16 + k], (unsigned long long)(k == 1 ? pf_inner_lanes : k == 2 ? pf_leaf_lanes_sum : 0u));
        }
    }
}

} // namespace crt
