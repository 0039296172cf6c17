// packet.h — wave-packet traversal for launches whose rays are coherent by construction (gfx950, wave64).
//
// north_star's "LDS-staged BVH-node packets with a per-wave traversal stack". The rays of bounce 0 -- the camera rays of
// 4 pixels x 16 spp (any spp: a wave covers a compact Morton block of one tile, kernels.hip slot_to_pixel) and their
// occlusion rays to the scene's quad light -- walk nearly the same nodes, and the per-lane kernel of traverse.h then
// sends 64 identical 16-byte requests per load instruction through the CU's vector-memory front end and runs a
// per-lane stack, refill and phase logic nobody needs. Here a wave traverses its 64 rays TOGETHER:
//   * ONE traversal stack per wave, in LDS, holding node references (wave-uniform);
//   * the node (64 B) or leaf slot (64 B) is fetched ONCE per wave by the scalar unit (s_load_dwordx16): no vector-memory
//     request at all for the tree, the scalar cache and the L2 serve it;
//   * every lane tests the node's four boxes against ITS ray (same slab arithmetic: slab.h) with ITS current hit
//     distance; a child is visited if any lane wants it, nearest-first by the first active lane's entry distances;
//   * every lane tests the slot's one or two triangles (same tri_test) and keeps its own best hit.
// The closest hit is the lexicographic minimum of (t, inst, geom, prim) over all valid hits, which
// does not depend on the order or the set of nodes visited as long as no node a ray enters before its hit is skipped --
// a lane's own box test with its own hit.t decides that here exactly as in the per-lane walk -- so hits, and therefore
// frames, are bit-identical to the per-lane kernels' (tests/test_gpu_packet.py). Occlusion rays stop a lane at its
// first hit; the wave goes on while any lane is still unoccluded.
// Scenes with a top-level tree over instances (SceneView::two_level == 1) keep the per-lane kernels: their rays change
// space lane by lane. One instance and world trees (a slot's instance is wave-uniform: its transform is applied by all
// lanes at once, when the slot's tag differs from the last one) are handled here.
#pragma once
#include "traverse.h"

namespace crt {

#ifndef CRT_PKT_STACK
#define CRT_PKT_STACK 128 // entries of a wave's stack: <= 3 pending siblings per level; a deeper tree keeps the per-lane kernels (launchers)
#endif
#ifndef CRT_PKT_CHUNK
#define CRT_PKT_CHUNK 256 // rays fetched per atomic on the queue cursor: four packets (a single word sustains ~88 returning atomics/us)
#endif

typedef uint32_t tv_u16 __attribute__((ext_vector_type(16)));
#define TV_CONST __attribute__((address_space(4)))

// 64 bytes at a wave-uniform address -> SGPRs (the compiler emits s_load_dwordx16 for a uniform constant-address-space load)
CRT_DEV tv_u16 pkt_load64(const void *base, uint32_t index_uniform)
{
    const TV_CONST tv_u16 *p = (const TV_CONST tv_u16 *)(reinterpret_cast<const char *>(base) + (size_t)index_uniform * 64u);
    return *p;
}
CRT_DEV uint32_t pkt_uniform(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

template <bool ANY_HIT, bool INST_TRIS, typename Source>
CRT_DEV void trace_packets(const SceneView &sc, TV_LDS int32_t *wstack, uint32_t n, uint32_t *cursor, float tnear, const Source &src)
{
    const uint32_t lane = tv_lane_id();
    for (;;) {
        uint32_t base = 0;
        if (lane == 0) {
            base = atomicAdd(cursor, (uint32_t)CRT_PKT_CHUNK);
        }
        base = pkt_uniform(base);
        if (base >= n) {
            break;
        }
        const uint32_t chunk_end = min(base + (uint32_t)CRT_PKT_CHUNK, n);
        for (uint32_t first = base; first < chunk_end; first += 64u) {
            const uint32_t i = first + lane;
            V3 wo = v3(0.f), wd = v3(0.f); // this lane's ray in world space
            float tfar_var = 0.f;
            uint32_t stage = 0, carry = 0;
            bool again = i < chunk_end;
            if (again) {
                src.load(i, wo, wd, tfar_var);
            }
            // one round per ray of the item: a second one only for the few occlusion items with a BSDF-sample ray
            while (__ballot(again) != 0ull) {
                const bool in_round = again;
                bool active = in_round; // still traversing (an occluded any-hit lane drops out)
                const float tfar = Source::CONST_TFAR ? RAY_TFAR : tfar_var;
                RayHit hit;
                hit.t = tfar;
                hit.u = hit.v = 0.f;
                hit.tri = -1;
                hit.inst = 0;
                // the ray the boxes are tested with: world space, or the single instance's object space
                V3 o = wo, d = wd;
                if (!INST_TRIS) {
                    const InstanceRec &in = sc.instances[0];
                    if (!in.identity) {
                        o = xfm_point(in.w2o, wo);
                        d = xfm_vector(in.w2o, wd);
                    }
                }
                SlabRay sr;
                {
                    const QFrame &f = sc.root_frame;
                    const V3 inv = v3(1.f / box_dir(d.x), 1.f / box_dir(d.y), 1.f / box_dir(d.z));
                    sr.qa[0] = f.step[0] * inv.x;
                    sr.qa[1] = f.step[1] * inv.y;
                    sr.qa[2] = f.step[2] * inv.z;
                    sr.qb[0] = (f.base[0] - o.x) * inv.x;
                    sr.qb[1] = (f.base[1] - o.y) * inv.y;
                    sr.qb[2] = (f.base[2] - o.z) * inv.z;
                }
                // world tree: the ray in the object space of the instance whose slot was tested last (wave-uniform space)
                V3 xo = o, xd = d;
                uint32_t xf_space = 1u;
                uint32_t sp = 0;
                int32_t cur = sc.root;
                for (;;) {
                    if (cur >= 0) {
                        // ---- inner node: one scalar fetch, four box tests per lane
                        const tv_u16 nd = pkt_load64(sc.nodes, (uint32_t)cur);
                        const float tmax = hit.t; // (any-hit: tfar until the lane drops out)
                        const uint32_t s0 = active ? slab_child_key(nd[0], nd[1], nd[2], 0u, sr, tnear, tmax) : 0xffffffffu;
                        const uint32_t s1 = active ? slab_child_key(nd[4], nd[5], nd[6], 1u, sr, tnear, tmax) : 0xffffffffu;
                        const uint32_t s2 = active ? slab_child_key(nd[8], nd[9], nd[10], 2u, sr, tnear, tmax) : 0xffffffffu;
                        const uint32_t s3 = active ? slab_child_key(nd[12], nd[13], nd[14], 3u, sr, tnear, tmax) : 0xffffffffu;
                        const bool any0 = __ballot(s0 != 0xffffffffu) != 0ull, any1 = __ballot(s1 != 0xffffffffu) != 0ull;
                        const bool any2 = __ballot(s2 != 0xffffffffu) != 0ull, any3 = __ballot(s3 != 0xffffffffu) != 0ull;
                        // order: the first active lane's entry distances; a child only other lanes enter sorts behind them
                        const int leader = __ffsll((unsigned long long)__ballot(active)) - 1;
                        auto wave_key = [&](uint32_t s, bool any, uint32_t slot) -> uint32_t {
                            const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)s, leader);
                            return any ? min(l, 0xfffffff0u | slot) : 0xffffffffu;
                        };
                        const uint32_t m0 = wave_key(s0, any0, 0u), m1 = wave_key(s1, any1, 1u), m2 = wave_key(s2, any2, 2u), m3 = wave_key(s3, any3, 3u);
                        const uint32_t a0 = min(m0, m1), a1 = max(m0, m1), a2 = min(m2, m3), a3 = max(m2, m3);
                        const uint32_t b0 = min(a0, a2), b2 = max(a0, a2), b1 = min(a1, a3), b3 = max(a1, a3);
                        const uint32_t c1 = min(b1, b2), c2 = max(b1, b2);
                        auto ref_of = [&](uint32_t key) -> int32_t {
                            const uint32_t slot = key & 3u;
                            return (int32_t)(slot == 0u ? nd[3] : slot == 1u ? nd[7] : slot == 2u ? nd[11] : nd[15]);
                        };
                        if (b0 != 0xffffffffu) {
                            if (lane == 0) {
                                uint32_t w = sp;
                                if (b3 != 0xffffffffu) {
                                    wstack[w++] = ref_of(b3);
                                }
                                if (c2 != 0xffffffffu) {
                                    wstack[w++] = ref_of(c2);
                                }
                                if (c1 != 0xffffffffu) {
                                    wstack[w++] = ref_of(c1);
                                }
                            }
                            sp += (b3 != 0xffffffffu ? 1u : 0u) + (c2 != 0xffffffffu ? 1u : 0u) + (c1 != 0xffffffffu ? 1u : 0u);
                            cur = ref_of(b0);
                            continue;
                        }
                    } else {
                        // ---- leaf: one scalar fetch per slot, one or two triangle tests per lane
                        const uint32_t x = ~(uint32_t)cur;
                        const uint32_t first_slot = x >> 3, count = (x & 7u) + 1u;
                        for (uint32_t k = first_slot; k < first_slot + count; ++k) {
                            const tv_u16 sl = pkt_load64(sc.slots, k);
                            const uint32_t geom = sl[12] & SLOT_GEOM_MASK, sel = sl[12] >> SLOT_GEOM_BITS;
                            const uint32_t prim0 = sl[13], prim1 = sl[14], tag = sl[15];
                            int32_t cur_inst = 0;
                            if (INST_TRIS) {
                                cur_inst = (int32_t)(tag >> 1);
                                const uint32_t space = (tag & 1u) != 0u ? 1u : tag;
                                if (space != xf_space) { // wave-uniform: every lane changes space at once
                                    if (space == 1u) {
                                        xo = wo;
                                        xd = wd;
                                    } else { // the two-level entry's expressions: same bits as entering the instance
                                        const tv_u16 m = pkt_load64(sc.instances, 2u * (tag >> 1)); // first 64 of the record's 128 bytes: w2o
                                        float w2o[12];
                                        for (int c = 0; c < 12; ++c) {
                                            w2o[c] = __uint_as_float(m[c]);
                                        }
                                        xo = xfm_point(w2o, wo);
                                        xd = xfm_vector(w2o, wd);
                                    }
                                    xf_space = space;
                                }
                            }
                            const V3 to = INST_TRIS ? xo : o, td = INST_TRIS ? xd : d;
                            auto vert = [&](uint32_t j) -> V3 { // j wave-uniform: scalar selects
                                j &= 3u;
                                return v3(__uint_as_float(j == 0u ? sl[0] : j == 1u ? sl[3] : j == 2u ? sl[6] : sl[9]),
                                          __uint_as_float(j == 0u ? sl[1] : j == 1u ? sl[4] : j == 2u ? sl[7] : sl[10]),
                                          __uint_as_float(j == 0u ? sl[2] : j == 1u ? sl[5] : j == 2u ? sl[8] : sl[11]));
                            };
                            auto accept = [&](float t, float u, float v, uint32_t prim, uint32_t which) {
                                bool take = t < hit.t;
                                if (t == hit.t && hit.tri >= 0) { // exact tie with the best hit so far: (inst, geom, prim) decides
                                    const LeafSlot &b = sc.slots[(uint32_t)hit.tri >> 1];
                                    const int32_t bi = INST_TRIS ? (int32_t)(b.tag >> 1) : 0;
                                    const uint32_t bg = b.geom_sel & SLOT_GEOM_MASK, bp = (hit.tri & 1) != 0 ? b.prim1 : b.prim0;
                                    take = cur_inst != bi ? cur_inst < bi : (geom != bg ? geom < bg : prim < bp);
                                } else if (t == hit.t) {
                                    take = true; // first hit exactly at tfar
                                }
                                if (take) {
                                    hit.t = t;
                                    hit.u = u;
                                    hit.v = v;
                                    hit.tri = (int32_t)(2u * k + which);
                                }
                            };
                            float t, u, v;
                            if (active && tri_test(vert(0u), vert(1u), vert(2u), to, td, tnear, tfar, t, u, v)) {
                                if (ANY_HIT) {
                                    active = false;
                                    hit.tri = 0;
                                } else {
                                    accept(t, u, v, prim0, 0u);
                                }
                            }
                            if (prim1 != SLOT_NO_SECOND) {
                                if (active && tri_test(vert(sel), vert(sel >> 2), vert(sel >> 4), to, td, tnear, tfar, t, u, v)) {
                                    if (ANY_HIT) {
                                        active = false;
                                        hit.tri = 0;
                                    } else {
                                        accept(t, u, v, prim1, 1u);
                                    }
                                }
                            }
                        }
                        if (ANY_HIT && __ballot(active) == 0ull) {
                            break; // every ray of the packet is occluded
                        }
                    }
                    // pop
                    if (sp == 0u) {
                        break;
                    }
                    --sp;
                    cur = (int32_t)pkt_uniform((uint32_t)wstack[sp]);
                }
                // retire: 64 consecutive items -> coalesced records; a lane may be handed a follow-up ray
                again = false;
                if (in_round) {
                    if (ANY_HIT && hit.tri == 0) {
                        hit.t = 0.f;
                    }
                    again = src.retire(i, stage, hit, wo, wd, tfar_var, carry);
                }
            }
        }
    }
}

} // namespace crt
