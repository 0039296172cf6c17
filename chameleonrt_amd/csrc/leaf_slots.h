// leaf_slots.h — which triangles share a 64-byte leaf slot (crt_types.h LeafSlot), and the slot records themselves.
//
// Embree builds its triangle-mesh BVHs over pairs of triangles that share an edge where it can (the reference's
// rtcCommitScene, embree_utils.cpp:63-76); so does this backend: two such triangles have four distinct vertices, which
// with the ids fill exactly one 64-byte cache line, and one leaf visit then tests both. Pairing is by vertex INDEX inside
// one geometry (the importers re-index on unique (position, normal, uv) triples, util/scene.cpp:114-186, so the two
// halves of a quad share two indices), over ALL edge neighbours -- not only the next triangle in the index buffer, which
// is what tessellators emit but not what every exporter does.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bvh_builder.h"
#include "crt_types.h"

namespace crt {

struct SlotTris {
    uint32_t a, b; // primitive ids inside the geometry; b == SLOT_NO_SECOND: a single triangle
};

// Two edge neighbours are paired if the box around both is no larger than what two separate leaves would cost a ray
// (surface-area measure): area(union) <= max_ratio * (area(A) + area(B)). The halves of a planar quad give 0.5.
inline float pair_max_ratio()
{
    static const float r = std::getenv("CRT_PAIR_MAX_RATIO") ? (float)std::atof(std::getenv("CRT_PAIR_MAX_RATIO")) : 1.0f;
    return r;
}

// Slots of one geometry, in ascending order of their first triangle. Greedy matching of edge-neighbour candidates in
// ascending order of the box ratio (ties: lower primitive ids first): deterministic, independent of the thread count.
inline std::vector<SlotTris> pair_triangles(const float *verts, const uint32_t *indices, uint64_t n_tris, float max_ratio)
{
    std::vector<SlotTris> slots;
    if (n_tris == 0) {
        return slots;
    }
    const uint32_t n = (uint32_t)n_tris;
    std::vector<uint32_t> mate(n, SLOT_NO_SECOND);
    if (max_ratio > 0.f && n > 1) {
        struct Edge {
            uint64_t key;
            uint32_t tri;
        };
        std::vector<Edge> edges;
        edges.reserve(3 * (size_t)n);
        for (uint32_t t = 0; t < n; ++t) {
            const uint32_t *ix = indices + 3 * (size_t)t;
            if (ix[0] == ix[1] || ix[1] == ix[2] || ix[0] == ix[2]) {
                continue; // degenerate: stays single
            }
            for (int e = 0; e < 3; ++e) {
                const uint32_t u = ix[e], v = ix[(e + 1) % 3];
                edges.push_back(Edge{((uint64_t)std::min(u, v) << 32) | std::max(u, v), t});
            }
        }
        std::sort(edges.begin(), edges.end(), [](const Edge &x, const Edge &y) { return x.key != y.key ? x.key < y.key : x.tri < y.tri; });
        auto half_area_of = [&](const uint32_t *ta, const uint32_t *tb) {
            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int k = 0; k < (tb ? 6 : 3); ++k) {
                const float *p = verts + 3 * (size_t)(k < 3 ? ta[k] : tb[k - 3]);
                for (int a = 0; a < 3; ++a) {
                    lo[a] = std::min(lo[a], p[a]);
                    hi[a] = std::max(hi[a], p[a]);
                }
            }
            const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
            return dx * dy + dy * dz + dz * dx;
        };
        struct Cand {
            float ratio;
            uint32_t a, b;
        };
        std::vector<Cand> cands;
        for (size_t i = 0; i + 1 < edges.size(); ++i) {
            if (edges[i].key != edges[i + 1].key || edges[i].tri == edges[i + 1].tri) {
                continue;
            }
            const uint32_t a = edges[i].tri, b = edges[i + 1].tri; // a < b
            const uint32_t *ta = indices + 3 * (size_t)a, *tb = indices + 3 * (size_t)b;
            const float sum = half_area_of(ta, nullptr) + half_area_of(tb, nullptr), both = half_area_of(ta, tb);
            const float ratio = sum > 0.f ? both / sum : (both > 0.f ? INFINITY : 0.5f);
            if (ratio <= max_ratio) { // (NaN coordinates never pair)
                cands.push_back(Cand{ratio, a, b});
            }
        }
        std::sort(cands.begin(), cands.end(), [](const Cand &x, const Cand &y) {
            return x.ratio != y.ratio ? x.ratio < y.ratio : (x.a != y.a ? x.a < y.a : x.b < y.b);
        });
        for (const Cand &c : cands) {
            if (mate[c.a] == SLOT_NO_SECOND && mate[c.b] == SLOT_NO_SECOND) {
                mate[c.a] = c.b;
                mate[c.b] = c.a;
            }
        }
    }
    slots.reserve(n);
    for (uint32_t t = 0; t < n; ++t) {
        if (mate[t] == SLOT_NO_SECOND) {
            slots.push_back(SlotTris{t, SLOT_NO_SECOND});
        } else if (t < mate[t]) {
            slots.push_back(SlotTris{t, mate[t]});
        }
    }
    return slots;
}

// The record of one slot of geometry `geom` (vertices in the geometry's own space) and what instance it belongs to.
inline LeafSlot make_leaf_slot(const float *verts, const uint32_t *indices, uint32_t geom, SlotTris st, uint32_t tag)
{
    LeafSlot s;
    const uint32_t *ia = indices + 3 * (size_t)st.a;
    for (int k = 0; k < 3; ++k) {
        std::memcpy(s.v[k], verts + 3 * (size_t)ia[k], 12);
    }
    std::memcpy(s.v[3], s.v[0], 12);
    uint32_t sel = 0;
    if (st.b != SLOT_NO_SECOND) {
        const uint32_t *ib = indices + 3 * (size_t)st.b;
        bool have_fourth = false;
        for (int k = 0; k < 3; ++k) {
            uint32_t where = 3;
            for (uint32_t j = 0; j < 3; ++j) {
                if (ib[k] == ia[j]) {
                    where = j;
                    break;
                }
            }
            if (where == 3) {
                // B's vertex that A does not have (at most one: the two share an edge); a second one would be a pairing bug
                if (have_fourth && std::memcmp(s.v[3], verts + 3 * (size_t)ib[k], 12) != 0) {
                    std::abort();
                }
                std::memcpy(s.v[3], verts + 3 * (size_t)ib[k], 12);
                have_fourth = true;
            }
            sel |= where << (2 * k);
        }
    }
    s.geom_sel = geom | (sel << SLOT_GEOM_BITS);
    s.prim0 = st.a;
    s.prim1 = st.b;
    s.tag = tag;
    return s;
}

} // namespace crt
