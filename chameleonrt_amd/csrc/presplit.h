// presplit.h — spatial PRE-SPLITTING of loosely boxed leaf items before the SAH build (opt-in: CRT_BVH_SPLITS=<fraction>).
//
// The builders work on one axis-aligned box per leaf slot. A long triangle (or quad) that is not aligned with the axes has a
// box many times its own size, which overlaps everything near it: the textbook weakness that spatial splits (Stich et al.
// 2009, SBVH) and their cheap cousin, early split clipping (Ernst & Greiner 2007; Embree's "presplits" of its high-quality
// builder), address by letting several leaves reference ONE primitive, each with the part of its box inside a cell. Here:
// the items with the most wasted box area (half area of the box minus the area of the geometry inside it) are cut at the
// midpoint of their box's longest axis, the triangles clipped against both halves and re-boxed, until `fraction` x n extra
// items have been made. A cut item's leaf slot is simply duplicated (64 B; its uv record with it): the kernels need no
// change -- a triangle reached through two leaves yields the same (t, instance, geomID, primID) twice, and the
// lexicographic closest-hit rule keeps one. Role in the reference: rtcCommitScene's builder (embree_utils.cpp:63-76); the
// reference builds at Embree's default quality, which does not split either, so this is an optimisation for real assets
// with large polygons, priced by tools/tree_cost.py, not a parity matter.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <queue>
#include <vector>

#include "bvh_builder.h"
#include "crt_types.h"

namespace crt {

struct PresplitStats {
    uint64_t items_in = 0, items_out = 0, cuts = 0;
    double area_in = 0.0, area_out = 0.0; // summed half areas of the item boxes
};

namespace presplit_detail {

struct P3 {
    float x[3];
};

// polygon clipped to the slab lo <= x[a] <= hi. A triangle cut by three slabs has at most 9 vertices when the arithmetic is
// exact; clip points are ROUNDED, though, and vertices within rounding of a later plane can make the in / out pattern
// alternate, each sign change adding a vertex (at most n / 2 more per half-plane). Inputs of more than CLIP_IN vertices are
// refused (-1: the caller leaves the item uncut), which bounds both stages inside arrays of CLIP_CAP (12 -> 18 -> 27).
constexpr int CLIP_IN = 12, CLIP_CAP = 32;
inline int clip_axis(const P3 *in, int n, int a, float lo, float hi, P3 *out)
{
    if (n > CLIP_IN) {
        return -1;
    }
    P3 tmp[CLIP_CAP];
    int m = 0;
    for (int i = 0; i < n; ++i) { // keep x[a] >= lo
        const P3 &p = in[i], &q = in[(i + 1) % n];
        const bool pin = p.x[a] >= lo, qin = q.x[a] >= lo;
        if (pin) {
            tmp[m++] = p;
        }
        if (pin != qin) {
            const float t = (lo - p.x[a]) / (q.x[a] - p.x[a]);
            P3 r;
            for (int k = 0; k < 3; ++k) {
                r.x[k] = p.x[k] + t * (q.x[k] - p.x[k]);
            }
            r.x[a] = lo;
            tmp[m++] = r;
        }
    }
    int o = 0;
    for (int i = 0; i < m; ++i) { // keep x[a] <= hi
        const P3 &p = tmp[i], &q = tmp[(i + 1) % m];
        const bool pin = p.x[a] <= hi, qin = q.x[a] <= hi;
        if (pin) {
            out[o++] = p;
        }
        if (pin != qin) {
            const float t = (hi - p.x[a]) / (q.x[a] - p.x[a]);
            P3 r;
            for (int k = 0; k < 3; ++k) {
                r.x[k] = p.x[k] + t * (q.x[k] - p.x[k]);
            }
            r.x[a] = hi;
            out[o++] = r;
        }
    }
    return o;
}

inline float poly_area(const P3 *p, int n)
{
    double sx = 0, sy = 0, sz = 0;
    for (int i = 1; i + 1 < n; ++i) {
        const double ax = p[i].x[0] - p[0].x[0], ay = p[i].x[1] - p[0].x[1], az = p[i].x[2] - p[0].x[2];
        const double bx = p[i + 1].x[0] - p[0].x[0], by = p[i + 1].x[1] - p[0].x[1], bz = p[i + 1].x[2] - p[0].x[2];
        sx += ay * bz - az * by;
        sy += az * bx - ax * bz;
        sz += ax * by - ay * bx;
    }
    return (float)(0.5 * std::sqrt(sx * sx + sy * sy + sz * sz));
}

inline float half_area(const Aabb &b)
{
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

} // namespace presplit_detail

// recs / boxes: one leaf slot and its box per item (the box in the space the tree is built in). xform(rec, m, pad): the
// column-major 4x4 that takes the slot's vertices into that space (m = nullptr: they are in it already) and the outward
// pad its box carries (scene_prepare.cpp instance_pad). Appends the new items; boxes only ever shrink.
template <typename Xform>
inline PresplitStats presplit_items(std::vector<LeafSlot> &recs, std::vector<Aabb> &boxes, Xform xform, double fraction)
{
    using namespace presplit_detail;
    PresplitStats st;
    st.items_in = recs.size();
    for (const Aabb &b : boxes) {
        st.area_in += half_area(b);
    }
    const uint64_t budget = (uint64_t)(fraction * (double)recs.size());
    // an item is worth cutting while more than `min_waste` of its box area is empty, and a cut is kept if the two halves'
    // boxes together are smaller than `max_kept` of the box they replace (tuning: CRT_BVH_SPLIT_WASTE / CRT_BVH_SPLIT_KEEP)
    const float min_waste = std::getenv("CRT_BVH_SPLIT_WASTE") ? (float)std::atof(std::getenv("CRT_BVH_SPLIT_WASTE")) : 0.25f;
    const float max_kept = std::getenv("CRT_BVH_SPLIT_KEEP") ? (float)std::atof(std::getenv("CRT_BVH_SPLIT_KEEP")) : 0.9f;
    // An item OWNS the part of its triangles inside its `cell` (all of space at first, halved by every cut: the halves are
    // closed and share the cut plane, so the parts cover the triangles exactly); its BOX is the box of that part pushed
    // out by the item's pad (a transformed instance's box has to cover where the rounded object-space test may put a
    // hit, scene_prepare.cpp instance_pad) plus the rounding of the clip points, and never leaves the item's original
    // box. So the two boxes of a cut overlap by twice the pad around the cut plane, and every point that the original
    // box had to cover for some part of the geometry is still covered by the box of the item that owns that part.
    auto part_box = [&](size_t item, const Aabb &cell, const Aabb &limit, Aabb &out, float &area) -> bool {
        const LeafSlot &s = recs[item];
        const float *m = nullptr;
        float pad = 0.f;
        xform(s, m, pad);
        P3 v[4];
        for (int k = 0; k < 4; ++k) {
            for (int a = 0; a < 3; ++a) {
                v[k].x[a] = m ? m[a] * s.v[k][0] + m[4 + a] * s.v[k][1] + m[8 + a] * s.v[k][2] + m[12 + a] : s.v[k][a];
            }
        }
        for (int a = 0; a < 3; ++a) {
            out.lo[a] = INFINITY;
            out.hi[a] = -INFINITY;
        }
        area = 0.f;
        const uint32_t sel = s.geom_sel >> SLOT_GEOM_BITS;
        for (int which = 0; which < (s.prim1 == SLOT_NO_SECOND ? 1 : 2); ++which) {
            P3 a0[CLIP_CAP], a1[CLIP_CAP];
            if (which == 0) {
                a0[0] = v[0], a0[1] = v[1], a0[2] = v[2];
            } else {
                a0[0] = v[sel & 3u], a0[1] = v[(sel >> 2) & 3u], a0[2] = v[(sel >> 4) & 3u];
            }
            int n = 3;
            for (int a = 0; a < 3 && n >= 3; ++a) {
                if (cell.lo[a] > -INFINITY || cell.hi[a] < INFINITY) {
                    n = clip_axis(a0, n, a, cell.lo[a], cell.hi[a], a1);
                    if (n < 0) {
                        return false; // a pathological polygon (see clip_axis): no cut for this item
                    }
                    std::copy(a1, a1 + n, a0);
                }
            }
            if (n < 3) {
                continue;
            }
            area += poly_area(a0, n);
            for (int i = 0; i < n; ++i) {
                for (int a = 0; a < 3; ++a) {
                    out.lo[a] = std::min(out.lo[a], a0[i].x[a]);
                    out.hi[a] = std::max(out.hi[a], a0[i].x[a]);
                }
            }
        }
        if (!(out.lo[0] <= out.hi[0])) {
            return false;
        }
        for (int a = 0; a < 3; ++a) {
            const float mag = std::max(std::fabs(out.lo[a]), std::fabs(out.hi[a]));
            const float eps = pad + 4e-7f * mag + 1e-30f; // (clip points: a few ulps of the coordinates)
            out.lo[a] = std::max(limit.lo[a], out.lo[a] - eps);
            out.hi[a] = std::min(limit.hi[a], out.hi[a] + eps);
        }
        return true;
    };
    struct Entry {
        float priority;
        uint64_t item;
        bool operator<(const Entry &o) const { return priority != o.priority ? priority < o.priority : item > o.item; }
    };
    const Aabb everywhere{{-INFINITY, -INFINITY, -INFINITY}, {INFINITY, INFINITY, INFINITY}};
    std::vector<Aabb> cells(recs.size(), everywhere), limits(boxes); // limits: the original box of the item a part descends from
    std::priority_queue<Entry> heap;
    for (uint64_t i = 0; i < recs.size(); ++i) {
        Aabb tight;
        float area = 0.f;
        const float h = half_area(boxes[i]);
        if (std::isfinite(h) && h > 0.f && part_box(i, everywhere, boxes[i], tight, area)) {
            const float waste = h - area;
            if (waste > min_waste * h) { // a box that is mostly its geometry (an axis-aligned quad) gains nothing from a cut
                heap.push(Entry{waste, i});
            }
        }
    }
    while (st.cuts < budget && !heap.empty()) {
        const Entry e = heap.top();
        heap.pop();
        const Aabb box = boxes[e.item];
        int axis = 0;
        for (int a = 1; a < 3; ++a) {
            if (box.hi[a] - box.lo[a] > box.hi[axis] - box.lo[axis]) {
                axis = a;
            }
        }
        const float mid = 0.5f * (box.lo[axis] + box.hi[axis]);
        if (!(mid > box.lo[axis] && mid < box.hi[axis])) {
            continue;
        }
        Aabb cl = cells[e.item], cr = cells[e.item], bl, br;
        cl.hi[axis] = std::min(cl.hi[axis], mid);
        cr.lo[axis] = std::max(cr.lo[axis], mid);
        float al = 0.f, ar = 0.f;
        const Aabb limit = limits[e.item];
        const bool okl = part_box(e.item, cl, limit, bl, al), okr = part_box(e.item, cr, limit, br, ar);
        if (!okl || !okr) {
            continue; // the geometry lies in one half only (the box was loose by its pad alone)
        }
        const float hl = half_area(bl), hr = half_area(br);
        if (!(hl + hr < max_kept * half_area(box))) {
            continue; // the cut does not pay for a second reference
        }
        boxes[e.item] = bl;
        cells[e.item] = cl;
        const uint64_t j = recs.size();
        recs.push_back(recs[e.item]);
        boxes.push_back(br);
        cells.push_back(cr);
        limits.push_back(limit);
        ++st.cuts;
        if (hl - al > min_waste * hl) {
            heap.push(Entry{hl - al, e.item});
        }
        if (hr - ar > min_waste * hr) {
            heap.push(Entry{hr - ar, j});
        }
    }
    st.items_out = recs.size();
    for (const Aabb &b : boxes) {
        st.area_out += half_area(b);
    }
    return st;
}

inline double presplit_fraction()
{
    const double f = std::getenv("CRT_BVH_SPLITS") ? std::atof(std::getenv("CRT_BVH_SPLITS")) : 0.0; // (read at every build: a test sets it per scene)
    return f > 0.0 ? std::min(f, 4.0) : 0.0;
}

} // namespace crt
