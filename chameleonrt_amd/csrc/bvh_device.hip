// bvh_device.hip — BLAS construction on the MI355X (the step before the hot path,
// rtcCommitScene in the reference, embree_utils.cpp:63-76).
//
// The host SAH builder (bvh_builder.cpp) takes seconds for a 10 M-triangle mesh; this path builds the same
// kind of tree -- 4-wide, 64-byte quantised nodes, leaves of one 64-byte slot (one or two triangles), slots in leaf order -- on
// the device in a fraction of that, at a lower tree quality (Morton-order splits instead of SAH; DESIGN.md
// section 7 has the measured node-visit ratio). Opt-in: CRT_HIP_BUILD=device.
//
//   0. (host)       which triangles share a leaf slot (leaf_slots.h: edge neighbours, paired once per geometry)
//   1. k_setup      leaf slots (four vertices, geomID, primIDs), boxes, mesh bounds            (per slot)
//   2. k_keys       63-bit Morton code of the box centre in the mesh bounds                   (per slot)
//   3. rocPRIM      radix sort of (key, slot)                                                 (library sort)
//   4. k_karras     binary radix tree over the sorted keys (Karras 2012; lbvh.h)             (per internal node)
//   5. k_refit      boxes bottom-up, second arriver at a node continues                      (per leaf)
//   6. k_collapse   level by level: binary subtree -> wide node, children allocated in the next level (BFS order)
//   7. k_emit       slots and their triangles' vertex UVs in leaf (= sorted) order
//
// The result is copied back into the host-side prepared scene, so everything after the build (TLAS,
// instance records, sharing between the GPUs of a node) is the one code path of scene_prepare.cpp.
#include <hip/hip_runtime.h>
#include <string.h> // rocprim's texture_cache_iterator.hpp calls ::memset without including it

#include <rocprim/rocprim.hpp>

#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "bvh_device.h"
#include "lbvh.h"
#include "leaf_slots.h"

namespace crt {
namespace {

#define BD_CHECK(expr)                                                                                    \
    do {                                                                                                  \
        hipError_t err__ = (expr);                                                                        \
        if (err__ != hipSuccess) {                                                                        \
            throw std::runtime_error(std::string("device BVH build: ") + #expr + ": " + hipGetErrorString(err__)); \
        }                                                                                                 \
    } while (0)

struct Buf {
    void *p = nullptr;
    void alloc(size_t n) { BD_CHECK(hipMalloc(&p, n ? n : 16)); }
    ~Buf()
    {
        if (p) {
            (void)hipFree(p);
        }
    }
    template <typename T> T *as() const { return static_cast<T *>(p); }
};

struct GeomDev {
    uint32_t tri_begin;  // first triangle of this geometry in the mesh-wide numbering
    uint32_t vert_begin; // first vertex in the concatenated vertex array
    int32_t uv_begin;    // first float2 in the concatenated uv array, -1: the geometry has no UVs
    uint32_t pad;
};

// order-preserving float <-> uint (for atomicMin / atomicMax on bounds)
__device__ inline uint32_t f2ord(float f)
{
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
inline float ord2f(uint32_t o)
{
    const uint32_t b = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    float f;
    std::memcpy(&f, &b, 4);
    return f;
}

// One entry per leaf slot of the tree, from the host's pairing: the geometry (index into the uploaded GeomDev array) and
// the one or two primitives it holds; for a world tree also the instance the slot belongs to.
struct SlotDev {
    uint32_t geom, a, b, inst; // b == SLOT_NO_SECOND: single; inst == NO_INSTANCE: a mesh's own BLAS
};
constexpr uint32_t NO_INSTANCE = 0xffffffffu;
// World tree: what a slot needs of its instance (scene_prepare.cpp build_world_tree: the same expressions on the host)
struct InstDev {
    float m[16];         // object_to_world, column-major
    float pad;           // how far the slot's world box is pushed out (instance_pad): 0 for an identity instance
    uint32_t identity;
    uint32_t geom_first; // first geometry of the instance's mesh: LeafSlot::geom_sel holds the geomID INSIDE the mesh
    uint32_t unused;
};

__global__ __launch_bounds__(256) void k_setup(uint32_t n, const GeomDev *geoms, const SlotDev *table, const InstDev *insts,
                                               const float *verts, const uint32_t *indices, LeafSlot *recs, Aabb *boxes, uint32_t *bounds)
{
    __shared__ uint32_t s_b[6];
    if (threadIdx.x < 6) {
        s_b[threadIdx.x] = threadIdx.x < 3 ? 0xffffffffu : 0u;
    }
    __syncthreads();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        const SlotDev sd = table[t];
        const GeomDev g = geoms[sd.geom];
        const bool instanced = sd.inst != NO_INSTANCE;
        const InstDev *in = instanced ? insts + sd.inst : nullptr;
        const bool xform = instanced && in->identity == 0u;
        // the expressions of make_leaf_slot / slot_box (leaf_slots.h, scene_prepare.cpp): same record, same box
        auto to_box = [&](const float *p, int a) -> float {
            return xform ? in->m[a] * p[0] + in->m[4 + a] * p[1] + in->m[8 + a] * p[2] + in->m[12 + a] : p[a];
        };
        const uint32_t *ia = indices + 3 * (size_t)(g.tri_begin + sd.a);
        const float *va[3] = {verts + 3 * (size_t)(g.vert_begin + ia[0]), verts + 3 * (size_t)(g.vert_begin + ia[1]),
                              verts + 3 * (size_t)(g.vert_begin + ia[2])};
        LeafSlot r;
        Aabb b;
        for (int a = 0; a < 3; ++a) {
            r.v[0][a] = va[0][a];
            r.v[1][a] = va[1][a];
            r.v[2][a] = va[2][a];
            r.v[3][a] = va[0][a];
            const float w0 = to_box(va[0], a), w1 = to_box(va[1], a), w2 = to_box(va[2], a);
            b.lo[a] = fminf(w0, fminf(w1, w2));
            b.hi[a] = fmaxf(w0, fmaxf(w1, w2));
        }
        uint32_t sel = 0;
        if (sd.b != SLOT_NO_SECOND) {
            const uint32_t *ib = indices + 3 * (size_t)(g.tri_begin + sd.b);
            for (int k = 0; k < 3; ++k) {
                uint32_t where = 3;
                for (uint32_t j = 0; j < 3; ++j) {
                    if (where == 3 && ib[k] == ia[j]) {
                        where = j;
                    }
                }
                const float *p = verts + 3 * (size_t)(g.vert_begin + ib[k]);
                if (where == 3) {
                    r.v[3][0] = p[0];
                    r.v[3][1] = p[1];
                    r.v[3][2] = p[2];
                }
                for (int a = 0; a < 3; ++a) {
                    const float w = to_box(p, a);
                    b.lo[a] = fminf(b.lo[a], w);
                    b.hi[a] = fmaxf(b.hi[a], w);
                }
                sel |= where << (2 * k);
            }
        }
        const float pad = instanced ? in->pad : 0.f;
        for (int a = 0; a < 3; ++a) {
            b.lo[a] -= pad;
            b.hi[a] += pad;
        }
        r.geom_sel = (sd.geom - (instanced ? in->geom_first : 0u)) | (sel << SLOT_GEOM_BITS);
        r.prim0 = sd.a;
        r.prim1 = sd.b;
        r.tag = instanced ? (sd.inst << 1) | (in->identity != 0u ? 1u : 0u) : 0u;
        recs[t] = r;
        boxes[t] = b;
        for (int a = 0; a < 3; ++a) {
            atomicMin(&s_b[a], f2ord(b.lo[a]));
            atomicMax(&s_b[3 + a], f2ord(b.hi[a]));
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        atomicMin(&bounds[threadIdx.x], s_b[threadIdx.x]);
    } else if (threadIdx.x < 6) {
        atomicMax(&bounds[threadIdx.x], s_b[threadIdx.x]);
    }
}

__global__ __launch_bounds__(256) void k_keys(uint32_t n, const Aabb *boxes, Aabb bounds, int mode, uint64_t *keys, uint32_t *idx)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) {
        return;
    }
    keys[t] = lbvh_key(boxes[t], bounds, mode);
    idx[t] = t;
}

// summed half-area of the internal nodes (the SAH estimate of node visits), one partial sum per block
__global__ __launch_bounds__(256) void k_cost(uint32_t n_internal, const Aabb *ibox, double *partial)
{
    __shared__ double s_sum[256];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    s_sum[threadIdx.x] = i < n_internal ? (double)lbvh_half_area(ibox[i]) : 0.0;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            s_sum[threadIdx.x] += s_sum[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = s_sum[0];
    }
}

__global__ __launch_bounds__(256) void k_sorted_boxes(uint32_t n, const Aabb *boxes, const uint32_t *idx, Aabb *pbox)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        pbox[t] = boxes[idx[t]];
    }
}

__global__ __launch_bounds__(256) void k_karras(uint32_t n, const uint64_t *keys, int32_t *left, int32_t *right, int32_t *lo,
                                                int32_t *hi, int32_t *parent_of_node, int32_t *parent_of_leaf)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 >= n) {
        return;
    }
    int32_t l, r;
    int a, b;
    lbvh_node(keys, (int)n, (int)i, l, r, a, b);
    left[i] = l;
    right[i] = r;
    lo[i] = a;
    hi[i] = b;
    if (l >= 0) {
        parent_of_node[l] = (int32_t)i;
    } else {
        parent_of_leaf[~l] = (int32_t)i;
    }
    if (r >= 0) {
        parent_of_node[r] = (int32_t)i;
    } else {
        parent_of_leaf[~r] = (int32_t)i;
    }
    if (i == 0) {
        parent_of_node[0] = -1;
    }
}

// One thread per leaf climbs towards the root; at every node the FIRST arriver stops and the second,
// which then sees both children complete, computes the node's box and continues.
__global__ __launch_bounds__(256) void k_refit(uint32_t n, const int32_t *left, const int32_t *right, const int32_t *parent_of_node,
                                               const int32_t *parent_of_leaf, const Aabb *pbox, Aabb *ibox, uint32_t *arrived)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) {
        return;
    }
    int32_t k = parent_of_leaf[p];
    while (k >= 0) {
        __threadfence(); // the boxes written below must be visible to whoever arrives second
        if (atomicAdd(&arrived[k], 1u) == 0u) {
            return;
        }
        __threadfence();
        const int32_t l = left[k], r = right[k];
        // (volatile: the sibling's box was written by another CU; the loads must not be served from this CU's L1)
        const volatile float *pl = reinterpret_cast<const volatile float *>(l >= 0 ? ibox + l : pbox + ~l);
        const volatile float *pr = reinterpret_cast<const volatile float *>(r >= 0 ? ibox + r : pbox + ~r);
        Aabb b;
        for (int a = 0; a < 3; ++a) {
            b.lo[a] = fminf(pl[a], pr[a]);
            b.hi[a] = fmaxf(pl[3 + a], pr[3 + a]);
        }
        ibox[k] = b;
        k = parent_of_node[k];
    }
}

__global__ __launch_bounds__(256) void k_collapse(LbvhTree t, const int32_t *frontier_in, uint32_t n_in, int32_t *frontier_out,
                                                  uint32_t *n_out, uint32_t level_base, QNode *nodes, QFrame frame,
                                                  uint32_t max_leaf)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_in) {
        return;
    }
    const int32_t k = frontier_in[i];
    int32_t sub[BVH_WIDTH];
    const int n = lbvh_wide_children(t, k, max_leaf, sub);
    QNode node;
    for (int c = 0; c < n; ++c) {
        const uint32_t count = lbvh_count(t, sub[c]);
        int32_t ref;
        if (count <= max_leaf) {
            ref = lbvh_leaf_ref(sub[c] >= 0 ? (uint32_t)t.lo[sub[c]] : (uint32_t)~sub[c], count);
        } else {
            const uint32_t slot = atomicAdd(n_out, 1u);
            frontier_out[slot] = sub[c];
            ref = (int32_t)(level_base + n_in + slot);
        }
        lbvh_quantise_child(node.child[c], lbvh_box(t, sub[c]), ref, frame);
    }
    for (int c = n; c < BVH_WIDTH; ++c) {
        lbvh_unused_child(node.child[c], node.child[0].ref);
    }
    nodes[level_base + i] = node;
}

__global__ __launch_bounds__(256) void k_emit(uint32_t n, const LeafSlot *recs, const uint32_t *idx, const GeomDev *geoms, const SlotDev *table,
                                              const uint32_t *indices, const float *uvs, LeafSlot *slots, float *tri_uvs)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) {
        return;
    }
    const LeafSlot r = recs[idx[p]];
    slots[p] = r;
    const GeomDev g = geoms[table[idx[p]].geom];
    for (int which = 0; which < 2; ++which) {
        const uint32_t prim = which == 0 ? r.prim0 : r.prim1;
        float out[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (g.uv_begin >= 0 && prim != SLOT_NO_SECOND) { // uv_buf[indices.x|y|z], render_embree.ispc:278-283
            for (int c = 0; c < 3; ++c) {
                const uint32_t vi = indices[3 * (size_t)(g.tri_begin + prim) + c];
                out[2 * c] = uvs[2 * (size_t)((uint32_t)g.uv_begin + vi)];
                out[2 * c + 1] = uvs[2 * (size_t)((uint32_t)g.uv_begin + vi) + 1];
            }
        }
        for (int c = 0; c < TRI_UV_STRIDE; ++c) {
            tri_uvs[(size_t)TRI_UV_STRIDE * (2 * (size_t)p + (size_t)which) + c] = c < 6 ? out[c] : 0.f;
        }
    }
}

inline unsigned grid_for(uint64_t n) { return (unsigned)((n + 255) / 256); }

} // namespace

namespace {

// The pipeline behind both entry points: geometry upload, k_setup over the slot table, keys / sort / radix tree / refit with
// both key normalisations, collapse, emit.
bool build_on_device(int device, const crt_geometry_desc *geoms, uint32_t n_geoms, const std::vector<SlotDev> &table,
                     const std::vector<InstDev> &inst_table, uint32_t max_leaf, uint32_t max_top_nodes, DeviceBuiltMesh &out)
{
    uint64_t n_tris = 0, n_verts = 0, n_uvs = 0;
    for (uint32_t g = 0; g < n_geoms; ++g) {
        n_tris += geoms[g].n_triangles;
        n_verts += geoms[g].n_vertices;
        n_uvs += geoms[g].uvs ? geoms[g].n_vertices : 0;
    }
    const uint64_t n_slots = table.size();
    if (n_slots < 2048 || n_slots >= (1ull << 28) || n_tris >= (1ull << 32) || n_verts >= (1ull << 32)) {
        return false; // small trees: the host builder takes milliseconds and builds the better tree
    }
    // the caller's current device is restored on every way out (a thread that drives another GPU, e.g. the GL
    // interop path, must not find its device changed), and the build runs on a stream of its own: the legacy default
    // stream would synchronise with every blocking stream of the device
    struct DeviceGuard {
        int prev = -1;
        hipStream_t stream = nullptr;
        ~DeviceGuard()
        {
            if (stream) {
                (void)hipStreamDestroy(stream);
            }
            if (prev >= 0) {
                (void)hipSetDevice(prev);
            }
        }
    } guard;
    BD_CHECK(hipGetDevice(&guard.prev));
    BD_CHECK(hipSetDevice(device));
    BD_CHECK(hipStreamCreateWithFlags(&guard.stream, hipStreamNonBlocking));
    hipStream_t s = guard.stream;
    const uint32_t n = (uint32_t)n_slots; // the items of the tree: leaf slots

    // inputs, straight from the caller's arrays into concatenated device arrays
    Buf d_verts, d_indices, d_uvs, d_geoms;
    d_verts.alloc(n_verts * 12);
    d_indices.alloc(n_tris * 12);
    d_uvs.alloc(n_uvs * 8);
    std::vector<GeomDev> gd(n_geoms);
    {
        uint64_t t0 = 0, v0 = 0, u0 = 0;
        for (uint32_t g = 0; g < n_geoms; ++g) {
            gd[g].tri_begin = (uint32_t)t0;
            gd[g].vert_begin = (uint32_t)v0;
            gd[g].uv_begin = geoms[g].uvs ? (int32_t)u0 : -1;
            gd[g].pad = 0;
            if (geoms[g].n_vertices) {
                BD_CHECK(hipMemcpyAsync(d_verts.as<float>() + 3 * v0, geoms[g].vertices, geoms[g].n_vertices * 12, hipMemcpyHostToDevice, s));
            }
            if (geoms[g].n_triangles) {
                BD_CHECK(hipMemcpyAsync(d_indices.as<uint32_t>() + 3 * t0, geoms[g].indices, geoms[g].n_triangles * 12,
                                        hipMemcpyHostToDevice, s));
            }
            if (geoms[g].uvs && geoms[g].n_vertices) {
                BD_CHECK(hipMemcpyAsync(d_uvs.as<float>() + 2 * u0, geoms[g].uvs, geoms[g].n_vertices * 8, hipMemcpyHostToDevice, s));
                u0 += geoms[g].n_vertices;
            }
            t0 += geoms[g].n_triangles;
            v0 += geoms[g].n_vertices;
        }
    }
    d_geoms.alloc(n_geoms * sizeof(GeomDev));
    BD_CHECK(hipMemcpyAsync(d_geoms.p, gd.data(), n_geoms * sizeof(GeomDev), hipMemcpyHostToDevice, s));
    Buf d_table, d_insts;
    d_table.alloc((size_t)n * sizeof(SlotDev));
    BD_CHECK(hipMemcpyAsync(d_table.p, table.data(), (size_t)n * sizeof(SlotDev), hipMemcpyHostToDevice, s));
    d_insts.alloc(inst_table.size() * sizeof(InstDev));
    if (!inst_table.empty()) {
        BD_CHECK(hipMemcpyAsync(d_insts.p, inst_table.data(), inst_table.size() * sizeof(InstDev), hipMemcpyHostToDevice, s));
    }

    Buf d_recs, d_boxes, d_bounds;
    d_recs.alloc((size_t)n * sizeof(LeafSlot));
    d_boxes.alloc((size_t)n * sizeof(Aabb));
    d_bounds.alloc(6 * 4);
    {
        const uint32_t init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
        BD_CHECK(hipMemcpyAsync(d_bounds.p, init, sizeof(init), hipMemcpyHostToDevice, s));
    }
    k_setup<<<grid_for(n), 256, 0, s>>>(n, d_geoms.as<GeomDev>(), d_table.as<SlotDev>(), d_insts.as<InstDev>(), d_verts.as<float>(), d_indices.as<uint32_t>(),
                                        d_recs.as<LeafSlot>(), d_boxes.as<Aabb>(), d_bounds.as<uint32_t>());
    uint32_t hb[6];
    BD_CHECK(hipMemcpyAsync(hb, d_bounds.p, sizeof(hb), hipMemcpyDeviceToHost, s));
    BD_CHECK(hipStreamSynchronize(s));
    Aabb bounds;
    for (int a = 0; a < 3; ++a) {
        bounds.lo[a] = ord2f(hb[a]);
        bounds.hi[a] = ord2f(hb[3 + a]);
    }
    out.bounds = bounds;
    const QFrame frame = make_frame(bounds);

    // The binary radix tree, built with both key normalisations (lbvh.h); the one with the smaller summed
    // surface area of its internal nodes is kept.
    struct TreeBufs {
        Buf keys, idx, pbox, ibox, left, right, lo, hi;
    };
    Buf d_keys_in, d_idx_in, d_tmp, d_pn, d_pl, d_arrived, d_partial;
    d_keys_in.alloc((size_t)n * 8);
    d_idx_in.alloc((size_t)n * 4);
    for (Buf *b : {&d_pn, &d_pl, &d_arrived}) {
        b->alloc((size_t)n * 4);
    }
    const unsigned cost_blocks = grid_for(n - 1);
    d_partial.alloc((size_t)cost_blocks * 8);
    size_t tmp_bytes = 0;
    TreeBufs trees[LBVH_KEY_MODES];
    double cost[LBVH_KEY_MODES];
    for (int mode = 0; mode < LBVH_KEY_MODES; ++mode) {
        TreeBufs &tb = trees[mode];
        tb.keys.alloc((size_t)n * 8);
        tb.idx.alloc((size_t)n * 4);
        tb.pbox.alloc((size_t)n * sizeof(Aabb));
        tb.ibox.alloc((size_t)n * sizeof(Aabb));
        for (Buf *b : {&tb.left, &tb.right, &tb.lo, &tb.hi}) {
            b->alloc((size_t)n * 4);
        }
        k_keys<<<grid_for(n), 256, 0, s>>>(n, d_boxes.as<Aabb>(), bounds, mode, d_keys_in.as<uint64_t>(), d_idx_in.as<uint32_t>());
        if (!d_tmp.p) {
            BD_CHECK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys_in.as<uint64_t>(), tb.keys.as<uint64_t>(),
                                               d_idx_in.as<uint32_t>(), tb.idx.as<uint32_t>(), n, 0, 63, s));
            d_tmp.alloc(tmp_bytes);
        }
        BD_CHECK(rocprim::radix_sort_pairs(d_tmp.p, tmp_bytes, d_keys_in.as<uint64_t>(), tb.keys.as<uint64_t>(), d_idx_in.as<uint32_t>(),
                                           tb.idx.as<uint32_t>(), n, 0, 63, s));
        BD_CHECK(hipMemsetAsync(d_arrived.p, 0, (size_t)n * 4, s));
        k_sorted_boxes<<<grid_for(n), 256, 0, s>>>(n, d_boxes.as<Aabb>(), tb.idx.as<uint32_t>(), tb.pbox.as<Aabb>());
        k_karras<<<grid_for(n), 256, 0, s>>>(n, tb.keys.as<uint64_t>(), tb.left.as<int32_t>(), tb.right.as<int32_t>(), tb.lo.as<int32_t>(),
                                             tb.hi.as<int32_t>(), d_pn.as<int32_t>(), d_pl.as<int32_t>());
        k_refit<<<grid_for(n), 256, 0, s>>>(n, tb.left.as<int32_t>(), tb.right.as<int32_t>(), d_pn.as<int32_t>(), d_pl.as<int32_t>(),
                                            tb.pbox.as<Aabb>(), tb.ibox.as<Aabb>(), d_arrived.as<uint32_t>());
        k_cost<<<cost_blocks, 256, 0, s>>>(n - 1, tb.ibox.as<Aabb>(), d_partial.as<double>());
        std::vector<double> partial(cost_blocks);
        BD_CHECK(hipMemcpyAsync(partial.data(), d_partial.p, (size_t)cost_blocks * 8, hipMemcpyDeviceToHost, s));
        BD_CHECK(hipStreamSynchronize(s));
        cost[mode] = 0.0;
        for (double v : partial) {
            cost[mode] += v;
        }
    }
    const TreeBufs &best = cost[0] <= cost[1] ? trees[0] : trees[1];
    const uint32_t *idx = best.idx.as<uint32_t>();

    // collapse, level by level; node indices come out in BFS order
    Buf d_nodes, d_front_a, d_front_b, d_count;
    d_nodes.alloc((size_t)n * sizeof(QNode));
    d_front_a.alloc((size_t)n * 4);
    d_front_b.alloc((size_t)n * 4);
    d_count.alloc(4);
    const LbvhTree tree{best.left.as<int32_t>(), best.right.as<int32_t>(), best.lo.as<int32_t>(), best.hi.as<int32_t>(),
                        best.ibox.as<Aabb>(), best.pbox.as<Aabb>()};
    const int32_t root = 0;
    BD_CHECK(hipMemcpyAsync(d_front_a.p, &root, 4, hipMemcpyHostToDevice, s));
    uint32_t n_in = 1, level_base = 0, depth = 0;
    int32_t *fin = d_front_a.as<int32_t>(), *fout = d_front_b.as<int32_t>();
    while (n_in > 0) {
        BD_CHECK(hipMemsetAsync(d_count.p, 0, 4, s));
        k_collapse<<<grid_for(n_in), 256, 0, s>>>(tree, fin, n_in, fout, d_count.as<uint32_t>(), level_base, d_nodes.as<QNode>(), frame,
                                                  max_leaf);
        uint32_t n_next = 0;
        BD_CHECK(hipMemcpyAsync(&n_next, d_count.p, 4, hipMemcpyDeviceToHost, s));
        BD_CHECK(hipStreamSynchronize(s));
        level_base += n_in;
        n_in = n_next;
        std::swap(fin, fout);
        ++depth;
        if (depth > 4096) {
            throw std::runtime_error("device BVH build: collapse does not terminate");
        }
    }
    const uint32_t n_nodes = level_base;

    // slots + their triangles' vertex UVs in leaf order
    Buf d_slots, d_tuv;
    d_slots.alloc((size_t)n * sizeof(LeafSlot));
    d_tuv.alloc((size_t)n * 2 * TRI_UV_STRIDE * 4);
    k_emit<<<grid_for(n), 256, 0, s>>>(n, d_recs.as<LeafSlot>(), idx, d_geoms.as<GeomDev>(), d_table.as<SlotDev>(), d_indices.as<uint32_t>(), d_uvs.as<float>(),
                                       d_slots.as<LeafSlot>(), d_tuv.as<float>());
    BD_CHECK(hipGetLastError());
    out.nodes.resize(n_nodes);
    out.slots.resize(n);
    out.tri_uvs.resize((size_t)n * 2 * TRI_UV_STRIDE);
    BD_CHECK(hipMemcpyAsync(out.nodes.data(), d_nodes.p, (size_t)n_nodes * sizeof(QNode), hipMemcpyDeviceToHost, s));
    BD_CHECK(hipMemcpyAsync(out.slots.data(), d_slots.p, (size_t)n * sizeof(LeafSlot), hipMemcpyDeviceToHost, s));
    BD_CHECK(hipMemcpyAsync(out.tri_uvs.data(), d_tuv.p, (size_t)n * 2 * TRI_UV_STRIDE * 4, hipMemcpyDeviceToHost, s));
    BD_CHECK(hipStreamSynchronize(s));
    out.max_depth = depth;
    out.n_top = std::min(n_nodes, max_top_nodes);
    out.frame = frame;
    return true;
}

} // namespace

bool device_build_mesh(int device, const crt_geometry_desc *geoms, uint32_t n_geoms, const std::vector<SlotTris> *geom_slots,
                       uint32_t max_leaf, uint32_t max_top_nodes, DeviceBuiltMesh &out)
{
    std::vector<SlotDev> table;
    for (uint32_t g = 0; g < n_geoms; ++g) {
        for (const SlotTris &st : geom_slots[g]) {
            table.push_back(SlotDev{g, st.a, st.b, NO_INSTANCE});
        }
    }
    return build_on_device(device, geoms, n_geoms, table, {}, max_leaf, max_top_nodes, out);
}

bool device_build_world(int device, const crt_scene_desc *scene, const std::vector<SlotTris> *geom_slots, const uint32_t *inst_identity,
                        const float *inst_pad, uint32_t max_leaf, uint32_t max_top_nodes, DeviceBuiltMesh &out)
{
    std::vector<InstDev> insts(scene->n_instances);
    std::vector<SlotDev> table;
    for (uint32_t i = 0; i < scene->n_instances; ++i) {
        const crt_instance_desc &id = scene->instances[i];
        const crt_mesh_desc &md = scene->meshes[scene->parameterized_meshes[id.parameterized_mesh_id].mesh_id];
        InstDev &d = insts[i];
        std::memcpy(d.m, id.transform, sizeof(d.m));
        d.pad = inst_pad[i];
        d.identity = inst_identity[i];
        d.geom_first = md.first_geometry;
        d.unused = 0;
        for (uint32_t k = 0; k < md.n_geometries; ++k) { // the order of scene_prepare.cpp's host loop: instance, geometry, slot
            for (const SlotTris &st : geom_slots[md.first_geometry + k]) {
                table.push_back(SlotDev{md.first_geometry + k, st.a, st.b, i});
            }
        }
    }
    return build_on_device(device, scene->geometries, scene->n_geometries, table, insts, max_leaf, max_top_nodes, out);
}

} // namespace crt
