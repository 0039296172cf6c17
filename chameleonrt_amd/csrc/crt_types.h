// crt_types.h — device-resident scene layout shared by the host core and the HIP kernels.
// (DESIGN.md "Data layout in HBM".) Everything is plain 32-bit words; records are sized and
// aligned so a lane fetches them with dwordx4 loads.
#pragma once
#include <stdint.h>

namespace crt {

// Child reference c: c >= 0 -> inner node index (global, into Scene::nodes)
//                    c <  0 -> leaf, x = ~c: first = x >> 3, count = (x & 7) + 1
//                              BLAS: leaf slots [first, first+count) of SceneView::slots (the builders make count == 1)
//                              top-level tree of a scene with more than one instance: count == 8 (a value no
//                              triangle leaf has: the builder makes leaves of <= 7) marks an INSTANCE leaf, instance
//                              `first`; any other count is a triangle leaf of the instance that was grafted into the
//                              top-level tree (SceneView::world_inst, scene_prepare.cpp), tested with the world-space ray
//                    c == EMPTY_CHILD -> unused slot of a node with fewer than BVH_WIDTH children (builder output only)
// The BVH is 4-wide: one fetch decides four children, which about halves the chain of dependent
// node fetches of a ray and the per-node bookkeeping (DESIGN.md "Traversal"; what bounds the kernel
// is analysed there with PMC counters, not assumed here). BvhNode is what the host builder produces (full-precision
// boxes of all children), QNode its 16-bit fixed-point form (what every builder delivers), PNode the packed form the
// traversal kernels read (48 bytes of a 64-byte record).
constexpr int BVH_WIDTH = 4;
constexpr int32_t EMPTY_CHILD = (int32_t)0x80000002;
#if defined(__HIPCC__)
#define CRT_TYPES_HD __host__ __device__ inline
#else
#define CRT_TYPES_HD inline
#endif
// (none of EMPTY_CHILD and the traversal's CUR_DONE / CUR_EXIT markers has its low three bits clear; the stack
// sentinel 0x80000000 has -- it would be instance 2^28 - 1, which check_scene refuses for every kind of scene -- and is
// excluded explicitly, so that no pop path depends on rewriting it first)
CRT_TYPES_HD int32_t instance_leaf_ref(uint32_t instance) { return (int32_t)~((instance << 3) | 7u); }
CRT_TYPES_HD bool is_instance_leaf(int32_t ref) { return ref < 0 && (ref & 7) == 0 && ref != (int32_t)0x80000000; }
static_assert((EMPTY_CHILD & 7) != 0, "EMPTY_CHILD must not look like an instance leaf");
struct alignas(16) BvhNode {
    float lo[BVH_WIDTH][3], hi[BVH_WIDTH][3];
    int32_t c[BVH_WIDTH];
};
static_assert(sizeof(BvhNode) == 112, "BvhNode must be 112 bytes");

// Fixed-point frame of one BVH: world/object coordinate = base + q * step, q in [0, 65535].
struct QFrame {
    float base[3];
    float step[3];
};

// One BVH4 node as the BUILDERS deliver it (rounds 1-3: also what the kernels read): 64 B, one 16-byte quarter per child:
// its AABB as 16-bit fixed point in the BVH's QFrame (one lo|hi dword per axis), rounded OUTWARD by at least one quantum
// (conservative: a box may only grow, so no hit can be missed; which triangle wins never depends
// on the boxes), and its reference. scene_prepare.cpp packs it into a PNode (below) at the end.
struct alignas(16) QChild {
    uint16_t q[3][2]; // per axis: {lo, hi} -> one dword per axis, lo in the low half
    int32_t ref;
};
// An unused slot holds an INVERTED box (lo = 65535, hi = 0 on every axis), which the slab test
// rejects by itself because it picks the near plane by the sign of the ray direction instead of
// symmetrising with min/max (slab.h), and a COPY of slot 0's reference: should a degenerate ray get
// through (origin so far away that both planes round to the same parameter), it revisits a sibling,
// which cannot change any result. EMPTY_CHILD marks unused slots in the builder's BvhNode only.
struct alignas(16) QNode {
    QChild child[BVH_WIDTH];
};
static_assert(sizeof(QNode) == 64, "QNode must be 64 bytes");

// The node as the TRAVERSAL KERNELS read it: the same four boxes and references in 48 bytes of a 64-byte record, fetched
// with THREE dwordx4 per lane instead of four. tools/node_bytes_microbench.hip (profiles/r04_node_bytes_microbench.txt):
// an incoherent 64-byte node visit costs the CU's vector-memory front end 2.77 cycles per lane, the 48 bytes of a
// 64-byte-aligned record 2.06 -- the cost follows the number of 16-byte requests, and the front end is what binds the
// traversal kernels. QNode stays the builders' output (host SAH, host and device LBVH, the world-tree
// cut); scene_prepare.cpp packs every node once, at the end (pack_node below), and the packed form is what is uploaded,
// saved, and copied out (crt_hip_bvh_copy).
//
// Per axis the node has an ORIGIN on the BVH's 16-bit grid (the smallest `lo` of its used children) and a SCALE
// 2^e or 1.5 x 2^e grid units (five bits: e << 1 | m); a child's plane is origin + byte x scale: lo rounded down, hi
// rounded up to a multiple of the scale, the scale the smallest of that sequence with which the farthest `hi` still
// fits a byte. Boxes only ever grow (by less than one scale step, i.e. less than 1/170 of the node's extent), so the
// conservative-box argument of QNode carries over; nodes no wider than 255 grid units -- every node near the leaves --
// have scale 1 and exactly the boxes they had. An unused slot has lo = 255, hi = 0 (inverted for every ray direction,
// slab.h) and a copy of slot 0's reference, as in QNode.
// (The format has room for half steps -- scales 1.5 x 2^e, the m bit of the code -- which make the boxes of C2 / C3 / C4
// 0.5 / 0.7 / 0.3 % tighter in node visits; the kernels then need a multiply by a constructed float per axis instead of one
// ldexp, and that costs more than the visits save: C4 56.7 against 55.8 ms, profiles/r04_issue_bound_ab.txt. Off.)
#ifndef CRT_PNODE_HALF_STEPS
#define CRT_PNODE_HALF_STEPS 0
#endif
struct alignas(16) PNode {
    uint32_t frame[2];  // [0] = origin_x | origin_y << 16; [1] = origin_z | scale_x << 16 | scale_y << 21 | scale_z << 26
    uint32_t lo_x, hi_x; // byte c = child c
    uint32_t lo_y, hi_y, lo_z, hi_z;
    int32_t ref[BVH_WIDTH];
    uint32_t unused[4]; // (zero; never fetched by the kernels)
};
static_assert(sizeof(PNode) == 64, "PNode must be 64 bytes");
// (the part of a PNode the kernels read: what the LDS copy of the top levels holds per node)
struct alignas(16) PNodeHead {
    uint32_t w[12];
};

CRT_TYPES_HD PNode pack_node(const QNode &q)
{
    PNode p;
    uint32_t origin[3], scale[3], lo[3] = {0u, 0u, 0u}, hi[3] = {0u, 0u, 0u};
    for (int a = 0; a < 3; ++a) {
        uint32_t omin = 0xffffffffu, hmax = 0u;
        for (int c = 0; c < BVH_WIDTH; ++c) {
            if (q.child[c].q[0][0] <= q.child[c].q[0][1]) { // (an unused slot is inverted on every axis)
                omin = q.child[c].q[a][0] < omin ? q.child[c].q[a][0] : omin;
                hmax = q.child[c].q[a][1] > hmax ? q.child[c].q[a][1] : hmax;
            }
        }
        if (omin == 0xffffffffu) {
            omin = hmax = 0u; // no used child (the empty scene's node)
        }
        // twice the scale, so that 1.5 x 2^e stays an integer: 2, 3, 4, 6, 8, 12, ... = (2 + m) << e
        // (CRT_PNODE_HALF_STEPS = 0: powers of two only -- the kernels then scale with one ldexp)
        uint32_t code = 0u, scale2 = 2u;
        while ((2u * (hmax - omin) + scale2 - 1u) / scale2 > 255u) {
            code += CRT_PNODE_HALF_STEPS ? 1u : 2u;
            scale2 = (2u + (code & 1u)) << (code >> 1);
        }
        origin[a] = omin;
        scale[a] = code;
        for (int c = 0; c < BVH_WIDTH; ++c) {
            uint32_t l = 255u, h = 0u;
            if (q.child[c].q[0][0] <= q.child[c].q[0][1]) {
                l = 2u * (q.child[c].q[a][0] - omin) / scale2;
                h = (2u * (q.child[c].q[a][1] - omin) + scale2 - 1u) / scale2;
            }
            lo[a] |= l << (8 * c);
            hi[a] |= h << (8 * c);
        }
    }
    p.frame[0] = origin[0] | origin[1] << 16;
    p.frame[1] = origin[2] | scale[0] << 16 | scale[1] << 21 | scale[2] << 26;
    p.lo_x = lo[0], p.hi_x = hi[0], p.lo_y = lo[1], p.hi_y = hi[1], p.lo_z = lo[2], p.hi_z = hi[2];
    for (int c = 0; c < BVH_WIDTH; ++c) {
        p.ref[c] = q.child[c].ref;
        p.unused[c] = 0u;
    }
    return p;
}

// One LEAF of a BVH = one 64-byte slot = 4 x dwordx4 in ONE cache line: a triangle, or two triangles of one geometry
// (and one instance) that share an edge -- a quad, Embree's own leaf form for triangle meshes -- stored as the four
// distinct vertices in full precision plus the ids. A leaf visit is a dependent step that costs a line fill whatever
// it fetches (tools/line_microbench.hip: 2.9 CU-cycles for 4 requests to one line, 4.1 for the 6 requests of two
// 48-byte triangle records), and the second triangle of the earlier two-record leaf was a second dependent round trip
// in the kernels that could not afford to preload it (profiles/r03_wave_phase_profile.txt: leaf steps of 13 000 cycles,
// 41 % of the closest-hit kernel's wave time on C4).
//   triangle A = (v[0], v[1], v[2])                      the vertices of primitive prim0 in index order
//   triangle B = (v[s0], v[s1], v[s2]), s_k two bits     the vertices of primitive prim1 in ITS index order: each is one
//                                                        of A's vertices or v[3]
// The kernels form e1 = a - b, e2 = c - a from the selected vertices (a, b, c): the same IEEE subtractions the host made
// for the 48-byte (v0, e1, e2) records of rounds 1-2, so t / u / v keep their bits (e1 = v0 - v1,
// e2 = v2 - v0, Ng = cross(e2, e1); geom = Embree geomID, prim = Embree primID).
// A hit is named by tri = 2 * slot + (0 for A, 1 for B): HitBuf, SceneView::tri_uvs (two uv records per slot).
struct alignas(16) LeafSlot {
    float v[4][3];
    uint32_t geom_sel; // bits 0..25: geomID; bits 26..31: s0 | s1 << 2 | s2 << 4 (0 in a single-triangle slot)
    uint32_t prim0;
    uint32_t prim1;    // SLOT_NO_SECOND: the slot holds triangle A only (v[3] then repeats v[0])
    uint32_t tag;      // 0, except in a world tree (LEVELS_WORLD_TREE below): (instance << 1) | 1 if its transform is the identity
};
static_assert(sizeof(LeafSlot) == 64, "LeafSlot must be 64 bytes");
constexpr uint32_t SLOT_NO_SECOND = 0xffffffffu;
constexpr uint32_t SLOT_GEOM_BITS = 26;
constexpr uint32_t SLOT_GEOM_MASK = (1u << SLOT_GEOM_BITS) - 1u;

// One instance (util/mesh.h:40-47 + embree_utils.cpp:90-104), 128 B. What a ray needs when it enters the
// instance sits in the first 80 bytes, laid out for five 16-byte requests: the affine part of
// world_to_object (three requests), then BLAS root, identity flag and the BLAS's frame (two).
struct alignas(16) InstanceRec {
    float w2o[12];      // world_to_object without its constant last row: column c, row r at w2o[c*3 + r]
    int32_t blas_root;  // node index of the mesh's BLAS root
    uint32_t identity;  // 1 if the transform is bit-exactly the identity (ray not transformed)
    QFrame frame;       // fixed-point frame of the mesh's BLAS
    uint32_t geom_base; // global geometry index of the mesh's geometry 0
    uint32_t mat_base;  // offset into Scene::material_ids for this instance's geomID 0
    uint32_t pad[10];
};
static_assert(sizeof(InstanceRec) == 128, "InstanceRec must be 128 bytes");

// ISPCTexture2D (backends/embree/texture2d.ih:6-11); texels live in one byte blob.
struct alignas(16) TexRec { // 16 bytes: one request fetches it
    int32_t width, height, channels;
    uint32_t offset16; // byte offset of the texture in Scene::texels, in units of 16 bytes
};
// Texels are stored in tiles of 8 x 4: with 4 channels (every texture the reference's importers load, stb_image is
// asked for 4) a tile is one 128-byte cache line, and the 2 x 2 footprint of a bilinear lookup lies in 1.4 lines on
// average where rows of texels put it in 2.06 (incoherent lookups pay per LINE, tools/line_microbench.hip). Slot of
// texel (x, y), in texels; the texture occupies tex_tiled_texels() slots (edge tiles are padded, never addressed).
constexpr int TEX_TILE_W_LOG2 = 3, TEX_TILE_H_LOG2 = 2;
CRT_TYPES_HD uint32_t tex_tiles_x(int32_t width) { return ((uint32_t)width + (1u << TEX_TILE_W_LOG2) - 1u) >> TEX_TILE_W_LOG2; }
CRT_TYPES_HD uint32_t tex_row_part(uint32_t tiles_x, int32_t y)
{
    return ((((uint32_t)y >> TEX_TILE_H_LOG2) * tiles_x) << (TEX_TILE_W_LOG2 + TEX_TILE_H_LOG2)) |
           (((uint32_t)y & ((1u << TEX_TILE_H_LOG2) - 1u)) << TEX_TILE_W_LOG2);
}
CRT_TYPES_HD uint32_t tex_col_part(int32_t x)
{
    return (((uint32_t)x >> TEX_TILE_W_LOG2) << (TEX_TILE_W_LOG2 + TEX_TILE_H_LOG2)) | ((uint32_t)x & ((1u << TEX_TILE_W_LOG2) - 1u));
}
CRT_TYPES_HD uint32_t tex_slot(int32_t width, int32_t x, int32_t y) { return tex_row_part(tex_tiles_x(width), y) + tex_col_part(x); }
CRT_TYPES_HD uint64_t tex_tiled_texels(int32_t width, int32_t height)
{
    const uint64_t tiles_y = ((uint64_t)height + (1u << TEX_TILE_H_LOG2) - 1u) >> TEX_TILE_H_LOG2;
    return ((uint64_t)tex_tiles_x(width) * tiles_y) << (TEX_TILE_W_LOG2 + TEX_TILE_H_LOG2);
}

// ViewParams (backends/embree/embree_utils.h:137-140) + framebuffer geometry.
struct ViewParams {
    float pos[3], dir_du[3], dir_dv[3], dir_top_left[3];
    uint32_t frame_id;
    uint32_t fb_width, fb_height;
    uint32_t spp;
    uint32_t n_tiles_x;
};

// Device pointers of the whole scene, passed to kernels by value.
struct SceneView {
    const PNode *nodes;
    const LeafSlot *slots;        // leaves: one or two triangles each (LeafSlot above)
    const InstanceRec *instances;
    const float *tri_uvs;         // TRI_UV_STRIDE floats per TRIANGLE index 2 * slot + which (uv of its three vertices, two of padding: 2 x dwordx4)
    const uint32_t *material_ids; // per instance per geomID; bit 31 (MATERIAL_TEXTURED): the material reads a texture
    const float *materials;       // 16 floats per material (14 used, MaterialParams order)
    const TexRec *textures;
    const uint8_t *texels;
    const float *lights;          // 20 floats per QuadLight
    uint32_t n_lights;
    uint32_t n_instances;
    QFrame root_frame;            // frame of the BVH `root` belongs to
    int32_t root;                 // TLAS root (two-level) or the single BLAS root
    uint32_t two_level;           // 0: exactly one instance, traverse its BLAS directly; 1: top-level tree over instances;
                                  // LEVELS_WORLD_TREE: one tree over the triangles of all instances, in world space
    int32_t world_inst;           // two level: the instance whose triangles sit in the top-level tree itself, or -1
    uint32_t n_top_nodes;         // nodes [root, root + n_top_nodes) are the BFS-ordered top levels
    int32_t *stack_spill;         // traversal-stack overflow slab, [wave of the persistent grid][depth][lane]
    uint32_t spill_stride;        // threads the slab was sized for
    uint32_t spill_depth;         // entries per lane in the slab (sized at set_scene from the depth of this scene's BVH)
};

// SceneView::two_level == LEVELS_WORLD_TREE ("world tree"): the scene has several instances, but memory is not what an
// MI355X is short of (288 GB): every instance gets leaf slots of its own -- still in ITS object space, with
// LeafSlot::tag = (instance << 1) | identity -- and ONE tree is built over all of them from the boxes of their
// transformed vertices. A ray walks that tree in world space from start to end (no instance entry, no second root,
// no frame change, no exit) and is transformed into an instance's object space only to test a triangle of it, with
// the two-level entry's expressions, so hits are bit-identical to the two-level walk and to the reference's
// per-instance intersection. scene_prepare.cpp decides per scene (memory budget); traverse.h INST_TRIS.
constexpr uint32_t LEVELS_WORLD_TREE = 2u;

constexpr uint32_t MATERIAL_TEXTURED = 0x80000000u; // flag on the entries of SceneView::material_ids (and HitBuf::mat)
constexpr int TRI_UV_STRIDE = 8;      // floats per triangle (2 per leaf slot) in SceneView::tri_uvs
constexpr int TILE = 64;              // the reference's tile edge (render_embree.h:25)
constexpr int TILE_PIXELS = TILE * TILE;
constexpr float RAY_EPS = 0.0001f;    // EPSILON, backends/embree/util.ih:8
constexpr int MAX_PATH_DEPTH = 5;     // backends/embree/util.ih:10
// PathQueue::path: the path's index within its pass (bits 0..26) and, above it, the rays the path has traced on
// earlier bounces (at most 3 per bounce: closest hit + up to two occlusion rays); k_shade adds them to
// radiance[path].w when the path ends.
constexpr int PATH_ID_BITS = 27;
constexpr uint32_t PATH_ID_MASK = (1u << PATH_ID_BITS) - 1u;
static_assert(3 * MAX_PATH_DEPTH < (1 << (32 - PATH_ID_BITS)), "ray count of a path must fit above its index");
constexpr float RAY_TFAR = 1e20f;     // set_ray_hit, backends/embree/util.ih:118

} // namespace crt
