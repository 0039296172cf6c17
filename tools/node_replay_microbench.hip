// node_replay_microbench.hip -- would a 32-byte, two-request BVH node pay on the REAL access pattern? (round-5 review, item 3 (ii))
//
// tools/node_bytes_microbench.hip priced node records of 64 / 48 / 32 bytes on chains of uniformly random indices: 2.06 against
// 1.44 CU-cycles per visit L2-resident with all lanes on, the gap gone at half the lanes. This replays RECORDED visit sequences
// instead -- bounce-1 / bounce-2 closest-hit rays and occlusion rays of a workload, walked by the oracle's walker of the product's
// own packed tree (tools/make_visit_trace.py -> build/trace_<W>.bin) -- over buffers sized like the real arrays: every lane owns a
// ray, walks its sequence of node and leaf-slot visits as a dependent chain (the next step waits for the data of this one), takes
// the next ray when it is done; a node visit fetches 3 x dwordx4 of a 64-byte record (today) or 2 x dwordx4 of a 32-byte record
// (the candidate; the node array is then half as large), a leaf-slot visit 4 x dwordx4 in both. `lanes`: share of the lanes that
// take part in a step (the production kernels run at ~42 of 64).
// Build: hipcc --offload-arch=gfx950 -O3 -o build/node_replay_microbench tools/node_replay_microbench.hip
// Run:   build/node_replay_microbench build/trace_C4.bin [passes]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                                          \
    do {                                                                                                               \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess) {                                                                                        \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                                  \
            exit(1);                                                                                                   \
        }                                                                                                              \
    } while (0)

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int NODE_LOADS, int NODE_STRIDE16>
__global__ __launch_bounds__(256) void k_replay(const u4 *nodes, const u4 *slots, const uint32_t *offsets, const uint32_t *visits,
                                                uint32_t n_rays, int passes, uint32_t active_thresh, uint32_t *out,
                                                unsigned long long *n_done)
{
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x, total = gridDim.x * 256u;
    uint32_t acc = 0, rnd = gid * 2654435761u + 12345u;
    unsigned long long done = 0;
    for (int pass = 0; pass < passes; ++pass) {
        const uint32_t ray = (uint32_t)(((unsigned long long)pass * total + gid) % n_rays);
        uint32_t pos = offsets[ray];
        const uint32_t end = offsets[ray + 1];
        while (__ballot(pos < end) != 0ull) { // the wave stays together, like the production kernels' step loop
            rnd = rnd * 1664525u + 1013904223u;
            const bool want = pos < end && (rnd >> 8) < active_thresh;
            if (want) {
                const uint32_t v = visits[pos];
                uint32_t fold;
                if (v >> 31) {
                    const u4 *p = slots + (size_t)4 * (v & 0x7fffffffu);
                    const u4 x = p[0] ^ p[1] * 3u ^ p[2] * 5u ^ p[3] * 7u;
                    fold = x.x ^ x.y ^ x.z ^ x.w;
                } else {
                    const u4 *p = nodes + (size_t)NODE_STRIDE16 * v;
                    u4 x = p[0];
#pragma unroll
                    for (int k = 1; k < NODE_LOADS; ++k) {
                        x ^= p[k] * (uint32_t)(2 * k + 1);
                    }
                    fold = x.x ^ x.y ^ x.z ^ x.w;
                }
                acc += fold;
                pos += 1u + (fold == 0x9e3779b9u ? 1u : 0u); // the next visit depends on this one's data
                ++done;
            }
        }
    }
    out[gid] = acc;
    if (done) {
        atomicAdd(n_done, done);
    }
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: %s trace.bin [passes]\n", argv[0]);
        return 2;
    }
    FILE *f = fopen(argv[1], "rb");
    if (!f) {
        perror(argv[1]);
        return 1;
    }
    uint32_t head[4];
    if (fread(head, 4, 4, f) != 4) {
        return 1;
    }
    const uint32_t n_rays = head[0], n_visits = head[1], n_nodes = head[2], n_slots = head[3];
    std::vector<uint32_t> off(n_rays + 1), vis(n_visits);
    if (fread(off.data(), 4, off.size(), f) != off.size() || fread(vis.data(), 4, vis.size(), f) != vis.size()) {
        return 1;
    }
    fclose(f);
    unsigned long long node_visits = 0;
    for (uint32_t v : vis) {
        node_visits += (v >> 31) == 0u;
    }
    const int passes = argc > 2 ? atoi(argv[2]) : 8;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 7; // 7 waves per SIMD, like the traversal kernels
    u4 *d_nodes, *d_slots;
    uint32_t *d_off, *d_vis, *d_out;
    unsigned long long *d_done;
    CK(hipMalloc(&d_nodes, (size_t)n_nodes * 64));
    CK(hipMalloc(&d_slots, (size_t)n_slots * 64));
    CK(hipMemset(d_nodes, 0x5a, (size_t)n_nodes * 64));
    CK(hipMemset(d_slots, 0xa5, (size_t)n_slots * 64));
    CK(hipMalloc(&d_off, off.size() * 4));
    CK(hipMalloc(&d_vis, vis.size() * 4));
    CK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
    CK(hipMalloc(&d_done, 8));
    CK(hipMemcpy(d_off, off.data(), off.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_vis, vis.data(), vis.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("%s: %u rays, %.2f node + %.2f leaf-slot visits per ray; tree %u nodes (%.0f MB at 64 B, %.0f MB at 32 B) + %u leaf slots (%.0f MB); "
           "%d passes of %d lanes; CU-cycles @2.3 GHz\n",
           argv[1], n_rays, (double)node_visits / n_rays, (double)(n_visits - node_visits) / n_rays, n_nodes, n_nodes * 64e-6, n_nodes * 32e-6,
           n_slots, n_slots * 64e-6, passes, blocks * 256);
    for (uint32_t act : {100u, 66u, 50u}) {
        const uint32_t thresh = (uint32_t)(0x1000000ull * act / 100u);
        double ms_mode[2] = {0, 0}, visits_mode[2] = {0, 0}, nodes_share = (double)node_visits / n_visits;
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e30f;
            unsigned long long done = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(d_done, 0, 8));
                CK(hipEventRecord(e0));
                if (mode == 0) {
                    k_replay<3, 4><<<blocks, 256>>>(d_nodes, d_slots, d_off, d_vis, n_rays, passes, thresh, d_out, d_done);
                } else {
                    k_replay<2, 2><<<blocks, 256>>>(d_nodes, d_slots, d_off, d_vis, n_rays, passes, thresh, d_out, d_done);
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
                CK(hipMemcpy(&done, d_done, 8, hipMemcpyDeviceToHost));
            }
            ms_mode[mode] = best;
            visits_mode[mode] = (double)done;
        }
        const double c3 = ms_mode[0] * 1e-3 * 2.3e9 * cus / visits_mode[0], c2 = ms_mode[1] * 1e-3 * 2.3e9 * cus / visits_mode[1];
        printf("  %3u%% of the lanes per step: 48 B of a 64-byte record (3 requests) %.3f ms = %.2f cycles per visit | 32-byte record (2 requests) %.3f ms = %.2f "
               "cycles per visit | %+.1f %%; per NODE visit %.2f -> %.2f cycles if the leaf slots cost the same\n",
               act, ms_mode[0], c3, ms_mode[1], c2, (ms_mode[1] / ms_mode[0] - 1.0) * 100.0, c3, c3 - (c3 - c2) / nodes_share);
    }
    return 0;
}
