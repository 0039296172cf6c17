// ta_microbench.hip — how does the per-CU vector-memory front end (TA) price divergent 16-byte
// lane requests on gfx950? Each lane walks a dependent chain of random 32-byte "nodes" (the QNode
// access pattern of incoherent BVH traversal) in one of three ways:
//   A  own:   each lane issues two dwordx4 loads for its own node (what traverse.h does today)
//   B  pair:  lanes 2k / 2k+1 load the two halves of node(2k), then the two halves of node(2k+1):
//             same two instructions, but each lane pair touches ONE cache line per instruction;
//             halves are exchanged with DPP quad_perm
//   C  half:  each lane issues ONE dwordx4 load (lower bound: half the requests)
// Build: hipcc --offload-arch=gfx950 -O3 -o ta_microbench tools/ta_microbench.hip
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

__device__ inline uint32_t mix(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

__device__ inline uint32_t swap_pair(uint32_t v)
{
    // quad_perm [1,0,3,2]
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_walk(const uint4 *nodes, uint32_t mask, int steps, uint32_t *out)
{
    uint32_t idx = mix(blockIdx.x * 256u + threadIdx.x) & mask;
    uint32_t acc = 0;
    const bool odd = threadIdx.x & 1;
    for (int s = 0; s < steps; ++s) {
        uint4 h0, h1;
        if (MODE == 0) {
            h0 = nodes[2 * (size_t)idx];
            h1 = nodes[2 * (size_t)idx + 1];
        } else if (MODE == 1) {
            const uint32_t other = swap_pair(idx);
            const uint32_t n_even = odd ? other : idx, n_odd = odd ? idx : other;
            const uint4 r1 = nodes[2 * (size_t)n_even + (odd ? 1 : 0)];
            const uint4 r2 = nodes[2 * (size_t)n_odd + (odd ? 1 : 0)];
            // even lane: own H0 = r1, partner holds own H1 in its r1; odd lane: own H1 = r2, partner holds H0 in its r2
            const uint4 mine = odd ? r2 : r1;
            const uint4 give = odd ? r1 : r2;
            uint4 got;
            got.x = swap_pair(give.x);
            got.y = swap_pair(give.y);
            got.z = swap_pair(give.z);
            got.w = swap_pair(give.w);
            h0 = mine; // odd lanes see the halves swapped; fine for a symmetric node layout
            h1 = got;
        } else {
            h0 = nodes[2 * (size_t)idx];
            h1 = h0;
        }
        const uint32_t v = h0.x ^ h0.w ^ h1.y ^ h1.z;
        acc += v;
        idx = mix(idx + v + s) & mask;
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

int main(int argc, char **argv)
{
    const int steps = 2000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    for (int log_nodes : {14, 17, 20, 24}) { // 512 KB (L2-resident everywhere), 4 MB, 32 MB, 512 MB
        const size_t n = (size_t)1 << log_nodes;
        std::vector<uint4> h(2 * n);
        uint32_t s = 12345;
        for (auto &v : h) {
            s = s * 1664525u + 1013904223u;
            v.x = s;
            s = s * 1664525u + 1013904223u;
            v.y = s;
            v.z = s >> 3;
            v.w = s >> 7;
        }
        uint4 *d;
        CK(hipMalloc(&d, h.size() * sizeof(uint4)));
        CK(hipMemcpy(d, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice));
        for (int bpc : {4, 7}) {
            const int blocks = cus * bpc;
            uint32_t *out;
            CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            float ms[3];
            for (int mode = 0; mode < 3; ++mode) {
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(e0));
                    if (mode == 0) {
                        k_walk<0><<<blocks, 256>>>(d, (uint32_t)n - 1, steps, out);
                    } else if (mode == 1) {
                        k_walk<1><<<blocks, 256>>>(d, (uint32_t)n - 1, steps, out);
                    } else {
                        k_walk<2><<<blocks, 256>>>(d, (uint32_t)n - 1, steps, out);
                    }
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms[mode], e0, e1));
                }
            }
            const double lane_steps = (double)blocks * 256 * steps;
            printf("nodes 2^%d (%zu KB) blocks/CU %d: own %.1f G lane-steps/s (%.2f cyc/CU/lane-step @2.4GHz), pair %.1f, "
                   "half %.1f\n",
                   log_nodes, n * 32 / 1024, bpc, lane_steps / ms[0] * 1e-6,
                   ms[0] * 1e-3 * 2.4e9 * cus / lane_steps, lane_steps / ms[1] * 1e-6, lane_steps / ms[2] * 1e-6);
            CK(hipFree(out));
        }
        CK(hipFree(d));
    }
    return 0;
}
