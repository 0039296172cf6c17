// node_bytes_microbench.hip — what would a SMALLER BVH node buy on gfx950's vector-memory front end?
//
// tools/ta_quad_microbench.hip showed that one divergent 16-byte load per lane costs 1.35 CU-cycles per lane and four of
// them (a 64-byte QNode, what traverse.h fetches) 2.8: the cost of an incoherent node visit follows the number of
// dwordx4 instructions, not only the number of cache lines. This walks the same dependent chain of random records with
// records of 64 B (4 loads), 48 B (3 loads; packed at a 48-byte stride, and padded to a 64-byte stride) and 32 B
// (2 loads), at four working-set sizes, to price a 48-byte node layout before any of it is written.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/node_bytes_microbench tools/node_bytes_microbench.hip
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

__device__ inline uint32_t mix(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

// LOADS dwordx4 per visit from a record at index * STRIDE16 sixteen-byte units
template <int LOADS, int STRIDE16>
__global__ __launch_bounds__(256) void k_walk(const u4 *nodes, uint32_t mask, int steps, uint32_t active_thresh, uint32_t *out)
{
    uint32_t idx = mix(blockIdx.x * 256u + threadIdx.x) & mask;
    uint32_t acc = 0, rnd = mix(idx + 77u);
    for (int s = 0; s < steps; ++s) {
        rnd = rnd * 1664525u + 1013904223u;
        const bool want = (rnd >> 8) < active_thresh;
        if (want) {
            const u4 *p = nodes + (size_t)STRIDE16 * idx;
            u4 x = p[0];
#pragma unroll
            for (int k = 1; k < LOADS; ++k) {
                x ^= p[k] * (uint32_t)(2 * k + 1);
            }
            const uint32_t v = x.x ^ (x.y * 11u) ^ (x.z * 13u) ^ (x.w * 17u);
            acc += v;
            idx = mix(idx + v + (uint32_t)s) & mask;
        }
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

int main()
{
    const int steps = 1000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 6; // 6 waves per SIMD, like the traversal kernels
    uint32_t *out;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("CU-cycles per lane-visit @2.3 GHz; records = nodes of the tree (2^21 = C4's node count)\n");
    for (int log_nodes : {14, 18, 21, 24}) {
        const size_t n = (size_t)1 << log_nodes;
        std::vector<u4> h(4 * n);
        uint32_t s = 12345;
        for (auto &v : h) {
            s = s * 1664525u + 1013904223u;
            v.x = s;
            s = s * 1664525u + 1013904223u;
            v.y = s;
            v.z = s >> 3;
            v.w = s >> 7;
        }
        u4 *d;
        CK(hipMalloc(&d, h.size() * sizeof(u4)));
        CK(hipMemcpy(d, h.data(), h.size() * sizeof(u4), hipMemcpyHostToDevice));
        for (uint32_t act : {100u, 50u}) {
            const uint32_t thresh = act == 100u ? 0x1000000u : 0x800000u;
            const double visits = (double)blocks * 256 * steps * (act / 100.0);
            printf("  2^%d records, %3u%% of the lanes active:", log_nodes, act);
            for (int mode = 0; mode < 6; ++mode) {
                float ms = 0.f;
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(e0));
                    switch (mode) {
                    case 0: k_walk<4, 4><<<blocks, 256>>>(d, (uint32_t)n - 1, steps, thresh, out); break;
                    case 1: k_walk<3, 4><<<blocks, 256>>>(d, (uint32_t)n - 1, steps, thresh, out); break;
                    case 2: k_walk<3, 3><<<blocks, 256>>>(d, (uint32_t)n - 1, steps, thresh, out); break;
                    case 3: k_walk<2, 4><<<blocks, 256>>>(d, (uint32_t)n - 1, steps, thresh, out); break;
                    case 4: k_walk<2, 2><<<blocks, 256>>>(d, (uint32_t)n - 1, steps, thresh, out); break;
                    default: k_walk<1, 4><<<blocks, 256>>>(d, (uint32_t)n - 1, steps, thresh, out); break;
                    }
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                }
                static const char *names[6] = {"64B", "48B/stride64", "48B/stride48", "32B/stride64", "32B/stride32", "16B/stride64"};
                printf("  %s %.2f", names[mode], ms * 1e-3 * 2.3e9 * cus / visits);
            }
            printf("\n");
        }
        CK(hipFree(d));
    }
    return 0;
}
