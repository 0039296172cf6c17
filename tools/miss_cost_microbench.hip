// miss_cost_microbench.hip -- what does an L2 MISS of a divergent node / leaf-slot visit cost the CU, as a function of how far away
// the data is? (round 6: gives bench.py's front-end ceiling a middle regime between "hits L2" and "comes from HBM".)
//
// One configuration per run, so that a rocprofv3 counter pass around it is unambiguous (tools/miss_cost_session.sh):
//     build/miss_cost_microbench LOG2_RECORDS LOADS LANES_PERCENT
// walks dependent chains of uniformly random 64-byte records (LOADS = 3: the 48 bytes of a packed node; 4: a leaf slot), 7 waves per
// SIMD, and prints CU-cycles per lane-visit. Under `rocprofv3 --pmc TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum`
// the same launch yields the L2 hit rate and the AVERAGE LATENCY of the L2's memory-side reads (LEVEL / RDREQ, in L2 clocks): short
// when the Infinity Cache serves them, long from HBM. The working set (records x 64 B) moves from inside L2 (1 MB) through the
// Infinity Cache's reach (16 ... 128 MB) to HBM (1 GB).
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

template <int LOADS>
__global__ __launch_bounds__(256) void k_miss_walk(const u4 *recs, uint32_t mask, int steps, uint32_t active_thresh, uint32_t *out)
{
    uint32_t idx = mix(blockIdx.x * 256u + threadIdx.x) & mask;
    uint32_t acc = 0, rnd = mix(idx + 77u);
    for (int s = 0; s < steps; ++s) {
        rnd = rnd * 1664525u + 1013904223u;
        if ((rnd >> 8) < active_thresh) {
            const u4 *p = recs + (size_t)4 * idx;
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

int main(int argc, char **argv)
{
    if (argc < 4) {
        fprintf(stderr, "usage: %s log2_records loads(3|4) lanes_percent\n", argv[0]);
        return 2;
    }
    const int log_n = atoi(argv[1]), loads = atoi(argv[2]), act = atoi(argv[3]);
    const int steps = 600;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, blocks = cus * 7;
    const size_t n = (size_t)1 << log_n;
    std::vector<u4> h(4 * n);
    uint32_t s = 12345;
    for (auto &v : h) {
        s = s * 1664525u + 1013904223u;
        v.x = s;
        v.y = s * 2654435761u;
        v.z = s >> 3;
        v.w = s >> 7;
    }
    u4 *d;
    uint32_t *out;
    CK(hipMalloc(&d, h.size() * sizeof(u4)));
    CK(hipMemcpy(d, h.data(), h.size() * sizeof(u4), hipMemcpyHostToDevice));
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint32_t thresh = (uint32_t)(0x1000000ull * act / 100);
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) { // the first launch warms the caches up; the second is the one reported (and the one to read counters of)
        CK(hipEventRecord(e0));
        if (loads == 3) {
            k_miss_walk<3><<<blocks, 256>>>(d, (uint32_t)n - 1, steps, thresh, out);
        } else {
            k_miss_walk<4><<<blocks, 256>>>(d, (uint32_t)n - 1, steps, thresh, out);
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double visits = (double)blocks * 256 * steps * (act / 100.0);
    printf("records 2^%d = %.0f MB, %d loads, %d%% lanes: %.3f ms, %.2f CU-cycles per lane-visit @2.3 GHz\n", log_n, n * 64e-6, loads, act, ms,
           ms * 1e-3 * 2.3e9 * cus / visits);
    return 0;
}
