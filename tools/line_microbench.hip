// line_microbench.hip — is an incoherent node visit on gfx950 paid per 16-byte REQUEST or per cache LINE?
// (Follow-up to ta_quad_microbench.hip after the product experiment "request only the used child quarters"
// saved 14 % of the node requests and no time.) Each lane walks a dependent chain of random records:
//   n32    : 32-byte record, 2 x dwordx4
//   n64    : 64-byte record, 4 x dwordx4 (the QNode of traverse.h)
//   n128   : 128-byte record aligned to 128, 8 x dwordx4 (what an 8-wide node would be)
//   pair   : a 64-byte record, then -- dependent on its contents -- the OTHER half of the same 128-byte line
//            (a parent followed by the child it is laid out next to); counted as two visits
//   two64  : two dependent random 64-byte records per step (what `pair` replaces); counted as two visits
//   tri48  : a 48-byte record at a multiple of 48 bytes (the TriRec of traverse.h: may straddle a 64- / 128-byte boundary)
//   tri64  : the same 48 bytes at the start of a 64-byte slot
//   leaf48 : two consecutive 48-byte records at a multiple of 48 bytes (a two-triangle leaf), 6 x dwordx4
//   leaf64 : two 48-byte records in two 64-byte slots of one 128-byte line, 6 x dwordx4
// Build: hipcc --offload-arch=gfx950 -O3 -o build/line_microbench tools/line_microbench.hip
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
__device__ inline uint32_t fold(u4 x) { return x.x ^ (x.y * 11u) ^ (x.z * 13u) ^ (x.w * 17u); }

// `mask` selects a 128-byte line; the buffer holds (mask + 1) lines
template <int MODE>
__global__ __launch_bounds__(256) void k_walk(const u4 *buf, uint32_t mask, int steps, uint32_t active_thresh, uint32_t *out)
{
    uint32_t idx = mix(blockIdx.x * 256u + threadIdx.x);
    uint32_t acc = 0, rnd = mix(idx + 77u);
    for (int s = 0; s < steps; ++s) {
        rnd = rnd * 1664525u + 1013904223u;
        const bool want = (rnd >> 8) < active_thresh;
        if (want) {
            const u4 *line = buf + 8 * (size_t)(idx & mask);
            const uint32_t half = (idx >> 31) * 4u; // which 64-byte half of the line
            uint32_t v;
            if (MODE == 0) {
                const u4 *p = line + half + ((idx >> 30) & 1u) * 2u;
                v = fold(p[0] ^ (p[1] * 3u));
            } else if (MODE == 1) {
                const u4 *p = line + half;
                v = fold(p[0] ^ (p[1] * 3u) ^ (p[2] * 5u) ^ (p[3] * 7u));
            } else if (MODE == 2) {
                const u4 *p = line;
                v = fold(p[0] ^ (p[1] * 3u) ^ (p[2] * 5u) ^ (p[3] * 7u) ^ (p[4] * 9u) ^ (p[5] * 19u) ^ (p[6] * 23u) ^ (p[7] * 29u));
            } else if (MODE == 3) {
                const u4 *p = line + half;
                const uint32_t a = fold(p[0] ^ (p[1] * 3u) ^ (p[2] * 5u) ^ (p[3] * 7u));
                const u4 *q = line + (half ^ 4u) + (a & 0u); // address depends on the first record's contents
                v = a ^ fold(q[0] ^ (q[1] * 3u) ^ (q[2] * 5u) ^ (q[3] * 7u));
            } else if (MODE == 5) {
                const u4 *p = buf + 3 * (size_t)((idx & ((mask << 1) | 1u)) ); // 48-byte stride over 3/4 of the buffer
                v = fold(p[0] ^ (p[1] * 3u) ^ (p[2] * 5u));
            } else if (MODE == 6) {
                const u4 *p = line + half;
                v = fold(p[0] ^ (p[1] * 3u) ^ (p[2] * 5u));
            } else if (MODE == 7) {
                const u4 *p = buf + 3 * (size_t)((idx & ((mask << 1) | 1u)));
                v = fold(p[0] ^ (p[1] * 3u) ^ (p[2] * 5u) ^ (p[3] * 7u) ^ (p[4] * 9u) ^ (p[5] * 19u));
            } else if (MODE == 8) {
                const u4 *p = line;
                v = fold(p[0] ^ (p[1] * 3u) ^ (p[2] * 5u) ^ (p[4] * 7u) ^ (p[5] * 9u) ^ (p[6] * 19u));
            } else {
                const u4 *p = line + half;
                const uint32_t a = fold(p[0] ^ (p[1] * 3u) ^ (p[2] * 5u) ^ (p[3] * 7u));
                const uint32_t j = mix(idx + a);
                const u4 *q = buf + 8 * (size_t)(j & mask) + (j >> 31) * 4u;
                v = a ^ fold(q[0] ^ (q[1] * 3u) ^ (q[2] * 5u) ^ (q[3] * 7u));
            }
            acc += v;
            idx = mix(idx + v + (uint32_t)s);
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
    const int blocks = cus * 6; // 6 blocks of 256 per CU, like the traversal kernels
    uint32_t *out;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char *names[9] = {"n32", "n64", "n128", "pair", "two64", "tri48", "tri64", "leaf48", "leaf64"};
    const double visits_per_step[9] = {1, 1, 1, 2, 2, 1, 1, 1, 1};
    for (int log_lines : {13, 17, 20, 23}) { // 1 MB (L2), 16 MB, 128 MB (Infinity Cache), 1 GB (HBM)
        const size_t n = (size_t)1 << log_lines;
        std::vector<u4> h(8 * n);
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
            printf("%5zu MB, %3u%% of the lanes active:", n * 128 >> 20, act);
            for (int mode = 0; mode < 9; ++mode) {
                float ms = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(e0));
                    switch (mode) {
                    case 0: k_walk<0><<<blocks, 256>>>(d, (uint32_t)n - 1, steps, thresh, out); break;
                    case 1: k_walk<1><<<blocks, 256>>>(d, (uint32_t)n - 1, steps, thresh, out); break;
                    case 2: k_walk<2><<<blocks, 256>>>(d, (uint32_t)n - 1, steps, thresh, out); break;
                    case 3: k_walk<3><<<blocks, 256>>>(d, (uint32_t)n - 1, steps, thresh, out); break;
                    case 4: k_walk<4><<<blocks, 256>>>(d, (uint32_t)n - 1, steps, thresh, out); break;
                    case 5: k_walk<5><<<blocks, 256>>>(d, (uint32_t)n - 1, steps, thresh, out); break;
                    case 6: k_walk<6><<<blocks, 256>>>(d, (uint32_t)n - 1, steps, thresh, out); break;
                    case 7: k_walk<7><<<blocks, 256>>>(d, (uint32_t)n - 1, steps, thresh, out); break;
                    default: k_walk<8><<<blocks, 256>>>(d, (uint32_t)n - 1, steps, thresh, out); break;
                    }
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                }
                const double visits = (double)blocks * 256 * steps * (act / 100.0) * visits_per_step[mode];
                printf("  %s %.2f", names[mode], ms * 1e-3 * 2.3e9 * cus / visits);
            }
            printf("   (CU-cycles per visit @2.3 GHz)\n");
        }
        CK(hipFree(d));
    }
    return 0;
}
