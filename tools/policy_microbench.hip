// policy_microbench.hip — what does a dependent random 64-byte node visit cost on gfx950 under each cache policy,
// and what would a wave-coherent ("packet") visit cost?
// Follow-up to line_microbench.hip (round 2: a visit costs ~2.9 CU-cycles while its line is in L2 whatever the
// number of 16-byte requests). Each lane walks a dependent chain of random 64-byte records (4 x dwordx4):
//   plain    : global_load_dwordx4
//   nt       : global_load_dwordx4 nt
//   b_plain  : buffer_load_dwordx4 (raw buffer, offen)
//   b_sc0 / b_nt / b_sc1 / b_sc0sc1 / b_sc1nt : the same with the cache-policy bits
//   uni_v    : every lane of a wave visits the SAME record (vector loads): bounce-0 rays in the per-lane kernel
//   uni_s    : the same record fetched once per wave by the scalar unit (s_load_dwordx16): the packet mode
//   grp16    : lanes in groups of 16 share a record (4 distinct lines per wave instruction)
// Build: hipcc --offload-arch=gfx950 -O3 -o build/policy_microbench tools/policy_microbench.hip
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

template <int AUX> __device__ inline uint32_t visit_buffer(__amdgpu_buffer_rsrc_t r, uint32_t byte_off)
{
    const u4 a = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, AUX);
    const u4 b = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off + 16, 0, AUX);
    const u4 c = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off + 32, 0, AUX);
    const u4 d = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off + 48, 0, AUX);
    return fold(a ^ (b * 3u) ^ (c * 5u) ^ (d * 7u));
}

// `mask` selects a 64-byte record; the buffer holds (mask + 1) records
template <int MODE>
__global__ __launch_bounds__(256) void k_walk(const u4 *buf, uint32_t mask, int steps, uint32_t active_thresh, uint32_t *out)
{
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)buf, 0, 0x7fffffff, 0x00027000);
    uint32_t idx = mix(blockIdx.x * 256u + threadIdx.x);
    if (MODE == 8 || MODE == 9) {
        idx = mix(blockIdx.x * 4u + threadIdx.x / 64u); // wave-uniform chain
    } else if (MODE == 10) {
        idx = mix(blockIdx.x * 16u + threadIdx.x / 16u);
    }
    uint32_t acc = 0, rnd = mix(blockIdx.x * 256u + threadIdx.x + 77u);
    for (int s = 0; s < steps; ++s) {
        rnd = rnd * 1664525u + 1013904223u;
        const bool want = (rnd >> 8) < active_thresh;
        uint32_t v = 0;
        if (MODE == 9) {
            // one scalar fetch per wave, whatever the lanes do with it
            const uint32_t u = __builtin_amdgcn_readfirstlane(idx);
            const __attribute__((address_space(4))) u4 *p = (const __attribute__((address_space(4))) u4 *)(buf + 4 * (size_t)(u & mask));
            const u4 a = p[0], b = p[1], c = p[2], d = p[3];
            v = fold(a ^ (b * 3u) ^ (c * 5u) ^ (d * 7u));
            acc += want ? v : 0u;
            idx = mix(u + v + (uint32_t)s);
            continue;
        }
        if (want || MODE == 8 || MODE == 10) { // the coherent modes keep their groups in step
            const u4 *p = buf + 4 * (size_t)(idx & mask);
            const uint32_t off = (idx & mask) * 64u;
            if (MODE == 0 || MODE == 8 || MODE == 10) {
                v = fold(p[0] ^ (p[1] * 3u) ^ (p[2] * 5u) ^ (p[3] * 7u));
            } else if (MODE == 1) {
                const u4 a = __builtin_nontemporal_load(p), b = __builtin_nontemporal_load(p + 1), c = __builtin_nontemporal_load(p + 2),
                         d = __builtin_nontemporal_load(p + 3);
                v = fold(a ^ (b * 3u) ^ (c * 5u) ^ (d * 7u));
            } else if (MODE == 2) {
                v = visit_buffer<0>(rsrc, off);
            } else if (MODE == 3) {
                v = visit_buffer<1>(rsrc, off);
            } else if (MODE == 4) {
                v = visit_buffer<2>(rsrc, off);
            } else if (MODE == 5) {
                v = visit_buffer<16>(rsrc, off);
            } else if (MODE == 6) {
                v = visit_buffer<17>(rsrc, off);
            } else if (MODE == 7) {
                v = visit_buffer<18>(rsrc, off);
            }
            acc += v;
            idx = mix(idx + v + (uint32_t)s);
        }
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

template <int MODE> float run(const u4 *d, uint32_t mask, int steps, uint32_t thresh, uint32_t *out, int blocks, hipEvent_t e0, hipEvent_t e1)
{
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        k_walk<MODE><<<blocks, 256>>>(d, mask, steps, thresh, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    return ms;
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
    const char *names[11] = {"plain", "nt", "b_plain", "b_sc0", "b_nt", "b_sc1", "b_sc0sc1", "b_sc1nt", "uni_v", "uni_s", "grp16"};
    for (int log_recs : {14, 18, 21, 24}) { // 1 MB (L2), 16 MB, 128 MB (Infinity Cache), 1 GB (HBM)
        const size_t n = (size_t)1 << log_recs;
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
            printf("%5zu MB, %3u%% of the lanes active:", n * 64 >> 20, act);
            for (int mode = 0; mode < 11; ++mode) {
                float ms = 0;
                const uint32_t mask = (uint32_t)n - 1;
                switch (mode) {
                case 0: ms = run<0>(d, mask, steps, thresh, out, blocks, e0, e1); break;
                case 1: ms = run<1>(d, mask, steps, thresh, out, blocks, e0, e1); break;
                case 2: ms = run<2>(d, mask, steps, thresh, out, blocks, e0, e1); break;
                case 3: ms = run<3>(d, mask, steps, thresh, out, blocks, e0, e1); break;
                case 4: ms = run<4>(d, mask, steps, thresh, out, blocks, e0, e1); break;
                case 5: ms = run<5>(d, mask, steps, thresh, out, blocks, e0, e1); break;
                case 6: ms = run<6>(d, mask, steps, thresh, out, blocks, e0, e1); break;
                case 7: ms = run<7>(d, mask, steps, thresh, out, blocks, e0, e1); break;
                case 8: ms = run<8>(d, mask, steps, thresh, out, blocks, e0, e1); break;
                case 9: ms = run<9>(d, mask, steps, thresh, out, blocks, e0, e1); break;
                default: ms = run<10>(d, mask, steps, thresh, out, blocks, e0, e1); break;
                }
                // lane-visits: in the coherent modes every lane takes part in every step
                const double frac = (mode >= 8) ? 1.0 : act / 100.0;
                const double visits = (double)blocks * 256 * steps * frac;
                printf("  %s %.2f", names[mode], ms * 1e-3 * 2.3e9 * cus / visits);
            }
            printf("   (CU-cycles per lane-visit @2.3 GHz)\n");
        }
        CK(hipFree(d));
    }
    return 0;
}
