// ta_quad_microbench.hip — what does one incoherent 64-byte BVH4 node visit cost the per-CU vector-memory
// front end on gfx950, and does fetching it by QUAD (4 lanes x 16 B of ONE node per instruction, then a
// 4x4 register transpose inside the quad) buy anything?
//
// Each lane walks a dependent chain of random 64-byte nodes (the QNode access pattern of traverse.h):
//   own   : 4 x global_load_dwordx4 per lane on its own node (what traverse.h does) -- every lane touches its
//           own cache line in each of the four instructions
//   quad  : instruction j fetches the node of quad-lane j, lane q of the quad reading quarter q: one line per
//           quad per instruction; then the 16 dwords are transposed with DPP (2 butterfly stages)
//   one   : a single dwordx4 per lane (lower bound: a quarter of the requests, no transpose)
// `active` emulates divergence: only that fraction of the lanes (random per step) wants a node.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/ta_quad_microbench tools/ta_quad_microbench.hip
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

template <int CTRL> __device__ inline uint32_t dpp(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
// quad_perm encodings: [a,b,c,d] -> a | b<<2 | c<<4 | d<<6
constexpr int QP_XOR1 = 1 | (0 << 2) | (3 << 4) | (2 << 6); // [1,0,3,2]
constexpr int QP_XOR2 = 2 | (3 << 2) | (0 << 4) | (1 << 6); // [2,3,0,1]
template <int J> constexpr int qp_bcast() { return J | (J << 2) | (J << 4) | (J << 6); }

// 4x4 transpose inside each quad: in: k[r] = quarter (lane & 3) of the node of quad-lane r;
// out: k[q] = quarter q of this lane's own node
__device__ inline void quad_transpose(u4 k[4], uint32_t lane)
{
    const bool b0 = lane & 1u, b1 = lane & 2u;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        // stage 1: swap bit 0 of the lane with bit 0 of the register index
        uint32_t a0 = k[0][c], a1 = k[1][c], a2 = k[2][c], a3 = k[3][c];
        const uint32_t p0 = dpp<QP_XOR1>(a0), p1 = dpp<QP_XOR1>(a1), p2 = dpp<QP_XOR1>(a2), p3 = dpp<QP_XOR1>(a3);
        uint32_t n0 = b0 ? p1 : a0, n1 = b0 ? a1 : p0, n2 = b0 ? p3 : a2, n3 = b0 ? a3 : p2;
        // stage 2: bit 1
        const uint32_t q0 = dpp<QP_XOR2>(n0), q1 = dpp<QP_XOR2>(n1), q2 = dpp<QP_XOR2>(n2), q3 = dpp<QP_XOR2>(n3);
        k[0][c] = b1 ? q2 : n0;
        k[2][c] = b1 ? n2 : q0;
        k[1][c] = b1 ? q3 : n1;
        k[3][c] = b1 ? n3 : q1;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_walk(const u4 *nodes, uint32_t mask, int steps, uint32_t active_thresh,
                                              uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t idx = mix(blockIdx.x * 256u + threadIdx.x) & mask;
    uint32_t acc = 0, rnd = mix(idx + 77u);
    for (int s = 0; s < steps; ++s) {
        rnd = rnd * 1664525u + 1013904223u;
        const bool want = (rnd >> 8) < active_thresh; // this lane visits a node in this step
        u4 k[4];
        if (MODE == 0) {
            if (want) {
                const u4 *p = nodes + 4 * (size_t)idx;
                k[0] = p[0];
                k[1] = p[1];
                k[2] = p[2];
                k[3] = p[3];
            }
        } else if (MODE == 1) {
            const int32_t mine = want ? (int32_t)idx : -1;
            const uint32_t q = lane & 3u;
            const int32_t n0 = (int32_t)dpp<qp_bcast<0>()>((uint32_t)mine), n1 = (int32_t)dpp<qp_bcast<1>()>((uint32_t)mine);
            const int32_t n2 = (int32_t)dpp<qp_bcast<2>()>((uint32_t)mine), n3 = (int32_t)dpp<qp_bcast<3>()>((uint32_t)mine);
            k[0] = k[1] = k[2] = k[3] = (u4){0u, 0u, 0u, 0u};
            if (n0 >= 0) {
                k[0] = nodes[4 * (size_t)n0 + q];
            }
            if (n1 >= 0) {
                k[1] = nodes[4 * (size_t)n1 + q];
            }
            if (n2 >= 0) {
                k[2] = nodes[4 * (size_t)n2 + q];
            }
            if (n3 >= 0) {
                k[3] = nodes[4 * (size_t)n3 + q];
            }
            quad_transpose(k, lane);
        } else {
            if (want) {
                k[0] = nodes[4 * (size_t)idx];
                k[1] = k[0] + 1u;
                k[2] = k[0] + 2u;
                k[3] = k[0] + 3u;
            }
        }
        if (want) {
            // every dword is consumed, like the four box tests + references of a real node visit
            const u4 x = k[0] ^ (k[1] * 3u) ^ (k[2] * 5u) ^ (k[3] * 7u);
            const uint32_t v = x.x ^ (x.y * 11u) ^ (x.z * 13u) ^ (x.w * 17u);
            acc += v;
            idx = mix(idx + v + (uint32_t)s) & mask;
        }
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

int main(int argc, char **argv)
{
    const int steps = 1000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int bpc = 7;
    const int blocks = cus * bpc;
    uint32_t *out;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int log_nodes : {14, 18, 21, 24}) { // 1 MB (L2), 16 MB, 128 MB (Infinity Cache), 1 GB (HBM)
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
            float ms[3] = {0, 0, 0};
            uint32_t chk[3] = {0, 0, 0};
            for (int mode = 0; mode < 3; ++mode) {
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(e0));
                    if (mode == 0) {
                        k_walk<0><<<blocks, 256>>>(d, (uint32_t)n - 1, steps, thresh, out);
                    } else if (mode == 1) {
                        k_walk<1><<<blocks, 256>>>(d, (uint32_t)n - 1, steps, thresh, out);
                    } else {
                        k_walk<2><<<blocks, 256>>>(d, (uint32_t)n - 1, steps, thresh, out);
                    }
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms[mode], e0, e1));
                }
                std::vector<uint32_t> o(1024);
                CK(hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost));
                for (uint32_t v : o) {
                    chk[mode] ^= v;
                }
            }
            const double visits = (double)blocks * 256 * steps * (act / 100.0);
            printf("nodes 2^%d (%5zu MB) active %3u%%: own %6.1f G visits/s (%.2f TA-cyc/visit/CU @2.3GHz) | quad %6.1f G (%.2f) %s | "
                   "one %6.1f G\n",
                   log_nodes, n * 64 >> 20, act, visits / ms[0] * 1e-6, ms[0] * 1e-3 * 2.3e9 * cus / visits,
                   visits / ms[1] * 1e-6, ms[1] * 1e-3 * 2.3e9 * cus / visits, chk[0] == chk[1] ? "same result" : "RESULT DIFFERS",
                   visits / ms[2] * 1e-6);
        }
        CK(hipFree(d));
    }
    return 0;
}
