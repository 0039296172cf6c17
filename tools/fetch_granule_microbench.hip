// fetch_granule_microbench.hip — how many HBM bytes does ONE random 4-byte load cost?  (development aid, not product)
// Every lane reads 4 bytes at a hashed address of a 2 GiB buffer (far beyond L2 and the Infinity Cache), `reps` times with
// independent addresses. Run under `rocprofv3 --pmc FETCH_SIZE` (and TCC_EA_RDREQ_sum / TCC_EA_RDREQ_32B_sum): bytes fetched per
// load = 2 x FETCH_SIZE x 1024 / loads on gfx950 (MI355X_MICROARCH.md). Decides whether 64-byte texel tiles would halve
// k_shade's texel traffic or whether a miss fills a whole 128-byte line anyway.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/fetch_granule_microbench tools/fetch_granule_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void k_random4(const uint32_t *buf, uint64_t n_words, uint32_t reps, uint32_t stride_words, uint32_t second, uint32_t *out)
{
    uint64_t x = (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345u;
    uint32_t acc = 0;
    for (uint32_t r = 0; r < reps; ++r) {
        x ^= x >> 29;
        x *= 0xBF58476D1CE4E5B9ull;
        x ^= x >> 32;
        const uint64_t w = (x % (n_words / stride_words)) * stride_words; // aligned to `stride_words` words
        acc += buf[w];
        if (second) { // a second word of the same aligned block, `second` words further on
            acc += buf[w + second];
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char **argv)
{
    const uint64_t bytes = 2ull << 30;
    const uint32_t stride_words = argc > 1 ? (uint32_t)atoi(argv[1]) : 1u; // 1: any word; 16: first word of a 64-byte block; 32: of a 128-byte line
    const uint32_t second = argc > 2 ? (uint32_t)atoi(argv[2]) : 0u;      // e.g. 32 16: both 64-byte halves of a random 128-byte line
    uint32_t *buf, *out;
    const int threads = 256 * 256 * 16, reps = 64;
    (void)hipMalloc(&buf, bytes);
    (void)hipMalloc(&out, threads * 4);
    (void)hipMemset(buf, 1, bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        (void)hipEventRecord(e0);
        k_random4<<<threads / 256, 256>>>(buf, bytes / 4, reps, stride_words, second, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("stride %u words, second +%u: %d visits in %.3f ms = %.1f G visits/s\n", stride_words, second, threads * reps, ms, threads * (double)reps / ms / 1e6);
    }
    return 0;
}
