// Micro-benchmarks behind the scan-kernel design choices (run on the GPU box: tools/run_ubench.sh).
// Each kernel: 64-thread blocks (one wave), conflict-free [bucket][lane] addressing like xmh_scan.hip.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int NB = 65;

template <int MODE>
__global__ __launch_bounds__(64) void k(const uint32_t* __restrict__ seed, uint32_t* __restrict__ out, int iters) {
    extern __shared__ unsigned long long lds64[];
    uint32_t* lds32 = reinterpret_cast<uint32_t*>(lds64);
    const int lane = threadIdx.x;
    for (int d = 0; d < NB; ++d) lds64[d * 64 + lane] = 0;
    uint32_t x = seed[blockIdx.x * 64 + lane] | 1u;
    unsigned long long acc = 0;
    for (int it = 0; it < iters; ++it) {
        int d[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x = x * 1664525u + 1013904223u;
            d[u] = (x >> 16) % NB;
        }
        if (MODE == 0) {                 // ds_add_u32, no return
#pragma unroll
            for (int u = 0; u < 8; ++u) atomicAdd(&lds32[d[u] * 64 + lane], 1u);
        } else if (MODE == 1) {          // ds_add_rtn_u64
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += atomicAdd(&lds64[d[u] * 64 + lane], 0x100000001ull);
        } else if (MODE == 2) {          // ds_add_rtn_u32
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += atomicAdd(&lds32[d[u] * 64 + lane], 1u);
        } else if (MODE == 3) {          // plain read-modify-write u64 (batched: 8 reads, 8 writes; ignores in-group collisions)
            unsigned long long v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = lds64[d[u] * 64 + lane];
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc += v[u]; lds64[d[u] * 64 + lane] = v[u] + 0x100000001ull; }
        } else if (MODE == 4) {          // plain RMW u32
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = lds32[d[u] * 64 + lane];
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc += v[u]; lds32[d[u] * 64 + lane] = v[u] + 1u; }
        } else if (MODE == 5) {          // no LDS at all (VALU baseline of this loop)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += d[u];
        } else if (MODE == 6) {          // readlane broadcast + xor/popc (5 readlanes per item)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                uint32_t a = __builtin_amdgcn_readlane((int)x, u), b = __builtin_amdgcn_readlane((int)(x >> 3), u + 8);
                uint32_t c = __builtin_amdgcn_readlane((int)(x >> 5), u + 16), e = __builtin_amdgcn_readlane((int)(x >> 7), u + 24);
                uint32_t f = __builtin_amdgcn_readlane((int)(x >> 9), u + 32);
                acc += __popc(a ^ x) + __popc(b ^ (x >> 1)) + (((c & x) | (e & (x >> 2)) | (f & (x >> 4))) != 0);
            }
        }
    }
    out[blockIdx.x * 64 + lane] = (uint32_t)acc + (uint32_t)(acc >> 32) + lds32[lane];
}

template <int MODE>
int run(const char* name, int blocks, int iters, size_t lds, uint32_t* seed, uint32_t* out) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), lds, 0, seed, out, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), lds, 0, seed, out, iters);
    CHECK(hipEventRecord(b));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    // cycles per item per wave assuming 2.4 GHz
    printf("%-34s blocks=%5d  %8.3f ms  %7.2f ns/item/wave  (~%6.1f cyc@2.4GHz)\n", name, blocks, ms, ms * 1e6 / (iters * 8.0), ms * 1e6 / (iters * 8.0) * 2.4);
    return 0;
}

int main() {
    uint32_t *seed, *out;
    const int maxb = 256 * 16;
    CHECK(hipMalloc(&seed, maxb * 64 * 4)); CHECK(hipMalloc(&out, maxb * 64 * 4));
    uint32_t* h = (uint32_t*)malloc(maxb * 64 * 4);
    for (int i = 0; i < maxb * 64; ++i) h[i] = 2654435761u * (i + 1);
    CHECK(hipMemcpy(seed, h, maxb * 64 * 4, hipMemcpyHostToDevice));
    const int iters = 20000;
    const size_t lds = NB * 64 * 8;
    for (int wpc : {1, 4, 8}) {       // waves per CU (LDS 33 KB each -> at most 4 resident per CU with 64-bit counters)
        int blocks = 256 * wpc;
        printf("--- %d wave(s) per CU (if evenly spread) ---\n", wpc);
        if (run<5>("valu-only baseline", blocks, iters, lds, seed, out)) return 1;
        if (run<0>("ds_add_u32 (no rtn)", blocks, iters, lds, seed, out)) return 1;
        if (run<2>("ds_add_rtn_u32", blocks, iters, lds, seed, out)) return 1;
        if (run<1>("ds_add_rtn_u64", blocks, iters, lds, seed, out)) return 1;
        if (run<4>("ds_read_b32+ds_write_b32", blocks, iters, lds, seed, out)) return 1;
        if (run<3>("ds_read_b64+ds_write_b64", blocks, iters, lds, seed, out)) return 1;
        if (run<6>("5 readlane + xor/popc/and", blocks, iters, lds, seed, out)) return 1;
    }
    return 0;
}
