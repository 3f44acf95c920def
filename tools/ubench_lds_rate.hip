// Throughput of LDS atomics in the access pattern of the scan's pass 2 (lane = slot * 16 + query; counter [bucket][query] of 8 or 4
// bytes; the 4 slots of a query hit buckets drawn from a narrow range, so some instructions carry same-address lanes):
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_lds_rate.hip -o /tmp/ubench_lds_rate && /tmp/ubench_lds_rate
// Prints cycles per wave-instruction per CU (2.4 GHz assumed) at several waves per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int NB = 65;

// MODE 0: ds_add_rtn_u64   1: ds_add_rtn_u32   2: ds_add_u32 (no return)   3: ds_add_u64 (no return)   4: ds_read_b64
template <int MODE, int SPREAD>
__global__ __launch_bounds__(64) void k(uint32_t* __restrict__ out, int iters) {
    extern __shared__ unsigned long long lds64[];
    uint32_t* lds32 = reinterpret_cast<uint32_t*>(lds64);
    const int lane = threadIdx.x, ql = lane & 15, slot = lane >> 4;
    for (int e = lane; e < NB * 16; e += 64) lds64[e] = 0;
    __syncthreads();
    // 8 bucket indices per lane around the middle of the range; SPREAD = how far the 4 slots of a query are apart
    int d[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) d[u] = 24 + ((u * 5 + slot * SPREAD + (ql * 3 >> 2)) % 17);
    unsigned long long acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int dd = d[u] ^ (it & 1);
            if (MODE == 0) acc += atomicAdd(&lds64[dd * 16 + ql], 0x100000001ull);
            if (MODE == 1) acc += atomicAdd(&lds32[dd * 16 + ql], 0x10001u);
            if (MODE == 2) atomicAdd(&lds32[dd * 16 + ql], 1u);
            if (MODE == 3) atomicAdd(&lds64[dd * 16 + ql], 0x100000001ull);
            if (MODE == 4) acc += lds64[dd * 16 + ql];
        }
    }
    out[blockIdx.x * 64 + lane] = (uint32_t)acc + (uint32_t)(acc >> 32) + lds32[lane];
}

template <int MODE, int SPREAD>
int run(const char* name, int waves_per_cu, int cus) {
    uint32_t* out;
    const int blocks = cus * waves_per_cu, iters = 4000;
    CHECK(hipMalloc(&out, (size_t)blocks * 64 * 4));
    const size_t lds = (size_t)NB * 16 * 8;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE, SPREAD>), dim3(blocks), dim3(64), lds, 0, out, 100);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, SPREAD>), dim3(blocks), dim3(64), lds, 0, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_cu = (double)waves_per_cu * iters * 8;
    printf("%-22s slots %s  waves/CU %2d  %.3f ms  %.2f cycles per wave-instruction per CU\n", name, SPREAD ? "apart " : "collide", waves_per_cu, ms,
           ms * 1e-3 * 2.4e9 / instr_per_cu);
    CHECK(hipFree(out));
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    for (int w : {4, 8, 16}) {
        run<0, 4>("ds_add_rtn_u64", w, cus);
        run<0, 0>("ds_add_rtn_u64", w, cus);
        run<1, 4>("ds_add_rtn_u32", w, cus);
        run<1, 0>("ds_add_rtn_u32", w, cus);
        run<2, 4>("ds_add_u32", w, cus);
        run<3, 4>("ds_add_u64", w, cus);
        run<4, 4>("ds_read_b64", w, cus);
    }
    return 0;
}
