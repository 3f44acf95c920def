// VALU issue-rate micro-benchmark for the ops of the scan inner loop (run on the GPU box: hipcc --offload-arch=gfx950 -O3
// tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu).  One wave per block, WPS waves per SIMD, 8 independent
// chains per lane; prints cycles per wave-instruction per SIMD assuming a 2.4 GHz clock, and s_memtime-based real cycles.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(64) void k(const uint32_t* __restrict__ seed, uint32_t* __restrict__ out, int iters, unsigned long long* clk) {
    uint32_t x[8];
    const uint32_t s0 = seed[threadIdx.x], s1 = seed[64 + threadIdx.x];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = s0 * (j + 1) + s1;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (OP == 0) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(x[j]) : "v"(s0));
                if (OP == 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(x[j]) : "v"(s0));
                if (OP == 2) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(x[j]) : "v"(s0), "v"(s1));
                if (OP == 3) asm volatile("v_lshl_add_u32 %0, %0, 8, %1" : "+v"(x[j]) : "v"(s0));
                if (OP == 4) asm volatile("v_min_u32 %0, %0, 1" : "+v"(x[j]));
                if (OP == 5) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x[j]) : "v"(s0));
                if (OP == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[j]));
                if (OP == 7) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x[j]));
                if (OP == 8) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(x[j]) : "v"(s0));
                if (OP == 9) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[j]) : "v"(s0), "v"(s1));
                if (OP == 10) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(x[j]) : "s"(iters), "v"(s1));
                if (OP == 11) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x[j]) : "v"(s0));
                if (OP == 12) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(x[j]) : "v"(s0), "v"(s1));
                if (OP == 13) asm volatile("v_mov_b32 %0, %1" : "=v"(x[j]) : "v"(s0));
                if (OP == 14) asm volatile("v_min_u32_e32 %0, 1, %0" : "+v"(x[j]));
                if (OP == 15) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0" : "+v"(x[j]) : "v"(s0));
                if (OP == 16) asm volatile("v_cmp_ne_u32_e32 vcc, 0, %1\n\tv_cndmask_b32_e32 %0, 1, %1, vcc" : "+v"(x[j]) : "v"(s0) : "vcc");
                if (OP == 17) asm volatile("v_lshlrev_b32_e32 %0, 8, %0" : "+v"(x[j]));
                if (OP == 18) asm volatile("v_xor_b32 %0, %1, %0\n\tv_bcnt_u32_b32 %0, %0, %1\n\tv_and_b32 %0, %1, %0\n\tv_and_or_b32 %0, %1, %2, %0\n\tv_xor_b32 %0, %1, %0" : "+v"(x[j]) : "v"(s0), "v"(s1));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += x[j];
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    if (blockIdx.x == 0 && threadIdx.x == 0) *clk = t1 - t0;
}

template <int OP>
int run(const char* name, int wps, uint32_t* seed, uint32_t* out, unsigned long long* clk) {
    const int blocks = 256 * 4 * wps, iters = 4096;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<OP><<<blocks, 64>>>(seed, out, 16, clk);
    CHECK(hipEventRecord(e0));
    k<OP><<<blocks, 64>>>(seed, out, iters, clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c; CHECK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
    const double insts_per_simd = (double)iters * 32 * wps;
    printf("%-22s wps=%d  %.3f ms  %.2f cyc/inst@2.4GHz  wave0 clk-counter ticks/inst(all waves of the SIMD)=%.2f\n", name, wps, ms, ms * 1e-3 * 2.4e9 / insts_per_simd,
           (double)c / (iters * 32.0 * wps));
    return 0;
}

int main() {
    uint32_t *seed, *out; unsigned long long* clk;
    CHECK(hipMalloc(&seed, 4096)); CHECK(hipMalloc(&out, 256 * 4 * 8 * 64 * 4)); CHECK(hipMalloc(&clk, 8));
    CHECK(hipMemset(seed, 0x5a, 4096));
    const int wlist[] = {1, 2, 3, 4, 6, 8};
    for (int wi = 0; wi < 6; ++wi) {
        const int wps = wlist[wi];
        run<0>("v_xor_b32", wps, seed, out, clk);
        run<1>("v_bcnt_u32_b32", wps, seed, out, clk);
        run<2>("v_and_or_b32 vvv", wps, seed, out, clk);
        run<10>("v_and_or_b32 svv", wps, seed, out, clk);
        run<3>("v_lshl_add_u32", wps, seed, out, clk);
        run<4>("v_min_u32 imm", wps, seed, out, clk);
        run<5>("v_and_b32", wps, seed, out, clk);
        run<6>("v_rcp_f32", wps, seed, out, clk);
        run<7>("v_cvt_f32_u32", wps, seed, out, clk);
        run<8>("v_mul_u32_u24", wps, seed, out, clk);
        run<9>("v_fmac_f32", wps, seed, out, clk);
        run<11>("v_add_u32", wps, seed, out, clk);
        run<12>("v_add3_u32", wps, seed, out, clk);
        run<13>("v_mov_b32", wps, seed, out, clk);
        run<14>("v_min_u32_e32 1,v", wps, seed, out, clk);
        run<15>("v_mov_b32_sdwa", wps, seed, out, clk);
        run<16>("v_cmp_e32+v_cndmask_e32 (x2)", wps, seed, out, clk);
        run<17>("v_lshlrev_b32_e32", wps, seed, out, clk);
        run<18>("mix xor,bcnt,and,and_or,xor (x5)", wps, seed, out, clk);
    }
    return 0;
}
