// What the MFMA pipe sustains on this chip as a function of operand DATA (DVFS / power), with nothing else running:
// 1024 waves (one per SIMD) or 2048 (two per SIMD), each a loop of independent v_mfma_f32_32x32x16_f16 (or bf16 / 16x16x32) on
// operands loaded once.  Prints TFLOP/s and the effective shader clock (s_memtime ticks per wall-clock tick) for zero, constant,
// small-integer, uniform [-1,1) and "parity lo-plane like" (tiny magnitude, random mantissa) operands.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_mfma_power.hip -o /tmp/mp && /tmp/mp
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(512) void k_mfma(const uint4* in, float* out, int iters, unsigned long long* tim) {
    const int tid = threadIdx.x + blockIdx.x * blockDim.x;
    uint4 ra[4], rb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { ra[i] = in[(tid * 8 + i) & 0xfffff]; rb[i] = in[(tid * 8 + 4 + i) & 0xfffff]; }
    f32x16 acc[4];
    f32x4 acc4[8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc4[i][e] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (KIND == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[(i + u) & 3]), __builtin_bit_cast(f16x8, rb[i]), acc[i], 0, 0, 0);
            } else if (KIND == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[(i + u) & 3]), __builtin_bit_cast(bf16x8, rb[i]), acc[i], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ra[(i + u) & 3]), __builtin_bit_cast(f16x8, rb[i & 3]), acc4[i], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += acc4[i][e];
    out[tid] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { tim[0] = t1 - t0; tim[1] = w1 - w0; }
}

static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x8000u) >> 16); }

int main() {
    const size_t n16 = (size_t)(1 << 20) * 8;      // 16-byte words x 8 halves
    std::vector<uint16_t> h(n16);
    uint4* din; float* dout; unsigned long long* tim;
    CK(hipMalloc(&din, n16 * 2)); CK(hipMalloc(&dout, 4096 * 512 * 4)); CK(hipMalloc(&tim, 16));
    const char* fills[] = {"zero", "const 1.0", "small ints -3..3", "uniform [-1,1)", "normal-ish sum of 4 uniforms", "tiny |x|<1e-3 random mantissa", "sign-only random (+-0.5)"};
    uint32_t seed = 12345;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (float)((seed >> 8) & 0xffffff) / 8388608.0f - 1.0f; };
    for (int kind = 0; kind < 3; ++kind) {
        printf("%s\n", kind == 0 ? "v_mfma_f32_32x32x16_f16" : kind == 1 ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_16x16x32_f16");
        for (int f = 0; f < 7; ++f) {
            for (size_t i = 0; i < n16; ++i) {
                float v;
                switch (f) {
                    case 0: v = 0.f; break;
                    case 1: v = 1.f; break;
                    case 2: v = (float)((int)(rnd() * 3.49f)); break;
                    case 3: v = rnd(); break;
                    case 4: v = 0.5f * (rnd() + rnd() + rnd() + rnd()); break;
                    case 5: v = rnd() * 1e-3f; break;
                    default: v = rnd() < 0 ? -0.5f : 0.5f; break;
                }
                h[i] = kind == 1 ? f2bf(v) : f2h(v);
            }
            CK(hipMemcpy(din, h.data(), n16 * 2, hipMemcpyHostToDevice));
            for (int nthr = 256; nthr <= 512; nthr += 256) {
                const int iters = 40000;      // 16 MFMAs (32 of the 16x16) per iteration
                auto launch = [&]() {
                    if (kind == 0) hipLaunchKernelGGL(k_mfma<0>, dim3(256), dim3(nthr), 0, 0, din, dout, iters, tim);
                    else if (kind == 1) hipLaunchKernelGGL(k_mfma<1>, dim3(256), dim3(nthr), 0, 0, din, dout, iters, tim);
                    else hipLaunchKernelGGL(k_mfma<2>, dim3(256), dim3(nthr), 0, 0, din, dout, iters, tim);
                };
                launch();
                CK(hipDeviceSynchronize());
                hipEvent_t e0, e1;
                CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                CK(hipEventRecord(e0));
                launch();
                CK(hipEventRecord(e1));
                CK(hipDeviceSynchronize());
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                unsigned long long ht[2];
                CK(hipMemcpy(ht, tim, 16, hipMemcpyDeviceToHost));
                const double flops = 256.0 * (nthr / 64) * iters * 16.0 * 32768.0;
                printf("  %-34s %d waves/SIMD  %7.3f ms  %7.1f TF  clock %.0f MHz  cycles per 32x32x16-equivalent %.1f\n", fills[f], nthr / 256, ms, flops / ms / 1e9,
                       ht[0] / (ht[1] / 100.0), (double)ht[0] / ((double)iters * 16 * (nthr / 256)));
            }
        }
    }
    return 0;
}
