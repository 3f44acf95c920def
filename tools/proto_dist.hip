// Timing harness for the materialised-output kernels (csrc/xmh_dist.hip compiled into this binary, with -D variants):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I clip-based-cross-modal-hash_amd/csrc -I include [-DXMH_DIST_PLAIN] [-DXMH_DIST_QTILE=n] \
//         tools/proto_dist.hip clip-based-cross-modal-hash_amd/csrc/xmh_dist.hip clip-based-cross-modal-hash_amd/csrc/xmh_core.hip -o tools/proto_dist.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "xmh.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_fill(float4* p, size_t n4) { for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = make_float4(1.f, 2.f, 3.f, 4.f); }
int main(int argc, char** argv) {
    const int64_t Q = 2000, R = argc > 1 ? atoll(argv[1]) : 117218;
    const int K = argc > 2 ? atoi(argv[2]) : 64, W = (K + 31) / 32;
    std::vector<uint32_t> hq(Q * W), hr(R * W);
    uint64_t x = 88172645463325252ull;
    for (auto& v : hq) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)(x >> 16); }
    for (auto& v : hr) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)(x >> 16); }
    uint32_t *dq, *dr; float* out;
    CK(hipMalloc(&dq, hq.size() * 4)); CK(hipMalloc(&dr, hr.size() * 4)); CK(hipMalloc(&out, (size_t)Q * R * 4 + 64));
    CK(hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dr, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timed = [&](const char* what, auto fn) {
        for (int i = 0; i < 3; ++i) fn();
        CK(hipEventRecord(a));
        for (int i = 0; i < 10; ++i) fn();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-36s %8.1f us  %5.2f TB/s\n", what, ms / 10 * 1e3, (double)Q * R * 4 / (ms / 10 * 1e-3) / 1e12);
    };
    timed("xmh_hamming_dist f32", [&]() { if (xmh_hamming_dist(dq, nullptr, dr, nullptr, Q, R, K, out, nullptr, nullptr)) { fprintf(stderr, "%s\n", xmh_last_error()); exit(1); } });
    timed("float4 fill of the same bytes", [&]() { hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, (float4*)out, (size_t)Q * R / 4); });
    return 0;
}
