// Ablation harness for the MFMA scan kernels: compiles csrc/xmh_scan_mfma.h -- the library's own source -- in seconds and times
// k_scan_hist_r2 (pass 1) and k_scan_ap_c (pass 2) at the headline shape with pieces of their loops switched off by -DXMH_ABL_* macros,
// to see what each pipe (matrix, VALU, LDS atomics, vector memory) costs and how much of it overlaps.  Results are WRONG under an
// ablation; only the time means something.  Without macros the histogram checksum is compared with a plain reference kernel.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I clip-based-cross-modal-hash_amd/csrc [-DXMH_ABL_...] tools/proto_scan_ablate.hip -o tools/proto_scan_ablate.bin
//   gpurun -- ./tools/proto_scan_ablate.bin            (tools/run_scan_ablate.sh builds and runs the whole matrix)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "xmh_scan_mfma.h"

#ifndef PNW
#define PNW 4
#endif
#ifndef PNQ
#define PNQ 4
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

namespace {
__global__ void k_ref_hist(const uint32_t* qbits, const uint32_t* qlab, const uint32_t* rbits, const uint32_t* rlab, int Q, int R, int W, int LW, int nb,
                           unsigned long long* sum) {
    // checksum of the (distance, relevant) histogram over all pairs: sum over pairs of (d + 1) * (1 + 1000 rel)
    unsigned long long acc = 0;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < (int64_t)Q * R; p += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(p / R), r = (int)(p % R);
        int d = 0;
        for (int w = 0; w < W; ++w) d += __popc(qbits[q * W + w] ^ rbits[(int64_t)r * W + w]);
        uint32_t h = 0;
        for (int w = 0; w < LW; ++w) h |= qlab[q * LW + w] & rlab[(int64_t)r * LW + w];
        acc += (unsigned long long)(d + 1) * (h ? 1001ull : 1ull);
    }
    atomicAdd(sum, acc);
}
__global__ void k_sum_hist(const uint32_t* chunk_hist, int nchunk, int nb, int qpad, int Q, unsigned long long* sum) {
    unsigned long long acc = 0;
    const int64_t cells = (int64_t)nchunk * nb * qpad;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < cells; e += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(e % qpad), d = (int)((e / qpad) % nb);
        if (q >= Q) continue;
        const uint32_t v = chunk_hist[e];
        const uint32_t all = v >> 16, rel = v & 0xffffu;
        acc += (unsigned long long)(d + 1) * ((unsigned long long)(all - rel) + 1001ull * rel);
    }
    atomicAdd(sum, acc);
}
}  // namespace

int main(int argc, char** argv) {
    const int Q = 5000, R = 117218, K = 64, C = 80, W = 2, LW = 3, nb = K + 1;
    const int iters = argc > 1 ? atoi(argv[1]) : 50;
    // the library's plan for this shape (xmh_scan_plan_make): 256-query blocks, two per CU, one set
    const int qpad = 5120, nqt256 = qpad / 256;
    int nchunk = 256 * 2 / nqt256;
    nchunk = (nchunk + 4) / 8 * 8;
    int chunk = (R + nchunk - 1) / nchunk;
    chunk = (chunk + 63) / 64 * 64;
    nchunk = (R + chunk - 1) / chunk;
    std::vector<uint32_t> hq((size_t)Q * W), hr((size_t)R * W), hql((size_t)Q * LW), hrl((size_t)R * LW);
    uint64_t x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)(x >> 16); };
    for (auto& v : hq) v = rnd();
    for (auto& v : hr) v = rnd();
    for (size_t i = 0; i < hql.size(); ++i) hql[i] = rnd() & rnd() & rnd() & rnd() & (i % LW == LW - 1 ? 0xffffu : 0xffffffffu);
    for (size_t i = 0; i < hrl.size(); ++i) hrl[i] = rnd() & rnd() & rnd() & rnd() & (i % LW == LW - 1 ? 0xffffu : 0xffffffffu);
    uint32_t *dq, *dr, *dql, *drl, *chunk_hist, *ctl;
    uint4* cache;
    uint2 *below, *dpre;
    float* ap_part;
    unsigned long long* sums;
    const size_t cells = (size_t)nchunk * nb * qpad;
    const size_t cache_bytes = (size_t)nchunk * (qpad / 16) * ((chunk + 63) / 64) * 1024;
    CK(hipMalloc(&dq, hq.size() * 4)); CK(hipMalloc(&dr, hr.size() * 4)); CK(hipMalloc(&dql, hql.size() * 4)); CK(hipMalloc(&drl, hrl.size() * 4));
    CK(hipMalloc(&chunk_hist, cells * 4)); CK(hipMalloc(&cache, cache_bytes)); CK(hipMalloc(&ctl, 65536)); CK(hipMalloc(&sums, 16));
    CK(hipMalloc(&below, cells * 8)); CK(hipMalloc(&dpre, (size_t)nb * qpad * 8)); CK(hipMalloc(&ap_part, (size_t)nchunk * qpad * 4));
    CK(hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dr, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dql, hql.data(), hql.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(drl, hrl.data(), hrl.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(below, 0, cells * 8)); CK(hipMemset(dpre, 0, (size_t)nb * qpad * 8)); CK(hipMemset(sums, 0, 16));

    MfmaArgs a{dq, Q, R, K, W, chunk, nchunk, qpad / (PNW * PNQ * 16), nb, qpad};
    a.rbits = dr; a.rlab = drl; a.qlab = dql; a.LW = LW;
    constexpr int NW = PNW, NQ = PNQ;
    const size_t lds1 = (size_t)NW * NQ * nb * 16 * 4;
    auto k1 = k_scan_hist_r2<2, NW, NQ, true>;
    auto k1n = k_scan_hist_r2<2, NW, NQ, false>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k1n), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    const dim3 grid1((unsigned)(8 * a.nqt * ((nchunk + 7) / 8)));
    ScanArgs s{};
    s.qbits = dq; s.qlab = dql; s.rbits = dr; s.rlab = drl; s.Q = Q; s.R = R; s.K = K;
    s.chunk = chunk; s.nchunk = nchunk; s.nqt = qpad / 16; s.qpad = qpad; s.nb = nb; s.pair_cache = cache;
    const size_t lds2 = (size_t)nb * 16 * 8;
    auto k2 = k_scan_ap_c<false, 8, false>;
    const dim3 grid2((unsigned)(8 * s.nqt * ((nchunk + 7) / 8)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](const char* what, auto launch) {
        for (int i = 0; i < 10; ++i) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double t = ms / iters * 1e-3;
        printf("%-34s %8.4f ms   %6.1f cycles per 64 pairs and SIMD\n", what, t * 1e3, t * 2.4e9 * 1024 / ((double)Q * R / 64));
        return t;
    };
    auto touch = [&]() { hipLaunchKernelGGL(k_scan_touch, dim3((unsigned)(8 * kTouchPerChunk * ((nchunk + 7) / 8))), dim3(256), 0, 0, dr, drl, (int64_t)R, W, LW, (int64_t)chunk, nchunk, ctl, 64); };
    {
        int occ = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k1, 64 * NW, lds1));
        int occn = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occn, k1n, 64 * NW, lds1));
        hipFuncAttributes fa, fan;
        CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k1)));
        CK(hipFuncGetAttributes(&fan, reinterpret_cast<const void*>(k1n)));
        printf("pass 1: %d waves x %d query groups, LDS %zu B per block; with cache: %d VGPRs, %d blocks per CU; without: %d VGPRs, %d blocks per CU\n", NW, NQ, lds1,
               fa.numRegs, occ, fan.numRegs, occn);
    }
    printf("plan: %d chunks of %d items, grid pass 1 %u blocks, pass 2 %u blocks\n", nchunk, chunk, grid1.x, grid2.x);
    timed("pass 1 with pair cache", [&]() { touch(); hipLaunchKernelGGL(k1, grid1, dim3(64 * NW), lds1, 0, a, chunk_hist, cache); });
    timed("pass 1 without pair cache", [&]() { touch(); hipLaunchKernelGGL(k1n, grid1, dim3(64 * NW), lds1, 0, a, chunk_hist, (uint4*)nullptr); });
    hipLaunchKernelGGL(k1, grid1, dim3(64 * NW), lds1, 0, a, chunk_hist, cache);
#if !defined(XMH_ABL_ANY)
    hipLaunchKernelGGL(k_sum_hist, dim3(1024), dim3(256), 0, 0, chunk_hist, nchunk, nb, qpad, Q, sums);
    hipLaunchKernelGGL(k_ref_hist, dim3(4096), dim3(256), 0, 0, dq, dql, dr, drl, Q, R, W, LW, nb, sums + 1);
    unsigned long long h[2];
    CK(hipMemcpy(h, sums, 16, hipMemcpyDeviceToHost));
    printf("histogram checksum %llu, reference %llu: %s\n", h[0], h[1], h[0] == h[1] ? "equal" : "DIFFERENT");
#endif
    timed("pass 2 (k_scan_ap_c)", [&]() { hipLaunchKernelGGL(k2, grid2, dim3(64), lds2, 0, s, (const uint2*)below, (const uint2*)dpre, (const uint32_t*)nullptr, ap_part, (const uint32_t*)nullptr, 0xffffffffu, (const uint32_t*)nullptr, (const uint32_t*)nullptr, 0); });
    return 0;
}
