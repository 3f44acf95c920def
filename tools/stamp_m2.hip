// Where does a wave of k_scan_hist_m2 spend its cycles?  Standalone: includes the library's own kernel source, runs the STAMP
// instantiation (s_memtime around the phases of a batch) on the configs[1] shape and prints the per-phase averages.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/stamp_m2.hip -o tools/stamp_m2.bin && tools/stamp_m2.bin
#include "../clip-based-cross-modal-hash_amd/csrc/xmh_core.hip"
#include "../clip-based-cross-modal-hash_amd/csrc/xmh_scan.hip"

#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NML, int NW, int NQ>
int run(int Q, int R, int K, int C) {
    constexpr int NMI = 1 + NML, NMQ = 2 + NML;
    const int W = (K + 31) / 32, LW = (C + 31) / 32;
    std::vector<uint32_t> qb((size_t)Q * W), rb((size_t)R * W), ql((size_t)Q * LW), rl((size_t)R * LW);
    uint32_t st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return st ^ (st >> 15); };
    for (auto& x : qb) x = rnd() ^ (rnd() << 16);
    for (auto& x : rb) x = rnd() ^ (rnd() << 16);
    for (auto& x : ql) x = rnd() & rnd() & rnd() & rnd() & rnd();
    for (auto& x : rl) x = rnd() & rnd() & rnd() & rnd() & rnd();
    xmh_scan_plan p;
    if (xmh_scan_plan_make(Q, R, K, 0, &p)) { printf("plan: %s\n", xmh_last_error()); return 1; }
    uint32_t *d_qb, *d_rb, *d_ql, *d_rl;
    char* ws;
    CK(hipMalloc(&d_qb, qb.size() * 4)); CK(hipMalloc(&d_rb, rb.size() * 4)); CK(hipMalloc(&d_ql, ql.size() * 4)); CK(hipMalloc(&d_rl, rl.size() * 4));
    CK(hipMemcpy(d_qb, qb.data(), qb.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_rb, rb.data(), rb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ql, ql.data(), ql.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_rl, rl.data(), rl.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&ws, p.ws_bytes));
    const size_t cache_bytes = pair_cache_bytes(p, K, false);
    const WsLayout L = ws_layout(p, cache_bytes, R, true);
    uint4* gimg = reinterpret_cast<uint4*>(ws + L.gimg);
    uint4* qimg = reinterpret_cast<uint4*>(ws + L.qimg32);
    const int64_t gpieces = xmh::ceil_div(R, 64) * 4 * NMI * 64, qpieces = (p.qpad / 16) * NMQ * 64;
    const unsigned gblocks = (unsigned)xmh::ceil_div(gpieces, 256), qblocks = (unsigned)xmh::ceil_div(qpieces, 256);
    hipLaunchKernelGGL((k_scan_expand2<NML>), dim3(gblocks + qblocks), dim3(256), 0, 0, d_rb, d_rl, (int64_t)R, W, LW, K, gimg, gpieces, gblocks, d_qb, d_ql, (int64_t)Q,
                       qimg, qpieces, reinterpret_cast<uint32_t*>(ws + L.tick), (int)((L.gate + 256 - L.tick) / 4));
    MfmaArgs a{gimg, qimg, d_qb, Q, R, K, W, (int)p.chunk, (int)p.nchunk, (int)(p.qpad / (NW * NQ * 16)), (int)p.nbuckets, (int)p.qpad};
#ifdef XMH_ABL_SUB2
    const size_t lds = (size_t)NW * NQ * p.nbuckets * 32 * 4 + 3 * 4 * NMI * 1024;
#else
    const size_t lds = (size_t)NW * NQ * p.nbuckets * 16 * 4 + 3 * 4 * NMI * 1024;
#endif
    const unsigned grid = (unsigned)(8 * a.nqt * xmh::ceil_div(p.nchunk, 8));
    unsigned long long* d_st;
    CK(hipMalloc(&d_st, (size_t)grid * NW * 8 * 8));
    CK(hipMemset(d_st, 0, (size_t)grid * NW * 8 * 8));
    #ifdef STAMP_NOCACHE
    constexpr bool CACHE = false;
#else
    constexpr bool CACHE = true;
#endif
    auto kern = k_scan_hist_m2<NML, NW, NQ, CACHE, true>;
    auto plain = k_scan_hist_m2<NML, NW, NQ, CACHE, false>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(plain), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    uint32_t* chunk_hist = reinterpret_cast<uint32_t*>(ws + L.chunk_hist);
    uint4* cache = cache_bytes ? reinterpret_cast<uint4*>(ws + L.pair_cache) : nullptr;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best[2] = {1e9f, 1e9f};
    for (int it = 0; it < 12; ++it) {
        for (int v = 0; v < 2; ++v) {
            CK(hipEventRecord(e0));
            if (v == 0) hipLaunchKernelGGL(plain, dim3(grid), dim3(64 * NW), lds, 0, a, chunk_hist, cache, (unsigned long long*)nullptr);
            else hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, 0, a, chunk_hist, cache, d_st);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best[v]) best[v] = ms;
        }
    }
    CK(hipGetLastError());
    std::vector<unsigned long long> h((size_t)grid * NW * 8);
    CK(hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost));
    double sum[8] = {0}; size_t nw = 0;
    for (size_t w = 0; w < (size_t)grid * NW; ++w) {
        if (!h[w * 8 + 5]) continue;                     // blocks past the last chunk
        ++nw;
        for (int k = 0; k < 8; ++k) sum[k] += (double)h[w * 8 + k];
    }
    const double nbat = (double)((p.chunk + 63) / 64);
    static const char* names[8] = {"prologue (once)", "wait vmcnt (pieces landed)", "barrier", "LDS-DMA issue", "A tiles ds_read", "MFMA + consume", "epilogue (once)", "cache stores + loop"};
    printf("NML=%d NW=%d NQ=%d  Q=%d R=%d K=%d C=%d: chunk %lld x %lld, grid %u, lds %zu; plain %.4f ms, stamped %.4f ms; %zu waves, %.0f batches each\n", NML, NW, NQ, Q, R,
           K, C, (long long)p.chunk, (long long)p.nchunk, grid, lds, best[0], best[1], nw, nbat);
    double tot = 0; for (int k = 0; k < 8; ++k) tot += sum[k];
    for (int k = 0; k < 8; ++k)
        printf("  %-28s %9.0f cycles per wave  %7.1f per batch  %5.1f %%\n", names[k], sum[k] / nw, sum[k] / nw / nbat, 100.0 * sum[k] / tot);
    printf("  total %.0f cycles per wave = %.1f us at 2.4 GHz (s_memtime ticks at 100 MHz? then x24)\n", tot / nw, tot / nw / 2400.0);
    hipFree(d_qb); hipFree(d_rb); hipFree(d_ql); hipFree(d_rl); hipFree(ws); hipFree(d_st);
    return 0;
}

int main(int argc, char** argv) {
    int rc = 0;
    const int K = argc > 1 ? atoi(argv[1]) : 64;
    if (K != 64) return run<2, 4, 2>(5000, 117218, K, 80);      // forced 4 x 2 geometry at another code length (XMH_SCAN_M2_GEOM=0 in the environment for the plan)
    const M2Geom g = m2_geom(64);
    if (g.nw == 4 && g.nq == 2) rc |= run<2, 4, 2>(5000, 117218, 64, 80);
    else if (g.nw == 8 && g.nq == 1) rc |= run<2, 8, 1>(5000, 117218, 64, 80);
    else if (g.nw == 4 && g.nq == 1) rc |= run<2, 4, 1>(5000, 117218, 64, 80);
    else printf("no instance for this XMH_SCAN_M2_GEOM\n");
    return rc;
}
