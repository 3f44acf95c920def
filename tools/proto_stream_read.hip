// What a read-only pass over a gallery of 256-bit codes can reach from HBM (no Infinity Cache help: `ngal` galleries of 320 MB in
// rotation, 1.28 GB in all), by load form.  The top-k filter of rounds 1-4 (k_topk_filter<8, 2, 1, 1>, removed in round 5) is variant "rec": one lane
// per item, two 16-byte loads 32 bytes apart between neighbouring lanes.  Every variant computes the same thing -- the Hamming
// distance of each item to one query, a count of the items under a threshold -- so the loads cannot be dropped and the integer work
// is the filter's.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/proto_stream_read.hip -o tools/proto_stream_read.bin && gpurun -- ./tools/proto_stream_read.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#ifdef WITH_LIB                                                  // -DWITH_LIB -I clip-based-cross-modal-hash_amd/csrc -I include: the library's own filter kernels in the same loop
#include "xmh_topk.hip"
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ u32x4 ld16(const u32x4* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
__device__ __forceinline__ int pc4(u32x4 a, u32x4 q) { return __popc(a.x ^ q.x) + __popc(a.y ^ q.y) + __popc(a.z ^ q.z) + __popc(a.w ^ q.w); }

// rec: lane = item, two loads per item (the shipped form).  IPT items per thread per tile, the next tile's loads issued before this tile's work.
template <int IPT, bool NT>
__global__ __launch_bounds__(256) void k_rec(const u32x4* __restrict__ g, int64_t R, const u32x4* __restrict__ q, int thr, unsigned* __restrict__ out) {
    const u32x4 q0 = q[0], q1 = q[1];
    constexpr int TILE = 256 * IPT;
    const int64_t ntiles = R / TILE;
    u32x4 cur[IPT][2], nxt[IPT][2];
    int64_t tile = blockIdx.x;
    if (tile < ntiles)
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const int64_t it = tile * TILE + j * 256 + threadIdx.x;
            cur[j][0] = ld16<NT>(g + it * 2); cur[j][1] = ld16<NT>(g + it * 2 + 1);
        }
    unsigned hits = 0;
    for (; tile < ntiles; tile += gridDim.x) {
        const int64_t tn = tile + gridDim.x;
        if (tn < ntiles)
#pragma unroll
            for (int j = 0; j < IPT; ++j) {
                const int64_t it = tn * TILE + j * 256 + threadIdx.x;
                nxt[j][0] = ld16<NT>(g + it * 2); nxt[j][1] = ld16<NT>(g + it * 2 + 1);
            }
#pragma unroll
        for (int j = 0; j < IPT; ++j) hits += (pc4(cur[j][0], q0) + pc4(cur[j][1], q1)) <= thr;
        if (tn < ntiles)
#pragma unroll
            for (int j = 0; j < IPT; ++j) { cur[j][0] = nxt[j][0]; cur[j][1] = nxt[j][1]; }
    }
    if (__ballot(hits != 0) && hits) atomicAdd(out, hits);
}

// seq: every load instruction of a wave covers 1 KB contiguous; lanes 2i, 2i+1 hold the halves of item i, one DPP add joins them.
// NLD loads per thread per tile.
template <int NLD, bool NT>
__global__ __launch_bounds__(256) void k_seq(const u32x4* __restrict__ g, int64_t R, const u32x4* __restrict__ q, int thr, unsigned* __restrict__ out) {
    const u32x4 qh = q[threadIdx.x & 1];
    constexpr int TILE16 = 256 * NLD;                          // 16-byte pieces per tile
    const int64_t ntiles = R * 2 / TILE16;
    u32x4 cur[NLD], nxt[NLD];
    int64_t tile = blockIdx.x;
    if (tile < ntiles)
#pragma unroll
        for (int j = 0; j < NLD; ++j) cur[j] = ld16<NT>(g + tile * TILE16 + j * 256 + threadIdx.x);
    unsigned hits = 0;
    for (; tile < ntiles; tile += gridDim.x) {
        const int64_t tn = tile + gridDim.x;
        if (tn < ntiles)
#pragma unroll
            for (int j = 0; j < NLD; ++j) nxt[j] = ld16<NT>(g + tn * TILE16 + j * 256 + threadIdx.x);
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int h = pc4(cur[j], qh);
            const int d = h + __builtin_amdgcn_mov_dpp(h, 0xB1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]: the neighbour's half
            hits += (d <= thr) & (int)(~threadIdx.x & 1);
        }
        if (tn < ntiles)
#pragma unroll
            for (int j = 0; j < NLD; ++j) cur[j] = nxt[j];
    }
    if (__ballot(hits != 0) && hits) atomicAdd(out, hits);
}

// chunk: each block owns ONE contiguous range of the gallery (no grid stride): block b reads [b, b+1) * R / grid
template <int NLD, bool NT>
__global__ __launch_bounds__(256) void k_chunk(const u32x4* __restrict__ g, int64_t R, const u32x4* __restrict__ q, int thr, unsigned* __restrict__ out) {
    const u32x4 qh = q[threadIdx.x & 1];
    constexpr int TILE16 = 256 * NLD;
    const int64_t ntiles = R * 2 / TILE16;
    const int64_t t0 = ntiles * blockIdx.x / gridDim.x, t1 = ntiles * (blockIdx.x + 1) / gridDim.x;
    u32x4 cur[NLD], nxt[NLD];
    if (t0 < t1)
#pragma unroll
        for (int j = 0; j < NLD; ++j) cur[j] = ld16<NT>(g + t0 * TILE16 + j * 256 + threadIdx.x);
    unsigned hits = 0;
    for (int64_t tile = t0; tile < t1; ++tile) {
        if (tile + 1 < t1)
#pragma unroll
            for (int j = 0; j < NLD; ++j) nxt[j] = ld16<NT>(g + (tile + 1) * TILE16 + j * 256 + threadIdx.x);
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int h = pc4(cur[j], qh);
            const int d = h + __builtin_amdgcn_mov_dpp(h, 0xB1, 0xf, 0xf, true);
            hits += (d <= thr) & (int)(~threadIdx.x & 1);
        }
        if (tile + 1 < t1)
#pragma unroll
            for (int j = 0; j < NLD; ++j) cur[j] = nxt[j];
    }
    if (__ballot(hits != 0) && hits) atomicAdd(out, hits);
}

int main(int argc, char** argv) {
    const int ngal = argc > 1 ? atoi(argv[1]) : 4, iters = argc > 2 ? atoi(argv[2]) : 40;
    const int64_t R = (argc > 3 ? atoll(argv[3]) : 10000000) / 2048 * 2048;      // whole tiles for every variant
    const size_t bytes = (size_t)R * 32;
    std::vector<uint32_t> h((size_t)R * 8);
    uint64_t x = 88172645463325252ull;
    for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)(x >> 16); }
    std::vector<u32x4*> gal(ngal);
    for (int i = 0; i < ngal; ++i) { CK(hipMalloc(&gal[i], bytes)); CK(hipMemcpy(gal[i], h.data(), bytes, hipMemcpyHostToDevice)); }
    u32x4* dq; unsigned* out;
    CK(hipMalloc(&dq, 32)); CK(hipMemcpy(dq, h.data(), 32, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, 4)); CK(hipMemset(out, 0, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int thr = 96;
    unsigned ref = 0;
    auto run = [&](const char* name, auto kern, int grid) {
        CK(hipMemset(out, 0, 4));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, gal[0], R, dq, thr, out);
        unsigned got = 0;
        CK(hipMemcpy(&got, out, 4, hipMemcpyDeviceToHost));
        if (!ref) ref = got;
        for (int i = 0; i < 2 * ngal; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, gal[i % ngal], R, dq, thr, out);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, gal[i % ngal], R, dq, thr, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double t = ms / iters * 1e-3;
        printf("%-28s grid %5d  %7.2f us  %5.2f TB/s  (%.3f of 8)  hits %u%s\n", name, grid, t * 1e6, bytes / t / 1e12, bytes / t / 8e12, got, got == ref ? "" : "  MISMATCH");
    };
    printf("%d galleries of %.0f MB in rotation, %d launches each variant (time includes the launch gap)\n", ngal, bytes / 1e6, iters);
#ifdef WITH_LIB
    {
        uint32_t* t_est; uint32_t* bnd; uint32_t* cnt; unsigned long long* cand;
        CK(hipMalloc(&t_est, 64)); CK(hipMalloc(&bnd, 64)); CK(hipMemset(bnd, 0x7f, 64));      // index bound = none (0x7f7f7f7f rows)
        CK(hipMalloc(&cnt, 4096)); CK(hipMalloc(&cand, (size_t)kCandCap * 8 * 2));
        const uint32_t th = (uint32_t)(getenv("LIB_THR") ? atoi(getenv("LIB_THR")) : thr);
        CK(hipMemcpy(t_est, &th, 4, hipMemcpyHostToDevice));
        auto runlib = [&](const char* name, auto kern, int grid) {
            for (int i = 0; i < 2 * ngal; ++i) { CK(hipMemsetAsync(cnt, 0, 4096, 0)); hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, (const uint32_t*)dq, (const uint32_t*)gal[i % ngal], 1, R, (const uint32_t*)t_est, (const uint32_t*)bnd, cnt, cand); }
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, (const uint32_t*)dq, (const uint32_t*)gal[i % ngal], 1, R, (const uint32_t*)t_est, (const uint32_t*)bnd, cnt, cand);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double t = ms / iters * 1e-3;
            unsigned got = 0;
            CK(hipMemcpy(&got, cnt, 4, hipMemcpyDeviceToHost));
            printf("%-28s grid %5d  %7.2f us  %5.2f TB/s  (%.3f of 8)  candidates %u (over %d launches)\n", name, grid, t * 1e6, bytes / t / 1e12, bytes / t / 8e12, got, iters);
        };
        for (int grid : {512, 1024, 2048}) {
            runlib("LIB k_topk_filter_seq<8,4,1>", k_topk_filter_seq<8, 4, 1>, grid);
        }
        // the same bytes as 64-bit and 32-bit codes (4 x / 8 x the items; thresholds low enough for a few hundred candidates)
        const int64_t R8 = R;
        auto runshort = [&](const char* name, auto kern, int grid, int64_t Rs, uint32_t ths) {
            CK(hipMemcpy(t_est, &ths, 4, hipMemcpyHostToDevice));
            for (int i = 0; i < 2 * ngal; ++i) { CK(hipMemsetAsync(cnt, 0, 4096, 0)); hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, (const uint32_t*)dq, (const uint32_t*)gal[i % ngal], 1, Rs, (const uint32_t*)t_est, (const uint32_t*)bnd, cnt, cand); }
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, (const uint32_t*)dq, (const uint32_t*)gal[i % ngal], 1, Rs, (const uint32_t*)t_est, (const uint32_t*)bnd, cnt, cand);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double t = ms / iters * 1e-3;
            unsigned got = 0;
            CK(hipMemcpy(&got, cnt, 4, hipMemcpyDeviceToHost));
            printf("%-28s grid %5d  %7.2f us  %5.2f TB/s  (%.3f of 8)  candidates %u (over %d launches)\n", name, grid, t * 1e6, bytes / t / 1e12, bytes / t / 8e12, got, iters);
        };
        for (int grid : {512, 1024}) {
            runshort("LIB k_topk_filter_short<2,4,1>", k_topk_filter_short<2, 4, 1>, grid, R8 * 4, (uint32_t)(getenv("LIB_THR2") ? atoi(getenv("LIB_THR2")) : 12));
            runshort("LIB k_topk_filter_short<1,4,1>", k_topk_filter_short<1, 4, 1>, grid, R8 * 8, 3u);
            runshort("LIB k_topk_filter_short<2,4,8>", k_topk_filter_short<2, 4, 8>, grid, R8 * 4, 12u);
            runshort("LIB k_topk_filter_short<1,4,8>", k_topk_filter_short<1, 4, 8>, grid, R8 * 8, 3u);
        }
    }
#endif
    for (int grid : {512, 1024, 2048}) {
        run("rec  ipt2 (shipped form)", k_rec<2, false>, grid);
        run("rec  ipt2 nt", k_rec<2, true>, grid);
        run("rec  ipt4", k_rec<4, false>, grid);
        run("rec  ipt4 nt", k_rec<4, true>, grid);
        run("seq  4 loads", k_seq<4, false>, grid);
        run("seq  4 loads nt", k_seq<4, true>, grid);
        run("seq  8 loads", k_seq<8, false>, grid);
        run("seq  8 loads nt", k_seq<8, true>, grid);
        run("chunk 4 loads", k_chunk<4, false>, grid);
        run("chunk 4 loads nt", k_chunk<4, true>, grid);
        run("chunk 8 loads nt", k_chunk<8, true>, grid);
    }
    return 0;
}
