// Float "similarities" of common/calc_utils.py on un-quantised inputs (UMoED-style raw tanh codes, the
// DCMHT loss inputs): everything that is a contraction goes through xmh_gemm_nt_f32; this file holds the row-wise
// pieces around it and the small-set float ranking used when calc_map_k is handed non-binary codes.
//
//   xmh_row_l2normalize   a / a.norm(dim=-1, keepdim=True)  (no eps, cosine_similarity :38-49)
//   xmh_pairwise_l2       torch.cdist(a, b, p=2) from the Gram matrix: sqrt(max(|a|^2 + |b|^2 - 2 a.b, 0)) (:28-36)
//   xmh_affine_inplace    y = alpha * x + beta  (0.5 * (K - q.r), calc_hammingDist :51-56 on float codes)
//   xmh_gemm_f32_sort_map calc_map_k for float "codes" as the reference computes it (:72-89): distances by exact-fp32 GEMM, ONE
//                         stable sort per query row (a segmented LSD radix sort of the order-preserving 32-bit image of the
//                         distance, payload = gallery index | relevance << 31: ties keep index order = torch.sort(stable=True)),
//                         then one pass over the sorted row for sum(ordinal / rank).  Replaces the round-2 comparison-counting
//                         kernel (O(R * n_rel) per query: 8e12 comparisons at the COCO shape).
#include "xmh_common.h"

namespace {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ __launch_bounds__(256) void k_row_norm(const float* __restrict__ x, int64_t rows, int D, float* __restrict__ y,
                                                  float* __restrict__ sqnorm) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float s = 0.0f;
    for (int c = lane; c < D; c += 64) s = fmaf(x[r * D + c], x[r * D + c], s);
    s = wave_sum_f(s);
    if (sqnorm && lane == 0) sqnorm[r] = s;
    if (y) {
        const float n = sqrtf(s);
        for (int c = lane; c < D; c += 64) y[r * D + c] = x[r * D + c] / n;
    }
}

__global__ __launch_bounds__(256) void k_l2_from_gram(float* __restrict__ g, const float* __restrict__ na, const float* __restrict__ nb,
                                                      int64_t M, int64_t N) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < M * N; e += (int64_t)gridDim.x * 256) {
        const float d2 = na[e / N] + nb[e % N] - 2.0f * g[e];
        g[e] = sqrtf(fmaxf(d2, 0.0f));
    }
}

__global__ __launch_bounds__(256) void k_affine_inplace(float* __restrict__ x, int64_t n, float alpha, float beta) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) x[e] = fmaf(alpha, x[e], beta);
}

// ---- segmented radix sort of the distance rows + AP pass (round 6) ------------------------------------------------------------------
// One block of 16 waves per query row; wave w owns a contiguous SEGMENT of the row.  An LSD pass over an 8-bit digit is: per-wave
// histogram of the segment (LDS atomics), an exclusive prefix over (digit, wave) -- digit-major, wave-minor, which is what makes the
// pass stable across segments --, then every wave scatters its segment in order: the 64 lanes of a step find the lanes that share
// their digit with eight ballots (multi-split), take consecutive slots behind the wave's running counter of that digit and the lowest
// lane of each group moves the counter on.  LDS operations of one wave execute in order, so the next step sees the counters of this
// one; nothing waits on another wave.  Four passes; the first reads the distances (and the labels: the relevance bit rides in the
// payload's top bit, so the AP pass gathers nothing), the last writes payloads only.
constexpr int kSortWaves = 16;
constexpr int kSortThreads = 64 * kSortWaves;

__device__ __forceinline__ uint32_t float_order_key(float d) {
    const uint32_t u = __float_as_uint(d + 0.0f);             // -0.0 -> +0.0: torch.sort holds them equal
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);       // ascending floats <=> ascending unsigned keys
}

template <bool FIRST, bool LAST>
__global__ __launch_bounds__(kSortThreads) void k_fsort_pass(const float* __restrict__ gram, float alpha, float beta,
                                                             const uint32_t* __restrict__ in_key, const uint32_t* __restrict__ in_pay,
                                                             uint32_t* __restrict__ out_key, uint32_t* __restrict__ out_pay, int64_t R, int shift,
                                                             const uint32_t* __restrict__ qlab, const uint32_t* __restrict__ rlab, int Lw) {
    __shared__ uint32_t cnt[kSortWaves][256];
    __shared__ uint32_t wtot[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t row = blockIdx.x;
    const int64_t seg = ((R + kSortWaves - 1) / kSortWaves + 63) & ~(int64_t)63;
    const int64_t lo = (int64_t)w * seg, hi = lo + seg < R ? lo + seg : R;
    const float* __restrict__ g = FIRST ? gram + row * R : nullptr;
    const uint32_t* __restrict__ ik = FIRST ? nullptr : in_key + row * R;
    const uint32_t* __restrict__ ip = FIRST ? nullptr : in_pay + row * R;
    auto key_at = [&](int64_t e) -> uint32_t { return FIRST ? float_order_key(fmaf(alpha, g[e], beta)) : ik[e]; };
    for (int e = threadIdx.x; e < kSortWaves * 256; e += kSortThreads) (&cnt[0][0])[e] = 0u;
    __syncthreads();
    for (int64_t e = lo + lane; e < hi; e += 64) atomicAdd(&cnt[w][(key_at(e) >> shift) & 0xFFu], 1u);
    __syncthreads();
    uint32_t run = 0;
    if (threadIdx.x < 256) {                                  // digit t: exclusive prefix over the waves, then over the digits
        for (int x = 0; x < kSortWaves; ++x) {
            const uint32_t c = cnt[x][threadIdx.x];
            cnt[x][threadIdx.x] = run;
            run += c;
        }
        uint32_t incl = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(incl, o, 64);
            if (lane >= o) incl += u;
        }
        if (lane == 63) wtot[w] = incl;
        run = incl - run;                                     // exclusive inside the wave
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        for (int x = 0; x < w; ++x) run += wtot[x];
        for (int x = 0; x < kSortWaves; ++x) cnt[x][threadIdx.x] += run;
    }
    __syncthreads();
    const uint32_t* __restrict__ ql = qlab + row * Lw;
    uint32_t* __restrict__ ok = LAST ? nullptr : out_key + row * R;
    uint32_t* __restrict__ op = out_pay + row * R;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int64_t e0 = lo; e0 < hi; e0 += 64) {
        const int64_t e = e0 + lane;
        const bool ok_ = e < hi;
        uint32_t key = 0, pay = 0;
        if (ok_) {
            key = key_at(e);
            if constexpr (FIRST) {
                uint32_t hit = 0;
                for (int x = 0; x < Lw; ++x) hit |= ql[x] & rlab[e * Lw + x];
                pay = (uint32_t)e | (hit ? 0x80000000u : 0u);
            } else pay = ip[e];
        }
        const uint32_t digit = (key >> shift) & 0xFFu;
        unsigned long long same = __ballot(ok_);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long m = __ballot((digit >> b) & 1u);
            same &= ((digit >> b) & 1u) ? m : ~m;
        }
        if (ok_) {
            const uint32_t base = cnt[w][digit];
            const uint32_t pos = base + (uint32_t)__popcll(same & below);
            if ((same & below) == 0ull) cnt[w][digit] = base + (uint32_t)__popcll(same);
            if constexpr (!LAST) ok[pos] = key;
            op[pos] = pay;
        }
    }
}

// sum(ordinal / rank) over the sorted row: payload top bit = relevant.  cap = min(k, n_rel): an ordinal never exceeds n_rel, so the
// reference's "first `total` relevant ranks" (:86-88) is ordinal <= k.
__global__ __launch_bounds__(kSortThreads) void k_fsort_ap(const uint32_t* __restrict__ pay, int64_t R, int64_t kcap, double* __restrict__ ap_sum,
                                                           int32_t* __restrict__ cap_out) {
    __shared__ uint32_t wrel[kSortWaves];
    __shared__ double wsum[kSortWaves];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t* __restrict__ p = pay + (int64_t)blockIdx.x * R;
    const int64_t seg = ((R + kSortWaves - 1) / kSortWaves + 63) & ~(int64_t)63;
    const int64_t lo = (int64_t)w * seg, hi = lo + seg < R ? lo + seg : R;
    uint32_t mine = 0;
    for (int64_t e = lo + lane; e < hi; e += 64) mine += p[e] >> 31;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if (lane == 0) wrel[w] = mine;
    __syncthreads();
    uint32_t before = 0, nrel = 0;
    for (int x = 0; x < kSortWaves; ++x) {
        if (x < w) before += wrel[x];
        nrel += wrel[x];
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    double s = 0.0;
    for (int64_t e0 = lo; e0 < hi; e0 += 64) {
        const int64_t e = e0 + lane;
        const bool rel = e < hi && (p[e] >> 31);
        const unsigned long long m = __ballot(rel);
        if (rel) {
            const uint32_t ord = before + (uint32_t)__popcll(m & below) + 1u;
            if (kcap <= 0 || (int64_t)ord <= kcap) s += (double)((float)ord / (float)(e + 1));
        }
        before += (uint32_t)__popcll(m);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) wsum[w] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int x = 0; x < kSortWaves; ++x) t += wsum[x];
        ap_sum[blockIdx.x] = t;
        cap_out[blockIdx.x] = (int32_t)((kcap > 0 && kcap < (int64_t)nrel) ? kcap : (int64_t)nrel);
    }
}

inline int grid1(int64_t work) {
    int64_t g = xmh::ceil_div(work, 256);
    const int64_t cap = (int64_t)xmh::device_cu_count() * 16;
    return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int xmh_row_l2normalize(const float* x, int64_t rows, int D, float* y, float* sqnorm, xmh_stream_t stream) {
    if (rows < 0 || D <= 0) return xmh::fail(XMH_EINVAL, "xmh_row_l2normalize: bad shape");
    if (rows == 0) return XMH_OK;
    if (!x || (!y && !sqnorm)) return xmh::fail(XMH_EINVAL, "xmh_row_l2normalize: null pointer");
    hipLaunchKernelGGL(k_row_norm, dim3((unsigned)xmh::ceil_div(rows, 4)), dim3(256), 0, xmh::as_stream(stream), x, rows, D, y, sqnorm);
    XMH_LAUNCH_CHECK("xmh_row_l2normalize");
    return XMH_OK;
}

extern "C" int xmh_pairwise_l2_from_gram(float* gram_inout, const float* sqnorm_a, const float* sqnorm_b, int64_t M, int64_t N,
                                         xmh_stream_t stream) {
    if (M < 0 || N < 0) return xmh::fail(XMH_EINVAL, "xmh_pairwise_l2_from_gram: bad shape");
    if (M == 0 || N == 0) return XMH_OK;
    if (!gram_inout || !sqnorm_a || !sqnorm_b) return xmh::fail(XMH_EINVAL, "xmh_pairwise_l2_from_gram: null pointer");
    hipLaunchKernelGGL(k_l2_from_gram, dim3(grid1(M * N)), dim3(256), 0, xmh::as_stream(stream), gram_inout, sqnorm_a, sqnorm_b, M, N);
    XMH_LAUNCH_CHECK("xmh_pairwise_l2_from_gram");
    return XMH_OK;
}

extern "C" int xmh_affine_inplace(float* x, int64_t n, float alpha, float beta, xmh_stream_t stream) {
    if (n < 0) return xmh::fail(XMH_EINVAL, "xmh_affine_inplace: bad shape");
    if (n == 0) return XMH_OK;
    if (!x) return xmh::fail(XMH_EINVAL, "xmh_affine_inplace: null pointer");
    hipLaunchKernelGGL(k_affine_inplace, dim3(grid1(n)), dim3(256), 0, xmh::as_stream(stream), x, n, alpha, beta);
    XMH_LAUNCH_CHECK("xmh_affine_inplace");
    return XMH_OK;
}

// ranking + AP of `rows` distance rows: gram[rows][R] holds q.r (alpha, beta turn it into the distance: 0.5 * (K - q.r))
static int float_sort_ap(const float* gram, float alpha, float beta, const uint32_t* qlab, const uint32_t* rlab, int64_t rows, int64_t R, int Lw,
                         int64_t k, uint32_t* ka, uint32_t* pa, uint32_t* kb, uint32_t* pb, double* ap_sum, int32_t* cap, hipStream_t st) {
    const dim3 grid((unsigned)rows), block(kSortThreads);
    hipLaunchKernelGGL((k_fsort_pass<true, false>), grid, block, 0, st, gram, alpha, beta, nullptr, nullptr, ka, pa, R, 0, qlab, rlab, Lw);
    hipLaunchKernelGGL((k_fsort_pass<false, false>), grid, block, 0, st, nullptr, 0.f, 0.f, ka, pa, kb, pb, R, 8, qlab, rlab, Lw);
    hipLaunchKernelGGL((k_fsort_pass<false, false>), grid, block, 0, st, nullptr, 0.f, 0.f, kb, pb, ka, pa, R, 16, qlab, rlab, Lw);
    hipLaunchKernelGGL((k_fsort_pass<false, true>), grid, block, 0, st, nullptr, 0.f, 0.f, ka, pa, nullptr, pb, R, 24, qlab, rlab, Lw);
    hipLaunchKernelGGL(k_fsort_ap, grid, block, 0, st, pb, R, k, ap_sum, cap);
    XMH_LAUNCH_CHECK("xmh_gemm_f32_sort_map");
    return XMH_OK;
}

// Workspace: per query row of a tile R * (4 gram + 16 two key/payload buffers) bytes.  The size asked for keeps a tile under 1.5 GB
// (Q = 500 x R = 117 218: one tile); any size from one row up is accepted and decides the tile.
extern "C" size_t xmh_gemm_f32_sort_ws_bytes(int64_t Q, int64_t R) {
    if (Q <= 0 || R <= 0) return 0;
    const size_t per_row = (size_t)R * 20 + 256;
    size_t rows = ((size_t)3 << 29) / per_row;
    if (rows < 1) rows = 1;
    if (rows > (size_t)Q) rows = (size_t)Q;
    return rows * per_row + 1024;
}

extern "C" int xmh_gemm_f32_sort_map(const float* qB, const float* rB, const uint32_t* qlab, const uint32_t* rlab, int64_t Q, int64_t R, int K,
                                     int C, int64_t k, void* ws, size_t ws_bytes, double* ap_sum, int32_t* cap, double* map_out,
                                     xmh_stream_t stream) {
    XMH_RANGE("xmh_gemm_f32_sort_map");
    if (Q <= 0 || R <= 0 || K <= 0 || C <= 0) return xmh::fail(XMH_EINVAL, "xmh_gemm_f32_sort_map: bad shape Q=%lld R=%lld K=%d C=%d", (long long)Q, (long long)R, K, C);
    if (R >= (1ll << 31)) return xmh::fail(XMH_ENOTSUP, "xmh_gemm_f32_sort_map: %lld gallery rows (the payload keeps 31 index bits)", (long long)R);
    if (!qB || !rB || !qlab || !rlab || !ws || !ap_sum || !cap) return xmh::fail(XMH_EINVAL, "xmh_gemm_f32_sort_map: null pointer");
    const size_t per_row = (size_t)R * 20 + 256;
    if (ws_bytes < per_row + 1024) return xmh::fail(XMH_EINVAL, "xmh_gemm_f32_sort_map: workspace too small (%zu < %zu: one row)", ws_bytes, per_row + 1024);
    int64_t tile = (int64_t)((ws_bytes - 1024) / per_row);
    if (tile > Q) tile = Q;
    char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
    const size_t plane = (((size_t)tile * R * 4) + 255) & ~(size_t)255;
    if (5 * plane + 256 > ws_bytes) --tile;                      // (alignment slack)
    if (tile < 1) return xmh::fail(XMH_EINVAL, "xmh_gemm_f32_sort_map: workspace too small");
    float* gram = reinterpret_cast<float*>(base);
    uint32_t *ka = reinterpret_cast<uint32_t*>(base + plane), *pa = reinterpret_cast<uint32_t*>(base + 2 * plane);
    uint32_t *kb = reinterpret_cast<uint32_t*>(base + 3 * plane), *pb = reinterpret_cast<uint32_t*>(base + 4 * plane);
    hipStream_t st = xmh::as_stream(stream);
    const int Lw = (C + 31) / 32;
    for (int64_t q0 = 0; q0 < Q; q0 += tile) {
        const int64_t rows = Q - q0 < tile ? Q - q0 : tile;
        // q.r^T with exact fp32 products (v_mfma_f32: the reference's mm on fp32 codes, calc_utils.py:51-56)
        if (const int rc = xmh_gemm_nt_f32(qB + q0 * K, K, rB, K, nullptr, nullptr, 0, gram, R, rows, R, K, XMH_ACT_NONE, XMH_PREC_F32, stream)) return rc;
        if (const int rc = float_sort_ap(gram, -0.5f, 0.5f * (float)K, qlab + q0 * Lw, rlab, rows, R, Lw, k, ka, pa, kb, pb, ap_sum + q0, cap + q0, st)) return rc;
    }
    if (map_out) return xmh_map_finalize(ap_sum, cap, Q, map_out, stream);
    return XMH_OK;
}

// the same ranking for a distance matrix the caller already holds (dist[Q][R], any float values): ws = Q * R * 16 bytes + 1 KB
extern "C" int xmh_float_sort_ap(const float* dist, const uint32_t* qlab, const uint32_t* rlab, int64_t Q, int64_t R, int C, int64_t k, void* ws,
                                 size_t ws_bytes, double* ap_sum, int32_t* cap, xmh_stream_t stream) {
    XMH_RANGE("xmh_float_sort_ap");
    if (Q <= 0 || R <= 0 || C <= 0) return xmh::fail(XMH_EINVAL, "xmh_float_sort_ap: bad shape");
    if (R >= (1ll << 31)) return xmh::fail(XMH_ENOTSUP, "xmh_float_sort_ap: %lld gallery rows (the payload keeps 31 index bits)", (long long)R);
    if (!dist || !qlab || !rlab || !ws || !ap_sum || !cap) return xmh::fail(XMH_EINVAL, "xmh_float_sort_ap: null pointer");
    const size_t plane = (((size_t)Q * R * 4) + 255) & ~(size_t)255;
    if (ws_bytes < 4 * plane + 256) return xmh::fail(XMH_EINVAL, "xmh_float_sort_ap: workspace too small (%zu < %zu)", ws_bytes, 4 * plane + 256);
    char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
    return float_sort_ap(dist, 1.0f, 0.0f, qlab, rlab, Q, R, (C + 31) / 32, k, reinterpret_cast<uint32_t*>(base), reinterpret_cast<uint32_t*>(base + plane),
                         reinterpret_cast<uint32_t*>(base + 2 * plane), reinterpret_cast<uint32_t*>(base + 3 * plane), ap_sum, cap, xmh::as_stream(stream));
}
