// Float "similarities" of common/calc_utils.py on un-quantised inputs (UMoED-style raw tanh codes, the
// DCMHT loss inputs): everything that is a contraction goes through xmh_gemm_nt_f32; this file holds the row-wise
// pieces around it and the small-set float ranking used when calc_map_k is handed non-binary codes.
//
//   xmh_row_l2normalize   a / a.norm(dim=-1, keepdim=True)  (no eps, cosine_similarity :38-49)
//   xmh_pairwise_l2       torch.cdist(a, b, p=2) from the Gram matrix: sqrt(max(|a|^2 + |b|^2 - 2 a.b, 0)) (:28-36)
//   xmh_affine_inplace    y = alpha * x + beta  (0.5 * (K - q.r), calc_hammingDist :51-56 on float codes)
//   xmh_float_rank_ap     calc_map_k ranking for float distances: rank of each relevant item by direct counting under
//                         the (distance, index) order -- O(R * n_rel) per query, meant for small evaluation sets; the
//                         bit-packed scan (xmh_scan.hip) is the production path.
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

// one block per query: relevant items are spread over the threads; each counts its rank and ordinal directly.
__global__ __launch_bounds__(256) void k_float_rank_ap(const float* __restrict__ dist, const uint32_t* __restrict__ qlab,
                                                       const uint32_t* __restrict__ rlab, int64_t R, int Lw, int64_t kcap,
                                                       double* __restrict__ ap_sum, int32_t* __restrict__ cap_out) {
    __shared__ double part[256];
    __shared__ int nrel_s;
    const int64_t q = blockIdx.x;
    const float* d = dist + q * R;
    const uint32_t* ql = qlab + q * Lw;
    auto rel = [&](int64_t r) {
        uint32_t hit = 0;
        for (int w = 0; w < Lw; ++w) hit |= ql[w] & rlab[r * Lw + w];
        return hit != 0;
    };
    if (threadIdx.x == 0) nrel_s = 0;
    __syncthreads();
    int mine = 0;
    for (int64_t r = threadIdx.x; r < R; r += 256) mine += rel(r);
    atomicAdd(&nrel_s, mine);
    __syncthreads();
    const int nrel = nrel_s;
    const int64_t cap = (kcap > 0 && kcap < nrel) ? kcap : nrel;
    double s = 0.0;
    for (int64_t r = threadIdx.x; r < R; r += 256) {
        if (!rel(r)) continue;
        const float dr = d[r];
        int64_t rank = 1, ord = 1;
        for (int64_t j = 0; j < R; ++j) {
            const float dj = d[j];
            const bool before = dj < dr || (dj == dr && j < r);
            if (before) {
                ++rank;
                ord += rel(j);
            }
        }
        if (ord <= cap) s += (double)((float)ord / (float)rank);
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        ap_sum[q] = part[0];
        cap_out[q] = (int32_t)cap;
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

extern "C" int xmh_float_rank_ap(const float* dist, const uint32_t* qlab, const uint32_t* rlab, int64_t Q, int64_t R, int C,
                                 int64_t k, double* ap_sum, int32_t* cap, xmh_stream_t stream) {
    if (Q <= 0 || R <= 0 || C <= 0) return xmh::fail(XMH_EINVAL, "xmh_float_rank_ap: bad shape");
    if (!dist || !qlab || !rlab || !ap_sum || !cap) return xmh::fail(XMH_EINVAL, "xmh_float_rank_ap: null pointer");
    hipLaunchKernelGGL(k_float_rank_ap, dim3((unsigned)Q), dim3(256), 0, xmh::as_stream(stream), dist, qlab, rlab, R, (C + 31) / 32, k, ap_sum, cap);
    XMH_LAUNCH_CHECK("xmh_float_rank_ap");
    return XMH_OK;
}
