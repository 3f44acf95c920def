// Loss forward of the DCMHT objective (reference models/DCMHT/DCMHT.py:72-149): the [B, B] similarity terms and the
// quantisation term, computed straight from the head outputs without materialising a [B, B] matrix.
//
//   xmh_pair_similarity_loss   similarity_loss (:72-98) for one pair of code matrices a, b [B, D]:
//                              euclidean: s = ||a_i - b_j||; positive = mean((s L)^2); negative = mean((m (1 - L) - min(s (1 - L), m))^2)
//                                         with L = (labels_i . labels_j > 0), m = sqrt(2 K vartheta);
//                              cosine:    s = clip(cos(a_i, b_j), t, 1 - t); both outputs mean(-L log s - (1 - L) log(1 - s))
//   xmh_quant_loss             soft_argmax_hash_loss (:100-105): 1 - mean((2 c - 1)^2)
//   xmh_pair_similarity_loss_grad   d(positive + negative)/da of the same term (what autograd returns for the reference's expression):
//                              euclidean: grad a_i = sum_j r_ij (a_i - b_j),  r_ij = 2/B^2 (L - (m/s - 1)(1 - L)[s <= m]), 0 where s = 0
//                                         (torch.cdist's backward masks zero distances the same way);
//                              cosine:    grad a_i = sum_j h_ij (b^_j - c_ij a^_i) / |a_i|,  h_ij = 2/B^2 (-L/s + (1 - L)/(1 - s))[t <= c_ij <= 1 - t]
//                              The gradient with respect to b is the same call with a and b swapped (s, c and L are symmetric), and
//                              for a term with a == b the caller passes scale = 2.
//   xmh_quant_loss_grad        d/dcode = -4 (2 c - 1) / n
// Bound: B = 128 rows, D <= 4096: a few hundred KB of L2-resident reads -- launch-latency, not bandwidth.  Sums are kept in fp64
// (forward) / fp32 (gradient rows).  The backward of the encoders and the optimiser step stay outside this path.
#include "xmh_common.h"

namespace {

__device__ __forceinline__ double block_sum(double v, double* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sh[w];
    __syncthreads();
    return s;                                                    // valid on thread 0
}

// one block per row i of a; threads take columns j
__global__ __launch_bounds__(256) void k_pair_similarity_loss(const float* __restrict__ a, const float* __restrict__ b, int B, int D,
                                                              const uint32_t* __restrict__ lab, int Lw, int cosine, float max_value,
                                                              float threshold, double* __restrict__ out2) {
    extern __shared__ __attribute__((aligned(16))) float row[];  // a_i
    __shared__ double sh[8];
    const int i = blockIdx.x;
    for (int c = threadIdx.x; c < D; c += blockDim.x) row[c] = a[(int64_t)i * D + c];
    __syncthreads();
    float na = 0.0f;
    if (cosine)
        for (int c = 0; c < D; ++c) na = fmaf(row[c], row[c], na);
    double pos = 0.0, neg = 0.0;
    for (int j = threadIdx.x; j < B; j += blockDim.x) {
        const float* bj = b + (int64_t)j * D;
        bool rel = false;
        for (int w = 0; w < Lw; ++w) rel |= (lab[(int64_t)i * Lw + w] & lab[(int64_t)j * Lw + w]) != 0u;
        const float L = rel ? 1.0f : 0.0f;
        if (cosine) {
            float dot = 0.0f, nb = 0.0f;
            for (int c = 0; c < D; ++c) {
                dot = fmaf(row[c], bj[c], dot);
                nb = fmaf(bj[c], bj[c], nb);
            }
            float s = dot / (sqrtf(na) * sqrtf(nb));
            s = fminf(fmaxf(s, threshold), 1.0f - threshold);
            const float l = -L * logf(s) - (1.0f - L) * logf(1.0f - s);
            pos += (double)l;
            neg += (double)l;
        } else {
            float d2 = 0.0f;
            for (int c = 0; c < D; ++c) {
                const float d = row[c] - bj[c];
                d2 = fmaf(d, d, d2);
            }
            const float s = sqrtf(d2);
            const float p = s * L;
            float n = fminf(s * (1.0f - L), max_value);
            n = max_value * (1.0f - L) - n;
            pos += (double)p * (double)p;
            neg += (double)n * (double)n;
        }
    }
    const double ps = block_sum(pos, sh), ns = block_sum(neg, sh);
    if (threadIdx.x == 0) {
        atomicAdd(&out2[0], ps / ((double)B * (double)B));
        atomicAdd(&out2[1], ns / ((double)B * (double)B));
    }
}

__global__ __launch_bounds__(256) void k_quant_loss(const float* __restrict__ code, int64_t n, double* __restrict__ out) {
    __shared__ double sh[8];
    double acc = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const float t = 2.0f * code[e] - 1.0f;
        acc += (double)(t * t);
    }
    const double s = block_sum(acc, sh);
    if (threadIdx.x == 0) atomicAdd(out, -s / (double)n);
}

// gradient of (positive + negative) with respect to row i of a: one block per row.  Phase 1 (threads over j): the coefficient of
// every pair into LDS; phase 2 (threads over columns, coalesced rows of b): the weighted sum.  `up` (device, may be null) is the
// upstream gradient of the loss, multiplied in so that backward() needs no host synchronisation.
__global__ __launch_bounds__(256) void k_pair_similarity_grad(const float* __restrict__ a, const float* __restrict__ b, int B, int D,
                                                              const uint32_t* __restrict__ lab, int Lw, int cosine, float max_value,
                                                              float threshold, float scale, const float* __restrict__ up,
                                                              float* __restrict__ grad, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float row[];  // a_i [D], then coef [B]
    __shared__ double sh[8];
    float* coef = row + D;
    const int i = blockIdx.x;
    for (int c = threadIdx.x; c < D; c += blockDim.x) row[c] = a[(int64_t)i * D + c];
    __syncthreads();
    const float w = 2.0f / ((float)B * (float)B);
    float na = 0.0f;
    if (cosine)
        for (int c = 0; c < D; ++c) na = fmaf(row[c], row[c], na);
    const float inv_na = cosine ? 1.0f / sqrtf(na) : 0.0f;
    double hc = 0.0;                                             // cosine: sum_j h_ij c_ij
    for (int j = threadIdx.x; j < B; j += blockDim.x) {
        const float* bj = b + (int64_t)j * D;
        bool rel = false;
        for (int x = 0; x < Lw; ++x) rel |= (lab[(int64_t)i * Lw + x] & lab[(int64_t)j * Lw + x]) != 0u;
        float cf = 0.0f;
        if (cosine) {
            float dot = 0.0f, nb = 0.0f;
            for (int c = 0; c < D; ++c) {
                dot = fmaf(row[c], bj[c], dot);
                nb = fmaf(bj[c], bj[c], nb);
            }
            const float inv_nb = 1.0f / sqrtf(nb);
            const float cs = dot * inv_na * inv_nb;
            if (cs >= threshold && cs <= 1.0f - threshold) {     // clamp passes the gradient on the closed interval
                const float h = w * (rel ? -1.0f / cs : 1.0f / (1.0f - cs));
                cf = h * inv_na * inv_nb;
                hc += (double)h * (double)cs;
            }
        } else {
            float d2 = 0.0f;
            for (int c = 0; c < D; ++c) {
                const float d = row[c] - bj[c];
                d2 = fmaf(d, d, d2);
            }
            const float s = sqrtf(d2);
            if (s > 0.0f) cf = rel ? w : (s <= max_value ? -w * (max_value / s - 1.0f) : 0.0f);
        }
        coef[j] = cf;
    }
    const double hcs = block_sum(hc, sh);                        // also the barrier that publishes coef[]
    if (threadIdx.x == 0) sh[0] = hcs;
    __syncthreads();
    const float self = cosine ? (float)(sh[0] * (double)inv_na * (double)inv_na) : 0.0f;
    const float g = scale * (up ? up[0] : 1.0f);
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        float acc = 0.0f;
        if (cosine) {
            for (int j = 0; j < B; ++j) acc = fmaf(coef[j], b[(int64_t)j * D + c], acc);
            acc -= row[c] * self;
        } else {
            for (int j = 0; j < B; ++j) acc = fmaf(coef[j], row[c] - b[(int64_t)j * D + c], acc);
        }
        float* o = grad + (int64_t)i * D + c;
        *o = accumulate ? *o + g * acc : g * acc;
    }
}

__global__ __launch_bounds__(256) void k_quant_grad(const float* __restrict__ code, int64_t n, float scale, const float* __restrict__ up,
                                                    float* __restrict__ grad, int accumulate) {
    const float g = scale * (up ? up[0] : 1.0f) * (-4.0f / (float)n);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const float v = g * (2.0f * code[e] - 1.0f);
        grad[e] = accumulate ? grad[e] + v : v;
    }
}

__global__ void k_set_double(double* p, double v0, double v1, int n) {
    if (threadIdx.x == 0) {
        p[0] = v0;
        if (n > 1) p[1] = v1;
    }
}

}  // namespace

extern "C" int xmh_pair_similarity_loss(const float* a, const float* b, int64_t B, int D, const uint32_t* lab, int C, int cosine,
                                        float max_value, float threshold, double* out2, xmh_stream_t stream) {
    XMH_RANGE("xmh_pair_similarity_loss");
    if (B <= 0 || D <= 0 || C <= 0) return xmh::fail(XMH_EINVAL, "xmh_pair_similarity_loss: bad shape B=%lld D=%d C=%d", (long long)B, D, C);
    if (!a || !b || !lab || !out2) return xmh::fail(XMH_EINVAL, "xmh_pair_similarity_loss: null pointer");
    if (D > 12288) return xmh::fail(XMH_ENOTSUP, "xmh_pair_similarity_loss: D=%d > 12288 (one row must fit LDS)", D);
    if (B >= (1ll << 31)) return xmh::fail(XMH_ENOTSUP, "xmh_pair_similarity_loss: B too large");
    hipStream_t st = xmh::as_stream(stream);
    hipLaunchKernelGGL(k_set_double, dim3(1), dim3(64), 0, st, out2, 0.0, 0.0, 2);
    hipLaunchKernelGGL(k_pair_similarity_loss, dim3((unsigned)B), dim3(256), (size_t)D * 4, st, a, b, (int)B, D, lab, (C + 31) / 32, cosine,
                       max_value, threshold, out2);
    XMH_LAUNCH_CHECK("xmh_pair_similarity_loss");
    return XMH_OK;
}

extern "C" int xmh_quant_loss(const float* code, int64_t n, double* out, xmh_stream_t stream) {
    XMH_RANGE("xmh_quant_loss");
    if (n <= 0) return xmh::fail(XMH_EINVAL, "xmh_quant_loss: empty input");
    if (!code || !out) return xmh::fail(XMH_EINVAL, "xmh_quant_loss: null pointer");
    hipStream_t st = xmh::as_stream(stream);
    hipLaunchKernelGGL(k_set_double, dim3(1), dim3(64), 0, st, out, 1.0, 0.0, 1);
    int64_t grid = xmh::ceil_div(n, 256 * 8);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(k_quant_loss, dim3((unsigned)grid), dim3(256), 0, st, code, n, out);
    XMH_LAUNCH_CHECK("xmh_quant_loss");
    return XMH_OK;
}

extern "C" int xmh_pair_similarity_loss_grad(const float* a, const float* b, int64_t B, int D, const uint32_t* lab, int C, int cosine,
                                             float max_value, float threshold, float scale, const float* upstream, float* grad_a,
                                             int accumulate, xmh_stream_t stream) {
    XMH_RANGE("xmh_pair_similarity_loss_grad");
    if (B <= 0 || D <= 0 || C <= 0) return xmh::fail(XMH_EINVAL, "xmh_pair_similarity_loss_grad: bad shape B=%lld D=%d C=%d", (long long)B, D, C);
    if (!a || !b || !lab || !grad_a) return xmh::fail(XMH_EINVAL, "xmh_pair_similarity_loss_grad: null pointer");
    if (B >= (1ll << 31)) return xmh::fail(XMH_ENOTSUP, "xmh_pair_similarity_loss_grad: B too large");
    const size_t lds = ((size_t)D + (size_t)B) * 4;
    if (lds > 64 * 1024) return xmh::fail(XMH_ENOTSUP, "xmh_pair_similarity_loss_grad: (D + B) * 4 = %zu bytes > 64 KB of LDS", lds);
    hipStream_t st = xmh::as_stream(stream);
    hipLaunchKernelGGL(k_pair_similarity_grad, dim3((unsigned)B), dim3(256), lds, st, a, b, (int)B, D, lab, (C + 31) / 32, cosine, max_value,
                       threshold, scale, upstream, grad_a, accumulate);
    XMH_LAUNCH_CHECK("xmh_pair_similarity_loss_grad");
    return XMH_OK;
}

extern "C" int xmh_quant_loss_grad(const float* code, int64_t n, float scale, const float* upstream, float* grad, int accumulate,
                                   xmh_stream_t stream) {
    XMH_RANGE("xmh_quant_loss_grad");
    if (n <= 0) return xmh::fail(XMH_EINVAL, "xmh_quant_loss_grad: empty input");
    if (!code || !grad) return xmh::fail(XMH_EINVAL, "xmh_quant_loss_grad: null pointer");
    hipStream_t st = xmh::as_stream(stream);
    int64_t grid = xmh::ceil_div(n, 256 * 8);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(k_quant_grad, dim3((unsigned)grid), dim3(256), 0, st, code, n, scale, upstream, grad, accumulate);
    XMH_LAUNCH_CHECK("xmh_quant_loss_grad");
    return XMH_OK;
}
