// Quantise + bit-pack kernels: float codes / pair probabilities / multi-hot labels -> u32 words.
//
// One lane per (row, bit position) slot of the padded [n][W*32] grid; a wave covers 64 consecutive
// slots, so its two ballot halves ARE the two output words: reads are coalesced, one 4-byte store per
// half-wave.  HBM-bound: algorithmic bytes = 4*n*K read + n*W*4 (x2 with the zero plane) written.
#include "xmh_common.h"

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ void store_ballot_words(unsigned long long m, int lane, uint32_t* dst_word) {
    // lane 0 owns the low word, lane 32 the high word of the ballot
    if ((lane & 31) == 0) *dst_word = (lane == 0) ? (uint32_t)m : (uint32_t)(m >> 32);
}

__global__ __launch_bounds__(kBlock) void k_pack_sign(const float* __restrict__ codes, int64_t n, int K, int W,
                                                      const int64_t* __restrict__ row_index,
                                                      uint32_t* __restrict__ bits, uint32_t* __restrict__ zero,
                                                      int32_t* __restrict__ flags) {
    const int64_t slots = n * (int64_t)W * 32;
    const int lane = threadIdx.x & 63;
    for (int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x; s - lane < slots; s += (int64_t)gridDim.x * kBlock) {
        const bool in = s < slots;
        const int64_t row = in ? s / (W * 32) : 0;
        const int col = in ? (int)(s % (W * 32)) : 0;
        const bool valid = in && col < K;
        const float x = valid ? codes[row * K + col] : 0.0f;
        const unsigned long long mp = __ballot(valid && x > 0.0f);
        const unsigned long long mz = __ballot(!valid || x == 0.0f);       // padding counts as "zero" (dead bit)
        const unsigned long long real_zero = __ballot(valid && x == 0.0f);
        const unsigned long long other = __ballot(valid && x != 0.0f && fabsf(x) != 1.0f);   // incl. NaN
        if (in) {
            const int64_t drow = row_index ? row_index[row] : row;
            const int64_t w = drow * W + col / 32;
            store_ballot_words(mp, lane, bits + w);
            if (zero) store_ballot_words(mz, lane, zero + w);
        }
        if (flags && (real_zero | other) && lane == 0) atomicOr(flags, (real_zero ? 1 : 0) | (other ? 2 : 0));
    }
}

__global__ __launch_bounds__(kBlock) void k_pack_pair(const float2* __restrict__ probs, int64_t n, int K, int W,
                                                      const int64_t* __restrict__ row_index,
                                                      uint32_t* __restrict__ bits) {
    const int64_t slots = n * (int64_t)W * 32;
    const int lane = threadIdx.x & 63;
    for (int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x; s - lane < slots; s += (int64_t)gridDim.x * kBlock) {
        const bool in = s < slots;
        const int64_t row = in ? s / (W * 32) : 0;
        const int col = in ? (int)(s % (W * 32)) : 0;
        const bool valid = in && col < K;
        float2 p = make_float2(0.f, 0.f);
        if (valid) p = probs[row * K + col];
        const unsigned long long mp = __ballot(valid && p.y > p.x);          // strict: a tie is -1 (argmax -> 0)
        if (in) {
            const int64_t drow = row_index ? row_index[row] : row;
            store_ballot_words(mp, lane, bits + drow * W + col / 32);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_pack_labels(const T* __restrict__ L, int64_t n, int C, int Lw,
                                                        uint32_t* __restrict__ lab) {
    const int64_t slots = n * (int64_t)Lw * 32;
    const int lane = threadIdx.x & 63;
    for (int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x; s - lane < slots; s += (int64_t)gridDim.x * kBlock) {
        const bool in = s < slots;
        const int64_t row = in ? s / (Lw * 32) : 0;
        const int col = in ? (int)(s % (Lw * 32)) : 0;
        const bool on = in && col < C && L[row * C + col] > (T)0;
        const unsigned long long m = __ballot(on);
        if (in) store_ballot_words(m, lane, lab + row * Lw + col / 32);
    }
}

__global__ __launch_bounds__(kBlock) void k_unpack(const uint32_t* __restrict__ bits, const uint32_t* __restrict__ zero,
                                                   int64_t n, int K, int W, float* __restrict__ out) {
    const int64_t total = n * (int64_t)K;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
        const int64_t row = e / K;
        const int col = (int)(e % K);
        const uint32_t b = bits[row * W + col / 32] >> (col & 31) & 1u;
        const uint32_t z = zero ? (zero[row * W + col / 32] >> (col & 31) & 1u) : 0u;
        out[e] = z ? 0.0f : (b ? 1.0f : -1.0f);
    }
}

inline int grid_for(int64_t work) {
    int64_t g = xmh::ceil_div(work, kBlock);
    const int64_t cap = (int64_t)xmh::device_cu_count() * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int xmh_pack_sign(const float* codes, int64_t n, int K, const int64_t* row_index, uint32_t* bits,
                             uint32_t* zero, int32_t* flags, xmh_stream_t stream) {
    if (n < 0 || K <= 0) return xmh::fail(XMH_EINVAL, "xmh_pack_sign: bad shape n=%lld K=%d", (long long)n, K);
    if (n == 0) return XMH_OK;
    if (!codes || !bits) return xmh::fail(XMH_EINVAL, "xmh_pack_sign: null pointer");
    const int W = (K + 31) / 32;
    hipLaunchKernelGGL(k_pack_sign, dim3(grid_for(n * W * 32)), dim3(kBlock), 0, xmh::as_stream(stream), codes, n, K, W,
                       row_index, bits, zero, flags);
    XMH_LAUNCH_CHECK("xmh_pack_sign");
    return XMH_OK;
}

extern "C" int xmh_pack_pair_argmax(const float* probs, int64_t n, int K, const int64_t* row_index, uint32_t* bits,
                                    xmh_stream_t stream) {
    if (n < 0 || K <= 0) return xmh::fail(XMH_EINVAL, "xmh_pack_pair_argmax: bad shape n=%lld K=%d", (long long)n, K);
    if (n == 0) return XMH_OK;
    if (!probs || !bits) return xmh::fail(XMH_EINVAL, "xmh_pack_pair_argmax: null pointer");
    const int W = (K + 31) / 32;
    hipLaunchKernelGGL(k_pack_pair, dim3(grid_for(n * W * 32)), dim3(kBlock), 0, xmh::as_stream(stream),
                       reinterpret_cast<const float2*>(probs), n, K, W, row_index, bits);
    XMH_LAUNCH_CHECK("xmh_pack_pair_argmax");
    return XMH_OK;
}

extern "C" int xmh_unpack_pm1(const uint32_t* bits, const uint32_t* zero, int64_t n, int K, float* out,
                              xmh_stream_t stream) {
    if (n < 0 || K <= 0) return xmh::fail(XMH_EINVAL, "xmh_unpack_pm1: bad shape n=%lld K=%d", (long long)n, K);
    if (n == 0) return XMH_OK;
    if (!bits || !out) return xmh::fail(XMH_EINVAL, "xmh_unpack_pm1: null pointer");
    const int W = (K + 31) / 32;
    hipLaunchKernelGGL(k_unpack, dim3(grid_for(n * K)), dim3(kBlock), 0, xmh::as_stream(stream), bits, zero, n, K, W, out);
    XMH_LAUNCH_CHECK("xmh_unpack_pm1");
    return XMH_OK;
}

extern "C" int xmh_pack_labels(const void* labels, int dt, int64_t n, int C, uint32_t* lab, xmh_stream_t stream) {
    if (n < 0 || C <= 0) return xmh::fail(XMH_EINVAL, "xmh_pack_labels: bad shape n=%lld C=%d", (long long)n, C);
    if (n == 0) return XMH_OK;
    if (!labels || !lab) return xmh::fail(XMH_EINVAL, "xmh_pack_labels: null pointer");
    const int Lw = (C + 31) / 32;
    const dim3 grid(grid_for(n * Lw * 32)), block(kBlock);
    hipStream_t st = xmh::as_stream(stream);
    switch (dt) {
        case XMH_DT_F32: hipLaunchKernelGGL(k_pack_labels<float>, grid, block, 0, st, (const float*)labels, n, C, Lw, lab); break;
        case XMH_DT_I64: hipLaunchKernelGGL(k_pack_labels<int64_t>, grid, block, 0, st, (const int64_t*)labels, n, C, Lw, lab); break;
        case XMH_DT_I32: hipLaunchKernelGGL(k_pack_labels<int32_t>, grid, block, 0, st, (const int32_t*)labels, n, C, Lw, lab); break;
        case XMH_DT_U8: hipLaunchKernelGGL(k_pack_labels<uint8_t>, grid, block, 0, st, (const uint8_t*)labels, n, C, Lw, lab); break;
        default: return xmh::fail(XMH_EINVAL, "xmh_pack_labels: unknown dtype code %d", dt);
    }
    XMH_LAUNCH_CHECK("xmh_pack_labels");
    return XMH_OK;
}
