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

// Round 5: the slot-per-lane kernels above read 4 bytes per lane (256 bytes per wave instruction) and divide a 64-bit index per element:
// 4 M x 64 float codes went in at 1.6 TB/s, 0.75 at 16 bits (tools/bench_pack.py).  When the code length is a multiple of 32 the codes are
// one flat stream of n*K floats -> n*K bits with no padding: a lane loads 16 bytes (four columns, non-temporal) and holds their four
// sign bits as a nibble; three DPP steps (row_shl 1, 2, 4) gather the eight nibbles of a word into the lane that leads them, which
// stores it.  Same words, same zero plane, same flags.  (A first form took one ballot per component and let lanes 0..7 weave the bytes:
// 3.1-3.3 TB/s, bound by its ~150 instructions per KB.)
typedef float pack_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t gather8(uint32_t nib) {     // lane 8 k ends with the nibbles of lanes 8 k .. 8 k + 7, lowest first
    uint32_t x = nib | (uint32_t)__builtin_amdgcn_update_dpp(0, (int)nib, 0x101, 0xf, 0xf, true) << 4;     // row_shl:1: the next lane's
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x102, 0xf, 0xf, true) << 8;                      // row_shl:2
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x104, 0xf, 0xf, true) << 16;                     // row_shl:4
    return x;
}

// The reference's default code length, 16 bits (and 8): one word per ROW with 16 (8) live bits, i.e. four (two) 16-byte pieces per word and
// nothing else changes: the stream is still flat, the upper bits of the word are padding (0 in the sign plane, 1 = dead in the zero plane).
template <int PPW>
__global__ __launch_bounds__(kBlock) void k_pack_sign_short(const pack_f4* __restrict__ codes4, int64_t n, const int64_t* __restrict__ row_index,
                                                            uint32_t* __restrict__ bits, uint32_t* __restrict__ zero, int32_t* __restrict__ flags) {
    static_assert(PPW == 4 || PPW == 2, "16- or 8-bit codes");
    const int64_t total4 = n * PPW;
    const int lane = threadIdx.x & 63;
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i - lane < total4; i += (int64_t)gridDim.x * kBlock) {
        const bool in = i < total4;
        pack_f4 v = {0.f, 0.f, 0.f, 0.f};
        if (in) v = __builtin_nontemporal_load(codes4 + i);
        uint32_t np = 0, nz = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            np |= (v[j] > 0.0f ? 1u : 0u) << j;
            nz |= (v[j] == 0.0f ? 1u : 0u) << j;
            bad |= (in && v[j] == 0.0f ? 1 : 0) | (v[j] != 0.0f && fabsf(v[j]) != 1.0f ? 2 : 0);
        }
        uint32_t wp = np | (uint32_t)__builtin_amdgcn_update_dpp(0, (int)np, 0x101, 0xf, 0xf, true) << 4;
        uint32_t wz = nz | (uint32_t)__builtin_amdgcn_update_dpp(0, (int)nz, 0x101, 0xf, 0xf, true) << 4;
        if (PPW == 4) {
            wp |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wp, 0x102, 0xf, 0xf, true) << 8;
            wz |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wz, 0x102, 0xf, 0xf, true) << 8;
        }
        if ((lane & (PPW - 1)) == 0 && in) {
            const int64_t row = i / PPW;
            const int64_t dst = row_index ? row_index[row] : row;
            constexpr uint32_t live = PPW == 4 ? 0xffffu : 0xffu;
            bits[dst] = wp & live;
            if (zero) zero[dst] = (wz & live) | ~live;       // padding counts as "zero" (dead bit), like the slot kernel
        }
    }
    if (flags) {
        const int any = (__ballot(bad & 1) ? 1 : 0) | (__ballot(bad & 2) ? 2 : 0);
        if (any && lane == 0) atomicOr(flags, any);
    }
}

template <bool POW2>
__global__ __launch_bounds__(kBlock) void k_pack_sign_flat(const pack_f4* __restrict__ codes4, int64_t nwords, int W, int wshift,
                                                           const int64_t* __restrict__ row_index, uint32_t* __restrict__ bits,
                                                           uint32_t* __restrict__ zero, int32_t* __restrict__ flags) {
    const int64_t total4 = nwords * 8;                       // 16-byte pieces
    const int lane = threadIdx.x & 63;
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i - lane < total4; i += (int64_t)gridDim.x * kBlock) {
        const bool in = i < total4;
        pack_f4 v = {0.f, 0.f, 0.f, 0.f};
        if (in) v = __builtin_nontemporal_load(codes4 + i);
        uint32_t np = 0, nz = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            np |= (v[j] > 0.0f ? 1u : 0u) << j;
            nz |= (v[j] == 0.0f ? 1u : 0u) << j;
            bad |= (in && v[j] == 0.0f ? 1 : 0) | (v[j] != 0.0f && fabsf(v[j]) != 1.0f ? 2 : 0);      // (2: incl. NaN)
        }
        const uint32_t wp = gather8(np), wz = gather8(nz);
        if ((lane & 7) == 0 && in) {                         // i % 8 == 0: this lane leads a word
            const int64_t wi = i >> 3;
            int64_t dst = wi;
            if (row_index) {
                const int64_t row = POW2 ? wi >> wshift : wi / W;
                dst = row_index[row] * W + (POW2 ? (wi & (W - 1)) : wi % W);
            }
            bits[dst] = wp;
            if (zero) zero[dst] = wz;
        }
    }
    if (flags) {
        const int any = (__ballot(bad & 1) ? 1 : 0) | (__ballot(bad & 2) ? 2 : 0);
        if (any && lane == 0) atomicOr(flags, any);
    }
}

// unpack, four columns per lane as one 16-byte non-temporal store (K % 4 == 0: the four share a row and a word)
__global__ __launch_bounds__(kBlock) void k_unpack_flat(const uint32_t* __restrict__ bits, const uint32_t* __restrict__ zero,
                                                        int64_t n, int K, int W, pack_f4* __restrict__ out4) {
    const int64_t total4 = n * (int64_t)K / 4;
    const int k4 = K / 4;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total4; i += (int64_t)gridDim.x * kBlock) {
        const int64_t row = i / k4;
        const int col = (int)(i % k4) * 4;
        const uint32_t b = bits[row * W + col / 32] >> (col & 31);
        const uint32_t z = zero ? zero[row * W + col / 32] >> (col & 31) : 0u;
        pack_f4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (z >> j & 1u) ? 0.0f : ((b >> j & 1u) ? 1.0f : -1.0f);
        __builtin_nontemporal_store(v, out4 + i);
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
    XMH_RANGE("xmh_pack_sign");
    if (n < 0 || K <= 0) return xmh::fail(XMH_EINVAL, "xmh_pack_sign: bad shape n=%lld K=%d", (long long)n, K);
    if (n == 0) return XMH_OK;
    if (!codes || !bits) return xmh::fail(XMH_EINVAL, "xmh_pack_sign: null pointer");
    const int W = (K + 31) / 32;
    if ((K == 16 || K == 8) && (reinterpret_cast<uintptr_t>(codes) & 15) == 0) {
        const pack_f4* c4 = reinterpret_cast<const pack_f4*>(codes);
        if (K == 16) hipLaunchKernelGGL(k_pack_sign_short<4>, dim3(grid_for(n * 4)), dim3(kBlock), 0, xmh::as_stream(stream), c4, n, row_index, bits, zero, flags);
        else hipLaunchKernelGGL(k_pack_sign_short<2>, dim3(grid_for(n * 2)), dim3(kBlock), 0, xmh::as_stream(stream), c4, n, row_index, bits, zero, flags);
    } else if (K % 32 == 0 && (reinterpret_cast<uintptr_t>(codes) & 15) == 0)
    {
        int wshift = 0;
        while ((1 << wshift) < W) ++wshift;
        if ((1 << wshift) == W)
            hipLaunchKernelGGL(k_pack_sign_flat<true>, dim3(grid_for(n * W * 8)), dim3(kBlock), 0, xmh::as_stream(stream),
                               reinterpret_cast<const pack_f4*>(codes), n * W, W, wshift, row_index, bits, zero, flags);
        else
            hipLaunchKernelGGL(k_pack_sign_flat<false>, dim3(grid_for(n * W * 8)), dim3(kBlock), 0, xmh::as_stream(stream),
                               reinterpret_cast<const pack_f4*>(codes), n * W, W, 0, row_index, bits, zero, flags);
    }
    else
        hipLaunchKernelGGL(k_pack_sign, dim3(grid_for(n * W * 32)), dim3(kBlock), 0, xmh::as_stream(stream), codes, n, K, W,
                           row_index, bits, zero, flags);
    XMH_LAUNCH_CHECK("xmh_pack_sign");
    return XMH_OK;
}

extern "C" int xmh_pack_pair_argmax(const float* probs, int64_t n, int K, const int64_t* row_index, uint32_t* bits,
                                    xmh_stream_t stream) {
    XMH_RANGE("xmh_pack_pair_argmax");
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
    XMH_RANGE("xmh_unpack_pm1");
    if (n < 0 || K <= 0) return xmh::fail(XMH_EINVAL, "xmh_unpack_pm1: bad shape n=%lld K=%d", (long long)n, K);
    if (n == 0) return XMH_OK;
    if (!bits || !out) return xmh::fail(XMH_EINVAL, "xmh_unpack_pm1: null pointer");
    const int W = (K + 31) / 32;
    if (K % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0)
        hipLaunchKernelGGL(k_unpack_flat, dim3(grid_for(n * K / 4)), dim3(kBlock), 0, xmh::as_stream(stream), bits, zero, n, K, W,
                           reinterpret_cast<pack_f4*>(out));
    else
        hipLaunchKernelGGL(k_unpack, dim3(grid_for(n * K)), dim3(kBlock), 0, xmh::as_stream(stream), bits, zero, n, K, W, out);
    XMH_LAUNCH_CHECK("xmh_unpack_pm1");
    return XMH_OK;
}

extern "C" int xmh_pack_labels(const void* labels, int dt, int64_t n, int C, uint32_t* lab, xmh_stream_t stream) {
    XMH_RANGE("xmh_pack_labels");
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
