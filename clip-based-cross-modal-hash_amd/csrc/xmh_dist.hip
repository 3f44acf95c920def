// Materialised pairwise outputs: calc_hammingDist (reference common/calc_utils.py:51-56) and
// calc_label_sim (:8-10) on bit-packed inputs.
//
// A lane owns IPL consecutive gallery items (their words stay in VGPRs), queries are walked in tiles through scalar loads, and every
// store instruction of a wave writes 1 KB of one output row: 4 floats / 8 int16 per lane as ONE 16-byte non-temporal store.  Write-bound
// by construction: algorithmic bytes = Q*R*sizeof(out) (+ inputs, negligible); the fused scan (xmh_scan.hip) exists so that calc_map_k
// never has to materialise this matrix.  Round 5: the one-item-per-lane form of rounds 1-4 (a dword store per lane, 256 bytes per wave
// instruction) wrote 3.2 TB/s where torch's fill_ of the same matrix writes 6.9 (tools/bench_dist.py); a row whose start is not 16-byte
// aligned (R % 4 != 0) takes the same stores at dword alignment, which the hardware splits: 4.8 TB/s instead of 6.7.
#include "xmh_common.h"

namespace {

#ifndef XMH_DIST_QTILE
#define XMH_DIST_QTILE 32
#endif
constexpr int kQTile = XMH_DIST_QTILE;
typedef float dist_f4 __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned short dist_us8 __attribute__((ext_vector_type(8), aligned(4)));

template <typename OutT, int IPL>
__device__ __forceinline__ void store_items(OutT* __restrict__ row, int64_t r0, int64_t R, const OutT (&v)[IPL]) {
    if constexpr (IPL == 1) {
        if (r0 < R) row[r0] = v[0];
    } else {
        bool wide = r0 + IPL <= R;
        if constexpr (sizeof(OutT) == 2) wide = wide && ((reinterpret_cast<uintptr_t>(row + r0) & 3) == 0);       // dword alignment at least
        if (wide) {
            if constexpr (sizeof(OutT) == 4) {
                static_assert(IPL == 4, "four floats per store");
                const dist_f4 x = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
#ifdef XMH_DIST_PLAIN
                *reinterpret_cast<dist_f4*>(row + r0) = x;
#else
                __builtin_nontemporal_store(x, reinterpret_cast<dist_f4*>(row + r0));
#endif
            } else {
                static_assert(IPL == 8, "eight int16 per store");
                const dist_us8 x = {(unsigned short)v[0], (unsigned short)v[1], (unsigned short)v[2], (unsigned short)v[3],
                                    (unsigned short)v[4], (unsigned short)v[5], (unsigned short)v[6], (unsigned short)v[7]};
                __builtin_nontemporal_store(x, reinterpret_cast<dist_us8*>(row + r0));
            }
        } else {
#pragma unroll
            for (int i = 0; i < IPL; ++i)
                if (r0 + i < R) row[r0 + i] = v[i];
        }
    }
}

template <int W, bool TERN, typename OutT, int IPL>
__global__ __launch_bounds__(256) void k_dist(const uint32_t* __restrict__ qbits, const uint32_t* __restrict__ qzero,
                                              const uint32_t* __restrict__ rbits, const uint32_t* __restrict__ rzero,
                                              int64_t Q, int64_t R, int K, int Wrt, OutT* __restrict__ out) {
    static_assert(W > 0 || IPL == 1, "code lengths without an instance: one item per lane, words re-read per query");
    const int64_t r0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * IPL;
    const int64_t q0 = (int64_t)blockIdx.y * kQTile;
    const int64_t q1 = q0 + kQTile < Q ? q0 + kQTile : Q;
    if (r0 >= R) return;
    const int Wn = W > 0 ? W : Wrt;
    uint32_t rb[IPL][W > 0 ? W : 1], rz[IPL][W > 0 ? W : 1];
    if (W > 0) {
#pragma unroll
        for (int i = 0; i < IPL; ++i) {
            const bool ok = r0 + i < R;
#pragma unroll
            for (int w = 0; w < W; ++w) {
                rb[i][w] = ok ? rbits[(r0 + i) * W + w] : 0u;
                rz[i][w] = (TERN && ok) ? rzero[(r0 + i) * W + w] : 0u;
            }
        }
    }
    for (int64_t q = q0; q < q1; ++q) {
        const uint32_t* __restrict__ qb = qbits + q * Wn;        // uniform -> scalar loads
        const uint32_t* __restrict__ qz = TERN ? qzero + q * Wn : nullptr;
        OutT v[IPL];
#pragma unroll
        for (int i = 0; i < IPL; ++i) {
            int diff = 0, live_n = 0;
            if (W > 0) {
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    if (TERN) {
                        const uint32_t live = ~(qz[w] | rz[i][w]);
                        live_n += __popc(live);
                        diff += __popc((qb[w] ^ rb[i][w]) & live);
                    } else {
                        diff += __popc(qb[w] ^ rb[i][w]);
                    }
                }
            } else {
                for (int w = 0; w < Wn; ++w) {
                    const uint32_t x = rbits[r0 * Wn + w];
                    if (TERN) {
                        const uint32_t live = ~(qz[w] | rzero[r0 * Wn + w]);
                        live_n += __popc(live);
                        diff += __popc((qb[w] ^ x) & live);
                    } else {
                        diff += __popc(qb[w] ^ x);
                    }
                }
            }
            v[i] = TERN ? (OutT)(0.5f * (float)(K - live_n + 2 * diff)) : (OutT)diff;      // ternary: = (K - q.r) / 2
        }
        store_items<OutT, IPL>(out + q * R, r0, R, v);
    }
}

// float32 outputs, round 5.  What a row of the output costs depends on how its start sits in memory: with rows of whole 128-byte lines
// (R % 32 == 0) the 16-byte stores wrote 5.3 TB/s, with R % 16 == 0 4.9, R % 8 == 0 3.9, anything else 3.7 (tools/proto_dist.hip: every
// wave's 1 KB run then begins and ends inside a line that its neighbour also writes -- partial lines).  So the lanes' columns SLIDE
// with the row: for row q the block covers the columns [1024 b + s - 32, 1024 (b + 1) + s - 32) where s = the columns up to the row's
// first line boundary (s = 0: no shift), which makes every wave's run a whole number of lines of THAT row.  The block's items -- 1024 + 32
// of them -- are staged in LDS once, and a lane reads the four it needs for the row at hand from there (s is uniform per row).  Groups
// that hang over either end of the row are written element by element.
template <int W>
__device__ __forceinline__ void lds_item(const uint32_t* __restrict__ lds, int idx, uint32_t (&dst)[W]) {
    const uint32_t* p = lds + idx * W;
    if constexpr (W % 4 == 0) {
#pragma unroll
        for (int x = 0; x < W / 4; ++x) {
            const uint4 v = reinterpret_cast<const uint4*>(p)[x];
            dst[4 * x] = v.x; dst[4 * x + 1] = v.y; dst[4 * x + 2] = v.z; dst[4 * x + 3] = v.w;
        }
    } else if constexpr (W == 2) {
        const uint2 v = *reinterpret_cast<const uint2*>(p);
        dst[0] = v.x; dst[1] = v.y;
    } else {
#pragma unroll
        for (int x = 0; x < W; ++x) dst[x] = p[x];
    }
}

__device__ __forceinline__ void store_row4(float* __restrict__ row, int64_t col, int64_t R, const float (&v)[4]) {
    if (col >= 0 && col + 4 <= R) {
        const dist_f4 x = {v[0], v[1], v[2], v[3]};
#ifdef XMH_DIST_PLAIN
        *reinterpret_cast<dist_f4*>(row + col) = x;
#else
        __builtin_nontemporal_store(x, reinterpret_cast<dist_f4*>(row + col));
#endif
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (col + i >= 0 && col + i < R) row[col + i] = v[i];
    }
}

// columns up to the first 128-byte boundary of a row that starts at `row` (0..31 floats)
__device__ __forceinline__ int line_shift(const float* row) { return (int)((32 - ((reinterpret_cast<uintptr_t>(row) >> 2) & 31)) & 31); }

constexpr int kStageItems = 1024 + 32;

template <int W, bool TERN>
__global__ __launch_bounds__(256) void k_dist_f32(const uint32_t* __restrict__ qbits, const uint32_t* __restrict__ qzero,
                                                  const uint32_t* __restrict__ rbits, const uint32_t* __restrict__ rzero,
                                                  int64_t Q, int64_t R, int K, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sm_items[];      // [kStageItems][W] bits (+ the same of zero masks)
    uint32_t* sb = sm_items;
    uint32_t* sz = sm_items + kStageItems * W;
    const int64_t first = (int64_t)blockIdx.x * 1024 - 32;       // item held at LDS index 0
    const int64_t q0 = (int64_t)blockIdx.y * kQTile;
    const int64_t q1 = q0 + kQTile < Q ? q0 + kQTile : Q;
    for (int e = threadIdx.x; e < kStageItems * W; e += 256) {
        const int64_t it = first + e / W;
        const bool ok = it >= 0 && it < R;
        sb[e] = ok ? rbits[it * W + e % W] : 0u;
        if (TERN) sz[e] = ok ? rzero[it * W + e % W] : 0u;
    }
    __syncthreads();
    for (int64_t q = q0; q < q1; ++q) {
        const uint32_t* __restrict__ qb = qbits + q * W;         // uniform -> scalar loads
        const uint32_t* __restrict__ qz = TERN ? qzero + q * W : nullptr;
        float* __restrict__ row = out + q * R;
        const int s = line_shift(row);
        const int off = s ? s : 32;                              // LDS index of the lane-0 item: s = 0 keeps the unshifted columns
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t rb[W], rz[W];
            lds_item<W>(sb, 4 * threadIdx.x + off + i, rb);
            if constexpr (TERN) lds_item<W>(sz, 4 * threadIdx.x + off + i, rz);
            int diff = 0, live_n = 0;
#pragma unroll
            for (int w = 0; w < W; ++w) {
                if (TERN) {
                    const uint32_t live = ~(qz[w] | rz[w]);
                    live_n += __popc(live);
                    diff += __popc((qb[w] ^ rb[w]) & live);
                } else {
                    diff += __popc(qb[w] ^ rb[w]);
                }
            }
            v[i] = TERN ? 0.5f * (float)(K - live_n + 2 * diff) : (float)diff;          // ternary: = (K - q.r) / 2
        }
        store_row4(row, first + off + 4 * (int64_t)threadIdx.x, R, v);
    }
}

template <int LW>
__global__ __launch_bounds__(256) void k_label_sim_f32(const uint32_t* __restrict__ qlab, const uint32_t* __restrict__ rlab,
                                                       int64_t Q, int64_t R, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sm_items[];      // [kStageItems][LW]
    const int64_t first = (int64_t)blockIdx.x * 1024 - 32;
    const int64_t q0 = (int64_t)blockIdx.y * kQTile;
    const int64_t q1 = q0 + kQTile < Q ? q0 + kQTile : Q;
    for (int e = threadIdx.x; e < kStageItems * LW; e += 256) {
        const int64_t it = first + e / LW;
        sm_items[e] = (it >= 0 && it < R) ? rlab[it * LW + e % LW] : 0u;
    }
    __syncthreads();
    for (int64_t q = q0; q < q1; ++q) {
        const uint32_t* __restrict__ ql = qlab + q * LW;
        float* __restrict__ row = out + q * R;
        const int s = line_shift(row);
        const int off = s ? s : 32;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t hit = 0;
#pragma unroll
            for (int w = 0; w < LW; ++w) hit |= ql[w] & sm_items[(4 * threadIdx.x + off + i) * LW + w];
            v[i] = hit ? 1.0f : 0.0f;
        }
        store_row4(row, first + off + 4 * (int64_t)threadIdx.x, R, v);
    }
}

template <int LW, int IPL>
__global__ __launch_bounds__(256) void k_label_sim(const uint32_t* __restrict__ qlab, const uint32_t* __restrict__ rlab,
                                                   int64_t Q, int64_t R, int Lrt, float* __restrict__ out) {
    static_assert(LW > 0 || IPL == 1, "label widths without an instance: one item per lane");
    const int64_t r0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * IPL;
    const int64_t q0 = (int64_t)blockIdx.y * kQTile;
    const int64_t q1 = q0 + kQTile < Q ? q0 + kQTile : Q;
    if (r0 >= R) return;
    const int Ln = LW > 0 ? LW : Lrt;
    uint32_t rl[IPL][LW > 0 ? LW : 1];
    if (LW > 0) {
#pragma unroll
        for (int i = 0; i < IPL; ++i)
#pragma unroll
            for (int w = 0; w < LW; ++w) rl[i][w] = r0 + i < R ? rlab[(r0 + i) * LW + w] : 0u;
    }
    for (int64_t q = q0; q < q1; ++q) {
        float v[IPL];
#pragma unroll
        for (int i = 0; i < IPL; ++i) {
            uint32_t hit = 0;
            if (LW > 0) {
#pragma unroll
                for (int w = 0; w < LW; ++w) hit |= qlab[q * LW + w] & rl[i][w];
            } else {
                for (int w = 0; w < Ln; ++w) hit |= qlab[q * Ln + w] & rlab[r0 * Ln + w];
            }
            v[i] = hit ? 1.0f : 0.0f;
        }
        store_items<float, IPL>(out + q * R, r0, R, v);
    }
}

template <bool TERN, typename OutT>
void launch_dist(int W, hipStream_t st, const uint32_t* qb, const uint32_t* qz, const uint32_t* rb,
                 const uint32_t* rz, int64_t Q, int64_t R, int K, OutT* out) {
    constexpr int IPL = sizeof(OutT) == 4 ? 4 : 8;
    const unsigned gy = (unsigned)xmh::ceil_div(Q, kQTile);
    const dim3 wide((unsigned)xmh::ceil_div(R, (int64_t)256 * IPL), gy), one((unsigned)xmh::ceil_div(R, 256), gy);
    if constexpr (sizeof(OutT) == 4) {
        float* o = reinterpret_cast<float*>(out);
        const dim3 slid((unsigned)xmh::ceil_div(R + 32, (int64_t)1024), gy);       // the shifted columns of the last block end up to 31 lower
        const size_t lds = (size_t)kStageItems * W * 4 * (TERN ? 2 : 1);          // <= 17 KB
        switch (W) {
            case 1: hipLaunchKernelGGL((k_dist_f32<1, TERN>), slid, dim3(256), lds, st, qb, qz, rb, rz, Q, R, K, o); return;
            case 2: hipLaunchKernelGGL((k_dist_f32<2, TERN>), slid, dim3(256), lds, st, qb, qz, rb, rz, Q, R, K, o); return;
            default: break;       // 128 bits and more: a lane's four items are 64+ bytes apart in LDS (bank conflicts; 256 bit ran 420 us against
                                  // 261 with the registers-only form below, whose rows take partial-line stores instead)
        }
    }
    switch (W) {
        case 1: hipLaunchKernelGGL((k_dist<1, TERN, OutT, IPL>), wide, dim3(256), 0, st, qb, qz, rb, rz, Q, R, K, W, out); break;
        case 2: hipLaunchKernelGGL((k_dist<2, TERN, OutT, IPL>), wide, dim3(256), 0, st, qb, qz, rb, rz, Q, R, K, W, out); break;
        case 4: hipLaunchKernelGGL((k_dist<4, TERN, OutT, IPL>), wide, dim3(256), 0, st, qb, qz, rb, rz, Q, R, K, W, out); break;
        case 8: hipLaunchKernelGGL((k_dist<8, TERN, OutT, IPL>), wide, dim3(256), 0, st, qb, qz, rb, rz, Q, R, K, W, out); break;
        default: hipLaunchKernelGGL((k_dist<0, TERN, OutT, 1>), one, dim3(256), 0, st, qb, qz, rb, rz, Q, R, K, W, out); break;
    }
}

}  // namespace

extern "C" int xmh_hamming_dist(const uint32_t* qbits, const uint32_t* qzero, const uint32_t* rbits, const uint32_t* rzero,
                                int64_t Q, int64_t R, int K, float* out_f32, uint16_t* out_u16, xmh_stream_t stream) {
    XMH_RANGE("xmh_hamming_dist");
    if (Q < 0 || R < 0 || K <= 0) return xmh::fail(XMH_EINVAL, "xmh_hamming_dist: bad shape");
    if (Q == 0 || R == 0) return XMH_OK;
    if (!qbits || !rbits) return xmh::fail(XMH_EINVAL, "xmh_hamming_dist: null pointer");
    if ((out_f32 == nullptr) == (out_u16 == nullptr)) return xmh::fail(XMH_EINVAL, "xmh_hamming_dist: exactly one of out_f32/out_u16");
    if ((qzero == nullptr) != (rzero == nullptr)) return xmh::fail(XMH_EINVAL, "xmh_hamming_dist: zero masks for both sides or neither");
    const bool tern = qzero != nullptr;
    if (tern && out_u16) return xmh::fail(XMH_EINVAL, "xmh_hamming_dist: ternary codes have half-integer distances; use out_f32");
    const int W = (K + 31) / 32;
    hipStream_t st = xmh::as_stream(stream);
    if (out_u16) launch_dist<false, uint16_t>(W, st, qbits, qzero, rbits, rzero, Q, R, K, out_u16);
    else if (tern) launch_dist<true, float>(W, st, qbits, qzero, rbits, rzero, Q, R, K, out_f32);
    else launch_dist<false, float>(W, st, qbits, qzero, rbits, rzero, Q, R, K, out_f32);
    XMH_LAUNCH_CHECK("xmh_hamming_dist");
    return XMH_OK;
}

extern "C" int xmh_label_sim(const uint32_t* qlab, const uint32_t* rlab, int64_t Q, int64_t R, int C, float* out,
                             xmh_stream_t stream) {
    XMH_RANGE("xmh_label_sim");
    if (Q < 0 || R < 0 || C <= 0) return xmh::fail(XMH_EINVAL, "xmh_label_sim: bad shape");
    if (Q == 0 || R == 0) return XMH_OK;
    if (!qlab || !rlab || !out) return xmh::fail(XMH_EINVAL, "xmh_label_sim: null pointer");
    const int Lw = (C + 31) / 32;
    const unsigned gy = (unsigned)xmh::ceil_div(Q, kQTile);
    const dim3 slid((unsigned)xmh::ceil_div(R + 32, (int64_t)1024), gy), one((unsigned)xmh::ceil_div(R, 256), gy);
    hipStream_t st = xmh::as_stream(stream);
    switch (Lw) {
        case 1: hipLaunchKernelGGL((k_label_sim_f32<1>), slid, dim3(256), (size_t)kStageItems * 4, st, qlab, rlab, Q, R, out); break;
        case 2: hipLaunchKernelGGL((k_label_sim_f32<2>), slid, dim3(256), (size_t)kStageItems * 8, st, qlab, rlab, Q, R, out); break;
        case 3: hipLaunchKernelGGL((k_label_sim_f32<3>), slid, dim3(256), (size_t)kStageItems * 12, st, qlab, rlab, Q, R, out); break;
        default: hipLaunchKernelGGL((k_label_sim<0, 1>), one, dim3(256), 0, st, qlab, rlab, Q, R, Lw, out); break;
    }
    XMH_LAUNCH_CHECK("xmh_label_sim");
    return XMH_OK;
}
