// Materialised pairwise outputs: calc_hammingDist (reference common/calc_utils.py:51-56) and
// calc_label_sim (:8-10) on bit-packed inputs.
//
// One lane per gallery item (its words stay in VGPRs), queries walked in tiles through scalar loads;
// every store instruction writes 64 consecutive floats of one output row.  Write-bound by construction:
// algorithmic bytes = Q*R*sizeof(out) (+ inputs, negligible); the fused scan (xmh_scan.hip) exists so
// that calc_map_k never has to materialise this matrix.
#include "xmh_common.h"

namespace {

constexpr int kQTile = 32;

template <int W, bool TERN, typename OutT>
__global__ __launch_bounds__(256) void k_dist(const uint32_t* __restrict__ qbits, const uint32_t* __restrict__ qzero,
                                              const uint32_t* __restrict__ rbits, const uint32_t* __restrict__ rzero,
                                              int64_t Q, int64_t R, int K, int Wrt, OutT* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t q0 = (int64_t)blockIdx.y * kQTile;
    const int64_t q1 = q0 + kQTile < Q ? q0 + kQTile : Q;
    const bool ok = r < R;
    const int Wn = W > 0 ? W : Wrt;
    uint32_t rb[W > 0 ? W : 1], rz[W > 0 ? W : 1];
    if (W > 0) {
#pragma unroll
        for (int w = 0; w < W; ++w) {
            rb[w] = ok ? rbits[r * W + w] : 0u;
            rz[w] = (TERN && ok) ? rzero[r * W + w] : 0u;
        }
    }
    for (int64_t q = q0; q < q1; ++q) {
        const uint32_t* __restrict__ qb = qbits + q * Wn;        // uniform -> scalar loads
        const uint32_t* __restrict__ qz = TERN ? qzero + q * Wn : nullptr;
        int diff = 0, live_n = 0;
        if (W > 0) {
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
        } else if (ok) {
            for (int w = 0; w < Wn; ++w) {
                const uint32_t x = rbits[r * Wn + w];
                if (TERN) {
                    const uint32_t live = ~(qz[w] | rzero[r * Wn + w]);
                    live_n += __popc(live);
                    diff += __popc((qb[w] ^ x) & live);
                } else {
                    diff += __popc(qb[w] ^ x);
                }
            }
        }
        if (ok) {
            if (TERN) {
                const int d2 = K - live_n + 2 * diff;              // = K - q.r
                out[q * R + r] = (OutT)(0.5f * (float)d2);
            } else {
                out[q * R + r] = (OutT)diff;
            }
        }
    }
}

template <int LW>
__global__ __launch_bounds__(256) void k_label_sim(const uint32_t* __restrict__ qlab, const uint32_t* __restrict__ rlab,
                                                   int64_t Q, int64_t R, int Lrt, float* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t q0 = (int64_t)blockIdx.y * kQTile;
    const int64_t q1 = q0 + kQTile < Q ? q0 + kQTile : Q;
    if (r >= R) return;
    const int Ln = LW > 0 ? LW : Lrt;
    for (int64_t q = q0; q < q1; ++q) {
        uint32_t hit = 0;
        if (LW > 0) {
#pragma unroll
            for (int w = 0; w < LW; ++w) hit |= qlab[q * LW + w] & rlab[r * LW + w];
        } else {
            for (int w = 0; w < Ln; ++w) hit |= qlab[q * Ln + w] & rlab[r * Ln + w];
        }
        out[q * R + r] = hit ? 1.0f : 0.0f;
    }
}

template <bool TERN, typename OutT>
void launch_dist(int W, dim3 grid, hipStream_t st, const uint32_t* qb, const uint32_t* qz, const uint32_t* rb,
                 const uint32_t* rz, int64_t Q, int64_t R, int K, OutT* out) {
    switch (W) {
        case 1: hipLaunchKernelGGL((k_dist<1, TERN, OutT>), grid, dim3(256), 0, st, qb, qz, rb, rz, Q, R, K, W, out); break;
        case 2: hipLaunchKernelGGL((k_dist<2, TERN, OutT>), grid, dim3(256), 0, st, qb, qz, rb, rz, Q, R, K, W, out); break;
        case 4: hipLaunchKernelGGL((k_dist<4, TERN, OutT>), grid, dim3(256), 0, st, qb, qz, rb, rz, Q, R, K, W, out); break;
        case 8: hipLaunchKernelGGL((k_dist<8, TERN, OutT>), grid, dim3(256), 0, st, qb, qz, rb, rz, Q, R, K, W, out); break;
        default: hipLaunchKernelGGL((k_dist<0, TERN, OutT>), grid, dim3(256), 0, st, qb, qz, rb, rz, Q, R, K, W, out); break;
    }
}

}  // namespace

extern "C" int xmh_hamming_dist(const uint32_t* qbits, const uint32_t* qzero, const uint32_t* rbits, const uint32_t* rzero,
                                int64_t Q, int64_t R, int K, float* out_f32, uint16_t* out_u16, xmh_stream_t stream) {
    if (Q < 0 || R < 0 || K <= 0) return xmh::fail(XMH_EINVAL, "xmh_hamming_dist: bad shape");
    if (Q == 0 || R == 0) return XMH_OK;
    if (!qbits || !rbits) return xmh::fail(XMH_EINVAL, "xmh_hamming_dist: null pointer");
    if ((out_f32 == nullptr) == (out_u16 == nullptr)) return xmh::fail(XMH_EINVAL, "xmh_hamming_dist: exactly one of out_f32/out_u16");
    if ((qzero == nullptr) != (rzero == nullptr)) return xmh::fail(XMH_EINVAL, "xmh_hamming_dist: zero masks for both sides or neither");
    const bool tern = qzero != nullptr;
    if (tern && out_u16) return xmh::fail(XMH_EINVAL, "xmh_hamming_dist: ternary codes have half-integer distances; use out_f32");
    const int W = (K + 31) / 32;
    const dim3 grid((unsigned)xmh::ceil_div(R, 256), (unsigned)xmh::ceil_div(Q, kQTile));
    hipStream_t st = xmh::as_stream(stream);
    if (out_u16) launch_dist<false, uint16_t>(W, grid, st, qbits, qzero, rbits, rzero, Q, R, K, out_u16);
    else if (tern) launch_dist<true, float>(W, grid, st, qbits, qzero, rbits, rzero, Q, R, K, out_f32);
    else launch_dist<false, float>(W, grid, st, qbits, qzero, rbits, rzero, Q, R, K, out_f32);
    XMH_LAUNCH_CHECK("xmh_hamming_dist");
    return XMH_OK;
}

extern "C" int xmh_label_sim(const uint32_t* qlab, const uint32_t* rlab, int64_t Q, int64_t R, int C, float* out,
                             xmh_stream_t stream) {
    if (Q < 0 || R < 0 || C <= 0) return xmh::fail(XMH_EINVAL, "xmh_label_sim: bad shape");
    if (Q == 0 || R == 0) return XMH_OK;
    if (!qlab || !rlab || !out) return xmh::fail(XMH_EINVAL, "xmh_label_sim: null pointer");
    const int Lw = (C + 31) / 32;
    const dim3 grid((unsigned)xmh::ceil_div(R, 256), (unsigned)xmh::ceil_div(Q, kQTile));
    hipStream_t st = xmh::as_stream(stream);
    switch (Lw) {
        case 1: hipLaunchKernelGGL(k_label_sim<1>, grid, dim3(256), 0, st, qlab, rlab, Q, R, Lw, out); break;
        case 2: hipLaunchKernelGGL(k_label_sim<2>, grid, dim3(256), 0, st, qlab, rlab, Q, R, Lw, out); break;
        case 3: hipLaunchKernelGGL(k_label_sim<3>, grid, dim3(256), 0, st, qlab, rlab, Q, R, Lw, out); break;
        default: hipLaunchKernelGGL(k_label_sim<0>, grid, dim3(256), 0, st, qlab, rlab, Q, R, Lw, out); break;
    }
    XMH_LAUNCH_CHECK("xmh_label_sim");
    return XMH_OK;
}
