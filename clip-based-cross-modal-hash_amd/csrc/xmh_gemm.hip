// Dense contraction for the encoder: C[M,N] = act(A[M,K] . W[N,K]^T + bias[N]) (+ residual[M,N]).
// W is in nn.Linear layout (row n = output feature n), i.e. an "NT" GEMM -- what every Linear / in_proj /
// out_proj / c_fc / c_proj of the reference's CLIP (models/CLIP/model.py:167-197) and hash heads computes.
//
// Kernels:
//   g16   k_gemm_g16: the fp16-MFMA GEMM of parity and fast mode.  Operands are fp16 "planes" in memory (xmh_planes.h), staged by
//         LDS-DMA into a swizzled, double-buffered LDS tile; parity mode = two activation planes (x = hi + lo) x one or two
//         weight planes, every fp16 x fp16 product exact in fp32, 2^-22 relative error per product, 2-3 MFMAs per product;
//         fast mode = one plane per operand.  Peak 2.5 PFLOP/s (1.25 useful in parity mode); what the chip sustains on random
//         operands with nothing but MFMAs issued is 1.64-1.85 PFLOP/s (power).  See the comment at the kernel.
//   f32   k_gemm_nt_f32, v_mfma_f32_32x32x2_f32: exact fp32 products -- "exact mode" and unaligned shapes; peak 157 TFLOP/s;
//         register-staged, LDS rows padded so that the ds_read_b128 of a 16-lane group lands on 16 distinct 16-byte slots.
//   f16   k_gemm_nt_f16: fp32 operands rounded to fp16 while staged -- fast mode for shapes the planes kernel does not take (K % 32).
// In all of them the k-order inside a slab is permuted identically for A and W (a sum over k does not care), which lets every
// lane fetch its MFMA operands with 16-byte LDS reads.
//
// Bound in practice: the L2 -> LDS path (about 18 TB/s at 128x128 tiles), with the MFMA peak above it;
// bytes per flop (tile size) and waves per tile are the levers.  Algorithmic flops per launch = 2*M*N*K.
#include "xmh_common.h"
#include "xmh_planes.h"
#include <string.h>

#include <stdlib.h>

#include <hip/hip_fp16.h>

namespace {

constexpr int BM = 128, BN = 128;
constexpr int kThreads = 256;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

enum Act { ACT_NONE = 0, ACT_QUICKGELU = 1, ACT_GELU_ERF = 2, ACT_TANH = 3, ACT_RELU = 4 };

// x * sigmoid(1.702 x) = x / (1 + 2^(-1.702 log2(e) x)) on v_exp_f32 and v_rcp_f32 (1 ulp each: relative error 3e-7, the level of the
// split product itself) -- 5 VALU instructions; expf + an IEEE division cost 12 us of a 53 us c_fc GEMM (5000 x 3072 x 768, fast mode)
__device__ __forceinline__ float quick_gelu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930156f * x));
}

// nn.GELU() = 0.5 x (1 + erf(x / sqrt 2)) with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute, i.e. <= 1.5e-7 relative on
// the factor 1 + erf in [1, 2] and <= 1.5e-7 |x| / 2 absolute where it is near 0): branch-free, one v_rcp_f32 and one v_exp_f32 --
// libm's erff costs 14 us of a 55 us GEMM epilogue (5000 x 3072 x 768), this 9; measured |error| <= 4.6e-7 over [-12, 12]
// (tools/check_gelu_erf.py)
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = 1.0f - p * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);      // erf(|x| / sqrt 2)
    return 0.5f * x * (1.0f + copysignf(e, x));
}

__device__ __forceinline__ float apply_act(float x, int act) {
    switch (act) {
        case ACT_QUICKGELU: return quick_gelu(x);                              // x * sigmoid(1.702 x), model.py:162-164
        case ACT_GELU_ERF: return gelu_erf(x);
        case ACT_TANH: return tanhf(x);
        case ACT_RELU: return x > 0.0f ? x : 0.0f;
        default: return x;
    }
}

struct GemmArgs {
    const float* A;
    const float* W;
    const float* bias;
    const float* residual;
    float* C;
    int64_t lda, ldw, ldr, ldc;
    int M, N, K, act;
};

// XCD-aware tile order: consecutive blocks on one XCD (b, b+8, ...) walk down one column panel of W so the
// panel stays in that XCD's L2 (guide T1, bijective form).
__device__ __forceinline__ void tile_of_block(int nbm, int nbn, int& tm, int& tn) {
    const int nwg = nbm * nbn;
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    tn = id / nbm;
    tm = id % nbm;
}

// ---------------------------------------------------------------------------------------------------
// f32 path: BK = 32, LDS row = 32 floats + 4 pad (144 B: a 16-lane ds_read_b128 group covers 16 distinct slots)
// ---------------------------------------------------------------------------------------------------
constexpr int BK32 = 32, LD32 = 36;

// MI = 32-row MFMA tiles per wave along M: 2 -> 128x128 block tile (large grids), 1 -> 64x128 block tile, chosen
// by the launcher when the 128x128 grid would leave less than two waves per SIMD (text tower, N = 768 GEMMs).
template <bool FAST, int MI>
__global__ __launch_bounds__(kThreads) void k_gemm_nt_f32(GemmArgs g) {
    constexpr int TBM = 64 * MI;                               // block rows
    constexpr int AH = MI * 2;                                 // float4 of A staged per thread (TBM rows / 32 rows per pass)
    extern __shared__ __attribute__((aligned(16))) float smem32[];
    float* sA = smem32;                        // [2][TBM * LD32]
    float* sW = smem32 + 2 * TBM * LD32;       // [2][BN * LD32]
    const int nbm = (g.M + TBM - 1) / TBM, nbn = (g.N + BN - 1) / BN;
    int tm, tn;
    tile_of_block(nbm, nbn, tm, tn);
    const int m0 = tm * TBM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 32 * MI, wn = (wave & 1) * 64;

    // staging: rows srow + 32*h, 8 float4 columns per row
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    const bool k_vec = (g.K % 4 == 0) && (g.lda % 4 == 0) && (g.ldw % 4 == 0);
    float4 ra[AH], rw[4];

    auto load_row = [&](const float* base, int64_t ld, int row0, int nrows, int r, int k0) -> float4 {
        if (FAST) {       // K % BK == 0, 16-byte aligned rows: branch-free, rows clamped (clamped rows are never stored)
            const int rr = row0 + r < nrows ? row0 + r : nrows - 1;
            return *reinterpret_cast<const float4*>(base + (int64_t)rr * ld + k0 + scol);
        }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int k = k0 + scol;
        if (row0 + r < nrows) {
            const float* p = base + (int64_t)(row0 + r) * ld + k;
            if (k_vec && k + 3 < g.K) v = *reinterpret_cast<const float4*>(p);
            else {
                if (k < g.K) v.x = p[0];
                if (k + 1 < g.K) v.y = p[1];
                if (k + 2 < g.K) v.z = p[2];
                if (k + 3 < g.K) v.w = p[3];
            }
        }
        return v;
    };
    auto gload = [&](int k0) {
#pragma unroll
        for (int h = 0; h < AH; ++h) ra[h] = load_row(g.A, g.lda, m0, g.M, srow + 32 * h, k0);
#pragma unroll
        for (int h = 0; h < 4; ++h) rw[h] = load_row(g.W, g.ldw, n0, g.N, srow + 32 * h, k0);
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int h = 0; h < AH; ++h) *reinterpret_cast<float4*>(&sA[buf * TBM * LD32 + (srow + 32 * h) * LD32 + scol]) = ra[h];
#pragma unroll
        for (int h = 0; h < 4; ++h) *reinterpret_cast<float4*>(&sW[buf * BN * LD32 + (srow + 32 * h) * LD32 + scol]) = rw[h];
    };

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int nk = (g.K + BK32 - 1) / BK32;
    gload(0);
    swrite(0);
    __syncthreads();
    const int fr = lane & 31, fh = lane >> 5;                  // fragment row/col, k half
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK32);               // global loads fly under the MFMAs below
#pragma unroll
        for (int half = 0; half < 2; ++half) {                 // two k-slabs of 16: lane half fh reads k = half*16 + fh*8 ..+7
            float4 a[MI][2], b[2][2];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const float* pa = &sA[buf * TBM * LD32 + (wm + i * 32 + fr) * LD32 + half * 16 + fh * 8];
                a[i][0] = *reinterpret_cast<const float4*>(pa);
                a[i][1] = *reinterpret_cast<const float4*>(pa + 4);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float* pb = &sW[buf * BN * LD32 + (wn + j * 32 + fr) * LD32 + half * 16 + fh * 8];
                b[j][0] = *reinterpret_cast<const float4*>(pb);
                b[j][1] = *reinterpret_cast<const float4*>(pb + 4);
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const float av = reinterpret_cast<const float*>(&a[i][0])[s];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float bv = reinterpret_cast<const float*>(&b[j][0])[s];
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
        if (kt + 1 < nk) {
            swrite(buf ^ 1);
            __syncthreads();
        }
    }

    // epilogue: lane holds column (lane&31) and rows (e&3) + 8*(e>>2) + 4*(lane>>5) of each 32x32 tile
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn + j * 32 + fr;
            if (col >= g.N) continue;
            const float bv = g.bias ? g.bias[col] : 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                if (row < g.M) {
                    float v = apply_act(acc[i][j][e] + bv, g.act);
                    if (g.residual) v += g.residual[(int64_t)row * g.ldr + col];
                    g.C[(int64_t)row * g.ldc + col] = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// f16 path: BK = 32, operands rounded to fp16 when staged; LDS row = 32 halves + 8 pad (80 B)
// v_mfma_f32_32x32x16_f16: lane l supplies A[i=l&31][k = 8*(l>>5) .. +7] (8 halves), same for B.
// ---------------------------------------------------------------------------------------------------
constexpr int BK16 = 32, LD16 = 40;

template <bool FAST>
__global__ __launch_bounds__(kThreads) void k_gemm_nt_f16(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) _Float16 sA[2][BM * LD16];
    __shared__ __attribute__((aligned(16))) _Float16 sW[2][BN * LD16];
    const int nbm = (g.M + BM - 1) / BM, nbn = (g.N + BN - 1) / BN;
    int tm, tn;
    tile_of_block(nbm, nbn, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

    // staging: 128 rows x 32 floats = 1024 float4 per operand -> 4 per thread
    const int srow = tid >> 3, scol = (tid & 7) * 4;           // rows srow + 32*h
    const bool k_vec = (g.K % 4 == 0) && (g.lda % 4 == 0) && (g.ldw % 4 == 0);
    float4 ra[4], rw[4];
    auto gload = [&](int k0) {
        if (FAST) {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int r = srow + h * 32;
                const int ra_ = m0 + r < g.M ? m0 + r : g.M - 1;
                const int rw_ = n0 + r < g.N ? n0 + r : g.N - 1;
                ra[h] = *reinterpret_cast<const float4*>(g.A + (int64_t)ra_ * g.lda + k0 + scol);
                rw[h] = *reinterpret_cast<const float4*>(g.W + (int64_t)rw_ * g.ldw + k0 + scol);
            }
            return;
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int r = srow + h * 32;
            const int k = k0 + scol;
            float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vw = va;
            if (m0 + r < g.M) {
                const float* p = g.A + (int64_t)(m0 + r) * g.lda + k;
                if (k_vec && k + 3 < g.K) va = *reinterpret_cast<const float4*>(p);
                else {
                    if (k < g.K) va.x = p[0];
                    if (k + 1 < g.K) va.y = p[1];
                    if (k + 2 < g.K) va.z = p[2];
                    if (k + 3 < g.K) va.w = p[3];
                }
            }
            if (n0 + r < g.N) {
                const float* p = g.W + (int64_t)(n0 + r) * g.ldw + k;
                if (k_vec && k + 3 < g.K) vw = *reinterpret_cast<const float4*>(p);
                else {
                    if (k < g.K) vw.x = p[0];
                    if (k + 1 < g.K) vw.y = p[1];
                    if (k + 2 < g.K) vw.z = p[2];
                    if (k + 3 < g.K) vw.w = p[3];
                }
            }
            ra[h] = va;
            rw[h] = vw;
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int r = srow + h * 32;
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            h4 pa, pw;
            pa[0] = (_Float16)ra[h].x; pa[1] = (_Float16)ra[h].y; pa[2] = (_Float16)ra[h].z; pa[3] = (_Float16)ra[h].w;
            pw[0] = (_Float16)rw[h].x; pw[1] = (_Float16)rw[h].y; pw[2] = (_Float16)rw[h].z; pw[3] = (_Float16)rw[h].w;
            *reinterpret_cast<h4*>(&sA[buf][r * LD16 + scol]) = pa;
            *reinterpret_cast<h4*>(&sW[buf][r * LD16 + scol]) = pw;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int nk = (g.K + BK16 - 1) / BK16;
    gload(0);
    swrite(0);
    __syncthreads();
    const int fr = lane & 31, fh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK16);
#pragma unroll
        for (int s = 0; s < 2; ++s) {                          // two k-slabs of 16 per BK
            f16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const f16x8*>(&sA[buf][(wm + i * 32 + fr) * LD16 + s * 16 + fh * 8]);
                b[i] = *reinterpret_cast<const f16x8*>(&sW[buf][(wn + i * 32 + fr) * LD16 + s * 16 + fh * 8]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            swrite(buf ^ 1);
            __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn + j * 32 + fr;
            if (col >= g.N) continue;
            const float bv = g.bias ? g.bias[col] : 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                if (row < g.M) {
                    float v = apply_act(acc[i][j][e] + bv, g.act);
                    if (g.residual) v += g.residual[(int64_t)row * g.ldr + col];
                    g.C[(int64_t)row * g.ldc + col] = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// g16: the fp16-MFMA GEMM on operand planes (xmh_planes.h), staged by LDS-DMA.
//
// Operands are fp16 in memory -- one plane per operand in fast mode; activations as (lo, hi) planes in parity mode, weights
// as one plane when they are fp16-exact (CLIP as released) or (hi, lo) when not -- produced in that format by whatever kernel
// computed them, so the k-loop moves bytes and issues MFMAs and nothing else:
//   * global_load_lds_dwordx4 (1 KB per wave instruction) straight into a double-buffered LDS tile, next tile in flight under the
//     MFMAs of the current one, ONE barrier per k-step (s_waitcnt vmcnt(0); s_barrier; issue next; compute);
//   * LDS tile = [rows][BK halves], 16-byte chunk c of row r stored at chunk c ^ g16_chunk_swz(r): the 16 lanes of every
//     ds_read_b128 group cover all 16 slots of the 256-byte bank row (derivation at g16_chunk_swz).  LDS-DMA writes lane-linear, so
//     the permutation is applied to the per-lane SOURCE address (same 128-byte line, chunks swapped) and again on the read;
//   * v_mfma_f32_16x16x32_f16, slabs of 32 in k order, per slab  acc += a_lo*w_hi; acc += a_hi*w_lo; acc += a_hi*w_hi  (the
//     terms that exist): every tile shape and BK walks k in the same order, so results do not depend on the dispatch.
//     Round 4: 16x16x32 instead of 32x32x16.  The chip is POWER-bound on these kernels (tools/ubench_mfma_power.hip: a loop of
//     nothing but independent MFMAs on uniform random fp16 data sustains 1.64 PFLOP/s with 32x32x16 -- the clock drops to 1.62 GHz
//     -- and 1.85 PFLOP/s with 16x16x32, which touches each accumulator half as often per flop; zeros run 2.45 on both).  Same
//     LDS reads, same registers, twice the MFMA instructions: +9-16 % on the ViT-B/32 shapes (tools/proto_gemm_pp.hip);
//   * tiles are numbered for the XCD's L2: block -> XCD-contiguous id range, inside it groups of 8 tile rows x all tile columns;
//   * epilogue through LDS: accumulators -> the wave's own fp32 region -> 16-byte row pieces; bias / activation / residual there,
//     then fp32 C (dwordx4) and / or the operand planes of the result (8 bytes per plane and lane) for the next GEMM.
// Measured against the register-staged BK = 32 kernels this replaces (M = 5000 ViT-B/32 shapes): fast mode 190-320 -> 400-640
// TFLOP/s, parity mode 184-270 -> 260-430 useful TFLOP/s; tools/proto_gemm_glds.hip holds the ablations (what the C store,
// the staging and the MFMAs each cost) that picked these shapes.
// ---------------------------------------------------------------------------------------------------

struct GArgsP {
    const _Float16 *A0, *A1;      // A0 = lo plane (or the only plane), A1 = hi plane
    const _Float16 *W0, *W1;      // W0 = hi plane, W1 = lo plane or null
    const float* bias;
    const float* residual;
    float* C;
    _Float16 *O_hi, *O_lo;
    int64_t lda, ldw, ldr, ldc, ldo;
    int M, N, K, act;
    const int32_t* m_dev;         // null, or the device word holding the REAL row count (<= M, which then only sizes the grid): the blocks
                                  // whose rows lie behind it return at once -- packed caption rows, counted on the device (no host sync)
};

template <int ACT>
__device__ __forceinline__ float act_ct(float x) {
    if (ACT == ACT_QUICKGELU) return quick_gelu(x);
    if (ACT == ACT_GELU_ERF) return gelu_erf(x);
    if (ACT == ACT_TANH) return tanhf(x);
    if (ACT == ACT_RELU) return x > 0.0f ? x : 0.0f;
    return x;
}

// the wave's (32 MI) x (32 NJ) accumulators -> memory, 32 rows at a time through the wave's own [32][32 NJ + 4] fp32 LDS region:
// written in the 16x16 C layout (lane = column l & 15, rows 4 (l >> 4) + e; the 4 floats of row padding put the four lane groups
// of a ds_write_b32 on two bank halves: 2-way, which a store does not pay for), read back as 16-byte row pieces, so that bias /
// activation / residual and the stores work on 4 consecutive columns: fp32 C as dwordx4, operand planes as 8 bytes per plane.
template <int NJ>
struct G16Epi {
    static constexpr int TW = 32 * NJ, RS = TW + 4, LPR = TW / 4;   // region width in floats, row stride, 16-byte pieces per row
    static constexpr int kWaveBytes = 32 * RS * 4;
};

template <int ACT, int MI, int NJ>
__device__ __forceinline__ void g16_epilogue(const GArgsP& g, f32x4 (&acc)[2 * MI][2 * NJ], char* lds, int wave, int lane, int row0, int col0, int Mr) {
    typedef G16Epi<NJ> E;
    constexpr int RS = E::RS, LPR = E::LPR;
    constexpr bool kFixedCol = 64 % LPR == 0;                      // every pass of the 64 lanes covers whole rows: a lane keeps its columns
    float* reg = reinterpret_cast<float*>(lds + wave * E::kWaveBytes);
    const int c16 = lane & 15, g4 = lane >> 4;
    const bool vec = (g.N % 4 == 0) && (g.ldc % 4 == 0) && (g.ldr % 4 == 0) && (g.ldo % 4 == 0);
    const xmh::Planes op{g.O_hi, g.O_lo, g.ldo};
    // C and residual are the same buffer in the blocks' x += ... GEMMs.  Every lane reads exactly the addresses it then writes and
    // each store takes its value from that load, so letting hipcc treat the two as disjoint only frees it to issue the residual loads
    // of later passes above the stores of earlier ones (otherwise: one exposed memory latency per pass).
    const float* __restrict__ resid = g.residual;
    float* __restrict__ cout = g.C;
    auto bias_at = [&](int col) {
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.bias) {
            if (vec && col < g.N) bv = *reinterpret_cast<const float4*>(g.bias + col);
            else {
                bv.x = col < g.N ? g.bias[col] : 0.0f; bv.y = col + 1 < g.N ? g.bias[col + 1] : 0.0f;
                bv.z = col + 2 < g.N ? g.bias[col + 2] : 0.0f; bv.w = col + 3 < g.N ? g.bias[col + 3] : 0.0f;
            }
        }
        return bv;
    };
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kFixedCol) bv = bias_at(col0 + (lane % LPR) * 4);
    constexpr int NIT = 32 * LPR / 64;                             // passes of the 64 lanes over one 32-row slab
    // The passes read the slab back from LDS, so only the slab WRITE needs the accumulators by constant index.  The towers' two
    // activations keep the passes unrolled (the residual loads of later passes issue above the stores of earlier ones); erf-GELU,
    // tanh and ReLU -- the heads' small GEMMs -- and the ragged-N path loop instead: their inlined bodies times 16 passes times
    // every tile shape were most of the library's code (2.8 MB of GEMM objects, round 5).
    constexpr int kUnroll = (ACT == ACT_NONE || ACT == ACT_QUICKGELU) ? NIT : 1;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j = 0; j < 2 * NJ; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) reg[(i2 * 16 + 4 * g4 + e) * RS + j * 16 + c16] = acc[2 * i + i2][j][e];
        if (vec) {
#pragma unroll kUnroll
            for (int it = 0; it < NIT; ++it) {
                const int f = it * 64 + lane;
                const int r = f / LPR, col = col0 + (f % LPR) * 4;
                const float4 v4 = *reinterpret_cast<const float4*>(reg + r * RS + (f % LPR) * 4);
                const int64_t row = row0 + i * 32 + r;
                if (row >= Mr || col >= g.N) continue;
                if (!kFixedCol) bv = bias_at(col);
                float4 v = make_float4(act_ct<ACT>(v4.x + bv.x), act_ct<ACT>(v4.y + bv.y), act_ct<ACT>(v4.z + bv.z), act_ct<ACT>(v4.w + bv.w));
                if (resid) {
                    const float4 rr = *reinterpret_cast<const float4*>(resid + row * g.ldr + col);
                    v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                }
                if (cout) *reinterpret_cast<float4*>(cout + row * g.ldc + col) = v;
                if (g.O_hi) xmh::store_planes4(op, row, col, v.x, v.y, v.z, v.w);
            }
        } else {
#pragma unroll 1
            for (int it = 0; it < NIT; ++it) {
                const int f = it * 64 + lane;
                const int r = f / LPR, col = col0 + (f % LPR) * 4;
                const float4 v4 = *reinterpret_cast<const float4*>(reg + r * RS + (f % LPR) * 4);
                const int64_t row = row0 + i * 32 + r;
                if (row >= Mr || col >= g.N) continue;
                if (!kFixedCol) bv = bias_at(col);
                const float vv[4] = {act_ct<ACT>(v4.x + bv.x), act_ct<ACT>(v4.y + bv.y), act_ct<ACT>(v4.z + bv.z), act_ct<ACT>(v4.w + bv.w)};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (col + t < g.N) {
                        float x = vv[t];
                        if (g.residual) x += g.residual[row * g.ldr + col + t];
                        if (g.C) g.C[row * g.ldc + col + t] = x;
                        if (g.O_hi) xmh::store_planes1(op, row, col + t, x);
                    }
                }
            }
        }
    }
}

// Swizzle of the 16-byte chunks of an LDS row for v_mfma_f32_16x16x32_f16 fragments (lane l = row l & 15, k chunk l >> 4 of a slab
// of 32).  The lane groups of a ds_read_b128 are {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, {32-35, 44-47, 52-59}, {36-43, 48-51,
// 60-63}: each is the 16 rows of the fragment with rows 4-11 one chunk further than rows 0-3 and 12-15.  With BK = 64 (two rows
// per 256-byte bank row) chunk ^ ((r / 2) & 7) puts the 8 rows of either parity on 8 distinct chunks; with BK = 32 (four rows per
// bank row) the rows r, r + 4, r + 8, r + 12 share their 64 bytes of the bank row and chunk ^ (-(r / 4) & 3) separates them
// (chunk ^ ((r / 4) & 3), right for the 32-row fragments this kernel used before, is 2-way here).
template <int BK>
__device__ __forceinline__ int g16_chunk_swz(int r) {
    constexpr int CH = BK / 8, RPB = 16 / CH;
    return (BK == 32 ? -(r / RPB) : r / RPB) & (CH - 1);
}

template <int WM, int WN, int MI, int NJ, int NA, int NW, int BK, int MINB>
__global__ __launch_bounds__(64 * WM * WN, MINB) void k_gemm_g16(GArgsP g) {
    constexpr int NWAVE = WM * WN;
    constexpr int TBM = 32 * MI * WM, TBN = 32 * NJ * WN;
    constexpr int CH = BK / 8, RPP = 64 / CH;                       // chunks per row, rows per 1 KB piece
    constexpr int ROWB = BK * 2;
    constexpr int PA = TBM / RPP, PW = TBN / RPP;                   // 1 KB pieces per operand plane
    constexpr int NPIECE = NA * PA + NW * PW;
    static_assert(NPIECE % NWAVE == 0, "pieces per wave");
    constexpr int PPW = NPIECE / NWAVE;
    constexpr int BUFB = NPIECE * 1024;
    constexpr int MF = 2 * MI, NF = 2 * NJ;                         // 16 x 16 fragments per wave
    static_assert((32 * 32 * NJ / 4) % 64 == 0, "epilogue: whole passes of 64 lanes");
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    // g.m_dev: the real row count lives on the device and g.M only sized the grid.  The tiles are numbered for the REAL count -- the first
    // nbm * nbn blocks map onto them bijectively, spread evenly over the XCDs (tile_of_block), the surplus blocks return.  (Numbering for g.M
    // and dropping the empty tile rows left the XCDs that own the tail of the id range idle: batch-400 captions 204 k -> 140 k per second.)
    int Mr = g.M;
    if (g.m_dev) {
        const int md = *g.m_dev;
        Mr = md < Mr ? md : Mr;
    }
    const int nbm = (Mr + TBM - 1) / TBM, nbn = (g.N + TBN - 1) / TBN;
    if ((int)blockIdx.x >= nbm * nbn) return;                       // (block-uniform: no barrier is left waiting)
    int tm, tn;
    tile_of_block(nbm, nbn, tm, tn);
    {   // inside the XCD's contiguous id range: groups of kGroupM tile rows x all tile columns, tile rows fastest -- the A rows
        // of a group and the W panels in flight stay in that XCD's L2 (measured +5-15 % on the ViT shapes)
        constexpr int kGroupM = 8;
        const int id = tn * nbm + tm;
        const int per = kGroupM * nbn;
        const int grp = id / per, rem = id % per;
        const int gm0 = grp * kGroupM;
        const int gsz = nbm - gm0 < kGroupM ? nbm - gm0 : kGroupM;
        tm = gm0 + rem % gsz;
        tn = rem / gsz;
    }
    const int m0 = tm * TBM, n0 = tn * TBN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave / WN) * 32 * MI, wn = (wave % WN) * 32 * NJ;
    const int r16 = lane & 15, kc = lane >> 4;

    // staging: piece p = j * NWAVE + wave; pieces [0, NA*PA) the A planes, then the W planes
    const _Float16* src[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int p = j * NWAVE + wave;
        const int prow = lane / CH;
        int r;
        const _Float16* base;
        if (p < NA * PA) {
            r = (p % PA) * RPP + prow;
            const int rg = m0 + r < Mr ? m0 + r : Mr - 1;            // clamped rows are never stored
            base = (p / PA == 0 ? g.A0 : g.A1) + (int64_t)rg * g.lda;
        } else {
            const int q = p - NA * PA;
            r = (q % PW) * RPP + prow;
            const int rg = n0 + r < g.N ? n0 + r : g.N - 1;
            base = (q / PW == 0 ? g.W0 : g.W1) + (int64_t)rg * g.ldw;
        }
        src[j] = base + ((lane % CH) ^ g16_chunk_swz<BK>(r)) * 8;
    }
    auto stage = [&](int buf, int k0) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int p = j * NWAVE + wave;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + k0),
                                             (__attribute__((address_space(3))) void*)(lds + buf * BUFB + p * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0f;

    const int swz = g16_chunk_swz<BK>(r16);     // wm, wn, i*16 are multiples of 16: the swizzle depends on r16 only
    const int nk = g.K / BK;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's pieces of tile kt have landed ...
        __builtin_amdgcn_s_barrier();                              // ... and everyone's; all reads of the other buffer are done
        if (kt + 1 < nk) stage(buf ^ 1, (kt + 1) * BK);
        const char* bA = lds + buf * BUFB;
        const char* bW = bA + NA * PA * 1024;
#pragma unroll
        for (int s = 0; s < BK / 32; ++s) {                        // k slabs of 32: lane group kc takes chunk 4s + kc
            const int coff = ((4 * s + kc) ^ swz) * 16;
            f16x8 b[NF], bl[NW == 2 ? NF : 1], a[MF], ah[NA == 2 ? MF : 1];
#pragma unroll
            for (int j = 0; j < NF; ++j) b[j] = *reinterpret_cast<const f16x8*>(bW + (wn + j * 16 + r16) * ROWB + coff);
#pragma unroll
            for (int i = 0; i < MF; ++i) a[i] = *reinterpret_cast<const f16x8*>(bA + (wm + i * 16 + r16) * ROWB + coff);
            // low parts first: the small terms meet the accumulator before the large ones of this slab
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
            if (NA == 2) {
#pragma unroll
                for (int i = 0; i < MF; ++i) ah[i] = *reinterpret_cast<const f16x8*>(bA + PA * 1024 + (wm + i * 16 + r16) * ROWB + coff);
                if (NW == 2) {
#pragma unroll
                    for (int j = 0; j < NF; ++j) bl[j] = *reinterpret_cast<const f16x8*>(bW + PW * 1024 + (wn + j * 16 + r16) * ROWB + coff);
#pragma unroll
                    for (int i = 0; i < MF; ++i)
#pragma unroll
                        for (int j = 0; j < NF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < MF; ++i)
#pragma unroll
                    for (int j = 0; j < NF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
    }

    // epilogue, 32 accumulator rows at a time: C layout (lane = column, 4 rows) -> LDS -> row pieces of 4 consecutive columns
    __builtin_amdgcn_s_barrier();                                  // every wave is done with the staging buffers
    const int row0 = m0 + wm, col0 = n0 + wn;
    switch (g.act) {                                               // one straight-line epilogue per activation
        case ACT_QUICKGELU: g16_epilogue<ACT_QUICKGELU, MI, NJ>(g, acc, lds, wave, lane, row0, col0, Mr); break;
        case ACT_GELU_ERF: g16_epilogue<ACT_GELU_ERF, MI, NJ>(g, acc, lds, wave, lane, row0, col0, Mr); break;
        case ACT_TANH: g16_epilogue<ACT_TANH, MI, NJ>(g, acc, lds, wave, lane, row0, col0, Mr); break;
        case ACT_RELU: g16_epilogue<ACT_RELU, MI, NJ>(g, acc, lds, wave, lane, row0, col0, Mr); break;
        default: g16_epilogue<ACT_NONE, MI, NJ>(g, acc, lds, wave, lane, row0, col0, Mr); break;
    }
}

// fp32 [rows][cols] -> operand planes, 8 elements per thread
__global__ __launch_bounds__(256) void k_split_planes(const float* __restrict__ x, int64_t ldx, int64_t rows, int cols8, xmh::Planes p) {
    const int64_t total = rows * cols8;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / cols8;
        const int c = (int)(e % cols8) * 8;
        const float4 a = *reinterpret_cast<const float4*>(x + row * ldx + c), b = *reinterpret_cast<const float4*>(x + row * ldx + c + 4);
        if (p.lo) {
            uint4 h, l;
            xmh::split2(a.x, a.y, h.x, l.x); xmh::split2(a.z, a.w, h.y, l.y);
            xmh::split2(b.x, b.y, h.z, l.z); xmh::split2(b.z, b.w, h.w, l.w);
            *reinterpret_cast<uint4*>(p.hi + row * p.ld + c) = h;
            *reinterpret_cast<uint4*>(p.lo + row * p.ld + c) = l;
        } else {
            *reinterpret_cast<uint4*>(p.hi + row * p.ld + c) = make_uint4(xmh::round2(a.x, a.y), xmh::round2(a.z, a.w), xmh::round2(b.x, b.y), xmh::round2(b.z, b.w));
        }
    }
}

// f = QuickGELU(u), four columns per thread: fp32 and / or operand planes out (the saved-activation forward keeps both sides of the
// c_fc activation; same function as the fused epilogue's ACT_QUICKGELU on the same fp32 value)
__global__ __launch_bounds__(256) void k_quickgelu_planes(const float* __restrict__ u, int64_t rows, int cols4, float* __restrict__ f, xmh::Planes p) {
    const int64_t total = rows * cols4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / cols4;
        const int c = (int)(e % cols4) * 4;
        const float4 v = *reinterpret_cast<const float4*>(u + row * (int64_t)cols4 * 4 + c);
        const float o0 = quick_gelu(v.x), o1 = quick_gelu(v.y), o2 = quick_gelu(v.z), o3 = quick_gelu(v.w);
        if (f) *reinterpret_cast<float4*>(f + row * (int64_t)cols4 * 4 + c) = make_float4(o0, o1, o2, o3);
        if (p.hi) xmh::store_planes4(p, row, c, o0, o1, o2, o3);
    }
}

template <int WM, int WN, int MI, int NJ, int NA, int NW, int BK, int MINB>
int launch_g16(const GArgsP& a, hipStream_t st) {
    constexpr int TBM = 32 * MI * WM, TBN = 32 * NJ * WN;
    constexpr size_t stage_b = (size_t)2 * (NA * TBM + NW * TBN) * BK * 2, epi_b = (size_t)WM * WN * G16Epi<NJ>::kWaveBytes;
    constexpr size_t lds = stage_b > epi_b ? stage_b : epi_b;     // the epilogue regions reuse the staging buffers
    auto kern = k_gemm_g16<WM, WN, MI, NJ, NA, NW, BK, MINB>;
    if (const int rl = xmh::raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds, "xmh gemm")) return rl;
    const int64_t nblk = xmh::ceil_div(a.M, TBM) * xmh::ceil_div(a.N, TBN);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(64 * WM * WN), lds, st, a);
    return XMH_OK;
}

}  // namespace

extern "C" int xmh_gemm_nt_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                               const float* residual, int64_t ldr, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                               int act, int precision, xmh_stream_t stream) {
    if (M < 0 || N < 0 || K <= 0) return xmh::fail(XMH_EINVAL, "xmh_gemm_nt_f32: bad shape M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
    if (M == 0 || N == 0) return XMH_OK;
    if (!A || !W || !C) return xmh::fail(XMH_EINVAL, "xmh_gemm_nt_f32: null pointer");
    if (lda < K || ldw < K || ldc < N || (residual && ldr < N)) return xmh::fail(XMH_EINVAL, "xmh_gemm_nt_f32: leading dimension too small");
    if (act < 0 || act > 4) return xmh::fail(XMH_EINVAL, "xmh_gemm_nt_f32: unknown activation %d", act);
    if (M >= (1ll << 31) || N >= (1ll << 31) || K >= (1ll << 31)) return xmh::fail(XMH_ENOTSUP, "xmh_gemm_nt_f32: dimension >= 2^31");
    GemmArgs g;
    g.A = A; g.W = W; g.bias = bias; g.residual = residual; g.C = C;
    g.lda = lda; g.ldw = ldw; g.ldr = ldr; g.ldc = ldc;
    g.M = (int)M; g.N = (int)N; g.K = (int)K; g.act = act;
    int64_t nblk = xmh::ceil_div(M, BM) * xmh::ceil_div(N, BN);
    hipStream_t st = xmh::as_stream(stream);
    xmh::ProfScope prof(precision == 1 ? "gemm_f16" : "gemm_f32", st);
    const bool aligned = (lda % 4 == 0) && (ldw % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W)) % 16 == 0);
    if (precision == 0) {
        const bool fast = aligned && K % BK32 == 0;
        const bool small = nblk < 2ll * xmh::device_cu_count();          // fewer than two 128x128 blocks per CU -> 64x128 tiles
        const size_t lds = (size_t)2 * ((small ? 64 : 128) + BN) * LD32 * 4;
        {
            const size_t big = (size_t)2 * (128 + BN) * LD32 * 4;        // 73,728 B: above the 64 KB default, opt in once per device
            if (const int rl = xmh::raise_dynamic_lds(reinterpret_cast<const void*>(k_gemm_nt_f32<true, 2>), big, "xmh_gemm_nt_f32")) return rl;
            if (const int rl = xmh::raise_dynamic_lds(reinterpret_cast<const void*>(k_gemm_nt_f32<false, 2>), big, "xmh_gemm_nt_f32")) return rl;
        }
        if (small) {
            nblk = xmh::ceil_div(M, 64) * xmh::ceil_div(N, BN);
            if (fast) hipLaunchKernelGGL((k_gemm_nt_f32<true, 1>), dim3((unsigned)nblk), dim3(kThreads), lds, st, g);
            else hipLaunchKernelGGL((k_gemm_nt_f32<false, 1>), dim3((unsigned)nblk), dim3(kThreads), lds, st, g);
        } else {
            if (fast) hipLaunchKernelGGL((k_gemm_nt_f32<true, 2>), dim3((unsigned)nblk), dim3(kThreads), lds, st, g);
            else hipLaunchKernelGGL((k_gemm_nt_f32<false, 2>), dim3((unsigned)nblk), dim3(kThreads), lds, st, g);
        }
    } else if (precision == 1) {
        if (aligned && K % BK16 == 0) hipLaunchKernelGGL(k_gemm_nt_f16<true>, dim3((unsigned)nblk), dim3(kThreads), 0, st, g);
        else hipLaunchKernelGGL(k_gemm_nt_f16<false>, dim3((unsigned)nblk), dim3(kThreads), 0, st, g);
    } else return xmh::fail(XMH_EINVAL, "xmh_gemm_nt_f32: precision must be 0 (f32 MFMA) or 1 (f16 MFMA, f32 accumulate)");
    XMH_LAUNCH_CHECK("xmh_gemm_nt_f32");
    return XMH_OK;
}

namespace xmh {

bool gemm_planes_ok(int64_t K, int64_t lda, int64_t ldw, const void* A, const void* W) {
    return K > 0 && K % 32 == 0 && lda % 8 == 0 && ldw % 8 == 0 && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W)) % 16 == 0);
}

int split_planes(const float* x, int64_t ldx, int64_t rows, int64_t cols, const Planes& p, hipStream_t st) {
    if (rows <= 0 || cols <= 0) return XMH_OK;
    if (cols % 8 || ldx % 4 || p.ld % 8 || reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(p.hi) % 16 || reinterpret_cast<uintptr_t>(p.lo) % 16)
        return fail(XMH_ENOTSUP, "xmh split_planes: needs cols %% 8 == 0 and 16-byte aligned rows (cols=%lld ldx=%lld)", (long long)cols, (long long)ldx);
    int64_t grid = ceil_div(rows * (cols / 8), 256);
    const int64_t cap = (int64_t)device_cu_count() * 16;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(k_split_planes, dim3((unsigned)grid), dim3(256), 0, st, x, ldx, rows, (int)(cols / 8), p);
    XMH_LAUNCH_CHECK("xmh split_planes");
    return XMH_OK;
}

int quickgelu_planes(const float* u, int64_t rows, int64_t cols, float* f, const Planes& p, hipStream_t st) {
    if (rows <= 0 || cols <= 0) return XMH_OK;
    if (cols % 4 || reinterpret_cast<uintptr_t>(u) % 16 || reinterpret_cast<uintptr_t>(f) % 16 || (p.hi && p.ld % 4))
        return fail(XMH_ENOTSUP, "xmh quickgelu_planes: needs cols %% 4 == 0 and 16-byte aligned rows (cols=%lld)", (long long)cols);
    int64_t grid = ceil_div(rows * (cols / 4), 256);
    const int64_t cap = (int64_t)device_cu_count() * 16;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(k_quickgelu_planes, dim3((unsigned)grid), dim3(256), 0, st, u, rows, (int)(cols / 4), f, p);
    XMH_LAUNCH_CHECK("xmh quickgelu_planes");
    return XMH_OK;
}

// Tile shapes (tools/proto_gemm_glds.hip, M = 5000 / 20000 ViT-B/32 shapes, useful TFLOP/s):
//   fast   128x128, BK 64, 2 blocks per CU: 550-770; 256x256 (8 waves) once there are >= 0.7 such tiles per CU: 660-1000;
//          64x128 for grids that leave most CUs empty;
//   parity 128x128, BK 32, 3 blocks per CU: 310-360; 128x256 with 8 waves once there are >= 1.5 such tiles per CU: 370-430;
//   parity with split weights (three terms): 128x128, BK 32, 2 blocks per CU.
int gemm_planes(const GemmPlanes& g, hipStream_t st) {
    if (g.M < 0 || g.N < 0 || g.K <= 0) return fail(XMH_EINVAL, "xmh gemm: bad shape M=%lld N=%lld K=%lld", (long long)g.M, (long long)g.N, (long long)g.K);
    if (g.M == 0 || g.N == 0) return XMH_OK;
    if (!g.A_hi || !g.W_hi || (!g.C && !g.O.hi)) return fail(XMH_EINVAL, "xmh gemm: null pointer");
    if (!gemm_planes_ok(g.K, g.lda, g.ldw, g.A_hi, g.W_hi) || reinterpret_cast<uintptr_t>(g.A_lo) % 16 || reinterpret_cast<uintptr_t>(g.W_lo) % 16)
        return fail(XMH_ENOTSUP, "xmh gemm: needs K %% 32 == 0 and 16-byte aligned rows (K=%lld lda=%lld ldw=%lld)", (long long)g.K, (long long)g.lda, (long long)g.ldw);
    if (g.W_lo && !g.A_lo) return fail(XMH_EINVAL, "xmh gemm: split weights go with split activations");
    if (g.lda < g.K || g.ldw < g.K || (g.C && g.ldc < g.N) || (g.residual && g.ldr < g.N) || (g.O.hi && g.O.ld < g.N))
        return fail(XMH_EINVAL, "xmh gemm: leading dimension too small");
    if (g.act < 0 || g.act > 4) return fail(XMH_EINVAL, "xmh gemm: unknown activation %d", g.act);
    if (g.M >= (1ll << 31) || g.N >= (1ll << 31) || g.K >= (1ll << 31)) return fail(XMH_ENOTSUP, "xmh gemm: dimension >= 2^31");
    const bool vec = g.N % 4 == 0 && (!g.C || g.ldc % 4 == 0) && (!g.residual || g.ldr % 4 == 0) && (!g.O.hi || g.O.ld % 4 == 0);
    if (vec && ((reinterpret_cast<uintptr_t>(g.C) | reinterpret_cast<uintptr_t>(g.residual)) % 16 || (reinterpret_cast<uintptr_t>(g.O.hi) | reinterpret_cast<uintptr_t>(g.O.lo)) % 8))
        return fail(XMH_ENOTSUP, "xmh gemm: C / residual must be 16-byte aligned");
    GArgsP a;
    a.A0 = g.A_lo ? g.A_lo : g.A_hi; a.A1 = g.A_hi;
    a.W0 = g.W_hi; a.W1 = g.W_lo;
    a.bias = g.bias; a.residual = g.residual; a.C = g.C;
    a.O_hi = g.O.hi; a.O_lo = g.O.lo;
    a.lda = g.lda; a.ldw = g.ldw; a.ldr = g.residual ? g.ldr : 0; a.ldc = g.C ? g.ldc : 0; a.ldo = g.O.hi ? g.O.ld : 0;
    a.M = (int)g.M; a.N = (int)g.N; a.K = (int)g.K; a.act = g.act;
    a.m_dev = g.m_dev;
    const int64_t cus = device_cu_count();
    const int64_t n128 = ceil_div(g.M, 128) * ceil_div(g.N, 128);
    int rc;
    // Tile shape by how the grid fills the chip (every shape walks k in the same order, so this never changes a result):
    //   grids that leave more than half the CUs without a 128x128 tile: 64-row tiles (with the prototype's bare epilogue they also won
    //   at one tile per CU, with bias / residual / planes they lose there: 63.6 vs 80.7 us at 5000 x 768 x 3072 parity);
    //   a grid that needs a second, mostly empty round of 128x128 tiles: 192x128 (fast) / 128x192 (parity) when those fit one round
    //   (5000 x 2304 x 768 parity 53.4 -> 51.5 us);
    //   large grids: 256x256 (fast), 128x256 with 8 waves (parity).
    static const bool no_wide = xmh_experiment_env("XMH_GEMM_NO_WIDE") != nullptr;
    static const int tile_rules = xmh_experiment_env("XMH_GEMM_TILE_RULES") ? atoi(xmh_experiment_env("XMH_GEMM_TILE_RULES")) : 30;    // bit 4: 8-wave tiles for the three-term product; bit 3: 64x128 tiles of 8 waves for grids of at most two 128x128 tiles per CU; bit 0: 64-row tiles already for grids of <= one 128x128 tile per CU (measured slower with the real epilogues: off), bit 1: 192-wide tiles, bit 2: 8-wave 128x128 tiles
    const bool under = (tile_rules & 1) ? n128 <= cus : 2 * n128 < cus;
    const bool k64 = g.K % 64 == 0;
    if (!g.A_lo) {
        ProfScope prof("gemm_f16", st);
        const int64_t n192 = ceil_div(g.M, 192) * ceil_div(g.N, 128);
        if ((tile_rules & 8) && k64 && n128 <= cus) rc = launch_g16<2, 4, 1, 1, 1, 1, 64, 2>(a, st);          // small grids: 64 x 128 as 8 waves of 32 x 32 (up to one 128 x 128 tile per CU; at two the 128 x 128 tile is ahead in fast mode)
        else if (under) rc = k64 ? launch_g16<1, 2, 2, 2, 1, 1, 64, 3>(a, st) : launch_g16<1, 2, 2, 2, 1, 1, 32, 4>(a, st);
        else if (!no_wide && k64 && g.K >= 2048 && ceil_div(g.M, 256) * ceil_div(g.N, 256) * 10 >= 7 * cus)
            rc = launch_g16<2, 4, 4, 2, 1, 1, 64, 1>(a, st);       // 256 x 256, 8 waves of 128 x 64: half the L2 -> LDS bytes per flop.  Long k only: one block per CU
                                                                   // leaves the CU idle while it stores C, which at K = 768 costs more than the bytes save
        else if ((tile_rules & 2) && !no_wide && k64 && n128 > 2 * cus && n128 < 3 * cus && n192 <= 2 * cus) rc = launch_g16<2, 2, 3, 2, 1, 1, 64, 2>(a, st);
        else if (k64 && (tile_rules & 4)) rc = launch_g16<4, 2, 1, 2, 1, 1, 64, 2>(a, st);      // 128 x 128 as 8 waves of 32 x 64: twice the waves per tile hide more of the staging (5-12 % over 4 waves of 64 x 64)
        else rc = k64 ? launch_g16<2, 2, 2, 2, 1, 1, 64, 2>(a, st) : launch_g16<2, 2, 2, 2, 1, 1, 32, 4>(a, st);
    } else {
        ProfScope prof("gemm_s16", st);
        const int64_t n192 = ceil_div(g.M, 128) * ceil_div(g.N, 192);
        if (g.W_lo && (tile_rules & 16)) rc = n128 <= 2 * cus ? launch_g16<2, 4, 1, 1, 2, 2, 32, 2>(a, st) : launch_g16<4, 2, 1, 2, 2, 2, 32, 2>(a, st);   // 8 waves per tile here too
        else if (g.W_lo) rc = 2 * n128 < cus ? launch_g16<1, 2, 2, 2, 2, 2, 32, 2>(a, st) : launch_g16<2, 2, 2, 2, 2, 2, 32, 2>(a, st);
        else if (!no_wide && ceil_div(g.M, 128) * ceil_div(g.N, 256) * 2 >= 3 * cus) rc = launch_g16<2, 4, 2, 2, 2, 1, 32, 1>(a, st);
        else if ((tile_rules & 8) && k64 && n128 <= 2 * cus) rc = launch_g16<2, 4, 1, 1, 2, 1, 64, 2>(a, st);  // small grids: 64 x 128 as 8 waves of 32 x 32 (3200 x 512 x 2048: 34.6 -> 24.8 us)
        else if (under) rc = launch_g16<1, 2, 2, 2, 2, 1, 32, 3>(a, st);
        else if ((tile_rules & 2) && !no_wide && n128 > 2 * cus && n192 <= 2 * cus) rc = launch_g16<2, 2, 2, 3, 2, 1, 32, 2>(a, st);
        else if ((tile_rules & 4) && n128 <= cus && k64) rc = launch_g16<4, 2, 1, 2, 2, 1, 64, 1>(a, st);      // one block per CU: 8 waves of 32 x 64
        else if (!(tile_rules & 1) && n128 <= cus && g.K >= 2048 && k64) rc = launch_g16<2, 2, 2, 2, 2, 1, 64, 1>(a, st);
        else rc = launch_g16<2, 2, 2, 2, 2, 1, 32, 3>(a, st);
    }
    if (rc) return rc;
    XMH_LAUNCH_CHECK("xmh gemm (fp16 MFMA on operand planes)");
    return XMH_OK;
}

}  // namespace xmh

namespace {
// stream-ordered scratch for the entry points that take fp32 activations (the fused forwards carry their planes in the
// caller's workspace instead)
struct AsyncScratch {
    void* p = nullptr;
    hipStream_t st;
    explicit AsyncScratch(hipStream_t s) : st(s) {}
    int get(size_t bytes) { return hipMallocAsync(&p, bytes, st) == hipSuccess ? 0 : xmh::fail(XMH_EHIP, "xmh gemm: hipMallocAsync of %zu bytes failed", bytes); }
    ~AsyncScratch() { if (p) (void)hipFreeAsync(p, st); }
};
}  // namespace

extern "C" int xmh_cast_f32_to_f16(const float* x, void* y_half, int64_t n, xmh_stream_t stream) {
    if (n < 0 || n % 8) return xmh::fail(XMH_EINVAL, "xmh_cast_f32_to_f16: n=%lld must be a non-negative multiple of 8", (long long)n);
    if (n == 0) return XMH_OK;
    if (!x || !y_half) return xmh::fail(XMH_EINVAL, "xmh_cast_f32_to_f16: null pointer");
    return xmh::split_planes(x, n, 1, n, xmh::Planes{static_cast<_Float16*>(y_half), nullptr, n}, xmh::as_stream(stream));
}

extern "C" int xmh_gemm_nt_h16(const void* A_half, int64_t lda, const void* W_half, int64_t ldw, const float* bias,
                               const float* residual, int64_t ldr, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                               int act, xmh_stream_t stream) {
    if (!C && M > 0 && N > 0) return xmh::fail(XMH_EINVAL, "xmh_gemm_nt_h16: null pointer");
    xmh::GemmPlanes g{};
    g.A_hi = static_cast<const _Float16*>(A_half); g.lda = lda;
    g.W_hi = static_cast<const _Float16*>(W_half); g.ldw = ldw;
    g.bias = bias; g.residual = residual; g.ldr = ldr; g.C = C; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.act = act;
    return xmh::gemm_planes(g, xmh::as_stream(stream));
}

extern "C" int xmh_gemm_nt_split16(const float* A, int64_t lda, const void* W_half, const void* W_lo_half, int64_t ldw, const float* bias,
                                   const float* residual, int64_t ldr, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                                   int act, xmh_stream_t stream) {
    if (M < 0 || N < 0 || K <= 0) return xmh::fail(XMH_EINVAL, "xmh_gemm_nt_split16: bad shape");
    if (M == 0 || N == 0) return XMH_OK;
    if (!A || !W_half || !C) return xmh::fail(XMH_EINVAL, "xmh_gemm_nt_split16: null pointer");
    if (K % 32 || lda % 4 || ldw % 8 || (reinterpret_cast<uintptr_t>(A) % 16) || (reinterpret_cast<uintptr_t>(W_half) % 16) ||
        (reinterpret_cast<uintptr_t>(W_lo_half) % 16))
        return xmh::fail(XMH_ENOTSUP, "xmh_gemm_nt_split16: needs K %% 32 == 0 and 16-byte aligned rows (K=%lld lda=%lld ldw=%lld)", (long long)K, (long long)lda, (long long)ldw);
    if (lda < K) return xmh::fail(XMH_EINVAL, "xmh_gemm_nt_split16: leading dimension too small");
    hipStream_t st = xmh::as_stream(stream);
    AsyncScratch sc(st);
    const size_t plane = ((size_t)M * K * sizeof(_Float16) + 255) & ~size_t(255);
    if (int rc = sc.get(2 * plane)) return rc;
    xmh::Planes ap{static_cast<_Float16*>(sc.p), reinterpret_cast<_Float16*>(static_cast<char*>(sc.p) + plane), K};
    if (int rc = xmh::split_planes(A, lda, M, K, ap, st)) return rc;
    xmh::GemmPlanes g{};
    g.A_hi = ap.hi; g.A_lo = ap.lo; g.lda = K;
    g.W_hi = static_cast<const _Float16*>(W_half); g.W_lo = static_cast<const _Float16*>(W_lo_half); g.ldw = ldw;
    g.bias = bias; g.residual = residual; g.ldr = ldr; g.C = C; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.act = act;
    return xmh::gemm_planes(g, st);
}
