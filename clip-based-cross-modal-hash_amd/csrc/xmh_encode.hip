// Non-GEMM pieces of the CLIP ViT-B/32 image / text forward and of the hash heads.
// All activations are fp32, token-major [B, L, D] (the reference works in LND; only the layout differs).
//
//   xmh_layernorm_f32     models/CLIP/model.py:153-159 (fp32 LayerNorm, eps 1e-5), one wave per row
//   xmh_attention_f32     nn.MultiheadAttention inside ResidualAttentionBlock (model.py:167-197): per (batch, head)
//                         softmax(Q K^T / sqrt(dh) + mask) V with the whole head resident in LDS (L <= 128)
//   xmh_im2col_patch      the non-overlapping Conv2d(3, width, k=32, s=32) of VisionTransformer (model.py:219,235)
//                         as a gather into GEMM rows (k index = c*P*P + dy*P + dx, the conv weight's own order)
//   xmh_vit_assemble      cls token concat + positional embedding + ln_pre (model.py:237-243)
//   xmh_text_embed        token embedding + positional embedding, EOS = argmax(ids) (model.py:374-379)
//   xmh_gather_rows       cls / EOS row selection (model.py:262-265, :392)
//   xmh_affine_cols       eval-mode BatchNorm1d of the DCMHT image head (models/DCMHT/hash/hash.py:22,40)
//   xmh_pair_softmax      softmax_hash (models/common/hash.py:21-31) on relu'd logits
//   xmh_lta_aggregate     MITH LocalizedTokenAggregation + positional encoding (models/MITH/hash/hash.py:41-65,109-169)
//   xmh_bitwise_hash      MITH BitwiseHashing (models/MITH/hash/hash.py:68-85)
// Every one of these is HBM-bound elementwise / row-reduction work; the GEMMs around them dominate the time.
#include "xmh_common.h"
#include "xmh_planes.h"

#include <stdlib.h>

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// ---- LayerNorm: one wave per row, row cached in registers (D <= 64*16) -------------------------------
constexpr int kLnMaxPerLane = 16;

__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, float eps, float* __restrict__ y, int64_t ldy,
                                                   int64_t rows, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * ldx;
    float v[kLnMaxPerLane];
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < kLnMaxPerLane; ++j) {
        const int c = j * 64 + lane;
        v[j] = c < D ? xr[c] : 0.0f;
        s += v[j];
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.0f;
#pragma unroll
    for (int j = 0; j < kLnMaxPerLane; ++j) {
        const int c = j * 64 + lane;
        const float d = c < D ? v[j] - mean : 0.0f;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    float* yr = y + row * ldy;
#pragma unroll
    for (int j = 0; j < kLnMaxPerLane; ++j) {
        const int c = j * 64 + lane;
        if (c < D) yr[c] = (v[j] - mean) * rstd * gamma[c] + beta[c];
    }
}

// D % 4 == 0, 16-byte aligned rows: four consecutive columns per lane, and the result goes out as fp32 and / or as the fp16
// operand planes of the GEMM that consumes it (xmh_planes.h) -- the same values either way.
__global__ __launch_bounds__(256) void k_layernorm4(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float eps, float* __restrict__ y, int64_t ldy,
                                                    xmh::Planes p, int64_t rows, int D, const int32_t* __restrict__ rows_dev) {
    constexpr int NV = kLnMaxPerLane / 4;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows || (rows_dev && row >= *rows_dev)) return;       // rows_dev: the real row count, known on the device only
    const float* xr = x + row * ldx;
    float4 v[NV];
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = (j * 64 + lane) * 4;
        v[j] = c < D ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.0f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = (j * 64 + lane) * 4;
        if (c < D) {
            const float d0 = v[j].x - mean, d1 = v[j].y - mean, d2 = v[j].z - mean, d3 = v[j].w - mean;
            q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = (j * 64 + lane) * 4;
        if (c < D) {
            const float4 gm = *reinterpret_cast<const float4*>(gamma + c), bt = *reinterpret_cast<const float4*>(beta + c);
            const float o0 = (v[j].x - mean) * rstd * gm.x + bt.x, o1 = (v[j].y - mean) * rstd * gm.y + bt.y;
            const float o2 = (v[j].z - mean) * rstd * gm.z + bt.z, o3 = (v[j].w - mean) * rstd * gm.w + bt.w;
            if (y) *reinterpret_cast<float4*>(y + row * ldy + c) = make_float4(o0, o1, o2, o3);
            if (p.hi) xmh::store_planes4(p, row, c, o0, o1, o2, o3);
        }
    }
}

// ---- small-sequence attention: one block per (batch, head), one thread per query row -------------------
// qkv: [B, L, 3*H*dh] rows = tokens, columns [q | k | v] each H*dh wide (nn.MultiheadAttention in_proj order).
// LDS: K and V of this head [L][dh] (float4 rows), scores [L][L+1].
template <int DH>
__global__ __launch_bounds__(128) void k_attention(const float* __restrict__ qkv, int L, int H, int causal, const uint8_t* __restrict__ kpm,
                            float* __restrict__ out, xmh::Planes pl) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int D = H * DH;
    // K and V rows are read by all lanes at the same address (LDS broadcast: no bank conflicts whatever the stride), so
    // they stay unpadded and 16-byte aligned for ds_read_b128; the score rows are per-lane and padded by one float.
    float4* sK = reinterpret_cast<float4*>(smem);                    // [L][DH/4]
    float4* sV = sK + L * (DH / 4);                                  // [L][DH/4]
    float* sS = reinterpret_cast<float*>(sV + L * (DH / 4));         // [L][L+1]
    const float* base = qkv + (int64_t)b * L * 3 * D;
    for (int e = threadIdx.x; e < L * (DH / 4); e += blockDim.x) {
        const int j = e / (DH / 4), c4 = e % (DH / 4);
        sK[e] = *reinterpret_cast<const float4*>(base + (int64_t)j * 3 * D + D + h * DH + c4 * 4);
        sV[e] = *reinterpret_cast<const float4*>(base + (int64_t)j * 3 * D + 2 * D + h * DH + c4 * 4);
    }
    __syncthreads();
    const int i = threadIdx.x;
    if (i >= L) return;
    float q[DH];
    const float scale = rsqrtf((float)DH);
#pragma unroll
    for (int c = 0; c < DH / 4; ++c) {                               // PyTorch scales q before QK^T
        const float4 v = *reinterpret_cast<const float4*>(base + (int64_t)i * 3 * D + h * DH + c * 4);
        q[4 * c + 0] = v.x * scale;
        q[4 * c + 1] = v.y * scale;
        q[4 * c + 2] = v.z * scale;
        q[4 * c + 3] = v.w * scale;
    }
    float* srow = sS + i * (L + 1);
    float mx = -INFINITY;
    for (int j = 0; j < L; ++j) {
        float s0 = 0.0f, s1 = 0.0f;                                  // two chains: shorter dependent-FMA latency
#pragma unroll
        for (int c = 0; c < DH / 4; ++c) {
            const float4 kv = sK[j * (DH / 4) + c];
            s0 = fmaf(q[4 * c + 0], kv.x, s0);
            s1 = fmaf(q[4 * c + 1], kv.y, s1);
            s0 = fmaf(q[4 * c + 2], kv.z, s0);
            s1 = fmaf(q[4 * c + 3], kv.w, s1);
        }
        float s = s0 + s1;
        if ((causal && j > i) || (kpm && kpm[(int64_t)b * L + j])) s = -INFINITY;
        srow[j] = s;
        mx = fmaxf(mx, s);
    }
    float sum = 0.0f;
    for (int j = 0; j < L; ++j) {
        const float p = expf(srow[j] - mx);
        srow[j] = p;
        sum += p;
    }
    const float inv = 1.0f / sum;
    float o[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) o[c] = 0.0f;
    for (int j = 0; j < L; ++j) {
        const float p = srow[j] * inv;
#pragma unroll
        for (int c = 0; c < DH / 4; ++c) {
            const float4 vv = sV[j * (DH / 4) + c];
            o[4 * c + 0] = fmaf(p, vv.x, o[4 * c + 0]);
            o[4 * c + 1] = fmaf(p, vv.y, o[4 * c + 1]);
            o[4 * c + 2] = fmaf(p, vv.z, o[4 * c + 2]);
            o[4 * c + 3] = fmaf(p, vv.w, o[4 * c + 3]);
        }
    }
    const int64_t orow_i = (int64_t)b * L + i;
#pragma unroll
    for (int c = 0; c < DH / 4; ++c) {
        if (out) *reinterpret_cast<float4*>(out + orow_i * D + h * DH + c * 4) = make_float4(o[4 * c + 0], o[4 * c + 1], o[4 * c + 2], o[4 * c + 3]);
        if (pl.hi) xmh::store_planes4(pl, orow_i, h * DH + c * 4, o[4 * c + 0], o[4 * c + 1], o[4 * c + 2], o[4 * c + 3]);
    }
}

// ---- the same attention on the fp32 MFMA for L <= 64 (ViT-B/32: L = 50, text: L = 32) ----------------------------
// One wave per (batch, head).  S = (q/sqrt(dh)) K^T and O = P V both run on v_mfma_f32_32x32x2_f32 (exact fp32 products,
// like the VALU kernel above).  Operand layout of that instruction: A and B give one float per lane, lane&31 = row of A /
// column of B, lane>>5 = which of the two k; the 32x32 result has column lane&31 and rows (e&3) + 8*(e>>2) + 4*(lane>>5).
// The sum over k does not care which k sits in which step, so lane (r, kk) simply owns the CONTIGUOUS half
// k in [32*kk, 32*kk+32) of its row: 8 float4 loads straight from global, no LDS staging of Q, K or V.
// The score tile goes through LDS once (C layout -> row layout); there lane (r, part) finishes the softmax of row r over
// columns [32*part, 32*part+32) with one cross-half exchange -- and the probabilities it then holds are exactly the A
// operand of the PV product (row r, k-half part), so they never go back to memory.
typedef float attn_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 attn_f16x8 __attribute__((ext_vector_type(8)));

// S16: the same dataflow on the fp16 MFMA with split operands (the GEMMs' parity scheme, xmh_planes.h): every operand register
// set is converted once to packed (hi, lo) halves -- x = hi + lo to 22 mantissa bits -- and every product becomes
// lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_f16 (fp32 accumulate, each fp16 x fp16 product exact; the dropped lo*lo term is
// 2^-22 relative).  12 MFMAs of 32 cycles per 32x32 block instead of 32 of 64: the matrix time of a head drops 5.3x, which is
// what bounds this kernel (one wave per head, two waves per SIMD).  The slab structure is the same trick as above: lane (r, kk)
// owns k in [32 kk, 32 kk + 32) of its row, slab s takes its elements [8 s, 8 s + 8).
struct AttnSplit {
    uint32_t hi[16], lo[16];
    __device__ __forceinline__ void set(const float (&f)[32]) {
#pragma unroll
        for (int t = 0; t < 16; ++t) xmh::split2(f[2 * t], f[2 * t + 1], hi[t], lo[t]);
    }
    __device__ __forceinline__ void set4(int c, float x, float y, float z, float w) {      // elements 4c .. 4c+3
        xmh::split2(x, y, hi[2 * c], lo[2 * c]);
        xmh::split2(z, w, hi[2 * c + 1], lo[2 * c + 1]);
    }
    __device__ __forceinline__ void set2(int t, float x, float y) { xmh::split2(x, y, hi[t], lo[t]); }                // elements 2t, 2t+1
    __device__ __forceinline__ attn_f16x8 h(int s) const { return __builtin_bit_cast(attn_f16x8, make_uint4(hi[4 * s], hi[4 * s + 1], hi[4 * s + 2], hi[4 * s + 3])); }
    __device__ __forceinline__ attn_f16x8 l(int s) const { return __builtin_bit_cast(attn_f16x8, make_uint4(lo[4 * s], lo[4 * s + 1], lo[4 * s + 2], lo[4 * s + 3])); }
};
__device__ __forceinline__ attn_f32x16 attn_mm_split(const AttnSplit& a, const AttnSplit& b) {
    attn_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {                                    // low parts first: the small terms meet the accumulator first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.l(s), b.h(s), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h(s), b.l(s), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h(s), b.h(s), acc, 0, 0, 0);
    }
    return acc;
}
__device__ __forceinline__ attn_f32x16 attn_mm_f32(const float (&a)[32], const float (&b)[32]) {
    attn_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
    for (int t = 0; t < 32; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
    return acc;
}

// Both variants keep the head's K and V blocks resident in registers (S16: as packed hi / lo halves, converted as they arrive) and
// run two waves per SIMD.  Reloading the K / V block per product to fit three or four waves was measured: 54.8 us against 20.1 us
// per ViT layer (the V operand is 32 strided loads per block and lane).
template <bool S16>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_attention_mfma64(const float* __restrict__ qkv, int Lmax, int H, int causal,
                                                         const uint8_t* __restrict__ kpm, float* __restrict__ out, xmh::Planes pl,
                                                         const int32_t* __restrict__ offs) {
    constexpr int DH = 64, SP = 68;                                  // score row stride: 16-byte aligned, rows on distinct 16-B slots
    __shared__ __attribute__((aligned(16))) float sS[32 * SP];
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    // packed sequences (the text tower without its padding, xmh_text_forward_packed): this one's rows start at offs[b], and it is
    // offs[b + 1] - offs[b] tokens long; otherwise B sequences of Lmax rows
    const int64_t row0 = offs ? (int64_t)offs[b] : (int64_t)b * Lmax;
    const int L = offs ? offs[b + 1] - offs[b] : Lmax;
    const int nblk = (L + 31) / 32;                                  // 32-row blocks of queries / keys (1 or 2)
    const int D = H * DH;
    const int lane = threadIdx.x, r = lane & 31, kk = lane >> 5;
    const float* base = qkv + row0 * 3 * D + h * DH;
    const float scale = rsqrtf((float)DH);

    // K rows (columns of S) and V columns.  All loads are unconditional on a clamped row and zeroed by a select afterwards: a
    // branch per load costs more than the load.
    auto load_k = [&](int jb, float (&kf)[S16 ? 1 : 32], AttnSplit& ksp) {
        const int j = jb * 32 + r;
        const bool ok = j < L;
        const float4* p = reinterpret_cast<const float4*>(base + (int64_t)(ok ? j : L - 1) * 3 * D + D + 32 * kk);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float4 v = p[c];
            const float x = ok ? v.x : 0.0f, y = ok ? v.y : 0.0f, z = ok ? v.z : 0.0f, w = ok ? v.w : 0.0f;
            if constexpr (S16) ksp.set4(c, x, y, z, w);              // converted as it arrives: no fp32 copy of the operand is kept
            else { kf[4 * c] = x; kf[4 * c + 1] = y; kf[4 * c + 2] = z; kf[4 * c + 3] = w; }
        }
    };
    auto load_v = [&](int cb, float (&vf)[S16 ? 1 : 32], AttnSplit& vsp) {
#pragma unroll
        for (int t = 0; t < 32; t += 2) {
            float pair[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = 32 * kk + t + u;
                const int jc = j < L ? j : L - 1;
                const float v0 = base[(int64_t)jc * 3 * D + 2 * D + cb * 32 + r];
                pair[u] = j < L ? v0 : 0.0f;                         // key index = k of the PV product
            }
            if constexpr (S16) vsp.set2(t / 2, pair[0], pair[1]);
            else { vf[t] = pair[0]; vf[t + 1] = pair[1]; }
        }
    };
    uint32_t dead_keys = 0;                                          // bit t: key 32*kk + t is padding or masked for this batch row
#pragma unroll
    for (int t = 0; t < 32; ++t) {
        const int j = 32 * kk + t;
        const uint32_t m = (kpm && j < L) ? (uint32_t)kpm[(int64_t)b * Lmax + j] : 0u;      // the mask keeps the padded [B, Lmax] layout in packed mode too
        dead_keys |= ((j >= L || m != 0u) ? 1u : 0u) << t;
    }
    float kb[2][S16 ? 1 : 32], vb[2][S16 ? 1 : 32];
    AttnSplit ks[S16 ? 2 : 1], vs[S16 ? 2 : 1];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        load_k(t, kb[t], ks[S16 ? t : 0]);
        load_v(t, vb[t], vs[S16 ? t : 0]);
    }

    for (int ib = 0; ib < nblk; ++ib) {
        const int i = ib * 32 + r;                                   // this lane's query row (A operand / softmax row)
        float qa[S16 ? 1 : 32];
        AttnSplit qs;
        {
            const float4* p = reinterpret_cast<const float4*>(base + (int64_t)(i < L ? i : L - 1) * 3 * D + 32 * kk);
#pragma unroll
            for (int c = 0; c < 8; ++c) {                            // PyTorch scales q before QK^T; rows >= L are never stored
                const float4 v = p[c];
                if constexpr (S16) qs.set4(c, v.x * scale, v.y * scale, v.z * scale, v.w * scale);
                else { qa[4 * c] = v.x * scale; qa[4 * c + 1] = v.y * scale; qa[4 * c + 2] = v.z * scale; qa[4 * c + 3] = v.w * scale; }
            }
        }
        __syncthreads();                                             // previous block's reads of the score tile are done
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            if (jb >= nblk) break;
            attn_f32x16 acc;
            if constexpr (S16) acc = attn_mm_split(qs, ks[jb]);
            else acc = attn_mm_f32(qa, kb[jb]);
#pragma unroll
            for (int e = 0; e < 16; ++e) sS[((e & 3) + 8 * (e >> 2) + 4 * kk) * SP + jb * 32 + r] = acc[e];
        }
        __syncthreads();
        // softmax of row r (query i) over columns [32*kk, 32*kk+32); the other half sits in lane r + 32 * (1 - kk)
        float pa[32];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(&sS[r * SP + 32 * kk + 4 * c]);
            pa[4 * c] = v.x; pa[4 * c + 1] = v.y; pa[4 * c + 2] = v.z; pa[4 * c + 3] = v.w;
        }
        uint32_t dead = dead_keys;
        if (causal) {                                                // keys j > i: bits t > i - 32*kk
            const int first = i + 1 - 32 * kk;                       // first dead bit
            dead |= first <= 0 ? 0xffffffffu : (first >= 32 ? 0u : (0xffffffffu << first));
        }
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            pa[t] = ((dead >> t) & 1u) ? -INFINITY : pa[t];
            mx = fmaxf(mx, pa[t]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.0f;
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            // S16: e^x as 2^(x log2 e) on v_exp_f32 (1 ulp; the arguments are <= 0 and, for weights that matter, small): the split
            // product that follows carries 2^-22 itself.  The fp32 variant keeps expf.
            pa[t] = S16 ? __builtin_amdgcn_exp2f((pa[t] - mx) * 1.4426950408889634f) : expf(pa[t] - mx);
            sum += pa[t];
        }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        AttnSplit ps;
        if constexpr (S16) {
#pragma unroll
            for (int t = 0; t < 16; ++t) ps.set2(t, pa[2 * t] * inv, pa[2 * t + 1] * inv);
        } else {
#pragma unroll
            for (int t = 0; t < 32; ++t) pa[t] *= inv;
        }
        // O[i][c] = sum_j P[i][j] V[j][c]; the 32 x 64 result goes through the score tile once more (C layout -> rows) so that
        // every lane stores 4 consecutive columns: 16 lanes cover 256 bytes of one fp32 row / 128 bytes of each fp16 plane
        __syncthreads();                                             // all softmax reads of sS are done
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            attn_f32x16 acc;
            if constexpr (S16) acc = attn_mm_split(ps, vs[cb]);
            else acc = attn_mm_f32(pa, vb[cb]);
#pragma unroll
            for (int e = 0; e < 16; ++e) sS[((e & 3) + 8 * (e >> 2) + 4 * kk) * SP + cb * 32 + r] = acc[e];
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int rr = it * 4 + (lane >> 4), cc = (lane & 15) * 4;
            const int row = ib * 32 + rr;
            const float4 v = *reinterpret_cast<const float4*>(&sS[rr * SP + cc]);
            if (row < L) {
                const int64_t orow = row0 + row;
                if (out) *reinterpret_cast<float4*>(out + orow * D + h * DH + cc) = v;
                if (pl.hi) xmh::store_planes4(pl, orow, h * DH + cc, v.x, v.y, v.z, v.w);
            }
        }
    }
}

// ---- patch gather: cols[(b*G*G + gy*G + gx)][c*P*P + dy*P + dx] = image[b][c][gy*P+dy][gx*P+dx] --------
__global__ __launch_bounds__(256) void k_im2col_patch(const float* __restrict__ img, int64_t B, int Cin, int res, int P,
                                                      float* __restrict__ cols, xmh::Planes pl) {
    const int G = res / P;
    const int64_t kdim = (int64_t)Cin * P * P;
    const int64_t total4 = B * G * G * kdim / 4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total4; e += (int64_t)gridDim.x * 256) {
        const int64_t flat = e * 4;
        const int64_t row = flat / kdim;
        const int k = (int)(flat % kdim);
        const int c = k / (P * P), dy = (k / P) % P, dx = k % P;              // dx is a multiple of 4
        const int64_t b = row / (G * G);
        const int g = (int)(row % (G * G)), gy = g / G, gx = g % G;
        const float4 v = *reinterpret_cast<const float4*>(img + ((b * Cin + c) * res + gy * P + dy) * res + gx * P + dx);
        if (cols) *reinterpret_cast<float4*>(cols + flat) = v;
        if (pl.hi) xmh::store_planes4(pl, row, k, v.x, v.y, v.z, v.w);
    }
}

// ---- x[b][0] = cls + pos[0]; x[b][1+p] = patch[b*NP+p] + pos[1+p]; then LayerNorm (ln_pre) ----------------
__global__ __launch_bounds__(256) void k_vit_assemble(const float* __restrict__ patch, const float* __restrict__ cls,
                                                      const float* __restrict__ pos, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps, float* __restrict__ x,
                                                      int64_t B, int NP, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int L = NP + 1;
    if (row >= B * L) return;
    const int64_t b = row / L;
    const int t = (int)(row % L);
    const float* src = t == 0 ? cls : patch + (b * NP + (t - 1)) * D;
    float v[kLnMaxPerLane];
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < kLnMaxPerLane; ++j) {
        const int c = j * 64 + lane;
        v[j] = c < D ? src[c] + pos[(int64_t)t * D + c] : 0.0f;
        s += v[j];
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.0f;
#pragma unroll
    for (int j = 0; j < kLnMaxPerLane; ++j) {
        const int c = j * 64 + lane;
        const float d = c < D ? v[j] - mean : 0.0f;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    float* yr = x + row * D;
#pragma unroll
    for (int j = 0; j < kLnMaxPerLane; ++j) {
        const int c = j * 64 + lane;
        if (c < D) yr[c] = (v[j] - mean) * rstd * gamma[c] + beta[c];
    }
}

// ---- text: x[b][l] = tok_emb[ids[b][l]] + pos[l]; eos[b] = argmax_l ids[b][l] (first maximum) -----------
__global__ __launch_bounds__(256) void k_text_embed(const int64_t* __restrict__ ids, const float* __restrict__ tok,
                                                    const float* __restrict__ pos, float* __restrict__ x,
                                                    int32_t* __restrict__ eos, int64_t B, int L, int D, int vocab) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * L) return;
    const int64_t b = row / L;
    const int l = (int)(row % L);
    int64_t id = ids[row];
    if (id < 0) id = 0;
    if (id >= vocab) id = vocab - 1;
    for (int c = lane; c < D; c += 64) x[row * D + c] = tok[id * D + c] + pos[(int64_t)l * D + c];
    if (l == 0 && lane == 0 && eos) {
        int best = 0;
        int64_t bv = ids[b * L];
        for (int j = 1; j < L; ++j) {
            const int64_t v = ids[b * L + j];
            if (v > bv) {
                bv = v;
                best = j;
            }
        }
        eos[b] = best;
    }
}

// the same rows without the padding: caption b keeps its first offs[b + 1] - offs[b] tokens (up to and including EOS), stored from
// row offs[b] on.  Same arithmetic per element as k_text_embed.
__global__ __launch_bounds__(256) void k_text_embed_packed(const int64_t* __restrict__ ids, const float* __restrict__ tok, const float* __restrict__ pos,
                                                           float* __restrict__ x, const int32_t* __restrict__ offs, int64_t B, int L, int D, int vocab) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * L) return;
    const int64_t b = row / L;
    const int l = (int)(row % L);
    const int o0 = offs[b];
    if (l >= offs[b + 1] - o0) return;
    int64_t id = ids[row];
    if (id < 0) id = 0;
    if (id >= vocab) id = vocab - 1;
    float* dst = x + (int64_t)(o0 + l) * D;
    for (int c = lane; c < D; c += 64) dst[c] = tok[id * D + c] + pos[(int64_t)l * D + c];
}

// Caption lengths, counted where the ids are (xmh_text_forward_packed_dev): one block; thread t takes captions t, t + 1024, ...;
// eos = first maximum of the ids (what CLIP.encode_text's argmax picks, models/CLIP/model.py:392); rows kept = up to EOS, or up to the last
// position the key padding mask leaves visible if that lies further back (rows the mask hides behind it are never kept: nothing may read
// them); offs = exclusive prefix over the captions in tiles of 1024 with a running carry.
__global__ __launch_bounds__(1024) void k_caption_offsets(const int64_t* __restrict__ ids, const uint8_t* __restrict__ kpm, int64_t B, int L,
                                                          int32_t* __restrict__ offs, int32_t* __restrict__ eos) {
    __shared__ int32_t part[1024];
    __shared__ int32_t carry;
    const int t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int64_t b0 = 0; b0 < B; b0 += 1024) {
        const int64_t b = b0 + t;
        int len = 0;
        if (b < B) {
            // eight ids (and mask bytes) per round, their loads issued together: one dependent load per token made this one-block kernel
            // 37 us at batch 400 (a thread walks its own 256-byte row, nothing coalesces)
            int best = 0, last_visible = -1;
            int64_t bv = ids[b * L];
            for (int j0 = 0; j0 < L; j0 += 8) {
                int64_t v[8];
                uint8_t m[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = j0 + u < L ? j0 + u : L - 1;
                    v[u] = ids[b * L + j];
                    m[u] = kpm ? kpm[b * L + j] : (uint8_t)1;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = j0 + u;
                    if (j >= L) break;
                    if (v[u] > bv) { bv = v[u]; best = j; }            // first maximum: strictly greater only
                    if (m[u] == 0) last_visible = j;
                }
            }
            len = (last_visible > best ? last_visible : best) + 1;
            if (eos) eos[b] = best;
        }
        part[t] = len;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {                          // inclusive scan of the tile (Hillis-Steele: 10 steps)
            const int v = t >= d ? part[t - d] : 0;
            __syncthreads();
            part[t] += v;
            __syncthreads();
        }
        if (b < B) offs[b] = carry + part[t] - len;
        __syncthreads();
        if (t == 1023) carry += part[1023];
        __syncthreads();
    }
    if (t == 0) offs[B] = carry;
}

__global__ __launch_bounds__(256) void k_unpack_rows(const float* __restrict__ packed, int64_t ldp, const int32_t* __restrict__ offs, float* __restrict__ out,
                                                     int64_t B, int L, int D4) {
    const int64_t total = B * L * D4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / D4;
        const int c = (int)(e % D4) * 4;
        const int64_t b = row / L;
        const int l = (int)(row % L);
        const int o0 = offs[b];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (l < offs[b + 1] - o0) v = *reinterpret_cast<const float4*>(packed + (int64_t)(o0 + l) * ldp + c);
        *reinterpret_cast<float4*>(out + row * (int64_t)D4 * 4 + c) = v;
    }
}

__global__ __launch_bounds__(256) void k_gather_packed_rows(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ offs, const int32_t* __restrict__ idx,
                                                            float* __restrict__ out, int64_t rows, int D) {
    const int64_t total = rows * D;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / D;
        out[e] = x[(int64_t)(offs[r] + idx[r]) * ldx + (int)(e % D)];
    }
}

__global__ __launch_bounds__(256) void k_gather_last_rows(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ offs, float* __restrict__ out,
                                                          int64_t rows, int D) {
    const int64_t total = rows * D;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / D;
        out[e] = x[(int64_t)(offs[r + 1] - 1) * ldx + (int)(e % D)];
    }
}

// out[r] = x[(r*group + (idx ? idx[r] : offset))]: idx==NULL -> fixed offset inside each group of rows
__global__ __launch_bounds__(256) void k_gather_rows(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ idx,
                                                     int offset, int group, float* __restrict__ out, int64_t rows, int D) {
    const int64_t total = rows * D;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / D;
        const int c = (int)(e % D);
        const int64_t srcrow = r * group + (idx ? idx[r] : offset);
        out[e] = x[srcrow * ldx + c];
    }
}

__global__ __launch_bounds__(256) void k_affine_cols(const float* __restrict__ x, const float* __restrict__ mean,
                                                     const float* __restrict__ var, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, float* __restrict__ y,
                                                     int64_t rows, int D) {
    const int64_t total = rows * D;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % D);
        // ATen's eval-mode batch_norm: alpha = weight / sqrt(var + eps), beta' = bias - mean * alpha, y = x * alpha + beta'
        const float alpha = (1.0f / sqrtf(var[c] + eps)) * gamma[c];
        y[e] = x[e] * alpha + (beta[c] - mean[c] * alpha);
    }
}

__global__ __launch_bounds__(256) void k_pair_softmax(const float* __restrict__ x, float* __restrict__ y, int64_t pairs) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < pairs; e += (int64_t)gridDim.x * 256) {
        const float2 v = reinterpret_cast<const float2*>(x)[e];
        const float m = fmaxf(v.x, v.y);
        const float e0 = expf(v.x - m), e1 = expf(v.y - m);
        const float s = e0 + e1;
        reinterpret_cast<float2*>(y)[e] = make_float2(e0 / s, e1 / s);
    }
}


// ---- MITH LocalizedTokenAggregation (models/MITH/hash/hash.py:109-169) + positional encoding (:41-65) ----------
// One block per sample.  S [B, L, K] concept scores (tanh outputs), X [B, L, D] raw CLIP tokens.
//   sim = S (+ -inf on masked tokens); sim = sim > 0 ? sim : -inf                                   (:142-153)
//   per token keep the concepts >= its 8th-largest score (ties kept), others -inf                     (:114-124)
//   softmax over the TOKEN axis per (sample, concept); a concept with no token gives NaN -> 0         (:159-160)
//   M[b,k,:] = sum_l A[b,k,l] X[b,l,:]   (+ pe[k,:], the sin/cos table already divided by sqrt(D))    (:164-168, :63)
__global__ __launch_bounds__(256) void k_lta_aggregate(const float* __restrict__ S, const float* __restrict__ X,
                                                       const uint8_t* __restrict__ mask, const float* __restrict__ pe,
                                                       float* __restrict__ M, int L, int K, int D, int topk) {
    // grid (B, ceil(D / 64)): every block rebuilds the [L][K] attention of its sample (cheap, all 256 threads) and aggregates
    // 64 feature columns -- one block per sample left 60 % of the CUs idle and ran 0.46 ms
    extern __shared__ __attribute__((aligned(16))) float sm[];      // [L][K+1] scores -> attention, then thr[L]
    const int b = blockIdx.x;
    const int ld = K + 1;
    float* thr_s = sm + L * ld;
    for (int e = threadIdx.x; e < L * K; e += 256) {
        const int l = e / K, k = e % K;
        float v = S[((int64_t)b * L + l) * K + k];
        if (mask && mask[(int64_t)b * L + l]) v = -INFINITY;
        sm[l * ld + k] = v > 0.0f ? v : -INFINITY;
    }
    for (int l = threadIdx.x; l < L; l += 256) thr_s[l] = -INFINITY;
    __syncthreads();
    // per-token top-k threshold: the value v with #(> v) < topk <= #(>= v); one (token, candidate) pair per thread step.
    // Several candidates of a row can qualify only if they are equal, so the racing stores write the same value.
    for (int e = threadIdx.x; e < L * K; e += 256) {
        const int l = e / K, i = e % K;
        const float* row = sm + l * ld;
        const float v = row[i];
        int gt = 0, ge = 0;
        for (int j = 0; j < K; ++j) {
            gt += row[j] > v;
            ge += row[j] >= v;
        }
        if (gt < topk && topk <= ge) thr_s[l] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < L * K; e += 256) {
        const int l = e / K, i = e % K;
        if (!(sm[l * ld + i] >= thr_s[l])) sm[l * ld + i] = -INFINITY;
    }
    __syncthreads();
    // softmax over tokens per concept
    for (int k = threadIdx.x; k < K; k += 256) {
        float mx = -INFINITY;
        for (int l = 0; l < L; ++l) mx = fmaxf(mx, sm[l * ld + k]);
        if (mx == -INFINITY) {
            for (int l = 0; l < L; ++l) sm[l * ld + k] = 0.0f;                       // NaN -> 0 in the reference
        } else {
            float sum = 0.0f;
            for (int l = 0; l < L; ++l) {
                const float p = expf(sm[l * ld + k] - mx);
                sm[l * ld + k] = p;
                sum += p;
            }
            for (int l = 0; l < L; ++l) sm[l * ld + k] = sm[l * ld + k] / sum;
        }
    }
    __syncthreads();
    // aggregate 64 feature columns: thread = (column, quarter of the concepts), 16 concepts at a time in registers
    const int d = blockIdx.y * 64 + (threadIdx.x & 63);
    const int kq = threadIdx.x >> 6;                                 // 0..3
    if (d >= D) return;
    for (int k0 = kq * 16; k0 < K; k0 += 64) {
        float acc[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u] = 0.0f;
        for (int l = 0; l < L; ++l) {
            const float x = X[((int64_t)b * L + l) * D + d];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (k0 + u < K) acc[u] = fmaf(sm[l * ld + k0 + u], x, acc[u]);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (k0 + u < K) M[((int64_t)b * K + k0 + u) * D + d] = acc[u] + (pe ? pe[(int64_t)(k0 + u) * D + d] : 0.0f);
    }
}

// ---- MITH BitwiseHashing (models/MITH/hash/hash.py:68-85): out[b,k] = tanh(w_k . z[b,k,:] + bias_k) (+ addend) ----
__global__ __launch_bounds__(256) void k_bitwise_hash(const float* __restrict__ Z, const float* __restrict__ Wb,
                                                      const float* __restrict__ bias, const float* __restrict__ addend,
                                                      float* __restrict__ out, int64_t rows, int K, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);          // r = b*K + k
    if (r >= rows) return;
    const int k = (int)(r % K);
    float s = 0.0f;
    for (int c = lane; c < D; c += 64) s = fmaf(Z[r * D + c], Wb[(int64_t)k * D + c], s);
    s = wave_sum(s);
    if (lane == 0) out[r] = tanhf(s + bias[k]) + (addend ? addend[r] : 0.0f);
}

inline int grid1d(int64_t work, int per_block = 256) {
    int64_t g = xmh::ceil_div(work, per_block);
    const int64_t cap = (int64_t)xmh::device_cu_count() * 16;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

namespace xmh {

int layernorm_planes(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, float* y, int64_t ldy, const Planes& p,
                     int64_t rows, int D, hipStream_t st, const int32_t* rows_dev) {
    if (rows < 0 || D <= 0 || D > 64 * kLnMaxPerLane) return fail(XMH_EINVAL, "xmh_layernorm_f32: bad shape rows=%lld D=%d (D <= %d)", (long long)rows, D, 64 * kLnMaxPerLane);
    if (rows == 0) return XMH_OK;
    if (!x || !gamma || !beta || (!y && !p.hi)) return fail(XMH_EINVAL, "xmh_layernorm_f32: null pointer");
    const bool vec = D % 4 == 0 && ldx % 4 == 0 && (!y || ldy % 4 == 0) && (!p.hi || p.ld % 4 == 0) &&
                     (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) % 16 == 0 &&
                     (reinterpret_cast<uintptr_t>(p.hi) | reinterpret_cast<uintptr_t>(p.lo)) % 8 == 0;
    if (vec) hipLaunchKernelGGL(k_layernorm4, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, st, x, ldx, gamma, beta, eps, y, ldy, p, rows, D, rows_dev);
    else if (rows_dev) return fail(XMH_ENOTSUP, "xmh layernorm: a device-side row count needs D %% 4 == 0 and 16-byte aligned rows (D=%d)", D);
    else if (p.hi) return fail(XMH_ENOTSUP, "xmh layernorm: operand planes need D %% 4 == 0 and 16-byte aligned rows (D=%d)", D);
    else hipLaunchKernelGGL(k_layernorm, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, st, x, ldx, gamma, beta, eps, y, ldy, rows, D);
    XMH_LAUNCH_CHECK("xmh_layernorm_f32");
    return XMH_OK;
}

int attention_planes(const float* qkv, int64_t B, int L, int H, int dh, int causal, const uint8_t* key_padding_mask, float* out, const Planes& p,
                     bool split16, hipStream_t st, const int32_t* row_offsets) {
    if (B < 0 || L <= 0 || H <= 0) return fail(XMH_EINVAL, "xmh_attention_f32: bad shape");
    if (B == 0) return XMH_OK;
    if (dh != 64) return fail(XMH_ENOTSUP, "xmh_attention_f32: head dim %d (only 64, CLIP's width/heads)", dh);
    if (L > 128) return fail(XMH_ENOTSUP, "xmh_attention_f32: L=%d > 128 (whole head must fit LDS)", L);
    if (!qkv || (!out && !p.hi)) return fail(XMH_EINVAL, "xmh_attention_f32: null pointer");
    static const bool valu_only = xmh_experiment_env("XMH_ATTENTION_VALU") != nullptr;
    if (row_offsets && L > 64) return fail(XMH_ENOTSUP, "xmh attention: packed sequences need L <= 64 (L=%d)", L);
    if (L <= 64 && (!valu_only || row_offsets)) {                    // fp32-MFMA kernel: one wave per head
        if (split16) hipLaunchKernelGGL(k_attention_mfma64<true>, dim3((unsigned)(B * H)), dim3(64), 0, st, qkv, L, H, causal, key_padding_mask, out, p, row_offsets);
        else hipLaunchKernelGGL(k_attention_mfma64<false>, dim3((unsigned)(B * H)), dim3(64), 0, st, qkv, L, H, causal, key_padding_mask, out, p, row_offsets);
        XMH_LAUNCH_CHECK("xmh_attention_f32");
        return XMH_OK;
    }
    const size_t lds = ((size_t)2 * L * dh + (size_t)L * (L + 1)) * 4;
    const int threads = L <= 64 ? 64 : 128;
    auto kern = k_attention<64>;
    if (const int rl = xmh::raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds, "xmh_attention_f32")) return rl;
    hipLaunchKernelGGL(kern, dim3((unsigned)(B * H)), dim3(threads), lds, st, qkv, L, H, causal, key_padding_mask, out, p);
    XMH_LAUNCH_CHECK("xmh_attention_f32");
    return XMH_OK;
}

int text_embed_packed(const int64_t* ids, const float* tok_emb, const float* pos, float* x, const int32_t* row_offsets, int64_t B, int L, int D,
                      int vocab, hipStream_t st) {
    if (B < 0 || L <= 0 || D <= 0 || vocab <= 0) return fail(XMH_EINVAL, "xmh text_embed_packed: bad shape");
    if (B == 0) return XMH_OK;
    if (!ids || !tok_emb || !pos || !x || !row_offsets) return fail(XMH_EINVAL, "xmh text_embed_packed: null pointer");
    hipLaunchKernelGGL(k_text_embed_packed, dim3((unsigned)ceil_div(B * L, 4)), dim3(256), 0, st, ids, tok_emb, pos, x, row_offsets, B, L, D, vocab);
    XMH_LAUNCH_CHECK("xmh text_embed_packed");
    return XMH_OK;
}

int gather_last_rows(const float* x, int64_t ldx, const int32_t* row_offsets, float* out, int64_t B, int D, hipStream_t st) {
    if (B < 0 || D <= 0) return fail(XMH_EINVAL, "xmh gather_last_rows: bad shape");
    if (B == 0) return XMH_OK;
    if (!x || !out || !row_offsets) return fail(XMH_EINVAL, "xmh gather_last_rows: null pointer");
    hipLaunchKernelGGL(k_gather_last_rows, dim3(grid1d(B * D)), dim3(256), 0, st, x, ldx, row_offsets, out, B, D);
    XMH_LAUNCH_CHECK("xmh gather_last_rows");
    return XMH_OK;
}

int caption_offsets(const int64_t* ids, const uint8_t* key_padding_mask, int64_t B, int L, int32_t* offs, int32_t* eos, hipStream_t st) {
    if (B < 0 || L <= 0) return fail(XMH_EINVAL, "xmh caption_offsets: bad shape");
    if (!ids || !offs) return fail(XMH_EINVAL, "xmh caption_offsets: null pointer");
    if (B * (int64_t)L >= (1ll << 31)) return fail(XMH_ENOTSUP, "xmh caption_offsets: %lld x %d tokens", (long long)B, L);
    hipLaunchKernelGGL(k_caption_offsets, dim3(1), dim3(1024), 0, st, ids, key_padding_mask, B, L, offs, eos);
    XMH_LAUNCH_CHECK("xmh caption_offsets");
    return XMH_OK;
}

int unpack_rows(const float* packed, int64_t ldp, const int32_t* offs, float* out, int64_t B, int L, int D, hipStream_t st) {
    if (B < 0 || L <= 0 || D <= 0 || D % 4 || ldp % 4) return fail(XMH_EINVAL, "xmh unpack_rows: bad shape");
    if (B == 0) return XMH_OK;
    if (!packed || !offs || !out || (reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(out)) % 16) return fail(XMH_EINVAL, "xmh unpack_rows: null or unaligned pointer");
    hipLaunchKernelGGL(k_unpack_rows, dim3(grid1d(B * L * (D / 4))), dim3(256), 0, st, packed, ldp, offs, out, B, L, D / 4);
    XMH_LAUNCH_CHECK("xmh unpack_rows");
    return XMH_OK;
}

int gather_packed_rows(const float* packed, int64_t ldp, const int32_t* offs, const int32_t* idx, float* out, int64_t B, int D, hipStream_t st) {
    if (B < 0 || D <= 0) return fail(XMH_EINVAL, "xmh gather_packed_rows: bad shape");
    if (B == 0) return XMH_OK;
    if (!packed || !offs || !idx || !out) return fail(XMH_EINVAL, "xmh gather_packed_rows: null pointer");
    hipLaunchKernelGGL(k_gather_packed_rows, dim3(grid1d(B * D)), dim3(256), 0, st, packed, ldp, offs, idx, out, B, D);
    XMH_LAUNCH_CHECK("xmh gather_packed_rows");
    return XMH_OK;
}

int im2col_planes(const float* image, int64_t B, int channels, int resolution, int patch, float* cols, const Planes& p, hipStream_t st) {
    if (B < 0 || channels <= 0 || patch <= 0 || resolution % patch || patch % 4) return fail(XMH_EINVAL, "xmh_im2col_patch: bad geometry res=%d patch=%d", resolution, patch);
    if (B == 0) return XMH_OK;
    if (!image || (!cols && !p.hi)) return fail(XMH_EINVAL, "xmh_im2col_patch: null pointer");
    const int64_t total4 = B * channels * resolution * resolution / 4;
    hipLaunchKernelGGL(k_im2col_patch, dim3(grid1d(total4)), dim3(256), 0, st, image, B, channels, resolution, patch, cols, p);
    XMH_LAUNCH_CHECK("xmh_im2col_patch");
    return XMH_OK;
}

}  // namespace xmh

extern "C" int xmh_layernorm_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, float* y,
                                 int64_t ldy, int64_t rows, int D, xmh_stream_t stream) {
    if (rows > 0 && !y) return xmh::fail(XMH_EINVAL, "xmh_layernorm_f32: null pointer");
    return xmh::layernorm_planes(x, ldx, gamma, beta, eps, y, ldy, xmh::Planes{nullptr, nullptr, 0}, rows, D, xmh::as_stream(stream));
}

extern "C" int xmh_attention_f32(const float* qkv, int64_t B, int L, int H, int dh, int causal, const uint8_t* key_padding_mask,
                                 float* out, xmh_stream_t stream) {
    if (B > 0 && !out) return xmh::fail(XMH_EINVAL, "xmh_attention_f32: null pointer");
    return xmh::attention_planes(qkv, B, L, H, dh, causal, key_padding_mask, out, xmh::Planes{nullptr, nullptr, 0}, false, xmh::as_stream(stream));
}

extern "C" int xmh_attention_split16(const float* qkv, int64_t B, int L, int H, int dh, int causal, const uint8_t* key_padding_mask,
                                     float* out, xmh_stream_t stream) {
    if (B > 0 && !out) return xmh::fail(XMH_EINVAL, "xmh_attention_split16: null pointer");
    return xmh::attention_planes(qkv, B, L, H, dh, causal, key_padding_mask, out, xmh::Planes{nullptr, nullptr, 0}, true, xmh::as_stream(stream));
}

extern "C" int xmh_im2col_patch(const float* image, int64_t B, int channels, int resolution, int patch, float* cols,
                                xmh_stream_t stream) {
    if (B > 0 && !cols) return xmh::fail(XMH_EINVAL, "xmh_im2col_patch: null pointer");
    return xmh::im2col_planes(image, B, channels, resolution, patch, cols, xmh::Planes{nullptr, nullptr, 0}, xmh::as_stream(stream));
}

extern "C" int xmh_vit_assemble(const float* patch_out, const float* cls, const float* pos, const float* gamma, const float* beta,
                                float eps, float* x, int64_t B, int n_patches, int D, xmh_stream_t stream) {
    if (B < 0 || n_patches <= 0 || D <= 0 || D > 64 * kLnMaxPerLane) return xmh::fail(XMH_EINVAL, "xmh_vit_assemble: bad shape");
    if (B == 0) return XMH_OK;
    if (!patch_out || !cls || !pos || !gamma || !beta || !x) return xmh::fail(XMH_EINVAL, "xmh_vit_assemble: null pointer");
    hipLaunchKernelGGL(k_vit_assemble, dim3((unsigned)xmh::ceil_div(B * (n_patches + 1), 4)), dim3(256), 0, xmh::as_stream(stream), patch_out, cls, pos,
                       gamma, beta, eps, x, B, n_patches, D);
    XMH_LAUNCH_CHECK("xmh_vit_assemble");
    return XMH_OK;
}

extern "C" int xmh_text_embed(const int64_t* ids, const float* tok_emb, const float* pos, float* x, int32_t* eos_index, int64_t B,
                              int L, int D, int vocab, xmh_stream_t stream) {
    if (B < 0 || L <= 0 || D <= 0 || vocab <= 0) return xmh::fail(XMH_EINVAL, "xmh_text_embed: bad shape");
    if (B == 0) return XMH_OK;
    if (!ids || !tok_emb || !pos || !x) return xmh::fail(XMH_EINVAL, "xmh_text_embed: null pointer");
    hipLaunchKernelGGL(k_text_embed, dim3((unsigned)xmh::ceil_div(B * L, 4)), dim3(256), 0, xmh::as_stream(stream), ids, tok_emb, pos, x, eos_index, B, L, D, vocab);
    XMH_LAUNCH_CHECK("xmh_text_embed");
    return XMH_OK;
}

extern "C" int xmh_gather_rows(const float* x, int64_t ldx, const int32_t* idx, int offset, int group, float* out, int64_t rows,
                               int D, xmh_stream_t stream) {
    if (rows < 0 || D <= 0 || group <= 0) return xmh::fail(XMH_EINVAL, "xmh_gather_rows: bad shape");
    if (rows == 0) return XMH_OK;
    if (!x || !out) return xmh::fail(XMH_EINVAL, "xmh_gather_rows: null pointer");
    hipLaunchKernelGGL(k_gather_rows, dim3(grid1d(rows * D)), dim3(256), 0, xmh::as_stream(stream), x, ldx, idx, offset, group, out, rows, D);
    XMH_LAUNCH_CHECK("xmh_gather_rows");
    return XMH_OK;
}

extern "C" int xmh_affine_cols(const float* x, const float* mean, const float* var, const float* gamma, const float* beta,
                               float eps, float* y, int64_t rows, int D, xmh_stream_t stream) {
    if (rows < 0 || D <= 0) return xmh::fail(XMH_EINVAL, "xmh_affine_cols: bad shape");
    if (rows == 0) return XMH_OK;
    if (!x || !mean || !var || !gamma || !beta || !y) return xmh::fail(XMH_EINVAL, "xmh_affine_cols: null pointer");
    hipLaunchKernelGGL(k_affine_cols, dim3(grid1d(rows * D)), dim3(256), 0, xmh::as_stream(stream), x, mean, var, gamma, beta, eps, y, rows, D);
    XMH_LAUNCH_CHECK("xmh_affine_cols");
    return XMH_OK;
}

extern "C" int xmh_pair_softmax(const float* x, float* y, int64_t rows, int K, xmh_stream_t stream) {
    if (rows < 0 || K <= 0) return xmh::fail(XMH_EINVAL, "xmh_pair_softmax: bad shape");
    if (rows == 0) return XMH_OK;
    if (!x || !y) return xmh::fail(XMH_EINVAL, "xmh_pair_softmax: null pointer");
    hipLaunchKernelGGL(k_pair_softmax, dim3(grid1d(rows * K)), dim3(256), 0, xmh::as_stream(stream), x, y, rows * K);
    XMH_LAUNCH_CHECK("xmh_pair_softmax");
    return XMH_OK;
}

extern "C" int xmh_lta_aggregate(const float* scores, const float* tokens, const uint8_t* token_mask, const float* pos_enc,
                                 float* out, int64_t B, int L, int K, int D, int top_k, xmh_stream_t stream) {
    if (B < 0 || L <= 0 || K <= 0 || D <= 0 || top_k <= 0 || top_k > K) return xmh::fail(XMH_EINVAL, "xmh_lta_aggregate: bad shape L=%d K=%d top_k=%d", L, K, top_k);
    if (B == 0) return XMH_OK;
    if (!scores || !tokens || !out) return xmh::fail(XMH_EINVAL, "xmh_lta_aggregate: null pointer");
    const size_t lds = ((size_t)L * (K + 1) + L) * 4;
    if (lds > 160 * 1024) return xmh::fail(XMH_ENOTSUP, "xmh_lta_aggregate: L*K too large for LDS");
    if (const int rl = xmh::raise_dynamic_lds(reinterpret_cast<const void*>(k_lta_aggregate), lds, "xmh_lta_aggregate")) return rl;
    hipLaunchKernelGGL(k_lta_aggregate, dim3((unsigned)B, (unsigned)xmh::ceil_div(D, 64)), dim3(256), lds, xmh::as_stream(stream), scores, tokens, token_mask, pos_enc, out, L, K, D, top_k);
    XMH_LAUNCH_CHECK("xmh_lta_aggregate");
    return XMH_OK;
}

extern "C" int xmh_bitwise_hash(const float* z, const float* w, const float* bias, const float* addend, float* out, int64_t B,
                                int K, int D, xmh_stream_t stream) {
    if (B < 0 || K <= 0 || D <= 0) return xmh::fail(XMH_EINVAL, "xmh_bitwise_hash: bad shape");
    if (B == 0) return XMH_OK;
    if (!z || !w || !bias || !out) return xmh::fail(XMH_EINVAL, "xmh_bitwise_hash: null pointer");
    hipLaunchKernelGGL(k_bitwise_hash, dim3((unsigned)xmh::ceil_div(B * K, 4)), dim3(256), 0, xmh::as_stream(stream), z, w, bias, addend, out, B * K, K, D);
    XMH_LAUNCH_CHECK("xmh_bitwise_hash");
    return XMH_OK;
}
