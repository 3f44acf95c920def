// fp16 "operand planes": how activations travel between the encoder kernels and the LDS-DMA staged GEMM (xmh_gemm.hip).
//
// The fp16 MFMA GEMM (k_gemm_g16) stages its operands with global_load_lds (no registers, no conversion on the way), so
// whatever feeds a GEMM stores its result directly in the GEMM's operand format instead of fp32:
//   fast mode   one plane   hi = half(x), round to nearest;
//   parity mode two planes  x = hi + lo + r: hi = x truncated to 11 significant bits (a mask: exactly an fp16 value inside the
//               fp16 exponent range), lo = half(x - hi) rounded toward zero, |r| <= 2^-21 |x| (for |x| below 2^-3 the low part is
//               subnormal: absolute error <= 2^-24).  Both parts come from v_cvt_pkrtz_f16_f32, which saturates at 65504
//               instead of producing inf.  Every fp16 x fp16 product is exact in fp32, so  acc += lo*w; acc += hi*w
//               reproduces the fp32 product to 2^-22 relative (reference weights are fp16 values held in fp32,
//               models/CLIP/model.py:415-436).
// Same bytes per element as fp32 in parity mode (2 + 2), half in fast mode.  The split is a pure function of the fp32 value, so
// a producer that emits planes directly (LayerNorm, attention, a GEMM epilogue, im2col) and the stand-alone pass
// (xmh::split_planes) over the same fp32 values give the same bits -- which is what keeps the fused forward and the
// per-primitive chain identical.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace xmh {

struct Planes {
    _Float16* hi;
    _Float16* lo;      // nullptr: one plane (fast mode)
    int64_t ld;
};

#ifdef __HIPCC__
// two floats -> packed (hi, hi) and (lo, lo) halves
__device__ __forceinline__ void split2(float f0, float f1, uint32_t& H, uint32_t& L) {
    const float h0 = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, f0) & 0xffffe000u);
    const float h1 = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, f1) & 0xffffe000u);
    H = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(h0, h1));
    L = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(f0 - h0, f1 - h1));
}
__device__ __forceinline__ uint32_t round2(float f0, float f1) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 v;
    v[0] = (_Float16)f0;
    v[1] = (_Float16)f1;
    return __builtin_bit_cast(uint32_t, v);
}
// four consecutive elements of a row (8-byte stores; col % 4 == 0, ld % 4 == 0)
__device__ __forceinline__ void store_planes4(const Planes& p, int64_t row, int64_t col, float a, float b, float c, float d) {
    if (p.lo) {
        uint2 h, l;
        split2(a, b, h.x, l.x);
        split2(c, d, h.y, l.y);
        *reinterpret_cast<uint2*>(p.hi + row * p.ld + col) = h;
        *reinterpret_cast<uint2*>(p.lo + row * p.ld + col) = l;
    } else {
        *reinterpret_cast<uint2*>(p.hi + row * p.ld + col) = make_uint2(round2(a, b), round2(c, d));
    }
}
__device__ __forceinline__ void store_planes1(const Planes& p, int64_t row, int64_t col, float a) {
    if (p.lo) {
        uint32_t h, l;
        split2(a, 0.0f, h, l);
        reinterpret_cast<uint16_t*>(p.hi)[row * p.ld + col] = (uint16_t)h;
        reinterpret_cast<uint16_t*>(p.lo)[row * p.ld + col] = (uint16_t)l;
    } else {
        p.hi[row * p.ld + col] = (_Float16)a;
    }
}
#endif

// x [rows][cols] fp32 (row stride ldx) -> planes (p.lo == nullptr: one rounded plane).  cols % 4 == 0, 16-byte aligned rows.
int split_planes(const float* x, int64_t ldx, int64_t rows, int64_t cols, const Planes& p, hipStream_t st);

// C[M,N] = act(A . W^T + bias) (+ residual) on the fp16 MFMA from operand planes.  A_lo / W_lo may be null (one plane each);
// K % 32 == 0, lda % 8 == 0, ldw % 8 == 0, 16-byte aligned bases.  Outputs: C (fp32, may be null) and / or the operand planes of
// the result for a following GEMM (O.hi may be null).
struct GemmPlanes {
    const _Float16 *A_hi, *A_lo;
    int64_t lda;
    const _Float16 *W_hi, *W_lo;
    int64_t ldw;
    const float* bias;
    const float* residual;
    int64_t ldr;
    float* C;
    int64_t ldc;
    Planes O;
    int64_t M, N, K;
    int act;
    const int32_t* m_dev = nullptr;   // device word with the real row count (<= M); M then only sizes the grid (xmh_gemm.hip, GArgsP)
};
bool gemm_planes_ok(int64_t K, int64_t lda, int64_t ldw, const void* A, const void* W);
int gemm_planes(const GemmPlanes& g, hipStream_t st);


// producers that emit operand planes directly (xmh_encode.hip); the fp32 output pointer may be null when only planes are wanted,
// p.hi may be null when only fp32 is wanted
int layernorm_planes(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, float* y, int64_t ldy, const Planes& p,
                     int64_t rows, int D, hipStream_t st, const int32_t* rows_dev = nullptr);      // rows_dev: the real row count on the device (<= rows)
// split16: the products on the fp16 MFMA with hi/lo split operands (parity / fast mode) instead of the fp32 MFMA (exact mode)
// row_offsets (device, [B + 1] i32, may be null): PACKED sequences -- sequence b owns rows [row_offsets[b], row_offsets[b + 1]) of qkv
// and of the outputs and is that many (1 .. L) tokens long; null: B sequences of exactly L rows each.  Packed form: L <= 64, no key
// padding mask.
int attention_planes(const float* qkv, int64_t B, int L, int H, int dh, int causal, const uint8_t* key_padding_mask, float* out, const Planes& p,
                     bool split16, hipStream_t st, const int32_t* row_offsets = nullptr);
// the packed text tower's first and last step (xmh_text_forward_packed): x[row_offsets[b] + l] = tok[ids[b][l]] + pos[l] for l below the
// caption's length; out[b] = x[row_offsets[b + 1] - 1] (the EOS row)
int text_embed_packed(const int64_t* ids, const float* tok_emb, const float* pos, float* x, const int32_t* row_offsets, int64_t B, int L, int D,
                      int vocab, hipStream_t st);
int gather_last_rows(const float* x, int64_t ldx, const int32_t* row_offsets, float* out, int64_t B, int D, hipStream_t st);
// caption lengths counted ON THE DEVICE: eos[b] = first maximum of ids[b] (CLIP.encode_text's argmax, models/CLIP/model.py:392), kept rows of
// caption b = max(eos[b] + 1, 1 + last position the key padding mask leaves visible), offs = their exclusive prefix (offs[B] = all rows)
int caption_offsets(const int64_t* ids, const uint8_t* key_padding_mask, int64_t B, int L, int32_t* offs, int32_t* eos, hipStream_t st);
// out[b][l] = l < offs[b + 1] - offs[b] ? packed[offs[b] + l] : 0   (the padded [B, L, D] layout of the reference; dropped rows are zero)
int unpack_rows(const float* packed, int64_t ldp, const int32_t* offs, float* out, int64_t B, int L, int D, hipStream_t st);
// out[b] = packed[offs[b] + idx[b]]
int gather_packed_rows(const float* packed, int64_t ldp, const int32_t* offs, const int32_t* idx, float* out, int64_t B, int D, hipStream_t st);
// f = QuickGELU(u) elementwise on [rows, cols] (cols % 4 == 0), as fp32 and / or operand planes: the c_fc epilogue's activation as a
// pass of its own, for the saved-activation forward (xmh_gemm.hip)
int quickgelu_planes(const float* u, int64_t rows, int64_t cols, float* f, const Planes& p, hipStream_t st);
int im2col_planes(const float* image, int64_t B, int channels, int resolution, int patch, float* cols, const Planes& p, hipStream_t st);

}  // namespace xmh
