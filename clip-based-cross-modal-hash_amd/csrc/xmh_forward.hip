// Whole-tower entry points (include/xmh.h, "Whole-tower entry points"): the CLIP ViT-B/32 image forward and the text forward
// of the reference (models/CLIP/model.py:167-268, :373-396) as one C call each.  Nothing is computed here: this file is the
// native executor that enqueues the kernel chain of xmh_encode.hip / xmh_gemm.hip on one stream, with every intermediate in a
// caller-owned workspace -- no allocation, no host synchronisation, so a caller may capture a call in a hipGraph.
//
//   image: im2col -> GEMM(conv1) -> cls/pos/ln_pre -> blocks -> ln_post -> GEMM proj   (cls row only, or every token)
//   text : embed + pos -> blocks (causal [+ key padding]) -> ln_final -> GEMM text_projection -> EOS row
//   block: LN, GEMM qkv, attention, GEMM out + residual, LN, GEMM c_fc + QuickGELU, GEMM c_proj + residual
#include "xmh_common.h"
#include "xmh_planes.h"

namespace {

constexpr int kActNone = 0, kActQuickGelu = 1, kActGeluErf = 2, kActTanh = 3, kActRelu = 4;
constexpr int kPrecParity = 0, kPrecFast = 1, kPrecExact = 2;
constexpr float kLnEps = 1e-5f;                      // nn.LayerNorm default, as in the reference

inline size_t align_up(size_t v) { return (v + 255) & ~size_t(255); }

// carve-out of the workspace; the same arithmetic sizes it (xmh_clip_workspace_bytes) and hands out the pieces
struct Arena {
    char* base;
    size_t used = 0;
    explicit Arena(void* p) : base(static_cast<char*>(p)) {}
    template <typename T>
    T* take(size_t count) {
        T* p = base ? reinterpret_cast<T*>(base + used) : nullptr;
        used += align_up(count * sizeof(T));
        return p;
    }
};

// Parity and fast mode keep every GEMM input as fp16 operand planes (xmh_planes.h), written by the kernel that produces it:
// LayerNorm -> qkv / c_fc, attention -> out_proj, the c_fc epilogue (QuickGELU) -> c_proj.  Only the residual stream x and the
// qkv rows the attention kernel reads stay fp32.  Exact mode (fp32 MFMA) keeps the fp32 buffers.
struct BlockScratch {
    float *h, *qkv, *a, *f;                          // exact mode: fp32 activations
    xmh::Planes hP, aP, fP;                          // parity / fast mode
    xmh::Planes any;                                 // planes of an fp32 activation that was not produced as planes (linear_any)
};

xmh::Planes carve_planes(Arena& ar, size_t rows, size_t cols, int precision) {
    xmh::Planes p;
    p.hi = reinterpret_cast<_Float16*>(ar.take<uint16_t>(rows * cols));
    p.lo = precision == kPrecParity ? reinterpret_cast<_Float16*>(ar.take<uint16_t>(rows * cols)) : nullptr;
    p.ld = (int64_t)cols;
    return p;
}

BlockScratch carve_blocks(Arena& ar, int64_t M, int width, int precision) {
    BlockScratch s{};
    s.qkv = ar.take<float>((size_t)M * width * 3);
    if (precision == kPrecExact) {
        s.h = ar.take<float>((size_t)M * width);
        s.a = ar.take<float>((size_t)M * width);
        s.f = ar.take<float>((size_t)M * width * 4);
    } else {
        s.hP = carve_planes(ar, (size_t)M, (size_t)width, precision);
        s.aP = carve_planes(ar, (size_t)M, (size_t)width, precision);
        s.fP = carve_planes(ar, (size_t)M, (size_t)width * 4, precision);
        s.any = s.fP;                                // free between blocks: the towers' other GEMMs run before / after the stack
    }
    return s;
}

bool planes_layer(const xmh_linear& l) { return l.w_hi && l.k % 32 == 0; }

// act(A @ W^T + bias) (+ residual) from operand planes; C and / or the result's own planes
int linear_p(const xmh_linear& l, const xmh::Planes& A, const float* residual, int64_t ldr, float* C, int64_t ldc, const xmh::Planes* out,
             int64_t M, int act, int precision, xmh_stream_t st, const int32_t* m_dev = nullptr) {
    xmh::GemmPlanes g{};
    g.A_hi = A.hi; g.A_lo = precision == kPrecParity ? A.lo : nullptr; g.lda = A.ld;
    g.W_hi = static_cast<const _Float16*>(l.w_hi);
    g.W_lo = precision == kPrecParity ? static_cast<const _Float16*>(l.w_lo) : nullptr;
    g.ldw = l.k;
    g.bias = l.bias; g.residual = residual; g.ldr = ldr; g.C = C; g.ldc = ldc;
    if (out) g.O = *out;
    g.M = M; g.N = l.n; g.K = l.k; g.act = act;
    g.m_dev = m_dev;
    return xmh::gemm_planes(g, xmh::as_stream(st));
}

// the same from an fp32 activation: one split pass into `scratch` (rows x K planes), or the fp32 kernels when the layer has no
// fp16 weights / an unaligned K -- dispatch as xmh/ops.py:gemm_nt does it
int linear_any(const xmh_linear& l, const float* A, int64_t lda, const float* residual, int64_t ldr, float* C, int64_t ldc, int64_t M,
               int act, int precision, const xmh::Planes& scratch, xmh_stream_t st) {
    const int64_t N = l.n, K = l.k;
    if (precision != kPrecExact && planes_layer(l) && scratch.hi && lda % 4 == 0 && reinterpret_cast<uintptr_t>(A) % 16 == 0) {
        xmh::Planes p{scratch.hi, precision == kPrecParity ? scratch.lo : nullptr, K};
        int rc = xmh::split_planes(A, lda, M, K, p, xmh::as_stream(st));
        if (rc) return rc;
        return linear_p(l, p, residual, ldr, C, ldc, nullptr, M, act, precision, st);
    }
    if (!l.w_f32) return xmh::fail(-22, "xmh forward: a %lld x %lld layer needs its fp32 weight for this shape / precision", (long long)N, (long long)K);
    return xmh_gemm_nt_f32(A, lda, l.w_f32, K, l.bias, residual, ldr, C, ldc, M, N, K, act, precision == kPrecFast ? 1 : 0, st);
}

// Which rows of the stack's output the caller keeps.  A tower that returns one embedding per sequence (cls / EOS, return_patches =
// False: models/CLIP/model.py:262-265, :392) needs ONE row of the last block's output per sequence; behind the last attention every
// operation is row-wise (out_proj + residual, ln_2, c_fc, QuickGELU, c_proj + residual), so the last block runs them on those B rows
// only -- the reference computes all B * L and discards them.  Same kernels, same per-element arithmetic: bit-identical rows.
struct TailRows {
    int mode = 0;                  // 0: every row; 1: row 0 of every group of L (cls); 2: row idx[b] of group b (EOS); 3: last row of every packed sequence
    const int32_t* idx = nullptr;
    float* x_tail = nullptr;       // [B, width] fp32: the kept rows of the stack's output (x itself is then stale in the last block)
};

// offs / M_packed: packed sequences (xmh_text_forward_packed) -- the row-wise kernels see M_packed rows, attention finds sequence b at
// rows [offs[b], offs[b + 1])
int run_blocks(const xmh_clip_block* blocks, int layers, int width, int heads, float* x, int64_t B, int L, int causal,
               const uint8_t* kpm, int precision, const BlockScratch& s, xmh_stream_t st, const int32_t* offs = nullptr, int64_t M_packed = 0,
               TailRows tail = TailRows{}, const int32_t* m_dev = nullptr) {
    // m_dev (xmh_text_forward_packed_dev): the packed row count lives in a device word; M_packed is then its upper bound (B * L), which
    // sizes the launches -- the row-wise kernels return on the rows behind the real count.  Parity / fast mode only.
    static const bool tail_off = xmh_experiment_env("XMH_TAIL_ROWS") && atoi(xmh_experiment_env("XMH_TAIL_ROWS")) == 0;      // A/B switch: 0 = the full last block + a gather
    const bool want_tail = tail.mode != 0;
    const bool tail_fused = want_tail && !tail_off && width % 2 == 0;
    // rows of a float-typed [*, cols] view (fp16 planes: two halves per float) -> the B kept rows
    auto keep_rows = [&](const void* src, int64_t ld_f, void* dst, int cols_f) -> int {
        if (tail.mode == 3) return xmh::gather_last_rows(static_cast<const float*>(src), ld_f, offs, static_cast<float*>(dst), B, cols_f, xmh::as_stream(st));
        return xmh_gather_rows(static_cast<const float*>(src), ld_f, tail.mode == 2 ? tail.idx : nullptr, 0, L, static_cast<float*>(dst), B, cols_f, st);
    };
    const int64_t M = offs ? M_packed : B * L;
    const int D = width;
    hipStream_t hs = xmh::as_stream(st);
    const xmh::Planes none{nullptr, nullptr, 0};
    for (int i = 0; i < layers; ++i) {
        const xmh_clip_block& b = blocks[i];
        if (b.qkv.n != 3 * D || b.qkv.k != D || b.out.n != D || b.out.k != D || b.fc.k != D || b.fc.n != 4 * D || b.proj.n != D || b.proj.k != b.fc.n)
            return xmh::fail(-22, "xmh forward: block %d has layer shapes that do not fit width %d", i, D);
        int rc;
        if (precision == kPrecExact) {
            if (m_dev) return xmh::fail(-95, "xmh forward: a device-side row count needs parity or fast mode");
            if (!b.qkv.w_f32 || !b.out.w_f32 || !b.fc.w_f32 || !b.proj.w_f32) return xmh::fail(-22, "xmh forward: block %d lacks fp32 weights (exact mode)", i);
            rc = xmh_layernorm_f32(x, D, b.ln1_w, b.ln1_b, kLnEps, s.h, D, M, D, st);
            if (rc) return rc;
            rc = xmh_gemm_nt_f32(s.h, D, b.qkv.w_f32, D, b.qkv.bias, nullptr, 0, s.qkv, 3 * D, M, 3 * D, D, kActNone, 0, st);
            if (rc) return rc;
            rc = offs ? xmh::attention_planes(s.qkv, B, L, heads, D / heads, causal, kpm, s.a, none, false, hs, offs)
                      : xmh_attention_f32(s.qkv, B, L, heads, D / heads, causal, kpm, s.a, st);
            if (rc) return rc;
            if (tail_fused && i == layers - 1) {             // the last block's row-wise half on the kept rows only
                float* xt = tail.x_tail;
                if ((rc = keep_rows(s.a, D, s.h, D))) return rc;
                if ((rc = keep_rows(x, D, xt, D))) return rc;
                if ((rc = xmh_gemm_nt_f32(s.h, D, b.out.w_f32, D, b.out.bias, xt, D, xt, D, B, D, D, kActNone, 0, st))) return rc;
                if ((rc = xmh_layernorm_f32(xt, D, b.ln2_w, b.ln2_b, kLnEps, s.a, D, B, D, st))) return rc;
                if ((rc = xmh_gemm_nt_f32(s.a, D, b.fc.w_f32, D, b.fc.bias, nullptr, 0, s.f, 4 * D, B, 4 * D, D, kActQuickGelu, 0, st))) return rc;
                if ((rc = xmh_gemm_nt_f32(s.f, 4 * D, b.proj.w_f32, 4 * D, b.proj.bias, xt, D, xt, D, B, D, 4 * D, kActNone, 0, st))) return rc;
                return 0;
            }
            rc = xmh_gemm_nt_f32(s.a, D, b.out.w_f32, D, b.out.bias, x, D, x, D, M, D, D, kActNone, 0, st);
            if (rc) return rc;
            rc = xmh_layernorm_f32(x, D, b.ln2_w, b.ln2_b, kLnEps, s.h, D, M, D, st);
            if (rc) return rc;
            rc = xmh_gemm_nt_f32(s.h, D, b.fc.w_f32, D, b.fc.bias, nullptr, 0, s.f, 4 * D, M, 4 * D, D, kActQuickGelu, 0, st);
            if (rc) return rc;
            rc = xmh_gemm_nt_f32(s.f, 4 * D, b.proj.w_f32, 4 * D, b.proj.bias, x, D, x, D, M, D, 4 * D, kActNone, 0, st);
            if (rc) return rc;
            continue;
        }
        if (!planes_layer(b.qkv) || !planes_layer(b.out) || !planes_layer(b.fc) || !planes_layer(b.proj))
            return xmh::fail(-22, "xmh forward: block %d lacks fp16 weights (w_hi) for width %d", i, D);
        rc = xmh::layernorm_planes(x, D, b.ln1_w, b.ln1_b, kLnEps, nullptr, 0, s.hP, M, D, hs, m_dev);
        if (rc) return rc;
        rc = linear_p(b.qkv, s.hP, nullptr, 0, s.qkv, 3 * D, nullptr, M, kActNone, precision, st, m_dev);
        if (rc) return rc;
        rc = xmh::attention_planes(s.qkv, B, L, heads, D / heads, causal, kpm, nullptr, s.aP, true, hs, offs);
        if (rc) return rc;
        if (tail_fused && i == layers - 1) {                 // the last block's row-wise half on the kept rows only
            float* xt = tail.x_tail;
            const xmh::Planes hc{s.hP.hi, s.hP.lo, D}, ac{s.aP.hi, s.aP.lo, D}, fc{s.fP.hi, s.fP.lo, 4 * (int64_t)D};
            if ((rc = keep_rows(s.aP.hi, s.aP.ld / 2, hc.hi, D / 2))) return rc;       // attention output planes -> the kept rows (hP is free)
            if (s.aP.lo && (rc = keep_rows(s.aP.lo, s.aP.ld / 2, hc.lo, D / 2))) return rc;
            if ((rc = keep_rows(x, D, xt, D))) return rc;
            if ((rc = linear_p(b.out, hc, xt, D, xt, D, nullptr, B, kActNone, precision, st))) return rc;
            if ((rc = xmh::layernorm_planes(xt, D, b.ln2_w, b.ln2_b, kLnEps, nullptr, 0, ac, B, D, hs))) return rc;
            if ((rc = linear_p(b.fc, ac, nullptr, 0, nullptr, 0, &fc, B, kActQuickGelu, precision, st))) return rc;
            if ((rc = linear_p(b.proj, fc, xt, D, xt, D, nullptr, B, kActNone, precision, st))) return rc;
            return 0;
        }
        rc = linear_p(b.out, s.aP, x, D, x, D, nullptr, M, kActNone, precision, st, m_dev);
        if (rc) return rc;
        rc = xmh::layernorm_planes(x, D, b.ln2_w, b.ln2_b, kLnEps, nullptr, 0, s.hP, M, D, hs, m_dev);
        if (rc) return rc;
        rc = linear_p(b.fc, s.hP, nullptr, 0, nullptr, 0, &s.fP, M, kActQuickGelu, precision, st, m_dev);
        if (rc) return rc;
        rc = linear_p(b.proj, s.fP, x, D, x, D, nullptr, M, kActNone, precision, st, m_dev);
        if (rc) return rc;
    }
    (void)none;
    if (want_tail) return keep_rows(x, D, tail.x_tail, D);    // no block ran its tail on the kept rows (switched off, or no layers): gather them
    return 0;
}

// ---- the block stack with saved activations ----------------------------------------------------------------
// What a backward pass of ResidualAttentionBlock (models/CLIP/model.py:167-197) needs from the forward, kept per layer instead
// of living in the shared scratch: one record of 16 * M * D floats per layer, fields in this order (xmh.h, xmh_clip_saved):
//   x_in [M,D] | ln1 [M,D] | qkv [M,3D] | attn [M,D] | x_mid [M,D] | ln2 [M,D] | fc_pre [M,4D] | fc_act [M,4D]
// Same kernels, same order and the same numbers as run_blocks: the producers write their fp32 result next to the operand planes
// (LayerNorm, attention), the GEMMs write straight into the record, the residual stream hops from record to record (x_in of
// layer i+1 is the output of layer i; the last layer writes x), and QuickGELU runs as its own elementwise pass between c_fc and
// c_proj (xmh::quickgelu_planes: the epilogue's function on the stored pre-activation) so that both sides of it are kept.
struct SavedRecord {
    float *x_in, *ln1, *qkv, *attn, *x_mid, *ln2, *fc_pre, *fc_act;
};

constexpr int kSavedFloatsPerElement = 16;           // per (token, channel): 1 + 1 + 3 + 1 + 1 + 1 + 4 + 4

SavedRecord saved_record(float* base, int layer, int64_t M, int D) {
    float* p = base + (size_t)layer * kSavedFloatsPerElement * (size_t)M * D;
    const size_t md = (size_t)M * D;
    SavedRecord r;
    r.x_in = p; r.ln1 = p + md; r.qkv = p + 2 * md; r.attn = p + 5 * md; r.x_mid = p + 6 * md; r.ln2 = p + 7 * md;
    r.fc_pre = p + 8 * md; r.fc_act = p + 12 * md;
    return r;
}

int run_blocks_saved(const xmh_clip_block* blocks, int layers, int width, int heads, float* x, int64_t B, int L, int causal,
                     const uint8_t* kpm, int precision, const BlockScratch& s, float* saved, xmh_stream_t st) {
    const int64_t M = B * L;
    const int D = width;
    hipStream_t hs = xmh::as_stream(st);
    const xmh::Planes none{nullptr, nullptr, 0};
    if (layers > 0) {
        hipError_t e = hipMemcpyAsync(saved_record(saved, 0, M, D).x_in, x, (size_t)M * D * sizeof(float), hipMemcpyDeviceToDevice, hs);
        if (e != hipSuccess) return xmh::fail(-5, "xmh_clip_blocks_forward_saved: copy of x failed: %s", hipGetErrorString(e));
    }
    for (int i = 0; i < layers; ++i) {
        const xmh_clip_block& b = blocks[i];
        if (b.qkv.n != 3 * D || b.qkv.k != D || b.out.n != D || b.out.k != D || b.fc.k != D || b.fc.n != 4 * D || b.proj.n != D || b.proj.k != b.fc.n)
            return xmh::fail(-22, "xmh forward: block %d has layer shapes that do not fit width %d", i, D);
        const SavedRecord r = saved_record(saved, i, M, D);
        float* x_out = i + 1 < layers ? saved_record(saved, i + 1, M, D).x_in : x;
        int rc;
        if (precision == kPrecExact) {
            if (!b.qkv.w_f32 || !b.out.w_f32 || !b.fc.w_f32 || !b.proj.w_f32) return xmh::fail(-22, "xmh forward: block %d lacks fp32 weights (exact mode)", i);
            rc = xmh_layernorm_f32(r.x_in, D, b.ln1_w, b.ln1_b, kLnEps, r.ln1, D, M, D, st);
            if (rc) return rc;
            rc = xmh_gemm_nt_f32(r.ln1, D, b.qkv.w_f32, D, b.qkv.bias, nullptr, 0, r.qkv, 3 * D, M, 3 * D, D, kActNone, 0, st);
            if (rc) return rc;
            rc = xmh_attention_f32(r.qkv, B, L, heads, D / heads, causal, kpm, r.attn, st);
            if (rc) return rc;
            rc = xmh_gemm_nt_f32(r.attn, D, b.out.w_f32, D, b.out.bias, r.x_in, D, r.x_mid, D, M, D, D, kActNone, 0, st);
            if (rc) return rc;
            rc = xmh_layernorm_f32(r.x_mid, D, b.ln2_w, b.ln2_b, kLnEps, r.ln2, D, M, D, st);
            if (rc) return rc;
            rc = xmh_gemm_nt_f32(r.ln2, D, b.fc.w_f32, D, b.fc.bias, nullptr, 0, r.fc_pre, 4 * D, M, 4 * D, D, kActNone, 0, st);
            if (rc) return rc;
            rc = xmh::quickgelu_planes(r.fc_pre, M, 4 * D, r.fc_act, none, hs);
            if (rc) return rc;
            rc = xmh_gemm_nt_f32(r.fc_act, 4 * D, b.proj.w_f32, 4 * D, b.proj.bias, r.x_mid, D, x_out, D, M, D, 4 * D, kActNone, 0, st);
            if (rc) return rc;
            continue;
        }
        if (!planes_layer(b.qkv) || !planes_layer(b.out) || !planes_layer(b.fc) || !planes_layer(b.proj))
            return xmh::fail(-22, "xmh forward: block %d lacks fp16 weights (w_hi) for width %d", i, D);
        rc = xmh::layernorm_planes(r.x_in, D, b.ln1_w, b.ln1_b, kLnEps, r.ln1, D, s.hP, M, D, hs);
        if (rc) return rc;
        rc = linear_p(b.qkv, s.hP, nullptr, 0, r.qkv, 3 * D, nullptr, M, kActNone, precision, st);
        if (rc) return rc;
        rc = xmh::attention_planes(r.qkv, B, L, heads, D / heads, causal, kpm, r.attn, s.aP, true, hs);
        if (rc) return rc;
        rc = linear_p(b.out, s.aP, r.x_in, D, r.x_mid, D, nullptr, M, kActNone, precision, st);
        if (rc) return rc;
        rc = xmh::layernorm_planes(r.x_mid, D, b.ln2_w, b.ln2_b, kLnEps, r.ln2, D, s.hP, M, D, hs);
        if (rc) return rc;
        rc = linear_p(b.fc, s.hP, nullptr, 0, r.fc_pre, 4 * D, nullptr, M, kActNone, precision, st);
        if (rc) return rc;
        rc = xmh::quickgelu_planes(r.fc_pre, M, 4 * D, r.fc_act, s.fP, hs);
        if (rc) return rc;
        rc = linear_p(b.proj, s.fP, r.x_mid, D, x_out, D, nullptr, M, kActNone, precision, st);
        if (rc) return rc;
    }
    return 0;
}

struct TowerScratch {
    BlockScratch blk;
    float *x, *cols, *patches, *row_a, *row_b, *y;
    xmh::Planes colsP;                               // parity / fast mode: im2col writes the conv1 GEMM's operand planes
    int32_t* eos;
    int32_t* offs;                                   // [B + 1] packed row offsets counted on the device (xmh_text_forward_packed_dev)
};

// conv_k > 0: image tower (im2col columns + patch embeddings); out_dim > 0: all tokens go through the final LN + projection
TowerScratch carve_tower(Arena& ar, int64_t B, int L, int width, int conv_k, int out_dim, int precision) {
    TowerScratch t{};
    const int64_t M = B * L;
    t.blk = carve_blocks(ar, M, width, precision);
    t.x = ar.take<float>((size_t)M * width);
    const bool cols_planes = conv_k > 0 && precision != kPrecExact && conv_k % 32 == 0;
    if (cols_planes) t.colsP = carve_planes(ar, (size_t)B * (L - 1), (size_t)conv_k, precision);
    t.cols = conv_k > 0 && !cols_planes ? ar.take<float>((size_t)B * (L - 1) * conv_k) : nullptr;
    t.patches = conv_k > 0 ? ar.take<float>((size_t)B * (L - 1) * width) : nullptr;
    t.row_a = ar.take<float>((size_t)B * width);
    t.row_b = ar.take<float>((size_t)B * width);
    t.y = out_dim > 0 ? ar.take<float>((size_t)M * width) : nullptr;
    t.eos = ar.take<int32_t>((size_t)B);
    t.offs = ar.take<int32_t>((size_t)B + 1);
    return t;
}

// LayerNorm + projection of `rows` rows (the tail of both towers): planes straight out of the LayerNorm when the layer allows
int ln_linear(const float* x, int64_t rows, int D, const float* gamma, const float* beta, const xmh_linear& l, float* ytmp, float* C,
              int precision, const BlockScratch& blk, xmh_stream_t st) {
    if (precision != kPrecExact && planes_layer(l) && blk.hP.hi) {
        int rc = xmh::layernorm_planes(x, D, gamma, beta, kLnEps, nullptr, 0, blk.hP, rows, D, xmh::as_stream(st));
        if (rc) return rc;
        return linear_p(l, blk.hP, nullptr, 0, C, l.n, nullptr, rows, kActNone, precision, st);
    }
    int rc = xmh_layernorm_f32(x, D, gamma, beta, kLnEps, ytmp, D, rows, D, st);
    if (rc) return rc;
    return linear_any(l, ytmp, D, nullptr, 0, C, l.n, rows, kActNone, precision, blk.any, st);
}

int check_precision(int precision) {
    if (precision != kPrecParity && precision != kPrecFast && precision != kPrecExact) return xmh::fail(-22, "xmh forward: precision must be 0 (parity), 1 (fast) or 2 (exact), got %d", precision);
    return 0;
}

}  // namespace

extern "C" size_t xmh_clip_workspace_bytes(int64_t B, int L, int width, int conv_k, int out_dim, int precision) {
    if (B <= 0 || L <= 0 || width <= 0) return 0;
    Arena ar(nullptr);
    carve_tower(ar, B, L, width, conv_k, out_dim, precision);
    return ar.used;
}

extern "C" int xmh_clip_blocks_forward(const xmh_clip_block* blocks, int layers, int width, int heads, float* x, int64_t B, int L,
                                       int causal, const uint8_t* key_padding_mask, int precision, void* workspace,
                                       size_t workspace_bytes, xmh_stream_t stream) {
    XMH_RANGE("xmh_clip_blocks_forward");
    if (int rc = check_precision(precision)) return rc;
    if (B == 0) return 0;
    if (!blocks || !x || !workspace || layers < 0 || heads <= 0 || width % heads) return xmh::fail(-22, "xmh_clip_blocks_forward: bad arguments");
    Arena ar(workspace);
    const BlockScratch s = carve_blocks(ar, B * L, width, precision);
    if (ar.used > workspace_bytes)
        return xmh::fail(-12, "xmh_clip_blocks_forward: workspace of %zu bytes, %zu needed", workspace_bytes, ar.used);
    return run_blocks(blocks, layers, width, heads, x, B, L, causal, key_padding_mask, precision, s, stream);
}

extern "C" size_t xmh_clip_saved_bytes(int64_t B, int L, int width, int layers) {
    if (B <= 0 || L <= 0 || width <= 0 || layers <= 0) return 0;
    return (size_t)layers * kSavedFloatsPerElement * (size_t)B * L * width * sizeof(float);
}

extern "C" int xmh_clip_blocks_forward_saved(const xmh_clip_block* blocks, int layers, int width, int heads, float* x, int64_t B, int L,
                                             int causal, const uint8_t* key_padding_mask, int precision, void* workspace,
                                             size_t workspace_bytes, float* saved, size_t saved_bytes, xmh_stream_t stream) {
    XMH_RANGE("xmh_clip_blocks_forward_saved");
    if (int rc = check_precision(precision)) return rc;
    if (B == 0) return 0;
    if (!blocks || !x || !workspace || !saved || layers < 0 || heads <= 0 || width % heads)
        return xmh::fail(-22, "xmh_clip_blocks_forward_saved: bad arguments");
    if (width % 4) return xmh::fail(-95, "xmh_clip_blocks_forward_saved: width %d is not a multiple of 4", width);
    const size_t need = xmh_clip_saved_bytes(B, L, width, layers);
    if (saved_bytes < need) return xmh::fail(-12, "xmh_clip_blocks_forward_saved: saved buffer of %zu bytes, %zu needed", saved_bytes, need);
    Arena ar(workspace);
    const BlockScratch s = carve_blocks(ar, B * L, width, precision);
    if (ar.used > workspace_bytes)
        return xmh::fail(-12, "xmh_clip_blocks_forward_saved: workspace of %zu bytes, %zu needed", workspace_bytes, ar.used);
    return run_blocks_saved(blocks, layers, width, heads, x, B, L, causal, key_padding_mask, precision, s, saved, stream);
}

extern "C" int xmh_vit_b32_forward(const xmh_vit_weights* w, const float* image, int64_t B, int precision, float* out_cls,
                                   float* out_tokens, void* workspace, size_t workspace_bytes, xmh_stream_t stream) {
    XMH_RANGE("xmh_vit_b32_forward (image tower)");
    if (int rc = check_precision(precision)) return rc;
    if (B == 0) return 0;
    if (!w || !image || !workspace || (!out_cls && !out_tokens)) return xmh::fail(-22, "xmh_vit_b32_forward: bad arguments");
    if (w->patch <= 0 || w->resolution % w->patch || w->heads <= 0 || w->width % w->heads)
        return xmh::fail(-22, "xmh_vit_b32_forward: resolution %d / patch %d / width %d / heads %d do not fit", w->resolution, w->patch, w->width, w->heads);
    const int G = w->resolution / w->patch, P = G * G, L = P + 1, D = w->width, conv_k = 3 * w->patch * w->patch;
    if (w->conv1.n != D || w->conv1.k != conv_k || w->proj.n != w->out_dim || w->proj.k != D)
        return xmh::fail(-22, "xmh_vit_b32_forward: conv1 / proj shapes do not fit the tower");
    Arena ar(workspace);
    const TowerScratch t = carve_tower(ar, B, L, D, conv_k, out_tokens ? w->out_dim : 0, precision);
    if (ar.used > workspace_bytes) return xmh::fail(-12, "xmh_vit_b32_forward: workspace of %zu bytes, %zu needed", workspace_bytes, ar.used);
    const int64_t M = B * L;
    int rc;
    if (t.colsP.hi && planes_layer(w->conv1)) {
        rc = xmh::im2col_planes(image, B, 3, w->resolution, w->patch, nullptr, t.colsP, xmh::as_stream(stream));
        if (rc) return rc;
        rc = linear_p(w->conv1, t.colsP, nullptr, 0, t.patches, D, nullptr, B * P, kActNone, precision, stream);
    } else {
        if (!t.cols) return xmh::fail(-22, "xmh_vit_b32_forward: conv1 lacks fp16 weights (w_hi) for a %d-wide patch row", conv_k);
        rc = xmh_im2col_patch(image, B, 3, w->resolution, w->patch, t.cols, stream);
        if (rc) return rc;
        rc = linear_any(w->conv1, t.cols, conv_k, nullptr, 0, t.patches, D, B * P, kActNone, precision, t.blk.any, stream);
    }
    if (rc) return rc;
    rc = xmh_vit_assemble(t.patches, w->cls, w->pos, w->ln_pre_w, w->ln_pre_b, kLnEps, t.x, B, P, D, stream);
    if (rc) return rc;
    TailRows tail;
    if (!out_tokens) { tail.mode = 1; tail.x_tail = t.row_a; }      // the cls row is all the caller keeps
    rc = run_blocks(w->blocks, w->layers, D, w->heads, t.x, B, L, 0, nullptr, precision, t.blk, stream, nullptr, 0, tail);
    if (rc) return rc;
    if (out_tokens) {                                 // return_patches: ln_post + proj on every token (model.py:257-265)
        rc = ln_linear(t.x, M, D, w->ln_post_w, w->ln_post_b, w->proj, t.y, out_tokens, precision, t.blk, stream);
        if (rc) return rc;
        if (out_cls) rc = xmh_gather_rows(out_tokens, w->out_dim, nullptr, 0, L, out_cls, B, w->out_dim, stream);
        return rc;
    }
    return ln_linear(t.row_a, B, D, w->ln_post_w, w->ln_post_b, w->proj, t.row_b, out_cls, precision, t.blk, stream);
}

extern "C" int xmh_text_forward(const xmh_text_weights* w, const int64_t* ids, const uint8_t* key_padding_mask, int64_t B, int L,
                                int precision, float* out_eos, float* out_tokens, int32_t* eos_index, void* workspace,
                                size_t workspace_bytes, xmh_stream_t stream) {
    XMH_RANGE("xmh_text_forward (text tower)");
    if (int rc = check_precision(precision)) return rc;
    if (B == 0) return 0;
    if (!w || !ids || !workspace || (!out_eos && !out_tokens)) return xmh::fail(-22, "xmh_text_forward: bad arguments");
    if (L <= 0 || L > w->context) return xmh::fail(-22, "xmh_text_forward: %d tokens, the positional embedding holds %d", L, w->context);
    const int D = w->width;
    if (w->heads <= 0 || D % w->heads || w->proj.n != w->out_dim || w->proj.k != D) return xmh::fail(-22, "xmh_text_forward: shapes do not fit the tower");
    Arena ar(workspace);
    const TowerScratch t = carve_tower(ar, B, L, D, 0, out_tokens ? w->out_dim : 0, precision);
    if (ar.used > workspace_bytes) return xmh::fail(-12, "xmh_text_forward: workspace of %zu bytes, %zu needed", workspace_bytes, ar.used);
    const int64_t M = B * L;
    int32_t* eos = eos_index ? eos_index : t.eos;
    int rc = xmh_text_embed(ids, w->tok_emb, w->pos, t.x, eos, B, L, D, w->vocab, stream);
    if (rc) return rc;
    TailRows tail;
    if (!out_tokens) { tail.mode = 2; tail.idx = eos; tail.x_tail = t.row_a; }      // only the EOS row of every caption is kept
    rc = run_blocks(w->blocks, w->layers, D, w->heads, t.x, B, L, 1, key_padding_mask, precision, t.blk, stream, nullptr, 0, tail);
    if (rc) return rc;
    if (out_tokens) {
        rc = ln_linear(t.x, M, D, w->ln_final_w, w->ln_final_b, w->proj, t.y, out_tokens, precision, t.blk, stream);
        if (rc) return rc;
        if (out_eos) rc = xmh_gather_rows(out_tokens, w->out_dim, eos, 0, L, out_eos, B, w->out_dim, stream);
        return rc;
    }
    return ln_linear(t.row_a, B, D, w->ln_final_w, w->ln_final_b, w->proj, t.row_b, out_eos, precision, t.blk, stream);
}

// CLIP.encode_text (models/CLIP/model.py:373-396) when only the EOS embedding is wanted, WITHOUT the padding: under the causal mask
// (build_attention_mask, :358-364) no token behind a caption's EOS can reach the row x[b, argmax(ids[b])] that :392 selects, so only
// the rows up to and including EOS are embedded, normalised, multiplied and attended -- sum_b (eos_b + 1) rows instead of B * L.
// Every kept row goes through the same kernels with the same per-element arithmetic as in xmh_text_forward (GEMM and LayerNorm are
// row-wise; in the attention kernel the keys behind a row are masked either way), so out_eos is bit-identical to the padded call.
extern "C" int xmh_text_forward_packed(const xmh_text_weights* w, const int64_t* ids, const int32_t* row_offsets, int64_t total_rows, int64_t B,
                                       int L, int precision, float* out_eos, void* workspace, size_t workspace_bytes, xmh_stream_t stream) {
    XMH_RANGE("xmh_text_forward_packed (text tower)");
    if (int rc = check_precision(precision)) return rc;
    if (B == 0) return 0;
    if (!w || !ids || !row_offsets || !workspace || !out_eos) return xmh::fail(-22, "xmh_text_forward_packed: bad arguments");
    if (L <= 0 || L > w->context || L > 64) return xmh::fail(-22, "xmh_text_forward_packed: %d tokens (positional embedding %d, packed attention 64)", L, w->context);
    if (total_rows < B || total_rows > B * L) return xmh::fail(-22, "xmh_text_forward_packed: %lld rows for %lld captions of at most %d tokens", (long long)total_rows, (long long)B, L);
    const int D = w->width;
    if (w->heads <= 0 || D % w->heads || w->proj.n != w->out_dim || w->proj.k != D) return xmh::fail(-22, "xmh_text_forward_packed: shapes do not fit the tower");
    Arena ar(workspace);
    const TowerScratch t = carve_tower(ar, B, L, D, 0, 0, precision);      // sized for B * L rows: total_rows <= that
    if (ar.used > workspace_bytes) return xmh::fail(-12, "xmh_text_forward_packed: workspace of %zu bytes, %zu needed", workspace_bytes, ar.used);
    hipStream_t hs = xmh::as_stream(stream);
    int rc = xmh::text_embed_packed(ids, w->tok_emb, w->pos, t.x, row_offsets, B, L, D, w->vocab, hs);
    if (rc) return rc;
    TailRows tail;
    tail.mode = 3;
    tail.x_tail = t.row_a;
    rc = run_blocks(w->blocks, w->layers, D, w->heads, t.x, B, L, 1, nullptr, precision, t.blk, stream, row_offsets, total_rows, tail);
    if (rc) return rc;
    return ln_linear(t.row_a, B, D, w->ln_final_w, w->ln_final_b, w->proj, t.row_b, out_eos, precision, t.blk, stream);
}

// The packed text forward with the caption lengths counted ON THE DEVICE (round 5): no host synchronisation, so the two towers' streams
// never wait on each other's host thread and the call can be captured in a hipGraph.  The lengths (xmh::caption_offsets) stay in the
// workspace; every row-wise launch is sized for B * L rows and returns on the rows behind offs[B] (GArgsP::m_dev, k_layernorm4).
//   * key_padding_mask (MITH: models/MITH/MITH.py:59-66 -> models/CLIP/model.py:378): applied to the keys as in the padded call; a caption
//     keeps its rows up to EOS or up to the last position the mask leaves visible, whichever is further back, so every row that any later
//     consumer may read unmasked is computed exactly as in xmh_text_forward;
//   * out_tokens ([B, L, out_dim], return_patches): the kept rows in the reference's padded layout, the dropped rows (all hidden by the
//     mask) ZERO where the padded call returns what attention made of padding -- callers that read masked rows must use xmh_text_forward.
// out_eos is bit-identical to xmh_text_forward's, and so is every kept row of out_tokens.  Parity / fast mode (exact mode: -ENOTSUP).
extern "C" int xmh_text_forward_packed_dev(const xmh_text_weights* w, const int64_t* ids, const uint8_t* key_padding_mask, int64_t B, int L,
                                           int precision, float* out_eos, float* out_tokens, void* workspace, size_t workspace_bytes,
                                           xmh_stream_t stream) {
    XMH_RANGE("xmh_text_forward_packed_dev (text tower)");
    if (int rc = check_precision(precision)) return rc;
    if (precision == kPrecExact) return xmh::fail(-95, "xmh_text_forward_packed_dev: parity or fast mode only");
    if (B == 0) return 0;
    if (!w || !ids || !workspace || (!out_eos && !out_tokens)) return xmh::fail(-22, "xmh_text_forward_packed_dev: bad arguments");
    if (L <= 0 || L > w->context || L > 64) return xmh::fail(-22, "xmh_text_forward_packed_dev: %d tokens (positional embedding %d, packed attention 64)", L, w->context);
    const int D = w->width;
    if (w->heads <= 0 || D % w->heads || w->proj.n != w->out_dim || w->proj.k != D) return xmh::fail(-22, "xmh_text_forward_packed_dev: shapes do not fit the tower");
    Arena ar(workspace);
    const TowerScratch t = carve_tower(ar, B, L, D, 0, out_tokens ? w->out_dim : 0, precision);
    if (ar.used > workspace_bytes) return xmh::fail(-12, "xmh_text_forward_packed_dev: workspace of %zu bytes, %zu needed", workspace_bytes, ar.used);
    hipStream_t hs = xmh::as_stream(stream);
    const int64_t Mub = B * L;
    const int32_t* m_dev = t.offs + B;
    int rc = xmh::caption_offsets(ids, key_padding_mask, B, L, t.offs, t.eos, hs);
    if (rc) return rc;
    rc = xmh::text_embed_packed(ids, w->tok_emb, w->pos, t.x, t.offs, B, L, D, w->vocab, hs);
    if (rc) return rc;
    if (!out_tokens) {
        TailRows tail;
        if (key_padding_mask) { tail.mode = 0; }                     // the kept rows may run past EOS: the EOS row is picked by index below
        else { tail.mode = 3; tail.x_tail = t.row_a; }
        rc = run_blocks(w->blocks, w->layers, D, w->heads, t.x, B, L, 1, key_padding_mask, precision, t.blk, stream, t.offs, Mub, tail, m_dev);
        if (rc) return rc;
        if (key_padding_mask) {
            rc = xmh::gather_packed_rows(t.x, D, t.offs, t.eos, t.row_a, B, D, hs);
            if (rc) return rc;
        }
        return ln_linear(t.row_a, B, D, w->ln_final_w, w->ln_final_b, w->proj, t.row_b, out_eos, precision, t.blk, stream);
    }
    rc = run_blocks(w->blocks, w->layers, D, w->heads, t.x, B, L, 1, key_padding_mask, precision, t.blk, stream, t.offs, Mub, TailRows{}, m_dev);
    if (rc) return rc;
    // ln_final + text_projection on the packed rows (t.y holds the projected rows, [rows, out_dim] inside its M * width floats), then the
    // reference's padded layout
    if (!planes_layer(w->proj) || !t.blk.hP.hi || w->out_dim > D) return xmh::fail(-95, "xmh_text_forward_packed_dev: the projection needs fp16 weights and out_dim <= width");
    rc = xmh::layernorm_planes(t.x, D, w->ln_final_w, w->ln_final_b, kLnEps, nullptr, 0, t.blk.hP, Mub, D, hs, m_dev);
    if (rc) return rc;
    rc = linear_p(w->proj, t.blk.hP, nullptr, 0, t.y, w->out_dim, nullptr, Mub, kActNone, precision, stream, m_dev);
    if (rc) return rc;
    rc = xmh::unpack_rows(t.y, w->out_dim, t.offs, out_tokens, B, L, w->out_dim, hs);
    if (rc) return rc;
    if (out_eos) rc = xmh::gather_packed_rows(t.y, w->out_dim, t.offs, t.eos, out_eos, B, w->out_dim, hs);
    return rc;
}

// ---- hash heads ---------------------------------------------------------------------------------------

namespace {
struct HeadScratch {
    float *a, *b, *wide, *wide2;
    xmh::Planes any;                                 // operand planes of the current GEMM's input (B x E)
};
void carve_head_arena(Arena& ar, int64_t B, int E, int precision, HeadScratch& s) {
    s.a = ar.take<float>((size_t)B * E);
    s.b = ar.take<float>((size_t)B * E);
    s.wide = ar.take<float>((size_t)B * E * 2);          // fc2 / fc output (2K <= 2E is checked) ...
    s.wide2 = ar.take<float>((size_t)B * E * 2);         // ... and the probabilities when the caller only wants bits
    s.any = precision == kPrecExact ? xmh::Planes{nullptr, nullptr, 0} : carve_planes(ar, (size_t)B, (size_t)E, precision);
}
int carve_head(void* workspace, size_t workspace_bytes, int64_t B, int E, int precision, HeadScratch& s, const char* who) {
    Arena ar(workspace);
    carve_head_arena(ar, B, E, precision, s);
    if (ar.used > workspace_bytes) return xmh::fail(-12, "%s: workspace of %zu bytes, %zu needed", who, workspace_bytes, ar.used);
    return 0;
}
}  // namespace

extern "C" size_t xmh_head_workspace_bytes(int64_t B, int E, int precision) {
    if (B <= 0 || E <= 0) return 0;
    Arena ar(nullptr);
    HeadScratch s;
    carve_head_arena(ar, B, E, precision, s);
    return ar.used;
}

extern "C" int xmh_head_dcmht(const xmh_dcmht_head* h, const float* emb, int64_t B, int precision, float* probs, uint32_t* bits,
                              const int64_t* row_index, void* workspace, size_t workspace_bytes, xmh_stream_t stream) {
    XMH_RANGE("xmh_head_dcmht");
    if (int rc = check_precision(precision)) return rc;
    if (B == 0) return 0;
    if (!h || !emb || !workspace || (!probs && !bits)) return xmh::fail(-22, "xmh_head_dcmht: bad arguments");
    const int E = (int)h->v_proj.k;
    const int64_t K2 = h->fc2.n;
    if (h->v_proj.n != E || h->out_proj.n != E || h->out_proj.k != E || h->fc2.k != E || K2 % 2 || K2 > 2 * E)
        return xmh::fail(-22, "xmh_head_dcmht: layer shapes do not fit (E = %d, fc2 %lld x %lld)", E, (long long)K2, (long long)h->fc2.k);
    HeadScratch s;
    if (int rc = carve_head(workspace, workspace_bytes, B, E, precision, s, "xmh_head_dcmht")) return rc;
    int rc = linear_any(h->v_proj, emb, E, nullptr, 0, s.a, E, B, kActNone, precision, s.any, stream);
    if (rc) return rc;
    rc = linear_any(h->out_proj, s.a, E, nullptr, 0, s.b, E, B, kActNone, precision, s.any, stream);
    if (rc) return rc;
    rc = h->norm_is_batchnorm ? xmh_affine_cols(s.b, h->bn_mean, h->bn_var, h->norm_w, h->norm_b, h->norm_eps, s.a, B, E, stream)
                              : xmh_layernorm_f32(s.b, E, h->norm_w, h->norm_b, h->norm_eps, s.a, E, B, E, stream);
    if (rc) return rc;
    rc = linear_any(h->fc2, s.a, E, nullptr, 0, s.wide, K2, B, kActRelu, precision, s.any, stream);
    if (rc) return rc;
    float* p = probs ? probs : s.wide2;
    rc = xmh_pair_softmax(s.wide, p, B, (int)(K2 / 2), stream);
    if (rc) return rc;
    if (bits) rc = xmh_pack_pair_argmax(p, B, (int)(K2 / 2), row_index, bits, stream);
    return rc;
}

extern "C" int xmh_head_dsph(const xmh_linear* fc, const float* emb, int64_t B, int precision, float* out, uint32_t* bits,
                             uint32_t* zero, int32_t* flags, const int64_t* row_index, void* workspace, size_t workspace_bytes,
                             xmh_stream_t stream) {
    XMH_RANGE("xmh_head_dsph");
    if (int rc = check_precision(precision)) return rc;
    if (B == 0) return 0;
    if (!fc || !emb || (!out && !bits)) return xmh::fail(-22, "xmh_head_dsph: bad arguments");
    const int E = (int)fc->k;
    const int64_t K = fc->n;
    if (K > 2 * E) return xmh::fail(-22, "xmh_head_dsph: %lld bits from %d features exceed the workspace layout", (long long)K, E);
    HeadScratch s{};
    if (!out || precision != kPrecExact) {
        if (!workspace) return xmh::fail(-22, "xmh_head_dsph: workspace needed");
        if (int rc = carve_head(workspace, workspace_bytes, B, E, precision, s, "xmh_head_dsph")) return rc;
    }
    float* o = out ? out : s.wide;
    int rc = linear_any(*fc, emb, E, nullptr, 0, o, K, B, kActTanh, precision, s.any, stream);
    if (rc) return rc;
    if (bits) rc = xmh_pack_sign(o, B, (int)K, row_index, bits, zero, flags, stream);
    return rc;
}

// ---- MITH head ---------------------------------------------------------------------------------------------------------

namespace {
struct MithScratch {
    float *y, *hbuf, *f, *scores, *m, *ycls, *hcls, *fcls;
    xmh::Planes hP, fP;                              // parity / fast mode: LayerNorm output and fc1 output as operand planes
    BlockScratch blk;
};
void carve_mith(Arena& ar, int64_t B, int L, int D, int K, int precision, MithScratch& s) {
    const int64_t M = B * L, M2 = B * K;
    s.y = ar.take<float>((size_t)M * D);
    s.scores = ar.take<float>((size_t)M * K);
    s.m = ar.take<float>((size_t)M2 * D);
    s.ycls = ar.take<float>((size_t)B * D);
    if (precision == kPrecExact) {
        s.hbuf = ar.take<float>((size_t)M * D);
        s.f = ar.take<float>((size_t)M * D * 4);
        s.hcls = ar.take<float>((size_t)B * D);
        s.fcls = ar.take<float>((size_t)B * D * 4);
        s.hP = s.fP = xmh::Planes{nullptr, nullptr, 0};
    } else {
        s.hbuf = s.f = s.hcls = s.fcls = nullptr;
        s.hP = carve_planes(ar, (size_t)M, (size_t)D, precision);
        s.fP = carve_planes(ar, (size_t)M, (size_t)D * 4, precision);
    }
    s.blk = carve_blocks(ar, M2, D, precision);
}
// GlobalConceptLearning on `rows` rows: y = ResidualMLPs(x) (x is not modified), scores = tanh(concept(y))
int mith_gcl(const xmh_mith_head* h, const float* x, int64_t rows, float* y, float* hb, float* f, float* scores, int precision,
             const MithScratch& s, xmh_stream_t st) {
    const int D = h->width;
    const float* cur = x;
    for (int i = 0; i < h->res_layers; ++i) {
        const xmh_mith_mlp& m = h->mlps[i];
        if (m.fc1.k != D || m.fc2.n != D || m.fc2.k != m.fc1.n || m.fc1.n > 4 * D) return xmh::fail(-22, "xmh_head_mith: MLP %d shapes do not fit width %d", i, D);
        int rc;
        if (precision != kPrecExact && planes_layer(m.fc1) && planes_layer(m.fc2)) {
            rc = xmh::layernorm_planes(cur, D, m.ln_w, m.ln_b, m.ln_eps, nullptr, 0, s.hP, rows, D, xmh::as_stream(st));
            if (rc) return rc;
            const xmh::Planes fo{s.fP.hi, s.fP.lo, m.fc1.n};
            rc = linear_p(m.fc1, s.hP, nullptr, 0, nullptr, 0, &fo, rows, kActGeluErf, precision, st);
            if (rc) return rc;
            rc = linear_p(m.fc2, fo, cur, D, y, D, nullptr, rows, kActNone, precision, st);      // y = cur + fc2(..): no clone of x needed
            if (rc) return rc;
        } else {
            if (!hb || !f) return xmh::fail(-22, "xmh_head_mith: MLP %d lacks fp16 weights (w_hi)", i);
            rc = xmh_layernorm_f32(cur, D, m.ln_w, m.ln_b, m.ln_eps, hb, D, rows, D, st);
            if (rc) return rc;
            rc = linear_any(m.fc1, hb, D, nullptr, 0, f, m.fc1.n, rows, kActGeluErf, precision, s.hP, st);
            if (rc) return rc;
            rc = linear_any(m.fc2, f, m.fc1.n, cur, D, y, D, rows, kActNone, precision, s.fP, st);
            if (rc) return rc;
        }
        cur = y;
    }
    return linear_any(h->concept, cur, D, nullptr, 0, scores, h->k_bits, rows, kActTanh, precision, s.hP, st);
}
}  // namespace

extern "C" size_t xmh_head_mith_workspace_bytes(int64_t B, int L, int width, int k_bits, int precision) {
    if (B <= 0 || L <= 0 || width <= 0 || k_bits <= 0) return 0;
    Arena ar(nullptr);
    MithScratch s;
    carve_mith(ar, B, L, width, k_bits, precision, s);
    return ar.used;
}

extern "C" int xmh_head_mith(const xmh_mith_head* h, const float* cls, const float* tokens, const uint8_t* token_mask, int64_t B, int L,
                             int precision, float* cls_hash, float* tokens_hash, void* workspace, size_t workspace_bytes,
                             xmh_stream_t stream) {
    XMH_RANGE("xmh_head_mith");
    if (int rc = check_precision(precision)) return rc;
    if (B == 0) return 0;
    if (!h || !cls || !tokens || !cls_hash || !tokens_hash || !workspace || L <= 0) return xmh::fail(-22, "xmh_head_mith: bad arguments");
    const int D = h->width, K = h->k_bits;
    if (D <= 0 || K <= 0 || h->heads <= 0 || D % h->heads || h->concept.n != K || h->concept.k != D || h->res_layers < 0 || h->layers < 0)
        return xmh::fail(-22, "xmh_head_mith: head shapes do not fit (width %d, %d bits, %d heads)", D, K, h->heads);
    Arena ar(workspace);
    MithScratch s;
    carve_mith(ar, B, L, D, K, precision, s);
    if (ar.used > workspace_bytes) return xmh::fail(-12, "xmh_head_mith: workspace of %zu bytes, %zu needed", workspace_bytes, ar.used);
    int rc = mith_gcl(h, cls, B, s.ycls, s.hcls, s.fcls, cls_hash, precision, s, stream);
    if (rc) return rc;
    rc = mith_gcl(h, tokens, B * L, s.y, s.hbuf, s.f, s.scores, precision, s, stream);
    if (rc) return rc;
    rc = xmh_lta_aggregate(s.scores, tokens, token_mask, h->pos_enc, s.m, B, L, K, D, h->top_k, stream);
    if (rc) return rc;
    rc = run_blocks(h->blocks, h->layers, D, h->heads, s.m, B, K, 0, nullptr, precision, s.blk, stream);
    if (rc) return rc;
    return xmh_bitwise_hash(s.m, h->hash_w, h->hash_b, nullptr, tokens_hash, B, K, D, stream);
}
