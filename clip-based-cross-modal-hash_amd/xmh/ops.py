"""Host wrappers of the encoder primitives in libxmh.so (see include/xmh.h).  They allocate outputs with
torch and pass raw device pointers + the current HIP stream; no arithmetic happens in Python."""
from __future__ import annotations

from typing import Optional

import torch

from ._lib import check, current_stream, lib, ptr

ACT_NONE, ACT_QUICKGELU, ACT_GELU_ERF, ACT_TANH, ACT_RELU = range(5)
PREC_F32, PREC_F16, PREC_F32X = 0, 1, 2

_precision = PREC_F32
_NAMES = {"f32": PREC_F32, "f16": PREC_F16, "f32x": PREC_F32X}


def set_precision(name: str) -> None:
    """'f32'  parity mode (default): fp32-grade results on the fp16 MFMA (xmh_gemm_nt_split16): the fp32 activations are split
              hi/lo; fp16-exact weights (CLIP weights as released) take two MFMAs per product, any other weight is split
              too and takes three.  Product error 2^-22.  Unaligned shapes fall through to the exact kernel.
       'f32x' exact fp32 MFMA everywhere (v_mfma_f32_32x32x2_f32).
       'f16'  fast mode: fp16 operands / fp32 accumulate."""
    global _precision
    _precision = _NAMES[name]


def get_precision() -> str:
    return {v: k for k, v in _NAMES.items()}[_precision]


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("xmh ops need CUDA/HIP tensors (got %s); there is no CPU fallback" % t.device)
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.stride(-1) == 1 else t.contiguous()


def _rows(t: torch.Tensor) -> torch.Tensor:
    """view as [rows, D] with a single row stride (what the kernels take as leading dimension)."""
    t = _f32c(t)
    t2 = t.reshape(-1, t.shape[-1])
    return t2 if t2.stride(-1) == 1 else t2.contiguous()


_half_weights = {}          # (data_ptr, version, shape, stride) -> (fp16 copy, exact, the fp32 tensor itself)


def clear_weight_cache() -> None:
    """drop the cached fp16 copies of weights (they pin their fp32 originals, see _half_weight)."""
    _half_weights.clear()


def cast_f16(x: torch.Tensor) -> torch.Tensor:
    """fp32 [rows, D] (D % 8 == 0, contiguous) -> fp16 through the HIP cast kernel."""
    x = _f32c(x).contiguous()
    y = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    check(lib.xmh_cast_f32_to_f16(ptr(x), ptr(y), x.numel(), current_stream()), "xmh_cast_f32_to_f16")
    return y


def _half_weight(W: torch.Tensor):
    """(hi, lo) fp16 parts of a weight: ``hi = half(W)``; ``lo`` is None when every value survives the fp32 -> fp16 -> fp32
    round trip (CLIP weights straight from convert_weights), else ``half(W - hi)`` (anything fine-tuned in fp32).
    The entry keeps a reference to W: while it is cached its memory cannot be handed to another tensor, so the
    address in the key cannot go stale; in-place updates bump ``_version``."""
    key = (W.data_ptr(), W._version, tuple(W.shape), tuple(W.stride()))
    e = _half_weights.get(key)
    if e is None:
        if len(_half_weights) > 512:
            _half_weights.clear()
        Wd = W.detach()
        h = cast_f16(Wd)
        rest = Wd - h.float()
        lo = None if not bool(rest.any()) else cast_f16(rest)     # one-time check per weight (host sync here only)
        e = (h, lo, W)
        _half_weights[key] = e
    return e[0], e[1]


def gemm_nt(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
            act: int = ACT_NONE, out: Optional[torch.Tensor] = None, precision: Optional[int] = None) -> torch.Tensor:
    """act(A @ W^T + bias) (+ residual); A [..., K], W [N, K] (nn.Linear layout) -> [..., N]."""
    lead = A.shape[:-1]
    A2, W2 = _rows(A), _rows(W)
    M, K = A2.shape
    N = W2.shape[0]
    if W2.shape[1] != K:
        raise ValueError("gemm_nt: A is [*, %d] but W is [%d, %d]" % (K, N, W2.shape[1]))
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=A2.device)
    out2 = out.reshape(-1, N)
    res2 = None if residual is None else _rows(residual)
    b = None if bias is None else _f32c(bias)
    prec = _precision if precision is None else precision
    ldr = 0 if res2 is None else res2.stride(0)
    if prec == PREC_F16 and K % 32 == 0 and W2.is_contiguous() and M * K % 8 == 0:
        # fast mode proper: fp16 operands in memory (weights converted once, activations by one cast pass), fp32 accumulate
        Ah = cast_f16(A2) if A2.is_contiguous() else cast_f16(A2.contiguous())
        Wh, _ = _half_weight(W2)
        check(lib.xmh_gemm_nt_h16(ptr(Ah), K, ptr(Wh), K, ptr(b), ptr(res2), ldr, ptr(out2),
                                  out2.stride(0), M, N, K, act, current_stream()), "xmh_gemm_nt_h16")
        return out2.reshape(*lead, N)
    if prec == PREC_F32 and K % 32 == 0 and W2.is_contiguous() and A2.stride(0) % 4 == 0 and A2.data_ptr() % 16 == 0:
        # parity mode: hi/lo split of the activations on the fp16 MFMA; the weight goes in as one (fp16-exact) or two fp16 parts
        Wh, Wl = _half_weight(W2)
        check(lib.xmh_gemm_nt_split16(ptr(A2), A2.stride(0), ptr(Wh), ptr(Wl), K, ptr(b), ptr(res2), ldr, ptr(out2),
                                      out2.stride(0), M, N, K, act, current_stream()), "xmh_gemm_nt_split16")
        return out2.reshape(*lead, N)
    check(lib.xmh_gemm_nt_f32(ptr(A2), A2.stride(0), ptr(W2), W2.stride(0), ptr(b), ptr(res2), ldr,
                              ptr(out2), out2.stride(0), M, N, K, act, PREC_F16 if prec == PREC_F16 else PREC_F32,
                              current_stream()), "xmh_gemm_nt_f32")
    return out2.reshape(*lead, N)


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    x2, gm, bt = _rows(x), _f32c(gamma), _f32c(beta)       # locals keep any converted temporaries alive past the launch
    y = torch.empty_like(x2)
    check(lib.xmh_layernorm_f32(ptr(x2), x2.stride(0), ptr(gm), ptr(bt), eps, ptr(y), y.stride(0), x2.shape[0],
                                x2.shape[1], current_stream()), "xmh_layernorm_f32")
    return y.reshape(x.shape)


def attention(qkv: torch.Tensor, heads: int, causal: bool = False, key_padding_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """qkv [B, L, 3*D] -> [B, L, D]; key_padding_mask [B, L] bool/uint8 (True = ignore that key)."""
    qkv = _f32c(qkv).contiguous()
    B, L, D3 = qkv.shape
    D = D3 // 3
    kpm = None
    if key_padding_mask is not None:
        kpm = key_padding_mask.to(device=qkv.device, dtype=torch.uint8).contiguous()
    out = torch.empty(B, L, D, dtype=torch.float32, device=qkv.device)
    fn = lib.xmh_attention_f32 if _precision == PREC_F32X else lib.xmh_attention_split16      # exact mode keeps fp32 products
    check(fn(ptr(qkv), B, L, heads, D // heads, int(causal), ptr(kpm), ptr(out), current_stream()), "xmh_attention")
    return out


def im2col_patch(image: torch.Tensor, patch: int) -> torch.Tensor:
    image = _f32c(image).contiguous()
    B, Cin, H, Wd = image.shape
    if H != Wd:
        raise ValueError("square images only")
    G = H // patch
    cols = torch.empty(B * G * G, Cin * patch * patch, dtype=torch.float32, device=image.device)
    check(lib.xmh_im2col_patch(ptr(image), B, Cin, H, patch, ptr(cols), current_stream()), "xmh_im2col_patch")
    return cols


def vit_assemble(patch_out, cls, pos, gamma, beta, B: int, n_patches: int, eps: float = 1e-5) -> torch.Tensor:
    D = patch_out.shape[-1]
    x = torch.empty(B, n_patches + 1, D, dtype=torch.float32, device=patch_out.device)
    po, c, p, gm, bt = _rows(patch_out), _f32c(cls), _f32c(pos).contiguous(), _f32c(gamma), _f32c(beta)
    check(lib.xmh_vit_assemble(ptr(po), ptr(c), ptr(p), ptr(gm), ptr(bt), eps, ptr(x), B, n_patches, D, current_stream()), "xmh_vit_assemble")
    return x


def text_embed(ids: torch.Tensor, tok_emb: torch.Tensor, pos: torch.Tensor):
    if not ids.is_cuda:
        raise RuntimeError("xmh ops need CUDA/HIP tensors; there is no CPU fallback")
    ids = ids.to(torch.int64).contiguous()
    B, L = ids.shape
    tok, pos = _f32c(tok_emb).contiguous(), _f32c(pos).contiguous()
    D = tok.shape[1]
    x = torch.empty(B, L, D, dtype=torch.float32, device=ids.device)
    eos = torch.empty(B, dtype=torch.int32, device=ids.device)
    check(lib.xmh_text_embed(ptr(ids), ptr(tok), ptr(pos), ptr(x), ptr(eos), B, L, D, tok.shape[0], current_stream()), "xmh_text_embed")
    return x, eos


def gather_rows(x: torch.Tensor, group: int, idx: Optional[torch.Tensor] = None, offset: int = 0) -> torch.Tensor:
    """x [R*group, D] -> [R, D]: row r*group + (idx[r] | offset)."""
    x2 = _rows(x)
    rows = x2.shape[0] // group
    out = torch.empty(rows, x2.shape[1], dtype=torch.float32, device=x2.device)
    check(lib.xmh_gather_rows(ptr(x2), x2.stride(0), ptr(idx), offset, group, ptr(out), rows, x2.shape[1], current_stream()), "xmh_gather_rows")
    return out


def affine_cols(x, mean, var, gamma, beta, eps: float = 1e-5) -> torch.Tensor:
    x2 = _rows(x).contiguous()
    y = torch.empty_like(x2)
    mn, vr, gm, bt = _f32c(mean), _f32c(var), _f32c(gamma), _f32c(beta)
    check(lib.xmh_affine_cols(ptr(x2), ptr(mn), ptr(vr), ptr(gm), ptr(bt), eps, ptr(y), x2.shape[0], x2.shape[1], current_stream()),
          "xmh_affine_cols")
    return y


def pair_softmax(x: torch.Tensor) -> torch.Tensor:
    x2 = _rows(x).contiguous()
    y = torch.empty_like(x2)
    check(lib.xmh_pair_softmax(ptr(x2), ptr(y), x2.shape[0], x2.shape[1] // 2, current_stream()), "xmh_pair_softmax")
    return y


def lta_aggregate(scores: torch.Tensor, tokens: torch.Tensor, token_mask: Optional[torch.Tensor], pos_enc: Optional[torch.Tensor],
                  top_k: int = 8) -> torch.Tensor:
    """scores [B, L, K], tokens [B, L, D] -> [B, K, D] (MITH LocalizedTokenAggregation + positional encoding)."""
    scores, tokens = _f32c(scores).contiguous(), _f32c(tokens).contiguous()
    B, L, K = scores.shape
    D = tokens.shape[-1]
    m = None if token_mask is None else token_mask.to(device=scores.device, dtype=torch.uint8).contiguous()
    pe = None if pos_enc is None else _f32c(pos_enc).reshape(-1, D)[:K].contiguous()
    out = torch.empty(B, K, D, dtype=torch.float32, device=scores.device)
    check(lib.xmh_lta_aggregate(ptr(scores), ptr(tokens), ptr(m), ptr(pe), ptr(out), B, L, K, D, top_k, current_stream()), "xmh_lta_aggregate")
    return out


def bitwise_hash(z: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, addend: Optional[torch.Tensor] = None) -> torch.Tensor:
    """z [B, K, D], w [K, D], bias [K] -> tanh(w_k . z[b,k] + bias_k) (+ addend) [B, K]."""
    z = _f32c(z).contiguous()
    B, K, D = z.shape
    out = torch.empty(B, K, dtype=torch.float32, device=z.device)
    a = None if addend is None else _f32c(addend).contiguous()
    w2, b2 = _f32c(w).contiguous(), _f32c(bias).contiguous()
    check(lib.xmh_bitwise_hash(ptr(z), ptr(w2), ptr(b2), ptr(a), ptr(out), B, K, D, current_stream()), "xmh_bitwise_hash")
    return out
