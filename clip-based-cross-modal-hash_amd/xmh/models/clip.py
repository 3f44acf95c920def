"""CLIP ViT-B/32 image and text towers executed by libxmh.so.

The ``nn.Module`` tree below exists to hold parameters under the reference's state_dict key names
(``visual.conv1.weight``, ``transformer.resblocks.3.attn.in_proj_weight`` ... -- SURVEY 8c), so reference
checkpoints load with ``load_state_dict``; none of its torch ``forward`` methods is ever called.  The forward
pass is the kernel chain of SURVEY 2.2:

  image: im2col -> GEMM(conv1) -> cls/pos/ln_pre -> 12 x [LN, GEMM qkv, attention, GEMM out + residual,
         LN, GEMM c_fc + QuickGELU, GEMM c_proj + residual] -> ln_post -> GEMM proj
         (reference models/CLIP/model.py:232-268, :167-197)
  text : embed + pos -> 12 blocks with the causal mask (+ key padding mask) -> ln_final -> GEMM
         text_projection -> EOS row (reference :373-396)

Differences, deliberate: activations are token-major [B, L, D] (the reference permutes to LND); attention weights
are not materialised (no in-scope caller consumes them); with ``return_patches=False`` ln_post / the projection run
on the cls / EOS row only (the reference projects every token and then discards all but one, :257-265)."""
from __future__ import annotations

import ctypes
import os
import weakref
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import _lib, ops
from .._lib import check, current_stream, lib, ptr

# The towers run through the whole-tower entry points of libxmh.so (xmh_vit_b32_forward / xmh_text_forward /
# xmh_clip_blocks_forward: one C call enqueues the ~150 launches).  XMH_FORWARD=ops keeps the per-primitive chain below
# (the same kernels in the same order, enqueued from Python) -- the tests run both and compare them bit for bit.
TEXT_PACKING = os.environ.get("XMH_TEXT_PACKING", "1") != "0"     # module switch (tests flip it): run the text tower without its padding rows
NATIVE_FORWARD = os.environ.get("XMH_FORWARD", "native") != "ops"



def _transposed(module, name: str) -> torch.Tensor:
    """`x @ proj` runs as gemm_nt(x, proj^T): keep ONE contiguous transpose per parameter version (a fresh copy per forward
    would defeat the fp16 weight cache of ops.gemm_nt and cost a host sync each time)."""
    p = getattr(module, name)
    key = (p.data_ptr(), p._version)
    cache = module.__dict__.setdefault("_xmh_transposed", {})
    hit = cache.get(name)
    if hit is None or hit[0] != key:
        hit = (key, p.detach().t().contiguous())
        cache[name] = hit
    return hit[1]


def _addr(t):
    return None if t is None else t.data_ptr()


def _linear_desc(W: torch.Tensor, bias, precision: int, keep: list) -> "_lib.Linear":
    """xmh_linear of a [N, K] weight; `keep` collects every tensor whose address goes into the descriptor."""
    W2 = ops._f32c(W.detach())
    W2 = W2 if W2.is_contiguous() else W2.contiguous()
    hi = lo = None
    if precision != ops.PREC_F32X and W2.shape[1] % 32 == 0:
        hi, lo = ops._half_weight(W2)
    b = None if bias is None else ops._f32c(bias.detach())
    keep.extend((W2, hi, lo, b))
    return _lib.Linear(_addr(W2), _addr(hi), _addr(lo), _addr(b), W2.shape[0], W2.shape[1])


def _blocks_desc(blocks, precision: int, keep: list):
    arr = (_lib.ClipBlock * len(blocks))()
    for i, blk in enumerate(blocks):
        ln = [ops._f32c(t.detach()) for t in (blk.ln_1.weight, blk.ln_1.bias, blk.ln_2.weight, blk.ln_2.bias)]
        keep.extend(ln)
        arr[i] = _lib.ClipBlock(_addr(ln[0]), _addr(ln[1]), _addr(ln[2]), _addr(ln[3]),
                                _linear_desc(blk.attn.in_proj_weight, blk.attn.in_proj_bias, precision, keep),
                                _linear_desc(blk.attn.out_proj.weight, blk.attn.out_proj.bias, precision, keep),
                                _linear_desc(blk.mlp.c_fc.weight, blk.mlp.c_fc.bias, precision, keep),
                                _linear_desc(blk.mlp.c_proj.weight, blk.mlp.c_proj.bias, precision, keep))
    return arr


_descriptors = weakref.WeakKeyDictionary()     # module -> {slot: (key, descriptor, tensors it points into)}; not in the module's
                                                # __dict__: ctypes structs holding pointers cannot be deep-copied / pickled


def _cached_desc(module, build, params=None, slot="tower"):
    """descriptor of `module`'s weights for the current precision, rebuilt when any parameter moved or changed in place."""
    precision = ops._precision
    key = (precision,) + tuple((p.data_ptr(), p._version) for p in (module.parameters() if params is None else params))
    slots = _descriptors.setdefault(module, {})
    hit = slots.get(slot)
    if hit is None or hit[0] != key:
        keep = []
        hit = (key, build(precision, keep), keep)
        slots[slot] = hit
    return hit[1], precision


def _workspace(nbytes: int, device) -> torch.Tensor:
    # per call, from torch's stream-aware caching allocator: the runner drives several forwards of ONE model on different
    # streams at a time, a workspace owned by the module would be shared between them
    return torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)


class _Block(nn.Module):
    def __init__(self, width: int, heads: int):
        super().__init__()
        self.attn = nn.MultiheadAttention(width, heads)
        self.ln_1 = nn.LayerNorm(width)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(width, width * 4)), ("gelu", nn.Identity()),
                                              ("c_proj", nn.Linear(width * 4, width))]))
        self.ln_2 = nn.LayerNorm(width)
        self.heads = heads

    def run(self, x: torch.Tensor, causal: bool, kpm) -> torch.Tensor:
        h = ops.layernorm(x, self.ln_1.weight, self.ln_1.bias)
        qkv = ops.gemm_nt(h, self.attn.in_proj_weight, self.attn.in_proj_bias)
        a = ops.attention(qkv, self.heads, causal=causal, key_padding_mask=kpm)
        x = ops.gemm_nt(a, self.attn.out_proj.weight, self.attn.out_proj.bias, residual=x, out=x)
        h = ops.layernorm(x, self.ln_2.weight, self.ln_2.bias)
        f = ops.gemm_nt(h, self.mlp.c_fc.weight, self.mlp.c_fc.bias, act=ops.ACT_QUICKGELU)
        return ops.gemm_nt(f, self.mlp.c_proj.weight, self.mlp.c_proj.bias, residual=x, out=x)


class Transformer(nn.Module):
    def __init__(self, width: int, layers: int, heads: int):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[_Block(width, heads) for _ in range(layers)])

    def run(self, x: torch.Tensor, causal: bool = False, key_padding_mask=None) -> torch.Tensor:
        """x [B, L, D] fp32 on the GPU; updated in place and returned."""
        if not NATIVE_FORWARD or not x.is_contiguous() or x.dtype != torch.float32:
            for blk in self.resblocks:
                x = blk.run(x, causal, key_padding_mask)
            return x
        B, L, D = x.shape
        blocks, precision = _cached_desc(self, lambda prec, keep: _blocks_desc(list(self.resblocks), prec, keep))
        kpm = None if key_padding_mask is None else key_padding_mask.to(device=x.device, dtype=torch.uint8).contiguous()
        nbytes = lib.xmh_clip_workspace_bytes(B, L, D, 0, 0, precision)
        ws = _workspace(nbytes, x.device)
        check(lib.xmh_clip_blocks_forward(blocks, len(self.resblocks), D, self.resblocks[0].heads if len(self.resblocks) else 1,
                                          ptr(x), B, L, int(causal), ptr(kpm), precision, ptr(ws), nbytes, current_stream()),
              "xmh_clip_blocks_forward")
        return x


    # per-layer record of xmh_clip_blocks_forward_saved (include/xmh.h): field -> width in units of D
    SAVED_FIELDS = (("x_in", 1), ("ln1", 1), ("qkv", 3), ("attn", 1), ("x_mid", 1), ("ln2", 1), ("fc_pre", 4), ("fc_act", 4))

    def run_saved(self, x: torch.Tensor, causal: bool = False, key_padding_mask=None):
        """The forward of `run` with the activations a backward pass of the blocks (models/CLIP/model.py:167-197) reads kept per
        layer: returns (x, saved) with x updated in place exactly as `run` leaves it and saved = one dict per layer
        of [B, L, n*D] fp32 views into one device buffer (16 * B*L*D floats per layer).  Native executor only."""
        if not x.is_cuda or not x.is_contiguous() or x.dtype != torch.float32 or x.dim() != 3:
            raise ValueError("run_saved takes a contiguous fp32 [B, L, D] tensor on the GPU")
        B, L, D = x.shape
        layers = len(self.resblocks)
        blocks, precision = _cached_desc(self, lambda prec, keep: _blocks_desc(list(self.resblocks), prec, keep))
        kpm = None if key_padding_mask is None else key_padding_mask.to(device=x.device, dtype=torch.uint8).contiguous()
        nbytes = lib.xmh_clip_workspace_bytes(B, L, D, 0, 0, precision)
        ws = _workspace(nbytes, x.device)
        sbytes = lib.xmh_clip_saved_bytes(B, L, D, layers)
        buf = torch.empty(sbytes // 4, dtype=torch.float32, device=x.device)
        check(lib.xmh_clip_blocks_forward_saved(blocks, layers, D, self.resblocks[0].heads if layers else 1, ptr(x), B, L, int(causal),
                                                ptr(kpm), precision, ptr(ws), nbytes, ptr(buf), sbytes, current_stream()),
              "xmh_clip_blocks_forward_saved")
        saved, per_layer, md = [], 16 * B * L * D, B * L * D
        for i in range(layers):
            rec, off = {}, i * per_layer
            for name, n in self.SAVED_FIELDS:
                rec[name] = buf[off:off + n * md].view(B, L, n * D)
                off += n * md
            saved.append(rec)
        return x, saved


class VisionTransformer(nn.Module):
    def __init__(self, input_resolution, patch_size, width, layers, heads, output_dim, return_patches=False):
        super().__init__()
        self.input_resolution, self.patch_size, self.output_dim = input_resolution, patch_size, output_dim
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        self.class_embedding = nn.Parameter(torch.zeros(width))
        self.positional_embedding = nn.Parameter(torch.zeros((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(torch.zeros(width, output_dim))
        self.return_patches = return_patches

    def _desc(self, precision: int, keep: list):
        width = self.conv1.weight.shape[0]
        blocks = _blocks_desc(list(self.transformer.resblocks), precision, keep)
        small = [ops._f32c(t.detach()).contiguous() for t in (self.class_embedding, self.positional_embedding, self.ln_pre.weight,
                                                              self.ln_pre.bias, self.ln_post.weight, self.ln_post.bias)]
        keep.extend(small)
        keep.append(blocks)
        heads = self.transformer.resblocks[0].heads if len(self.transformer.resblocks) else 1
        return _lib.VitWeights(self.input_resolution, self.patch_size, width, heads, len(self.transformer.resblocks), self.proj.shape[1],
                               _linear_desc(self.conv1.weight.reshape(width, -1), None, precision, keep),
                               *[_addr(t) for t in small],
                               _linear_desc(_transposed(self, "proj"), None, precision, keep), blocks)

    def run(self, image: torch.Tensor):
        if NATIVE_FORWARD:
            image = ops._f32c(image).contiguous()
            B, out_dim = image.shape[0], self.proj.shape[1]
            if image.shape[1:] != (3, self.input_resolution, self.input_resolution):
                raise ValueError("image batch is %s, the tower takes [B, 3, %d, %d]" % (tuple(image.shape), self.input_resolution, self.input_resolution))
            L = self.positional_embedding.shape[0]
            desc, precision = _cached_desc(self, self._desc)
            nbytes = lib.xmh_clip_workspace_bytes(B, L, desc.width, 3 * self.patch_size ** 2, out_dim if self.return_patches else 0, precision)
            ws = _workspace(nbytes, image.device)
            if not self.return_patches:
                out = torch.empty(B, out_dim, dtype=torch.float32, device=image.device)
                check(lib.xmh_vit_b32_forward(ctypes.byref(desc), ptr(image), B, precision, ptr(out), None, ptr(ws), nbytes, current_stream()),
                      "xmh_vit_b32_forward")
                return out
            y = torch.empty(B, L, out_dim, dtype=torch.float32, device=image.device)
            check(lib.xmh_vit_b32_forward(ctypes.byref(desc), ptr(image), B, precision, None, ptr(y), ptr(ws), nbytes, current_stream()),
                  "xmh_vit_b32_forward")
            return y[:, 0, :], y[:, 1:, :].permute(1, 0, 2), None                                # cls, tokens (LND), attn
        B = image.shape[0]
        width = self.conv1.weight.shape[0]
        n_patches = self.positional_embedding.shape[0] - 1
        cols = ops.im2col_patch(image, self.patch_size)
        patches = ops.gemm_nt(cols, self.conv1.weight.reshape(width, -1))
        x = ops.vit_assemble(patches, self.class_embedding, self.positional_embedding, self.ln_pre.weight, self.ln_pre.bias, B, n_patches)
        x = self.transformer.run(x)
        proj_t = _transposed(self, "proj")
        L = n_patches + 1
        if not self.return_patches:
            cls = ops.gather_rows(x, group=L, offset=0)
            cls = ops.layernorm(cls, self.ln_post.weight, self.ln_post.bias)
            return ops.gemm_nt(cls, proj_t)
        y = ops.gemm_nt(ops.layernorm(x, self.ln_post.weight, self.ln_post.bias), proj_t)      # [B, L, out]
        return y[:, 0, :], y[:, 1:, :].permute(1, 0, 2), None                                    # cls, tokens (LND), attn


class CLIP(nn.Module):
    def __init__(self, embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size, context_length,
                 vocab_size, transformer_width, transformer_heads, transformer_layers, return_patches=False):
        super().__init__()
        self.context_length, self.vocab_size, self.return_patches = context_length, vocab_size, return_patches
        self.visual = VisionTransformer(image_resolution, vision_patch_size, vision_width, vision_layers, vision_width // 64,
                                        embed_dim, return_patches=return_patches)
        self.transformer = Transformer(transformer_width, transformer_layers, transformer_heads)
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.zeros(context_length, transformer_width))
        self.ln_final = nn.LayerNorm(transformer_width)
        self.text_projection = nn.Parameter(torch.zeros(transformer_width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]))

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    @torch.no_grad()
    def encode_image(self, image):
        return self.visual.run(image)

    def _text_desc(self, precision: int, keep: list):
        blocks = _blocks_desc(list(self.transformer.resblocks), precision, keep)
        small = [ops._f32c(t.detach()).contiguous() for t in (self.token_embedding.weight, self.positional_embedding,
                                                              self.ln_final.weight, self.ln_final.bias)]
        keep.extend(small)
        keep.append(blocks)
        width = self.token_embedding.weight.shape[1]
        heads = self.transformer.resblocks[0].heads if len(self.transformer.resblocks) else 1
        return _lib.TextWeights(self.token_embedding.weight.shape[0], self.positional_embedding.shape[0], width, heads,
                                len(self.transformer.resblocks), self.text_projection.shape[1], *[_addr(t) for t in small],
                                _linear_desc(_transposed(self, "text_projection"), None, precision, keep), blocks)

    def _text_params(self):
        yield from self.transformer.parameters()
        yield from (self.token_embedding.weight, self.positional_embedding, self.ln_final.weight, self.ln_final.bias, self.text_projection)

    def _encode_text_native(self, text, key_padding_mask, masked_rows="exact"):
        if not text.is_cuda:
            raise RuntimeError("xmh ops need CUDA/HIP tensors; there is no CPU fallback")
        ids = text.to(torch.int64).contiguous()
        B, L = ids.shape
        desc, precision = _cached_desc(self, self._text_desc, params=self._text_params(), slot="text")
        out_dim = self.text_projection.shape[1]
        kpm = None if key_padding_mask is None else key_padding_mask.to(device=ids.device, dtype=torch.uint8).contiguous()
        nbytes = lib.xmh_clip_workspace_bytes(B, L, desc.width, 0, out_dim if self.return_patches else 0, precision)
        ws = _workspace(nbytes, ids.device)
        eos_tok = torch.empty(B, out_dim, dtype=torch.float32, device=ids.device)
        packable = L <= 64 and TEXT_PACKING and precision != 2 and out_dim <= desc.width
        if packable and not self.return_patches:
            # only the EOS embedding is wanted and the attention is causal: the tokens behind a caption's EOS cannot reach it, so the
            # tower runs on the rows up to EOS only (bit-identical output, sum(lengths) / (B L) of the work).  Round 5: the lengths are
            # counted on the device and every launch is sized by its upper bound (xmh_text_forward_packed_dev) -- round 4 read the row
            # count back with .item(), a host sync per caption batch that serialised the two towers' streams.
            check(lib.xmh_text_forward_packed_dev(ctypes.byref(desc), ptr(ids), ptr(kpm), B, L, precision, ptr(eos_tok), None, ptr(ws), nbytes,
                                                  current_stream()), "xmh_text_forward_packed_dev")
            return eos_tok
        if not self.return_patches and kpm is None and L <= 64 and TEXT_PACKING and precision == 2:
            # exact mode (fp32 MFMA; diagnostics): round 4's form of the same packing, the row count read back by the host
            offs = torch.zeros(B + 1, dtype=torch.int32, device=ids.device)
            offs[1:] = torch.cumsum(ids.argmax(dim=1) + 1, 0)
            total = int(offs[B].item())
            if total < 0.9 * B * L:
                check(lib.xmh_text_forward_packed(ctypes.byref(desc), ptr(ids), ptr(offs), total, B, L, precision, ptr(eos_tok), ptr(ws), nbytes,
                                                  current_stream()), "xmh_text_forward_packed")
                return eos_tok
        if not self.return_patches:
            check(lib.xmh_text_forward(ctypes.byref(desc), ptr(ids), ptr(kpm), B, L, precision, ptr(eos_tok), None, None, ptr(ws), nbytes,
                                       current_stream()), "xmh_text_forward")
            return eos_tok
        y = torch.empty(B, L, out_dim, dtype=torch.float32, device=ids.device)
        new_mask = None if kpm is None else (kpm.bool() | (ids == self.vocab_size - 1))
        if packable and masked_rows == "zero" and kpm is not None:
            # the caller reads no row the mask hides (MITH: every consumer of the tokens applies new_mask, models/MITH/hash/hash.py:142-148):
            # the rows behind the last visible position are not computed and come back as zeros; every other row is the padded call's
            check(lib.xmh_text_forward_packed_dev(ctypes.byref(desc), ptr(ids), ptr(kpm), B, L, precision, ptr(eos_tok), ptr(y), ptr(ws), nbytes,
                                                  current_stream()), "xmh_text_forward_packed_dev")
            return eos_tok, y.permute(1, 0, 2), None, new_mask
        check(lib.xmh_text_forward(ctypes.byref(desc), ptr(ids), ptr(kpm), B, L, precision, ptr(eos_tok), ptr(y), None, ptr(ws), nbytes,
                                   current_stream()), "xmh_text_forward")
        return eos_tok, y.permute(1, 0, 2), None, new_mask

    @torch.no_grad()
    def encode_text(self, text, key_padding_mask=None, masked_rows="exact"):
        """reference models/CLIP/model.py:373-396.  ``masked_rows`` (this package's callers only): "exact" returns every token row as the
        reference computes it; "zero" lets the rows that ``key_padding_mask`` hides behind the last visible position come back as zeros
        (not computed) -- for callers that apply the returned mask to everything they read, like the MITH head."""
        if NATIVE_FORWARD:
            return self._encode_text_native(text, key_padding_mask, masked_rows)
        x, eos = ops.text_embed(text, self.token_embedding.weight, self.positional_embedding)
        B, L, _ = x.shape
        kpm = None if key_padding_mask is None else key_padding_mask.to(x.device)
        x = self.transformer.run(x, causal=True, key_padding_mask=kpm)
        proj_t = _transposed(self, "text_projection")
        if not self.return_patches:
            e = ops.gather_rows(x, group=L, idx=eos)
            return ops.gemm_nt(ops.layernorm(e, self.ln_final.weight, self.ln_final.bias), proj_t)
        y = ops.gemm_nt(ops.layernorm(x, self.ln_final.weight, self.ln_final.bias), proj_t)        # [B, L, out]
        eos_tok = ops.gather_rows(y, group=L, idx=eos)
        new_mask = None if kpm is None else (kpm.bool() | (text.to(x.device) == self.vocab_size - 1))
        return eos_tok, y.permute(1, 0, 2), None, new_mask


def round_like_convert_weights(model: CLIP) -> None:
    """The reference casts Conv/Linear/MultiheadAttention/proj tensors to fp16 before loading the checkpoint
    (convert_weights, models/CLIP/model.py:415-436) and the runner later calls .float(): those tensors end up as
    fp16-rounded fp32.  Same rounding here, applied after load."""
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                m.weight.copy_(m.weight.half().float())
                if m.bias is not None:
                    m.bias.copy_(m.bias.half().float())
            if isinstance(m, nn.MultiheadAttention):
                m.in_proj_weight.copy_(m.in_proj_weight.half().float())
                m.in_proj_bias.copy_(m.in_proj_bias.half().float())
        for p in (model.visual.proj, model.text_projection):
            p.copy_(p.half().float())


def build_model(state_dict: dict, return_patches: bool = False) -> CLIP:
    """Infer the architecture from tensor shapes like the reference's build_model (models/CLIP/model.py:438-489);
    ViT checkpoints only (the ResNet branch is unused by every config)."""
    if "visual.proj" not in state_dict:
        raise NotImplementedError("only ViT CLIP checkpoints are supported (no config of the reference uses RN50)")
    vw = state_dict["visual.conv1.weight"].shape[0]
    v_layers = len([k for k in state_dict if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    patch = state_dict["visual.conv1.weight"].shape[-1]
    grid = round((state_dict["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    tw = state_dict["ln_final.weight"].shape[0]
    t_layers = len(set(k.split(".")[2] for k in state_dict if k.startswith("transformer.resblocks")))
    model = CLIP(state_dict["text_projection"].shape[1], patch * grid, v_layers, vw, patch, state_dict["positional_embedding"].shape[0],
                 state_dict["token_embedding.weight"].shape[0], tw, tw // 64, t_layers, return_patches=return_patches)
    sd = {k: v for k, v in state_dict.items() if k not in ("input_resolution", "context_length", "vocab_size")}
    model.load_state_dict(sd)
    round_like_convert_weights(model)
    return model.float().eval()
