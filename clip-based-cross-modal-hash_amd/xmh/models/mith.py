"""MITH: CLIP in return_patches mode + the concept/token hash head (reference models/MITH/MITH.py:11-76,
models/MITH/hash/hash.py:9-254), registered as "MITH".  Parameters sit under the reference's key names
(``hash.gcl_i.mlp.mlps.0.0.weight`` ..., ``gcl_t`` aliasing ``gcl_i`` like in the reference, :218); the eval-path
dataflow is SURVEY 2.4.  ``res_*_cls`` / ``trans_tokens_*`` only feed the training losses and are returned as None.
"""
from __future__ import annotations

import ctypes
import math

import torch
import torch.nn as nn

from .. import _lib, ops, towers
from .._lib import check, current_stream, lib, ptr
from ..common.register import registry
from . import clip as _clip
from .base import BaseModel
from .clip import Transformer


class ResidualMLPs(nn.Module):
    def __init__(self, org_dim, dropout=0.0, num_layers=2, activation="gelu"):
        super().__init__()
        if activation != "gelu":
            raise NotImplementedError("MITH ships with activation: gelu")
        self.num_layers = num_layers
        self.mlps = nn.ModuleList(nn.Sequential(nn.Linear(org_dim, 4 * org_dim), nn.GELU(), nn.Dropout(p=dropout),
                                                nn.Linear(4 * org_dim, org_dim)) for _ in range(num_layers))
        self.lns = nn.ModuleList(nn.LayerNorm(org_dim) for _ in range(num_layers))

    def run(self, x):
        x = x.clone()
        for mlp, ln in zip(self.mlps, self.lns):
            h = ops.layernorm(x, ln.weight, ln.bias, ln.eps)
            f = ops.gemm_nt(h, mlp[0].weight, mlp[0].bias, act=ops.ACT_GELU_ERF)
            x = ops.gemm_nt(f, mlp[3].weight, mlp[3].bias, residual=x, out=x)
        return x


class GlobalConceptLearning(nn.Module):
    def __init__(self, k_concept, org_dim, dropout=0.0, activation="gelu", res_mlp_layers=2):
        super().__init__()
        self.mlp = ResidualMLPs(org_dim, dropout, res_mlp_layers, activation) if res_mlp_layers else nn.Identity()
        self.common_concept_embedding = nn.Linear(org_dim, k_concept, bias=False)

    def run(self, x):
        y = self.mlp.run(x) if isinstance(self.mlp, ResidualMLPs) else x
        return y, ops.gemm_nt(y, self.common_concept_embedding.weight, act=ops.ACT_TANH)


class PositionalEncoding(nn.Module):
    def __init__(self, d_model, dropout=0.0, max_len=128):
        super().__init__()
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0).transpose(0, 1) / (d_model ** 0.5))       # [max_len, 1, d_model]


class BitwiseHashing(nn.Module):
    def __init__(self, org_dim, k_bits=32):
        super().__init__()
        self.k = k_bits
        self.fc_list = nn.ModuleList(nn.Linear(org_dim, 1) for _ in range(k_bits))

    def stacked(self):
        return torch.cat([fc.weight for fc in self.fc_list], 0), torch.cat([fc.bias for fc in self.fc_list], 0)


class LocalizedTokenAggregation(nn.Module):
    def __init__(self, top_k):
        super().__init__()
        self.top_k = top_k


class LocalConceptTransforming(nn.Module):
    def __init__(self, clip_embed_dim, k_bits, transformer_layers, dropout, top_k):
        super().__init__()
        self.lta = LocalizedTokenAggregation(top_k=top_k)
        self.position = PositionalEncoding(clip_embed_dim, dropout=dropout, max_len=k_bits)
        self.transformer = Transformer(width=clip_embed_dim, layers=transformer_layers, heads=clip_embed_dim // 64)
        self.hashing = BitwiseHashing(org_dim=clip_embed_dim, k_bits=k_bits)

    def run(self, tokens, scores, token_mask, addend=None):
        """tokens [B,L,D] raw CLIP tokens, scores [B,L,K] -> tokens_hash [B,K] (+ addend)."""
        m = ops.lta_aggregate(scores, tokens, token_mask, self.position.pe, self.lta.top_k)     # [B,K,D], pos-enc added
        z = self.transformer.run(m)
        w, b = self.hashing.stacked()
        return ops.bitwise_hash(z, w, b, addend)


class MITHHashLayer(nn.Module):
    def __init__(self, clip_embed_dim=512, k_bits=16, dropout=0.0, transformer_layers=2, activation="gelu", top_k_label=8,
                 res_mlp_layers=2):
        super().__init__()
        self.k_bits = k_bits
        self.gcl_i = self.gcl_t = GlobalConceptLearning(k_bits, clip_embed_dim, dropout, activation, res_mlp_layers)
        self.lct_i = LocalConceptTransforming(clip_embed_dim, k_bits, transformer_layers, dropout, top_k_label)
        self.lct_t = LocalConceptTransforming(clip_embed_dim, k_bits, transformer_layers, dropout, top_k_label)
        self.img_concept_proj = nn.Linear(clip_embed_dim, clip_embed_dim)
        self.txt_concept_proj = nn.Linear(clip_embed_dim, clip_embed_dim)

    def _desc(self, gcl, lct, precision: int, keep: list):
        """xmh_mith_head of one modality (its GlobalConceptLearning + LocalConceptTransforming)."""
        D = lct.hashing.fc_list[0].weight.shape[1]
        layers = list(gcl.mlp.mlps) if isinstance(gcl.mlp, ResidualMLPs) else []
        mlps = (_lib.MithMlp * max(len(layers), 1))()
        for i, (mlp, ln) in enumerate(zip(layers, gcl.mlp.lns if layers else [])):
            lw, lb = ops._f32c(ln.weight.detach()), ops._f32c(ln.bias.detach())
            keep.extend((lw, lb))
            mlps[i] = _lib.MithMlp(lw.data_ptr(), lb.data_ptr(), float(ln.eps), _clip._linear_desc(mlp[0].weight, mlp[0].bias, precision, keep),
                                   _clip._linear_desc(mlp[3].weight, mlp[3].bias, precision, keep))
        blocks = _clip._blocks_desc(list(lct.transformer.resblocks), precision, keep)
        w, b = lct.hashing.stacked()
        w, b, pe = ops._f32c(w.detach()).contiguous(), ops._f32c(b.detach()).contiguous(), ops._f32c(lct.position.pe.detach()).contiguous()
        keep.extend((mlps, blocks, w, b, pe))
        heads = lct.transformer.resblocks[0].heads if len(lct.transformer.resblocks) else 1
        return _lib.MithHead(D, self.k_bits, lct.lta.top_k, len(layers), len(lct.transformer.resblocks), heads, mlps,
                             _clip._linear_desc(gcl.common_concept_embedding.weight, None, precision, keep), pe.data_ptr(), blocks,
                             w.data_ptr(), b.data_ptr())

    def _encode_native(self, gcl, lct, cls, tokens, mask):
        """xmh_head_mith: the whole eval dataflow of one modality from one C call."""
        cls, tokens = ops._f32c(cls).contiguous(), ops._f32c(tokens).contiguous()
        B, L, D = tokens.shape
        params = list(gcl.parameters()) + list(lct.parameters()) + list(lct.buffers())
        desc, precision = _clip._cached_desc(lct, lambda prec, keep: self._desc(gcl, lct, prec, keep), params=params, slot="mith")
        m = None if mask is None else mask.to(device=tokens.device, dtype=torch.uint8).contiguous()
        nbytes = lib.xmh_head_mith_workspace_bytes(B, L, D, self.k_bits, precision)
        ws = _clip._workspace(nbytes, tokens.device)
        cls_hash = torch.empty(B, self.k_bits, dtype=torch.float32, device=tokens.device)
        tokens_hash = torch.empty_like(cls_hash)
        check(lib.xmh_head_mith(ctypes.byref(desc), ptr(cls), ptr(tokens), ptr(m), B, L, precision, ptr(cls_hash), ptr(tokens_hash), ptr(ws),
                                nbytes, current_stream()), "xmh_head_mith")
        return None, cls_hash, tokens_hash, None

    @torch.no_grad()
    def _encode(self, gcl, lct, cls, tokens_lnd, mask):
        tokens = tokens_lnd.permute(1, 0, 2)                       # LND view of a [B,L,D] buffer -> back to [B,L,D]
        if _clip.NATIVE_FORWARD and cls.dim() == 2 and tokens.dim() == 3:
            return self._encode_native(gcl, lct, cls, tokens, mask)
        _, cls_hash = gcl.run(cls)
        _, scores = gcl.run(tokens)                                # concept scores of every token, [B,L,K]
        tokens_hash = lct.run(tokens, scores, mask)
        return None, cls_hash, tokens_hash, None

    def encode_img(self, img_cls, img_tokens):
        return self._encode(self.gcl_i, self.lct_i, img_cls, img_tokens, None)

    def encode_txt(self, txt_eos, txt_tokens, key_padding_mask):
        return self._encode(self.gcl_t, self.lct_t, txt_eos, txt_tokens, key_padding_mask)


@registry.register_model("MITH")
class MITH(BaseModel):
    def __init__(self, cfg, outputDim=16, clipPath="./ViT-B-32.pt", train_num=10000, hash_func="tanh", dropout=0,
                 transformer_layers=2, activation="gelu", top_k_label=8, res_mlp_layers=2, **hyper):
        super().__init__(cfg)
        embed_dim, _, self.backbone = self.load_backbone(clipPath=clipPath, return_patches=True)
        self.hash = MITHHashLayer(embed_dim, outputDim, dropout, transformer_layers, activation, top_k_label, res_mlp_layers)
        self.output_dim, self.hash_func = outputDim, hash_func
        self.hyper = hyper

    def encode_image(self, image):
        cls_token, seq_tokens, _ = self.backbone.encode_image(image)
        return self.hash.encode_img(img_cls=cls_token, img_tokens=seq_tokens)

    def encode_text(self, text, key_padding_mask=None):
        if key_padding_mask is not None and key_padding_mask.device != text.device:
            key_padding_mask = key_padding_mask.to(text.device)
        # every consumer of the tokens below applies new_mask (LocalizedTokenAggregation, models/MITH/hash/hash.py:142-148), so the rows
        # the mask hides need not be computed (xmh_text_forward_packed_dev; XMH_TEXT_PACKING=0 runs them)
        txt_eos, txt_tokens, _, new_mask = self.backbone.encode_text(text, key_padding_mask=key_padding_mask, masked_rows="zero")
        return self.hash.encode_txt(txt_eos, txt_tokens, new_mask)

    def forward(self, image, text, key_padding_mask=None, labels=None, indexs=None, return_loss=False):
        if return_loss:
            return self.object_function()
        img, txt = towers.run_both(lambda: self.encode_image(image), lambda: self.encode_text(text, key_padding_mask=key_padding_mask))
        return (*img, *txt)

    def object_function(self, *a, **k):
        raise NotImplementedError("training losses are outside the encode-and-retrieve path (SURVEY 2.1 #8)")

    @classmethod
    def from_config(cls, cfg, output_dim=16, train_num=10000):
        keys = ("hyper_tokens_intra", "hyper_distill", "hyper_info_nce", "hyper_cls_inter", "hyper_quan", "hyper_alpha", "hyper_lambda")
        return cls(cfg=cfg, outputDim=output_dim, clipPath=cfg.get("clip_path", "./ViT-B-32.pt"), train_num=train_num,
                   hash_func=cfg.get("hash_func", "tanh"), dropout=cfg.get("dropout", 0), transformer_layers=cfg.get("transformer_layers", 2),
                   activation=cfg.get("activation", "gelu"), top_k_label=cfg.get("top_k_label", 8), res_mlp_layers=cfg.get("res_mlp_layers", 2),
                   **{k: cfg.get(k) for k in keys if cfg.get(k) is not None})
