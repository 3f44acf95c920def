"""Hash heads of the in-scope methods, parameters under the reference's key names, forward in libxmh.so.

  DCMHT  models/DCMHT/hash/hash.py:15-82   MHA on a length-1 sequence == out_proj(v_proj(x)) (softmax over one key
         is 1), BatchNorm1d (image) / LayerNorm (text), fc2, relu, pair softmax
  DSPH   models/DSPH/hash/hash.py:6-45     tanh(fc(x)) (dropout is identity in eval)
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn as nn

from .. import _lib, ops
from .._lib import check, current_stream, lib, ptr
from . import clip as _clip


class DCMHTModalityHash(nn.Module):
    def __init__(self, inputDim=512, outputDim=64, num_heads=8, layernorm=True):
        super().__init__()
        self.bit = outputDim
        self.atten = nn.MultiheadAttention(inputDim, num_heads=num_heads, batch_first=True)
        self.norm = nn.LayerNorm(inputDim) if layernorm else nn.BatchNorm1d(inputDim)
        self.fc2 = nn.Linear(inputDim, outputDim * 2)

    def _desc(self, precision: int, keep: list):
        E = self.atten.in_proj_weight.shape[1]
        bn = isinstance(self.norm, nn.BatchNorm1d)
        small = [ops._f32c(t.detach()).contiguous() for t in ((self.norm.weight, self.norm.bias, self.norm.running_mean, self.norm.running_var)
                                                              if bn else (self.norm.weight, self.norm.bias))]
        keep.extend(small)
        return _lib.DcmhtHead(_clip._linear_desc(self.atten.in_proj_weight[2 * E:3 * E], self.atten.in_proj_bias[2 * E:3 * E], precision, keep),
                              _clip._linear_desc(self.atten.out_proj.weight, self.atten.out_proj.bias, precision, keep),
                              int(bn), float(self.norm.eps), small[0].data_ptr(), small[1].data_ptr(),
                              small[2].data_ptr() if bn else None, small[3].data_ptr() if bn else None,
                              _clip._linear_desc(self.fc2.weight, self.fc2.bias, precision, keep))

    def _native(self, data: torch.Tensor) -> torch.Tensor:
        """xmh_head_dcmht: the five launches below from one C call."""
        data = ops._f32c(data).contiguous()
        B, E = data.shape
        desc, precision = _clip._cached_desc(self, self._desc, params=list(self.parameters()) + list(self.buffers()), slot="dcmht")
        nbytes = lib.xmh_head_workspace_bytes(B, E, precision)
        ws = _clip._workspace(nbytes, data.device)
        probs = torch.empty(B, self.fc2.weight.shape[0], dtype=torch.float32, device=data.device)
        check(lib.xmh_head_dcmht(ctypes.byref(desc), ptr(data), B, precision, ptr(probs), None, None, ptr(ws), nbytes, current_stream()),
              "xmh_head_dcmht")
        return probs

    @torch.no_grad()
    def forward(self, data: torch.Tensor) -> torch.Tensor:
        if isinstance(self.norm, nn.BatchNorm1d) and self.training:
            raise RuntimeError("the HIP path implements eval-mode BatchNorm only (running statistics)")
        if _clip.NATIVE_FORWARD and data.dim() == 2:
            return self._native(data)
        E = data.shape[1]
        wv, bv = self.atten.in_proj_weight[2 * E:3 * E], self.atten.in_proj_bias[2 * E:3 * E]
        v = ops.gemm_nt(data, wv, bv)
        o = ops.gemm_nt(v, self.atten.out_proj.weight, self.atten.out_proj.bias)
        if isinstance(self.norm, nn.BatchNorm1d):
            if self.training:
                raise RuntimeError("the HIP path implements eval-mode BatchNorm only (running statistics)")
            n = ops.affine_cols(o, self.norm.running_mean, self.norm.running_var, self.norm.weight, self.norm.bias, self.norm.eps)
        else:
            n = ops.layernorm(o, self.norm.weight, self.norm.bias, self.norm.eps)
        f = ops.gemm_nt(n, self.fc2.weight, self.fc2.bias, act=ops.ACT_RELU)
        return ops.pair_softmax(f)


class DCMHTHashLayer(nn.Module):
    def __init__(self, feature_size=512, outputDim=64, num_heads=8, batch_first=True, hash_func_="softmax"):
        super().__init__()
        if hash_func_ != "softmax":
            raise NotImplementedError("DCMHT is configured with hash_func: softmax in every shipped config")
        self.img_hash = DCMHTModalityHash(feature_size, outputDim, num_heads, layernorm=False)
        self.txt_hash = DCMHTModalityHash(feature_size, outputDim, num_heads, layernorm=True)

    def encode_img(self, embeds):
        return self.img_hash(embeds)

    def encode_txt(self, embeds):
        return self.txt_hash(embeds)

    def forward(self, img_embeds, txt_embeds):
        return self.encode_img(img_embeds), self.encode_txt(txt_embeds)


class DSPHLinearHash(nn.Module):
    def __init__(self, inputDim=512, outputDim=64):
        super().__init__()
        self.fc = nn.Linear(inputDim, outputDim)
        self.drop_out = nn.Dropout(p=0.2)

    @torch.no_grad()
    def forward(self, data):
        if self.training:
            raise RuntimeError("the HIP path is inference-only (dropout inactive)")
        if _clip.NATIVE_FORWARD and data.dim() == 2:
            data = ops._f32c(data).contiguous()
            B, E = data.shape
            desc, precision = _clip._cached_desc(self, lambda prec, keep: _clip._linear_desc(self.fc.weight, self.fc.bias, prec, keep), slot="dsph")
            nbytes = lib.xmh_head_workspace_bytes(B, E, precision)
            ws = _clip._workspace(nbytes, data.device)
            out = torch.empty(B, self.fc.weight.shape[0], dtype=torch.float32, device=data.device)
            check(lib.xmh_head_dsph(ctypes.byref(desc), ptr(data), B, precision, ptr(out), None, None, None, None, ptr(ws), nbytes,
                                    current_stream()), "xmh_head_dsph")
            return out
        return ops.gemm_nt(data, self.fc.weight, self.fc.bias, act=ops.ACT_TANH)


class DSPHHashLayer(nn.Module):
    def __init__(self, inputDim=512, outputDim=64):
        super().__init__()
        self.img_hash = DSPHLinearHash(inputDim, outputDim)
        self.txt_hash = DSPHLinearHash(inputDim, outputDim)

    def encode_img(self, embeds):
        return self.img_hash(embeds)

    def encode_txt(self, embeds):
        return self.txt_hash(embeds)

    def forward(self, img_embeds, txt_embeds):
        return self.encode_img(img_embeds), self.encode_txt(txt_embeds)
