"""Hash heads of the in-scope methods, parameters under the reference's key names, forward in libxmh.so.

  DCMHT  models/DCMHT/hash/hash.py:15-82   MHA on a length-1 sequence == out_proj(v_proj(x)) (softmax over one key
         is 1), BatchNorm1d (image) / LayerNorm (text), fc2, relu, pair softmax
  DSPH   models/DSPH/hash/hash.py:6-45     tanh(fc(x)) (dropout is identity in eval)
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops


class DCMHTModalityHash(nn.Module):
    def __init__(self, inputDim=512, outputDim=64, num_heads=8, layernorm=True):
        super().__init__()
        self.bit = outputDim
        self.atten = nn.MultiheadAttention(inputDim, num_heads=num_heads, batch_first=True)
        self.norm = nn.LayerNorm(inputDim) if layernorm else nn.BatchNorm1d(inputDim)
        self.fc2 = nn.Linear(inputDim, outputDim * 2)

    @torch.no_grad()
    def forward(self, data: torch.Tensor) -> torch.Tensor:
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
