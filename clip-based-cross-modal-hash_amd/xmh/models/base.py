"""``BaseModel`` with the surface the reference's runner relies on (models/base.py:10-70): ``load_backbone``,
``encode_image`` / ``encode_text`` / ``object_function`` to override, ``forward``, ``freezen`` / ``unfreezen``,
``from_config``.  Inference only: the forward pass runs in libxmh.so, so there is no autograd graph."""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from .clip import build_model as build_clip
from .weights import synth_clip_state_dict


class BaseModel(nn.Module):

    def __init__(self, cfg=None):
        super().__init__()
        self.cfg = cfg

    def load_backbone(self, clipPath: str, return_patches: bool = False) -> tuple:
        """reference models/base.py:18-31.  ``clipPath`` is a TorchScript archive or a plain state_dict file;
        additionally ``synthetic[:seed[:k=v,...]]`` builds the deterministic random-init weights used by the
        benchmarks and parity tests (there is no checkpoint in this repository)."""
        if clipPath.startswith("synthetic"):
            parts = clipPath.split(":")
            seed = int(parts[1]) if len(parts) > 1 and parts[1] else 1814
            over = {}
            if len(parts) > 2:
                over = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in parts[2].split(",") if kv}
            state_dict = synth_clip_state_dict(seed, **over)
        elif os.path.exists(clipPath):
            try:
                state_dict = torch.jit.load(clipPath, map_location="cpu").eval().state_dict()
            except RuntimeError:
                state_dict = torch.load(clipPath, map_location="cpu")
        else:
            print("pretrained CLIP model doesn't exist!")
            exit(0)
        embed_dim = state_dict["text_projection"].shape[1]
        n_tokens = state_dict["visual.positional_embedding"].shape[0]
        clip = build_clip(state_dict, return_patches=return_patches)
        if return_patches:
            return embed_dim, n_tokens, clip
        return embed_dim, clip

    def encode_image(self, x):
        raise NotImplementedError()

    def encode_text(self, x):
        raise NotImplementedError()

    def object_function(self, a, b, labels=None, indexs=None, **kwags):
        raise NotImplementedError()

    def forward(self, image, text, labels=None, indexs=None, return_loss=False):
        """both towers; with ``return_loss`` the method's objective instead of the pair (reference models/base.py:52-60)"""
        embeds = (self.encode_image(image), self.encode_text(text))
        return self.object_function(*embeds, labels=labels, indexs=indexs) if return_loss else embeds

    @classmethod
    def from_config(cls, cfg, output_dim=None, train_num=None):
        raise NotImplementedError()

    def freezen(self):
        for p in self.parameters():
            p.requires_grad = False

    def unfreezen(self):
        for p in self.parameters():
            p.requires_grad = True
