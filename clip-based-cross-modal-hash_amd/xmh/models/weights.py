"""Build-owned deterministic weights: no checkpoint is ever stored in the repo.

``synth_clip_state_dict(seed)`` produces a ViT-B/32-shaped CLIP state_dict -- the key names and shapes of the
weight-file contract in SURVEY 8c -- from a CPU ``torch.Generator`` keyed by (seed, key name), so the golden
script (feeding the *reference* ``build_model``) and the GPU box regenerate the very same tensors.  Scales follow
the reference's ``initialize_parameters`` orders of magnitude (models/CLIP/model.py:330-357) so activations stay
in a realistic range; LayerNorm gains are perturbed away from 1 so that affine mistakes show up in tests.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import torch

VIT_B32 = dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768, vision_patch_size=32,
               context_length=77, vocab_size=49408, transformer_width=512, transformer_heads=8, transformer_layers=12)


def _gen(seed: int, name: str) -> torch.Generator:
    return torch.Generator(device="cpu").manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2**63 - 1))


def _normal(seed, name, shape, std):
    return torch.randn(*shape, generator=_gen(seed, name), dtype=torch.float32) * std


def _ln(seed, prefix, width, sd):
    sd[prefix + ".weight"] = 1.0 + _normal(seed, prefix + ".weight", (width,), 0.05)
    sd[prefix + ".bias"] = _normal(seed, prefix + ".bias", (width,), 0.02)


def _blocks(seed, prefix, width, layers, sd):
    proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
    attn_std = width ** -0.5
    fc_std = (2 * width) ** -0.5
    for i in range(layers):
        p = "%sresblocks.%d." % (prefix, i)
        sd[p + "attn.in_proj_weight"] = _normal(seed, p + "attn.in_proj_weight", (3 * width, width), attn_std)
        sd[p + "attn.in_proj_bias"] = _normal(seed, p + "attn.in_proj_bias", (3 * width,), 0.01)
        sd[p + "attn.out_proj.weight"] = _normal(seed, p + "attn.out_proj.weight", (width, width), proj_std)
        sd[p + "attn.out_proj.bias"] = _normal(seed, p + "attn.out_proj.bias", (width,), 0.01)
        _ln(seed, p + "ln_1", width, sd)
        sd[p + "mlp.c_fc.weight"] = _normal(seed, p + "mlp.c_fc.weight", (4 * width, width), fc_std)
        sd[p + "mlp.c_fc.bias"] = _normal(seed, p + "mlp.c_fc.bias", (4 * width,), 0.01)
        sd[p + "mlp.c_proj.weight"] = _normal(seed, p + "mlp.c_proj.weight", (width, 4 * width), proj_std)
        sd[p + "mlp.c_proj.bias"] = _normal(seed, p + "mlp.c_proj.bias", (width,), 0.01)
        _ln(seed, p + "ln_2", width, sd)


def synth_clip_state_dict(seed: int = 1814, **overrides) -> "OrderedDict[str, torch.Tensor]":
    """CLIP state_dict with the reference's key names.  ``overrides`` shrink the architecture for fast CPU
    tests (e.g. vision_layers=2, transformer_layers=2); the default is full ViT-B/32."""
    c = dict(VIT_B32)
    c.update(overrides)
    vw, tw, ed = c["vision_width"], c["transformer_width"], c["embed_dim"]
    grid = c["image_resolution"] // c["vision_patch_size"]
    sd = OrderedDict()
    sd["visual.class_embedding"] = _normal(seed, "visual.class_embedding", (vw,), vw ** -0.5)
    sd["visual.positional_embedding"] = _normal(seed, "visual.positional_embedding", (grid * grid + 1, vw), vw ** -0.5)
    sd["visual.proj"] = _normal(seed, "visual.proj", (vw, ed), vw ** -0.5)
    sd["visual.conv1.weight"] = _normal(seed, "visual.conv1.weight", (vw, 3, c["vision_patch_size"], c["vision_patch_size"]), 0.02)
    _ln(seed, "visual.ln_pre", vw, sd)
    _blocks(seed, "visual.transformer.", vw, c["vision_layers"], sd)
    _ln(seed, "visual.ln_post", vw, sd)
    sd["positional_embedding"] = _normal(seed, "positional_embedding", (c["context_length"], tw), 0.01)
    sd["text_projection"] = _normal(seed, "text_projection", (tw, ed), tw ** -0.5)
    sd["logit_scale"] = torch.tensor(2.6592)
    sd["token_embedding.weight"] = _normal(seed, "token_embedding.weight", (c["vocab_size"], tw), 0.02)
    _blocks(seed, "transformer.", tw, c["transformer_layers"], sd)
    _ln(seed, "ln_final", tw, sd)
    return sd


def synth_tensor(seed: int, name: str, shape, std: float) -> torch.Tensor:
    """named deterministic tensor for head weights (same generator family)."""
    return _normal(seed, name, tuple(shape), std)


def synth_images(seed: int, B: int, res: int = 224) -> torch.Tensor:
    return torch.randn(B, 3, res, res, generator=_gen(seed, "images"), dtype=torch.float32)


def synth_text(seed: int, B: int, L: int = 32, vocab: int = 49408):
    """[SOS, U(1, vocab-3) x len, EOS, 0...] with len ~ U(4, L-2) and the padding mask ids == 0."""
    g = _gen(seed, "text")
    ids = torch.zeros(B, L, dtype=torch.int64)
    lens = torch.randint(4, L - 1, (B,), generator=g)
    for b in range(B):
        n = int(lens[b])
        ids[b, 0] = vocab - 2
        ids[b, 1:1 + n] = torch.randint(1, vocab - 3, (n,), generator=g)
        ids[b, 1 + n] = vocab - 1
    return ids, ids == 0
