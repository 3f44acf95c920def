"""Oracle (CPU restatement) of the reference's encoder maths.  TEST INFRASTRUCTURE ONLY.

Plain torch-CPU fp32 functional code over a CLIP ``state_dict`` (reference key names), no nn.Module: it is
independent of both the reference's module classes and the HIP path.  Pinned by ``tests/golden/encode_*.npz``
(generated from the imported reference by ``oracle/make_golden_encode.py``).

  clip_image / clip_text   models/CLIP/model.py:232-268, :373-396 (+ ResidualAttentionBlock :167-197,
                           LayerNorm :153-159, QuickGELU :162-164, build_attention_mask :358-364)
  dcmht_head               models/DCMHT/hash/hash.py:15-46 (+ softmax_hash, models/common/hash.py:21-31)
  dsph_head                models/DSPH/hash/hash.py:6-15
  mith_head_*              models/MITH/hash/hash.py:9-254, runners/MITH/runner.py:125-131
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def fp16_round_like_reference(sd: dict) -> dict:
    """convert_weights (models/CLIP/model.py:415-436) then .float(): Conv/Linear/MHA/proj tensors fp16-rounded."""
    out = {}
    for k, v in sd.items():
        v = v.float()
        hit = (k.endswith("conv1.weight") or ".attn.in_proj_" in k or ".attn.out_proj." in k or ".mlp.c_fc." in k
               or ".mlp.c_proj." in k or k in ("visual.proj", "text_projection"))
        out[k] = v.half().float() if hit else v
    return out


def _mha(x, w_in, b_in, w_out, b_out, heads, mask):
    """x [L, N, E] -> [L, N, E]; mask additive [N, L, L] or None (nn.MultiheadAttention, need_weights path)."""
    L, N, E = x.shape
    dh = E // heads
    qkv = F.linear(x, w_in, b_in)
    q, k, v = qkv.chunk(3, dim=-1)

    def split(t):
        return t.contiguous().view(L, N * heads, dh).transpose(0, 1)           # [N*H, L, dh]
    q, k, v = split(q) * (1.0 / math.sqrt(dh)), split(k), split(v)
    s = torch.bmm(q, k.transpose(1, 2))
    if mask is not None:
        s = s + mask.repeat_interleave(heads, dim=0)
    p = torch.softmax(s, dim=-1)
    o = torch.bmm(p, v).transpose(0, 1).contiguous().view(L, N, E)
    return F.linear(o, w_out, b_out)


def _blocks(x, sd, prefix, layers, heads, mask, probe=None):
    for i in range(layers):
        p = "%sresblocks.%d." % (prefix, i)
        E = x.shape[-1]
        h = F.layer_norm(x, (E,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], 1e-5)
        x = x + _mha(h, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"], sd[p + "attn.out_proj.weight"],
                     sd[p + "attn.out_proj.bias"], heads, mask)
        h = F.layer_norm(x, (E,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], 1e-5)
        f = F.linear(h, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"])
        f = f * torch.sigmoid(1.702 * f)
        x = x + F.linear(f, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])
        if probe is not None:
            probe.append(x[0].clone())                                           # token 0 of every sample
    return x


def blocks_saved(x, sd, prefix, layers, heads, mask):
    """ResidualAttentionBlock stack (models/CLIP/model.py:167-211) on x [L, N, E], every intermediate a backward pass reads kept:
    -> (x_out, [per layer {x_in, ln1, qkv, attn (heads' outputs before out_proj), x_mid, ln2, fc_pre, fc_act}])"""
    saved = []
    for i in range(layers):
        p = "%sresblocks.%d." % (prefix, i)
        L, N, E = x.shape
        dh = E // heads
        rec = {"x_in": x}
        rec["ln1"] = F.layer_norm(x, (E,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], 1e-5)
        rec["qkv"] = F.linear(rec["ln1"], sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"])
        q, k, v = rec["qkv"].chunk(3, dim=-1)

        def split(t):
            return t.contiguous().view(L, N * heads, dh).transpose(0, 1)
        s = torch.bmm(split(q) * (1.0 / math.sqrt(dh)), split(k).transpose(1, 2))
        if mask is not None:
            s = s + mask.repeat_interleave(heads, dim=0)
        rec["attn"] = torch.bmm(torch.softmax(s, dim=-1), split(v)).transpose(0, 1).contiguous().view(L, N, E)
        rec["x_mid"] = x + F.linear(rec["attn"], sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"])
        rec["ln2"] = F.layer_norm(rec["x_mid"], (E,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], 1e-5)
        rec["fc_pre"] = F.linear(rec["ln2"], sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"])
        rec["fc_act"] = rec["fc_pre"] * torch.sigmoid(1.702 * rec["fc_pre"])
        x = rec["x_mid"] + F.linear(rec["fc_act"], sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])
        saved.append(rec)
    return x, saved


def _count_layers(sd, prefix):
    return len({k.split("resblocks.")[1].split(".")[0] for k in sd if k.startswith(prefix + "resblocks.")})


def vit_front(sd, image):
    """conv1 + class / positional embedding + ln_pre (models/CLIP/model.py:232-244) -> x [B, L, width] entering the block stack"""
    w = sd["visual.conv1.weight"]
    width, patch = w.shape[0], w.shape[-1]
    x = F.conv2d(image.float(), w, stride=patch)
    x = x.reshape(x.shape[0], width, -1).permute(0, 2, 1)
    cls = sd["visual.class_embedding"] + torch.zeros(x.shape[0], 1, width)
    x = torch.cat([cls, x], dim=1) + sd["visual.positional_embedding"]
    return F.layer_norm(x, (width,), sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"], 1e-5)


def clip_image(sd, image, return_patches=False, probe=None):
    """-> cls [B, out] (and tokens [n_patches, B, out] when return_patches)."""
    x = vit_front(sd, image)
    width = x.shape[-1]
    x = _blocks(x.permute(1, 0, 2), sd, "visual.transformer.", _count_layers(sd, "visual.transformer."), width // 64, None, probe)
    x = F.layer_norm(x.permute(1, 0, 2), (width,), sd["visual.ln_post.weight"], sd["visual.ln_post.bias"], 1e-5)
    x = x @ sd["visual.proj"]
    if return_patches:
        return x[:, 0], x[:, 1:].permute(1, 0, 2)
    return x[:, 0]


def clip_text(sd, ids, key_padding_mask=None, return_patches=False, probe=None):
    """-> eos [B, out] (and tokens [L, B, out], new mask when return_patches)."""
    B, L = ids.shape
    width = sd["ln_final.weight"].shape[0]
    x = sd["token_embedding.weight"][ids] + sd["positional_embedding"][:L]
    mask = torch.full((L, L), float("-inf")).triu_(1)[None].repeat(B, 1, 1)
    if key_padding_mask is not None:
        mask = mask.masked_fill(key_padding_mask[:, None, :].bool(), float("-inf"))
    x = _blocks(x.permute(1, 0, 2), sd, "transformer.", _count_layers(sd, "transformer."), width // 64, mask, probe)
    x = F.layer_norm(x.permute(1, 0, 2), (width,), sd["ln_final.weight"], sd["ln_final.bias"], 1e-5)
    x = x @ sd["text_projection"]
    eos = ids.argmax(dim=-1)
    e = x[torch.arange(B), eos]
    if return_patches:
        new_mask = None if key_padding_mask is None else (key_padding_mask.bool() | (ids == 49407))
        return e, x.permute(1, 0, 2), new_mask
    return e


# ---------------------------------------------------------------------------------------------------
# heads.  `hp` maps the reference's head state_dict keys (relative to hash.img_hash / hash.txt_hash) to tensors.
# ---------------------------------------------------------------------------------------------------
def dcmht_head(hp, e, image: bool):
    E = e.shape[1]
    v = F.linear(e, hp["atten.in_proj_weight"][2 * E:], hp["atten.in_proj_bias"][2 * E:])
    o = F.linear(v, hp["atten.out_proj.weight"], hp["atten.out_proj.bias"])
    if image:
        n = F.batch_norm(o, hp["norm.running_mean"], hp["norm.running_var"], hp["norm.weight"], hp["norm.bias"], False, 0.0, 1e-5)
    else:
        n = F.layer_norm(o, (E,), hp["norm.weight"], hp["norm.bias"], 1e-5)
    f = torch.relu(F.linear(n, hp["fc2.weight"], hp["fc2.bias"]))
    return torch.softmax(f.view(f.shape[0], -1, 2), dim=-1).view(f.shape[0], -1)


def dsph_head(hp, e):
    return torch.tanh(F.linear(e, hp["fc.weight"], hp["fc.bias"]))


def _mith_gcl(hp, x):
    """GlobalConceptLearning (models/MITH/hash/hash.py:88-106) with ResidualMLPs (:9-38): -> (mlp(x), tanh(Wc mlp(x)))."""
    i = 0
    while "gcl_i.mlp.lns.%d.weight" % i in hp:
        p = "gcl_i.mlp."
        h = F.layer_norm(x, (x.shape[-1],), hp[p + "lns.%d.weight" % i], hp[p + "lns.%d.bias" % i], 1e-5)
        f = F.gelu(F.linear(h, hp[p + "mlps.%d.0.weight" % i], hp[p + "mlps.%d.0.bias" % i]))
        x = x + F.linear(f, hp[p + "mlps.%d.3.weight" % i], hp[p + "mlps.%d.3.bias" % i])
        i += 1
    return x, torch.tanh(F.linear(x, hp["gcl_i.common_concept_embedding.weight"]))


def mith_head(hp, cls, tokens_lnd, mask, modality: str, top_k: int = 8):
    """Eval path of models/MITH/hash/hash.py:231-247 for one modality ('i' or 't'): -> (cls_hash [B,K], tokens_hash [B,K]).
    tokens_lnd [L,B,D]; mask [B,L] bool or None (text: padding | EOS)."""
    _, cls_hash = _mith_gcl(hp, cls)
    _, sim = _mith_gcl(hp, tokens_lnd)                                   # [L,B,K]
    sim = sim.clone()
    L, B, K = sim.shape
    if mask is not None:
        sim = sim + torch.where(mask, float("-inf"), 0.0).t()[:, :, None]
    sim = torch.where(sim > 0, sim, torch.full_like(sim, float("-inf")))
    kth = torch.topk(sim, k=top_k, dim=-1).values.min(dim=-1, keepdim=True).values
    sim = torch.where(sim >= kth, sim, torch.full_like(sim, float("-inf")))
    att = torch.softmax(sim, dim=0)
    att = torch.where(torch.isnan(att), torch.zeros_like(att), att)
    merged = torch.bmm(att.permute(1, 2, 0), tokens_lnd.permute(1, 0, 2)).permute(1, 0, 2)       # [K,B,D]
    pre = "lct_%s." % modality
    x = merged + hp[pre + "position.pe"][:K]
    sd = {k[len(pre + "transformer."):]: v for k, v in hp.items() if k.startswith(pre + "transformer.")}
    x = _blocks(x, sd, "", _count_layers(sd, ""), x.shape[-1] // 64, None)
    bits = torch.stack([F.linear(x[k], hp[pre + "hashing.fc_list.%d.weight" % k], hp[pre + "hashing.fc_list.%d.bias" % k]) for k in range(K)])
    return cls_hash, torch.tanh(bits.permute(1, 0, 2).squeeze(-1))


def twdh_short(long_hash: torch.Tensor, trans: torch.Tensor) -> torch.Tensor:
    """TwDH short code of one transform (reference models/TwDH/TwDH.py:70-74 / :80-84):
    ``hash.quantization(long_hash.matmul(trans))`` with the softmax hash of models/common/hash.py:21-31 --
    [B, 2*long] x [2*long, 2*short] -> pair softmax -> [B, 2*short]."""
    z = long_hash.matmul(trans)
    return torch.softmax(z.view(z.shape[0], -1, 2), dim=-1).view(z.shape[0], -1)
