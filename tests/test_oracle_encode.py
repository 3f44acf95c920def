"""Pin the encoder oracle (oracle/encode.py) against goldens produced by the reference's own modules (CPU only)."""
import os

import numpy as np
import torch

from conftest import GOLDEN
from oracle import encode as enc


def _weights():
    from xmh.models import weights
    return weights


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-12)


def test_clip_towers_match_reference_goldens():
    g = np.load(os.path.join(GOLDEN, "encode_clip_b2.npz"))
    W = _weights()
    seed = int(g["seed"])
    sd = enc.fp16_round_like_reference(W.synth_clip_state_dict(seed))
    image = W.synth_images(seed, 2)
    ids, pad = W.synth_text(seed, 2)
    torch.set_num_threads(8)
    with torch.no_grad():
        probe = []
        cls = enc.clip_image(sd, image, probe=probe)
        assert _rel(cls, g["img_cls"]) < 2e-5
        assert _rel(torch.stack(probe), g["img_block_cls"]) < 2e-5
        assert _rel(enc.clip_text(sd, ids), g["txt_eos"]) < 2e-5
        cls2, tok = enc.clip_image(sd, image, return_patches=True)
        assert _rel(cls2, g["img_cls_rp"]) < 2e-5 and _rel(tok, g["img_tokens_rp"]) < 2e-5
        eos, ttok, nm = enc.clip_text(sd, ids, key_padding_mask=pad, return_patches=True)
        assert _rel(eos, g["txt_eos_rp"]) < 2e-5
        assert np.array_equal(nm.numpy(), g["txt_mask_rp"])
        keep = ~g["txt_mask_rp"].T                                    # [L, B]: padded rows are NaN-free but unspecified
        assert _rel(ttok.numpy()[keep], g["txt_tokens_rp"][keep]) < 2e-5


def saved_subset(saved, x_out_of, g):
    """the rows the golden file keeps: [layer, token, n] per field, from per-layer records of [L, B, n] tensors"""
    b = int(g["sample"])
    out = {}
    for name in ("x_in", "ln1", "x_mid", "ln2", "fc_pre", "fc_act"):
        out[name] = np.stack([np.stack([np.asarray(saved[li][name][t, b]) for t in g["tokens"]]) for li in g["layers"]])
    out["x_out"] = np.stack([np.stack([np.asarray(x_out_of(li)[t, b]) for t in g["tokens"]]) for li in g["layers"]])
    return out


def test_saved_activations_of_the_oracle_match_the_reference_hooks():
    """oracle.blocks_saved against forward hooks inside the reference's own ResidualAttentionBlocks (oracle/make_golden_saved.py)"""
    g = np.load(os.path.join(GOLDEN, "encode_saved_b2.npz"))
    W = _weights()
    seed = int(g["seed"])
    sd = enc.fp16_round_like_reference(W.synth_clip_state_dict(seed))
    torch.set_num_threads(8)
    with torch.no_grad():
        x = enc.vit_front(sd, W.synth_images(seed, 2)).permute(1, 0, 2)
        y, saved = enc.blocks_saved(x, sd, "visual.transformer.", 12, 12, None)
        assert _rel(y, enc._blocks(x, sd, "visual.transformer.", 12, 12, None)) < 1e-6
    got = saved_subset(saved, lambda li: saved[li + 1]["x_in"] if li + 1 < 12 else y, g)
    for name, v in got.items():
        assert _rel(v, g[name]) < 2e-5, name
    # the two fields no module output exposes: qkv is in_proj of ln1, attn is what out_proj maps to x_mid - x_in
    r = saved[5]
    p = "visual.transformer.resblocks.5."
    assert _rel(torch.nn.functional.linear(r["attn"], sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"]), r["x_mid"] - r["x_in"]) < 1e-5
    assert r["qkv"].shape == (50, 2, 2304)


def head_params(W, seed, prefix, shapes):
    return {k: v for k, v in shapes.items()}


def dcmht_params(W, seed, K, modality, family="dcmht"):
    """the same named tensors oracle/make_golden_encode.py (make_golden_twdh.py: family "twdh") loaded into the reference head."""
    pre = "%s%d.%s_hash." % (family, K, modality)
    p = {
        "atten.in_proj_weight": W.synth_tensor(seed, pre + "atten.in_proj_weight", (1536, 512), 0.05),
        "atten.in_proj_bias": W.synth_tensor(seed, pre + "atten.in_proj_bias", (1536,), 0.02),
        "atten.out_proj.weight": W.synth_tensor(seed, pre + "atten.out_proj.weight", (512, 512), 0.05),
        "atten.out_proj.bias": W.synth_tensor(seed, pre + "atten.out_proj.bias", (512,), 0.02),
        "norm.weight": 1.0 + W.synth_tensor(seed, pre + "norm.weight", (512,), 0.1),
        "norm.bias": W.synth_tensor(seed, pre + "norm.bias", (512,), 0.02),
        "fc2.weight": W.synth_tensor(seed, pre + "fc2.weight", (2 * K, 512), 0.05),
        "fc2.bias": W.synth_tensor(seed, pre + "fc2.bias", (2 * K,), 0.02),
    }
    if modality == "img":
        p["norm.running_mean"] = W.synth_tensor(seed, pre + "norm.running_mean", (512,), 0.02)
        p["norm.running_var"] = W.synth_tensor(seed, pre + "norm.running_var", (512,), 0.2).abs() + 0.5
    return p


def dsph_params(W, seed, K, modality):
    pre = "dsph%d.%s_hash." % (K, modality)
    return {"fc.weight": W.synth_tensor(seed, pre + "fc.weight", (K, 512), 0.05), "fc.bias": W.synth_tensor(seed, pre + "fc.bias", (K,), 0.02)}


def test_heads_match_reference_goldens():
    from oracle import retrieval as orc
    g = np.load(os.path.join(GOLDEN, "encode_heads.npz"))
    W = _weights()
    seed = int(g["seed"])
    emb = torch.from_numpy(g["emb"])
    assert torch.equal(emb, W.synth_tensor(seed, "head_input", (40, 512), 0.5))
    for K in (16, 64):
        for mod in ("img", "txt"):
            out = enc.dcmht_head(dcmht_params(W, seed, K, mod), emb, image=(mod == "img"))
            assert np.abs(out.numpy() - g["dcmht%d_%s" % (K, mod)]).max() < 2e-6
            code = orc.hash_code_pair_argmax(out.clone()).numpy()
            assert (code != g["dcmht%d_%s_code" % (K, mod)]).mean() < 0.002        # only 1-ulp near-ties may differ
    for mod in ("img", "txt"):
        out = enc.dsph_head(dsph_params(W, seed, 128, mod), emb)
        assert np.abs(out.numpy() - g["dsph128_%s" % mod]).max() < 2e-6
        assert (np.sign(out.numpy()) != g["dsph128_%s_code" % mod]).mean() < 0.002


def mith_params(W, seed, K):
    """named tensors identical to the ones oracle/make_golden_mith.py loaded into the reference HashLayer; keys are the
    reference head's state_dict keys (gcl_t.* aliases gcl_i.*)."""
    import math
    p = {}

    def t(name, shape, std, plus=0.0):
        p[name] = plus + W.synth_tensor(seed, "mith%d." % K + name, shape, std)
    for i in range(2):
        t("gcl_i.mlp.mlps.%d.0.weight" % i, (2048, 512), 0.04)
        t("gcl_i.mlp.mlps.%d.0.bias" % i, (2048,), 0.02)
        t("gcl_i.mlp.mlps.%d.3.weight" % i, (512, 2048), 0.04)
        t("gcl_i.mlp.mlps.%d.3.bias" % i, (512,), 0.02)
        t("gcl_i.mlp.lns.%d.weight" % i, (512,), 0.05, 1.0)
        t("gcl_i.mlp.lns.%d.bias" % i, (512,), 0.02)
    t("gcl_i.common_concept_embedding.weight", (K, 512), 0.04)
    for m in ("i", "t"):
        pre = "lct_%s." % m
        pe = torch.zeros(K, 512)
        pos = torch.arange(0, K, dtype=torch.float).unsqueeze(1)
        div = torch.exp(torch.arange(0, 512, 2).float() * (-math.log(10000.0) / 512))
        pe[:, 0::2], pe[:, 1::2] = torch.sin(pos * div), torch.cos(pos * div)
        p[pre + "position.pe"] = pe.unsqueeze(0).transpose(0, 1) / (512 ** 0.5)
        for i in range(2):
            b = pre + "transformer.resblocks.%d." % i
            t(b + "attn.in_proj_weight", (1536, 512), 0.04)
            t(b + "attn.in_proj_bias", (1536,), 0.02)
            t(b + "attn.out_proj.weight", (512, 512), 0.04)
            t(b + "attn.out_proj.bias", (512,), 0.02)
            t(b + "ln_1.weight", (512,), 0.05, 1.0)
            t(b + "ln_1.bias", (512,), 0.02)
            t(b + "mlp.c_fc.weight", (2048, 512), 0.04)
            t(b + "mlp.c_fc.bias", (2048,), 0.02)
            t(b + "mlp.c_proj.weight", (512, 2048), 0.04)
            t(b + "mlp.c_proj.bias", (512,), 0.02)
            t(b + "ln_2.weight", (512,), 0.05, 1.0)
            t(b + "ln_2.bias", (512,), 0.02)
        for k in range(K):
            t(pre + "hashing.fc_list.%d.weight" % k, (1, 512), 0.04)
            t(pre + "hashing.fc_list.%d.bias" % k, (1,), 0.02)
    for m in ("img", "txt"):
        t("%s_concept_proj.weight" % m, (512, 512), 0.04)
        t("%s_concept_proj.bias" % m, (512,), 0.02)
    return p


def mith_inputs(W, seed):
    B = 40
    mask = torch.zeros(B, 32, dtype=torch.bool)
    for b in range(B):
        mask[b, 4 + (7 * b) % 27:] = True                 # as in oracle/make_golden_mith.py
    return (W.synth_tensor(seed, "mith_in.cls_i", (B, 512), 0.6), W.synth_tensor(seed, "mith_in.tok_i", (49, B, 512), 0.6),
            W.synth_tensor(seed, "mith_in.cls_t", (B, 512), 0.6), W.synth_tensor(seed, "mith_in.tok_t", (32, B, 512), 0.6), mask)


def test_mith_head_matches_reference_goldens():
    g = np.load(os.path.join(GOLDEN, "encode_mith.npz"))
    W = _weights()
    seed = int(g["seed"])
    cls_i, tok_i, cls_t, tok_t, mask = mith_inputs(W, seed)
    assert np.array_equal(mask.numpy(), g["mask"])
    for K in (16, 64):
        hp = mith_params(W, seed, K)
        with torch.no_grad():
            ch_i, th_i = enc.mith_head(hp, cls_i, tok_i, None, "i")
            ch_t, th_t = enc.mith_head(hp, cls_t, tok_t, mask, "t")
        for got, key in ((ch_i, "cls_hash_i"), (th_i, "tok_hash_i"), (ch_t, "cls_hash_t"), (th_t, "tok_hash_t")):
            assert np.abs(got.numpy() - g["k%d_%s" % (K, key)]).max() < 5e-6, (K, key)
        assert np.array_equal((ch_i + th_i).sign().numpy(), g["k%d_code_i" % K])
        assert np.array_equal((ch_t + th_t).sign().numpy(), g["k%d_code_t" % K])


def test_twdh_long_and_short_heads_match_reference_golden():
    """DCMHT hash layer at long_dim = 512 and the TwDH short transforms (models/TwDH/TwDH.py:66-85)."""
    import importlib
    from oracle import encode as E
    W = importlib.import_module("xmh.models.weights")
    g = np.load(os.path.join(GOLDEN, "encode_twdh.npz"))
    emb = torch.from_numpy(g["emb"])
    for name in ("img", "txt"):
        long_hash = E.dcmht_head(dcmht_params(W, int(g["seed"]), 512, name, family="twdh"), emb, image=name == "img")
        assert np.allclose(long_hash.numpy(), g["long_%s" % name], rtol=0, atol=2e-6)
        for S in (16, 64):
            short = E.twdh_short(torch.from_numpy(g["long_%s" % name]), torch.from_numpy(g["trans%d" % S]))
            assert np.allclose(short.numpy(), g["short%d_%s" % (S, name)], rtol=0, atol=2e-6)
