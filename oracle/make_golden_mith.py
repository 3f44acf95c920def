"""Generate tests/golden/encode_mith.npz by RUNNING the reference's MITH HashLayer (models/MITH/hash/hash.py).
TEST INFRASTRUCTURE ONLY; runs in the build container.   python oracle/make_golden_mith.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import _ref_import  # noqa: E402
from oracle.make_golden_encode import load_weights_module  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEED = 1814


def mith_state(Wt, ref_sd, K):
    """deterministic tensors for every key of the reference head's state_dict (gcl_t aliases gcl_i)."""
    new = {}
    for k, v in ref_sd.items():
        name = k.replace("gcl_t.", "gcl_i.")
        if k.endswith("position.pe"):
            new[k] = v
        elif ".lns." in k and k.endswith("weight") or (".ln_" in k and k.endswith("weight")):
            new[k] = 1.0 + Wt.synth_tensor(SEED, "mith%d." % K + name, v.shape, 0.05)
        else:
            std = 0.04 if v.dim() > 1 else 0.02
            new[k] = Wt.synth_tensor(SEED, "mith%d." % K + name, v.shape, std)
    return new


def main():
    _ref_import.setup()
    import types
    for pkg in ("models.MITH.hash",):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(_ref_import.REF, *pkg.split("."))]
        sys.modules.setdefault(pkg, m)
    from models.MITH.hash.hash import HashLayer
    Wt = load_weights_module()
    rec = {}
    for K in (16, 64):
        ref = HashLayer(clip_embed_dim=512, k_bits=K, dropout=0.0, transformer_layers=2, activation="gelu", top_k_label=8, res_mlp_layers=2).eval()
        ref.load_state_dict(mith_state(Wt, ref.state_dict(), K))
        B = 40                                            # 640 / 2560 code bits per modality: the "<1 % flips" bound means something
        cls_i = Wt.synth_tensor(SEED, "mith_in.cls_i", (B, 512), 0.6)
        tok_i = Wt.synth_tensor(SEED, "mith_in.tok_i", (49, B, 512), 0.6)
        cls_t = Wt.synth_tensor(SEED, "mith_in.cls_t", (B, 512), 0.6)
        tok_t = Wt.synth_tensor(SEED, "mith_in.tok_t", (32, B, 512), 0.6)
        mask = torch.zeros(B, 32, dtype=torch.bool)
        for b in range(B):
            mask[b, 4 + (7 * b) % 27:] = True             # caption lengths 4..30 (tests/test_oracle_encode.py: mith_inputs)
        with torch.no_grad():
            _, ch_i, th_i, _ = ref.encode_img(cls_i, tok_i)
            _, ch_t, th_t, _ = ref.encode_txt(cls_t, tok_t, mask)
        rec.update({"k%d_cls_hash_i" % K: ch_i.numpy(), "k%d_tok_hash_i" % K: th_i.numpy(), "k%d_cls_hash_t" % K: ch_t.numpy(),
                    "k%d_tok_hash_t" % K: th_t.numpy(), "k%d_code_i" % K: (ch_i + th_i).sign().numpy(), "k%d_code_t" % K: (ch_t + th_t).sign().numpy()})
        rec["mask"] = mask.numpy()
        print(K, float(th_i.abs().mean()), float(th_t.abs().mean()), sorted(ref.state_dict().keys())[:3], len(ref.state_dict()))
    np.savez_compressed(os.path.join(OUT, "encode_mith.npz"), seed=SEED, **rec)


if __name__ == "__main__":
    main()
