#!/usr/bin/env python3
"""Golden vectors for the saved-activation forward: the activations inside the reference's own
ResidualAttentionBlocks (models/CLIP/model.py:167-197), read with forward hooks while the UNMODIFIED reference model runs the
image tower on the synthetic ViT-B/32 weights of oracle/make_golden_encode.py.  Only a few rows are kept (layers 0 / 5 / 11,
tokens 0 and 17 of sample 1): tests/golden/encode_saved_b2.npz.  Build-container only (/root/reference).

    python oracle/make_golden_saved.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import _ref_import  # noqa: E402
from oracle.make_golden_encode import SEED, load_weights_module  # noqa: E402

LAYERS, TOKENS, SAMPLE = (0, 5, 11), (0, 17), 1


def main():
    _ref_import.setup()
    torch.set_num_threads(8)
    Wt = load_weights_module()
    import models.CLIP.model as ref_clip
    model = ref_clip.build_model(Wt.synth_clip_state_dict(SEED), return_patches=False).float().eval()
    image = Wt.synth_images(SEED, 2)
    got = {}                                                  # (layer, field) -> [L, B, n]
    hooks = []
    for li in LAYERS:
        blk = model.visual.transformer.resblocks[li]
        hooks.append(blk.register_forward_pre_hook(lambda m, i, li=li: got.__setitem__((li, "x_in"), i[0].clone())))
        hooks.append(blk.ln_1.register_forward_hook(lambda m, i, o, li=li: got.__setitem__((li, "ln1"), o.clone())))
        hooks.append(blk.ln_2.register_forward_hook(lambda m, i, o, li=li: (got.__setitem__((li, "x_mid"), i[0].clone()),
                                                                               got.__setitem__((li, "ln2"), o.clone()))[0]))
        hooks.append(blk.mlp.c_fc.register_forward_hook(lambda m, i, o, li=li: got.__setitem__((li, "fc_pre"), o.clone())))
        hooks.append(blk.mlp.gelu.register_forward_hook(lambda m, i, o, li=li: got.__setitem__((li, "fc_act"), o.clone())))
        hooks.append(blk.register_forward_hook(lambda m, i, o, li=li: got.__setitem__((li, "x_out"), (o[0] if isinstance(o, tuple) else o).clone())))
    with torch.no_grad():
        model.encode_image(image)
    for h in hooks:
        h.remove()
    out = {"layers": np.array(LAYERS), "tokens": np.array(TOKENS), "sample": np.array(SAMPLE), "seed": np.array(SEED)}
    for name in ("x_in", "ln1", "x_mid", "ln2", "fc_pre", "fc_act", "x_out"):
        out[name] = np.stack([np.stack([got[(li, name)][t, SAMPLE].numpy() for t in TOKENS]) for li in LAYERS])     # [layer, token, n]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "encode_saved_b2.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
