#!/usr/bin/env python3
"""images/s of the ViT-B/32 + DCMHT forward at batch 100 and 400 under the tile-rule switches of the GEMM dispatcher
(XMH_GEMM_TILE_RULES bit mask, XMH_GEMM_NO_WIDE; read once per process): run on the GPU box, one process per setting
    for r in 30 26 22 14; do XMH_GEMM_TILE_RULES=$r python tools/sweep_gemm_rules.py; done"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
import torch
from xmh import ops, retrieval as R
from xmh.models.dcmht import DCMHT
from xmh.models import weights as W
from xmh.utils.config import Config
model = DCMHT.from_config(Config({"clip_path": "synthetic:1814"}), output_dim=64).cuda().eval()
img = W.synth_images(5, 100).cuda()
big = img.repeat(4, 1, 1, 1)
res = []
for mode in ("f32", "f16"):
    ops.set_precision(mode)
    for name, x in (("b100", img), ("b400", big)):
        for _ in range(3):
            R.pack_pair_argmax(model.encode_image(x))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            R.pack_pair_argmax(model.encode_image(x))
        torch.cuda.synchronize()
        res.append("%s %s %.0f img/s" % (mode, name, x.shape[0] * n / (time.perf_counter() - t0)))
ops.set_precision("f32")
print("rules=%s nowide=%s : " % (os.environ.get("XMH_GEMM_TILE_RULES", "30"), os.environ.get("XMH_GEMM_NO_WIDE", "-")) + " | ".join(res))
