#!/usr/bin/env python3
"""image + caption batches encoded repeatedly (both towers, two streams): the packed codes must be bit-identical every time (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
import torch
from xmh import ops, towers, retrieval as R
from xmh.models.dcmht import DCMHT
from xmh.models import weights as W
from xmh.utils.config import Config
model = DCMHT.from_config(Config({"clip_path": "synthetic:1814"}), output_dim=64).cuda().eval()
for B in (100, 400, 37):
    img = W.synth_images(5, 100).cuda().repeat((B + 99) // 100, 1, 1, 1)[:B]
    ids = W.synth_text(5, 100)[0].cuda().repeat((B + 99) // 100, 1)[:B]
    first = None
    bad = 0
    for i in range(40):
        with torch.no_grad():
            a, b = towers.run_both(lambda: model.encode_image(img), lambda: model.encode_text(ids))
        cur = (a.clone(), b.clone())
        if first is None: first = cur
        elif not (torch.equal(cur[0], first[0]) and torch.equal(cur[1], first[1])): bad += 1
    print("batch", B, "evaluations that differ from the first:", bad)
