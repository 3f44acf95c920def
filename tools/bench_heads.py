#!/usr/bin/env python3
"""encode throughput per method head (DCMHT / DSPH / MITH, 64-bit, B=100, parity mode) -- run on the GPU box"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
import torch
import xmh.models  # noqa: F401
from xmh.common.register import registry
from xmh.utils.config import Config
from xmh.models import weights as W

B = 100
image = W.synth_images(5, B).cuda()
ids, _ = W.synth_text(5, B)
ids = ids.cuda()
kpm = ids == 0
for arch in ("DCMHT", "DSPH", "MITH"):
    model = registry.get_model_class(arch).from_config(Config({"clip_path": "synthetic:1814"}), output_dim=64, train_num=1000).cuda().eval()
    fns = {"image": lambda: model.encode_image(image),
           "text": (lambda: model.encode_text(ids, kpm)) if arch == "MITH" else (lambda: model.encode_text(ids))}
    for name, fn in fns.items():
        with torch.no_grad():
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print("%-5s %-5s %7.3f ms/batch  %8.0f /s" % (arch, name, dt * 1e3, B / dt))
