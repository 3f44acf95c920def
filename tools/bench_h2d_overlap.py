#!/usr/bin/env python3
"""encode loop fed from pinned host batches: with and without the separate H2D stream (run on the GPU box).
    python tools/bench_h2d_overlap.py ; XMH_NO_COPY_STREAM=1 python tools/bench_h2d_overlap.py"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
import torch
import xmh.models, xmh.runners  # noqa: F401
from xmh.common.register import registry
from xmh.utils.config import Config
from xmh.models import weights as W

tmp = tempfile.mkdtemp()
cfg = Config({"model": {"arch": "DCMHT", "clip_path": "synthetic:1814"},
              "dataset": {"arch": "synthetic", "name": "synth", "num_classes": 24, "retrieval_num": 64, "max_word": 32, "image_resolution": 224},
              "run": {"arch": "DCMHTTrainer", "output_dim": 64, "device": 0, "batch_size": 32, "num_workers": 0, "is_train": False, "query_num": 32,
                      "train_num": 32, "save_dir": tmp, "log_dir": tmp, "seed": 1814}})
tr = registry.get_runner_class("DCMHTTrainer").from_config(cfg=cfg, autorun=False)
B, NB = 100, 16
ids, _ = W.synth_text(5, B)
batches = []
for b in range(NB):
    img = W.synth_images(b, B).pin_memory()
    batches.append((img, ids.clone().pin_memory(), None, torch.zeros(B, 24, dtype=torch.int64), torch.arange(b * B, (b + 1) * B)))
tr._shard = lambda n: (0, n)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.encode_streams(batches, B * NB)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("copy stream %s: %d images + captions in %.1f ms -> %.0f pairs/s" % ("OFF" if os.environ.get("XMH_NO_COPY_STREAM") else "ON", B * NB, dt * 1e3, B * NB / dt))
