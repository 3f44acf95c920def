#!/usr/bin/env python3
"""cProfile of one BaseTrainer.valid() at the configs[1] shape (GPU box): where the host time of the call goes besides the encode loop"""
import cProfile, pstats, sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
import torch
import bench_valid as BV
from xmh.runners.base import BaseTrainer
orig = BaseTrainer.valid
def profiled(self, *a, **k):
    pr = cProfile.Profile(); pr.enable()
    r = orig(self, *a, **k)
    torch.cuda.synchronize(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
    return r
BaseTrainer.valid = profiled
o = BV.measure()
print("valid", round(o["valid_seconds"], 3), "encode", round(o["encode_seconds"], 3))
