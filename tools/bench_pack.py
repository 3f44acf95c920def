#!/usr/bin/env python3
"""pack / unpack against their bound: pack_sign reads 4 n K bytes of float codes, unpack writes them.  GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
import torch
from xmh import retrieval as X

def timed(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3

for n, K in ((4_000_000, 16), (4_000_000, 64), (2_000_000, 256), (117_218, 64)):
    x = torch.randn(n, K, device="cuda").sign()
    t = timed(lambda: X.pack_sign(x))
    p = X.pack_sign(x)
    print("pack_sign   n %8d K %4d: %8.1f us  %5.2f TB/s read" % (n, K, t * 1e6, n * K * 4 / t / 1e12))
    t = timed(lambda: p.unpack())
    print("unpack_pm1  n %8d K %4d: %8.1f us  %5.2f TB/s written" % (n, K, t * 1e6, n * K * 4 / t / 1e12))
    L = (torch.rand(n, 80, device="cuda") < 0.05).long()
    t = timed(lambda: X.pack_labels(L))
    print("pack_labels n %8d C   80: %8.1f us  %5.2f TB/s read (int64 labels)" % (n, t * 1e6, n * 80 * 8 / t / 1e12))
    del x, p, L
    torch.cuda.empty_cache()
