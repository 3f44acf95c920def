#!/usr/bin/env python3
"""The materialised outputs of SURVEY 8 rows a-1 / a-5 against their bound (write bandwidth): calc_hammingDist as float32 [Q, R] and as
int16, calc_label_sim as float32, Q = 2000 x R = 117 218 (0.94 GB / 0.47 GB / 0.94 GB written per call).  GPU box.
    python tools/bench_dist.py"""
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

g = torch.Generator().manual_seed(3)
Q, R, C = 2000, 117218, 80
for K in (16, 64, 256):
    qB = torch.randn(Q, K, generator=g).sign(); rB = torch.randn(R, K, generator=g).sign()
    q, r = X.pack_sign(qB.cuda()), X.pack_sign(rB.cuda())
    q, r = X.PackedCodes(q.bits, None, K), X.PackedCodes(r.bits, None, K)      # pure +-1 codes: no zero plane
    t = timed(lambda: X.hamming_dist(q, r))
    print("hamming_dist f32  K %4d: %7.1f us  %5.2f TB/s written" % (K, t * 1e6, Q * R * 4 / t / 1e12))
    t = timed(lambda: X.hamming_dist(q, r, as_u16=True))
    print("hamming_dist u16  K %4d: %7.1f us  %5.2f TB/s written" % (K, t * 1e6, Q * R * 2 / t / 1e12))
qL = (torch.rand(Q, C, generator=g) < 0.05).long(); rL = (torch.rand(R, C, generator=g) < 0.05).long()
ql, rl = X.pack_labels(qL.cuda()), X.pack_labels(rL.cuda())
t = timed(lambda: X.label_sim(ql, rl, C))
print("label_sim f32 C 80      : %7.1f us  %5.2f TB/s written" % (t * 1e6, Q * R * 4 / t / 1e12))
x = torch.empty(Q, R, dtype=torch.float32, device="cuda")
t = timed(lambda: x.fill_(1.0))
print("torch fill_ of the same : %7.1f us  %5.2f TB/s written" % (t * 1e6, Q * R * 4 / t / 1e12))
