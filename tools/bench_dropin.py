#!/usr/bin/env python3
"""The drop-in calc_map_k handed DEVICE float codes and int64 labels -- what the reference's own valid() passes (runners/base.py:259-264:
the code buffers live on self.device) -- per call at the COCO shape: pack both code matrices + (cached) label packing + both scan passes
+ the D2H of the scalar.  GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
import torch
from xmh.common import calc_utils as cu
g = torch.Generator().manual_seed(1)
for K in (16, 64, 128):
    Q, R, C = 5000, 117218, 80
    qB = torch.randn(Q, K, generator=g).sign(); rB = torch.randn(R, K, generator=g).sign()
    qB[qB == 0] = 1; rB[rB == 0] = 1          # (one exact zero among the 7.5 M elements selects the ternary kernels: timed below)
    qB, rB = qB.cuda(), rB.cuda()
    qL = (torch.rand(Q, C, generator=g) < 0.05).long(); rL = (torch.rand(R, C, generator=g) < 0.05).long()
    qL[:, 0] = 1; rL[::3, 0] = 1
    qL, rL = qL.cuda(), rL.cuda()
    for _ in range(3): m = cu.calc_map_k(qB, rB, qL, rL)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): m = cu.calc_map_k(qB, rB, qL, rL)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("calc_map_k on device tensors, K %3d: %7.1f us per call (%.3g pairs/s), mAP %.6f" % (K, dt * 1e6, Q * R / dt, float(m)))
    rZ = rB.clone(); rZ[777, 3] = 0.0          # sign(0) = 0 (reference runners/base.py:407-410): ONE exact zero
    for _ in range(3): m = cu.calc_map_k(qB, rZ, qL, rL)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): m = cu.calc_map_k(qB, rZ, qL, rL)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("   ... with one exact 0.0 in the gallery codes:  %7.1f us per call (%.3g pairs/s), mAP %.6f" % (dt * 1e6, Q * R / dt, float(m)))
