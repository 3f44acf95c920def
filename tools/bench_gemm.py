#!/usr/bin/env python3
"""per-shape GEMM throughput (run on the GPU box): python tools/bench_gemm.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
import torch
from xmh import ops

shapes = [(5000, 2304, 768), (5000, 768, 768), (5000, 3072, 768), (5000, 768, 3072), (4900, 768, 3072), (3200, 1536, 512), (3200, 512, 512),
          (3200, 2048, 512), (3200, 512, 2048), (100, 512, 768), (8192, 8192, 8192)]
for prec, name in ((0, "f32"), (1, "f16")):
    for M, N, K in shapes:
        if prec == 0 and M == 8192:
            M = N = K = 4096
        A = torch.randn(M, K, device="cuda")
        W = torch.randn(N, K, device="cuda") * 0.05
        b = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda")
        for _ in range(3):
            ops.gemm_nt(A, W, b, act=1, out=out, precision=prec)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            ops.gemm_nt(A, W, b, act=1, out=out, precision=prec)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print("%s M=%5d N=%5d K=%5d  %8.3f ms  %8.1f TFLOP/s" % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
