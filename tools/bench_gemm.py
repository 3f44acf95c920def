#!/usr/bin/env python3
"""per-shape GEMM throughput and error of the three precisions (run on the GPU box): python tools/bench_gemm.py
f32x = exact fp32 MFMA, f32 = hi/lo split on fp16-exact weights (parity mode), f16 = fast mode.
Error = max |C - C64| / max |C64| against a float64 product of the same operands."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
import torch
from xmh import ops

shapes = [(5000, 2304, 768), (5000, 768, 768), (5000, 3072, 768), (5000, 768, 3072), (3200, 1536, 512), (3200, 512, 512),
          (3200, 2048, 512), (3200, 512, 2048), (100, 512, 768), (4096, 4096, 4096),
          (20000, 2304, 768), (20000, 768, 768), (20000, 3072, 768), (20000, 768, 3072)]      # fused evaluation batches (4 x 100 images)
for name in ("f32x", "f32", "f32w", "f16"):            # f32w = parity mode on weights that are not fp16-exact (3 MFMAs per product)
    prec = ops._NAMES["f32" if name == "f32w" else name]
    for M, N, K in shapes:
        A = torch.randn(M, K, device="cuda") * 3.0
        W = torch.randn(N, K, device="cuda") * 0.05
        if name != "f32w":
            W = W.half().float()                                                # fp16-exact like CLIP weights
        b = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda")
        for _ in range(3):
            ops.gemm_nt(A, W, b, act=0, out=out, precision=prec)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            ops.gemm_nt(A, W, b, act=0, out=out, precision=prec)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        rows = slice(0, min(M, 512))
        ref = A[rows].double() @ W.double().t() + b.double()
        err = float((out[rows].double() - ref).abs().max() / ref.abs().max())
        print("%-4s M=%5d N=%5d K=%5d  %8.3f ms  %8.1f TFLOP/s   max rel err %.2e" % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9, err))
