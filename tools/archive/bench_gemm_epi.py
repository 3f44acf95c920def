#!/usr/bin/env python3
"""plane GEMM with the epilogues the encoder uses (bias; bias + in-place residual), fast and parity: python tools/bench_gemm_epi.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
import torch
from xmh import ops, _lib
from xmh._lib import lib, ptr, check, current_stream
shapes = [(5000, 2304, 768, False), (5000, 768, 768, True), (5000, 3072, 768, False), (5000, 768, 3072, True), (3200, 1536, 512, False), (3200, 512, 512, True), (3200, 512, 2048, True), (20000, 2304, 768, False), (20000, 768, 3072, True)]
for M, N, K, res in shapes:
    A = (torch.rand(M, K, device="cuda") * 2 - 1)
    W = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.05).half().float()
    b = torch.randn(N, device="cuda")
    x = torch.randn(M, N, device="cuda")
    line = "M=%5d N=%5d K=%5d %s" % (M, N, K, "+res" if res else "    ")
    for mode, slot in (("f16", "gemm_f16"), ("f32", "gemm_s16")):
        ops.set_precision(mode)
        def fn():
            if res: ops.gemm_nt(A, W, b, residual=x, out=x)
            else: ops.gemm_nt(A, W, b, out=x)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        _lib.prof_enable(True)
        for _ in range(20): fn()
        torch.cuda.synchronize()
        ms, n = _lib.prof_read(slot)
        _lib.prof_enable(False)
        line += "   %s %6.1f us (%5.0f TF)" % (mode, ms * 1e3, 2.0 * M * N * K / ms / 1e9)
    print(line)
ops.set_precision("f32")
