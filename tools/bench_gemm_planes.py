#!/usr/bin/env python3
"""fp16-plane GEMM kernel alone (xmh_gemm_nt_h16: A already fp16), back-to-back launches vs launches separated by a pass that
rewrites A (what a forward does): python tools/bench_gemm_planes.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
import torch
from xmh import ops
from xmh._lib import lib, ptr, check, current_stream

shapes = [(5000, 2304, 768), (5000, 768, 768), (5000, 3072, 768), (5000, 768, 3072), (20000, 2304, 768), (4096, 4096, 4096)]
for M, N, K in shapes:
    A = (torch.rand(M, K, device="cuda") * 2 - 1)
    Ah = A.half()
    Wh = (torch.rand(N, K, device="cuda") * 2 - 1).half()
    b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    def gemm():
        check(lib.xmh_gemm_nt_h16(ptr(Ah), K, ptr(Wh), K, ptr(b), None, 0, ptr(out), N, M, N, K, 0, current_stream()), "h16")
    def timed(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    t_b2b = timed(gemm)
    def cast(): check(lib.xmh_cast_f32_to_f16(ptr(A), ptr(Ah), A.numel(), current_stream()), "cast")
    t_cast = timed(cast)
    def both(): cast(); gemm()
    t_both = timed(both)
    print("M=%5d N=%5d K=%5d  back-to-back %7.1f us (%6.1f TF)   after a pass that rewrites A %7.1f us (%6.1f TF)   [cast alone %5.1f us]" %
          (M, N, K, t_b2b, 2.0 * M * N * K / t_b2b / 1e6, t_both - t_cast, 2.0 * M * N * K / (t_both - t_cast) / 1e6, t_cast))
