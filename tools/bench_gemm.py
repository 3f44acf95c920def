#!/usr/bin/env python3
"""What is attainable on the encoder's GEMM shapes: k_gemm_g16 (this repo, fast = one fp16 plane per operand, parity = hi/lo split
activations x fp16 weights: two MFMAs per product) beside torch.matmul on fp16 operands -- hipBLASLt / Tensile underneath, the vendor's
tuned kernels.  torch.matmul is a YARDSTICK here and nothing else: no product path calls it.

The eight Linear shapes of CLIP ViT-B/32 (models/CLIP/model.py:167-197): vision width 768 (qkv 2304, out 768, c_fc 3072, c_proj K = 3072)
and text width 512 (1536, 512, 2048, K = 2048), at M = 5000 (100 images x 50 tokens) and M = 20000 (batch 400).  Kernel time by HIP
events over `iters` back-to-back launches on uniform random operands (zeros run ~35 % faster on this chip: power).

    python tools/bench_gemm.py [--iters 50] > profiles/r05_gemm_yardstick.txt
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]

import torch  # noqa: E402

SHAPES = [("vit qkv", 2304, 768), ("vit out", 768, 768), ("vit c_fc", 3072, 768), ("vit c_proj", 768, 3072),
          ("text qkv", 1536, 512), ("text out", 512, 512), ("text c_fc", 2048, 512), ("text c_proj", 512, 2048)]


def timed(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    from xmh import _lib
    from xmh._lib import check, current_stream, lib, ptr
    g = torch.Generator(device="cuda").manual_seed(7)
    print("GEMM yardstick on %s: useful TFLOP/s = 2 M N K / kernel time (the parity kernel issues twice that on the MFMA)" % torch.cuda.get_device_name(0))
    print("%-12s %6s %5s %5s | %22s | %22s | %22s | %s" % ("layer", "M", "N", "K", "torch.matmul fp16 (lib)", "k_gemm_g16 fast", "k_gemm_g16 parity", "fast / lib"))
    for M in (5000, 20000):
        for name, N, K in SHAPES:
            A = (torch.rand(M, K, device="cuda", generator=g) - 0.5)
            W = (torch.rand(N, K, device="cuda", generator=g) - 0.5) * 0.1
            bias = torch.zeros(N, device="cuda")
            Ah, Wh = A.half().contiguous(), W.half().contiguous()
            Wt = Wh.t()
            flops = 2.0 * M * N * K
            t_lib = timed(lambda: torch.matmul(Ah, Wt), args.iters)                      # fp16 in, fp16 out, fp32 accumulate
            C = torch.empty(M, N, device="cuda")

            def fast():
                check(lib.xmh_gemm_nt_h16(ptr(Ah), K, ptr(Wh), K, ptr(bias), None, 0, ptr(C), N, M, N, K, 0, current_stream()), "xmh_gemm_nt_h16")
            t_fast = timed(fast, args.iters)
            # parity: the split of A into planes is its own launch in this entry point (in the forward the producer writes the planes):
            # time the GEMM kernel alone through the library's event scopes
            Wf = Wh.float().contiguous()

            def parity():
                check(lib.xmh_gemm_nt_split16(ptr(A), K, ptr(Wh), None, K, ptr(bias), None, 0, ptr(C), N, M, N, K, 0, current_stream()), "xmh_gemm_nt_split16")
            for _ in range(3):
                parity()
            torch.cuda.synchronize()
            _lib.prof_enable(True)
            for _ in range(args.iters):
                parity()
            torch.cuda.synchronize()
            t_ms, n = _lib.prof_read("gemm_s16")                                   # mean launch ms
            _lib.prof_enable(False)
            t_par = t_ms * 1e-3
            ref = (Ah.float() @ Wf.t())
            err = float((C - (A @ Wf.t())).abs().max() / ref.abs().max())
            assert err < 1e-5, (name, err)
            del Wf
            print("%-12s %6d %5d %5d | %8.1f us %8.0f TF | %8.1f us %8.0f TF | %8.1f us %8.0f TF | %.2f" % (
                name, M, N, K, t_lib * 1e6, flops / t_lib / 1e12, t_fast * 1e6, flops / t_fast / 1e12, t_par * 1e6, flops / t_par / 1e12, t_lib / t_fast))


if __name__ == "__main__":
    main()
