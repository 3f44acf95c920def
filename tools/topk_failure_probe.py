#!/usr/bin/env python3
"""How often the fast path of xmh_hamming_topk hands over to the robust path on large galleries of short codes (coarse distance buckets:
one bucket more is 4-5 x the candidates), and what the call costs: whole-call time over `iters` calls on fresh random queries, and the
fraction of calls whose robust kernels really ran (library event scope "topk_robust").  GPU box.
    python tools/topk_failure_probe.py [R=40000000]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
import torch
from xmh import retrieval as X
R = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
g = torch.Generator(device="cuda").manual_seed(5)
for K in (16, 32, 64, 256):
    W = (K + 31) // 32
    rb = torch.randint(-2**31, 2**31 - 1, (R, W), dtype=torch.int32, device="cuda", generator=g)
    if K % 32:
        rb &= (1 << (K % 32)) - 1
    r = X.PackedCodes(rb, None, K)
    for Q in (1, 4, 16, 64):
        times = []
        for it in range(12):
            qb = torch.randint(-2**31, 2**31 - 1, (Q, W), dtype=torch.int32, device="cuda", generator=g)
            if K % 32:
                qb &= (1 << (K % 32)) - 1
            q = X.PackedCodes(qb, None, K)
            ws = X.TopkWorkspace(Q, R, K, 100, "cuda")
            X.hamming_topk(q, r, 100, workspace=ws)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); X.hamming_topk(q, r, 100, workspace=ws); e1.record(); torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) * 1e3)
        times.sort()
        print("K %4d R %d Q %3d: call us min %8.1f median %8.1f max %8.1f" % (K, R, Q, times[0], times[len(times) // 2], times[-1]), flush=True)
    del r, rb
    torch.cuda.empty_cache()
