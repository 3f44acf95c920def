#!/usr/bin/env python3
"""HBM-bound regime of the retrieval path: exact top-k of a handful of queries over a 10 M x 256-bit gallery
(BASELINE configs[4] shape, the whole gallery on ONE GPU: 320 MB of codes > 256 MB Infinity Cache).

Called by bench.py (``measure()``) and runnable on its own:  python bench_topk.py [--R 10000000 --K 256 --Q 8 --k 100]
Prints / returns the `roofline` object of k_topk_stream: algorithmic bytes = R*K/8 (gallery read once)
+ Q*K/8 + partial lists written, divided by the HIP-event time of the launch bracket.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0


def _traffic():
    try:
        import bench
        t = bench.pmc_traffic("k_topk_filter")
        return None if t is None else t["bytes"]
    except Exception:
        return None


def measure(R=10_000_000, K=256, Q=8, k=100, iters=20, warmup=3):
    from xmh import retrieval as X
    from xmh._lib import lib
    W = (K + 31) // 32
    g = torch.Generator(device="cuda").manual_seed(1814)
    rb = torch.randint(-2**31, 2**31 - 1, (R, W), dtype=torch.int32, device="cuda", generator=g)
    qb = torch.randint(-2**31, 2**31 - 1, (Q, W), dtype=torch.int32, device="cuda", generator=g)
    q, r = X.PackedCodes(qb, None, K), X.PackedCodes(rb, None, K)
    from xmh import _lib
    for _ in range(warmup):
        d, i = X.hamming_topk(q, r, k)
    torch.cuda.synchronize()
    _lib.prof_enable(True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for n in range(iters):
        d, i = X.hamming_topk(q, r, k)
        ev[n + 1].record()
    torch.cuda.synchronize()
    t_call = sum(ev[n].elapsed_time(ev[n + 1]) for n in range(iters)) / iters * 1e-3
    t, launches = _lib.prof_read("topk_filter")
    t *= 1e-3
    _lib.prof_enable(False)
    alg = R * W * 4 + Q * W * 4 + Q * 4                 # gallery read once + queries + thresholds
    return {"kernel": "k_topk_filter (streaming pass of xmh_hamming_topk), HIP events around the launch, %d launches" % launches,
            "bound": "hbm", "achieved": alg / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / t / 1e9 / HBM_PEAK_GBS,
            "traffic": _traffic(), "algorithmic_bytes": alg, "avg_launch_ms": t * 1e3,
            "whole_call_ms": t_call * 1e3, "whole_call_GBps": alg / t_call / 1e9,
            "workload": "exact top-%d of Q=%d queries over R=%d x %d-bit gallery on one GPU" % (k, Q, R, K),
            "pairs_per_s_whole_call": Q * R / t_call}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--R", type=int, default=10_000_000)
    ap.add_argument("--K", type=int, default=256)
    ap.add_argument("--Q", type=int, default=8)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    print(json.dumps(measure(a.R, a.K, a.Q, a.k, a.iters)))
