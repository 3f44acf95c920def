#!/usr/bin/env python3
"""top-k filter sweep over Q and code length (run on the GPU box): python tools/bench_topk_sweep.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
import bench_topk

for K, R in ((256, 10_000_000), (64, 10_000_000), (128, 10_000_000), (512, 4_000_000), (1024, 2_000_000)):
    for Q in (1, 8, 64, 512):
        if K != 256 and Q in (512,):
            continue
        m = bench_topk.measure(R=R, K=K, Q=Q)
        print("K=%4d R=%8d Q=%3d  filter %.4f ms  %.0f GB/s  whole %.4f ms  %.3e pairs/s" % (
            K, R, Q, m["avg_launch_ms"], m["achieved"], m["whole_call_ms"], m["pairs_per_s_whole_call"]), flush=True)
