#!/usr/bin/env python3
"""top-k filter over query counts and code lengths: the matrix-core filter (default rule) against the VALU filter
(XMH_TOPK_MFMA=0), run on the GPU box:  for v in -1 0; do XMH_TOPK_MFMA=$v python tools/sweep_topk_mfma.py; done
(-1 = the library's own rule: 3 and >= 5 queries at 128 / 256 / 512 bits)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
if os.environ.get("XMH_TOPK_MFMA") == "-1":
    del os.environ["XMH_TOPK_MFMA"]
import bench_topk

QS = tuple(int(x) for x in sys.argv[1].split(",")) if len(sys.argv) > 1 else (1, 2, 3, 4, 6, 8, 16, 32, 64, 512)
for K, R in ((256, 10_000_000), (128, 10_000_000), (512, 4_000_000)):
    for Q in QS:
        m = bench_topk.measure(R=R, K=K, Q=Q, iters=5, warmup=2)
        print("K=%4d R=%8d Q=%3d  filter %.4f ms  whole %.4f ms  %.3e pairs/s robust %d" % (
            K, R, Q, m["avg_launch_ms"], m["whole_call_ms"], m["pairs_per_s_whole_call"], m["robust_path_launches"]["launches"]), flush=True)
