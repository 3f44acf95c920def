cd $GRAFT_REPO_ROOT
timeout 600 python - <<'PY'
import sys, os, json
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "clip-based-cross-modal-hash_amd")]
import bench_topk
for Q in (1, 2, 4, 8):
    m = bench_topk.measure(Q=Q, robust=False)
    print("Q=%d whole call %.1f us = %.0f GB/s (%.3f of 8 TB/s); filter alone %.1f us frac %.3f; fused launch %s us" % (
        Q, m["whole_call_ms"] * 1e3, m["whole_call_GBps"], m["whole_call_GBps"] / 8000, m["avg_launch_ms"] * 1e3, m["frac"],
        ("%.1f" % (m["fused_launch_ms"] * 1e3)) if "fused_launch_ms" in m else "-"), flush=True)
for K, R in ((64, 40_000_000), (32, 80_000_000), (128, 20_000_000), (512, 5_000_000)):
    m = bench_topk.measure(R=R, K=K, Q=1, robust=False)
    print("K=%d R=%d Q=1 whole call %.1f us = %.0f GB/s; filter alone %.1f us frac %.3f" % (K, R, m["whole_call_ms"] * 1e3, m["whole_call_GBps"], m["avg_launch_ms"] * 1e3, m["frac"]), flush=True)
PY
