import os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "clip-based-cross-modal-hash_amd")]
import bench_topk
for kind in ("iid", "structured", "duplicates"):
    for Q in (8, 64):
        m = bench_topk.measure(R=10_000_000, K=256, Q=Q, iters=10, warmup=3, kind=kind)
        print(kind, Q, "filter %.4f ms call %.4f" % (m["avg_launch_ms"], m["whole_call_ms"]), m["robust_path_launches"])
