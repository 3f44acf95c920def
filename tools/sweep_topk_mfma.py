import os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "clip-based-cross-modal-hash_amd")]
import bench_topk
for K, R in ((256, 10_000_000),):
    for Q in (8, 16, 32, 64):
        m = bench_topk.measure(R=R, K=K, Q=Q, iters=5, warmup=2)
        print("K=%4d R=%8d Q=%3d  filter %.4f ms  whole %.4f ms  %.3e pairs/s robust %d" % (K, R, Q, m["avg_launch_ms"], m["whole_call_ms"], m["pairs_per_s_whole_call"], m["robust_path_launches"]["launches"]), flush=True)
