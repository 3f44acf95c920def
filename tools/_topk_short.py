import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,"clip-based-cross-modal-hash_amd")]
import bench_topk as B
for K,R in ((256, 10_000_000), (64, 40_000_000), (32, 40_000_000), (128, 10_000_000)):
    for Q in (1, 4, 8, 64):
        r=B.measure(K=K, Q=Q, R=R, iters=30)
        print("K",K,"R",R,"Q",Q, B.filter_instance(K,Q,R), "filter us", round(r["avg_launch_ms"]*1e3,1), "GB/s", round(r["achieved"]), "call us", round(r["whole_call_ms"]*1e3,1), flush=True)
