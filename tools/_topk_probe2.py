import sys, os, json
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,"clip-based-cross-modal-hash_amd")]
import bench_topk as B
for K in (128, 256, 512):
    for Q in (1, 3, 4, 5, 8, 12, 16):
        r=B.measure(K=K, Q=Q)
        print("K",K,"Q",Q, B.filter_instance(K,Q), round(r["avg_launch_ms"]*1e3,1), round(r["whole_call_ms"]*1e3,1), flush=True)
