import sys, os, json
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,"clip-based-cross-modal-hash_amd")]
import bench_topk as B
print(B.filter_instance(256,1))
for Q in (1,2,4):
    r=B.measure(Q=Q)
    print("Q",Q, r["avg_launch_ms"], r["whole_call_ms"])
r=B.measure_cache_defeat()
print({k:(v["filter_ms"], v["whole_call_ms"]) for k,v in r["Q1"].items() if isinstance(v,dict)})
for K in (128,512,1024):
    r=B.measure(K=K,Q=1,R=10_000_000 if K<=512 else 4_000_000)
    print("K",K, r["avg_launch_ms"], r["whole_call_ms"])
