import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,"clip-based-cross-modal-hash_amd")]
import torch, bench_topk as B
from xmh import retrieval as X
Q=int(sys.argv[1]) if len(sys.argv)>1 else 1
q, r = B._codes("iid", 10_000_000, 256, Q)
ws = X.TopkWorkspace(q.n, r.n, 256, 100, "cuda")
for _ in range(5): X.hamming_topk(q, r, 100, workspace=ws)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): X.hamming_topk(q, r, 100, workspace=ws)
e1.record(); torch.cuda.synchronize()
print("Q", Q, "whole call us", e0.elapsed_time(e1)/200*1e3)
