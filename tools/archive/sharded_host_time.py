#!/usr/bin/env python3
"""host time to enqueue one sharded mAP step (RCCL, world size 1) against its GPU time -- run on the GPU box"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import bench
from xmh import retrieval as R, sharded
Q, Rn, K, C = 5000, 117218, 64, 80
qB, qL, _, _ = bench.synth(Q, 8, K, C, seed=1814, p=0.04)
_, _, rB, rL = bench.synth(8, Rn, K, C, seed=1815, p=0.04)
q, ql, r, rl = R.pack_sign(qB.cuda()), R.pack_labels(qL.cuda()), R.pack_sign(rB.cuda()), R.pack_labels(rL.cuda())
ops = sharded.HipShardOps(q, ql, r, rl, C)
for _ in range(5):
    sharded.map_k_sharded(ops, None)
torch.cuda.synchronize()
n = 100
t0 = time.perf_counter()
for _ in range(n):
    sharded.map_k_sharded(ops, None)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.3f ms/step, wall %.3f ms/step" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(50):
    sharded.map_k_sharded(ops, None)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
dist.destroy_process_group()
