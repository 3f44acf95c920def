#!/bin/bash
# top-k filter: queries per group x groups per block (XMH_TOPK_QG), 10 M x 256 bit; run on the GPU box
for Q in 8 16 64; do
  for v in 8x1 4x2 2x4 8x2 8x4 4x4; do
    XMH_TOPK_QG=$v python - <<PY
import sys; sys.path[:0]=['.','clip-based-cross-modal-hash_amd']
import bench_topk as B
r=B.measure(Q=$Q, iters=10)
print("Q=$Q QG=$v filter %.1f us call %.1f us frac %.3f call_frac %.3f" % (r["avg_launch_ms"]*1e3, r["whole_call_ms"]*1e3, r["frac"], r["whole_call_GBps"]/8000))
PY
  done
done
