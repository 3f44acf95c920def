#!/bin/bash
# A/B of the matrix-core top-k filter (XMH_TOPK_MFMA = smallest query count that takes it, 0 = never); run on the GPU box
mkdir -p gpurun_out
for v in ${1:--1 0}; do
  echo "== XMH_TOPK_MFMA=$v"
  XMH_TOPK_MFMA=$v timeout 600 python bench.py --steps 5 --no-cpu-baseline --no-encode --no-extra-configs 2>>gpurun_out/ab_topk.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k in ('roofline_hbm_regime','roofline_hbm_regime_q8','roofline_hbm_regime_q64'):
    r=d[k]; print(k, 'filter %.4f ms call %.4f ms' % (r['avg_launch_ms'], r['whole_call_ms']), 'robust launches', r['robust_path_launches']['launches'], 'pairs/s %.3e' % r['pairs_per_s_whole_call'])
t=d.get('topk_structured_codes'); print(json.dumps(t)[:600])"
done
