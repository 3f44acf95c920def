#!/bin/bash
# A/B of the scan step: k_scan_hist_m2 (default) against k_scan_hist_m (XMH_SCAN_M2=0); run on the GPU box
mkdir -p gpurun_out
for v in 1 0; do
  echo "== XMH_SCAN_M2=$v"
  XMH_SCAN_M2=$v timeout 300 python bench.py --steps 50 --no-cpu-baseline --no-hbm-regime --no-encode --no-extra-configs "$@" 2>>gpurun_out/ab_m2.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step', d['ms_per_step'], 'mAP', d.get('mAP'), 'pass1', r.get('pass1_avg_launch_ms'), 'pass2', r.get('pass2_avg_launch_ms'))"
done
