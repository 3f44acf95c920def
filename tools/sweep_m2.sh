#!/bin/bash
# sweep of k_scan_hist_m2 shapes (XMH_SCAN_M2_GEOM: 0 = 4 waves x 2 groups, 1 = 8x1, 2 = 4x4, 3 = 8x2) and chunk rounds; run on the GPU box
mkdir -p gpurun_out
run() {
  env "$@" timeout 300 python bench.py --steps 40 --no-cpu-baseline --no-hbm-regime --no-encode --no-extra-configs $EXTRA 2>>gpurun_out/sweep_m2.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$*', 'ms/step %.4f' % d['ms_per_step'], 'mAP %.8f' % d.get('mAP'), 'pass1 %.4f' % r.get('pass1_avg_launch_ms'), 'pass2 %.4f' % r.get('pass2_avg_launch_ms'))"
}
for g in 0 4 5 6; do for r in 2 3; do run XMH_SCAN_M2_GEOM=$g XMH_SCAN_M2_ROUNDS=$r; done; done
run XMH_SCAN_M2_GEOM=0 XMH_SCAN_CACHE_MB=0
run XMH_SCAN_M2_GEOM=1 XMH_SCAN_CACHE_MB=0
run XMH_SCAN_M2=0
