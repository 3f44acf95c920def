#!/bin/bash
# sweep of k_scan_hist_m2 shapes (XMH_SCAN_M2_GEOM: see m2_geom() in xmh_scan.hip) and chunk rounds; run on the GPU box.  usage: tools/sweep_m2.sh "0 2 3" "2 3" [bench args]
mkdir -p gpurun_out
GEOMS=${1:-"0 1 2 3"}; ROUNDS=${2:-"2 3"}; shift 2
run() {
  env "$@" timeout 300 python bench.py --steps 40 --no-cpu-baseline --no-hbm-regime --no-encode --no-extra-configs $EXTRA 2>>gpurun_out/sweep_m2.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$*', 'ms/step %.4f' % d['ms_per_step'], 'mAP %.8f' % d.get('mAP'), 'pass1 %.4f' % r.get('pass1_avg_launch_ms'), 'pass2 %.4f' % r.get('pass2_avg_launch_ms'))"
}
for g in $GEOMS; do for r in $ROUNDS; do run XMH_SCAN_M2_GEOM=$g XMH_SCAN_M2_ROUNDS=$r; done; done
