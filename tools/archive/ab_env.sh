#!/bin/bash
# A/B of the scan step under one environment switch; run on the GPU box.  usage: tools/ab_env.sh VAR "values" [bench args]
mkdir -p gpurun_out
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  echo "== $VAR=$v $*"
  env $VAR=$v timeout 300 python bench.py --steps 50 --no-cpu-baseline --no-hbm-regime --no-encode --no-extra-configs "$@" 2>>gpurun_out/ab_env.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step %.4f' % d['ms_per_step'], 'mAP', d.get('mAP'), 'pass1 %.4f' % r.get('pass1_avg_launch_ms'), 'pass2 %.4f' % r.get('pass2_avg_launch_ms'))"
done
