#!/bin/bash
# sweep gallery-access mode x rounds for the fused mAP scan (run on the GPU box)
for r in 2 4 6 8; do for ga in 0 1; do
  echo -n "rounds=$r gm_ap=$ga : "
  XMH_SCAN_ROUNDS=$r XMH_SCAN_GM_AP=$ga python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-hbm-regime --no-encode 2>/dev/null \
   | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ms/step %.3f  pass1 %.3f  pass2 %.3f  mAP %.6f' % (d['ms_per_step'], d['roofline']['pass1_avg_launch_ms'], d['roofline']['avg_launch_ms'], d['mAP']))"
done; done
