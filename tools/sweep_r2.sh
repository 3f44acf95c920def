#!/bin/bash
# k_scan_hist_r2 (XMH_SCAN_M2_REGS=1) against k_scan_hist_m2 at the headline shape: geometry x blocks per CU x rounds (GPU box)
cd "${GRAFT_REPO_ROOT:-.}"
cat > /tmp/leg.py <<PY
import sys, os
sys.path[:0] = [".", "clip-based-cross-modal-hash_amd"]
import bench_roofline as RL
K = int(os.environ.get("KBITS", "64"))
o = RL.extra_scan_leg("shape", Q=5000, Rn=117218, K=K, C=80, p_label=0.04, seed=1814, steps=100)
print("ms %.4f p1 %.4f p2 %.4f mAP %.8f" % (o["ms_per_step"], o["pass1_ms"], o["pass2_ms"], o["mAP"]))
PY
echo -n "m2 default: "; python /tmp/leg.py 2>&1 | tail -1
for g in ${GEOMS:-0 2}; do for b in ${BPCS:-2 3 4}; do for r in ${ROUNDS:-1 2}; do
echo -n "r2 geom=$g bpc=$b rounds=$r: "; XMH_SCAN_M2_REGS=1 XMH_SCAN_M2_GEOM=$g XMH_SCAN_M2_BPC=$b XMH_SCAN_M2_ROUNDS=$r python /tmp/leg.py 2>&1 | tail -1
done; done; done
