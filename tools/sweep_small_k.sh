#!/bin/bash
# plan sweep (k_scan_hist_r2 geometry x blocks per CU x rounds) on the small / short-code legs of the bench (GPU box)
cd "${GRAFT_REPO_ROOT:-.}"
cat > /tmp/leg.py <<PY
import sys
sys.path[:0] = [".", "clip-based-cross-modal-hash_amd"]
import bench_roofline as RL
o = RL.extra_scan_leg(**dict(RL.EXTRA_LEGS[sys.argv[1]], steps=200))
print(sys.argv[1], "ms %.4f p1 %.4f p2 %.4f" % (o["ms_per_step"], o["pass1_ms"], o["pass2_ms"]), o["pass1_kernel"][:40])
PY
for leg in ${LEGS:-configs0_dcmht_16bit_mirflickr}; do
echo -n "default: "; python /tmp/leg.py $leg 2>&1 | tail -1
for g in ${GEOMS:-2 0}; do for r in ${ROUNDS:-1 2}; do for b in ${BPCS:-1 2}; do
echo -n "geom=$g rounds=$r bpc=$b: "; XMH_SCAN_M2_GEOM=$g XMH_SCAN_M2_ROUNDS=$r XMH_SCAN_M2_BPC=$b python /tmp/leg.py $leg 2>&1 | tail -1
done; done; done; done
