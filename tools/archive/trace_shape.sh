#!/bin/bash
# per-kernel times of the mAP scan at an arbitrary shape (GPU box):  bash tools/trace_shape.sh Q R K C [p_label]   (environment switches pass through)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=/tmp/trace_shape; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/shape.py <<PY
import sys
sys.path[:0] = [".", "clip-based-cross-modal-hash_amd"]
import bench_roofline as RL
o = RL.extra_scan_leg("shape", Q=$1, Rn=$2, K=$3, C=$4, p_label=${5:-0.04}, seed=1814, steps=200)
print({k: o[k] for k in ("ms_per_step", "pass1_ms", "pass2_ms", "pass1_kernel", "pass2_kernel", "mAP")})
PY
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT -o t -- python /tmp/shape.py > $OUT/log.txt 2>&1
grep -v rocprofv3 $OUT/log.txt | tail -1 | cut -c1-400
python - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:8]:
    print("  %-90s calls %6d  avg %8.2f us  %5.1f %%" % (r["Name"][:90], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
