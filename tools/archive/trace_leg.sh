#!/bin/bash
# per-kernel times of one extra scan leg of the bench (GPU box):  bash tools/trace_leg.sh configs0_dcmht_16bit_mirflickr
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
LEG=${1:-configs0_dcmht_16bit_mirflickr}
OUT=/tmp/trace_leg; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/leg.py <<PY
import sys
sys.path[:0] = [".", "clip-based-cross-modal-hash_amd"]
import bench_roofline as RL
print(RL.extra_scan_leg(**dict(RL.EXTRA_LEGS["$LEG"], steps=200)))
PY
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT -o t -- python /tmp/leg.py > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt | cut -c1-400
python - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:12]:
    print("  %-90s calls %6d  avg %8.2f us  %5.1f %%" % (r["Name"][:90], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
