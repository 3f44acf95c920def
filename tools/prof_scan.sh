#!/bin/bash
# rocprofv3 kernel trace + PMC passes for the fused mAP scan (run on the GPU box; outputs under gpurun_out/prof)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out/prof_scan
mkdir -p $OUT
CMD="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-hbm-regime --no-encode"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o scan -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d $OUT/pmc1 -o scan -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc2 -o scan -- $CMD > $OUT/pmc2.log 2>&1
find $OUT -name "*.csv" | head -20
python - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/prof_scan/trace/**/*kernel_stats.csv", recursive=True):
    print("==", f)
    for row in list(csv.reader(open(f)))[:12]:
        print([c[:60] for c in row])
for tag in ("pmc1", "pmc2"):
    for f in glob.glob("gpurun_out/prof_scan/%s/**/*counter_collection.csv" % tag, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        print("==", f)
        for k, v in agg.items():
            if "scan" in k:
                print(k, {a: "%.3g" % b for a, b in v.items()})
PY
