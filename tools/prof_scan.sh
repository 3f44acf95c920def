#!/bin/bash
# rocprofv3 kernel trace + PMC passes for the fused mAP scan (run on the GPU box; outputs under gpurun_out/prof_scan)
# usage: tools/prof_scan.sh [extra bench.py args, e.g. --K 2048]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out/prof_scan
rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-hbm-regime --no-encode $*"
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d $OUT/pmc1 -o scan -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc2 -o scan -- $CMD > $OUT/pmc2.log 2>&1
python - <<'PY'
import csv, glob, collections
for tag in ("pmc1", "pmc2"):
    for f in glob.glob("gpurun_out/prof_scan/%s/**/*counter_collection.csv" % tag, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:48]
            if "scan" not in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
        print("==", tag)
        for k, v in agg.items():
            print(k, {a: "%.3g" % (b / n[k][a]) for a, b in v.items()})
PY
rm -rf $OUT/pmc1 $OUT/pmc2
