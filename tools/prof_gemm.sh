#!/bin/bash
# rocprofv3 PMC passes over tools/bench_gemm.py (run on the GPU box; outputs under gpurun_out/prof_gemm)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out/prof_gemm
rm -rf $OUT; mkdir -p $OUT
CMD="python tools/bench_gemm.py"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o g -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d $OUT/pmc1 -o g -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/pmc2 -o g -- $CMD > $OUT/pmc2.log 2>&1
python - <<'PY'
import csv, glob, collections
for tag in ("pmc1", "pmc2"):
    for f in glob.glob("gpurun_out/prof_gemm/%s/**/*counter_collection.csv" % tag, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.Counter())
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60] + " grid=" + r.get("Grid_Size", "?")
            if "gemm" not in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
        print("==", tag)
        for k, v in agg.items():
            print(k, {a: "%.3g" % (b / n[k][a]) for a, b in v.items()})
PY
