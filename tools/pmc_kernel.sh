#!/bin/bash
# PMC counters of one kernel (name substring $1) while running command "$2" (GPU box): three counter passes, per-launch averages.
#   gpurun -- bash tools/pmc_kernel.sh k_attention_mfma64 "python tools/bench_attention.py"
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
K="$1"; CMD="$2"
OUT=/tmp/pmc_kernel; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d $OUT/p1 -o b -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d $OUT/p2 -o b -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_WAVE_CYCLES -d $OUT/p3 -o b -- $CMD > $OUT/p3.log 2>&1
rocprofv3 --output-format csv --pmc TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum -d $OUT/p4 -o b -- $CMD > $OUT/p4.log 2>&1
python - "$OUT" "$K" <<'PY'
import csv, glob, sys, collections
out, key = sys.argv[1], sys.argv[2]
for p in ("p1", "p2", "p3", "p4"):
    files = glob.glob(out + "/" + p + "/**/*counter_collection.csv", recursive=True)
    acc, n = collections.defaultdict(float), collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            if key in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for c in acc:
        print("%-34s per launch %14.0f   (%d launches)" % (c, acc[c] / n[c], n[c]))
PY
