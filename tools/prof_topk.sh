#!/bin/bash
# rocprofv3 PMC passes for the top-k filter (run on the GPU box; outputs under gpurun_out/prof_topk)
# usage: tools/prof_topk.sh [Q] [K]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out/prof_topk
rm -rf $OUT; mkdir -p $OUT
Q=${1:-8}; K=${2:-256}
cat > /tmp/topk_once.py <<PY
import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "clip-based-cross-modal-hash_amd")]
import bench_topk
print(bench_topk.measure(R=10_000_000, K=$K, Q=$Q, iters=5, warmup=2))
PY
CMD="python /tmp/topk_once.py"
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_INSTS_SMEM -d $OUT/pmc1 -o t -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/pmc2 -o t -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --output-format csv --pmc SQ_THREAD_CYCLES_VALU SQ_VALU_DEP_STALL SQ_INST_LEVEL_VMEM SQ_WAIT_INST_VMEM TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc3 -o t -- $CMD > $OUT/pmc3.log 2>&1
python - <<'PY'
import csv, glob, collections
for tag in ("pmc1", "pmc2", "pmc3"):
    for f in glob.glob("gpurun_out/prof_topk/%s/**/*counter_collection.csv" % tag, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            if "filter" not in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
        print("==", tag)
        for k, v in agg.items():
            print(k, {a: "%.4g" % (b / n[k][a]) for a, b in v.items()})
PY
tail -3 $OUT/pmc3.log
rm -rf $OUT/pmc1 $OUT/pmc2 $OUT/pmc3
