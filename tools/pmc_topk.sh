#!/bin/bash
# HBM traffic (PMC) and kernel-trace durations of a top-k call at Q queries ALONE (10 M x 256 bit): tools/pmc_topk.sh [Q]
# -> gpurun_out/topk_q<Q>_pmc.txt, gpurun_out/topk_q<Q>_kernel_stats.csv.  Counters in their own passes (no trace domains with --pmc).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
Q=${1:-1}
OUT=/tmp/pmc_q$Q; rm -rf $OUT; mkdir -p $OUT gpurun_out
CMD="python tools/topk_call_loop.py $Q"
for c in FETCH_SIZE WRITE_SIZE "TCC_MISS_sum TCC_HIT_sum"; do
  n=$(echo $c | tr ' ' '_')
  eval timeout 300 rocprofv3 --output-format csv --pmc $c -d $OUT/$n -o b -- $CMD > $OUT/$n.log 2>&1
done
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
cp $OUT/trace/t_kernel_stats.csv gpurun_out/topk_q${Q}_kernel_stats.csv 2>/dev/null
python - "$OUT" "$Q" <<'PY' | tee gpurun_out/topk_q${Q}_pmc.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if "topk" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
print("top-k at Q = %s alone, 10 M x 256 bit (algorithmic gallery bytes 320 MB); FETCH_SIZE / WRITE_SIZE in KB, TCC_MISS x 128 B" % sys.argv[2])
for k in sorted(acc):
    row = {c: acc[k][c] / n[k][c] for c in acc[k]}
    print("%-34s launches %3d  FETCH_SIZE %.1f MB raw (x2 for 16 B/lane streams: %.1f MB)  WRITE_SIZE %.2f MB  TCC_MISS*128 %.1f MB  TCC_HIT*128 %.1f MB" % (
        k, max(n[k].values()), row.get("FETCH_SIZE", 0) * 1024 / 1e6, row.get("FETCH_SIZE", 0) * 2048 / 1e6, row.get("WRITE_SIZE", 0) * 1024 / 1e6,
        row.get("TCC_MISS_sum", 0) * 128 / 1e6, row.get("TCC_HIT_sum", 0) * 128 / 1e6))
try:
    rows = list(csv.DictReader(open(sys.argv[1] + "/trace/t_kernel_stats.csv")))
    print("kernel-trace durations of the same command (rocprofv3 --kernel-trace --stats):")
    for r in rows:
        if "topk" in r["Name"]:
            print("  %-60s calls %4s  avg %8.1f us  min %8.1f us" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
except Exception as e:
    print("no kernel trace:", e)
PY
