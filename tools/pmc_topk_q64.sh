#!/bin/bash
# HBM traffic of the top-k filter at Q = 64 ALONE (10 M x 256 bit): round 4's profile averaged k_topk_filter_mfma<8, 4> over the Q = 64 leg
# and the Q = 5000 leg (79 passes over the gallery) of one process and read 1.47 GB per launch
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=/tmp/pmc_q64; rm -rf $OUT; mkdir -p $OUT gpurun_out
CMD="python -c 'import bench_topk, json; print(json.dumps(bench_topk.measure(Q=64)))'"
for c in FETCH_SIZE WRITE_SIZE "TCC_MISS_sum TCC_HIT_sum"; do
  n=$(echo $c | tr ' ' '_')
  eval rocprofv3 --output-format csv --pmc $c -d $OUT/$n -o b -- $CMD > $OUT/$n.log 2>&1
done
python - "$OUT" <<'PY' | tee gpurun_out/topk_q64_pmc.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if "topk" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
print("top-k at Q = 64 alone, 10 M x 256 bit (algorithmic gallery bytes 320 MB); FETCH_SIZE / WRITE_SIZE in KB, TCC_MISS x 128 B")
for k in sorted(acc):
    row = {c: acc[k][c] / n[k][c] for c in acc[k]}
    print("%-34s launches %3d  FETCH_SIZE %.1f MB raw (x2 for 8-16 B/lane streams: %.1f MB)  WRITE_SIZE %.2f MB  TCC_MISS*128 %.1f MB  TCC_HIT*128 %.1f MB" % (
        k, max(n[k].values()), row.get("FETCH_SIZE", 0) * 1024 / 1e6, row.get("FETCH_SIZE", 0) * 2048 / 1e6, row.get("WRITE_SIZE", 0) * 1024 / 1e6,
        row.get("TCC_MISS_sum", 0) * 128 / 1e6, row.get("TCC_HIT_sum", 0) * 128 / 1e6))
PY
