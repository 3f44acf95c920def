#!/bin/bash
# per-shape kernel time of the GEMM kernels alone (rocprofv3 kernel trace of tools/bench_gemm.py, grouped by kernel and grid)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out/prof_gemm_shapes
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --output-format csv --kernel-trace -d $OUT/t -o g -- python tools/bench_gemm.py > $OUT/bench.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "gemm" not in n and "split_planes" not in n: continue
    m = re.search(r"k_gemm_\w+<[^>]*>|k_split_planes", n)
    key = (m.group(0) if m else n[:60], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg.setdefault(key, []).append(d)
for (k, g), v in agg.items():
    v = sorted(v)
    print("%-44s blocks %6d  calls %3d  median %8.2f us  min %8.2f" % (k, g, len(v), v[len(v) // 2], v[0]))
PY
rm -rf $OUT/t
