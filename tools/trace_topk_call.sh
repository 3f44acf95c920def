#!/bin/bash
# timeline of ONE xmh_hamming_topk call (kernel start/end stamps from rocprofv3 --kernel-trace): where the whole-call time goes
# usage: tools/trace_topk_call.sh [Q] [K]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out/trace_topk
rm -rf $OUT; mkdir -p $OUT
Q=${1:-1}; K=${2:-256}
cat > /tmp/topk_once.py <<PY
import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "clip-based-cross-modal-hash_amd")]
import bench_topk
print(bench_topk.measure(R=10_000_000, K=$K, Q=$Q, iters=20, warmup=3))
PY
rocprofv3 --output-format csv --kernel-trace -d $OUT/t -o t -- python /tmp/topk_once.py > $OUT/run.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# find the last complete call: sequences starting at a k_topk_sample (or fill/memset before it)
idx = [i for i, r in enumerate(rows) if "k_topk_sample" in r["Kernel_Name"]]
if len(idx) >= 3:
    a, b = idx[-2], idx[-1]
    seg = rows[a - 1:b - 1]
    t0 = int(seg[0]["Start_Timestamp"])
    prev_end = None
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        print("%-58s start %8.2f us  dur %7.2f us  gap before %6.2f us" % (r["Kernel_Name"].replace("(anonymous namespace)::", "")[:58], (s - t0) / 1e3, (e - s) / 1e3, gap))
        prev_end = e
    print("call span %.2f us" % ((int(seg[-1]["End_Timestamp"]) - t0) / 1e3))
PY
tail -2 $OUT/run.log
rm -rf $OUT/t
