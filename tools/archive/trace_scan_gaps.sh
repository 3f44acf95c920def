#!/bin/bash
# rocprofv3 kernel trace of the headline step: per-kernel durations and the gaps between consecutive kernels of one step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out/trace_scan
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --output-format csv --kernel-trace -d $OUT/t -o s -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-hbm-regime --no-encode $* > $OUT/log.txt 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace_scan/t/**/s_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the timed steps: from one pass-1 kernel to the next
idx = [i for i, r in enumerate(rows) if "k_scan_hist_" in r["Kernel_Name"]]
i0 = idx[4]                                  # a steady-state step
i1 = idx[5]
prev_end = None
for r in rows[i0:i1 + 1]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev_end is None else (st - prev_end) / 1e3
    print("%-60s dur %8.2f us  gap before %6.2f us" % (r["Kernel_Name"][:60], (en - st) / 1e3, gap))
    prev_end = en
print("step period %.2f us" % ((int(rows[i1]["Start_Timestamp"]) - int(rows[i0]["Start_Timestamp"])) / 1e3))
PY
rm -rf $OUT/t
