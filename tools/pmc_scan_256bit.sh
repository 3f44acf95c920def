#!/bin/bash
# PMC rows of the two passes of the scan on configs[4]'s shard (Q 5000 x R 1.25 M x 256 bit): k_scan_hist_b and k_scan_ap_c<., 16>
#   gpurun -- bash tools/pmc_scan_256bit.sh  ->  gpurun_out/pmc_scan_256bit.txt
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
CMD="python tools/bench_scan_leg.py configs4_shard_scan_256bit"
OUT=/tmp/pmc256; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/tr -o b -- $CMD > $OUT/tr.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/f -o b -- $CMD > $OUT/f.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/w -o b -- $CMD > $OUT/w.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $OUT/p1 -o b -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d $OUT/p2 -o b -- $CMD > $OUT/p2.log 2>&1
python - "$OUT" <<'PY' | tee gpurun_out/pmc_scan_256bit.txt
import csv, glob, sys, collections
out = sys.argv[1]
print("configs[4] shard, Q 5000 x R 1 250 000 x 256 bit, C 80 (tools/pmc_scan_256bit.sh): per-launch averages")
st = glob.glob(out + "/tr/**/b_kernel_stats.csv", recursive=True)
if st:
    for r in csv.DictReader(open(st[0])):
        if "k_scan_hist_b" in r["Name"] or "k_scan_ap_c" in r["Name"] or "k_scan_below" in r["Name"]:
            print("  kernel trace: %-60s calls %4s  avg %10.1f us" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
for key in ("k_scan_hist_b", "k_scan_ap_c"):
    print("### " + key)
    for p in ("f", "w", "p1", "p2"):
        acc, n = collections.defaultdict(float), collections.Counter()
        for f in glob.glob(out + "/" + p + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if key in r["Kernel_Name"]:
                    acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for c in acc:
            v = acc[c] / n[c]
            extra = ""
            if c == "FETCH_SIZE": extra = "  = %.2f GB read (x 1024 x 2: the guide's gfx950 unit)" % (v * 1024 * 2 / 1e9)
            if c == "WRITE_SIZE": extra = "  = %.2f GB written (x 1024)" % (v * 1024 / 1e9)
            print("  %-30s %16.0f  (%d launches)%s" % (c, v, n[c], extra))
PY
