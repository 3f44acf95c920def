#!/bin/bash
# PMC rows of the two pass-2 designs at the headline shape: k_scan_ap_r2 (no pair cache, XMH_SCAN_AP_R2=1) and k_scan_ap_c (one-byte pair
# cache), with the pass 1 each runs behind; separate counter passes (FETCH_SIZE and WRITE_SIZE cannot share one).
#   gpurun -- bash tools/pmc_scan_ab.sh      -> gpurun_out/scan_ab_pmc.txt
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp XMH_SCAN_M2_SELFCHECK=0
CMD="python bench.py --no-encode --no-hbm-regime --no-extra-configs --no-cpu-baseline --steps 20 --warmup 3 --settle 10"
OUT=/tmp/pmc_ab; rm -rf $OUT; mkdir -p $OUT gpurun_out
for mode in 1 0; do
  export XMH_SCAN_AP_R2=$mode
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
             "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_I8"; do
    i=$((i+1))
    rocprofv3 --output-format csv --pmc $set -d $OUT/m$mode/p$i -o b -- $CMD > $OUT/m$mode.p$i.log 2>&1
  done
done
python - "$OUT" > gpurun_out/scan_ab_pmc.txt <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for mode, title in (("1", "XMH_SCAN_AP_R2=1: pass 2 = k_scan_ap_r2 (pairs evaluated again on the MFMA, no pair cache)"), ("0", "XMH_SCAN_AP_R2=0: pass 2 = k_scan_ap_c (one-byte pair cache)")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    for f in glob.glob(out + "/m" + mode + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if not k.startswith("k_scan") and not k.startswith("k_ap_"): continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    print("== " + title)
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            v = acc[k][c] / n[k][c]
            extra = "   = %.1f MB per launch" % (v * 1024 / 1e6) if c in ("FETCH_SIZE", "WRITE_SIZE") else ""
            print("    %-28s %16.0f  (%d launches)%s" % (c, v, n[k][c], extra))
PY
cat gpurun_out/scan_ab_pmc.txt | head -150
