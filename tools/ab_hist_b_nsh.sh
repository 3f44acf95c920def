#!/bin/bash
# A/B of k_scan_hist_b's waves per query tile (XMH_HIST_B_NSH = 1 .. 4): configs[4] shard (256 bit binary), binary / ternary scans at the COCO shape
cd "${GRAFT_REPO_ROOT:-.}"
for n in ${@:-1 2 3 4}; do
  echo "== XMH_HIST_B_NSH=$n"
  XMH_HIST_B_NSH=$n timeout 300 python tools/bench_scan_leg.py configs4_shard_scan_256bit 2>&1 | tail -1 | cut -c1-200
  XMH_HIST_B_NSH=$n timeout 300 python tools/bench_ternary_scan.py 64 128 256 2>&1 | grep -E "K=" | cut -c1-100
done
