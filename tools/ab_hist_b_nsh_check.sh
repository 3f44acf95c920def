#!/bin/bash
# correctness of k_scan_hist_b with NSH waves per query tile: per-bucket histograms against the oracle, the scan fuzz, the scan tests
cd "${GRAFT_REPO_ROOT:-.}"
for n in ${@:-2 3}; do
  echo "== XMH_HIST_B_NSH=$n"
  for shape in "40 3000 256 80" "200 70000 256 80" "130 20000 200 24" "64 9000 129 128"; do
    XMH_HIST_B_NSH=$n timeout 300 python tools/diag_bits.py $shape 2>&1 | grep -E "bad all"
  done
  XMH_HIST_B_NSH=$n timeout 900 python tools/fuzz_scan_extended.py 400 2>&1 | tail -2
  XMH_HIST_B_NSH=$n timeout 900 python -m pytest tests/test_gpu_retrieval.py -q -x -m gpu -k "ternary or 256 or bits or long or cache or verify or golden" 2>&1 | tail -2
done
