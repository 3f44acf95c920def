#!/bin/bash
# soak of the final library against the oracles (run on the GPU box):  gpurun -- bash tools/soak.sh [scale]
cd "${GRAFT_REPO_ROOT:-.}"
S=${1:-1}
set -x
timeout 1500 python tools/fuzz_scan_extended.py $((2000 * S)) 2>&1 | tail -2
XMH_SCAN_PACK32=0 timeout 900 python tools/fuzz_scan_extended.py $((800 * S)) 2>&1 | tail -2
timeout 1500 python tools/fuzz_topk_extended.py $((3000 * S)) 2>&1 | tail -2
timeout 1500 python tools/fuzz_topk_extended.py $((2000 * S)) ternary 2>&1 | tail -2
timeout 900 python tools/fuzz_determinism.py 200 6 2>&1 | tail -2
