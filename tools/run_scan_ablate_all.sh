#!/bin/bash
# every matrix of tools/run_scan_ablate.sh in one GPU call -> gpurun_out/scan_ablate_all.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; : > gpurun_out/scan_ablate_all.txt
for m in "" skel nops geom wide; do
  echo "#### matrix: ${m:-pipes}" >> gpurun_out/scan_ablate_all.txt
  bash tools/run_scan_ablate.sh run $m > /dev/null
  cat gpurun_out/scan_ablate.txt >> gpurun_out/scan_ablate_all.txt
done
grep -v "amdgpu.ids" gpurun_out/scan_ablate_all.txt | tail -5
