#!/bin/bash
# Round profile: rocprofv3 kernel trace + stats of the default bench command, then separate PMC passes
# (FETCH_SIZE / WRITE_SIZE cannot share a pass: TCC has 4 slots, FETCH_SIZE costs 3, WRITE_SIZE 2).
# Run on the GPU box:  gpurun -- ./tools/profile_round.sh r01      -> gpurun_out/profile_<tag>/..., summary JSON/MD
TAG=${1:-r01}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/profile_$TAG
rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --steps 20 --warmup 3 --no-cpu-baseline"
# the counter passes leave the extra scan legs out: they launch some of the headline's kernel instances at OTHER shapes (k_scan_ap_c<false>
# serves 16- and 64-bit codes alike), and a per-launch average over mixed shapes is nobody's number
PMC_CMD="$CMD --no-extra-configs"
# the per-device self-check of k_scan_hist_m2 launches the headline's kernel instances once at a 200 x 9000 shape: keep it out of the
# per-launch counter averages
export XMH_SCAN_M2_SELFCHECK=0
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o b -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o b -- $PMC_CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o b -- $PMC_CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d $OUT/pmc_sq1 -o b -- $PMC_CMD > $OUT/pmc_sq1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d $OUT/pmc_sq2 -o b -- $PMC_CMD > $OUT/pmc_sq2.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_WAVE_CYCLES -d $OUT/pmc_sq3 -o b -- $PMC_CMD > $OUT/pmc_sq3.log 2>&1
python tools/summarize_profile.py $OUT $TAG
# the raw per-dispatch CSVs are tens of MB (the encoder leg launches thousands of kernels); gpurun only carries 64 MiB back
cp $OUT/trace/b_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_sq3 $OUT/trace
