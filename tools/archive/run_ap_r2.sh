#!/bin/bash
# A/B of pass 2 without a pair cache (k_scan_ap_r2) against the cached k_scan_ap_c: parity tests with the switch on, then the headline bench either way
mkdir -p gpurun_out
export XMH_SCAN_AP_R2=1
timeout 1200 python -m pytest tests/test_gpu_retrieval.py -x -q -m gpu > gpurun_out/ap_r2_tests.log 2>&1
tail -5 gpurun_out/ap_r2_tests.log
B="python bench.py --no-encode --no-hbm-regime --no-extra-configs --no-cpu-baseline --steps 200"
for g in 0 1 2 3; do
  XMH_SCAN_AP_R2_GEOM=$g timeout 300 $B > gpurun_out/ap_r2_bench_g$g.json 2> gpurun_out/ap_r2_bench_g$g.err
  python - <<P
import json
d=json.load(open("gpurun_out/bench_detail.json")); r=d["roofline"]
print("geom $g", d["ms_per_step"], r.get("pass1_avg_launch_ms"), r.get("pass2_avg_launch_ms"), r["pass2"]["kernel"], d["mAP"])
P
done
XMH_SCAN_AP_R2=0 timeout 300 $B > gpurun_out/ap_r2_bench_off.json 2> gpurun_out/ap_r2_bench_off.err
python - <<P
import json
d=json.load(open("gpurun_out/bench_detail.json")); r=d["roofline"]
print("cached", d["ms_per_step"], r.get("pass1_avg_launch_ms"), r.get("pass2_avg_launch_ms"), r["pass2"]["kernel"], d["mAP"])
P
