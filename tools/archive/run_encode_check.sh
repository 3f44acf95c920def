#!/bin/bash
# encode-side GPU check: parity tests of the towers / runners, then the encode legs of the bench (DCMHT + MITH)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_runner.py -x -q -k "bench_sharded or distributed" -m gpu > gpurun_out/encode_tests.log 2>&1
tail -6 gpurun_out/encode_tests.log
timeout 600 python - <<'P' 2>&1 | tail -30
import json, bench_encode
e = bench_encode.measure()
print(json.dumps({k: v for k, v in e.items() if not isinstance(v, (dict, list)) or k in ("both_towers", "fused_batches")}, indent=1)[:3000])
m = bench_encode.measure_mith()
print(json.dumps(m, indent=1)[:1500])
P
