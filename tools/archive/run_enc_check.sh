#!/bin/bash
# GPU-box check after a GEMM change: encoder tests, GEMM shape bench, encode bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_encode.py tests/test_gpu_runner.py -x -q -m gpu > gpurun_out/enc_tests.log 2>&1; echo "tests rc=$?" 
tail -3 gpurun_out/enc_tests.log
timeout 300 python tools/bench_gemm_planes.py > gpurun_out/gemm_planes.log 2>&1; cat gpurun_out/gemm_planes.log
timeout 600 python bench_encode.py --batch 100 > gpurun_out/bench_encode_100.json 2> gpurun_out/bench_encode_100.err; tail -c 1500 gpurun_out/bench_encode_100.json
