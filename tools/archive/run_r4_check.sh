#!/bin/bash
# round-4 GPU check of the new legs and tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_retrieval.py -x -q -m gpu -k "self_check or unsharded or bench_legs or topk_sharded or merge" > gpurun_out/r4_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r4_tests.log
timeout 600 python bench_valid.py > gpurun_out/valid_e2e.json 2> gpurun_out/valid_e2e.err; echo "valid rc=$?"; cat gpurun_out/valid_e2e.json; tail -3 gpurun_out/valid_e2e.err
timeout 600 python - > gpurun_out/topk_extra.json 2> gpurun_out/topk_extra.err <<'PY'
import json, bench_topk
print(json.dumps({"many": bench_topk.measure_many_queries(), "defeat": bench_topk.measure_cache_defeat()}))
PY
echo "topk rc=$?"; cat gpurun_out/topk_extra.json; tail -3 gpurun_out/topk_extra.err
