set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/proto_gemm_pp.hip -o /tmp/pp_bin 2>&1 | grep -v "argument unused" || true
for s in 0 1 2 3 4 5 6 7 8; do timeout 300 /tmp/pp_bin $s; done > gpurun_out/pp1.log 2>&1
tail -5 gpurun_out/pp1.log
