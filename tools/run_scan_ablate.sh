#!/bin/bash
# build the ablation matrix of tools/proto_scan_ablate.hip here (hipcc cross-compiles), run it on the GPU box:
#   bash tools/run_scan_ablate.sh build     (in the build container)
#   gpurun -- bash tools/run_scan_ablate.sh run
cd "$(dirname "$0")/.."
if [ "$2" = wide ]; then
VARIANTS=("g44:" "g28:-DPNW=2 -DPNQ=8" "g18:-DPNW=1 -DPNQ=8" "g48:-DPNQ=8")
elif [ "$2" = skel ]; then
X="-DXMH_ABL_NOADD -DXMH_ABL_NOMFMA"
VARIANTS=("s0:$X" "s_nomin:$X -DXMH_ABL_NOMIN" "s_minfull:$X -DXMH_ABL_MINFULL" "s_nobuild:$X -DXMH_ABL_NOBUILD" "s_noloadw:$X -DXMH_ABL_NOLOADW" "s_nomin_nobuild:$X -DXMH_ABL_NOMIN -DXMH_ABL_NOBUILD"
          "s_all:$X -DXMH_ABL_NOMIN -DXMH_ABL_NOBUILD -DXMH_ABL_NOLOADW" "f_minfull:-DXMH_ABL_MINFULL" "f_nobuild:-DXMH_ABL_NOBUILD" "f_nomin:-DXMH_ABL_NOMIN")
elif [ "$2" = nops ]; then
VARIANTS=("n0:-DXMH_R2_NOPS=0" "n1:-DXMH_R2_NOPS=1" "n2:-DXMH_R2_NOPS=2" "n3:-DXMH_R2_NOPS=3" "n4:-DXMH_R2_NOPS=4")
elif [ "$2" = geom ]; then
VARIANTS=("g44:" "g42:-DPNQ=2" "g41:-DPNQ=1" "g24:-DPNW=2" "g22:-DPNW=2 -DPNQ=2" "g14:-DPNW=1" "g12:-DPNW=1 -DPNQ=2" "g82:-DPNW=8 -DPNQ=2"
          "g42w3:-DPNQ=2 -DXMH_R2_ATTR=__attribute__((amdgpu_waves_per_eu(3,3)))" "g42w4:-DPNQ=2 -DXMH_R2_ATTR=__attribute__((amdgpu_waves_per_eu(4,4)))"
          "g44w3:-DXMH_R2_ATTR=__attribute__((amdgpu_waves_per_eu(3,3)))")
else
VARIANTS=("base:" "noadd:-DXMH_ABL_NOADD" "nomfma:-DXMH_ABL_NOMFMA" "noadd_nomfma:-DXMH_ABL_NOADD -DXMH_ABL_NOMFMA" "nostore:-DXMH_ABL_NOSTORE" "nostore_noadd:-DXMH_ABL_NOSTORE -DXMH_ABL_NOADD"
          "ap_noatomic:-DXMH_ABL_AP_NOATOMIC" "ap_norcp:-DXMH_ABL_AP_NORCP" "ap_noload:-DXMH_ABL_AP_NOLOAD" "ap_noatomic_noload:-DXMH_ABL_AP_NOATOMIC -DXMH_ABL_AP_NOLOAD")
fi
if [ "$1" = build ]; then
  for v in "${VARIANTS[@]}"; do
    name=${v%%:*}; flags=${v#*:}
    extra=""; [[ "$flags" == *XMH_ABL* ]] && extra="-DXMH_ABL_ANY"
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -I clip-based-cross-modal-hash_amd/csrc -I include $flags $extra tools/proto_scan_ablate.hip -o tools/proto_scan_ablate_$name.bin &
  done
  wait; ls -la tools/proto_scan_ablate_*.bin
else
  mkdir -p gpurun_out
  for v in "${VARIANTS[@]}"; do
    name=${v%%:*}
    echo "== $name"; ./tools/proto_scan_ablate_$name.bin 100 2>&1 | grep -v amdgpu.ids
  done | tee gpurun_out/scan_ablate.txt
fi
