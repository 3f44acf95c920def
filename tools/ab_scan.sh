for n in 1 2 3; do
  echo "== ABL $n"; XMH_LIB=$PWD/clip-based-cross-modal-hash_amd/xmh/libxmh_abl$n.so XMH_SCAN_MFMA_AP=1 XMH_SCAN_CACHE_MB=0 timeout 200 python bench.py --steps 50 --no-cpu-baseline --no-hbm-regime --no-encode --no-extra-configs 2>>gpurun_out/b_mfma.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['mAP'], d['roofline']['pass1_avg_launch_ms'], d['roofline']['pass2_avg_launch_ms'])"
done
