#!/bin/bash
# A/B of k_scan_hist_b (XMH_SCAN_BITS=1, default) against k_scan_hist_m (=0) on the 128- and 256-bit bench legs; run on the GPU box
for v in 1 0 1 0; do
  echo "== XMH_SCAN_BITS=$v"
  XMH_SCAN_BITS=$v python -c "
import sys, os, json
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),'clip-based-cross-modal-hash_amd')]
import bench_roofline as B
for k in ('configs3_dsph_128bit','configs4_shard_scan_256bit'):
    r=B.extra_scan_leg(**B.EXTRA_LEGS[k])
    print(k, 'step %.4f ms' % r['ms_per_step'], 'mAP %.9f' % r['mAP'], {a: round(b,4) for a,b in r.items() if 'pass' in a and isinstance(b,float)})
r=B.extra_scan_leg(what='256-bit at the COCO shape', Q=5000, Rn=117218, K=256, C=80, p_label=0.04, seed=3815, steps=30)
print('k256_coco', 'step %.4f ms' % r['ms_per_step'], 'mAP %.9f' % r['mAP'], {a: round(b,4) for a,b in r.items() if 'pass' in a and isinstance(b,float)})
" 2>&1 | grep -v amdgpu.ids
done
