"""one extra scan leg of bench.py alone (names: bench_roofline.EXTRA_LEGS), with the pass times by the library's events:
    python tools/bench_scan_leg.py configs4_shard_scan_256bit [configs3_dsph_128bit ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
import bench_roofline as RL
for name in sys.argv[1:] or ["configs4_shard_scan_256bit"]:
    o = RL.extra_scan_leg(**RL.EXTRA_LEGS[name])
    print(name, json.dumps({k: o[k] for k in o if k in ("ms_per_step", "pairs_per_s", "mAP", "pass1_ms", "pass2_ms", "pass2_kernel", "kernels", "avg_launch_ms", "algorithmic")}, default=str)[:900], flush=True)
