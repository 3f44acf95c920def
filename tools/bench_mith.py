import sys, os, json
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "clip-based-cross-modal-hash_amd")]
import bench_encode
d = bench_encode.measure_mith()
print({k: round(v) for k, v in d.items() if "per_s" in k})
