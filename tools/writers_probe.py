import sys, time, os, io
sys.path[:0] = [".", "clip-based-cross-modal-hash_amd"]
import torch
import bench_valid as BV
from xmh.runners.base import BaseTrainer
T = {}
def wrap(name):
    orig = getattr(BaseTrainer, name)
    f = orig.__func__ if hasattr(orig, "__func__") else orig
    def timed(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T[name] = T.get(name, 0) + time.perf_counter() - t0; return r
    return timed
BaseTrainer.save_model = wrap("save_model")
sc = BaseTrainer._save_codes
def timed_sc(self, *a, **k):
    t0 = time.perf_counter(); r = sc(self, *a, **k); T["_save_codes"] = T.get("_save_codes", 0) + time.perf_counter() - t0; return r
BaseTrainer._save_codes = timed_sc
o = BV.measure()
print({k: round(v, 3) for k, v in T.items()}, "valid", round(o["valid_seconds"], 3), "encode", round(o["encode_seconds"], 3), "rest", round(o["writers_and_rest_seconds"], 3))
import subprocess
print(subprocess.run("df -h /tmp | tail -1", shell=True, capture_output=True, text=True).stdout)
