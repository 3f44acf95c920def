#!/bin/bash
mkdir -p gpurun_out
timeout 600 python - <<'P' > gpurun_out/encode_leg.json 2> gpurun_out/encode_leg.err
import json, bench_encode
e = bench_encode.measure()
m = bench_encode.measure_mith()
print(json.dumps({"encode": e, "mith": m}, indent=1))
P
python - <<'P'
import json
d = json.load(open("gpurun_out/encode_leg.json")); e = d["encode"]
for k in ("images_per_s_f32", "captions_per_s_f32", "images_per_s_f16", "captions_per_s_f16", "captions_per_s_f32_padded_tower"):
    print(k, e.get(k))
print("fused", {k: v for k, v in e["fused_batches"].items() if k != "workload"})
print("both", {k: v for k, v in e["both_towers"].items() if k != "workload"})
print("mith", {k: v for k, v in d["mith"].items() if k != "config"})
P
# text-only MITH profile
cat > /tmp/mith_text.py <<'PY'
import os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "clip-based-cross-modal-hash_amd")]
import torch
import xmh.models  # noqa
from xmh.common.register import registry
from xmh.utils.config import Config
from xmh.models import weights as W
B = 100
ids, _ = W.synth_text(5, B); ids = ids.cuda(); kpm = ids == 0
model = registry.get_model_class("MITH").from_config(Config({"clip_path": "synthetic:1814"}), output_dim=64, train_num=1000).cuda().eval()
with torch.no_grad():
    for _ in range(5):
        model.encode_text(ids, kpm)
torch.cuda.synchronize()
PY
export TMPDIR=/tmp
rm -rf /tmp/pm; rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/pm -o m -- python /tmp/mith_text.py > /tmp/pm.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/pm/**/m_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("MITH text: total kernel time per pass: %.3f ms" % (tot / 5 / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:25]:
    print("%-90s calls/pass %6.1f  avg %8.2f us  %5.1f %%" % (r["Name"][:90], int(r["Calls"]) / 5, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
