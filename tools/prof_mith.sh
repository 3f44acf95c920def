#!/bin/bash
# rocprofv3 kernel stats of the MITH encode (image + text, B=100) -- run on the GPU box; summary in gpurun_out/prof_mith.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out/prof_mith
rm -rf $OUT; mkdir -p $OUT
cat > /tmp/mith_once.py <<'PY'
import os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "clip-based-cross-modal-hash_amd")]
import torch
import xmh.models  # noqa
from xmh.common.register import registry
from xmh.utils.config import Config
from xmh.models import weights as W
B = 100
image = W.synth_images(5, B).cuda(); ids, _ = W.synth_text(5, B); ids = ids.cuda(); kpm = ids == 0
model = registry.get_model_class("MITH").from_config(Config({"clip_path": "synthetic:1814"}), output_dim=64, train_num=1000).cuda().eval()
with torch.no_grad():
    for _ in range(5):
        model.encode_image(image); model.encode_text(ids, kpm)
torch.cuda.synchronize()
PY
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o m -- python /tmp/mith_once.py > $OUT/trace.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_mith/trace/**/m_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per (image+text) pass: %.3f ms" % (tot / 5 / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print("%-70s calls/pass %6.1f  avg %8.2f us  %5.1f %%" % (r["Name"][:70], int(r["Calls"]) / 5, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
rm -rf $OUT/trace
