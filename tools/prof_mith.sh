#!/bin/bash
# kernel trace of the MITH encode legs (batch 100, parity mode): tools/prof_mith.sh -> gpurun_out/mith_{images,captions}_stats.txt
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
cat > /tmp/mith_loop.py <<'PY'
import sys, os, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
import xmh.models
from xmh.common.register import registry
from xmh.models import weights as W
from xmh.utils.config import Config
what = sys.argv[1]
model = registry.get_model_class("MITH").from_config(Config({"clip_path": "synthetic:1814"}), output_dim=64, train_num=1000).cuda().eval()
image = W.synth_images(5, 100).cuda()
ids, _ = W.synth_text(5, 100); ids = ids.cuda(); kpm = ids == 0
fn = (lambda: model.encode_image(image)) if what == "images" else (lambda: model.encode_text(ids, kpm))
with torch.no_grad():
    for _ in range(30): fn()
torch.cuda.synchronize()
PY
for w in images captions; do
  timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/mith_$w -o m -- python /tmp/mith_loop.py $w > /tmp/mith_$w.log 2>&1
  python - /tmp/mith_$w/m_kernel_stats.csv $w <<'PY' | tee gpurun_out/mith_${w}_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("MITH %s, batch 100, 30 forwards: GPU time per forward %.3f ms" % (sys.argv[2], tot / 30 / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print("%6.2f %%  %6d calls  avg %8.1f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:110]))
PY
done
python - <<'PY' | tee gpurun_out/mith_images_sequence.txt
import csv
rows = sorted(csv.DictReader(open("/tmp/mith_images/m_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
last = rows[-(n // 30):]
print("kernels of the last forward, in order (duration us):")
for r in last:
    print("%7.1f  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:90]))
PY
