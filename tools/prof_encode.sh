#!/bin/bash
# kernel-time breakdown of parity-mode image / text forwards (run on the GPU box)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out/prof_encode
rm -rf $OUT; mkdir -p $OUT
cat > /tmp/enc_loop.py <<'PY'
import sys, torch
sys.path[:0] = [".", "clip-based-cross-modal-hash_amd"]
from xmh import ops, retrieval as R
from xmh.models.dcmht import DCMHT
from xmh.utils.config import Config
from xmh.models import weights as W
which = sys.argv[1]
model = DCMHT.from_config(Config({"clip_path": "synthetic:1814"}), output_dim=64).cuda().eval()
import os
NB = int(os.environ.get("XMH_PROF_BATCH", "100"))
image = W.synth_images(5, 100).cuda().repeat(NB // 100, 1, 1, 1)
ids, _ = W.synth_text(5, 100); ids = ids.cuda().repeat(NB // 100, 1)
fn = (lambda: R.pack_pair_argmax(model.encode_image(image))) if which == "image" else (lambda: R.pack_pair_argmax(model.encode_text(ids)))
for _ in range(22): fn()
torch.cuda.synchronize()
PY
for w in image text; do
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/$w -o e -- python /tmp/enc_loop.py $w > $OUT/$w.log 2>&1
  python - "$OUT/$w" "$w" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("==", sys.argv[2], "total kernel ms per forward: %.3f" % (tot / 22 / 1e6))
for r in rows[:14]:
    print("  %-70s calls/fwd %6.1f  avg %8.2f us  %5.1f %%" % (r["Name"][:70], int(r["Calls"]) / 22, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / tot * 100))
PY
done
rm -rf $OUT/image $OUT/text
