#!/bin/bash
# rocprofv3 marker + kernel trace of one small valid() + retrieve_topk() with the library's roctx ranges on (xmh_prof_enable(2)):
# which kernels run inside which phase, and the phase durations.  Run on the GPU box:  gpurun -- bash tools/prof_valid_ranges.sh
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_ranges
rm -rf $OUT; mkdir -p $OUT
cat > /tmp/valid_ranges.py <<'PY'
import os, sys, pathlib, tempfile
root = os.getcwd()
sys.path[:0] = [root, os.path.join(root, "clip-based-cross-modal-hash_amd"), os.path.join(root, "tests")]
import torch
from test_gpu_runner import make_cfg
import xmh.models, xmh.runners
from xmh import _lib
from xmh.common.register import registry
tmp = pathlib.Path(tempfile.mkdtemp())
cfg = make_cfg(tmp, "MITH", "MITHTrainer", 64, layers=2)
cfg.dataset.retrieval_num, cfg.run.query_num = 2000, 200
t = registry.get_runner_class("MITHTrainer").from_config(cfg=cfg, autorun=False)
t.valid(0, k=None)                     # warm
torch.cuda.synchronize()
_lib.prof_enable(2)
t.valid(1, k=None)
t.retrieve_topk(10, tasks=("i2t",))
torch.cuda.synchronize()
_lib.prof_enable(0)
PY
rocprofv3 --output-format csv --marker-trace --kernel-trace --stats -d $OUT/t -o v -- python /tmp/valid_ranges.py > $OUT/run.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
mk = glob.glob(out + "/**/*marker_api_trace.csv", recursive=True)
kt = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)
if not mk or not kt:
    print("no marker / kernel trace found:", glob.glob(out + "/**/*.csv", recursive=True)); sys.exit(0)
ranges = [r for r in csv.DictReader(open(mk[0]))]
kernels = sorted(csv.DictReader(open(kt[0])), key=lambda r: int(r["Start_Timestamp"]))
agg = collections.OrderedDict()
for r in ranges:
    name = r.get("Function") or r.get("Name") or r.get("Message") or "?"
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    a = agg.setdefault(name, [0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3
with open(out + "/ranges_summary.txt", "w") as f:
    f.write("roctx ranges of one MITH valid() + retrieve_topk() (200 queries, 2000 gallery rows, 2-layer towers; host-side durations)\n")
    f.write("%-64s %8s %12s\n" % ("range", "count", "total us"))
    for name, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write("%-64s %8d %12.1f\n" % (name[:64], n, us))
    f.write("\nkernels launched while ranges were on: %d\n" % len(kernels))
print(open(out + "/ranges_summary.txt").read()[:3000])
PY
rm -rf $OUT/t
