#!/bin/bash
# kernel timeline of ONE calc_map_k call on device tensors (COCO shape, 64 bit): where the time between the kernels goes
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
cat > /tmp/dropin_loop.py <<'PY'
import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
import bench
from xmh.common import calc_utils as cu
qB, qL, rB, rL = bench.synth(5000, 117218, 64, 80, seed=1814, p=0.04)
qB, rB, qL, rL = qB.cuda(), rB.cuda(), qL.cuda(), rL.cuda()
for _ in range(12):
    m = cu.calc_map_k(qB, rB, qL, rL)
torch.cuda.synchronize()
PY
rm -rf /tmp/dropin_tr
timeout 300 rocprofv3 --output-format csv --kernel-trace --memory-copy-trace -d /tmp/dropin_tr -o d -- python /tmp/dropin_loop.py > /tmp/dropin_tr.log 2>&1
python - <<'PY'
import csv, glob
k = list(csv.DictReader(open(glob.glob("/tmp/dropin_tr/**/d_kernel_trace.csv", recursive=True)[0])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]) for r in k]
mc = glob.glob("/tmp/dropin_tr/**/d_memory_copy_trace.csv", recursive=True)
if mc:
    for r in csv.DictReader(open(mc[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Size", r.get("Bytes", ""))))
ev.sort()
# the last call = from the last k_pack... pair back to the end
idx = [i for i, e in enumerate(ev) if "pack_sign" in e[2] or "k_pack" in e[2]]
start = idx[-2]
t0 = ev[start][0]
prev_end = t0
print("one calc_map_k call (times in us from the first pack kernel):")
for s, e, n in ev[start:]:
    print("  +%8.1f  gap %6.1f  dur %7.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, n))
    prev_end = max(prev_end, e)
print("  span %.1f us" % ((prev_end - t0) / 1e3))
PY
