"""calc_map_k on float "codes" (UMoED-style tanh outputs): the reference's route -- fp32 GEMM + one stable sort per query -- as
xmh_gemm_f32_sort_map.  Prints ms per call at a few shapes (COCO-shaped gallery), with HIP-event time of the C call alone."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "clip-based-cross-modal-hash_amd"))
from xmh import dense, retrieval as R          # noqa: E402
from xmh.common import calc_utils as cu        # noqa: E402

gen = torch.Generator().manual_seed(1)
dense._warned_float = True
for Q, Rn, K, C in ((500, 117218, 64, 80), (5000, 117218, 64, 80), (500, 117218, 512, 80), (50, 1_000_000, 64, 24)):
    qL = (torch.rand(Q, C, generator=gen) < 0.04).long()
    rL = (torch.rand(Rn, C, generator=gen) < 0.04).long()
    qL[:, 0] = 1
    qB, rB = torch.tanh(torch.randn(Q, K, generator=gen)).cuda(), torch.tanh(torch.randn(Rn, K, generator=gen)).cuda()
    qLd, rLd = qL.cuda(), rL.cuda()
    cu.calc_map_k(qB, rB, qLd, rLd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        m = cu.calc_map_k(qB, rB, qLd, rLd)
    dt = (time.perf_counter() - t0) / 3
    ql, rl = R.pack_labels(qLd), R.pack_labels(rLd)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        dense.map_k_float(qB, rB, ql, rl, C, None)
    e1.record()
    torch.cuda.synchronize()
    print("Q %5d x R %8d x K %4d: calc_map_k %.2f ms per call (%.3g pairs/s), C call by events %.2f ms, mAP %.5f"
          % (Q, Rn, K, dt * 1e3, Q * Rn / dt, e0.elapsed_time(e1) / 3, float(m)), flush=True)
