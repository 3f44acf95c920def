#!/usr/bin/env python3
"""k_scan_hist_r2w against k_scan_hist_m (XMH_SCAN_R2W=0) on small shapes: histograms, caps, AP sums (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
import torch
from xmh import retrieval as xr
def synth(Q, R, K, C, seed):
    g = torch.Generator().manual_seed(seed)
    qB = torch.randn(Q, K, generator=g).sign(); rB = torch.randn(R, K, generator=g).sign()
    qL = (torch.rand(Q, C, generator=g) < 0.1).long(); rL = (torch.rand(R, C, generator=g) < 0.1).long()
    qL[:, 0] = 1; rL[::3, 0] = 1
    return qB, rB, qL, rL
for (Q, R, K, C) in ((16, 64, 128, 10), (16, 128, 128, 10), (40, 300, 128, 10), (40, 300, 128, 80), (40, 300, 100, 80), (200, 9000, 128, 80), (33, 70, 96, 40)):
    qB, rB, qL, rL = synth(Q, R, K, C, 7)
    outs = []
    for flag in ("1", "0"):
        os.environ["XMH_SCAN_R2W"] = flag
        s = xr.RankingScan(xr.pack_sign(qB.cuda()), xr.pack_labels(qL.cuda()), xr.pack_sign(rB.cuda()), xr.pack_labels(rL.cuda()), C)
        ha, hr = s.histograms(True)
        ap, cap = s.ap_sums(None)
        outs.append((ha.cpu(), hr.cpu(), cap.cpu(), ap.cpu(), (s.plan.chunk, s.plan.nchunk, s.plan.qpad)))
    a, b = outs
    print((Q, R, K, C), "plans", a[4], b[4], "hist_all", bool(torch.equal(a[0], b[0])), "hist_rel", bool(torch.equal(a[1], b[1])), "cap", bool(torch.equal(a[2], b[2])),
          "ap", bool(torch.allclose(a[3], b[3], rtol=1e-5)))
    if not torch.equal(a[0], b[0]):
        bad = (a[0] != b[0]).nonzero()
        print("   first diffs", bad[:6].tolist(), a[0][bad[0][0]].nonzero().flatten().tolist()[:10], b[0][bad[0][0]].nonzero().flatten().tolist()[:10])
