#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
import torch
from xmh import retrieval as xr
os.environ["XMH_SCAN_AP_HALF"] = "1"
for (Q, R, K, C) in ((513, 3180, 64, 5), (513, 3328, 64, 5), (513, 256, 64, 5), (513, 64, 64, 5), (513, 128, 64, 5), (64, 3180, 64, 5), (513, 3180, 64, 80), (513, 20000, 64, 5), (5000, 20015, 64, 24)):
    gen = torch.Generator().manual_seed(1)
    qB, rB = torch.randn(Q, K, generator=gen).sign(), torch.randn(R, K, generator=gen).sign()
    qL, rL = (torch.rand(Q, C, generator=gen) < 0.1).long(), (torch.rand(R, C, generator=gen) < 0.1).long()
    qL[:, 0] = 1; rL[0, 0] = 1
    s = xr.RankingScan(xr.pack_sign(qB.cuda()), xr.pack_labels(qL.cuda()), xr.pack_sign(rB.cuda()), xr.pack_labels(rL.cuda()), C)
    s.histograms(False)
    seen = {}
    first = None
    offq = set()
    for i in range(30):
        ap, cap = s.ap_sums(None)                      # pass 2 only, same tables and cache every time
        a = ap.cpu()
        if first is None: first = a
        else: offq |= set((a != first).nonzero().flatten().tolist())
        key = hash(a.numpy().tobytes()); seen[key] = seen.get(key, 0) + 1
    print((Q, R, K, C), "plan", s.plan.chunk, s.plan.nchunk, s.plan.qpad, "distinct pass-2 results", len(seen), "queries ever off", len(offq), sorted(offq)[:12])
