#!/usr/bin/env python3
"""exact top-k evaluated repeatedly on the same inputs: the lists must be bit-identical every time (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
import torch
from xmh import retrieval as xr
for (Q, R, K, k) in ((1, 1000003, 256, 100), (8, 1000003, 256, 100), (64, 300001, 128, 50), (513, 70001, 64, 10), (5, 200000, 512, 100), (3, 999, 256, 100))[:4] + ((5, 200000, 512, 100), (3, 999, 256, 100)):
    g = torch.Generator().manual_seed(3)
    qs = torch.randn(Q, K, generator=g).sign(); qs[qs == 0] = 1
    q = xr.pack_sign(qs.cuda())
    base = torch.randn(4096, K, generator=g).sign(); base[base == 0] = 1
    r = xr.pack_sign(base[torch.randint(0, 4096, (R,), generator=g)].cuda())      # duplicates: many ties at the cut
    first = None
    bad = 0
    junk = torch.empty(32 << 20, dtype=torch.uint8, device="cuda")
    for i in range(60):
        if i % 3 == 0: junk.random_(0, 255)
        d, idx = xr.hamming_topk(q, r, k)
        cur = (d.clone(), idx.clone())
        if first is None: first = cur
        elif not (torch.equal(cur[0], first[0]) and torch.equal(cur[1], first[1])): bad += 1
    print((Q, R, K, k), "evaluations that differ from the first:", bad)
