#!/usr/bin/env python3
"""Repeat one scan shape many times and compare every stage with its first evaluation (a rare, run-to-run difference = a race or an
unspaced hazard): histograms, divisors, the pair-cache bytes, the AP sums.  python tools/repro_rare.py [iterations] [Q R K C seed]"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
import numpy as np
import torch
from xmh import retrieval as xr
from xmh._lib import lib
it = int(sys.argv[1]) if len(sys.argv) > 1 else 300
Q, R, K, C, seed = (int(x) for x in sys.argv[2:7]) if len(sys.argv) > 6 else (513, 3180, 128, 5, 5911)
gen = torch.Generator().manual_seed(seed)
qB, rB = torch.randn(Q, K, generator=gen).sign(), torch.randn(R, K, generator=gen).sign()
qL, rL = (torch.rand(Q, C, generator=gen) < 0.1).long(), (torch.rand(R, C, generator=gen) < 0.1).long()
qL[:, 0] = 1; rL[0, 0] = 1
q, r = xr.pack_sign(qB.cuda()), xr.pack_sign(rB.cuda())
ql, rl = xr.pack_labels(qL.cuda()), xr.pack_labels(rL.cuda())
scan = xr.RankingScan(q, ql, r, rl, C)
nbytes = int(lib.xmh_scan_pair_cache_bytes(Q, R, scan.q.K, 0)); off = int(lib.xmh_scan_pair_cache_offset(Q, R, scan.q.K, 0))
print("plan", scan.plan.chunk, scan.plan.nchunk, scan.plan.qpad, "cache", nbytes)
ref = None
bad = {"hist": 0, "cap": 0, "cache": 0, "ap": 0, "map": 0}
junk = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
for i in range(it):
    if i % 3 == 0:
        junk.random_(0, 255)                     # disturb the caches / timing between evaluations
    ha, hr = scan.histograms(True)
    cache = scan.ws[off:off + nbytes].clone() if nbytes else None
    m, ap, cap = scan.map_all(None)
    cur = (ha.clone(), hr.clone(), cap.clone(), cache, ap.clone(), float(m))
    if ref is None:
        ref = cur
        continue
    if not (torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1])): bad["hist"] += 1
    if not torch.equal(cur[2], ref[2]): bad["cap"] += 1
    if cache is not None and not torch.equal(cur[3], ref[3]):
        bad["cache"] += 1
        d = (cur[3] != ref[3]).nonzero().flatten()
        print("  iter", i, "cache bytes differ:", d.numel(), "first", d[:6].tolist(), "values", cur[3][d[:6]].tolist(), ref[3][d[:6]].tolist())
    if not torch.equal(cur[4], ref[4]):
        bad["ap"] += 1
        d = (cur[4] != ref[4]).nonzero().flatten()
        print("  iter", i, "ap differs at queries", d[:8].tolist(), (cur[4][d[:4]] - ref[4][d[:4]]).tolist())
    if cur[5] != ref[5]: bad["map"] += 1
print("iterations", it, "differences", bad)
