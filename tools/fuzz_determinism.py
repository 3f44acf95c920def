#!/usr/bin/env python3
"""Random scan shapes, each evaluated several times on the same object: histograms, divisors, AP sums, mAP (and mAP@k) must be bit-identical
from one evaluation to the next (a difference = a race or an unspaced hazard on some path; round 4 found one this way).
    python tools/fuzz_determinism.py [shapes] [evaluations]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
import numpy as np
import torch
from xmh import retrieval as xr
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = np.random.default_rng(4242)
bad = 0
for case in range(n):
    K = int(rng.choice([16, 32, 33, 48, 64, 64, 96, 100, 128, 128, 160, 256, 512]))
    Q = int(rng.choice([3, 16, 17, 65, 127, 200, 513, 1025]))
    R = int(rng.integers(2, 20000)) if case % 3 else int(rng.choice([63, 64, 65, 255, 256, 257, 4096, 8191, 8193, 32769]))
    C = int(rng.choice([1, 5, 24, 33, 64, 80, 128]))
    tern = case % 11 == 0 and K <= 256
    gen = torch.Generator().manual_seed(9000 + case)
    qB, rB = torch.randn(Q, K, generator=gen).sign(), torch.randn(R, K, generator=gen).sign()
    qB[qB == 0] = 1; rB[rB == 0] = 1
    if tern:
        rB[torch.rand(R, K, generator=gen) < 0.1] = 0
    if case % 4 == 0 and R > 8:
        rB = rB[torch.randint(0, int(rng.integers(1, 12)), (R,), generator=gen)]
    qL, rL = (torch.rand(Q, C, generator=gen) < 0.1).long(), (torch.rand(R, C, generator=gen) < 0.1).long()
    qL[:, 0] = 1; rL[0, 0] = 1
    k = None if case % 2 else int(rng.integers(1, 200))
    scan = xr.RankingScan(xr.pack_sign(qB.cuda()), xr.pack_labels(qL.cuda()), xr.pack_sign(rB.cuda()), xr.pack_labels(rL.cuda()), C)
    first = None
    for i in range(reps):
        ha, hr = scan.histograms(True)
        m, ap, cap = scan.map_all(k)
        cur = (ha.clone(), hr.clone(), cap.clone(), ap.clone(), m.clone())
        if first is None:
            first = cur
        elif not all(torch.equal(x, y) for x, y in zip(cur, first)):
            bad += 1
            which = [name for name, x, y in zip(("hist_all", "hist_rel", "cap", "ap", "map"), cur, first) if not torch.equal(x, y)]
            print("NONDETERMINISTIC", case, (Q, R, K, C), "ternary" if tern else "binary", "k", k, "evaluation", i, which, flush=True)
            break
print("shapes %d, nondeterministic %d" % (n, bad))
sys.exit(1 if bad else 0)
