#!/usr/bin/env python3
"""One-off extended fuzz of xmh_hamming_topk against the C oracle (run on the GPU box): bit-exact (distance, index) lists.
    python tools/fuzz_topk_extended.py [cases] [ternary]
With a second argument the cases carry exact zeros (round 6: xmh_hamming_topk_ternary against orc_topk_ternary, half-unit distances)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
import numpy as np
import torch
from oracle import c_oracle as co
from xmh import retrieval as xr

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ternary = len(sys.argv) > 2
rng = np.random.default_rng(4242 + (1 if ternary else 0))
bad = 0
for case in range(n):
    K = int(rng.choice([8, 16, 32, 48, 64, 96, 128, 256, 512, 1024, 2048]))
    Q = int(rng.choice([1, 2, 3, 4, 7, 8, 9, 15, 16, 17, 33, 64, 65]))
    R = int(rng.integers(1, 60000)) if case % 2 else int(rng.choice([1, 2, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 65536, 100001]))
    k = int(rng.choice([1, 2, 5, 10, 100, 500, 1000]))
    W = (K + 31) // 32
    qb = rng.integers(0, 2**32, size=(Q, W), dtype=np.uint32)
    rb = rng.integers(0, 2**32, size=(R, W), dtype=np.uint32)
    if K % 32:
        qb[:, -1] &= (1 << (K % 32)) - 1
        rb[:, -1] &= (1 << (K % 32)) - 1
    mode = case % 5
    if mode == 0 and R > 16:
        rb = rb[rng.integers(0, int(rng.integers(1, 9)), size=R)]              # heavy ties
    elif mode == 1 and R > 4:
        rb[: R // 2] = qb[0]                                                    # many exact matches of query 0
    base = int(rng.integers(0, 1 << 22))
    if ternary:
        pz = float(rng.choice([0.003, 0.05, 0.15, 0.5]))
        def plane(bits):
            z = (rng.random((bits.shape[0], W * 32)) < pz)
            z[:, K:] = True                                                     # padding bits set, as xmh_pack_sign leaves them
            zw = np.packbits(z.reshape(-1, W, 4, 8)[:, :, :, ::-1], axis=-1).reshape(-1, W, 4)
            zw = (zw[:, :, 0].astype(np.uint32) | (zw[:, :, 1].astype(np.uint32) << 8) | (zw[:, :, 2].astype(np.uint32) << 16) | (zw[:, :, 3].astype(np.uint32) << 24))
            return zw
        qz, rz = plane(qb), plane(rb)
        if mode == 0 and R > 16:
            rz = rz[rng.integers(0, int(rng.integers(1, 9)), size=R)]
        qb, rb = qb & ~qz, rb & ~rz                                             # a zero element has its sign bit clear
        t = lambda a: torch.from_numpy(a.view(np.int32)).cuda()
        try:
            d, i = xr.hamming_topk(xr.PackedCodes(t(qb), t(qz), K), xr.PackedCodes(t(rb), t(rz), K), k, base)
        except Exception as exc:                                                # 4097 buckets x 8 queries + k = 1000 candidates: beyond one block's LDS, said so
            assert "LDS" in str(exc) and K >= 1024, exc
            continue
        wd, wi = co.topk_ternary(qb, qz, rb, rz, K, k, base)
        if not (np.array_equal(i.cpu().numpy(), wi) and np.array_equal(d.cpu().numpy().view(np.uint16), wd)):
            bad += 1
            print("MISMATCH ternary", case, Q, R, K, k, mode, pz, flush=True)
        continue
    q = xr.PackedCodes(torch.from_numpy(qb.view(np.int32)).cuda(), None, K)
    r = xr.PackedCodes(torch.from_numpy(rb.view(np.int32)).cuda(), None, K)
    d, i = xr.hamming_topk(q, r, k, base)
    wd, wi = co.topk(qb, rb, K + 1, k, base)
    if not (np.array_equal(i.cpu().numpy(), wi) and np.array_equal(d.cpu().numpy().view(np.uint16), wd)):
        bad += 1
        print("MISMATCH", case, Q, R, K, k, mode, flush=True)
print("cases %d, mismatches %d" % (n, bad))
sys.exit(1 if bad else 0)
