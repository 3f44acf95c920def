#!/usr/bin/env python3
"""Stress of the hand-scheduled pass 1 (k_scan_hist_m2; run on the GPU box): its hazards were found on hardware, so it is run many
times at large shapes -- alone and with a GEMM hammering the chip from a second stream (other clocks, other wave interleavings) --
and every run must reproduce the bucket histograms of the round-2 kernel (XMH_SCAN_M2=0) bit for bit and its own AP sums bit for bit.
    python tools/stress_m2.py [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
import torch
import bench_roofline as RL
from xmh import retrieval as R

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 25
bad = 0
side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device="cuda", dtype=torch.float16)
for (Q, Rn, K, C, p) in ((5000, 117218, 64, 80, .04), (5000, 117218, 16, 24, .1), (3000, 60011, 48, 33, .04), (777, 200003, 32, 80, .02)):
    q, ql, r, rl = RL.synth_gpu(Q, Rn, K, C, seed=K + Q, p=p)
    os.environ["XMH_SCAN_M2"] = "0"                       # the plan (chunking, workspace) depends on it: a scan object of its own
    ref = R.RankingScan(q, ql, r, rl, C)
    ref_ha, ref_hr = [t.clone() for t in ref.histograms()]
    ref_ap, ref_cap = [t.clone() for t in ref.ap_sums(None)]
    os.environ.pop("XMH_SCAN_M2")
    del ref
    scan = R.RankingScan(q, ql, r, rl, C)
    first = None
    for it in range(rounds):
        load = it % 2 == 1
        if load:
            with torch.cuda.stream(side):
                for _ in range(6):
                    a @ a
        ha, hr = scan.histograms()
        ap, cap = scan.ap_sums(None)
        torch.cuda.synchronize()
        ok = torch.equal(ha, ref_ha) and torch.equal(hr, ref_hr) and torch.equal(cap, ref_cap) and torch.allclose(ap, ref_ap, rtol=2e-6, atol=1e-9)
        if first is None:
            first = ap.clone()
        ok = ok and torch.equal(ap, first)
        if not ok:
            bad += 1
            print("MISMATCH", (Q, Rn, K, C), "round", it, "loaded" if load else "alone", flush=True)
    print((Q, Rn, K, C), "rounds", rounds, "bad so far", bad, flush=True)
print("mismatches", bad)
sys.exit(1 if bad else 0)
