"""the mAP scan on TERNARY codes (one exact 0.0 among the code elements selects these kernels) at the COCO shape, beside the binary scan:
    python tools/bench_ternary_scan.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
import torch
import bench
from oracle import retrieval as orc
from xmh import _lib, retrieval as R

def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for K in [int(a) for a in sys.argv[1:]] or (16, 64, 128):
    qB, qL, rB, rL = bench.synth(5000, 117218, K, 80, seed=1814, p=0.04)
    ql, rl = R.pack_labels(qL.cuda()), R.pack_labels(rL.cuda())
    out = {}
    for name, zero in (("binary", False), ("one zero", True), ("2% zeros", 0.02)):
        q2, r2 = qB.clone(), rB.clone()
        if zero is True:
            r2[777, 3] = 0.0
        elif zero:
            g = torch.Generator().manual_seed(3)
            r2[torch.rand(r2.shape, generator=g) < zero] = 0.0
            q2[torch.rand(q2.shape, generator=g) < zero] = 0.0
        q, r = R.pack_sign(q2.cuda()), R.pack_sign(r2.cuda())
        scan = R.RankingScan(q, ql, r, rl, 80)
        def step():
            scan.histograms(False)
            return scan.map_all(None)[0]
        ms = t(step)
        buf = ctypes.create_string_buffer(1024)
        _lib.check(_lib.lib.xmh_scan_describe(5000, 117218, K, 80, int(q.zero is not None or r.zero is not None), buf, 1024), "describe")
        sub = slice(0, 32)
        got = float(R.map_k_packed(R.pack_sign(q2[sub].cuda()), r, ql[sub].contiguous(), rl, 80).item())
        want = float(orc.map_k(q2[sub], r2, qL[sub], rL, stable=True))
        print("K=%3d %-9s %.3f ms per step   |subsample mAP - oracle| = %.1e   %s" % (K, name, ms, abs(got - want), buf.value.decode()), flush=True)
