"""diagnostic: pass-1 histograms of k_scan_hist_b against the oracle's, per bucket (run on the GPU box): python tools/diag_bits.py Q R K C"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
from xmh import retrieval as xr
from oracle import retrieval as orc
Q, R, K, C = (int(x) for x in (sys.argv[1:5] or (40, 3000, 256, 80)))
g = torch.Generator().manual_seed(5)
qB, rB = torch.randn(Q, K, generator=g).sign(), torch.randn(R, K, generator=g).sign()
qL, rL = (torch.rand(Q, C, generator=g) < .1).long(), (torch.rand(R, C, generator=g) < .1).long()
q, r = xr.pack_sign(qB.cuda()), xr.pack_sign(rB.cuda())
ql, rl = xr.pack_labels(qL.cuda()), xr.pack_labels(rL.cuda())
u32 = lambda t: t.cpu().numpy().view(np.uint32)
scan = xr.RankingScan(q, ql, r, rl, C)
ha, hr = scan.histograms(True)
ha, hr = ha.cpu().numpy().astype(np.int64), hr.cpu().numpy().astype(np.int64)
dist = orc.hamming_packed(u32(q.bits), u32(r.bits)).astype(np.int64); rel = orc.relevance_packed(u32(ql), u32(rl)).astype(np.int64)
wa = np.stack([np.bincount(dist[i], minlength=K + 1) for i in range(Q)])
wr = np.stack([np.bincount(dist[i], weights=rel[i], minlength=K + 1).astype(np.int64) for i in range(Q)])
print("all: row sums got", ha.sum(1)[:6], "want", wa.sum(1)[:6])
bad = np.argwhere(ha != wa)
print("bad all cells", len(bad), "of", ha.size, "; bad rel cells", int((hr != wr).sum()))
for (i, d) in bad[:12]:
    print(" q", i, "bucket", d, "got", ha[i, d], "want", wa[i, d])
if len(bad):
    i = bad[0][0]
    print("query", i, "popcount", int(np.unpackbits(u32(q.bits)[i].view(np.uint8)).sum()))
    nz = np.nonzero(ha[i] - wa[i])[0]; print(" diff buckets", nz[:20], (ha[i] - wa[i])[nz][:20])
