"""diagnostic: pass-1 histograms of the TERNARY k_scan_hist_b instances against an exact integer restatement, per bucket (run on the GPU box):
    python tools/diag_bits_ternary.py [Q R K C zero_fraction]
Distances in half units: K - q.r over {-1, 0, +1} codes (reference common/calc_utils.py:51-56 on sign_() outputs, runners/base.py:407-410)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
from xmh import retrieval as xr
from oracle import retrieval as orc
args = sys.argv[1:]
Q, R, K, C = (int(x) for x in (args[:4] or (200, 70000, 256, 80)))
pz = float(args[4]) if len(args) > 4 else 0.02
g = torch.Generator().manual_seed(7)
qB, rB = torch.randn(Q, K, generator=g).sign(), torch.randn(R, K, generator=g).sign()
qB[torch.rand(qB.shape, generator=g) < pz] = 0.0
rB[torch.rand(rB.shape, generator=g) < pz] = 0.0
qL, rL = (torch.rand(Q, C, generator=g) < .1).long(), (torch.rand(R, C, generator=g) < .1).long()
q, r = xr.pack_sign(qB.cuda()), xr.pack_sign(rB.cuda())
ql, rl = xr.pack_labels(qL.cuda()), xr.pack_labels(rL.cuda())
scan = xr.RankingScan(q, ql, r, rl, C)
ha, hr = scan.histograms(True)
ha, hr = ha.cpu().numpy().astype(np.int64), hr.cpu().numpy().astype(np.int64)
d2 = (K - (qB.double() @ rB.double().T)).round().long().numpy()                     # [Q, R] half units, exact
rel = ((qL.double() @ rL.double().T) > 0).long().numpy()
W = (K + 31) // 32
Wp = 1 << (W - 1).bit_length()                                                       # the scan runs 1, 2, 4, 8 code words: other lengths are widened,
shift = 32 * Wp - K if Wp != W else 0                                                             # the padding elements count as zeros: every distance grows by their number
d2 = d2 + shift
nb = 2 * 32 * Wp + 1
wa = np.stack([np.bincount(d2[i], minlength=nb) for i in range(Q)])
wr = np.stack([np.bincount(d2[i], weights=rel[i], minlength=nb).astype(np.int64) for i in range(Q)])
nb = min(nb, ha.shape[1])
wa, wr, ha, hr = wa[:, :nb], wr[:, :nb], ha[:, :nb], hr[:, :nb]
print("Q %d R %d K %d C %d zeros %.3f: bad all cells %d of %d; bad rel cells %d; plan %s" % (Q, R, K, C, pz, int((ha != wa).sum()), ha.size, int((hr != wr).sum()), scan.describe() if hasattr(scan, "describe") else ""))
