"""diagnostic: which pair-cache entries of pass 1 differ from the oracle (run on the GPU box)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
from xmh import retrieval as xr
from xmh._lib import lib
from oracle import retrieval as orc
Q, R, K, C = (int(x) for x in (sys.argv[1:5] or (130, 6000, 64, 80)))
g = torch.Generator().manual_seed(5)
qB, rB = torch.randn(Q, K, generator=g).sign(), torch.randn(R, K, generator=g).sign()
qL, rL = (torch.rand(Q, C, generator=g) < .1).long(), (torch.rand(R, C, generator=g) < .1).long()
q, r = xr.pack_sign(qB.cuda()), xr.pack_sign(rB.cuda())
ql, rl = xr.pack_labels(qL.cuda()), xr.pack_labels(rL.cuda())
u32 = lambda t: t.cpu().numpy().view(np.uint32)
scan = xr.RankingScan(q, ql, r, rl, C)
nbytes = int(lib.xmh_scan_pair_cache_bytes(Q, R, K, 0)); off = int(lib.xmh_scan_pair_cache_offset(Q, R, K, 0))
scan.ws.zero_(); scan.histograms(False); torch.cuda.synchronize()
raw = scan.ws[off:off + nbytes].cpu().numpy()
dist = orc.hamming_packed(u32(q.bits), u32(r.bits)).astype(np.int64); rel = orc.relevance_packed(u32(ql), u32(rl)).astype(np.int64)
want = (dist << 1) | rel
pl = scan.plan; nbatch = (pl.chunk + 63) // 64
print("plan chunk", pl.chunk, "nchunk", pl.nchunk, "qpad", pl.qpad, "nbatch", nbatch)
got = raw.reshape(pl.nchunk, pl.qpad // 16, nbatch, 64, 16).astype(np.int64)
lane = np.arange(64); slot, qin = lane // 16, lane % 16; t = np.arange(16)
nprinted = 0
for c in range(pl.nchunk):
    lo, hi = c * pl.chunk, min((c + 1) * pl.chunk, R)
    item = lo + 64 * np.arange(nbatch)[:, None, None] + 4 * t[None, None, :] + slot[None, :, None]
    for tile in range(pl.qpad // 16):
        qq = tile * 16 + qin
        m = (item < hi) & (qq < Q)[None, :, None]
        w = want[np.minimum(qq, Q - 1)[None, :, None], np.minimum(item, R - 1)]
        bad = (got[c, tile] != w) & m
        if bad.any(): print("chunk", c, "tile", tile, "bad", int(bad.sum()), "of", int(m.sum()))
        if bad.any() and nprinted < 3:
            nprinted += 1
            print("  bad per batch:", bad.sum((1, 2))[:12])
            print("  bad per entry t:", bad.sum((0, 1)))
            print("  bad per slot:", [int(bad[:, slot == s_].sum()) for s_ in range(4)], " per query-in-tile:", [int(bad[:, qin == x].sum()) for x in range(16)])
            b = np.argwhere(bad)[0]
            bb = b[0]
            print("  first bad batch", bb, "got row lane0:", got[c, tile, bb, 0], "want:", w[bb, 0])
            # does the got block equal the wanted block of another batch?
            for ob in range(nbatch):
                if np.array_equal(got[c, tile, bb][m[bb]], w[ob][m[bb]]): print("  -> equals wanted batch", ob)
            d_got, d_w = got[c, tile, bb] >> 1, w[bb] >> 1
            print("  rel bits equal:", np.array_equal(got[c, tile, bb] & 1, w[bb] & 1), " d diff stats:", np.unique(d_got - d_w, return_counts=True))
