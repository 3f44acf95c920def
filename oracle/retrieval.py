"""Oracle (CPU restatement) of the reference's retrieval maths.  TEST INFRASTRUCTURE ONLY.

Restates ``common/calc_utils.py`` and the two ``make_hash_code`` variants of the
reference.  Two independent formulations of mAP are kept on purpose:

* :func:`map_k` follows the reference step by step (float GEMM distances,
  integer label matmul, full row sort, per-query loop) -- it is what
  ``bench.py`` times as ``cpu_baseline`` (kind "port").
* :func:`map_k_ranked` is the integer formulation the HIP kernels implement
  (bit-packed XOR/popcount distances, (distance, index) ranking through bucket
  histograms, no sort).  ``tests/`` check both against the golden vectors and
  against each other.

Pinned by ``tests/golden/calc_utils_*.npz`` (generated from the imported
reference by ``oracle/make_golden_retrieval.py``).
"""
from __future__ import annotations

import numpy as np
import torch


# --------------------------------------------------------------------------
# common/calc_utils.py restatements
# --------------------------------------------------------------------------
def hamming_dist(B1: torch.Tensor, B2: torch.Tensor) -> torch.Tensor:
    """reference common/calc_utils.py:51-56 -- ``0.5 * (K - B1 @ B2^T)``; a 1-D B1 is one query."""
    K = B2.shape[1]
    if B1.dim() < 2:
        B1 = B1[None, :]
    return 0.5 * (K - B1.mm(B2.t()))


def label_sim(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """reference common/calc_utils.py:8-10 -- 1.0 where two items share a label."""
    return (a.matmul(b.t()) > 0).float()


def cosine_sim(a, b):
    """reference common/calc_utils.py:38-49 -- row-normalise (no eps) then a @ b^T."""
    if isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor):
        return torch.matmul(a / a.norm(dim=-1, keepdim=True), (b / b.norm(dim=-1, keepdim=True)).t())
    if isinstance(a, np.ndarray) and isinstance(b, np.ndarray):
        return (a / np.linalg.norm(a, axis=-1, keepdims=True)) @ (b / np.linalg.norm(b, axis=-1, keepdims=True)).T
    raise ValueError("input value must in [torch.Tensor, numpy.ndarray], but it is %s, %s" % (type(a), type(b)))


def euclid_sim(a, b):
    """reference common/calc_utils.py:28-36 -- pairwise L2 distance (torch.cdist / sklearn)."""
    if isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor):
        return torch.cdist(a, b, p=2.0)
    if isinstance(a, np.ndarray) and isinstance(b, np.ndarray):
        aa = (a * a).sum(-1)[:, None]
        bb = (b * b).sum(-1)[None, :]
        return np.sqrt(np.maximum(aa + bb - 2.0 * (a @ b.T), 0.0))
    raise ValueError("input value must in [torch.Tensor, numpy.ndarray], but it is %s, %s" % (type(a), type(b)))


def map_k(qB, rB, query_L, retrieval_L, k=None, stable: bool = True) -> torch.Tensor:
    """reference common/calc_utils.py:58-92, step for step.

    ``stable=True`` ranks ties by gallery index (the build's canonical order);
    ``stable=False`` uses torch's default sort like the reference does.
    Quirks kept: NaN when a query has no relevant item (mean of empty), IndexError for
    Q == 1 (the squeeze at :72), ``k`` caps the number of *relevant* items averaged (:81).
    """
    Q = query_L.shape[0]
    if isinstance(qB, torch.Tensor) and qB.is_cuda:
        qB, rB = qB.cpu(), rB.cpu()
    if query_L.device != qB.device:
        query_L, retrieval_L = query_L.to(qB.device), retrieval_L.to(qB.device)
    if k is None:
        k = retrieval_L.shape[0]
    rel = (query_L.mm(retrieval_L.t()) > 0).squeeze().to(torch.float32)            # :72
    n_rel = rel.sum(dim=-1, keepdim=True, dtype=torch.int32)                        # :75
    order = torch.sort(hamming_dist(qB, rB), dim=-1, stable=stable).indices         # :76-77
    cap = torch.minimum(n_rel, torch.full_like(n_rel, k))                           # :81
    acc = 0
    for i in range(Q):                                                              # :84-89
        hits = rel[i][order[i]]
        n = cap[i].squeeze()
        ordinal = torch.arange(1, n + 1).to(torch.float32)
        rank = torch.nonzero(hits)[:n].squeeze().to(torch.float32) + 1.0
        acc = acc + torch.mean(ordinal / rank)
    return acc / Q


def map_k_tie_bounds(qB, rB, query_L, retrieval_L, k=None):
    """(lowest, highest) mAP any tie order can give: the reference's default torch.sort (calc_utils.py:77) leaves the order of
    equal distances unspecified, so ITS value -- and the canonical (distance, index) value -- must lie between the
    ranking that puts every irrelevant item of a distance bucket first and the one that puts every relevant item first."""
    Q = query_L.shape[0]
    if k is None:
        k = retrieval_L.shape[0]
    rel = (query_L.to(torch.float32).mm(retrieval_L.to(torch.float32).t()) > 0)
    d = hamming_dist(qB.cpu(), rB.cpu())
    out = []
    for sign in (+1.0, -1.0):                                   # worst case: relevant last within a tie; best: relevant first
        key = d.to(torch.float64) * 4.0 + sign * rel.to(torch.float64)
        order = torch.sort(key, dim=-1, stable=True).indices
        acc = 0.0
        for i in range(Q):
            hits = rel[i][order[i]]
            n = int(min(int(hits.sum()), k))
            rank = torch.nonzero(hits)[:n].squeeze(-1).to(torch.float64) + 1.0
            acc += float((torch.arange(1, n + 1, dtype=torch.float64) / rank).mean()) if n else float("nan")
        out.append(acc / Q)
    return out[0], out[1]


# --------------------------------------------------------------------------
# make_hash_code variants (quantisers)
# --------------------------------------------------------------------------
def hash_code_sign(code: torch.Tensor) -> torch.Tensor:
    """reference runners/base.py:407-410 -- in-place sign: -1 / 0 / +1."""
    return code.sign_()


def hash_code_pair_argmax(code: torch.Tensor) -> torch.Tensor:
    """reference runners/DCMHT/runner.py:82-95 -- [B,2K] -> [B,K]: +1 iff p1 > p0 strictly, else -1."""
    if code.dim() < 3:
        code = code.view(code.shape[0], -1, 2)
    win = torch.argmax(code, dim=-1)
    return torch.where(win == 0, -torch.ones_like(win), win).float()


# --------------------------------------------------------------------------
# integer formulation (what the HIP kernels compute)
# --------------------------------------------------------------------------
def pack_bits(code: np.ndarray):
    """[N,K] of {-1,0,+1} -> (bits[N,W] u32 with bit j%32 of word j//32 set iff code>0,
    zero_mask[N,W] set iff code==0).  W = ceil(K/32); padding bits are 0 in both."""
    code = np.asarray(code)
    N, K = code.shape
    W = (K + 31) // 32
    pos = np.zeros((N, W * 32), dtype=np.uint64)
    zer = np.zeros((N, W * 32), dtype=np.uint64)
    pos[:, :K] = code > 0
    zer[:, :K] = code == 0
    weights = (np.uint64(1) << np.arange(32, dtype=np.uint64))[None, None, :]
    bits = (pos.reshape(N, W, 32) * weights).sum(-1).astype(np.uint32)
    zmask = (zer.reshape(N, W, 32) * weights).sum(-1).astype(np.uint32)
    return bits, zmask


def pack_labels(L: np.ndarray) -> np.ndarray:
    """[N,C] multi-hot (any dtype, >0 means set) -> [N,ceil(C/32)] u32 masks."""
    L = np.asarray(L)
    N, C = L.shape
    Lw = (C + 31) // 32
    on = np.zeros((N, Lw * 32), dtype=np.uint64)
    on[:, :C] = L > 0
    weights = (np.uint64(1) << np.arange(32, dtype=np.uint64))[None, None, :]
    return (on.reshape(N, Lw, 32) * weights).sum(-1).astype(np.uint32)


_POP8 = np.array([bin(i).count("1") for i in range(256)], dtype=np.uint16)


def popcount32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32)
    return (_POP8[x & 0xFF] + _POP8[(x >> 8) & 0xFF] + _POP8[(x >> 16) & 0xFF] + _POP8[(x >> 24) & 0xFF])


def hamming_packed(qbits: np.ndarray, rbits: np.ndarray) -> np.ndarray:
    """[Q,W],[R,W] u32 -> [Q,R] u16 popcount(q xor r).  Equals calc_hammingDist for +-1 codes."""
    Q, W = qbits.shape
    out = np.zeros((Q, rbits.shape[0]), dtype=np.uint16)
    for w in range(W):
        out += popcount32(qbits[:, w][:, None] ^ rbits[:, w][None, :])
    return out


def hamming2_ternary(qbits, qzero, rbits, rzero, K: int) -> np.ndarray:
    """Ternary codes (sign() can give 0, runners/base.py:410): returns 2*distance =
    K - q.r as an integer, with q.r = #agree - #disagree over positions where both are non-zero."""
    Q, W = qbits.shape
    dot = np.zeros((Q, rbits.shape[0]), dtype=np.int32)
    for w in range(W):
        live = ~(qzero[:, w][:, None] | rzero[:, w][None, :])
        if w == W - 1 and K % 32:
            live &= np.uint32((1 << (K % 32)) - 1)
        diff = (qbits[:, w][:, None] ^ rbits[:, w][None, :]) & live
        dot += popcount32(live).astype(np.int32) - 2 * popcount32(diff).astype(np.int32)
    return (K - dot).astype(np.int32)


def relevance_packed(qlab: np.ndarray, rlab: np.ndarray) -> np.ndarray:
    """[Q,Lw],[R,Lw] u32 -> [Q,R] bool: share at least one label (== (qL @ rL^T > 0), calc_utils.py:72)."""
    hit = np.zeros((qlab.shape[0], rlab.shape[0]), dtype=bool)
    for w in range(qlab.shape[1]):
        hit |= (qlab[:, w][:, None] & rlab[:, w][None, :]) != 0
    return hit


def bucket_histograms(dist: np.ndarray, rel: np.ndarray, nbuckets: int):
    """per query: hist_all[d], hist_rel[d] (u32) -- what xmh_hamming_hist returns."""
    Q = dist.shape[0]
    ha = np.zeros((Q, nbuckets), dtype=np.uint32)
    hr = np.zeros((Q, nbuckets), dtype=np.uint32)
    for q in range(Q):
        ha[q] = np.bincount(dist[q], minlength=nbuckets)
        hr[q] = np.bincount(dist[q][rel[q]], minlength=nbuckets)
    return ha, hr


def ap_from_ranking(dist: np.ndarray, rel: np.ndarray, k=None, base_all=None, base_rel=None,
                    n_rel_total=None) -> np.ndarray:
    """Per-query sum_j (j / rank_j) over the first min(n_rel,k) relevant items under the
    canonical (distance asc, gallery index asc) order, computed WITHOUT a sort:
    rank = (#items in lower buckets) + (#same-bucket items with smaller index) + 1.
    ``base_*`` [Q,nb] are optional extra offsets per bucket (lower-ranked shards);
    returns float64 partial sums [Q] (not yet divided by the cap)."""
    Q, R = dist.shape
    nb = int(dist.max()) + 1 if base_all is None else base_all.shape[1]
    out = np.zeros(Q, dtype=np.float64)
    for q in range(Q):
        d = dist[q].astype(np.int64)
        r = rel[q]
        ha = np.bincount(d, minlength=nb)
        hr = np.bincount(d[r], minlength=nb)
        if base_all is None:
            ba = np.concatenate([[0], np.cumsum(ha)[:-1]])
            br = np.concatenate([[0], np.cumsum(hr)[:-1]])
            ntot = int(hr.sum())
        else:
            ba, br = base_all[q].astype(np.int64), base_rel[q].astype(np.int64)
            ntot = int(n_rel_total[q])
        cap = ntot if k is None else min(ntot, k)
        order = np.argsort(d, kind="stable")
        ds = d[order]
        first = np.concatenate([[0], np.cumsum(ha)[:-1]])          # local bucket start in sorted order
        pos_in_bucket = np.arange(R) - first[ds]
        rs = r[order]
        rel_before = np.cumsum(rs) - rs                                # local relevant before (sorted order)
        rel_first = np.concatenate([[0], np.cumsum(hr)[:-1]])
        rel_in_bucket = rel_before - rel_first[ds]
        rank = ba[ds] + pos_in_bucket + 1
        ordinal = br[ds] + rel_in_bucket + 1
        take = rs & (ordinal <= cap)
        out[q] = (ordinal[take] / rank[take]).sum()
    return out


def map_k_ranked(qB, rB, query_L, retrieval_L, k=None) -> float:
    """mAP through the integer formulation (binary +-1 codes).  NaN if any query has no
    relevant item, exactly like the reference (calc_utils.py:87-89)."""
    qb, _ = pack_bits(np.asarray(qB))
    rb, _ = pack_bits(np.asarray(rB))
    dist = hamming_packed(qb, rb)
    rel = relevance_packed(pack_labels(np.asarray(query_L)), pack_labels(np.asarray(retrieval_L)))
    K = np.asarray(qB).shape[1]
    part = ap_from_ranking(dist, rel, k=k, base_all=None)
    n_rel = rel.sum(-1)
    cap = n_rel if k is None else np.minimum(n_rel, k)
    with np.errstate(invalid="ignore", divide="ignore"):
        ap = part / cap
    del K
    return float(ap.mean())
