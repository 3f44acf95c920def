"""calc_utils on un-quantised float inputs: GEMM + row-wise kernels from libxmh.so."""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from . import retrieval as R
from ._lib import check, current_stream, lib, ptr


def _sqnorm(x: torch.Tensor) -> torch.Tensor:
    n = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    check(lib.xmh_row_l2normalize(ptr(x), x.shape[0], x.shape[1], None, ptr(n), current_stream()), "xmh_row_l2normalize")
    return n


def _normalize(x: torch.Tensor) -> torch.Tensor:
    y = torch.empty_like(x)
    check(lib.xmh_row_l2normalize(ptr(x), x.shape[0], x.shape[1], ptr(y), None, current_stream()), "xmh_row_l2normalize")
    return y


def cosine(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    a, b = a.contiguous(), b.contiguous()
    return ops.gemm_nt(_normalize(a), _normalize(b), precision=ops.PREC_F32X)


def pairwise_l2(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    a, b = a.contiguous(), b.contiguous()
    g = ops.gemm_nt(a, b, precision=ops.PREC_F32X)
    na, nb = _sqnorm(a), _sqnorm(b)          # keep both alive until the launch is enqueued (a freed temporary's block can be re-used)
    check(lib.xmh_pairwise_l2_from_gram(ptr(g), ptr(na), ptr(nb), a.shape[0], b.shape[0], current_stream()), "xmh_pairwise_l2_from_gram")
    return g


def hamming_dist_float(B1: torch.Tensor, B2: torch.Tensor) -> torch.Tensor:
    """0.5 * (K - B1 @ B2^T) for arbitrary float 'codes'."""
    B1, B2 = B1.contiguous(), B2.contiguous()
    g = ops.gemm_nt(B1, B2, precision=ops.PREC_F32X)
    check(lib.xmh_affine_inplace(ptr(g), g.numel(), -0.5, 0.5 * B2.shape[1], current_stream()), "xmh_affine_inplace")
    return g


_FLOAT_TILE_BYTES = 1 << 30        # the [q-tile, R] fp32 distance block of the float path stays under 1 GiB


def map_k_float(qB, rB, qlab, rlab, C: int, k: Optional[int]) -> torch.Tensor:
    """calc_map_k on un-quantised float "codes": distances by exact-fp32 GEMM, ranks by comparison counting.  The queries are
    tiled so that at most 1 GiB of distances exists at a time (Q=5000 x R=117k would be 2.3 GB at once); the ranking itself
    is O(nrel * R) per query -- this is the slow path, and the caller is told so once."""
    global _warned_float
    if not _warned_float:
        import warnings
        warnings.warn("xmh: codes contain values outside {-1,0,+1}: using the float ranking path (GEMM + comparison counting), "
                      "orders of magnitude slower than the bit-packed scan; quantise the codes (make_hash_code) to avoid it")
        _warned_float = True
    qB, rB = qB.contiguous(), rB.contiguous()
    Q, Rn = qB.shape[0], rB.shape[0]
    ap = torch.empty(Q, dtype=torch.float64, device=qB.device)
    cap = torch.empty(Q, dtype=torch.int32, device=qB.device)
    tile = max(1, min(Q, _FLOAT_TILE_BYTES // (4 * max(Rn, 1))))
    for lo in range(0, Q, tile):
        hi = min(Q, lo + tile)
        d = hamming_dist_float(qB[lo:hi], rB)
        check(lib.xmh_float_rank_ap(ptr(d), ptr(qlab[lo:hi]), ptr(rlab), hi - lo, Rn, C, 0 if k is None else int(k), ptr(ap[lo:hi]),
                                    ptr(cap[lo:hi]), current_stream()), "xmh_float_rank_ap")
    return R.map_finalize(ap, cap)


_warned_float = False
