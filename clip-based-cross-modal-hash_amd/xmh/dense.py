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


def map_k_float(qB, rB, qlab, rlab, C: int, k: Optional[int], why: Optional[str] = None) -> torch.Tensor:
    """calc_map_k the way the reference computes it (common/calc_utils.py:72-89) for code sets the bit-packed scan has no kernel for:
    un-quantised float "codes" (UMoED-style tanh outputs), ternary codes above 256 bits, more than 2048 bits, more than 256 classes.
    xmh_gemm_f32_sort_map: exact-fp32 GEMM distances, one stable radix sort per query row, one AP pass -- query tiles of at most 1.5 GB
    (20 bytes per pair) inside the C call.  ``qlab`` / ``rlab``: packed label masks of any word count."""
    global _warned_float
    if not _warned_float:
        import warnings
        warnings.warn("xmh: %s: using the float ranking path (fp32 GEMM + one radix sort per query), slower than the bit-packed scan"
                      % (why or "codes contain values outside {-1,0,+1}"))
        _warned_float = True
    qB, rB = qB.contiguous(), rB.contiguous()
    qlab, rlab = qlab.contiguous(), rlab.contiguous()
    Q, Rn, K = qB.shape[0], rB.shape[0], qB.shape[1]
    if rB.shape[1] != K or qlab.shape[0] != Q or rlab.shape[0] != Rn or qlab.shape[1] != rlab.shape[1] or qlab.shape[1] != (C + 31) // 32:
        raise ValueError("map_k_float: shapes %s %s / labels %s %s do not fit %d classes"
                         % (tuple(qB.shape), tuple(rB.shape), tuple(qlab.shape), tuple(rlab.shape), C))
    dev = qB.device
    ap = torch.empty(Q, dtype=torch.float64, device=dev)
    cap = torch.empty(Q, dtype=torch.int32, device=dev)
    out = torch.empty(1, dtype=torch.float64, device=dev)
    need = int(lib.xmh_gemm_f32_sort_ws_bytes(Q, Rn))
    free, _ = torch.cuda.mem_get_info(dev)
    free += torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
    need = max(min(need, int(0.8 * free)), 20 * Rn + 4096)       # a smaller workspace is a smaller query tile, not an error
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    check(lib.xmh_gemm_f32_sort_map(ptr(qB), ptr(rB), ptr(qlab), ptr(rlab), Q, Rn, K, C, 0 if k is None else int(k), ptr(ws), need,
                                    ptr(ap), ptr(cap), ptr(out), current_stream()), "xmh_gemm_f32_sort_map")
    return out


def float_sort_ap(dist: torch.Tensor, qlab: torch.Tensor, rlab: torch.Tensor, C: int, k: Optional[int] = None):
    """the ranking half of map_k_float on a distance matrix the caller holds (float32 [Q, R], any values): stable ascending sort per
    row, sum(ordinal / rank) over the first min(n_rel, k) relevant items -> (ap_sum float64 [Q], cap int32 [Q])."""
    dist, qlab, rlab = dist.contiguous(), qlab.contiguous(), rlab.contiguous()
    Q, Rn = dist.shape
    ap = torch.empty(Q, dtype=torch.float64, device=dist.device)
    cap = torch.empty(Q, dtype=torch.int32, device=dist.device)
    ws = torch.empty(16 * Q * Rn + 2048, dtype=torch.uint8, device=dist.device)
    check(lib.xmh_float_sort_ap(ptr(dist), ptr(qlab), ptr(rlab), Q, Rn, C, 0 if k is None else int(k), ptr(ws), ws.numel(), ptr(ap), ptr(cap),
                                current_stream()), "xmh_float_sort_ap")
    return ap, cap


_warned_float = False
