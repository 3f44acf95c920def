"""Bit-packed hash codes on the GPU and the retrieval ops over them.

Thin host wrappers: allocate outputs with torch, hand raw device pointers and the
current HIP stream to ``libxmh.so``.  No arithmetic happens here.

Reference counterparts (file:line under the reference repo):
  pack_sign / pack_pair_argmax  -> make_hash_code, runners/base.py:407-410, runners/DCMHT/runner.py:82-95
  hamming_dist                  -> calc_hammingDist, common/calc_utils.py:51-56
  map_k_packed                  -> calc_map_k, common/calc_utils.py:58-92
"""
from __future__ import annotations

import ctypes as C
import os
import warnings
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import check, current_stream, lib, ptr

_DT = {torch.float32: 0, torch.int64: 1, torch.int32: 2, torch.uint8: 3, torch.bool: 3}


def _require_cuda(*tensors):
    """libxmh launches on the current device's current stream: every operand must live there."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("xmh ops need CUDA/HIP tensors (got a %s tensor); there is no CPU fallback" % t.device)
        if t.device.index != torch.cuda.current_device():
            raise RuntimeError("xmh op on a %s tensor while cuda:%d is current: wrap the call in torch.cuda.device(%d) "
                               "or torch.cuda.set_device first" % (t.device, torch.cuda.current_device(), t.device.index))


def words(K: int) -> int:
    return (K + 31) // 32


@dataclass
class PackedCodes:
    """n hash codes of K bits.  ``bits`` int32 [n, W] (bit set <=> +1); ``zero`` int32 [n, W] or None
    (bit set <=> the code element is exactly 0, padding bits set)."""
    bits: torch.Tensor
    zero: Optional[torch.Tensor]
    K: int
    flags: int = 0          # bit0: an exact 0 was seen, bit1: a value outside {-1,0,+1} was seen (pack_sign)

    @property
    def n(self) -> int:
        return self.bits.shape[0]

    @property
    def ternary(self) -> bool:
        return self.zero is not None

    def rows(self, lo: int, hi: int) -> "PackedCodes":
        return PackedCodes(self.bits[lo:hi], None if self.zero is None else self.zero[lo:hi], self.K, self.flags)

    def unpack(self) -> torch.Tensor:
        """-> float32 [n, K] of -1/0/+1 (what BaseTrainer.get_code hands to calc_map_k / save_mat)."""
        out = torch.empty(self.n, self.K, dtype=torch.float32, device=self.bits.device)
        check(lib.xmh_unpack_pm1(ptr(self.bits), ptr(self.zero), self.n, self.K, ptr(out), current_stream()), "xmh_unpack_pm1")
        return out


_KERNEL_WORDS = (1, 2, 4, 8, 16, 32, 64)


def widened(p: PackedCodes) -> PackedCodes:
    """The ranking kernels are instantiated for power-of-two word counts.  A code of, say, 96 bits (3 words) is handed to
    them as a 128-bit code whose extra word is zero on both sides (and flagged "exact zero" in the zero plane): binary
    distances are unchanged, ternary half-unit distances all shift by the same constant, the ranking is the same."""
    W = p.bits.shape[1]
    if W in _KERNEL_WORDS or W > _KERNEL_WORDS[-1]:             # beyond 2048 bits the C side reports the limit
        return p
    Wp = next(w for w in _KERNEL_WORDS if w >= W)
    bits = torch.zeros(p.n, Wp, dtype=torch.int32, device=p.bits.device)
    bits[:, :W] = p.bits
    zero = None
    if p.zero is not None:
        zero = torch.full((p.n, Wp), -1, dtype=torch.int32, device=p.bits.device)
        zero[:, :W] = p.zero
    return PackedCodes(bits, zero, 32 * Wp, p.flags)


def empty_packed(n: int, K: int, device, with_zero: bool = False) -> PackedCodes:
    bits = torch.zeros(n, words(K), dtype=torch.int32, device=device)
    zero = None
    if with_zero:
        zero = torch.zeros(n, words(K), dtype=torch.int32, device=device)
    return PackedCodes(bits, zero, K)


def pack_sign(codes: torch.Tensor, out: Optional[PackedCodes] = None, row_index: Optional[torch.Tensor] = None,
              flags: Optional[torch.Tensor] = None, defer: bool = False) -> PackedCodes:
    """sign-quantise + pack float codes [n, K] (BaseTrainer.make_hash_code + the row scatter of get_code).

    Stand-alone (``out is None``): returns a fresh PackedCodes; one 4-byte D2H read of the value flags
    decides whether the zero plane is kept (``sign(0) = 0`` seen) and records ``.flags`` (bit1 = some
    element was not in {-1,0,+1}, i.e. the input was not yet quantised -- harmless here, sign is applied).
    Scatter mode (``out``/``row_index``): rows land in ``out`` at ``row_index``; ``flags`` (int32 [1] device
    tensor) accumulates the value flags without any sync.  ``defer=True`` (stand-alone, with ``flags``): no read
    either -- the zero plane stays and the caller settles several packs with ONE read (``settle_flags``)."""
    _require_cuda(codes, row_index, flags)
    codes = codes.contiguous().float()
    n, K = codes.shape
    if out is not None:
        if out.zero is None or out.K != K:
            raise ValueError("scatter target needs a zero plane and the same K")
        check(lib.xmh_pack_sign(ptr(codes), n, K, ptr(row_index), ptr(out.bits), ptr(out.zero), ptr(flags), current_stream()),
              "xmh_pack_sign")
        return out
    res = empty_packed(n, K, codes.device, with_zero=True)
    fl = torch.zeros(1, dtype=torch.int32, device=codes.device) if flags is None else flags
    check(lib.xmh_pack_sign(ptr(codes), n, K, ptr(row_index), ptr(res.bits), ptr(res.zero), ptr(fl), current_stream()),
          "xmh_pack_sign")
    if defer:
        if flags is None:
            raise ValueError("pack_sign(defer=True) needs the caller's flags tensor")
        return res
    res.flags = int(fl.item())
    if not (res.flags & 1):
        res.zero = None
    return res


def settle_flags(flags: torch.Tensor, *packed: PackedCodes) -> int:
    """ONE 4-byte D2H read for several deferred packs that accumulated into the same flags word: every code set takes the OR of the
    value flags (a zero seen in any of them keeps all the zero planes -- an all-live plane changes no result -- and any unquantised
    value sends the caller down its float path, which is what the per-set flags decided too)."""
    f = int(flags.item())
    for p in packed:
        p.flags = f
        if not (f & 1):
            p.zero = None
    return f


def pack_pair_argmax(probs: torch.Tensor, out: Optional[PackedCodes] = None,
                     row_index: Optional[torch.Tensor] = None) -> PackedCodes:
    """DCMHT quantiser: probs [n, 2K] -> K bits, bit = p[2j+1] > p[2j]."""
    _require_cuda(probs, row_index)
    probs = probs.contiguous().float()
    n, K2 = probs.shape
    if K2 % 2:
        raise ValueError("pair-argmax needs an even last dimension")
    K = K2 // 2
    if out is None:
        out = empty_packed(n, K, probs.device)
    check(lib.xmh_pack_pair_argmax(ptr(probs), n, K, ptr(row_index), ptr(out.bits), current_stream()), "xmh_pack_pair_argmax")
    return out


def pack_labels(L: torch.Tensor) -> torch.Tensor:
    """multi-hot labels [n, C] (float32 / int64 / int32 / uint8 / bool) -> int32 [n, ceil(C/32)]."""
    _require_cuda(L)
    if L.dtype not in _DT:
        L = L.to(torch.float32)
    L = L.contiguous()
    n, Cn = L.shape
    lab = torch.empty(n, words(Cn), dtype=torch.int32, device=L.device)
    check(lib.xmh_pack_labels(ptr(L), _DT[L.dtype], n, Cn, ptr(lab), current_stream()), "xmh_pack_labels")
    return lab


def zero_plane_or_default(p: PackedCodes) -> torch.Tensor:
    """the code's zero plane, or the plane of a code without zeros (only the padding bits set)."""
    if p.zero is not None:
        return p.zero
    z = torch.zeros_like(p.bits)
    pad = p.bits.shape[1] * 32 - p.K
    if pad:
        z[:, -1] = -1 << (32 - pad)                  # padding bits set (int32 two's complement)
    return z


def _both_planes(q: PackedCodes, r: PackedCodes):
    """zero planes for both sides or neither (the kernels' contract)."""
    if q.zero is None and r.zero is None:
        return None, None
    return zero_plane_or_default(q), zero_plane_or_default(r)


def hamming_dist(q: PackedCodes, r: PackedCodes, as_u16: bool = False) -> torch.Tensor:
    """[Q,R] distances: float32 0.5*(K - q.r) (reference semantics) or raw popcounts as int16 storage."""
    _require_cuda(q.bits, r.bits)
    if q.K != r.K:
        raise ValueError("code lengths differ: %d vs %d" % (q.K, r.K))
    qz, rz = _both_planes(q, r)
    Q, R = q.n, r.n
    dev = q.bits.device
    if as_u16:
        out = torch.empty(Q, R, dtype=torch.int16, device=dev)
        check(lib.xmh_hamming_dist(ptr(q.bits), ptr(qz), ptr(r.bits), ptr(rz), Q, R, q.K, None, ptr(out), current_stream()),
              "xmh_hamming_dist")
    else:
        out = torch.empty(Q, R, dtype=torch.float32, device=dev)
        check(lib.xmh_hamming_dist(ptr(q.bits), ptr(qz), ptr(r.bits), ptr(rz), Q, R, q.K, ptr(out), None, current_stream()),
              "xmh_hamming_dist")
    return out


def label_sim(qlab: torch.Tensor, rlab: torch.Tensor, Cn: int) -> torch.Tensor:
    _require_cuda(qlab, rlab)
    out = torch.empty(qlab.shape[0], rlab.shape[0], dtype=torch.float32, device=qlab.device)
    check(lib.xmh_label_sim(ptr(qlab), ptr(rlab), qlab.shape[0], rlab.shape[0], Cn, ptr(out), current_stream()), "xmh_label_sim")
    return out


def scan_plan(Q: int, R: int, K: int, ternary: bool) -> _lib.ScanPlan:
    p = _lib.ScanPlan()
    check(lib.xmh_scan_plan_make(Q, R, K, int(ternary), C.byref(p)), "xmh_scan_plan_make")
    return p


def _plan_and_workspace(Q: int, R: int, K: int, ternary: bool, device):
    """Plan the scan and allocate its workspace.  The plan includes the pair cache (one or two bytes per (query, item) pair, up to
    XMH_SCAN_CACHE_MB, default 128 GB) whenever the shape has one -- sized for an empty 288 GB device, not for what is free next to
    a resident encoder.  When that does not fit, the workspace is allocated WITHOUT the cache (xmh_scan_ws_bytes_nocache): the library
    takes the size it is handed as the decision, per call, and runs the same plan uncached -- slower, never an out-of-memory error for
    a shape that ran before the cache existed, and no process-wide state (round 4 lowered XMH_SCAN_CACHE_MB in the environment here)."""
    plan = scan_plan(Q, R, K, ternary)
    free, _ = torch.cuda.mem_get_info(device)
    free += torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)      # the caching allocator's idle blocks count
    if plan.ws_bytes <= 0.9 * free:
        try:
            return plan, torch.empty(plan.ws_bytes, dtype=torch.uint8, device=device)
        except torch.cuda.OutOfMemoryError:
            pass
    small = int(lib.xmh_scan_ws_bytes_nocache(Q, R, K, 1 if ternary else 0))
    if small <= 0 or small >= plan.ws_bytes:
        return plan, torch.empty(plan.ws_bytes, dtype=torch.uint8, device=device)       # no cache in this plan: the allocator raises for real
    warnings.warn("xmh: scan workspace of %.1f GB does not fit in %.1f GB of free device memory; this evaluation runs without its pair cache "
                  "(workspace %.1f GB)" % (plan.ws_bytes / 2**30, free / 2**30, small / 2**30))
    return plan, torch.empty(small, dtype=torch.uint8, device=device)


class RankingScan:
    """The two-pass fused scan for one (query set, gallery shard): owns the workspace and exposes the
    pieces the sharded driver needs (histogram totals, AP partial sums)."""

    def __init__(self, q: PackedCodes, qlab: torch.Tensor, r: PackedCodes, rlab: torch.Tensor, Cn: int, workspace=None):
        """``workspace``: a (plan, buffer) pair of an earlier RankingScan of the SAME shape, ternary-ness and device (``.workspace``) to
        run in instead of allocating one -- a caller that evaluates the same shape again and again (valid(): four calls per epoch) skips
        the allocation; use it from one stream at a time."""
        _require_cuda(q.bits, r.bits, qlab, rlab)
        if q.K != r.K:
            raise ValueError("code lengths differ: %d vs %d" % (q.K, r.K))
        q, r = widened(q), widened(r)
        if 4 < qlab.shape[1] <= 8 and qlab.shape[1] == rlab.shape[1]:
            # 129 ... 256 classes (IAPR TC-12: 255): the library's instances take exactly eight label words -- absent classes are zero bits
            pad = 8 - qlab.shape[1]
            if pad:
                qlab, rlab = torch.nn.functional.pad(qlab, (0, pad)), torch.nn.functional.pad(rlab, (0, pad))
            Cn = 256
        self.q, self.r, self.qlab, self.rlab, self.C = q, r, qlab.contiguous(), rlab.contiguous(), Cn
        self.qz, self.rz = _both_planes(q, r)
        if workspace is not None:
            self.plan, self.ws = workspace
            if self.ws.device != q.bits.device or self.ws.numel() < int(lib.xmh_scan_ws_bytes_nocache(q.n, r.n, q.K, 1 if self.qz is not None else 0)):
                raise ValueError("RankingScan: the workspace handed in does not fit this shape")
        else:
            self.plan, self.ws = _plan_and_workspace(q.n, r.n, q.K, self.qz is not None, q.bits.device)

    @property
    def workspace(self):
        return self.plan, self.ws

    def _common(self):
        return (ptr(self.q.bits), ptr(self.qz), ptr(self.qlab), ptr(self.r.bits), ptr(self.rz), ptr(self.rlab),
                self.q.n, self.r.n, self.q.K, self.C, ptr(self.ws), self.ws.numel())      # the size decides: a buffer without room for the pair cache runs uncached

    def histograms(self, want_totals: bool = True):
        """pass 1.  Returns (hist_all, hist_rel) int32 [Q, nbuckets] shard totals (or (None, None))."""
        ha = hr = None
        if want_totals:
            pair = torch.empty(2, self.q.n, self.plan.nbuckets, dtype=torch.int32, device=self.ws.device)
            ha, hr = pair[0], pair[1]                  # one buffer: the sharded driver all-gathers both planes at once
        check(lib.xmh_hamming_hist(*self._common(), ptr(ha), ptr(hr), current_stream()), "xmh_hamming_hist")
        return ha, hr

    def ap_sums(self, k: Optional[int] = None, base_all=None, base_rel=None, nrel_total=None):
        """pass 2.  Returns (ap_sum float64 [Q], cap int32 [Q])."""
        dev = self.ws.device
        ap = torch.empty(self.q.n, dtype=torch.float64, device=dev)
        cap = torch.empty(self.q.n, dtype=torch.int32, device=dev)
        kk = 0 if k is None else int(k)
        if k is not None and kk <= 0:
            raise ValueError("k must be positive or None")
        check(lib.xmh_hamming_ap(*self._common(), ptr(base_all), ptr(base_rel), ptr(nrel_total), kk, ptr(ap), ptr(cap),
                                 current_stream()), "xmh_hamming_ap")
        return ap, cap


    def map_all(self, k: Optional[int] = None):
        """pass 2 of an unsharded gallery with the finalisation folded in -> (map float64 [1], ap_sum [Q], cap [Q])."""
        dev = self.ws.device
        ap = torch.empty(self.q.n, dtype=torch.float64, device=dev)
        cap = torch.empty(self.q.n, dtype=torch.int32, device=dev)
        out = torch.empty(1, dtype=torch.float64, device=dev)
        kk = 0 if k is None else int(k)
        if k is not None and kk <= 0:
            raise ValueError("k must be positive or None")
        check(lib.xmh_hamming_map(*self._common(), kk, ptr(ap), ptr(cap), ptr(out), current_stream()), "xmh_hamming_map")
        return out, ap, cap


def _totals(self) -> torch.Tensor:
    """this shard's totals table as pass 1 left it in the workspace: int32 view [nbuckets, qpad, 2] {all, relevant} -- what a
    sharded evaluation all-gathers (no export pass, no copy).  Valid until the next histograms() on this object."""
    nbytes = C.c_size_t(0)
    off = lib.xmh_scan_totals_offset(self.q.n, self.r.n, self.q.K, 1 if self.qz is not None else 0, C.byref(nbytes))
    if off == C.c_size_t(-1).value:
        check(1, "xmh_scan_totals_offset")
    return self.ws[off:off + nbytes.value].view(torch.int32).view(self.plan.nbuckets, self.plan.qpad, 2)


def _map_sharded(self, k, totals_gathered: torch.Tensor, rank: int):
    """pass 2 of ONE SHARD from the all-gathered totals tables of the shards ([world, nbuckets, qpad, 2] int32) with this shard's
    share of the mean folded in -> (map_partial float64 [1], ap_sum [Q] of this shard, cap [Q] global).  The mAP is the sum of
    map_partial over the shards."""
    if (totals_gathered.dim() != 4 or tuple(totals_gathered.shape[1:]) != (self.plan.nbuckets, self.plan.qpad, 2)
            or totals_gathered.dtype != torch.int32 or not totals_gathered.is_contiguous()):
        raise ValueError("map_sharded: expected a contiguous int32 [world, %d, %d, 2] tensor" % (self.plan.nbuckets, self.plan.qpad))
    world = totals_gathered.shape[0]
    dev = self.ws.device
    ap = torch.empty(self.q.n, dtype=torch.float64, device=dev)
    cap = torch.empty(self.q.n, dtype=torch.int32, device=dev)
    out = torch.empty(1, dtype=torch.float64, device=dev)
    kk = 0 if k is None else int(k)
    if k is not None and kk <= 0:
        raise ValueError("k must be positive or None")
    check(lib.xmh_hamming_map_sharded(*self._common(), ptr(totals_gathered), world, int(rank), kk, ptr(ap), ptr(cap), ptr(out),
                                      current_stream()), "xmh_hamming_map_sharded")
    return out, ap, cap


def slice_offsets(totals_slices: torch.Tensor) -> torch.Tensor:
    """all-to-all form of the sharded exchange, the slice owner's part: the columns of this rank's query slice from every shard,
    int32 [world, nbuckets, slice, 2], -> the offsets of those queries for EVERY shard, int32 [world, nbuckets + 1, slice, 2]
    (last row: {relevant items, items} over all shards)."""
    _require_cuda(totals_slices)
    if totals_slices.dim() != 4 or totals_slices.shape[3] != 2 or totals_slices.dtype != torch.int32 or not totals_slices.is_contiguous():
        raise ValueError("slice_offsets: expected a contiguous int32 [world, nbuckets, slice, 2] tensor")
    world, nb, S, _ = totals_slices.shape
    out = torch.empty(world, nb + 1, S, 2, dtype=torch.int32, device=totals_slices.device)
    check(lib.xmh_shard_slice_offsets(ptr(totals_slices), world, nb, S, ptr(out), current_stream()), "xmh_shard_slice_offsets")
    return out


def _map_sharded_offsets(self, k, offsets: torch.Tensor):
    """pass 2 of ONE SHARD from its offset rows as the second all-to-all delivered them ([world (slice owner), nbuckets + 1,
    slice, 2] int32) -> (map_partial float64 [1], ap_sum [Q] of this shard, cap [Q] global), like map_sharded."""
    world = offsets.shape[0]
    if (offsets.dim() != 4 or offsets.shape[1] != self.plan.nbuckets + 1 or offsets.shape[2] * world != self.plan.qpad or offsets.shape[3] != 2
            or offsets.dtype != torch.int32 or not offsets.is_contiguous()):
        raise ValueError("map_sharded_offsets: expected a contiguous int32 [world, %d, %d / world, 2] tensor" % (self.plan.nbuckets + 1, self.plan.qpad))
    dev = self.ws.device
    ap = torch.empty(self.q.n, dtype=torch.float64, device=dev)
    cap = torch.empty(self.q.n, dtype=torch.int32, device=dev)
    out = torch.empty(1, dtype=torch.float64, device=dev)
    kk = 0 if k is None else int(k)
    if k is not None and kk <= 0:
        raise ValueError("k must be positive or None")
    check(lib.xmh_hamming_map_sharded_offsets(*self._common(), ptr(offsets), world, kk, ptr(ap), ptr(cap), ptr(out), current_stream()),
          "xmh_hamming_map_sharded_offsets")
    return out, ap, cap


RankingScan.totals = _totals
RankingScan.map_sharded = _map_sharded
RankingScan.map_sharded_offsets = _map_sharded_offsets


def map_finalize(ap_sum: torch.Tensor, cap: torch.Tensor) -> torch.Tensor:
    out = torch.empty(1, dtype=torch.float64, device=ap_sum.device)
    check(lib.xmh_map_finalize(ptr(ap_sum), ptr(cap), ap_sum.shape[0], ptr(out), current_stream()), "xmh_map_finalize")
    return out


def shard_offsets(hist_gathered: torch.Tensor, rank: int):
    """[world, 2, Q, nb] int32 all-gathered shard histograms -> (base_all, base_rel [Q, nb], nrel_total [Q]) for ``rank``."""
    world, two, Q, nb = hist_gathered.shape
    if two != 2 or hist_gathered.dtype != torch.int32 or not hist_gathered.is_contiguous():
        raise ValueError("shard_offsets: expected a contiguous int32 [world, 2, Q, nb] tensor")
    dev = hist_gathered.device
    base_a = torch.empty(Q, nb, dtype=torch.int32, device=dev)
    base_r = torch.empty_like(base_a)
    nrel = torch.empty(Q, dtype=torch.int32, device=dev)
    check(lib.xmh_shard_offsets(ptr(hist_gathered), world, int(rank), Q, nb, ptr(base_a), ptr(base_r), ptr(nrel), current_stream()),
          "xmh_shard_offsets")
    return base_a, base_r, nrel


def map_k_packed(q: PackedCodes, r: PackedCodes, qlab: torch.Tensor, rlab: torch.Tensor, Cn: int,
                 k: Optional[int] = None, workspace=None, return_scan: bool = False):
    """mAP of one query set against one (unsharded) gallery; float64 [1] on the device (``workspace``: see RankingScan)."""
    scan = RankingScan(q, qlab, r, rlab, Cn, workspace=workspace)
    scan.histograms(want_totals=False)
    m = scan.map_all(k)[0]
    return (m, scan) if return_scan else m


class TopkWorkspace:
    """A prepared top-k workspace for one (Q, R, K, k) shape on one device: initialised once (xmh_topk_ws_init), then every call
    through it leaves its control words clean for the next one -- a query loop over a fixed gallery pays no memset launch.
    Use it from one stream at a time.  ``ternary``: for code sets with zero planes (2K + 1 half-unit buckets)."""

    def __init__(self, Q: int, R: int, K: int, k: int, device, ternary: bool = False):
        self.shape = (int(Q), int(R), int(K), int(k))
        self.ternary = bool(ternary)
        ws_bytes, ws_init = ((lib.xmh_topk_ternary_ws_bytes, lib.xmh_topk_ternary_ws_init) if self.ternary
                             else (lib.xmh_topk_ws_bytes, lib.xmh_topk_ws_init))
        self.bytes = ws_bytes(*self.shape)
        if self.bytes == 0:
            _topk_shape_error(self.shape, self.ternary)
        self.buf = torch.empty(self.bytes, dtype=torch.uint8, device=device)
        check(ws_init(*self.shape, ptr(self.buf), self.bytes, current_stream()), "xmh_topk_ws_init")


def _topk_shape_error(shape, ternary: bool):
    """a shape the planner refuses: make the library say why"""
    Q, R, K, k = shape
    if ternary:                                          # (the planner speaks first: the null workspace is never reached)
        check(lib.xmh_topk_ternary_ws_init(Q, R, K, k, None, 0, None), "xmh_hamming_topk_ternary")
    check(lib.xmh_hamming_topk(None, None, Q, R, K, k, 0, None, 0, None, None, None), "xmh_hamming_topk")


def _topk_out(out, Q: int, k: int, dev):
    if out is None:
        return torch.empty(Q, k, dtype=torch.int16, device=dev), torch.empty(Q, k, dtype=torch.int32, device=dev)
    dist, idx = out
    if (tuple(dist.shape) != (Q, k) or tuple(idx.shape) != (Q, k) or dist.dtype != torch.int16 or idx.dtype != torch.int32
            or dist.device != dev or idx.device != dev or not dist.is_contiguous() or not idx.is_contiguous()):
        raise ValueError("hamming_topk: out must be contiguous (int16 [%d, %d], int32 [%d, %d]) tensors on %s" % (Q, k, Q, k, dev))
    return dist, idx


def topk_ws_key(q: PackedCodes, r: PackedCodes, k: int, ternary: Optional[bool] = None):
    """(Q, R, K as the kernels see it, k, ternary) -- the shape a TopkWorkspace must have been prepared for"""
    tern = q.zero is not None or r.zero is not None or bool(ternary)
    return (q.n, r.n, widened(q).K, int(k), tern)


def hamming_topk(q: PackedCodes, r: PackedCodes, k: int, base_index: int = 0, workspace: Optional[TopkWorkspace] = None, out=None,
                 ternary: Optional[bool] = None):
    """Exact top-k of every query over this gallery shard under (distance, index) order.
    Returns (dist int16-storage [Q,k] (uint16 bit pattern, 0xFFFF = unused slot), idx int32 [Q,k] global
    indices = base_index + row, -1 = unused slot when the shard has fewer than k rows).
    Ternary code sets (a zero plane on either side: sign_() left an exact 0, reference runners/base.py:407-410) are ranked by the
    reference's 0.5 * (K - q.r); ``dist`` then holds HALF units (K - q.r, 0 ... 2K) -- ``q.ternary or r.ternary`` tells which.
    ``ternary=True`` asks for half units although neither side has a zero plane (a rank of a sharded call whose shard happens to hold
    no zero while another rank's does: the lists must merge in one unit).
    ``workspace``: a TopkWorkspace of this shape (see there); without one a scratch workspace is allocated and cleared per call.
    ``out``: (dist, idx) contiguous device tensors of those shapes and dtypes to write into (the sharded driver passes views of the
    record it all-gathers)."""
    _require_cuda(q.bits, r.bits)
    if q.K != r.K:
        raise ValueError("code lengths differ: %d vs %d" % (q.K, r.K))
    K_true = q.K
    qz, rz = _both_planes(q, r)
    if qz is None and ternary:
        qz, rz = zero_plane_or_default(q), zero_plane_or_default(r)
    if qz is not None and ternary is False:
        raise ValueError("hamming_topk: ternary=False for code sets with a zero plane")
    tern = qz is not None
    if tern:
        q, r = PackedCodes(q.bits, qz, q.K, q.flags), PackedCodes(r.bits, rz, r.K, r.flags)
    q, r = widened(q), widened(r)
    Q, R = q.n, r.n
    dev = q.bits.device
    shape = (Q, R, q.K, int(k))
    if workspace is not None:
        if workspace.shape != shape or workspace.buf.device != dev or workspace.ternary != tern:
            raise ValueError("top-k workspace was prepared for %r%s on %s, call is %r%s on %s"
                             % (workspace.shape, " ternary" if workspace.ternary else "", workspace.buf.device, shape, " ternary" if tern else "", dev))
        ws, need, prepared = workspace.buf, workspace.bytes, 1
    else:
        need = (lib.xmh_topk_ternary_ws_bytes if tern else lib.xmh_topk_ws_bytes)(*shape)
        if need == 0:
            _topk_shape_error(shape, tern)
        ws, prepared = torch.empty(need, dtype=torch.uint8, device=dev), 0
    dist, idx = _topk_out(out, Q, k, dev)
    if tern:
        check(lib.xmh_hamming_topk_ternary(ptr(q.bits), ptr(q.zero), ptr(r.bits), ptr(r.zero), Q, R, q.K, k, base_index, ptr(ws), need, prepared,
                                           ptr(dist), ptr(idx), current_stream()), "xmh_hamming_topk_ternary")
        if q.K != K_true:
            # codes widened to the next kernel word count: the extra words are "zero on both sides" and add the same constant to every
            # half-unit distance (the ranking is untouched); taken out again, unused slots (0xFFFF) stay
            dist.sub_(torch.where(dist == -1, 0, q.K - K_true).to(torch.int16))
    elif prepared:
        check(lib.xmh_hamming_topk_prepared(ptr(q.bits), ptr(r.bits), Q, R, q.K, k, base_index, ptr(ws), need,
                                            ptr(dist), ptr(idx), current_stream()), "xmh_hamming_topk_prepared")
    else:
        check(lib.xmh_hamming_topk(ptr(q.bits), ptr(r.bits), Q, R, q.K, k, base_index, ptr(ws), need, ptr(dist), ptr(idx),
                                   current_stream()), "xmh_hamming_topk")
    return dist, idx
