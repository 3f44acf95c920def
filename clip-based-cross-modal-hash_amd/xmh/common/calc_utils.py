"""Drop-in for the reference's ``common/calc_utils.py`` -- same names, positional arguments, return
types and error behaviour; the arithmetic runs in libxmh.so on the GPU.

    calc_hammingDist      common/calc_utils.py:51-56   -> xmh_hamming_dist (bit-packed XOR/popcount)
    calc_map_k            common/calc_utils.py:58-92   -> xmh_hamming_hist + xmh_hamming_ap + xmh_map_finalize
    calc_label_sim        common/calc_utils.py:8-10    -> xmh_label_sim
    cosine_similarity     common/calc_utils.py:38-49   -> xmh_rownorm + xmh_gemm_nt_f32
    euclidean_similarity  common/calc_utils.py:28-36   -> xmh_pairwise_l2

Differences that are deliberate and documented (DESIGN.md "Parity"):
  * ranking ties are broken by gallery index (torch.sort(stable=True)); the reference's unstable sort
    leaves them unspecified.
  * inputs may live on the CPU like in the reference (calc_map_k moves them there, :62-64); here they
    are moved TO the GPU instead; calc_map_k still returns a CPU 0-dim float32 tensor and the other
    functions return on the device their inputs came from (host in -> host out).
  * there is no CPU fallback: without a GPU these functions raise.
"""
from __future__ import annotations

from typing import Union

import numpy as np
import threading

import torch

from .. import retrieval as R


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("xmh.common.calc_utils needs an MI355X (torch.cuda.is_available() is False); no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _to_gpu(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_cuda else t.to(_device(), non_blocking=True)


class _on_device_of:
    """run the body with the device of the first CUDA operand current (libxmh launches on the current device's stream), and
    remember whether every operand came from the host: the reference returns its result on the inputs' device
    (common/calc_utils.py:8-56), so ``back()`` moves a result there."""

    def __init__(self, *tensors):
        cuda = [t for t in tensors if isinstance(t, torch.Tensor) and t.is_cuda]
        self.all_host = not cuda
        self.ctx = torch.cuda.device(cuda[0].device if cuda else _device())

    def __enter__(self):
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        return self.ctx.__exit__(*exc)

    def back(self, t: torch.Tensor) -> torch.Tensor:
        return t.cpu() if self.all_host else t


def _pack_codes(B: torch.Tensor) -> R.PackedCodes:
    p = R.pack_sign(_to_gpu(B))
    return p


_label_cache = {}            # (data_ptr, version, shape, dtype, device) -> (packed masks, the label tensor itself)
_scan_ws = threading.local()  # .entry = (shape key, (plan, workspace buffer)) of this thread's last calc_map_k
_KEEP_WS_BYTES = 2 << 30     # workspaces up to 2 GiB are kept between calls (0.6 GB at the COCO shape)


def release_scan_workspace() -> None:
    """drop the scan workspace calc_map_k keeps between calls of one shape (0.6 GB at the COCO shape: the pair cache)"""
    _scan_ws.__dict__.pop("entry", None)


def set_workspace_keep_bytes(nbytes: int) -> int:
    """how large a scan workspace calc_map_k may keep alive between calls (per thread; default 2 GiB; 0 = keep nothing: every call
    allocates and frees its own, +0.1 ms at the COCO shape).  Returns the previous limit.  The reference keeps nothing on the device
    between calls; this is the one piece of device memory the drop-in holds on to, so it has a knob."""
    global _KEEP_WS_BYTES
    old, _KEEP_WS_BYTES = _KEEP_WS_BYTES, max(0, int(nbytes))
    kept = _scan_ws.__dict__.get("entry")
    if kept is not None:
        buf = kept[1][1] if isinstance(kept[1], tuple) else kept[1]          # composed path keeps (plan, buffer), the one-call path the buffer
        if buf.numel() > _KEEP_WS_BYTES:
            release_scan_workspace()
    return old


def _packed_labels(L: torch.Tensor) -> torch.Tensor:
    """bit-packed label masks on the GPU.  valid() calls calc_map_k four times with the same two label matrices
    (runners/base.py:312-315), usually int64 on the CPU: they are moved and packed once.  The entry keeps a reference to the
    label tensor, so its address cannot be reused while it is cached; in-place edits bump ``_version``."""
    key = (L.data_ptr(), L._version, tuple(L.shape), L.dtype, str(L.device))
    hit = _label_cache.get(key)
    if hit is None:
        if len(_label_cache) >= 8:
            _label_cache.clear()
        hit = (R.pack_labels(_to_gpu(L)), L)
        _label_cache[key] = hit
    return hit[0]


def _calc_map_k_one_call(gq: torch.Tensor, gr: torch.Tensor, ql: torch.Tensor, rl: torch.Tensor, C: int, k):
    """xmh_calc_map_k: pack + both passes + the scalar back in one call of the C ABI (round 5: the composed path spent 140 us of host work
    around a 360 us scan).  None when the shape needs the composed path or the codes are not quantised."""
    import ctypes
    from .._lib import current_stream, lib, ptr
    if gq.dtype != torch.float32 or gr.dtype != torch.float32 or gq.dim() != 2 or gr.dim() != 2 or gq.shape[1] != gr.shape[1]:
        return None
    gq, gr = gq.contiguous(), gr.contiguous()
    Q, Rn, K = gq.shape[0], gr.shape[0], gq.shape[1]
    Lw = (C + 31) // 32
    if (K + 31) // 32 not in (1, 2, 4, 8, 16, 32, 64) or not (Lw <= 4 or Lw == 8):
        return None
    # the C ABI takes raw pointers: every extent it will read is checked here (mismatches make the reference's mm raise, :72)
    if tuple(ql.shape) != (Q, Lw) or tuple(rl.shape) != (Rn, Lw) or not ql.is_contiguous() or not rl.is_contiguous():
        raise ValueError("calc_map_k: labels %s / %s do not match %d queries, %d gallery rows, %d classes"
                         % (tuple(ql.shape), tuple(rl.shape), Q, Rn, C))
    need = int(lib.xmh_calc_map_k_ws_bytes(Q, Rn, K, C))
    if need == 0:
        return None
    key = ("fused", Q, Rn, K, C, str(gq.device), torch.cuda.current_stream(gq.device).cuda_stream)
    hit = _scan_ws.__dict__.get("entry")
    if hit is not None and hit[0] == key and hit[1].numel() >= need:    # (the plan size follows per-call switches: a kept buffer must still fit)
        ws = hit[1]
    else:
        release_scan_workspace()
        free, _ = torch.cuda.mem_get_info(gq.device)
        free += torch.cuda.memory_reserved(gq.device) - torch.cuda.memory_allocated(gq.device)
        if need > 0.9 * free:
            return None                                  # the composed path knows how to run without the pair cache
        ws = torch.empty(need, dtype=torch.uint8, device=gq.device)
    # the two host words the call writes (value flags, mAP) live in PINNED memory, one 16-byte buffer per thread: a device-to-host copy into
    # pageable memory goes through the runtime's staging buffer, 10-15 us each at this size (bench_dropin: 493 -> see INTEGRATION section 2)
    host = _scan_ws.__dict__.get("host")
    if host is None:
        host = _scan_ws.host = torch.empty(2, dtype=torch.float64).pin_memory()
    host[0] = float("nan")
    host.view(torch.int32)[2] = 0
    base = host.data_ptr()
    rc = lib.xmh_calc_map_k(ptr(gq), ptr(gr), ptr(ql), ptr(rl), Q, Rn, K, C, 0 if k is None else int(k), ptr(ws), ws.numel(),
                            ctypes.cast(base, ctypes.POINTER(ctypes.c_double)), ctypes.cast(base + 8, ctypes.POINTER(ctypes.c_int32)), current_stream())
    if ws.numel() <= _KEEP_WS_BYTES:
        _scan_ws.entry = (key, ws)
    if rc != 0 or (int(host.view(torch.int32)[2]) & 2):
        return None                                      # not supported in this form / unquantised codes: the composed path reports or handles it
    return torch.tensor(float(host[0]), dtype=torch.float32)


def _is_quantised(*packed: R.PackedCodes) -> bool:
    return not any(p.flags & 2 for p in packed)


def calc_label_sim(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """(a @ b^T > 0).float() for multi-hot label matrices; computed on the GPU, returned on the inputs' device."""
    with _on_device_of(a, b) as dv:
        a, b = _to_gpu(a), _to_gpu(b)
        return dv.back(R.label_sim(R.pack_labels(a), R.pack_labels(b), a.shape[1]))


def calc_hammingDist(B1: torch.Tensor, B2: torch.Tensor) -> torch.Tensor:
    """0.5 * (K - B1 @ B2^T), float32 [Q,R]; a 1-D B1 is one query (reference :53-54)."""
    if B1.dim() < 2:
        B1 = B1.unsqueeze(0)
    with _on_device_of(B1, B2) as dv:
        q, r = _pack_codes(B1), _pack_codes(B2)
        if _is_quantised(q, r):
            return dv.back(R.hamming_dist(q, r))
        from .. import dense                           # un-quantised float "codes" (UMoED-style)
        return dv.back(dense.hamming_dist_float(_to_gpu(B1).float(), _to_gpu(B2).float()))


def calc_map_k(qB, rB, query_L, retrieval_L, k=None) -> torch.Tensor:
    """mAP@k with the reference's semantics: ``k`` caps the number of relevant items averaged (:81),
    a query without relevant items makes the result NaN (:87-89), a single query raises IndexError
    (the squeeze at :72).  Returns a CPU 0-dim float32 tensor like the reference."""
    num_query = query_L.shape[0]
    if num_query == 1:
        raise IndexError("calc_map_k needs more than one query (reference squeezes the query axis, calc_utils.py:72)")
    if k is not None:
        k = int(k)
        if k == 0:                                       # reference: totals = 0 -> mean of an empty tensor per query (:81-89) -> nan
            return torch.tensor(float("nan"), dtype=torch.float32)
        if k < 0:                                        # reference: count / tindex of different lengths -> RuntimeError (:85-89)
            raise ValueError("calc_map_k: k must be positive or None (got %d)" % k)
    if query_L.shape[0] != qB.shape[0] or retrieval_L.shape[0] != rB.shape[0] or query_L.shape[1] != retrieval_L.shape[1]:
        raise ValueError("calc_map_k: %d / %d code rows against %d / %d label rows, %d / %d classes"
                         % (qB.shape[0], rB.shape[0], query_L.shape[0], retrieval_L.shape[0], query_L.shape[1], retrieval_L.shape[1]))
    with _on_device_of(qB, rB, query_L, retrieval_L):
        gq, gr = _to_gpu(qB), _to_gpu(rB)
        ql, rl, C = _packed_labels(query_L), _packed_labels(retrieval_L), query_L.shape[1]
        fused = _calc_map_k_one_call(gq, gr, ql, rl, C, k)
        if fused is not None:
            return fused
        # (code lengths that need widening, 129 ... 224 classes, unquantised values: the composed path below)
        K = gq.shape[1]
        # what the bit-packed kernels have no instance for goes down the reference's own route -- float GEMM + one sort per query
        # (xmh_gemm_f32_sort_map): the drop-in returns a number wherever the reference does
        why = None
        if K > 2048:
            why = "codes of %d bits (the bit-packed scan stops at 2048)" % K
        elif ql.shape[1] > 8:
            why = "%d classes (the bit-packed scan stops at 256)" % C
        if why is None:
            # both code matrices are packed before the ONE read of their value flags (round 5: two stand-alone packs were two syncs)
            fl = torch.zeros(1, dtype=torch.int32, device=gq.device)
            q, r = R.pack_sign(gq, flags=fl, defer=True), R.pack_sign(gr, flags=fl, defer=True)
            R.settle_flags(fl, q, r)
            if not _is_quantised(q, r):
                why = "codes contain values outside {-1,0,+1}"
            elif q.zero is not None and K > 256:
                why = "ternary codes (an exact 0 among the values) of %d bits (the bit-packed scan takes zero planes up to 256)" % K
        if why is None:
            # valid() evaluates the same shape four times per epoch (runners/base.py:312-315): the scan workspace of the last shape is
            # kept (release_scan_workspace() drops it; it is per thread, and used on the caller's current stream)
            key = (q.n, r.n, q.K, q.zero is not None or r.zero is not None, str(gq.device), torch.cuda.current_stream(gq.device).cuda_stream)
            hit = _scan_ws.__dict__.get("entry")
            res, scan = R.map_k_packed(q, r, ql, rl, C, k, workspace=hit[1] if hit is not None and hit[0] == key else None, return_scan=True)
            if scan.ws.numel() <= _KEEP_WS_BYTES:
                _scan_ws.entry = (key, scan.workspace)
            else:                                        # a 10 M-row gallery's pair cache is tens of GB: not something to sit on between calls
                release_scan_workspace()
        else:
            from .. import dense
            release_scan_workspace()                     # the float path's tile wants the room
            res = dense.map_k_float(gq.float(), gr.float(), ql, rl, C, k, why=why)
        return res.to(torch.float32).cpu().reshape(())


def cosine_similarity(a: Union[torch.Tensor, np.ndarray], b: Union[torch.Tensor, np.ndarray]):
    if isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor):
        from .. import dense
        with _on_device_of(a, b) as dv:
            return dv.back(dense.cosine(_to_gpu(a).float(), _to_gpu(b).float()))
    if isinstance(a, np.ndarray) and isinstance(b, np.ndarray):
        from .. import dense
        return dense.cosine(_to_gpu(torch.from_numpy(a)).float(), _to_gpu(torch.from_numpy(b)).float()).cpu().numpy()
    raise ValueError("input value must in [torch.Tensor, numpy.ndarray], but it is %s, %s" % (type(a), type(b)))


def euclidean_similarity(a: Union[torch.Tensor, np.ndarray], b: Union[torch.Tensor, np.ndarray]):
    if isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor):
        from .. import dense
        with _on_device_of(a, b) as dv:
            return dv.back(dense.pairwise_l2(_to_gpu(a).float(), _to_gpu(b).float()))
    if isinstance(a, np.ndarray) and isinstance(b, np.ndarray):
        from .. import dense
        return dense.pairwise_l2(_to_gpu(torch.from_numpy(a)).float(), _to_gpu(torch.from_numpy(b)).float()).cpu().numpy()
    raise ValueError("input value must in [torch.Tensor, numpy.ndarray], but it is %s, %s" % (type(a), type(b)))
