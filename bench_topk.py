#!/usr/bin/env python3
"""HBM-bound regime of the retrieval path: exact top-k of a handful of queries over a 10 M x 256-bit gallery
(BASELINE configs[4] shape, the whole gallery on ONE GPU: 320 MB of codes > 256 MB Infinity Cache).

Called by bench.py (``measure()``) and runnable on its own:  python bench_topk.py [--R 10000000 --K 256 --Q 8 --k 100]
Prints / returns the `roofline` object of k_topk_stream: algorithmic bytes = R*K/8 (gallery read once)
+ Q*K/8 + partial lists written, divided by the HIP-event time of the launch bracket.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0


def filter_instance(K, Q, R=10_000_000, k=100):
    """the streaming kernel instance this call launches, as the library reports it (xmh_topk_describe: one decision function for the
    launch and for this name)"""
    import ctypes
    from xmh._lib import check, lib
    buf = ctypes.create_string_buffer(256)
    check(lib.xmh_topk_describe(Q, R, K, k, buf, 256), "xmh_topk_describe")
    return buf.value.decode().split("=", 1)[1]


def filter_on_matrix_cores(K, Q):
    return filter_instance(K, Q).startswith("k_topk_filter_mfma")


def _traffic(K, Q):
    """PMC HBM bytes of exactly the filter instance this (K, Q) launches"""
    try:
        import bench_roofline
        t = bench_roofline.pmc_traffic(filter_instance(K, Q))
        return None if t is None else t["bytes"]
    except Exception:
        return None


def _codes(kind, R, K, Q, seed=1814):
    """packed gallery / query codes on the GPU.  "iid": uniform random bits (the sampled threshold's ideal case);
    "structured": SURVEY 8d's sign(L W + 0.8 randn) with 80 COCO-like classes -- items sharing labels cluster, so the distance
    distribution of a query is multi-modal and heavy near 0; "duplicates": the structured gallery with every item repeated 64
    times in a row (ties far beyond k at every distance: the exact (distance, index) tie-break does the work)."""
    from xmh import retrieval as X
    W = (K + 31) // 32
    g = torch.Generator(device="cuda").manual_seed(seed)
    if kind == "iid":
        rb = torch.randint(-2**31, 2**31 - 1, (R, W), dtype=torch.int32, device="cuda", generator=g)
        qb = torch.randint(-2**31, 2**31 - 1, (Q, W), dtype=torch.int32, device="cuda", generator=g)
        if K % 32:                                           # the contract of packed codes: bits beyond K are zero
            rb[:, -1] &= (1 << (K % 32)) - 1
            qb[:, -1] &= (1 << (K % 32)) - 1
        return X.PackedCodes(qb, None, K), X.PackedCodes(rb, None, K)
    C, p = 80, 0.04
    Wm = torch.randn(C, K, device="cuda", generator=g)

    def side(n):
        out = X.empty_packed(n, K, "cuda")
        for lo in range(0, n, 1 << 20):                      # 1 M rows at a time: the fp32 codes never exist in full
            m = min(1 << 20, n - lo)
            L = (torch.rand(m, C, device="cuda", generator=g) < p).float()
            L[torch.arange(m, device="cuda"), torch.randint(0, C, (m,), device="cuda", generator=g)] = 1.0
            B = L @ Wm + 0.8 * torch.randn(m, K, device="cuda", generator=g)
            out.bits[lo:lo + m] = X.pack_sign(torch.where(B == 0, torch.ones_like(B), B).sign()).bits
        return out
    if kind == "structured":
        return side(Q), side(R)
    base = side((R + 63) // 64)
    rep = X.PackedCodes(base.bits.repeat_interleave(64, dim=0)[:R].contiguous(), None, K)
    return side(Q), rep


def measure(R=10_000_000, K=256, Q=8, k=100, iters=20, warmup=3, kind="iid", robust=True):
    from xmh import retrieval as X
    from xmh._lib import lib
    W = (K + 31) // 32
    q, r = _codes(kind, R, K, Q)
    from xmh import _lib
    # the query loop of a serving process: one prepared workspace (xmh_topk_ws_init once), every call leaves it clean
    ws = X.TopkWorkspace(q.n, r.n, K, k, "cuda")
    for _ in range(warmup):
        d, i = X.hamming_topk(q, r, k, workspace=ws)
    torch.cuda.synchronize()
    def loop(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for n in range(iters):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e-3, out

    t_unprepared, (du, iu) = loop(lambda: X.hamming_topk(q, r, k))                   # scratch workspace, cleared per call
    t_call, (d, i) = loop(lambda: X.hamming_topk(q, r, k, workspace=ws))
    # the streaming launch alone: HIP events around it inside the library (they cost the call ~8 us, so the whole-call times
    # above are taken without them)
    _lib.prof_enable(True)
    for n in range(iters):
        X.hamming_topk(q, r, k, workspace=ws)
    torch.cuda.synchronize()
    assert torch.equal(du, d) and torch.equal(iu, i)
    t, launches = _lib.prof_read("topk_filter")
    t *= 1e-3
    _lib.prof_enable(False)
    alg = R * W * 4 + Q * W * 4 + Q * 4                 # gallery read once + queries + thresholds
    # the robust path alone (what a call costs when a candidate list overflows or the sample misjudges): XMH_TOPK_ROBUST_ONLY
    # makes the library skip the fast path; same call, same outputs
    if not robust:
        return _result(alg, t, t_call, launches, R, K, Q, k, kind)
    os.environ["XMH_TOPK_ROBUST_ONLY"] = "1"
    try:
        for _ in range(2):
            d2, i2 = X.hamming_topk(q, r, k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            d2, i2 = X.hamming_topk(q, r, k)
        e1.record()
        torch.cuda.synchronize()
        t_robust = e0.elapsed_time(e1) / 5 * 1e-3
        same = bool(torch.equal(d2, d) and torch.equal(i2, i))
    finally:
        os.environ.pop("XMH_TOPK_ROBUST_ONLY", None)
    robust_only = {"whole_call_ms": t_robust * 1e3, "whole_call_GBps": alg / t_robust / 1e9, "equals_fast_path": same}
    out = _result(alg, t, t_call, launches, R, K, Q, k, kind)
    out["whole_call_unprepared_ms"] = t_unprepared * 1e3
    out["robust_path_alone"] = robust_only
    return out


def _result(alg, t, t_call, launches, R, K, Q, k, kind):
    inst = filter_instance(K, Q)
    name = "k_topk_filter_mfma (distances on v_mfma_i32_16x16x64_i8)" if inst.startswith("k_topk_filter_mfma") else inst
    return {"kernel": "%s, the streaming pass of xmh_hamming_topk; HIP events around the launch, %d launches" % (name, launches),
            "bound": "hbm", "achieved": alg / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / t / 1e9 / HBM_PEAK_GBS,
            "traffic": _traffic(K, Q), "algorithmic_bytes": alg, "avg_launch_ms": t * 1e3,
            "whole_call_ms": t_call * 1e3, "whole_call_GBps": alg / t_call / 1e9,
            "workload": "exact top-%d of Q=%d queries over R=%d x %d-bit gallery on one GPU, %s codes" % (k, Q, R, K, kind),
            "pairs_per_s_whole_call": Q * R / t_call, "robust_path_launches": _robust_launches(launches), **_mfma_bound(inst, R, K, Q, t)}


I8_MFMA_TOPS = 3944.0      # v_mfma_i32_16x16x64_i8, the guide's measured ceiling (MI355X_MICROARCH.md, "I8 ... >= 3944 TOPS")


def _mfma_bound(inst, R, K, Q, t):
    """From a handful of queries per gallery pass the filter is the matrix-core kernel and its bound is the i8 MFMA rate, not HBM: report
    the operations it ISSUES (query tiles of 16 columns: 5 .. 16 queries issue the same MFMAs) and the useful ones, against that ceiling."""
    if not inst.startswith("k_topk_filter_mfma"):
        return {}
    tiles = (Q + 15) // 16
    issued, useful = 2.0 * 16 * tiles * R * K, 2.0 * Q * R * K
    return {"mfma": {"bound": "mfma", "unit": "TOPS", "peak": I8_MFMA_TOPS, "issued": issued / t / 1e12, "useful": useful / t / 1e12,
                     "frac_issued": issued / t / 1e12 / I8_MFMA_TOPS, "frac": useful / t / 1e12 / I8_MFMA_TOPS}}


def _robust_launches(filter_launches):
    """how often the gated robust path did the work (its kernels return at once when the fast path's lists held)"""
    try:
        from xmh import _lib
        t, n = _lib.prof_read("topk_robust")
        return {"launches": n, "avg_ms": t}
    except Exception:
        return None


def measure_ternary(R=10_000_000, K=256, Q=1, k=100, iters=20, p_zero=0.02):
    """round 6: the same call over TERNARY codes (sign_() left exact zeros: reference runners/base.py:407-410) -- two planes per item, so
    twice the bytes (640 MB); distances in half units.  Checked against the binary call on the zero-free subset property: every list is
    ascending in (distance, index) with indices in range."""
    from xmh import _lib
    from xmh import retrieval as X
    W = (K + 31) // 32
    g = torch.Generator(device="cuda").manual_seed(1815)
    q, r = _codes("iid", R, K, Q)
    def plane(n):
        z = torch.zeros(n, W, dtype=torch.int32, device="cuda")
        for b in range(2):                                   # a few zero bits per word: AND of random words thins them out
            m = torch.randint(-2**31, 2**31 - 1, (n, W), dtype=torch.int32, device="cuda", generator=g)
            for _ in range(5):
                m &= torch.randint(-2**31, 2**31 - 1, (n, W), dtype=torch.int32, device="cuda", generator=g)
            z |= m
        return z
    qz, rz = plane(Q), plane(R)
    qt = X.PackedCodes(q.bits & ~qz, qz, K)                  # a zero element has its sign bit clear
    rt = X.PackedCodes(r.bits & ~rz, rz, K)
    ws = X.TopkWorkspace(Q, R, K, k, "cuda", ternary=True)
    for _ in range(3):
        d, i = X.hamming_topk(qt, rt, k, workspace=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        d, i = X.hamming_topk(qt, rt, k, workspace=ws)
    e1.record()
    torch.cuda.synchronize()
    t_call = e0.elapsed_time(e1) / iters * 1e-3
    _lib.prof_enable(True)
    for _ in range(iters):
        X.hamming_topk(qt, rt, k, workspace=ws)
    torch.cuda.synchronize()
    t, launches = _lib.prof_read("topk_filter")
    _lib.prof_enable(False)
    t *= 1e-3
    dd = d.to(torch.int32) & 0xFFFF
    key = (dd.to(torch.int64) << 32) | i.to(torch.int64)
    ok = bool((key[:, 1:] > key[:, :-1]).all()) and bool((i >= 0).all()) and bool((i < R).all())
    alg = 2 * R * W * 4 + 2 * Q * W * 4
    return {"workload": "exact top-%d of Q=%d over R=%d x %d-bit TERNARY codes (bits + zero planes: %d MB), half-unit distances" % (k, Q, R, K, alg // 10**6),
            "whole_call_ms": t_call * 1e3, "whole_call_GBps": alg / t_call / 1e9, "filter_ms": t * 1e3, "filter_GBps": alg / t / 1e9,
            "filter_frac_of_8TBps": alg / t / 1e9 / HBM_PEAK_GBS, "lists_sorted_distinct_in_range": ok, "zero_fraction": float((rz[:4096] != 0).float().mean())}


def measure_structured(R=10_000_000, K=256, k=100):
    """the sampled threshold is ideal on i.i.d. codes -- the same call on label-correlated codes and on a
    duplicate-heavy gallery (lists overflow -> the robust path recomputes), Q = 1 and 8."""
    out = {}
    for kind in ("structured", "duplicates"):
        for Q in (1, 8):
            m = measure(R=R, K=K, Q=Q, k=k, iters=10, kind=kind)
            out["%s_Q%d" % (kind, Q)] = {x: m[x] for x in ("whole_call_ms", "whole_call_GBps", "avg_launch_ms", "achieved", "workload", "robust_path_launches", "robust_path_alone")}
    return out


def measure_many_queries(R=10_000_000, K=256, Q=5000, k=100, iters=2):
    """SURVEY 8d shape (5), the Q = 5000 end: exact top-k of a whole query set over the unsharded 10 M x 256-bit gallery (the reference
    would need a 5000 x 10 M fp32 distance matrix = 200 GB for this, common/calc_utils.py:51-56).  Whole call only: with this many
    queries per gallery pass the filter is matrix-core bound, not HBM bound."""
    from xmh import retrieval as X
    q, r = _codes("iid", R, K, Q)
    d, i = X.hamming_topk(q, r, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        d, i = X.hamming_topk(q, r, k)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e-3
    # size-independent checks on the full result: lists ascending in (distance, index), indices in range and distinct per query
    dd = d.to(torch.int32) & 0xFFFF
    key = (dd.to(torch.int64) << 32) | i.to(torch.int64)
    ok = bool((key[:, 1:] > key[:, :-1]).all()) and bool((i >= 0).all()) and bool((i < R).all())
    return {"workload": "exact top-%d of Q=%d queries over R=%d x %d-bit gallery on one GPU, iid codes" % (k, Q, R, K), "whole_call_ms": t * 1e3,
            "pairs_per_s_whole_call": Q * R / t, "lists_sorted_distinct_in_range": ok, "mean_kth_distance": float(dd[:, -1].float().mean()),
            "mfma_useful_TOPS_whole_call": 2.0 * Q * R * K / t / 1e12, "mfma_frac_whole_call": 2.0 * Q * R * K / t / 1e12 / I8_MFMA_TOPS}


def measure_cache_defeat(R=10_000_000, K=256, k=100, galleries=4, iters=20):
    """Is the HBM-regime number HBM?  One 10 M x 256-bit gallery is 320 MB against 256 MiB of Infinity Cache, and a loop over it re-reads
    the same bytes; FETCH_SIZE counts cache hits.  Here the calls rotate over `galleries` different 10 M galleries (1.28 GB in all), so
    every byte a call streams was evicted since it was last read.  Same workspace, same queries; filter time from the library's own
    HIP events, whole call from events around the loop."""
    from xmh import _lib
    from xmh import retrieval as X
    W = (K + 31) // 32
    out = {}
    gs = [_codes("iid", R, K, 8, seed=1814 + 7 * j)[1] for j in range(galleries)]
    for Q in (1, 8):
        q = _codes("iid", 16, K, Q)[0]
        ws = X.TopkWorkspace(Q, R, K, k, "cuda")
        alg = R * W * 4 + Q * W * 4 + Q * 4
        res = {}
        for name, pool in (("one_gallery", gs[:1]), ("rotating_%d_galleries" % galleries, gs)):
            for n in range(2 * len(pool)):
                X.hamming_topk(q, pool[n % len(pool)], k, workspace=ws)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for n in range(iters):
                X.hamming_topk(q, pool[n % len(pool)], k, workspace=ws)
            e1.record()
            torch.cuda.synchronize()
            t_call = e0.elapsed_time(e1) / iters * 1e-3
            _lib.prof_enable(True)
            for n in range(iters):
                X.hamming_topk(q, pool[n % len(pool)], k, workspace=ws)
            torch.cuda.synchronize()
            t, launches = _lib.prof_read("topk_filter")
            _lib.prof_enable(False)
            t *= 1e-3
            res[name] = {"filter_GBps": alg / t / 1e9, "filter_frac_of_8TBps": alg / t / 1e9 / HBM_PEAK_GBS, "filter_ms": t * 1e3,
                         "whole_call_ms": t_call * 1e3, "whole_call_GBps": alg / t_call / 1e9}
        a, b = res["one_gallery"], res["rotating_%d_galleries" % galleries]
        res["filter_rate_rotating_over_one"] = b["filter_GBps"] / a["filter_GBps"]
        out["Q%d" % Q] = res
    out["workload"] = "top-%d over 10 M x %d bit: the same gallery every call (320 MB, 1.2 x the Infinity Cache) against %d galleries in rotation (%.2f GB)" % (
        k, K, galleries, galleries * R * W * 4 / 1e9)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--R", type=int, default=10_000_000)
    ap.add_argument("--K", type=int, default=256)
    ap.add_argument("--Q", type=int, default=8)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    print(json.dumps(measure(a.R, a.K, a.Q, a.k, a.iters)))
