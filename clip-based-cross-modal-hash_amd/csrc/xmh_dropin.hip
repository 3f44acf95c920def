// calc_map_k (reference common/calc_utils.py:58-92) as ONE call of the C ABI on device-resident inputs: float codes as the reference's
// callers hold them (runners/base.py:259-264), packed label masks, the mAP back on the host -- the function a binding of the reference
// would bind.  Composes the library's own entry points (pack, pass 1, pass 2 + finalisation): the same kernels, the same bits; what it
// saves is the host work between them (the Python drop-in spent 140 us around a 360 us scan on allocations and six foreign calls).
//
// Two synchronisations inside: the value flags of the packs decide binary / ternary kernels (a 4-byte read), and the result goes back to
// the host like the reference's .cpu().
#include "xmh_common.h"

namespace {

inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

struct DropinLayout {
    int W, Lw;
    size_t qbits, qzero, rbits, rzero, flags, ap, cap, out, scan, scan_bytes, total;
};

int dropin_layout(int64_t Q, int64_t R, int K, int C, DropinLayout* L) {
    if (Q <= 0 || R <= 0 || K <= 0 || C <= 0) return xmh::fail(XMH_EINVAL, "xmh_calc_map_k: bad shape Q=%lld R=%lld K=%d C=%d", (long long)Q, (long long)R, K, C);
    L->W = (K + 31) / 32;
    L->Lw = (C + 31) / 32;
    if (L->W != 1 && L->W != 2 && L->W != 4 && L->W != 8 && L->W != 16 && L->W != 32 && L->W != 64)
        return xmh::fail(XMH_ENOTSUP, "xmh_calc_map_k: K=%d needs the caller to widen the codes to a power-of-two word count first", K);
    xmh_scan_plan pb, pt;
    if (const int rc = xmh_scan_plan_make(Q, R, K, 0, &pb)) return rc;
    size_t scan = pb.ws_bytes;
    if (K <= 256 && xmh_scan_plan_make(Q, R, K, 1, &pt) == XMH_OK && pt.ws_bytes > scan) scan = pt.ws_bytes;      // ternary codes: up to 256 bits
    size_t o = 0;
    auto take = [&](size_t b) { const size_t at = o; o += up256(b); return at; };
    L->qbits = take((size_t)Q * L->W * 4); L->qzero = take((size_t)Q * L->W * 4);
    L->rbits = take((size_t)R * L->W * 4); L->rzero = take((size_t)R * L->W * 4);
    L->flags = take(4); L->ap = take((size_t)Q * 8); L->cap = take((size_t)Q * 4); L->out = take(8);
    L->scan = take(scan);
    L->scan_bytes = scan;
    L->total = o;
    return XMH_OK;
}

}  // namespace

extern "C" size_t xmh_calc_map_k_ws_bytes(int64_t Q, int64_t R, int K, int C) {
    DropinLayout L;
    return dropin_layout(Q, R, K, C, &L) == XMH_OK ? L.total : 0;
}

extern "C" int xmh_calc_map_k(const float* qB, const float* rB, const uint32_t* qlab, const uint32_t* rlab, int64_t Q, int64_t R, int K, int C,
                              int64_t k, void* ws, size_t ws_bytes, double* map_host, int32_t* flags_host, xmh_stream_t stream) {
    XMH_RANGE("xmh_calc_map_k");
    DropinLayout L;
    if (const int rc = dropin_layout(Q, R, K, C, &L)) return rc;
    if (!qB || !rB || !qlab || !rlab || !ws || !map_host || !flags_host) return xmh::fail(XMH_EINVAL, "xmh_calc_map_k: null pointer");
    if (ws_bytes < L.total) return xmh::fail(XMH_EINVAL, "xmh_calc_map_k: workspace too small (%zu < %zu)", ws_bytes, L.total);
    char* w = static_cast<char*>(ws);
    uint32_t *qb = reinterpret_cast<uint32_t*>(w + L.qbits), *qz = reinterpret_cast<uint32_t*>(w + L.qzero);
    uint32_t *rb = reinterpret_cast<uint32_t*>(w + L.rbits), *rz = reinterpret_cast<uint32_t*>(w + L.rzero);
    int32_t* fl = reinterpret_cast<int32_t*>(w + L.flags);
    hipStream_t st = xmh::as_stream(stream);
    XMH_HIP(hipMemsetAsync(fl, 0, 4, st));
    if (const int rc = xmh_pack_sign(qB, Q, K, nullptr, qb, qz, fl, stream)) return rc;
    if (const int rc = xmh_pack_sign(rB, R, K, nullptr, rb, rz, fl, stream)) return rc;
    XMH_HIP(hipMemcpyAsync(flags_host, fl, 4, hipMemcpyDeviceToHost, st));
    XMH_HIP(hipStreamSynchronize(st));
    const int flags = *flags_host;
    if (flags & 2) return XMH_OK;                            // values outside {-1, 0, +1}: the caller's float path (no result written)
    const bool tern = flags & 1;                             // an exact zero somewhere: sign(0) = 0, the ternary kernels
    if (tern && K > 256) return xmh::fail(XMH_ENOTSUP, "xmh_calc_map_k: ternary codes (an exact 0 among the values) are supported up to 256 bits, K=%d", K);
    const uint32_t *qzp = tern ? qz : nullptr, *rzp = tern ? rz : nullptr;
    // the size handed to the scan decides about its pair cache: all of the scan region (sized for the larger of the two plans)
    if (const int rc = xmh_hamming_hist(qb, qzp, qlab, rb, rzp, rlab, Q, R, K, C, w + L.scan, L.scan_bytes, nullptr, nullptr, stream)) return rc;
    double* out = reinterpret_cast<double*>(w + L.out);
    if (const int rc = xmh_hamming_map(qb, qzp, qlab, rb, rzp, rlab, Q, R, K, C, w + L.scan, L.scan_bytes, k > 0 ? k : 0,
                                       reinterpret_cast<double*>(w + L.ap), reinterpret_cast<int32_t*>(w + L.cap), out, stream)) return rc;
    XMH_HIP(hipMemcpyAsync(map_host, out, 8, hipMemcpyDeviceToHost, st));
    XMH_HIP(hipStreamSynchronize(st));
    return XMH_OK;
}
