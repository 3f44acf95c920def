// Fused ranking scan: calc_map_k (reference common/calc_utils.py:58-92) as two streaming passes over
// bit-packed codes, no [Q,R] intermediate, no sort.
//
// Execution shape ("query per lane"): a wave owns 64 queries -- their code words and label masks live in
// VGPRs -- and walks a contiguous gallery chunk in index order.  The gallery record is wave-uniform, so it
// arrives through SCALAR loads (s_load_dwordxN into SGPRs) and feeds v_xor_b32 / v_bcnt_u32_b32 as the
// scalar operand; HBM sees every gallery byte once per (query tile, chunk) and the L2 of the XCD the
// chunk is pinned to serves the other query tiles.  Per-lane bucket counters sit in LDS as cnt[d][lane]
// (row stride = 64 lanes), so lane l always hits bank l%32: conflict-free ds_add for any distance pattern.
// Walking in index order makes "number of same-distance items with a smaller index" a running counter,
// which is exactly the tie-break of the canonical (distance, index) order.
//
// Bound: VALU: ~10 lane-ops/pair in pass 1, ~20 in pass 2, vs K/8+4*Lw bytes per gallery item
// shared by 64 queries.  Algorithmic HBM bytes per launch: R*(4W+4Lw) + Q*(4W+4Lw) + workspace.
#include "xmh_common.h"

#include <type_traits>

namespace {

constexpr int kMaxChunk = 32768;   // u16 halves of the packed pass-1 counters must not overflow
constexpr int kMinChunk = 256;

struct ScanArgs {
    const uint32_t* qbits;
    const uint32_t* qzero;
    const uint32_t* qlab;
    const uint32_t* rbits;
    const uint32_t* rzero;
    const uint32_t* rlab;
    int Q, R, K;
    int chunk, nchunk, nqt, qpad, nb;
};

// blockIdx -> (chunk, query tile).  Block b runs on XCD b%8 (observed, speed only): pin chunk c to XCD c%8
// and sweep the query tiles of one chunk back-to-back so the chunk stays in that XCD's L2.
__device__ __forceinline__ bool map_block(const ScanArgs& a, int& chunk_id, int& qtile) {
    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int t = b >> 3;
    qtile = t % a.nqt;
    chunk_id = xcd + 8 * (t / a.nqt);
    return chunk_id < a.nchunk;
}

template <int W, int LW, bool TERN>
struct QueryRegs {
    uint32_t b[W];
    uint32_t z[TERN ? W : 1];
    uint32_t l[LW];
    __device__ __forceinline__ void load(const ScanArgs& a, int q) {
        const bool ok = q < a.Q;
#pragma unroll
        for (int w = 0; w < W; ++w) b[w] = ok ? a.qbits[(int64_t)q * W + w] : 0u;
        if (TERN) {
#pragma unroll
            for (int w = 0; w < W; ++w) z[w] = ok ? a.qzero[(int64_t)q * W + w] : 0xffffffffu;
        }
#pragma unroll
        for (int w = 0; w < LW; ++w) l[w] = ok ? a.qlab[(int64_t)q * LW + w] : 0u;
    }
};

// A batch of 64 consecutive gallery records, one per lane (coalesced vector loads, tracked by vmcnt so the
// next batch can be in flight while the LDS counters of this one are being updated).  Record u of the batch
// is broadcast to all lanes with v_readlane (-> SGPR operands of the XOR/AND), which costs W+LW VALU issues
// per item but no LDS bandwidth and, unlike scalar loads, does not share a wait counter with the LDS atomics.
template <int W, int LW, bool TERN>
struct GalleryBatch {
    uint32_t b[W];
    uint32_t z[TERN ? W : 1];
    uint32_t l[LW];
    __device__ __forceinline__ void load(const ScanArgs& a, int64_t base, int64_t hi, int lane) {
        const int64_t i = base + lane;
        const bool ok = i < hi;
#pragma unroll
        for (int w = 0; w < W; ++w) b[w] = ok ? a.rbits[i * W + w] : 0u;
        if (TERN) {
#pragma unroll
            for (int w = 0; w < W; ++w) z[w] = ok ? a.rzero[i * W + w] : 0xffffffffu;
        }
#pragma unroll
        for (int w = 0; w < LW; ++w) l[w] = ok ? a.rlab[i * LW + w] : 0u;
    }
};

// distance bucket + relevance of (this lane's query, record u of the batch)
template <int W, int LW, bool TERN>
__device__ __forceinline__ void pair_eval(const QueryRegs<W, LW, TERN>& qr, const GalleryBatch<W, LW, TERN>& g, int u, int K,
                                          int& d, bool& rel) {
    if (!TERN) {
        int acc = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) acc += __popc(qr.b[w] ^ (uint32_t)__builtin_amdgcn_readlane((int)g.b[w], u));
        d = acc;
    } else {
        int live_n = 0, diff_n = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const uint32_t rb = (uint32_t)__builtin_amdgcn_readlane((int)g.b[w], u);
            const uint32_t rz = (uint32_t)__builtin_amdgcn_readlane((int)g.z[w], u);
            const uint32_t live = ~(qr.z[w] | rz);
            live_n += __popc(live);
            diff_n += __popc((qr.b[w] ^ rb) & live);
        }
        d = K - live_n + 2 * diff_n;                           // 2 * (0.5 * (K - q.r)), in [0, 2K]
    }
    uint32_t hit = 0;
#pragma unroll
    for (int w = 0; w < LW; ++w) hit |= qr.l[w] & (uint32_t)__builtin_amdgcn_readlane((int)g.l[w], u);
    rel = hit != 0;
}

template <int W>
struct Unroll {
    static constexpr int value = W <= 2 ? 8 : 4;
};

// ---------------------------------------------------------------------------------------------------
// pass 1: chunk_hist[chunk][d][q] = (#items at distance d) | (#relevant items at distance d) << 16
// ---------------------------------------------------------------------------------------------------
template <int W, int LW, bool TERN>
__global__ __launch_bounds__(64) void k_scan_hist(ScanArgs a, uint32_t* __restrict__ chunk_hist) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];   // [nb][64]
    int chunk_id, qtile;
    if (!map_block(a, chunk_id, qtile)) return;
    const int lane = threadIdx.x;
    const int q = qtile * 64 + lane;
    for (int d = 0; d < a.nb; ++d) lds[d * 64 + lane] = 0u;
    QueryRegs<W, LW, TERN> qr;
    qr.load(a, q);

    const int64_t lo = (int64_t)chunk_id * a.chunk;
    const int64_t hi = (lo + a.chunk < a.R) ? lo + a.chunk : a.R;
    constexpr int U = Unroll<W>::value;
    GalleryBatch<W, LW, TERN> cur, nxt;
    cur.load(a, lo, hi, lane);
    for (int64_t base = lo; base < hi; base += 64) {
        nxt.load(a, base + 64, hi, lane);                      // prefetch (all-zero past the end)
        const int cnt = (hi - base < 64) ? (int)(hi - base) : 64;
        int u0 = 0;
        for (; u0 + U <= cnt; u0 += U) {
            int d[U];
            bool rel[U];
#pragma unroll
            for (int u = 0; u < U; ++u) pair_eval<W, LW, TERN>(qr, cur, u0 + u, a.K, d[u], rel[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) atomicAdd(&lds[d[u] * 64 + lane], rel[u] ? 0x10001u : 1u);
        }
        for (; u0 < cnt; ++u0) {
            int d;
            bool rel;
            pair_eval<W, LW, TERN>(qr, cur, u0, a.K, d, rel);
            atomicAdd(&lds[d * 64 + lane], rel ? 0x10001u : 1u);
        }
        cur = nxt;
    }
    uint32_t* __restrict__ out = chunk_hist + ((int64_t)chunk_id * a.nb) * a.qpad + q;
    for (int d = 0; d < a.nb; ++d) out[(int64_t)d * a.qpad] = lds[d * 64 + lane];
}

// shard totals for the multi-GPU exchange: hist_all/hist_rel [Q][nb]
__global__ __launch_bounds__(256) void k_hist_totals(const uint32_t* __restrict__ chunk_hist, int Q, int qpad, int nb,
                                                     int nchunk, uint32_t* __restrict__ hist_all,
                                                     uint32_t* __restrict__ hist_rel) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int d = blockIdx.y;
    if (q >= Q) return;
    uint32_t sa = 0, sr = 0;
    for (int c = 0; c < nchunk; ++c) {
        const uint32_t h = chunk_hist[((int64_t)c * nb + d) * qpad + q];
        sa += h & 0xffffu;
        sr += h >> 16;
    }
    if (hist_all) hist_all[(int64_t)q * nb + d] = sa;
    if (hist_rel) hist_rel[(int64_t)q * nb + d] = sr;
}

// ---------------------------------------------------------------------------------------------------
// pass 2: ranks of relevant items.  cnt[d][lane] is a 64-bit counter {lo: rank of the NEXT item of bucket d,
// hi: ordinal of the NEXT relevant item of bucket d} (both 1-based); it starts at the bucket's global base
// and one ds_add_rtn_u64 per pair both advances it and returns (rank, ordinal) of the current item.
// The loop is software-pipelined by one group: the atomics of group g are in flight while group g+1's
// distances are computed, and only then are group g's returns credited.
// ---------------------------------------------------------------------------------------------------
template <int W, int LW, bool TERN, bool CAPPED>
__global__ __launch_bounds__(64) void k_scan_ap(ScanArgs a, const uint32_t* __restrict__ chunk_hist,
                                                const uint32_t* __restrict__ base_all,
                                                const uint32_t* __restrict__ base_rel,
                                                const uint32_t* __restrict__ nrel_total, int64_t kcap,
                                                float* __restrict__ ap_part, int32_t* __restrict__ cap_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long cnt[];   // [nb][64]
    int chunk_id, qtile;
    if (!map_block(a, chunk_id, qtile)) return;
    const int lane = threadIdx.x;
    const int q = qtile * 64 + lane;
    const bool qok = q < a.Q;

    // bucket bases: (everything in lower buckets) + (same bucket, lower chunks) [+ other shards]
    uint32_t run_all = 0, run_rel = 0;
    for (int d = 0; d < a.nb; ++d) {
        uint32_t tot_a = 0, tot_r = 0, below_a = 0, below_r = 0;
        for (int c = 0; c < a.nchunk; ++c) {
            const uint32_t h = chunk_hist[((int64_t)c * a.nb + d) * a.qpad + q];
            const uint32_t ha = h & 0xffffu, hr = h >> 16;
            tot_a += ha;
            tot_r += hr;
            if (c < chunk_id) {
                below_a += ha;
                below_r += hr;
            }
        }
        uint32_t ba, br;
        if (base_all) {
            ba = (qok ? base_all[(int64_t)q * a.nb + d] : 0u) + below_a;
            br = (qok ? base_rel[(int64_t)q * a.nb + d] : 0u) + below_r;
        } else {
            ba = run_all + below_a;
            br = run_rel + below_r;
        }
        run_all += tot_a;
        run_rel += tot_r;
        cnt[d * 64 + lane] = (unsigned long long)(ba + 1u) | ((unsigned long long)(br + 1u) << 32);
    }
    const uint32_t nrel = nrel_total ? (qok ? nrel_total[q] : 0u) : run_rel;
    const uint32_t cap = (kcap > 0 && (uint64_t)kcap < (uint64_t)nrel) ? (uint32_t)kcap : nrel;
    if (chunk_id == 0 && qok) cap_out[q] = (int32_t)cap;

    QueryRegs<W, LW, TERN> qr;
    qr.load(a, q);
    const int64_t lo = (int64_t)chunk_id * a.chunk;
    const int64_t hi = (lo + a.chunk < a.R) ? lo + a.chunk : a.R;
    float acc = 0.0f;

    // (rank, ordinal) of a relevant item -> ordinal / rank.  v_rcp_f32 is good to 1 ulp: a term is off by
    // <= 1.5e-7 relative, far inside the 1e-4 mAP tolerance (measured ~1e-8 on the goldens).
    auto credit = [&](unsigned long long old, bool rel) {
        const uint32_t rank = (uint32_t)old;
        const uint32_t ord = (uint32_t)(old >> 32);
        const float term = (float)ord * __builtin_amdgcn_rcpf((float)rank);
        const bool take = CAPPED ? (rel && ord <= cap) : rel;
        acc += take ? term : 0.0f;
    };

    constexpr int U = Unroll<W>::value;
    GalleryBatch<W, LW, TERN> cur, nxt;
    cur.load(a, lo, hi, lane);
    for (int64_t base = lo; base < hi; base += 64) {
        nxt.load(a, base + 64, hi, lane);                      // prefetch (all-zero past the end)
        const int cntb = (hi - base < 64) ? (int)(hi - base) : 64;
        const int ngroups = cntb / U;
        unsigned long long old[U];
        bool relp[U];
        if (ngroups > 0) {
            int d[U];
#pragma unroll
            for (int u = 0; u < U; ++u) pair_eval<W, LW, TERN>(qr, cur, u, a.K, d[u], relp[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) old[u] = atomicAdd(&cnt[d[u] * 64 + lane], relp[u] ? 0x100000001ull : 1ull);
        }
        for (int g = 1; g < ngroups; ++g) {
            int d[U];
            bool rel[U];
#pragma unroll
            for (int u = 0; u < U; ++u) pair_eval<W, LW, TERN>(qr, cur, g * U + u, a.K, d[u], rel[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) credit(old[u], relp[u]);                  // previous group's returns
#pragma unroll
            for (int u = 0; u < U; ++u) {
                old[u] = atomicAdd(&cnt[d[u] * 64 + lane], rel[u] ? 0x100000001ull : 1ull);
                relp[u] = rel[u];
            }
        }
        if (ngroups > 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) credit(old[u], relp[u]);
        }
        for (int u0 = ngroups * U; u0 < cntb; ++u0) {
            int d;
            bool rel;
            pair_eval<W, LW, TERN>(qr, cur, u0, a.K, d, rel);
            credit(atomicAdd(&cnt[d * 64 + lane], rel ? 0x100000001ull : 1ull), rel);
        }
        cur = nxt;
    }
    ap_part[(int64_t)chunk_id * a.qpad + q] = acc;
}

__global__ __launch_bounds__(256) void k_ap_reduce(const float* __restrict__ ap_part, int Q, int qpad, int nchunk,
                                                   double* __restrict__ ap_sum) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= Q) return;
    double s = 0.0;
    for (int c = 0; c < nchunk; ++c) s += (double)ap_part[(int64_t)c * qpad + q];
    ap_sum[q] = s;
}

__global__ __launch_bounds__(256) void k_map_finalize(const double* __restrict__ ap_sum, const int32_t* __restrict__ cap,
                                                      int64_t Q, double* __restrict__ map_out) {
    __shared__ double part[256];
    double s = 0.0;
    for (int64_t q = threadIdx.x; q < Q; q += 256) s += ap_sum[q] / (double)cap[q];   // cap == 0 -> NaN (0/0)
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) map_out[0] = part[0] / (double)Q;
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
int make_plan(int64_t Q, int64_t R, int K, int ternary, xmh_scan_plan* p) {
    if (Q <= 0 || R <= 0 || K <= 0) return xmh::fail(XMH_EINVAL, "scan plan: bad shape Q=%lld R=%lld K=%d", (long long)Q, (long long)R, K);
    if (R >= (1ll << 31) || Q >= (1ll << 24)) return xmh::fail(XMH_ENOTSUP, "scan plan: shard too large (R=%lld, Q=%lld)", (long long)R, (long long)Q);
    const int64_t nb = ternary ? 2 * (int64_t)K + 1 : (int64_t)K + 1;
    const int64_t lds_ap = nb * 64 * 8;
    if (lds_ap > 160 * 1024) return xmh::fail(XMH_ENOTSUP, "scan plan: %lld distance buckets need %lld B of LDS per wave (max 163840); K=%d%s", (long long)nb, (long long)lds_ap, K, ternary ? " ternary" : "");
    const int64_t nqt = xmh::ceil_div(Q, 64);
    int64_t wpc = (160 * 1024) / lds_ap;          // waves per CU the pass-2 LDS footprint admits
    if (wpc > 8) wpc = 8;
    const int64_t slots = (int64_t)xmh::device_cu_count() * wpc;
    int64_t nchunk = slots / nqt;
    if (nchunk < 1) nchunk = 1;
    int64_t chunk = xmh::ceil_div(R, nchunk);
    if (chunk < kMinChunk) chunk = kMinChunk;
    if (chunk > kMaxChunk) chunk = kMaxChunk;
    chunk = xmh::ceil_div(chunk, 8) * 8;
    nchunk = xmh::ceil_div(R, chunk);
    p->chunk = chunk;
    p->nchunk = nchunk;
    p->nqtile = nqt;
    p->qpad = nqt * 64;
    p->nbuckets = nb;
    p->ws_bytes = (size_t)(nchunk * nb * p->qpad) * 4 + (size_t)(nchunk * p->qpad) * 4;
    return XMH_OK;
}

template <bool TERN, typename F>
int dispatch_shape(int W, int LW, F&& f) {
#define XMH_CASE(WW, LL) \
    if (W == WW && LW == LL) return f(std::integral_constant<int, WW>{}, std::integral_constant<int, LL>{});
    XMH_CASE(1, 1) XMH_CASE(1, 2) XMH_CASE(1, 3) XMH_CASE(1, 4)
    XMH_CASE(2, 1) XMH_CASE(2, 2) XMH_CASE(2, 3) XMH_CASE(2, 4)
    XMH_CASE(4, 1) XMH_CASE(4, 2) XMH_CASE(4, 3) XMH_CASE(4, 4)
    XMH_CASE(8, 1) XMH_CASE(8, 2) XMH_CASE(8, 3) XMH_CASE(8, 4)
#undef XMH_CASE
    return xmh::fail(XMH_ENOTSUP, "scan: unsupported shape W=%d code words (K in {<=32,64,128,256}), Lw=%d label words (C<=128)", W, LW);
}

int check_common(const char* who, const uint32_t* qbits, const uint32_t* qzero, const uint32_t* qlab,
                 const uint32_t* rbits, const uint32_t* rzero, const uint32_t* rlab, int C, void* ws, size_t ws_bytes,
                 const xmh_scan_plan& p) {
    if (!qbits || !rbits || !qlab || !rlab || !ws) return xmh::fail(XMH_EINVAL, "%s: null pointer", who);
    if ((qzero == nullptr) != (rzero == nullptr)) return xmh::fail(XMH_EINVAL, "%s: zero masks must be given for both sides or neither", who);
    if (C <= 0) return xmh::fail(XMH_EINVAL, "%s: C=%d", who, C);
    if (ws_bytes < p.ws_bytes) return xmh::fail(XMH_EINVAL, "%s: workspace too small (%zu < %zu)", who, ws_bytes, p.ws_bytes);
    return XMH_OK;
}

ScanArgs make_args(const uint32_t* qbits, const uint32_t* qzero, const uint32_t* qlab, const uint32_t* rbits,
                   const uint32_t* rzero, const uint32_t* rlab, int64_t Q, int64_t R, int K, const xmh_scan_plan& p) {
    ScanArgs a;
    a.qbits = qbits; a.qzero = qzero; a.qlab = qlab;
    a.rbits = rbits; a.rzero = rzero; a.rlab = rlab;
    a.Q = (int)Q; a.R = (int)R; a.K = K;
    a.chunk = (int)p.chunk; a.nchunk = (int)p.nchunk; a.nqt = (int)p.nqtile; a.qpad = (int)p.qpad; a.nb = (int)p.nbuckets;
    return a;
}

inline int scan_grid(const xmh_scan_plan& p) { return (int)(8 * p.nqtile * xmh::ceil_div(p.nchunk, 8)); }

}  // namespace

extern "C" int xmh_scan_plan_make(int64_t Q, int64_t R, int K, int ternary, xmh_scan_plan* plan_host) {
    if (!plan_host) return xmh::fail(XMH_EINVAL, "xmh_scan_plan_make: null plan");
    return make_plan(Q, R, K, ternary, plan_host);
}

extern "C" int xmh_hamming_hist(const uint32_t* qbits, const uint32_t* qzero, const uint32_t* qlab,
                                const uint32_t* rbits, const uint32_t* rzero, const uint32_t* rlab, int64_t Q, int64_t R,
                                int K, int C, void* ws, size_t ws_bytes, uint32_t* hist_all, uint32_t* hist_rel,
                                xmh_stream_t stream) {
    const bool tern = qzero != nullptr;
    xmh_scan_plan p;
    int rc = make_plan(Q, R, K, tern, &p);
    if (rc) return rc;
    rc = check_common("xmh_hamming_hist", qbits, qzero, qlab, rbits, rzero, rlab, C, ws, ws_bytes, p);
    if (rc) return rc;
    const ScanArgs a = make_args(qbits, qzero, qlab, rbits, rzero, rlab, Q, R, K, p);
    uint32_t* chunk_hist = static_cast<uint32_t*>(ws);
    hipStream_t st = xmh::as_stream(stream);
    const size_t lds = (size_t)p.nbuckets * 64 * 4;
    const int W = (K + 31) / 32, LW = (C + 31) / 32;
    auto launch = [&](auto tern_c) {
        constexpr bool T = decltype(tern_c)::value;
        return dispatch_shape<T>(W, LW, [&](auto w, auto l) {
            auto kern = k_scan_hist<decltype(w)::value, decltype(l)::value, T>;
            if (lds > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return xmh::fail(XMH_EHIP, "xmh_hamming_hist: cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e));
            }
            hipLaunchKernelGGL(kern, dim3(scan_grid(p)), dim3(64), lds, st, a, chunk_hist);
            return XMH_OK;
        });
    };
    rc = tern ? launch(std::true_type{}) : launch(std::false_type{});
    if (rc) return rc;
    XMH_LAUNCH_CHECK("xmh_hamming_hist");
    if (hist_all || hist_rel) {
        hipLaunchKernelGGL(k_hist_totals, dim3((unsigned)xmh::ceil_div(Q, 256), (unsigned)p.nbuckets), dim3(256), 0, st, chunk_hist,
                           (int)Q, (int)p.qpad, (int)p.nbuckets, (int)p.nchunk, hist_all, hist_rel);
        XMH_LAUNCH_CHECK("xmh_hamming_hist totals");
    }
    return XMH_OK;
}

extern "C" int xmh_hamming_ap(const uint32_t* qbits, const uint32_t* qzero, const uint32_t* qlab, const uint32_t* rbits,
                              const uint32_t* rzero, const uint32_t* rlab, int64_t Q, int64_t R, int K, int C, void* ws,
                              size_t ws_bytes, const uint32_t* base_all, const uint32_t* base_rel,
                              const uint32_t* nrel_total, int64_t k, double* ap_sum, int32_t* cap, xmh_stream_t stream) {
    const bool tern = qzero != nullptr;
    xmh_scan_plan p;
    int rc = make_plan(Q, R, K, tern, &p);
    if (rc) return rc;
    rc = check_common("xmh_hamming_ap", qbits, qzero, qlab, rbits, rzero, rlab, C, ws, ws_bytes, p);
    if (rc) return rc;
    if (!ap_sum || !cap) return xmh::fail(XMH_EINVAL, "xmh_hamming_ap: null output");
    const int ext = (base_all != nullptr) + (base_rel != nullptr) + (nrel_total != nullptr);
    if (ext != 0 && ext != 3) return xmh::fail(XMH_EINVAL, "xmh_hamming_ap: base_all, base_rel and nrel_total go together");
    const ScanArgs a = make_args(qbits, qzero, qlab, rbits, rzero, rlab, Q, R, K, p);
    const uint32_t* chunk_hist = static_cast<const uint32_t*>(ws);
    float* ap_part = reinterpret_cast<float*>(static_cast<char*>(ws) + (size_t)(p.nchunk * p.nbuckets * p.qpad) * 4);
    hipStream_t st = xmh::as_stream(stream);
    const size_t lds = (size_t)p.nbuckets * 64 * 8;
    const int W = (K + 31) / 32, LW = (C + 31) / 32;
    auto launch = [&](auto tern_c) {
        constexpr bool T = decltype(tern_c)::value;
        return dispatch_shape<T>(W, LW, [&](auto w, auto l) {
            auto go = [&](auto kern) {
                if (lds > 64 * 1024) {
                    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    if (e != hipSuccess) return xmh::fail(XMH_EHIP, "xmh_hamming_ap: cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e));
                }
                hipLaunchKernelGGL(kern, dim3(scan_grid(p)), dim3(64), lds, st, a, chunk_hist, base_all, base_rel, nrel_total, k, ap_part, cap);
                return (int)XMH_OK;
            };
            return k > 0 ? go(k_scan_ap<decltype(w)::value, decltype(l)::value, T, true>)
                         : go(k_scan_ap<decltype(w)::value, decltype(l)::value, T, false>);
        });
    };
    rc = tern ? launch(std::true_type{}) : launch(std::false_type{});
    if (rc) return rc;
    XMH_LAUNCH_CHECK("xmh_hamming_ap");
    hipLaunchKernelGGL(k_ap_reduce, dim3((unsigned)xmh::ceil_div(Q, 256)), dim3(256), 0, st, ap_part, (int)Q, (int)p.qpad, (int)p.nchunk, ap_sum);
    XMH_LAUNCH_CHECK("xmh_hamming_ap reduce");
    return XMH_OK;
}

extern "C" int xmh_map_finalize(const double* ap_sum, const int32_t* cap, int64_t Q, double* map_out, xmh_stream_t stream) {
    if (!ap_sum || !cap || !map_out || Q <= 0) return xmh::fail(XMH_EINVAL, "xmh_map_finalize: bad arguments");
    hipLaunchKernelGGL(k_map_finalize, dim3(1), dim3(256), 0, xmh::as_stream(stream), ap_sum, cap, Q, map_out);
    XMH_LAUNCH_CHECK("xmh_map_finalize");
    return XMH_OK;
}
