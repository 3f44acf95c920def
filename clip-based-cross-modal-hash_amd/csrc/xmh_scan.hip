// Fused ranking scan: calc_map_k (reference common/calc_utils.py:58-92) as two streaming passes over
// bit-packed codes, no [Q,R] intermediate, no sort.
//
// Execution shape ("query per lane"): a wave owns 64 queries -- their code words and label masks live in
// VGPRs -- and walks a contiguous gallery chunk in index order.  Per-lane bucket counters sit in LDS as
// cnt[d][lane] (row stride = 64 lanes), so lane l always hits bank l%32: conflict-free for any distance
// pattern, and because every lane owns its column the read-modify-write needs no cross-lane atomicity.
// Walking in index order makes "number of same-distance items with a smaller index" a running counter,
// which is exactly the tie-break of the canonical (distance, index) order.
//
// The gallery record is wave-uniform.  Two ways to broadcast it (XMH_SCAN_GM_HIST / XMH_SCAN_GM_AP; both measured
// within a few percent of each other on MI355X):
//   GM_LDS (2, default)  the wave stages 64 records at a time into a small word-major LDS ring and fetches word x of
//              four consecutive items with ONE wave-uniform ds_read_b128 (LDS broadcast).  LDS ops of a wave
//              complete in order, so every wait in the loop is a counted lgkmcnt; the next batch's global loads
//              (vmcnt) fly during the whole batch.  Record words arrive in VGPRs.
//   GM_SCALAR (0)  hand-issued s_load_dwordxN into SGPRs (inline asm, invisible to the compiler's waitcnt pass), used
//              directly as the scalar operand of v_xor / v_and_or.  SMEM returns out of order and shares lgkmcnt
//              with LDS, so each group has exactly one explicit lgkmcnt(0); the next group's loads and the previous
//              group's atomics are issued right after it and complete under the current group's VALU work.
//
// Bound: VALU issue.  tools/ubench_valu.hip measures ~4.0-4.4 cycles per wave64 instruction per SIMD for
// the instruction mix of these loops at any occupancy (v_xor/v_and/v_add alone reach 2.4, VOP3 ops such as
// v_bcnt_u32_b32 / v_and_or_b32 / v_lshl_add_u32 4.2, v_rcp_f32 8.2): 10 (pass 1) / 14 (pass 2) VALU instructions per
// wave-item at K=64, C=80, against K/8+4*Lw gallery bytes shared by the 64 queries of a wave.  The relevance test
// is one v_and_or_b32 per label word + one v_min_u32 (inline asm: hipcc does not form them).
// Algorithmic HBM bytes per launch: R*(4W+4Lw) + Q*(4W+4Lw) (+ the bucket tables in the workspace).
#include "xmh_common.h"

#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int kMaxChunk = 32768;   // u16 halves of the packed pass-1 counters must not overflow
constexpr int kMinChunk = 256;
constexpr int GM_SCALAR = 0;
constexpr int GM_LDS = 2;

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

// one gallery record as plain words (uniform across the wave)
template <int W, int LW, bool TERN>
struct Rec {
    uint32_t b[W];
    uint32_t z[TERN ? W : 1];
    uint32_t l[LW];
};

// distance bucket d and relevance (as 0/1 in `hit01`) of (this lane's query, record r)
// VREC = the record words live in VGPRs (GM_LDS): relevance then uses v_and_or_b32 / v_min_u32 (one op per label word
// + one) through inline asm -- hipcc does not form them -- which would cost extra v_movs on SGPR-resident records.
template <int W, int LW, bool TERN, int VREC = 0>
__device__ __forceinline__ void rec_eval01(const QueryRegs<W, LW, TERN>& qr, const Rec<W, LW, TERN>& r, int K, int& d, uint32_t& hit01) {
    if (!TERN) {
        int acc = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) acc += __popc(qr.b[w] ^ r.b[w]);
        d = acc;
    } else {
        int live_n = 0, diff_n = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const uint32_t live = ~(qr.z[w] | r.z[w]);
            live_n += __popc(live);
            diff_n += __popc((qr.b[w] ^ r.b[w]) & live);
        }
        d = K - live_n + 2 * diff_n;                           // 2 * (0.5 * (K - q.r)), in [0, 2K]
    }
    if (VREC == 1) {
        uint32_t hit = qr.l[0] & r.l[0];
#pragma unroll
        for (int w = 1; w < LW; ++w) asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(hit) : "v"(qr.l[w]), "v"(r.l[w]));
        asm("v_min_u32 %0, %1, 1" : "=v"(hit01) : "v"(hit));
    } else if (VREC == 2) {                                      // record words wave-uniform in SGPRs (one SGPR per VOP3)
        uint32_t hit = qr.l[0] & r.l[0];
#pragma unroll
        for (int w = 1; w < LW; ++w) asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(hit) : "s"(r.l[w]), "v"(qr.l[w]));
        asm("v_min_u32 %0, %1, 1" : "=v"(hit01) : "v"(hit));
    } else {
        uint32_t hit = 0;
#pragma unroll
        for (int w = 0; w < LW; ++w) hit |= qr.l[w] & r.l[w];
        hit01 = hit != 0 ? 1u : 0u;
    }
}

template <int W, int LW, bool TERN>
__device__ __forceinline__ void rec_eval(const QueryRegs<W, LW, TERN>& qr, const Rec<W, LW, TERN>& r, int K, int& d, bool& rel) {
    uint32_t h;
    rec_eval01<W, LW, TERN>(qr, r, K, d, h);
    rel = h != 0;
}

// GM_SCALAR: record i through a wave-uniform address (the compiler emits s_load_dwordxN)
template <int W, int LW, bool TERN>
__device__ __forceinline__ void rec_load_uniform(Rec<W, LW, TERN>& r, const ScanArgs& a, int64_t i) {
    const uint32_t* __restrict__ pb = a.rbits + i * W;
#pragma unroll
    for (int w = 0; w < W; ++w) r.b[w] = pb[w];
    if (TERN) {
        const uint32_t* __restrict__ pz = a.rzero + i * W;
#pragma unroll
        for (int w = 0; w < W; ++w) r.z[w] = pz[w];
    }
    const uint32_t* __restrict__ pl = a.rlab + i * LW;
#pragma unroll
    for (int w = 0; w < LW; ++w) r.l[w] = pl[w];
}

// GM_SCALAR, explicit form: N consecutive dwords fetched into SGPRs by hand-issued s_load_dwordxN.  The compiler's
// waitcnt pass does not see these loads, which is the point: the kernel decides where the one lgkmcnt(0) per group goes
// (see k_scan_hist).  Protocol: issue() ... __builtin_amdgcn_s_waitcnt(lgkmcnt 0) ... fence() ... word(i).  fence() is
// an empty asm that makes every later use depend on a point after the wait.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
template <int P> struct SPiece;
template <> struct SPiece<1> {
    uint32_t v;
    template <int OFF> __device__ __forceinline__ void issue(const uint32_t* p) { asm volatile("s_load_dword %0, %1, %2" : "=&s"(v) : "s"(p), "n"(OFF)); }
    __device__ __forceinline__ void fence() { asm volatile("" : "+s"(v)); }
    __device__ __forceinline__ uint32_t word(int) const { return v; }
};
#define XMH_SPIECE(P, T, INSN)                                                                                              \
    template <> struct SPiece<P> {                                                                                          \
        T v;                                                                                                                \
        template <int OFF> __device__ __forceinline__ void issue(const uint32_t* p) {                                       \
            asm volatile(INSN " %0, %1, %2" : "=&s"(v) : "s"(p), "n"(OFF));                                                 \
        }                                                                                                                   \
        __device__ __forceinline__ void fence() { asm volatile("" : "+s"(v)); }                                             \
        __device__ __forceinline__ uint32_t word(int i) const { return v[i]; }                                              \
    };
XMH_SPIECE(2, u32x2, "s_load_dwordx2")
XMH_SPIECE(4, u32x4, "s_load_dwordx4")
XMH_SPIECE(8, u32x8, "s_load_dwordx8")
XMH_SPIECE(16, u32x16, "s_load_dwordx16")
#undef XMH_SPIECE

template <int N>
struct SRun {
    static constexpr int P = N >= 16 ? 16 : (N >= 8 ? 8 : (N >= 4 ? 4 : (N >= 2 ? 2 : 1)));
    SPiece<P> head;
    SRun<N - P> tail;
    template <int OFF = 0> __device__ __forceinline__ void issue(const uint32_t* p) {
        head.template issue<OFF>(p);
        tail.template issue<OFF + 4 * P>(p);
    }
    __device__ __forceinline__ void fence() { head.fence(); tail.fence(); }
    __device__ __forceinline__ uint32_t word(int i) const { return i < P ? head.word(i < P ? i : 0) : tail.word(i - P); }
};
template <>
struct SRun<0> {
    template <int OFF = 0> __device__ __forceinline__ void issue(const uint32_t*) {}
    __device__ __forceinline__ void fence() {}
    __device__ __forceinline__ uint32_t word(int) const { return 0u; }
};

// U consecutive records in SGPRs
template <int W, int LW, bool TERN, int U>
struct ScalarGroup {
    SRun<U * W> b;
    SRun<TERN ? U * W : 0> z;
    SRun<U * LW> l;
    __device__ __forceinline__ void issue(const ScanArgs& a, int64_t at) {
        b.issue(a.rbits + at * W);
        if (TERN) z.issue(a.rzero + at * W);
        l.issue(a.rlab + at * LW);
    }
    __device__ __forceinline__ void fence() { b.fence(); z.fence(); l.fence(); }
    __device__ __forceinline__ void get(Rec<W, LW, TERN>& r, int u) const {           // u is a constant after unrolling
#pragma unroll
        for (int w = 0; w < W; ++w) r.b[w] = b.word(u * W + w);
        if (TERN) {
#pragma unroll
            for (int w = 0; w < W; ++w) r.z[w] = z.word(u * W + w);
        }
#pragma unroll
        for (int w = 0; w < LW; ++w) r.l[w] = l.word(u * LW + w);
    }
};

// GM_LDS: the same 64-record batch, but staged through a small LDS ring, word-major: ring[word][item].  Lane i writes
// the words of record base+i (conflict-free ds_write_b32s); word x of four consecutive items is then ONE wave-uniform
// ds_read_b128 (an LDS broadcast: one address, no bank conflict), i.e. RW reads per 4 items with no padding.  No
// v_readlane on the VALU pipe, no SGPR pressure, and -- LDS being in-order per wave -- every wait in the loop is a
// counted lgkmcnt.  lgkmcnt has 4 bits on gfx9: the loops below keep <= 15 LDS ops between a read and its use so a
// wait never has to drain the atomics issued after it.
template <int W, int LW, bool TERN>
struct LdsBatch {
    static constexpr int RW = W * (TERN ? 2 : 1) + LW;
    static constexpr int RS = (RW + 3) / 4 * 4;                    // host sizes the ring as 64 * RS dwords
    using R = Rec<W, LW, TERN>;
    uint32_t w[RW];
    __device__ __forceinline__ void load(const ScanArgs& a, int64_t base, int64_t hi, int lane) {
        const int64_t i = base + lane;
        const bool ok = i < hi;
#pragma unroll
        for (int x = 0; x < W; ++x) w[x] = ok ? a.rbits[i * W + x] : 0u;
        if (TERN) {
#pragma unroll
            for (int x = 0; x < W; ++x) w[W + x] = ok ? a.rzero[i * W + x] : 0xffffffffu;
        }
#pragma unroll
        for (int x = 0; x < LW; ++x) w[W * (TERN ? 2 : 1) + x] = ok ? a.rlab[i * LW + x] : 0u;
    }
    __device__ __forceinline__ void publish(uint32_t* ring, int lane) const {
#pragma unroll
        for (int x = 0; x < RW; ++x) ring[x * 64 + lane] = w[x];
    }
    static __device__ __forceinline__ void set_word(R& r, int x, uint32_t v) {      // x is a constant after unrolling
        if (x < W) r.b[x] = v;
        else if (TERN && x < 2 * W) r.z[TERN ? x - W : 0] = v;
        else r.l[x - W * (TERN ? 2 : 1)] = v;
    }
    static __device__ __forceinline__ void get(R& r, const uint32_t* ring, int u) {
#pragma unroll
        for (int x = 0; x < RW; ++x) set_word(r, x, ring[x * 64 + u]);              // wave-uniform address
    }
    // items u0 .. u0+3 (u0 % 4 == 0)
    static __device__ __forceinline__ void get4(R (&g)[4], const uint32_t* ring, int u0) {
#pragma unroll
        for (int x = 0; x < RW; ++x) {
            const uint4 v = *reinterpret_cast<const uint4*>(ring + x * 64 + u0);    // wave-uniform address
            set_word(g[0], x, v.x);
            set_word(g[1], x, v.y);
            set_word(g[2], x, v.z);
            set_word(g[3], x, v.w);
        }
    }
};

// items per unrolled group: two groups of records must fit the SGPR file (<= ~80 of 102) in GM_SCALAR mode
template <int W, int LW, bool TERN>
struct Unroll {
    static constexpr int words = W * (TERN ? 2 : 1) + LW;
    static constexpr int value = words <= 5 ? 8 : (words <= 10 ? 4 : 2);
};

// ---------------------------------------------------------------------------------------------------
// pass 1: chunk_hist[chunk][d][q] = (#items at distance d) | (#relevant items at distance d) << 16
// ---------------------------------------------------------------------------------------------------
template <int W, int LW, bool TERN, int GM>
__global__ __launch_bounds__(64) void k_scan_hist(ScanArgs a, uint32_t* __restrict__ chunk_hist) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];   // [nb][64]
    int chunk_id, qtile;
    if (!map_block(a, chunk_id, qtile)) return;
    const int lane = threadIdx.x;
    const int q = qtile * 64 + lane;
    for (int d = 0; d < a.nb; ++d) lds[d * 64 + lane] = 0u;
    QueryRegs<W, LW, TERN> qr;
    qr.load(a, q);
    using R = Rec<W, LW, TERN>;
    constexpr int U = Unroll<W, LW, TERN>::value;
    const int64_t lo = (int64_t)chunk_id * a.chunk;
    const int64_t hi = (lo + a.chunk < a.R) ? lo + a.chunk : a.R;

    auto count_one = [&](const R& r) {
        int d;
        bool rel;
        rec_eval<W, LW, TERN>(qr, r, a.K, d, rel);
        atomicAdd(&lds[d * 64 + lane], rel ? 0x10001u : 1u);
    };
    auto count_four = [&](const R (&g)[4]) {                         // records in VGPRs (LDS mode): one-op-per-word relevance
        int d[4];
        uint32_t hit[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) rec_eval01<W, LW, TERN, 1>(qr, g[u], a.K, d[u], hit[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            atomicAdd(&lds[d[u] * 64 + lane], (hit[u] << 16) + 1u);
        }
    };

    if constexpr (GM == GM_SCALAR) {
        // Two SGPR-resident groups, ping-pong.  SMEM returns out of order, so the only wait that makes a group usable
        // is lgkmcnt(0) -- which also drains every LDS op in flight.  The loop therefore puts ONE explicit wait at the
        // top of each half-iteration and issues, right after it, the next group's s_loads AND the previous group's
        // atomics (their operands are held in VGPRs for one stage): both then have the whole evaluation of the current
        // group (~10 VALU/item) to complete, and the next wait finds the counters already at zero.
        using SG = ScalarGroup<W, LW, TERN, U>;
        SG ga, gb;
        int dA[U], dB[U];
        uint32_t vA[U], vB[U];
        const int64_t ngroups = (hi - lo) / U;
        auto load = [&](SG& g, int64_t at) { g.issue(a, at); };
        auto eval = [&](SG& g, int (&d)[U], uint32_t (&v)[U]) {
            g.fence();
#pragma unroll
            for (int u = 0; u < U; ++u) {
                R r;
                g.get(r, u);
                uint32_t hit;
                rec_eval01<W, LW, TERN, 2>(qr, r, a.K, d[u], hit);
                v[u] = (hit << 16) + 1u;
            }
        };
        auto issue = [&](const int (&d)[U], const uint32_t (&v)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) atomicAdd(&lds[d[u] * 64 + lane], v[u]);
        };
        int64_t g = 0;
        bool eA = false, eB = false;
        if (ngroups > 0) load(ga, lo);
        while (g < ngroups) {
            __builtin_amdgcn_s_waitcnt(0xc07f);                                  // lgkmcnt(0): ga landed
            if (g + 1 < ngroups) load(gb, lo + (g + 1) * U);
            if (eB) issue(dB, vB);
            eval(ga, dA, vA);
            eA = true; eB = false;
            if (++g >= ngroups) break;
            __builtin_amdgcn_s_waitcnt(0xc07f);                                  // gb landed
            if (g + 1 < ngroups) load(ga, lo + (g + 1) * U);
            issue(dA, vA);
            eval(gb, dB, vB);
            eB = true; eA = false;
            ++g;
        }
        if (eA) issue(dA, vA);
        if (eB) issue(dB, vB);
        for (int64_t i = lo + ngroups * U; i < hi; ++i) {
            R r;
            rec_load_uniform<W, LW, TERN>(r, a, i);
            count_one(r);
        }
    } else if constexpr (GM == GM_LDS) {
        using LB = LdsBatch<W, LW, TERN>;
        uint32_t* ring = lds + a.nb * 64;                          // [64][RS] after the counters
        LB cur, nxt;
        cur.load(a, lo, hi, lane);
        for (int64_t base = lo; base < hi; base += 64) {
            cur.publish(ring, lane);
            nxt.load(a, base + 64, hi, lane);                      // prefetch (vmcnt) while this batch is consumed from LDS
            const int cnt = (hi - base < 64) ? (int)(hi - base) : 64;
            int u0 = 0;
            if (cnt == 64) {                                         // full batch: unrolled, one-group LDS read-ahead
                R ga[4], gb[4];
                LB::get4(ga, ring, 0);
#pragma unroll
                for (int g = 0; g < 16; g += 2) {
                    LB::get4(gb, ring, (g + 1) * 4);
                    count_four(ga);
                    if (g + 2 < 16) LB::get4(ga, ring, (g + 2) * 4);
                    count_four(gb);
                }
                u0 = 64;
            }
            for (; u0 + 4 <= cnt; u0 += 4) {
                R g[4];
                LB::get4(g, ring, u0);
                count_four(g);
            }
            for (; u0 < cnt; ++u0) {
                R r;
                LB::get(r, ring, u0);
                count_one(r);
            }
            cur = nxt;
        }
    }
    uint32_t* __restrict__ out = chunk_hist + ((int64_t)chunk_id * a.nb) * a.qpad + q;
    for (int d = 0; d < a.nb; ++d) out[(int64_t)d * a.qpad] = lds[d * 64 + lane];
}

// ---------------------------------------------------------------------------------------------------
// bucket tables between the passes (tiny, latency-bound -> spread over many waves, loads independent)
//   below[c][d][q] = (#all, #relevant) items of bucket d in chunks < c of this shard
//   tot[d][q]      = (#all, #relevant) items of bucket d in this shard
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_scan_below(const uint32_t* __restrict__ chunk_hist, int qpad, int nb, int nchunk,
                                                    uint2* __restrict__ below, uint2* __restrict__ tot) {
    const int lane = threadIdx.x & 63;
    const int d = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int q = blockIdx.x * 64 + lane;
    if (d >= nb) return;
    uint32_t ra = 0, rr = 0;
    int c = 0;
    for (; c + 4 <= nchunk; c += 4) {
        uint32_t h[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = chunk_hist[((int64_t)(c + j) * nb + d) * qpad + q];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            below[((int64_t)(c + j) * nb + d) * qpad + q] = make_uint2(ra, rr);
            ra += h[j] & 0xffffu;
            rr += h[j] >> 16;
        }
    }
    for (; c < nchunk; ++c) {
        const uint32_t h = chunk_hist[((int64_t)c * nb + d) * qpad + q];
        below[((int64_t)c * nb + d) * qpad + q] = make_uint2(ra, rr);
        ra += h & 0xffffu;
        rr += h >> 16;
    }
    tot[(int64_t)d * qpad + q] = make_uint2(ra, rr);
}

//   dpre[d][q] = rank offset of bucket d from outside this shard's bucket d: all lower buckets
//                (locally: exclusive prefix of tot; sharded: the caller's base_all/base_rel)
//   cap[q]     = min(n_rel, k)
__global__ __launch_bounds__(64) void k_scan_dpre(const uint2* __restrict__ tot, int Q, int qpad, int nb,
                                                  const uint32_t* __restrict__ base_all,
                                                  const uint32_t* __restrict__ base_rel,
                                                  const uint32_t* __restrict__ nrel_total, int64_t kcap,
                                                  uint2* __restrict__ dpre, uint32_t* __restrict__ cap_ws,
                                                  int32_t* __restrict__ cap_out, uint32_t* __restrict__ hist_all,
                                                  uint32_t* __restrict__ hist_rel, uint32_t* __restrict__ nrel_max) {
    const int q = blockIdx.x * 64 + threadIdx.x;
    const bool qok = q < Q;
    uint32_t ra = 0, rr = 0;
    int d = 0;
    for (; d + 8 <= nb; d += 8) {
        uint2 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = tot[(int64_t)(d + j) * qpad + q];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint2 o = make_uint2(ra, rr);
            if (base_all && qok) o = make_uint2(base_all[(int64_t)q * nb + d + j], base_rel[(int64_t)q * nb + d + j]);
            if (dpre) dpre[(int64_t)(d + j) * qpad + q] = o;
            if (qok && hist_all) hist_all[(int64_t)q * nb + d + j] = t[j].x;
            if (qok && hist_rel) hist_rel[(int64_t)q * nb + d + j] = t[j].y;
            ra += t[j].x;
            rr += t[j].y;
        }
    }
    for (; d < nb; ++d) {
        const uint2 t = tot[(int64_t)d * qpad + q];
        uint2 o = make_uint2(ra, rr);
        if (base_all && qok) o = make_uint2(base_all[(int64_t)q * nb + d], base_rel[(int64_t)q * nb + d]);
        if (dpre) dpre[(int64_t)d * qpad + q] = o;
        if (qok && hist_all) hist_all[(int64_t)q * nb + d] = t.x;
        if (qok && hist_rel) hist_rel[(int64_t)q * nb + d] = t.y;
        ra += t.x;
        rr += t.y;
    }
    if (cap_ws) {
        const uint32_t nrel = nrel_total ? (qok ? nrel_total[q] : 0u) : rr;
        const uint32_t cap = (kcap > 0 && (uint64_t)kcap < (uint64_t)nrel) ? (uint32_t)kcap : nrel;
        cap_ws[q] = cap;
        if (qok) cap_out[q] = (int32_t)cap;
        if (qok && nrel_max) atomicMax(nrel_max, nrel);
    }
}

// ---------------------------------------------------------------------------------------------------
// pass 2: ranks of relevant items.  cnt[d][lane] is a 64-bit counter {lo: rank of the NEXT item of bucket d,
// hi: ordinal of the NEXT relevant item of bucket d} (both 1-based); it starts at the bucket's global base
// and one ds_add_rtn_u64 per pair both advances it and returns (rank, ordinal) of the current item.
// The loop is software-pipelined by one group: the returns of group g are consumed after group g+1's
// distances have been computed.
// ---------------------------------------------------------------------------------------------------
template <int W, int LW, bool TERN, bool CAPPED, int GM>
__global__ __launch_bounds__(64) void k_scan_ap(ScanArgs a, const uint2* __restrict__ below, const uint2* __restrict__ dpre,
                                                const uint32_t* __restrict__ cap_ws, float* __restrict__ ap_part,
                                                const uint32_t* __restrict__ nrel_max, int rank_bits) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long cnt[];   // [nb][64]
    int chunk_id, qtile;
    if (!map_block(a, chunk_id, qtile)) return;
    // the 32-bit packed variant (k_scan_ap32) handles this call when rank and ordinal fit one word together
    if (rank_bits > 0 && (uint64_t)(*nrel_max) + 2 < (1ull << (32 - rank_bits))) return;
    const int lane = threadIdx.x;
    const int q = qtile * 64 + lane;
    {
        const uint2* __restrict__ pb = below + ((int64_t)chunk_id * a.nb) * a.qpad + q;
        const uint2* __restrict__ pd = dpre + q;
        int d = 0;
        for (; d + 8 <= a.nb; d += 8) {
            uint2 x[8], y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                x[j] = pb[(int64_t)(d + j) * a.qpad];
                y[j] = pd[(int64_t)(d + j) * a.qpad];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                cnt[(d + j) * 64 + lane] = (unsigned long long)(x[j].x + y[j].x + 1u) | ((unsigned long long)(x[j].y + y[j].y + 1u) << 32);
        }
        for (; d < a.nb; ++d) {
            const uint2 x = pb[(int64_t)d * a.qpad], y = pd[(int64_t)d * a.qpad];
            cnt[d * 64 + lane] = (unsigned long long)(x.x + y.x + 1u) | ((unsigned long long)(x.y + y.y + 1u) << 32);
        }
    }
    const uint32_t cap = CAPPED ? cap_ws[q] : 0u;

    QueryRegs<W, LW, TERN> qr;
    qr.load(a, q);
    using R = Rec<W, LW, TERN>;
    constexpr int U = Unroll<W, LW, TERN>::value;
    const int64_t lo = (int64_t)chunk_id * a.chunk;
    const int64_t hi = (lo + a.chunk < a.R) ? lo + a.chunk : a.R;
    float acc = 0.0f;

    // (rank, ordinal) of a relevant item -> ordinal / rank.  v_rcp_f32 is good to 1 ulp: a term is off by
    // <= 1.5e-7 relative, far inside the 1e-4 mAP tolerance (measured ~1e-8 on the goldens).
    // `hit` is the 0/1 relevance that was also the high-word increment: ordinal*hit zeroes non-relevant terms
    // without keeping a lane mask alive across the pipeline stage.
    auto credit = [&](unsigned long long old, uint32_t hit) {
        const uint32_t rank = (uint32_t)old;
        uint32_t ord = (uint32_t)(old >> 32);
        if (CAPPED) hit = ord <= cap ? hit : 0u;
        const float of = (float)__umul24(ord, hit);            // ordinals < 2^24 (R < 16.7 M per shard checked in the plan)
        acc = fmaf(of, __builtin_amdgcn_rcpf((float)rank), acc);
    };
    constexpr int UG = (GM == GM_LDS) ? 4 : U;                       // items per pipelined group
    unsigned long long old[UG];
    uint32_t hitp[UG];
    auto eval_issue = [&](const R (&g)[UG], bool have_prev) {
        int d[UG];
        uint32_t hit[UG];
#pragma unroll
        for (int u = 0; u < UG; ++u) rec_eval01<W, LW, TERN, (GM == GM_LDS ? 1 : (GM == GM_SCALAR ? 2 : 0))>(qr, g[u], a.K, d[u], hit[u]);
        if (have_prev) {
#pragma unroll
            for (int u = 0; u < UG; ++u) credit(old[u], hitp[u]);             // previous group's returns
        }
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            hitp[u] = hit[u];
            old[u] = atomicAdd(&cnt[d[u] * 64 + lane], 1ull | ((unsigned long long)hit[u] << 32));
        }
    };
    auto drain = [&]() {
#pragma unroll
        for (int u = 0; u < UG; ++u) credit(old[u], hitp[u]);
    };
    auto one = [&](const R& r) {
        int d;
        bool rel;
        rec_eval<W, LW, TERN>(qr, r, a.K, d, rel);
        const uint32_t hit = rel ? 1u : 0u;
        credit(atomicAdd(&cnt[d * 64 + lane], 1ull | ((unsigned long long)hit << 32)), hit);
    };

    if constexpr (GM == GM_SCALAR) {
        // Same one-wait-per-group pipeline as k_scan_hist (GM_SCALAR), one stage deeper: after the lgkmcnt(0) at the top
        // of a half-iteration the wave issues the next group's s_loads and the atomics of the group evaluated in the
        // PREVIOUS half, credits the group whose returns that wait has just made valid, and evaluates the current one.
        // Atomics are issued in item order (A0 B0 A1 B1 ...), which is what the running counters need.
        using SG = ScalarGroup<W, LW, TERN, U>;
        SG ga, gb;
        int dA[U], dB[U];
        uint32_t hA[U], hB[U], cA[U], cB[U];
        unsigned long long oA[U], oB[U];
        const int64_t ngroups = (hi - lo) / U;
        auto load = [&](SG& g, int64_t at) { g.issue(a, at); };
        auto eval = [&](SG& g, int (&d)[U], uint32_t (&h)[U]) {
            g.fence();
#pragma unroll
            for (int u = 0; u < U; ++u) {
                R r;
                g.get(r, u);
                rec_eval01<W, LW, TERN, 2>(qr, r, a.K, d[u], h[u]);
            }
        };
        auto issue = [&](const int (&d)[U], const uint32_t (&h)[U], unsigned long long (&o)[U], uint32_t (&c)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                c[u] = h[u];
                o[u] = atomicAdd(&cnt[d[u] * 64 + lane], 1ull | ((unsigned long long)h[u] << 32));
            }
        };
        auto settle = [&](const unsigned long long (&o)[U], const uint32_t (&c)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) credit(o[u], c[u]);
        };
        int64_t g = 0;
        bool eA = false, eB = false, iA = false, iB = false;   // e: evaluated, not yet issued; i: issued, not yet credited
        if (ngroups > 0) load(ga, lo);
        while (g < ngroups) {
            __builtin_amdgcn_s_waitcnt(0xc07f);                                  // lgkmcnt(0): ga landed, oA returned
            if (g + 1 < ngroups) load(gb, lo + (g + 1) * U);
            if (eB) { issue(dB, hB, oB, cB); iB = true; eB = false; }
            if (iA) { settle(oA, cA); iA = false; }
            eval(ga, dA, hA);
            eA = true;
            if (++g >= ngroups) break;
            __builtin_amdgcn_s_waitcnt(0xc07f);                                  // gb landed, oB returned
            if (g + 1 < ngroups) load(ga, lo + (g + 1) * U);
            issue(dA, hA, oA, cA); iA = true; eA = false;
            if (iB) { settle(oB, cB); iB = false; }
            eval(gb, dB, hB);
            eB = true;
            ++g;
        }
        // tail: at most one group evaluated-not-issued and one issued-not-credited (the older one)
        if (eA) { if (iB) { settle(oB, cB); iB = false; } issue(dA, hA, oA, cA); iA = true; }
        if (eB) { if (iA) { settle(oA, cA); iA = false; } issue(dB, hB, oB, cB); iB = true; }
        if (iA) settle(oA, cA);
        if (iB) settle(oB, cB);
        for (int64_t i = lo + ngroups * U; i < hi; ++i) {
            R r;
            rec_load_uniform<W, LW, TERN>(r, a, i);
            one(r);
        }
    } else if constexpr (GM == GM_LDS) {
        // Batches of 64 records are staged through an LDS ring slot by the wave itself and read back with wave-uniform
        // ds_read_b128s (broadcast); the next batch's global loads are in flight (vmcnt) during the whole current batch.
        // LDS is in-order per wave, so every wait in the loop is a counted lgkmcnt.  (A one-group LDS read-ahead with a
        // two-slot ring was measured and bought nothing: 0.69 vs 0.67 ms.)
        using LB = LdsBatch<W, LW, TERN>;
        uint32_t* ring = reinterpret_cast<uint32_t*>(cnt + a.nb * 64);   // [64][RS] after the counters
        LB cur, nxt;
        cur.load(a, lo, hi, lane);
        bool prev = false;
        for (int64_t base = lo; base < hi; base += 64) {
            cur.publish(ring, lane);
            nxt.load(a, base + 64, hi, lane);
            const int cntb = (hi - base < 64) ? (int)(hi - base) : 64;
            int u0 = 0;
            if (cntb == 64) {
                // full batch, fully unrolled with a one-group LDS read-ahead: the ds_reads of group g+1 are issued
                // before group g is evaluated (counted lgkmcnt waits: LDS returns in order)
                R ga[4], gb[4];
                LB::get4(ga, ring, 0);
#pragma unroll
                for (int g = 0; g < 16; g += 2) {
                    LB::get4(gb, ring, (g + 1) * 4);
                    eval_issue(ga, prev);
                    prev = true;
                    if (g + 2 < 16) LB::get4(ga, ring, (g + 2) * 4);
                    eval_issue(gb, true);
                }
                u0 = 64;
            }
            for (; u0 + 4 <= cntb; u0 += 4) {
                R g[4];
                LB::get4(g, ring, u0);
                eval_issue(g, prev);
                prev = true;
            }
            if (u0 < cntb) {
                if (prev) drain();
                prev = false;
                for (; u0 < cntb; ++u0) {
                    R r;
                    LB::get(r, ring, u0);
                    one(r);
                }
            }
            cur = nxt;
        }
        if (prev) drain();
    }
    ap_part[(int64_t)chunk_id * a.qpad + q] = acc;
}

// ---------------------------------------------------------------------------------------------------
// pass 2, packed variant: when (rank bits) + (ordinal bits) <= 32 -- known on the device after pass 1 -- the two
// running numbers share ONE 32-bit counter {lo rank_bits: rank, hi: ordinal}.  Half the LDS (two waves per SIMD at
// K = 64) and a 32-bit returning add per pair.  Gated against k_scan_ap by the same device word, no host sync.
// ---------------------------------------------------------------------------------------------------
template <int W, int LW, bool TERN, bool CAPPED>
__global__ __launch_bounds__(64) void k_scan_ap32(ScanArgs a, const uint2* __restrict__ below, const uint2* __restrict__ dpre,
                                                  const uint32_t* __restrict__ cap_ws, float* __restrict__ ap_part,
                                                  const uint32_t* __restrict__ nrel_max, int rank_bits) {
    extern __shared__ __attribute__((aligned(16))) uint32_t cnt32[];            // [nb][64]
    int chunk_id, qtile;
    if (!map_block(a, chunk_id, qtile)) return;
    if (!((uint64_t)(*nrel_max) + 2 < (1ull << (32 - rank_bits)))) return;       // k_scan_ap takes this call
    const int lane = threadIdx.x;
    const int q = qtile * 64 + lane;
    {
        const uint2* __restrict__ pb = below + ((int64_t)chunk_id * a.nb) * a.qpad + q;
        const uint2* __restrict__ pd = dpre + q;
        int d = 0;
        for (; d + 8 <= a.nb; d += 8) {
            uint2 x[8], y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                x[j] = pb[(int64_t)(d + j) * a.qpad];
                y[j] = pd[(int64_t)(d + j) * a.qpad];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) cnt32[(d + j) * 64 + lane] = (x[j].x + y[j].x + 1u) | ((x[j].y + y[j].y + 1u) << rank_bits);
        }
        for (; d < a.nb; ++d) {
            const uint2 x = pb[(int64_t)d * a.qpad], y = pd[(int64_t)d * a.qpad];
            cnt32[d * 64 + lane] = (x.x + y.x + 1u) | ((x.y + y.y + 1u) << rank_bits);
        }
    }
    const uint32_t cap = CAPPED ? cap_ws[q] : 0u;
    const uint32_t rmask = (1u << rank_bits) - 1u;
    const uint32_t one_rel = 1u << rank_bits;

    QueryRegs<W, LW, TERN> qr;
    qr.load(a, q);
    using R = Rec<W, LW, TERN>;
    const int64_t lo = (int64_t)chunk_id * a.chunk;
    const int64_t hi = (lo + a.chunk < a.R) ? lo + a.chunk : a.R;
    float acc = 0.0f;
    auto credit = [&](uint32_t old, uint32_t hit) {
        const uint32_t rank = old & rmask;
        uint32_t ord = old >> rank_bits;
        if (CAPPED) hit = ord <= cap ? hit : 0u;
        const float of = (float)__umul24(ord, hit);
        acc = fmaf(of, __builtin_amdgcn_rcpf((float)rank), acc);
    };
    uint32_t old[4], hitp[4];
    auto eval_issue = [&](const R (&g)[4], bool have_prev) {
        int d[4];
        uint32_t hit[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) rec_eval01<W, LW, TERN, 1>(qr, g[u], a.K, d[u], hit[u]);
        if (have_prev) {
#pragma unroll
            for (int u = 0; u < 4; ++u) credit(old[u], hitp[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            hitp[u] = hit[u];
            old[u] = atomicAdd(&cnt32[d[u] * 64 + lane], (hit[u] << rank_bits) + 1u);      // v_lshl_add_u32
        }
    };
    // gallery through the LDS ring (see k_scan_ap, GM_LDS)
    using LB = LdsBatch<W, LW, TERN>;
    uint32_t* ring = cnt32 + a.nb * 64;                              // [64][RS] after the counters
    LB cur, nxt;
    cur.load(a, lo, hi, lane);
    bool prev = false;
    for (int64_t base = lo; base < hi; base += 64) {
        cur.publish(ring, lane);
        nxt.load(a, base + 64, hi, lane);
        const int cntb = (hi - base < 64) ? (int)(hi - base) : 64;
        int u0 = 0;
        if (cntb == 64) {                                            // full batch: unrolled, one-group LDS read-ahead
            R ga[4], gb[4];
            LB::get4(ga, ring, 0);
#pragma unroll
            for (int g = 0; g < 16; g += 2) {
                LB::get4(gb, ring, (g + 1) * 4);
                eval_issue(ga, prev);
                prev = true;
                if (g + 2 < 16) LB::get4(ga, ring, (g + 2) * 4);
                eval_issue(gb, true);
            }
            u0 = 64;
        }
        for (; u0 + 4 <= cntb; u0 += 4) {
            R g[4];
            LB::get4(g, ring, u0);
            eval_issue(g, prev);
            prev = true;
        }
        if (u0 < cntb) {
            if (prev) {
#pragma unroll
                for (int u = 0; u < 4; ++u) credit(old[u], hitp[u]);
            }
            prev = false;
            for (; u0 < cntb; ++u0) {
                R r;
                LB::get(r, ring, u0);
                int d;
                bool rel;
                rec_eval<W, LW, TERN>(qr, r, a.K, d, rel);
                const uint32_t hit = rel ? 1u : 0u;
                credit(atomicAdd(&cnt32[d * 64 + lane], rel ? one_rel + 1u : 1u), hit);
            }
        }
        cur = nxt;
    }
    if (prev) {
#pragma unroll
        for (int u = 0; u < 4; ++u) credit(old[u], hitp[u]);
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
struct WsLayout {
    size_t chunk_hist, below, tot, dpre, cap, gate, ap_part, total;
};

WsLayout ws_layout(const xmh_scan_plan& p) {
    WsLayout L;
    const size_t cells = (size_t)p.nchunk * p.nbuckets * p.qpad;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o += (bytes + 255) & ~(size_t)255;
        return at;
    };
    L.chunk_hist = take(cells * 4);
    L.below = take(cells * 8);
    L.tot = take((size_t)p.nbuckets * p.qpad * 8);
    L.dpre = take((size_t)p.nbuckets * p.qpad * 8);
    L.cap = take((size_t)p.qpad * 4);
    L.gate = take(256);
    L.ap_part = take((size_t)p.nchunk * p.qpad * 4);
    L.total = o;
    return L;
}

int make_plan(int64_t Q, int64_t R, int K, int ternary, xmh_scan_plan* p) {
    if (Q <= 0 || R <= 0 || K <= 0) return xmh::fail(XMH_EINVAL, "scan plan: bad shape Q=%lld R=%lld K=%d", (long long)Q, (long long)R, K);
    if (R >= (1ll << 24) || Q >= (1ll << 24)) return xmh::fail(XMH_ENOTSUP, "scan plan: shard too large (R=%lld, Q=%lld)", (long long)R, (long long)Q);
    const int64_t nb = ternary ? 2 * (int64_t)K + 1 : (int64_t)K + 1;
    const int64_t lds_ap = nb * 64 * 8;
    if (lds_ap > 160 * 1024) return xmh::fail(XMH_ENOTSUP, "scan plan: %lld distance buckets need %lld B of LDS per wave (max 163840); K=%d%s", (long long)nb, (long long)lds_ap, K, ternary ? " ternary" : "");
    const int64_t nqt = xmh::ceil_div(Q, 64);
    int64_t wpc = (160 * 1024) / (lds_ap / 2);    // waves per CU of the packed pass-2 variant (the 64-bit one fits half)
    if (wpc > 8) wpc = 8;
    // pass 2 runs `rounds` resident sets of waves; pass 1 (half the LDS) then gets 2 waves per SIMD
    const int64_t slots = (int64_t)xmh::device_cu_count() * wpc;
    static const int rounds_env = getenv("XMH_SCAN_ROUNDS") ? atoi(getenv("XMH_SCAN_ROUNDS")) : 0;
    const int64_t rounds = rounds_env > 0 ? rounds_env : 1;
    int64_t nchunk = rounds * slots / nqt;
    if (nchunk < 1) nchunk = 1;
    if (nchunk > 8) nchunk = (nchunk + 4) / 8 * 8;      // whole XCD groups: every XCD gets the same number of chunks
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
    p->ws_bytes = ws_layout(*p).total;
    return XMH_OK;
}

int gallery_mode(const char* env, int dflt) {
    const char* v = getenv(env);
    if (!v) return dflt;
    const int m = atoi(v);
    return m == 0 ? GM_SCALAR : GM_LDS;
}

template <typename F>
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

template <typename KernT>
int raise_lds(KernT kern, size_t lds, const char* who) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return xmh::fail(XMH_EHIP, "%s: cannot raise dynamic LDS to %zu: %s", who, lds, hipGetErrorString(e));
    }
    return XMH_OK;
}

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
    const WsLayout L = ws_layout(p);
    char* base = static_cast<char*>(ws);
    uint32_t* chunk_hist = reinterpret_cast<uint32_t*>(base + L.chunk_hist);
    uint2* below = reinterpret_cast<uint2*>(base + L.below);
    uint2* tot = reinterpret_cast<uint2*>(base + L.tot);
    hipStream_t st = xmh::as_stream(stream);
    const int W = (K + 31) / 32, LW = (C + 31) / 32;
    const int gm = gallery_mode("XMH_SCAN_GM_HIST", GM_LDS);
    const size_t ring = (size_t)64 * (((W * (tern ? 2 : 1) + LW) + 3) / 4 * 4) * 4;     // GM_LDS staging ring
    const size_t lds = (size_t)p.nbuckets * 64 * 4 + (gm == GM_LDS ? ring : 0);
    auto launch = [&](auto tern_c, auto gm_c) {
        constexpr bool T = decltype(tern_c)::value;
        constexpr int G = decltype(gm_c)::value;
        return dispatch_shape(W, LW, [&](auto w, auto l) {
            auto kern = k_scan_hist<decltype(w)::value, decltype(l)::value, T, G>;
            const int r2 = raise_lds(kern, lds, "xmh_hamming_hist");
            if (r2) return r2;
            xmh::ProfScope prof("scan_hist", st);
            hipLaunchKernelGGL(kern, dim3(scan_grid(p)), dim3(64), lds, st, a, chunk_hist);
            return (int)XMH_OK;
        });
    };
    using T1 = std::true_type;
    using T0 = std::false_type;
    using GS = std::integral_constant<int, GM_SCALAR>;
    using GD = std::integral_constant<int, GM_LDS>;
    if (tern) rc = gm == GM_SCALAR ? launch(T1{}, GS{}) : launch(T1{}, GD{});
    else rc = gm == GM_SCALAR ? launch(T0{}, GS{}) : launch(T0{}, GD{});
    if (rc) return rc;
    XMH_LAUNCH_CHECK("xmh_hamming_hist");
    hipLaunchKernelGGL(k_scan_below, dim3((unsigned)p.nqtile, (unsigned)xmh::ceil_div(p.nbuckets, 4)), dim3(256), 0, st, chunk_hist,
                       (int)p.qpad, (int)p.nbuckets, (int)p.nchunk, below, tot);
    XMH_LAUNCH_CHECK("xmh_hamming_hist below");
    if (hist_all || hist_rel) {
        hipLaunchKernelGGL(k_scan_dpre, dim3((unsigned)p.nqtile), dim3(64), 0, st, (const uint2*)tot, (int)Q, (int)p.qpad, (int)p.nbuckets,
                           (const uint32_t*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr, (int64_t)0,
                           (uint2*)nullptr, (uint32_t*)nullptr, (int32_t*)nullptr, hist_all, hist_rel, (uint32_t*)nullptr);
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
    const WsLayout L = ws_layout(p);
    char* base = static_cast<char*>(ws);
    const uint2* below = reinterpret_cast<const uint2*>(base + L.below);
    const uint2* tot = reinterpret_cast<const uint2*>(base + L.tot);
    uint2* dpre = reinterpret_cast<uint2*>(base + L.dpre);
    uint32_t* cap_ws = reinterpret_cast<uint32_t*>(base + L.cap);
    float* ap_part = reinterpret_cast<float*>(base + L.ap_part);
    hipStream_t st = xmh::as_stream(stream);
    uint32_t* nrel_max = reinterpret_cast<uint32_t*>(base + L.gate);
    // packed 32-bit counters apply to a single shard when rank fits rank_bits and the largest relevant count fits the rest
    int rank_bits = 0;
    if (!base_all && getenv("XMH_SCAN_NO_PACK32") == nullptr) {
        while ((1ll << rank_bits) < R + 2) ++rank_bits;
        if (rank_bits > 24) rank_bits = 0;
    }
    if (rank_bits) XMH_HIP(hipMemsetAsync(nrel_max, 0, 4, st));
    hipLaunchKernelGGL(k_scan_dpre, dim3((unsigned)p.nqtile), dim3(64), 0, st, tot, (int)Q, (int)p.qpad, (int)p.nbuckets, base_all,
                       base_rel, nrel_total, k, dpre, cap_ws, cap, (uint32_t*)nullptr, (uint32_t*)nullptr, rank_bits ? nrel_max : (uint32_t*)nullptr);
    XMH_LAUNCH_CHECK("xmh_hamming_ap dpre");
    const int W = (K + 31) / 32, LW = (C + 31) / 32;
    const int gm = gallery_mode("XMH_SCAN_GM_AP", GM_LDS);
    const size_t ring = (size_t)64 * (((W * (tern ? 2 : 1) + LW) + 3) / 4 * 4) * 4;     // GM_LDS staging ring
    const size_t lds = (size_t)p.nbuckets * 64 * 8 + (gm == GM_LDS ? ring : 0);
    auto launch = [&](auto tern_c, auto cap_c, auto gm_c) {
        constexpr bool T = decltype(tern_c)::value;
        constexpr bool CP = decltype(cap_c)::value;
        constexpr int G = decltype(gm_c)::value;
        return dispatch_shape(W, LW, [&](auto w, auto l) {
            auto kern = k_scan_ap<decltype(w)::value, decltype(l)::value, T, CP, G>;
            const int r2 = raise_lds(kern, lds, "xmh_hamming_ap");
            if (r2) return r2;
            xmh::ProfScope prof("scan_ap", st);
            hipLaunchKernelGGL(kern, dim3(scan_grid(p)), dim3(64), lds, st, a, below, (const uint2*)dpre, (const uint32_t*)cap_ws, ap_part,
                               (const uint32_t*)nrel_max, rank_bits);
            return (int)XMH_OK;
        });
    };
    using T1 = std::true_type;
    using T0 = std::false_type;
    using GS = std::integral_constant<int, GM_SCALAR>;
    const bool capped = k > 0;
    if (rank_bits) {
        const size_t lds32 = (size_t)p.nbuckets * 64 * 4 + ring;
        auto launch32 = [&](auto tern_c, auto cap_c) {
            constexpr bool T = decltype(tern_c)::value;
            constexpr bool CP = decltype(cap_c)::value;
            return dispatch_shape(W, LW, [&](auto w, auto l) {
                auto kern = k_scan_ap32<decltype(w)::value, decltype(l)::value, T, CP>;
                const int r2 = raise_lds(kern, lds32, "xmh_hamming_ap");
                if (r2) return r2;
                xmh::ProfScope prof("scan_ap32", st);
                hipLaunchKernelGGL(kern, dim3(scan_grid(p)), dim3(64), lds32, st, a, below, (const uint2*)dpre, (const uint32_t*)cap_ws, ap_part,
                                   (const uint32_t*)nrel_max, rank_bits);
                return (int)XMH_OK;
            });
        };
        if (tern) rc = capped ? launch32(T1{}, T1{}) : launch32(T1{}, T0{});
        else rc = capped ? launch32(T0{}, T1{}) : launch32(T0{}, T0{});
        if (rc) return rc;
        XMH_LAUNCH_CHECK("xmh_hamming_ap packed");
    }
    if (gm == GM_SCALAR) {
        if (tern) rc = capped ? launch(T1{}, T1{}, GS{}) : launch(T1{}, T0{}, GS{});
        else rc = capped ? launch(T0{}, T1{}, GS{}) : launch(T0{}, T0{}, GS{});
    } else {
        using GD = std::integral_constant<int, GM_LDS>;
        if (tern) rc = capped ? launch(T1{}, T1{}, GD{}) : launch(T1{}, T0{}, GD{});
        else rc = capped ? launch(T0{}, T1{}, GD{}) : launch(T0{}, T0{}, GD{});
    }
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
