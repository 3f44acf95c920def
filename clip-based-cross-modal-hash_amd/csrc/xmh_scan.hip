// Fused ranking scan: calc_map_k (reference common/calc_utils.py:58-92) as two streaming passes over
// bit-packed codes, no [Q,R] intermediate, no sort.
//
// Execution shape: a wave owns QW = 64/S queries and S "slots"; lane = slot * QW + ql holds query ql (code words and
// label masks in VGPRs) and, at step t of a 64-item batch, works on gallery item t*S + slot.  Per-query bucket counters
// sit in LDS as cnt[d][ql].  Walking the gallery in index order makes "number of same-distance items with a smaller
// index" a running counter, which is exactly the tie-break of the canonical (distance, index) order:
//   pass 1 (k_scan_hist_s)  histogram of (distance, relevant) per (chunk, query): one ds_add_u32 per pair;
//   tables                  exclusive prefixes over chunks and over buckets (k_scan_below, k_scan_dpre);
//   pass 2 (k_scan_ap_s)    counters start at the global base of (bucket, chunk); one returning ds_add per pair yields
//                           (rank, ordinal) of the item, relevant items add ordinal/rank to the query's AP sum.
// Binary codes of at most 256 bits take the MFMA-evaluated pass 1 further down (k_scan_hist_m2 up to 64 bits, k_scan_hist_m beyond),
// which also leaves (distance << 1 | relevant) of every pair in the workspace; pass 2 then reads that pair cache instead of the
// gallery (k_scan_ap_c up to 64 bits, the CACHE variant of k_scan_ap_s beyond).  What follows describes the VALU kernels, which
// remain for ternary codes, long codes (512..2048 bits) and as the checked alternative (XMH_SCAN_MFMA=0).
// Why slots: with one query per lane (S = 1) the counters of a wave cost nb*64*{4,8} bytes -- 33 KB at K = 64 with
// 64-bit counters (one wave per SIMD), 131 KB at K = 256 (one wave per CU).  S lanes per query shrink the footprint by
// S: 3-4 waves per SIMD for every supported K, which is what hides the LDS round trips of this loop (measured at
// K=64/128/256: 0.55/3.6/13.3 ms for S = 1 against 0.34/0.47/0.77 ms).
// Ordering: the S lanes of a query may hit the same counter in one instruction.  Pass 1 only needs totals.  Pass 2 needs
// the returns in item order; the LDS serialises same-address lanes in ascending lane order (= item order, slot being the
// high lane bits).  That is observed behaviour, not an ISA promise, so xmh_hamming_ap probes it once per process
// (k_probe_lane_order) and otherwise issues the add once per slot under an exec mask (MASKED, ~2x slower).
//
// The gallery batch: lane i loads record base+i (coalesced, vmcnt; the next batch is in flight during the current one),
// the wave stages it record-major in a small LDS ring, and a lane reads the record of its own item with ds_read_b128s
// (S distinct addresses per instruction, each broadcast to QW lanes).  LDS ops of a wave complete in order, so every
// wait in the loop is a counted lgkmcnt; reads run one group of steps (SlotGeom::G) ahead of their use.  Records of 64 B
// and more (K >= 512) are loaded as one contiguous range and scattered to padded rows, and at one query per wave
// (S = 64) eight waves share each staged batch (see AosBatch, k_scan_hist_s).
//
// Bound: VALU issue.  tools/ubench_valu.hip measures ~4.0-4.4 cycles per wave64 instruction per SIMD for
// the instruction mix of these loops (v_xor/v_and/v_add alone reach 2.4, VOP3 ops such as v_bcnt_u32_b32 /
// v_and_or_b32 / v_lshl_add_u32 4.2, v_rcp_f32 8.2): 10 (pass 1) / 14-15 (pass 2) VALU instructions per pair-step of a
// wave at K=64, C=80.  The relevance test is one v_and_or_b32 per label word + one v_min_u32 (inline asm: hipcc does
// not form them).  Algorithmic HBM bytes per launch: R*(4W+4Lw) + Q*(4W+4Lw) (+ the bucket tables in the workspace).
#include "xmh_common.h"
#include <atomic>
#include <math.h>
#include <mutex>
#include <string.h>
#include <vector>
#include "xmh_scan_bits.h"
#include "xmh_scan_mfma.h"

#include <stdlib.h>
#include <type_traits>

namespace {

// (kMaxChunk ... ScanArgs, map_block: xmh_scan_mfma.h)

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

// distance bucket d and relevance (as 0/1 in `hit01`) of (this lane's query, record r); all operands in VGPRs
template <int W, int LW, bool TERN>
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
    uint32_t hit = qr.l[0] & r.l[0];
#pragma unroll
    for (int w = 1; w < LW; ++w) asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(hit) : "v"(qr.l[w]), "v"(r.l[w]));
    asm("v_min_u32 %0, %1, 1" : "=v"(hit01) : "v"(hit));
}

// ---------------------------------------------------------------------------------------------------
// bucket tables between the passes (tiny, latency-bound -> spread over many waves, loads independent)
//   below[c][d][q] = (#all, #relevant) items of bucket d in chunks < c of this shard
//   tot[d][q]      = (#all, #relevant) items of bucket d in this shard
// ---------------------------------------------------------------------------------------------------
// rel_scale = 0: pass-1 counters are (all | relevant << 16); kRelHi16: (all << 16 | relevant) (k_scan_hist_m2); else all + relevant * rel_scale
// (k_scan_hist_m)
constexpr uint32_t kRelHi16 = 0xffffffffu;
// words of the gate block of the workspace (cleared at the start of every xmh_hamming_hist): 0 = nrel_max, 1 = finalize ticket,
// 2 = gallery items over all shards, 3 = "a distance wrapped in the one-byte pair cache of 65..128-bit codes"
constexpr int kGateWrapped = 3;
// The block that finishes a 64-query tile LAST (ticket per tile, zeroed ahead of the launch) also writes what the unsharded pass 2
// needs from the totals -- dpre[d][q] = exclusive prefix of tot over d, nrel[q], the nrel_max gate word -- which used to be a
// launch of its own between the passes (k_scan_dpre: 9 us + a launch gap; kept for the sharded call and the histogram export).
__global__ __launch_bounds__(256) void k_scan_below(const uint32_t* __restrict__ chunk_hist, int qpad, int nb, int nchunk, uint32_t rel_scale,
                                                    uint2* __restrict__ below, uint2* __restrict__ tot, uint32_t* __restrict__ tickets, int Q,
                                                    uint2* __restrict__ dpre, uint32_t* __restrict__ nrel_ws, uint32_t* __restrict__ nrel_max,
                                                    uint32_t* __restrict__ hist_all, uint32_t* __restrict__ hist_rel) {
    __shared__ uint2 part[4][64];
    __shared__ int last;
    const int lane = threadIdx.x & 63, wq = threadIdx.x >> 6;
    const int d = blockIdx.y * 4 + wq;
    const int q = blockIdx.x * 64 + lane;
    const bool hi16 = rel_scale == kRelHi16;
    if (d < nb) {
        uint32_t ra = 0, rr = 0;
        int c = 0;
        for (; c + 4 <= nchunk; c += 4) {
            uint32_t h[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) h[j] = chunk_hist[((int64_t)(c + j) * nb + d) * qpad + q];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                below[((int64_t)(c + j) * nb + d) * qpad + q] = make_uint2(ra, rr);
                const uint32_t rel = hi16 ? h[j] & 0xffffu : (rel_scale ? h[j] / rel_scale : h[j] >> 16);
                ra += hi16 ? h[j] >> 16 : (rel_scale ? h[j] - rel * rel_scale : h[j] & 0xffffu);
                rr += rel;
            }
        }
        for (; c < nchunk; ++c) {
            const uint32_t h = chunk_hist[((int64_t)c * nb + d) * qpad + q];
            below[((int64_t)c * nb + d) * qpad + q] = make_uint2(ra, rr);
            const uint32_t rel = hi16 ? h & 0xffffu : (rel_scale ? h / rel_scale : h >> 16);
            ra += hi16 ? h >> 16 : (rel_scale ? h - rel * rel_scale : h & 0xffffu);
            rr += rel;
        }
        // agent-scope store (sc1: written through this XCD's L2): the totals are the one thing another block of this launch reads
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(tot + (int64_t)d * qpad + q), (unsigned long long)ra | ((unsigned long long)rr << 32),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!tickets) return;
    // hand-off without fences (an agent-scope release writes back the whole L2 of the XCD -- with 84 MB of `below` in flight that
    // made this kernel 22 -> 152 us): the totals go out as agent-scope stores, every wave waits for their acknowledgement
    // (vmcnt(0)) before the barrier, then one thread takes the tile's ticket; the block that draws the last ticket reads the
    // totals with agent-scope loads, which do not hit in a stale L1 / L2 line.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = __hip_atomic_fetch_add(tickets + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = t == gridDim.y - 1;
    }
    __syncthreads();
    if (!last) return;
    auto tot_at = [&](int dd) {
        const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(tot + (int64_t)dd * qpad + q), __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);
        return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
    };
    // 64 queries, the bucket axis split over the 4 waves: quarter sums, prefixed through LDS, then the exclusive prefixes
    const int nbq = (nb + 3) / 4;
    const int d0 = wq * nbq, d1 = (d0 + nbq < nb) ? d0 + nbq : nb;
    uint32_t sa = 0, sr = 0;
    for (int dd = d0; dd < d1; dd += 8) {
        uint2 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = dd + j < d1 ? tot_at(dd + j) : make_uint2(0u, 0u);
#pragma unroll
        for (int j = 0; j < 8; ++j) { sa += t[j].x; sr += t[j].y; }
    }
    part[wq][lane] = make_uint2(sa, sr);
    __syncthreads();
    uint32_t ra = 0, rr = 0, tr = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint2 pw = part[w][lane];
        if (w < wq) { ra += pw.x; rr += pw.y; }
        tr += pw.y;
    }
    for (int dd = d0; dd < d1; dd += 8) {
        uint2 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = dd + j < d1 ? tot_at(dd + j) : make_uint2(0u, 0u);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (dd + j < d1) {
                dpre[(int64_t)(dd + j) * qpad + q] = make_uint2(ra, rr);
                if (hist_all && q < Q) hist_all[(int64_t)q * nb + dd + j] = t[j].x;      // the shard totals a sharded evaluation gathers
                if (hist_rel && q < Q) hist_rel[(int64_t)q * nb + dd + j] = t[j].y;
                ra += t[j].x;
                rr += t[j].y;
            }
        }
    }
    if (wq == 0) {
        nrel_ws[q] = tr;
        if (q < Q) atomicMax(nrel_max, tr);
    }
}

//   dpre[d][q] = rank offset of bucket d from outside this shard's bucket d: all lower buckets
//                (locally: exclusive prefix of tot; sharded: the caller's base_all/base_rel)
//   nrel_ws[q] = relevant items of query q over all shards (pass 2 caps at min(nrel, k) itself), cap_out[q] = min(nrel, k)
__global__ __launch_bounds__(256) void k_scan_dpre(const uint2* __restrict__ tot, int Q, int qpad, int nb,
                                                   const uint32_t* __restrict__ base_all,
                                                   const uint32_t* __restrict__ base_rel,
                                                   const uint32_t* __restrict__ nrel_total, int64_t kcap,
                                                   uint2* __restrict__ dpre, uint32_t* __restrict__ cap_ws,
                                                   int32_t* __restrict__ cap_out, uint32_t* __restrict__ hist_all,
                                                   uint32_t* __restrict__ hist_rel, uint32_t* __restrict__ nrel_max,
                                                   uint32_t* __restrict__ rank_max = nullptr) {
    // rank_max (explicit-offsets form): the largest rank this shard can hand out, max over (q, d) of base_all + tot -- the word the
    // float-bit pass 2 is gated on, as the gallery size is for the totals-table form.
    // 64 queries per block, the bucket axis split over the 4 waves: each wave sums its quarter, the quarter sums are
    // prefixed through LDS, then each wave walks its quarter again (L2 hits) writing the exclusive prefixes.  One wave per
    // 64 queries walking all buckets was a 13 us latency chain of dependent-free but serialised load groups.
    __shared__ uint2 part[4][64];
    const int lane = threadIdx.x & 63, wq = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + lane;
    const bool qok = q < Q;
    const int nbq = (nb + 3) / 4;
    const int d0 = wq * nbq, d1 = (d0 + nbq < nb) ? d0 + nbq : nb;
    uint32_t sa = 0, sr = 0;
    for (int d = d0; d < d1; d += 8) {
        uint2 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = d + j < d1 ? tot[(int64_t)(d + j) * qpad + q] : make_uint2(0u, 0u);
#pragma unroll
        for (int j = 0; j < 8; ++j) { sa += t[j].x; sr += t[j].y; }
    }
    part[wq][lane] = make_uint2(sa, sr);
    __syncthreads();
    uint32_t ra = 0, rr = 0, ta = 0, tr = 0, rmax = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint2 p = part[w][lane];
        if (w < wq) { ra += p.x; rr += p.y; }
        ta += p.x; tr += p.y;
    }
    for (int d = d0; d < d1; d += 8) {
        uint2 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = d + j < d1 ? tot[(int64_t)(d + j) * qpad + q] : make_uint2(0u, 0u);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (d + j < d1) {
                uint2 o = make_uint2(ra, rr);
                if (base_all && qok) {
                    o = make_uint2(base_all[(int64_t)q * nb + d + j], base_rel[(int64_t)q * nb + d + j]);
                    rmax = max(rmax, o.x + t[j].x);
                }
                if (dpre) dpre[(int64_t)(d + j) * qpad + q] = o;
                if (qok && hist_all) hist_all[(int64_t)q * nb + d + j] = t[j].x;
                if (qok && hist_rel) hist_rel[(int64_t)q * nb + d + j] = t[j].y;
                ra += t[j].x;
                rr += t[j].y;
            }
        }
    }
    if (cap_ws && wq == 0) {
        const uint32_t nrel = nrel_total ? (qok ? nrel_total[q] : 0u) : tr;
        const uint32_t cap = (kcap > 0 && (uint64_t)kcap < (uint64_t)nrel) ? (uint32_t)kcap : nrel;
        cap_ws[q] = nrel;
        if (qok && cap_out) cap_out[q] = (int32_t)cap;
        if (qok && nrel_max) atomicMax(nrel_max, nrel);
    }
    if (rank_max) {
        for (int o = 32; o; o >>= 1) rmax = max(rmax, (uint32_t)__shfl_xor((int)rmax, o));
        if (lane == 0 && rmax) atomicMax(rank_max, rmax);
    }
    (void)ta;
}

// ---------------------------------------------------------------------------------------------------
// the 64-record gallery batch: one record per lane in registers, staged record-major ([item][RS]) in the LDS ring
// ---------------------------------------------------------------------------------------------------
template <int W, int LW, bool TERN>
struct AosBatch {
    static constexpr int RW = W * (TERN ? 2 : 1) + LW;
    static constexpr int RS = (RW + 3) / 4 * 4;
    using R = Rec<W, LW, TERN>;
    // Long records (W >= 16 words, 64..256 B): a lane loading "its" record would touch one cache line per lane and
    // instruction (64 lines per load, 8x traffic once the L1 thrashes -- measured 12x slower).  The 64-record batch is
    // one contiguous 64*W-dword range instead: piece j of lane l is dwords [j*256 + 4l, +4) of it, which belongs to record
    // (j*256 + 4l) / W; publish() scatters the pieces to the padded record rows of the ring.
    static constexpr bool COAL = W >= 16 && !TERN;
    uint32_t w[RS];
    // NW > 1: the NW waves of a block share the ring; wave `wave` brings pieces wave, wave + NW, ... and wave 0 the labels
    template <int NW = 1>
    __device__ __forceinline__ void load(const ScanArgs& a, int64_t base, int64_t hi, int lane, int wave = 0) {
        const int64_t i = base + lane;
        const bool ok = i < hi;
        if (COAL) {
#pragma unroll
            for (int jj = 0; jj < W / 4 / NW; ++jj) {
                const int o = (jj * NW + wave) * 256 + lane * 4;
                const bool okj = base + o / W < hi;
                const uint4 v = okj ? *reinterpret_cast<const uint4*>(a.rbits + base * W + o) : make_uint4(0u, 0u, 0u, 0u);
                w[4 * jj] = v.x; w[4 * jj + 1] = v.y; w[4 * jj + 2] = v.z; w[4 * jj + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int x = 0; x < W; ++x) w[x] = ok ? a.rbits[i * W + x] : 0u;
        }
        if (TERN) {
#pragma unroll
            for (int x = 0; x < W; ++x) w[W + x] = ok ? a.rzero[i * W + x] : 0xffffffffu;
        }
        if (NW == 1 || wave == 0) {
#pragma unroll
            for (int x = 0; x < LW; ++x) w[W * (TERN ? 2 : 1) + x] = ok ? a.rlab[i * LW + x] : 0u;
        }
#pragma unroll
        for (int x = RW; x < RS; ++x) w[x] = 0u;
    }
    template <int NW = 1>
    __device__ __forceinline__ void publish(uint32_t* ring, int lane, int wave = 0) const {
        if (COAL) {
#pragma unroll
            for (int jj = 0; jj < W / 4 / NW; ++jj) {
                const int o = (jj * NW + wave) * 256 + lane * 4;
                *reinterpret_cast<uint4*>(ring + (o / W) * RS + (o % W)) = make_uint4(w[4 * jj], w[4 * jj + 1], w[4 * jj + 2], w[4 * jj + 3]);
            }
            if (NW == 1 || wave == 0)                                // labels + padding (W % 4 == 0)
                *reinterpret_cast<uint4*>(ring + lane * RS + W) = make_uint4(w[W], w[W + 1], w[W + 2], w[W + 3]);
            return;
        }
#pragma unroll
        for (int x = 0; x < RS; x += 4) *reinterpret_cast<uint4*>(ring + lane * RS + x) = make_uint4(w[x], w[x + 1], w[x + 2], w[x + 3]);
    }
    // `mine` = ring + slot * RS (per lane); item = step * S + slot -> constant offset step * S * RS from it
    static __device__ __forceinline__ void get(R& r, const uint32_t* mine, int off_dwords) {
        uint32_t t[RS];
#pragma unroll
        for (int x = 0; x < RS; x += 4) {
            if (x + 4 <= RW || RW - x > 2) {
                const uint4 v = *reinterpret_cast<const uint4*>(mine + off_dwords + x);
                t[x] = v.x; t[x + 1] = v.y; t[x + 2] = v.z; t[x + 3] = v.w;
            } else if (RW - x == 2) {
                const uint2 v = *reinterpret_cast<const uint2*>(mine + off_dwords + x);
                t[x] = v.x; t[x + 1] = v.y; t[x + 2] = 0u; t[x + 3] = 0u;
            } else {
                t[x] = mine[off_dwords + x]; t[x + 1] = 0u; t[x + 2] = 0u; t[x + 3] = 0u;
            }
        }
#pragma unroll
        for (int x = 0; x < W; ++x) r.b[x] = t[x];
        if (TERN) {
#pragma unroll
            for (int x = 0; x < W; ++x) r.z[x] = t[W + x];
        }
#pragma unroll
        for (int x = 0; x < LW; ++x) r.l[x] = t[W * (TERN ? 2 : 1) + x];
    }
};

template <int S> struct SlotGeom {
    static constexpr int QW = 64 / S;
    static constexpr int LOG_QW = QW == 64 ? 6 : (QW == 32 ? 5 : (QW == 16 ? 4 : (QW == 8 ? 3 : (QW == 4 ? 2 : (QW == 2 ? 1 : 0)))));
    static constexpr int G = QW >= 16 ? 8 : (QW < 4 ? QW : 4);   // steps per pipelined group (at K = 64: 8 beats 4 by 4 %, 2 loses 4 %)
    static constexpr int NG = QW / G;                  // groups per 64-item batch (QW steps)
};

// NW = waves per block.  1 except for S = 64 (one query per wave): there every wave would stream its whole chunk for a
// single query -- the gallery re-read Q times from L2 / Infinity Cache (measured 6.5 TB/s, 10x the VALU time) -- so NW = 8
// waves (8 queries) share ONE staged batch: each wave loads an eighth of it, two barriers per batch.
// CACHE (S = 4, one wave per block, codes of at most 127 bits): pass 1 also writes one byte per pair, distance << 1 | relevant,
// so that pass 2 does not evaluate the pair again (XOR / popcount / label AND: 8 of its 15 VALU instructions) nor stage the
// gallery: lane l of the wave of (chunk, query tile) owns 16 consecutive bytes = its 16 steps of a 64-item batch, a wave
// stores 1 KB per batch.  Q x R bytes in all (593 MB at the COCO shape), streamed once each way while both passes are VALU-bound.
template <int W, int LW, bool TERN, int S, int NW, bool CACHE>
__global__ __launch_bounds__(64 * NW) void k_scan_hist_s(ScanArgs a, uint32_t* __restrict__ chunk_hist) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];   // NW x [nb][QW] counters, then the ring
    using SG = SlotGeom<S>;
    constexpr int QW = SG::QW, G = SG::G, NG = SG::NG;
    int chunk_id, qtile;
    if (!map_block(a, chunk_id, qtile)) return;                      // a.nqt counts tiles of QW * NW queries here
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ql = lane & (QW - 1), slot = lane >> SG::LOG_QW;
    const int q0 = (qtile * NW + wave) * QW;                         // first query of this wave
    const int q = q0 + ql;
    const int ncell = a.nb * QW, cstride = (ncell + 3) & ~3;
    uint32_t* cnt = lds + wave * cstride;
    for (int e = lane; e < ncell; e += 64) cnt[e] = 0u;
    QueryRegs<W, LW, TERN> qr;
    qr.load(a, q);
    using R = Rec<W, LW, TERN>;
    using LB = AosBatch<W, LW, TERN>;
    uint32_t* ring = lds + NW * cstride;
    const uint32_t* mine = ring + slot * LB::RS;
    const int64_t lo = (int64_t)chunk_id * a.chunk;
    const int64_t hi = (lo + a.chunk < a.R) ? lo + a.chunk : a.R;

    // cache entry: EB bits = distance << 1 | relevant; a lane's QW steps of a batch fill exactly 16 bytes (8-bit entries at
    // 16 steps: codes up to 64 bits; 16-bit entries at 8 steps: 65..256 bits), each group of G steps two dwords
    constexpr int EB = 128 / QW, EPW = 32 / EB;
    static_assert(!CACHE || (NW == 1 && (S == 4 || S == 8) && NG == 2 && G == 2 * EPW), "pair cache geometry: 16 bytes per lane and batch, two groups");
    uint32_t cw0 = 0u, cw1 = 0u, cw2 = 0u, cw3 = 0u;                 // CACHE: this lane's 16 bytes of the current batch
    auto count = [&](const R (&g)[G], uint32_t& wa, uint32_t& wb) {
        int d[G];
        uint32_t hit[G];
#pragma unroll
        for (int u = 0; u < G; ++u) rec_eval01<W, LW, TERN>(qr, g[u], a.K, d[u], hit[u]);
#pragma unroll
        for (int u = 0; u < G; ++u) atomicAdd(&cnt[d[u] * QW + ql], (hit[u] << 16) + 1u);
        if (CACHE) {                                                 // append an entry per step: w = entry << (32 - EB) | w >> EB
            // (counters indexed by that byte, so that the add is a constant 1, were tried: 3x the LDS per wave, pass 1 0.27 -> 0.31 ms)
#pragma unroll
            for (int u = 0; u < G / 2; ++u) wa = __builtin_amdgcn_alignbyte(((uint32_t)d[u] << 1) | hit[u], wa, EB / 8);
#pragma unroll
            for (int u = G / 2; u < G; ++u) wb = __builtin_amdgcn_alignbyte(((uint32_t)d[u] << 1) | hit[u], wb, EB / 8);
        }
    };
    const int nbatch = (a.chunk + 63) >> 6;
    uint4* crow = CACHE ? a.pair_cache + ((int64_t)chunk_id * a.nqt + qtile) * nbatch * 64 + lane : nullptr;
    auto fetch = [&](R (&g)[G], int group) {
#pragma unroll
        for (int u = 0; u < G; ++u) LB::get(g[u], mine, (group * G + u) * S * LB::RS);
    };
    LB cur, nxt;                                                     // the next batch's global loads fly during this one
    cur.template load<NW>(a, lo, hi, lane, wave);
    for (int64_t base = lo; base < hi; base += 64) {
        if (NW > 1) __syncthreads();                                 // every wave is done reading the previous batch
        cur.template publish<NW>(ring, lane, wave);
        nxt.template load<NW>(a, base + 64, hi, lane, wave);
        if (NW > 1) __syncthreads();                                 // the batch is complete in the ring
        const int cntb = (hi - base < 64) ? (int)(hi - base) : 64;
        if (cntb == 64) {
            R ga[G], gb[G];
            fetch(ga, 0);
#pragma unroll
            for (int g = 0; g < NG; g += 2) {
                if (g + 1 < NG) fetch(gb, g + 1);
                count(ga, cw0, cw1);
                if (g + 2 < NG) fetch(ga, g + 2);
                if (g + 1 < NG) count(gb, cw2, cw3);
            }
        } else if (!CACHE) {
            for (int t = 0; t * S < cntb; ++t) {
                R r;
                LB::get(r, mine, t * S * LB::RS);
                int d;
                uint32_t hit;
                rec_eval01<W, LW, TERN>(qr, r, a.K, d, hit);
                if (t * S + slot < cntb) atomicAdd(&cnt[d * QW + ql], (hit << 16) + 1u);
            }
        } else {
#pragma unroll
            for (int t = 0; t < QW; ++t) {                           // unrolled: the bytes land in fixed registers
                uint32_t b = 0u;
                if (t * S < cntb) {
                    R r;
                    LB::get(r, mine, t * S * LB::RS);
                    int d;
                    uint32_t hit;
                    rec_eval01<W, LW, TERN>(qr, r, a.K, d, hit);
                    b = ((uint32_t)d << 1) | hit;
                    if (t * S + slot < cntb) atomicAdd(&cnt[d * QW + ql], (hit << 16) + 1u);
                }
                uint32_t& w = t < EPW ? cw0 : (t < 2 * EPW ? cw1 : (t < 3 * EPW ? cw2 : cw3));
                w = __builtin_amdgcn_alignbyte(b, w, EB / 8);
            }
        }
        if (CACHE && EB == 8) {                                      // streamed once: non-temporal, so that the 593 MB do not sit dirty in
            uint4* dst = crow + ((base - lo) >> 6) * 64;                 // L2 / Infinity Cache while the small table kernels run
            __builtin_nontemporal_store(cw0, &dst->x);
            __builtin_nontemporal_store(cw1, &dst->y);
            __builtin_nontemporal_store(cw2, &dst->z);
            __builtin_nontemporal_store(cw3, &dst->w);
        }
        if (CACHE && EB == 16) {                                     // two-byte class entries go out 12 bits each (xmh_common.h): the four words of
            // 16-bit pairs become three.  p = E_even | E_odd << 16 with both below 4096: bfi(0xfff, p, p >> 4) = E_even | E_odd << 12
            const uint32_t a0 = (cw0 & 0xfffu) | ((cw0 >> 4) & ~0xfffu), a1 = (cw1 & 0xfffu) | ((cw1 >> 4) & ~0xfffu);
            const uint32_t a2 = (cw2 & 0xfffu) | ((cw2 >> 4) & ~0xfffu), a3 = (cw3 & 0xfffu) | ((cw3 >> 4) & ~0xfffu);
            uint32_t* dst = reinterpret_cast<uint32_t*>(a.pair_cache) +
                            ((((int64_t)chunk_id * a.nqt + qtile) * nbatch + ((base - lo) >> 6)) * 64 + lane) * xmh::kCache12Dwords;
            __builtin_nontemporal_store(a0 | (a1 << 24), dst);
            __builtin_nontemporal_store((a1 >> 8) | (a2 << 16), dst + 1);
            __builtin_nontemporal_store((a2 >> 16) | (a3 << 8), dst + 2);
        }
        cur = nxt;
    }
    uint32_t* __restrict__ out = chunk_hist + ((int64_t)chunk_id * a.nb) * a.qpad + q0;
    for (int e = lane; e < ncell; e += 64) out[(int64_t)(e >> SG::LOG_QW) * a.qpad + (e & (QW - 1))] = cnt[e];
}

// P32 = packed 32-bit counters {lo rank_bits: rank, hi: ordinal} when both fit one word (known on the device after
// pass 1: the launch is gated by *nrel_max, no host sync), else 64-bit {lo: rank, hi: ordinal}.  Counters are 1-based
// and start at the global base of (bucket, chunk).  MASKED: see the header (lane-order fallback).
// CACHE: the pairs come from the byte cache of pass 1 (see k_scan_hist_s) instead of the gallery.
// items_total (sharded calls that also launch k_scan_ap_c): that kernel takes the call when the gallery over all shards is small enough for it
// run_if (65..128-bit codes with one-byte cache entries, k_scan_hist_m<.., BYTE>): this launch is the stand-in for k_scan_ap_c and runs
// only when pass 1 raised the word (a distance of 128 wrapped in the cache) -- or when items_total says the gallery is too large for it
template <int W, int LW, bool TERN, bool CAPPED, int S, bool P32, bool MASKED, int NW, bool CACHE>
__global__ __launch_bounds__(64 * NW) void k_scan_ap_s(ScanArgs a, const uint2* __restrict__ below, const uint2* __restrict__ dpre,
                                                  const uint32_t* __restrict__ cap_ws, float* __restrict__ ap_part,
                                                  const uint32_t* __restrict__ nrel_max, int rank_bits, uint32_t kcap,
                                                  const uint32_t* __restrict__ items_total, const uint32_t* __restrict__ run_if = nullptr) {
    using CT = typename std::conditional<P32, uint32_t, unsigned long long>::type;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    using SG = SlotGeom<S>;
    constexpr int QW = SG::QW, G = SG::G, NG = SG::NG;
    int chunk_id, qtile;
    if (!map_block(a, chunk_id, qtile)) return;
    {
        const bool fits32 = rank_bits > 0 && (uint64_t)(*nrel_max) + 2 < (1ull << (32 - rank_bits));
        if (P32 != fits32) return;                                   // the other variant takes this call
    }
    if (run_if) {
        const bool too_large = items_total && (int64_t)*items_total > kFloatBitsMaxItems;
        if (!too_large && *run_if == 0u) return;                     // k_scan_ap_c takes this call
    } else if (items_total && (int64_t)*items_total <= kFloatBitsMaxItems) return;      // k_scan_ap_c takes this call
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;    // NW waves share the ring, see k_scan_hist_s
    const int ql = lane & (QW - 1), slot = lane >> SG::LOG_QW;
    const int q0 = (qtile * NW + wave) * QW;
    const int q = q0 + ql;
    const int ncell = a.nb * QW;
    const int cstride = ((ncell * (int)(sizeof(CT) / 4) + 3) & ~3);  // dwords of one wave's counters [nb][QW]
    CT* cnt = reinterpret_cast<CT*>(lds + wave * cstride);
    auto pack = [&](uint2 x, uint2 y) -> CT {
        if (P32) return (CT)((x.x + y.x + 1u) | ((x.y + y.y + 1u) << rank_bits));
        return (CT)((unsigned long long)(x.x + y.x + 1u) | ((unsigned long long)(x.y + y.y + 1u) << 32));
    };
    {
        const uint2* __restrict__ pb = below + ((int64_t)chunk_id * a.nb) * a.qpad + q0;
        const uint2* __restrict__ pd = dpre + q0;
        int e = lane;
        for (; e + 7 * 64 < ncell; e += 8 * 64) {
            uint2 x[8], y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ee = e + j * 64;
                const int64_t at = (int64_t)(ee >> SG::LOG_QW) * a.qpad + (ee & (QW - 1));
                x[j] = pb[at];
                y[j] = pd[at];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) cnt[e + j * 64] = pack(x[j], y[j]);
        }
        for (; e < ncell; e += 64) {
            const int64_t at = (int64_t)(e >> SG::LOG_QW) * a.qpad + (e & (QW - 1));
            cnt[e] = pack(pb[at], pd[at]);
        }
    }
    const uint32_t cap = CAPPED ? min(cap_ws[q], kcap) : 0u;      // cap_ws[q] = relevant items of query q
    const uint32_t rmask = P32 ? (1u << rank_bits) - 1u : 0xffffffffu;

    QueryRegs<W, LW, TERN> qr;
    qr.load(a, q);
    using R = Rec<W, LW, TERN>;
    using LB = AosBatch<W, LW, TERN>;
    uint32_t* ring = lds + NW * cstride;
    const uint32_t* mine = ring + slot * LB::RS;
    const int64_t lo = (int64_t)chunk_id * a.chunk;
    const int64_t hi = (lo + a.chunk < a.R) ? lo + a.chunk : a.R;
    float acc = 0.0f;

    auto credit = [&](CT old, uint32_t hit) {
        uint32_t rank, ord;
        if (P32) { rank = (uint32_t)old & rmask; ord = (uint32_t)old >> rank_bits; }
        else { rank = (uint32_t)old; ord = (uint32_t)((unsigned long long)old >> 32); }
        if (CAPPED) hit = ord <= cap ? hit : 0u;
        const float of = (float)__umul24(ord, hit);
        acc = fmaf(of, __builtin_amdgcn_rcpf((float)rank), acc);
    };
    auto inc = [&](uint32_t hit) -> CT {
        if (P32) return (CT)((hit << rank_bits) + 1u);
        return (CT)(1ull | ((unsigned long long)hit << 32));
    };
    CT old[G];
    uint32_t hitp[G];
    auto eval_issue = [&](const R (&g)[G], bool have_prev) {
        int d[G];
        uint32_t hit[G];
#pragma unroll
        for (int u = 0; u < G; ++u) rec_eval01<W, LW, TERN>(qr, g[u], a.K, d[u], hit[u]);
        if (have_prev) {
#pragma unroll
            for (int u = 0; u < G; ++u) credit(old[u], hitp[u]);
        }
#pragma unroll
        for (int u = 0; u < G; ++u) {
            hitp[u] = hit[u];
            CT* cell = &cnt[d[u] * QW + ql];
            const CT v = inc(hit[u]);
            if (!MASKED) {
                old[u] = atomicAdd(cell, v);                          // same-address lanes resolve in lane = item order
            } else {
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    if (slot == s) old[u] = atomicAdd(cell, v);      // slot order = index order
                }
            }
        }
    };
    auto drain = [&]() {
#pragma unroll
        for (int u = 0; u < G; ++u) credit(old[u], hitp[u]);
    };
    auto fetch = [&](R (&g)[G], int group) {
#pragma unroll
        for (int u = 0; u < G; ++u) LB::get(g[u], mine, (group * G + u) * S * LB::RS);
    };

    if constexpr (CACHE) {
        constexpr int EB = 128 / QW, EPW = 32 / EB;                   // entry bits, entries per word (see k_scan_hist_s)
        static_assert(NW == 1 && (S == 4 || S == 8) && NG == 2 && G == 2 * EPW && !MASKED, "pair cache geometry");
        // one group of 8 steps from two cache words; the returned counters are credited while the next group's adds fly
        // entry k (a constant after unrolling) of a lane's batch record r: 8- and 16-bit class entries; the latter are stored 12 bits each in
        // three dwords (xmh_common.h), sx / sy = the two entries that straddle a dword, shifted down
        auto entry = [&](const uint4& r, uint32_t sx, uint32_t sy, int k, uint32_t& d, uint32_t& hit) {
            if constexpr (EB == 8) {
                const uint32_t w = k < EPW ? r.x : (k < 2 * EPW ? r.y : (k < 3 * EPW ? r.z : r.w));
                d = __builtin_amdgcn_ubfe(w, EB * (k % EPW) + 1, EB - 1);
                hit = __builtin_amdgcn_ubfe(w, EB * (k % EPW), 1);
            } else {
                const uint32_t w = XMH_CACHE12_WORD(k, r.x, r.y, r.z, sx, sy);
                d = __builtin_amdgcn_ubfe(w, XMH_CACHE12_BIT(k) + 1, 11);
                hit = __builtin_amdgcn_ubfe(w, XMH_CACHE12_BIT(k), 1);
            }
        };
        auto issue = [&](const uint4& r, uint32_t sx, uint32_t sy, int first, bool have_prev) {
            if (have_prev) {
#pragma unroll
                for (int u = 0; u < G; ++u) credit(old[u], hitp[u]);
            }
#pragma unroll
            for (int u = 0; u < G; ++u) {
                uint32_t d, hit;
                entry(r, sx, sy, first + u, d, hit);
                hitp[u] = hit;
                old[u] = atomicAdd(&cnt[d * QW + ql], inc(hit));      // same-address lanes resolve in lane = item order
            }
        };
        const int nbatch = (a.chunk + 63) >> 6;
        const uint4* crow = a.pair_cache + ((int64_t)chunk_id * a.nqt + qtile) * nbatch * 64 + lane;
        const uint32_t* crow12 = reinterpret_cast<const uint32_t*>(a.pair_cache) + (((int64_t)chunk_id * a.nqt + qtile) * nbatch * 64 + lane) * xmh::kCache12Dwords;
        const int nfull = (int)((hi - lo) >> 6);                     // whole batches of this chunk
        // read once, 1 KB (12-bit records: 768 B) contiguous per wave instruction: non-temporal (see k_scan_ap_c).  EB == 16: the record's three
        // dwords come back as x, y, z and its two straddling entries, shifted down, as w (low half: entry 2, high half: entry 5)
        auto cache_words = [&](int64_t batch) -> uint4 {
            if constexpr (EB == 8) {
                typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
                const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(crow + batch * 64));
                return make_uint4(v.x, v.y, v.z, v.w);
            } else {
                typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));
                const u32x3_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x3_t*>(crow12 + batch * 64 * xmh::kCache12Dwords));
                return make_uint4(v.x, v.y, v.z, 0u);
            }
        };
        uint4 cw = cache_words(0);
        bool prev = false;
        for (int bi = 0; bi < nfull; ++bi) {
            const uint4 nw = cache_words(bi + 1 < nbatch ? bi + 1 : bi);               // unconditional: counted vmcnt, no predication
            const uint32_t sx = EB == 16 ? __builtin_amdgcn_alignbit(cw.y, cw.x, 24) : 0u, sy = EB == 16 ? __builtin_amdgcn_alignbit(cw.z, cw.y, 28) : 0u;
            issue(cw, sx, sy, 0, prev);
            prev = true;
            issue(cw, sx, sy, G, true);
            cw = nw;
        }
        if (prev) drain();
        const int cntb = (int)(hi - lo) - nfull * 64;                // ragged last batch (cw holds its words)
        const uint32_t sx = EB == 16 ? __builtin_amdgcn_alignbit(cw.y, cw.x, 24) : 0u, sy = EB == 16 ? __builtin_amdgcn_alignbit(cw.z, cw.y, 28) : 0u;
#pragma unroll
        for (int t = 0; t < QW; ++t) {
            if (t * S + slot < cntb) {
                uint32_t d, hit;
                entry(cw, sx, sy, t, d, hit);
                const CT o = atomicAdd(&cnt[d * QW + ql], inc(hit));
                credit(o, hit);
            }
        }
#pragma unroll
        for (int o = QW; o < 64; o <<= 1) acc += __shfl_xor(acc, o, 64);
        if (slot == 0) ap_part[(int64_t)chunk_id * a.qpad + q] = acc;
        return;
    }
    LB cur, nxt;                                                     // the next batch's global loads fly during this one
    cur.template load<NW>(a, lo, hi, lane, wave);
    bool prev = false;
    for (int64_t base = lo; base < hi; base += 64) {
        if (NW > 1) __syncthreads();                                 // every wave is done reading the previous batch
        cur.template publish<NW>(ring, lane, wave);
        nxt.template load<NW>(a, base + 64, hi, lane, wave);
        if (NW > 1) __syncthreads();                                 // the batch is complete in the ring
        const int cntb = (hi - base < 64) ? (int)(hi - base) : 64;
        if (cntb == 64) {
            R ga[G], gb[G];
            fetch(ga, 0);
#pragma unroll
            for (int g = 0; g < NG; g += 2) {
                if (g + 1 < NG) fetch(gb, g + 1);
                eval_issue(ga, prev);
                prev = true;
                if (g + 2 < NG) fetch(ga, g + 2);
                if (g + 1 < NG) eval_issue(gb, true);
            }
        } else {
            if (prev) drain();
            prev = false;
            for (int t = 0; t * S < cntb; ++t) {
                R r;
                LB::get(r, mine, t * S * LB::RS);
                int d;
                uint32_t hit;
                rec_eval01<W, LW, TERN>(qr, r, a.K, d, hit);
                const bool valid = t * S + slot < cntb;
                CT o = 0;
                if (!MASKED) {
                    if (valid) o = atomicAdd(&cnt[d * QW + ql], inc(hit));       // lanes resolve in item order
                } else {
                    for (int s = 0; s < S; ++s) {
                        if (slot == s && valid) o = atomicAdd(&cnt[d * QW + ql], inc(hit));
                    }
                }
                if (valid) credit(o, hit);
            }
        }
        cur = nxt;
    }
    if (prev) drain();
#pragma unroll
    for (int o = QW; o < 64; o <<= 1) acc += __shfl_xor(acc, o, 64);
    if (slot == 0) ap_part[(int64_t)chunk_id * a.qpad + q] = acc;
}


// Does the LDS hand out same-address returning adds of one instruction in ascending lane order?  (see the header)
__global__ __launch_bounds__(64) void k_probe_lane_order(uint32_t* __restrict__ ok_out) {
    __shared__ unsigned long long c64[256];
    __shared__ uint32_t c32[256];
    const int lane = threadIdx.x;
    bool ok = true;
    for (int round = 0; round < 24; ++round) {
        const int S = 2 << (round % 3 == 2 ? (round % 6 == 5 ? 4 : 2) : round % 3);   // 2, 4, 8, 2, 4, 32, ...
        const int QW = 64 / S;
        const int ql = lane & (QW - 1);
        const int row = (lane * 7 + round * 5 + (lane >> 3)) % 3;                       // a few "buckets" per query
        const int cell = row * QW + ql;
        for (int e = lane; e < 256; e += 64) { c64[e] = 1000ull + round; c32[e] = 77u + round; }
        __syncthreads();
        int before = 0;
        for (int l = 0; l < lane; ++l) {
            const int lrow = (l * 7 + round * 5 + (l >> 3)) % 3;
            before += (lrow * QW + (l & (QW - 1))) == cell;
        }
        const unsigned long long r64 = atomicAdd(&c64[cell], 1ull | (1ull << 32));
        const uint32_t r32 = atomicAdd(&c32[cell], 0x10001u);
        ok = ok && r64 == (1000ull + round) + ((unsigned long long)before | ((unsigned long long)before << 32));
        ok = ok && r32 == 77u + round + (uint32_t)before * 0x10001u;
        __syncthreads();
    }
    const bool all_ok = __all(ok);
    if (lane == 0) *ok_out = all_ok ? 1u : 0u;
}

// The same question AT PASS 2'S GEOMETRY AND UNDER LOAD (round 6): one wave per block with k_scan_ap_c's counter
// array -- 65 bucket rows x 16 queries x 8 bytes, row stride 128 bytes, lane = slot * 16 + query --, five such waves per SIMD like
// the product launch, eight returning 64-bit adds {1, -relevant} in flight before a counted wait, and every fifth block an MFMA loop
// instead, so that probing waves share their SIMDs with matrix waves as pass 2 shares the chip with nothing less busy.  A few
// hundred million adds in all -- the size of the headline evaluation -- in a fraction of a millisecond, once per device and process.
// The check needs no model of the counters' history: the lanes of one instruction that hit the same cell must have been served in
// ascending slot order, i.e. a lane's return = the lowest such lane's return + {lower lanes of the cell, - relevant lower lanes}; a
// shadow array (one plain LDS update per cell and instruction) must equal the counters at the end.  `fault` (XMH_SCAN_PROBE_FAULT,
// the test hook) makes the expectation that of a device serving the lanes in DESCENDING order, which no device does: the probe then
// fails exactly as it would on hardware with another order.
constexpr int kProbeRows = 65, kProbeRounds = 96;
__global__ __launch_bounds__(64) void k_probe_lane_order_load(uint32_t* __restrict__ bad_out, int fault, float* __restrict__ sink) {
    __shared__ unsigned long long cnt[kProbeRows * 16];
    __shared__ unsigned long long shadow[kProbeRows * 16];
    const int lane = threadIdx.x;
    if (blockIdx.x % 5 == 4) {                                   // a matrix wave beside the probing ones
        typedef float v4f __attribute__((ext_vector_type(4)));
        typedef _Float16 v8h __attribute__((ext_vector_type(8)));
        v8h x, y;
        for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(0.001f * (float)(lane + i)); y[i] = (_Float16)(0.002f * (float)(lane ^ i)); }
        v4f acc = {0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < kProbeRounds * 24; ++it) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc, 0, 0, 0);
        if (acc[0] == 12345.678f) sink[lane] = acc[1];           // (never: keeps the loop)
        return;
    }
    const int ql = lane & 15, slot = lane >> 4;
    for (int e = lane; e < kProbeRows * 16; e += 64) {
        cnt[e] = (unsigned long long)(0x4b000000u + 3u * e) | ((unsigned long long)(0x4b7fffffu - 5u * e) << 32);      // float-bit counters as pass 2 starts them
        shadow[e] = cnt[e];
    }
    __syncthreads();
    const uint32_t cntbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned long long*)cnt + ql * 8;
    uint32_t bad = 0;
    uint32_t h = 0x9E3779B9u * (blockIdx.x + 1) + 0x85ebca6bu * (uint32_t)lane;
    for (int round = 0; round < kProbeRounds; ++round) {
        uint32_t d[8], m[8];
        unsigned long long old[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            h ^= h << 13; h ^= h >> 17; h ^= h << 5;
            // few distinct rows per instruction: the four slots of a query collide often, in every pattern
            d[u] = ((uint32_t)(round * 8 + u) * 7u + ((h >> 9) % 3u) * 11u) % (uint32_t)kProbeRows;
            m[u] = (h & 0x70u) ? 0u : ~0u;                        // about one lane in eight "relevant"
            const uint32_t addr = cntbase + d[u] * 128u;
            const unsigned long long inc = 1ull | ((unsigned long long)m[u] << 32);
            asm volatile("ds_add_rtn_u64 %0, %1, %2" : "=v"(old[u]) : "v"(addr), "v"(inc) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(old[0]), "+v"(old[1]), "+v"(old[2]), "+v"(old[3]), "+v"(old[4]), "+v"(old[5]), "+v"(old[6]), "+v"(old[7])::"memory");
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            // the other three slots of this query, same instruction
            uint32_t before = 0, before_rel = 0, group = 0, group_rel = 0;
            unsigned long long first = old[u];
            int first_slot = slot;
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const int src = s2 * 16 + ql;
                const uint32_t d2 = (uint32_t)__shfl((int)d[u], src, 64), m2 = (uint32_t)__shfl((int)m[u], src, 64);
                const unsigned long long o2 = ((unsigned long long)(uint32_t)__shfl((int)(old[u] >> 32), src, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)old[u], src, 64);
                if (d2 != d[u]) continue;
                ++group;
                group_rel += m2 & 1u;
                const bool lower = fault ? s2 > slot : s2 < slot;     // the order this device is expected to serve the lanes in
                if (lower) { ++before; before_rel += m2 & 1u; }
                const bool firster = fault ? s2 > first_slot : s2 < first_slot;
                if (firster) { first = o2; first_slot = s2; }
            }
            const uint32_t want_lo = (uint32_t)first + before, want_hi = (uint32_t)(first >> 32) - before_rel;
            if ((uint32_t)old[u] != want_lo || (uint32_t)(old[u] >> 32) != want_hi) bad = 1;
            if (first_slot == slot) {                             // one lane per cell and instruction keeps the books
                const unsigned long long sv = shadow[d[u] * 16 + ql];
                shadow[d[u] * 16 + ql] = (unsigned long long)((uint32_t)sv + group) | ((unsigned long long)((uint32_t)(sv >> 32) - group_rel) << 32);
            }
        }
    }
    __syncthreads();
    for (int e = lane; e < kProbeRows * 16; e += 64)
        if (cnt[e] != shadow[e]) bad = 1;
    if (__any(bad != 0) && lane == 0) atomicOr(bad_out, 1u);
}

// sharded evaluation: the all-gathered per-shard bucket histograms hist_g[world][2][Q][nb] (plane 0 = all
// items, plane 1 = relevant) -> the rank offsets pass 2 starts from on shard `rank`:
//   base[q][d] = (# items in buckets < d on ANY shard) + (# items in bucket d on shards < rank);  nrel[q] = all relevant.
// One wave per query, lanes over buckets (coalesced), wave prefix scan per 64-bucket segment.
__global__ __launch_bounds__(64) void k_shard_offsets(const uint32_t* __restrict__ hist_g, int world, int rank, int Q, int nb,
                                                      uint32_t* __restrict__ base_all, uint32_t* __restrict__ base_rel,
                                                      uint32_t* __restrict__ nrel_total) {
    const int q = blockIdx.x, lane = threadIdx.x;
    const int64_t plane = (int64_t)Q * nb;
    uint32_t carry_a = 0, carry_r = 0;
    for (int d0 = 0; d0 < nb; d0 += 64) {
        const int d = d0 + lane;
        uint32_t ta = 0, tr = 0, la = 0, lr = 0;
        if (d < nb) {
            for (int w = 0; w < world; ++w) {
                const uint32_t* h = hist_g + (int64_t)w * 2 * plane + (int64_t)q * nb + d;
                const uint32_t a = h[0], r = h[plane];
                ta += a; tr += r;
                if (w < rank) { la += a; lr += r; }
            }
        }
        uint32_t sa = ta, sr = tr;                              // inclusive scan over the segment
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t ua = __shfl_up(sa, o, 64), ur = __shfl_up(sr, o, 64);
            if (lane >= o) { sa += ua; sr += ur; }
        }
        if (d < nb) {
            base_all[(int64_t)q * nb + d] = carry_a + sa - ta + la;
            base_rel[(int64_t)q * nb + d] = carry_r + sr - tr + lr;
        }
        carry_a += __shfl(sa, 63, 64);
        carry_r += __shfl(sr, 63, 64);
    }
    if (lane == 0) nrel_total[q] = carry_r;
}

// The sharded call's offsets in one launch, from the all-gathered TOTALS TABLES of the shards in the workspace's own layout
// (tot_g[world][nb][qpad] {all, relevant}: what k_scan_below leaves behind, gathered as it is -- no export pass, no transposed
// copy): dpre[d][q] = items of lower buckets on any shard + items of bucket d on lower shards, cap_ws[q] = relevant items on all
// shards, cap_out[q] = min(that, k).  64 queries per block (lanes, coalesced), the bucket axis split over the 4 waves as in
// k_scan_dpre.
__global__ __launch_bounds__(256) void k_shard_offsets_dpre(const uint2* __restrict__ tot_g, int world, int rank, int Q, int qpad, int nb,
                                                            int64_t kcap, uint2* __restrict__ dpre, uint32_t* __restrict__ cap_ws,
                                                            int32_t* __restrict__ cap_out, uint32_t* __restrict__ items_total) {
    __shared__ uint2 part[4][64];
    const int lane = threadIdx.x & 63, wq = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + lane;
    const int64_t plane = (int64_t)nb * qpad;
    const int nbq = (nb + 3) / 4;
    const int d0 = wq * nbq, d1 = (d0 + nbq < nb) ? d0 + nbq : nb;
    uint32_t sa = 0, sr = 0;
    for (int d = d0; d < d1; ++d) {
        for (int w = 0; w < world; ++w) {
            const uint2 t = tot_g[(int64_t)w * plane + (int64_t)d * qpad + q];
            sa += t.x;
            sr += t.y;
        }
    }
    part[wq][lane] = make_uint2(sa, sr);
    __syncthreads();
    uint32_t ra = 0, rr = 0, tr = 0, ta_all = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint2 pw = part[w][lane];
        if (w < wq) { ra += pw.x; rr += pw.y; }
        tr += pw.y;
        ta_all += pw.x;
    }
    if (items_total && blockIdx.x == 0 && threadIdx.x == 0) *items_total = ta_all;      // every query sees every item: the gallery size over all shards
    for (int d = d0; d < d1; ++d) {
        uint32_t ta = 0, tr2 = 0, la = 0, lr = 0;
        for (int w = 0; w < world; ++w) {
            const uint2 t = tot_g[(int64_t)w * plane + (int64_t)d * qpad + q];       // second walk: L2 hits
            ta += t.x; tr2 += t.y;
            if (w < rank) { la += t.x; lr += t.y; }
        }
        dpre[(int64_t)d * qpad + q] = make_uint2(ra + la, rr + lr);
        ra += ta;
        rr += tr2;
    }
    if (wq == 0) {
        cap_ws[q] = tr;
        if (q < Q) cap_out[q] = (int32_t)((kcap > 0 && (uint64_t)kcap < (uint64_t)tr) ? (uint32_t)kcap : tr);
    }
}

// The all-to-all form of the sharded exchange: instead of every rank receiving every shard's whole totals table
// (world x 2.6 MB at Q 5000, K 64), rank j receives from every shard only the column slice of ITS S = qpad / world queries
// (tot_s[world][nb][S] {all, relevant}), resolves the offsets of those queries for EVERY shard
//   out[w][d][s] = items of lower buckets on any shard + items of bucket d on shards < w        (d < nb)
//   out[w][nb][s] = {relevant items on all shards, items on all shards}
// and a second all-to-all hands shard w its rows back.  64 queries per block (lanes), the bucket axis split over the 4 waves.
__global__ __launch_bounds__(256) void k_shard_slice_offsets(const uint2* __restrict__ tot_s, int world, int nb, int S, uint2* __restrict__ out) {
    __shared__ uint2 part[4][64];
    const int lane = threadIdx.x & 63, wq = threadIdx.x >> 6;
    const int s = blockIdx.x * 64 + lane;
    const bool ok = s < S;
    const int64_t plane_in = (int64_t)nb * S, plane_out = (int64_t)(nb + 1) * S;
    const int nbq = (nb + 3) / 4;
    const int d0 = wq * nbq, d1 = (d0 + nbq < nb) ? d0 + nbq : nb;
    uint32_t sa = 0, sr = 0;
    for (int d = d0; d < d1; ++d) {
        for (int w = 0; w < world; ++w) {
            const uint2 t = ok ? tot_s[(int64_t)w * plane_in + (int64_t)d * S + s] : make_uint2(0u, 0u);
            sa += t.x;
            sr += t.y;
        }
    }
    part[wq][lane] = make_uint2(sa, sr);
    __syncthreads();
    uint32_t ra = 0, rr = 0, ta_all = 0, tr_all = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint2 pw = part[w][lane];
        if (w < wq) { ra += pw.x; rr += pw.y; }
        ta_all += pw.x;
        tr_all += pw.y;
    }
    if (!ok) return;
    for (int d = d0; d < d1; ++d) {
        uint32_t la = 0, lr = 0;
        for (int w = 0; w < world; ++w) {                            // second walk: L2 hits
            const uint2 t = tot_s[(int64_t)w * plane_in + (int64_t)d * S + s];
            out[(int64_t)w * plane_out + (int64_t)d * S + s] = make_uint2(ra + la, rr + lr);
            la += t.x;
            lr += t.y;
        }
        ra += la;
        rr += lr;
    }
    if (wq == 0)
        for (int w = 0; w < world; ++w) out[(int64_t)w * plane_out + (int64_t)nb * S + s] = make_uint2(tr_all, ta_all);
}

// ... and on shard w the rows that came back, offs[owner][nb + 1][S], go where pass 2 reads them: dpre[d][q], the relevant count,
// the cap, and the gallery size over all shards (the gate word of k_scan_ap_c)
__global__ __launch_bounds__(256) void k_shard_scatter_offsets(const uint2* __restrict__ offs, int nb, int S, int Q, int qpad, int64_t kcap,
                                                               uint2* __restrict__ dpre, uint32_t* __restrict__ cap_ws, int32_t* __restrict__ cap_out,
                                                               uint32_t* __restrict__ items_total) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= qpad) return;
    const int owner = q / S, s = q - owner * S;
    const uint2* __restrict__ src = offs + (int64_t)owner * (nb + 1) * S + s;
    const int d = blockIdx.y;
    if (d < nb) {
        dpre[(int64_t)d * qpad + q] = src[(int64_t)d * S];
        return;
    }
    const uint2 t = src[(int64_t)nb * S];                           // {relevant items on all shards, items on all shards}
    cap_ws[q] = t.x;
    if (q < Q) cap_out[q] = (int32_t)((kcap > 0 && (uint64_t)kcap < (uint64_t)t.x) ? (uint32_t)kcap : t.x);
    if (q == 0) *items_total = t.y;
}

// Sum of a query's per-chunk credits.  A block takes 64 queries; its 4 waves take a quarter of the chunks each (all loads of a
// thread are issued together: one dependent load per chunk was a 13 us chain of misses for 32 chunks) and the quarters are added
// in a fixed order -- the same order in the sharded and the unsharded reduction, whatever the grid.
__device__ __forceinline__ double ap_chunk_sum(const float* __restrict__ ap_part, int qpad, int nchunk, int q, double (*quarter)[64]) {
    const int lane = threadIdx.x & 63, wq = threadIdx.x >> 6;
    const int per = (nchunk + 3) / 4;
    const int c0 = wq * per, c1 = (c0 + per < nchunk) ? c0 + per : nchunk;
    double s = 0.0;
    for (int c = c0; c < c1; c += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = c + j < c1 ? ap_part[(int64_t)(c + j) * qpad + q] : 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (c + j < c1) s += (double)v[j];
    }
    quarter[wq][lane] = s;
    __syncthreads();
    return ((quarter[0][lane] + quarter[1][lane]) + quarter[2][lane]) + quarter[3][lane];
}

// nrel_ws != null (unsharded call): cap_out[q] = min(nrel_ws[q], kcap) is written here; else the caller's cap_out is read.
// grid = qpad / 64 blocks of 256 threads.
__global__ __launch_bounds__(256) void k_ap_reduce(const float* __restrict__ ap_part, int Q, int qpad, int nchunk,
                                                   double* __restrict__ ap_sum, const uint32_t* __restrict__ nrel_ws, uint32_t kcap,
                                                   int32_t* __restrict__ cap_out) {
    __shared__ double quarter[4][64];
    const int q = blockIdx.x * 64 + (threadIdx.x & 63);
    const double s = ap_chunk_sum(ap_part, qpad, nchunk, q, quarter);
    if (threadIdx.x >= 64 || q >= Q) return;
    ap_sum[q] = s;
    if (nrel_ws) cap_out[q] = (int32_t)min(nrel_ws[q], kcap);
}

// k_ap_reduce + k_map_finalize in one launch (unsharded evaluation): every block also leaves the sum of ap/cap over its 64
// queries in part[], takes a ticket, and the block that draws the last ticket adds the partials IN BLOCK ORDER (deterministic)
// into the mean.  Hand-off as in k_scan_below: the partial goes out as an agent-scope store, vmcnt(0), then the ticket; the last
// block reads the partials with agent-scope loads, one per thread (a thread walking them was a chain of 20 dependent misses);
// no fence: a release would write back the XCD's whole L2.  The ticket word is zeroed by
// xmh_hamming_hist and put back to zero by the last block, so several evaluations can follow one pass 1.
// (One block doing all of it was tried: 40 us -- a single block cannot pull 1 MB fast.)
__global__ __launch_bounds__(256) void k_ap_reduce_map(const float* __restrict__ ap_part, int Q, int qpad, int nchunk,
                                                       const uint32_t* __restrict__ nrel_ws, uint32_t kcap, int32_t* __restrict__ cap_out,
                                                       double* __restrict__ ap_sum, double* __restrict__ part, uint32_t* __restrict__ ticket,
                                                       double* __restrict__ map_out) {
    __shared__ double quarter[4][64];
    __shared__ double red[64];
    __shared__ int last;
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 64 + lane;
    const double s = ap_chunk_sum(ap_part, qpad, nchunk, q, quarter);
    if (threadIdx.x < 64) {
        double term = 0.0;
        if (q < Q) {
            const int32_t cap = (int32_t)min(nrel_ws[q], kcap);
            ap_sum[q] = s;
            cap_out[q] = cap;
            term = s / (double)cap;                                  // cap == 0 -> NaN (0/0), like the reference
        }
        red[lane] = term;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double b = 0.0;
        for (int j = 0; j < 64; ++j) b += red[j];                    // in query order
        __hip_atomic_store(&part[blockIdx.x], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = t == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    double m = 0.0;                                                  // <= 4096 blocks: 16 rounds of 256 parallel loads, added in block order
    for (unsigned b0 = 0; b0 < gridDim.x; b0 += 256) {
        const unsigned b = b0 + threadIdx.x;
        __syncthreads();
        reinterpret_cast<double*>(quarter)[threadIdx.x] = b < gridDim.x ? __hip_atomic_load(&part[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned n = gridDim.x - b0 < 256u ? gridDim.x - b0 : 256u;
            for (unsigned j = 0; j < n; ++j) m += reinterpret_cast<double*>(quarter)[j];
        }
    }
    if (threadIdx.x == 0) {
        map_out[0] = m / (double)Q;
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
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
// lanes per query (S): smallest power of two that brings the counters of a wave under the budget.  Budgets from a sweep
// on MI355X (K = 16..256, dense and sparse relevance): 9 KB for 64-bit counters, 5 KB for 32-bit ones (3-4 waves/SIMD).
constexpr int kSlotBudget64 = 9 * 1024;
constexpr int kSlotBudget32 = 5 * 1024;
// Long codes (W > 8 words: TwDH-style 512..2048 bits) go up to S = 64, i.e. one query per wave and one
// gallery item per lane.
constexpr int slots_for(int W, bool tern, int counter_bytes) {
    const int nbmax = (tern ? 64 : 32) * W + 1;
    const int budget = counter_bytes == 8 ? kSlotBudget64 : kSlotBudget32;
    const int cap = W > 8 ? 64 : 8;
    int S = 2;
    while (S < cap && nbmax * (64 / S) * counter_bytes > budget) S *= 2;
    return S;
}
// the pair cache of k_scan_hist_s: two code words, binary, four slots in both passes and both counter widths
template <int W, bool TERN>
constexpr bool kPairCacheShape = !TERN && (W == 1 || W == 2 || W == 4 || W == 8);
// ... and the slots both passes run with when the cache is in use: 4 x 16 queries with one-byte entries up to 64 bits (for codes of at
// most 32 bits that is NOT the geometry the uncached kernels pick, 2 x 32), 8 x 8 queries with two-byte entries beyond
constexpr int cache_slots(int W) { return W <= 2 ? 4 : 8; }
static_assert(slots_for(2, false, 4) == 4 && slots_for(2, false, 8) == 4 && slots_for(4, false, 4) == 8 && slots_for(4, false, 8) == 8 &&
              slots_for(8, false, 4) == 8 && slots_for(8, false, 8) == 8, "33..256 bits: the cache geometry is the uncached kernels' own");
// waves per block: 8 where a wave owns a single query (S = 64), so that 8 queries share each staged gallery batch
constexpr int waves_for(int W, bool tern) { return (W >= 32 && !tern) ? 8 : 1; }
inline size_t aos_ring_bytes(int W, int LW, bool tern) { return (size_t)64 * (((W * (tern ? 2 : 1) + LW) + 3) / 4 * 4) * 4; }

struct WsLayout {
    size_t chunk_hist, below, tot, dpre, cap, tick, gate, ap_part, pair_cache, total;
};

// ---- which kernels take a shape (round 5: one MFMA path + the VALU kernels per code-length band; DESIGN 3.1 has the table) ----------
//   binary, K <= 64        k_scan_hist_r2            one-byte pair cache -> k_scan_ap_c     (no cache: k_scan_ap_r2)
//   binary, 65 .. 128      k_scan_hist_r2w           one-byte pair cache -> k_scan_ap_c<., 8, HALF>; a distance of 128 wraps: stand-in k_scan_ap_s
//   binary, 129 .. 256     k_scan_hist_b             two-byte pair cache -> cached k_scan_ap_s
//   anything else, more than 128 classes, XMH_SCAN_MFMA=0, a failed self-check or lane-order probe: k_scan_hist_s / k_scan_ap_s (VALU)
// k_scan_hist_r2 / r2w keep hazards the compiler cannot see apart by hand (s_nop counts, early-clobber operands, statement order: see
// the kernel), each of which was a wrong result on hardware before it was a comment.  A hipcc or ROCm change could break one of them
// silently, so the kernels have to EARN their place once per device and process: r2_selfcheck_ok runs a small scan (several chunks,
// ties, every geometry) with and without them and compares histograms and divisors bit for bit, AP sums to float rounding; a mismatch
// prints one line to stderr and every later plan of this process uses the VALU kernels, as XMH_SCAN_MFMA=0 does.
// XMH_SCAN_M2_SELFCHECK=0 skips it, =2 runs it and pretends it failed (the test of the fallback); g_r2_force pins the answer while the
// check itself runs.
static thread_local int g_r2_force = -1;
bool r2_selfcheck_ok();
inline bool mfma_env_on() {                              // read per call: the tests compare the two families in one process
    const char* e = getenv("XMH_SCAN_MFMA");
    return !(e && atoi(e) == 0);
}
inline bool r2_enabled() {
    if (g_r2_force >= 0) return g_r2_force != 0;
    return mfma_env_on() && r2_selfcheck_ok();
}
inline bool r2_shape(int K, bool ternary) { return !ternary && K <= 64 && r2_enabled(); }
inline bool r2w_shape(int K, bool ternary) { return !ternary && K > 64 && K <= 128 && r2_enabled(); }
// k_scan_hist_b (xmh_scan_bits.hip, round 3): 129..256-bit binary codes build their MFMA operands from the packed bits in registers;
// counters in the (all << 16 | relevant) form of k_scan_hist_r2.  Compiler-scheduled intrinsics: no hand-kept hazard, no self-check.
inline bool bits_shape(int K, bool ternary) { return !ternary && K > 128 && K <= 256 && mfma_env_on() && g_r2_force != 0; }
// Round 6: TERNARY codes (up to 256 bits: every length the zero planes exist for) on the same kernel (k_scan_hist_b<., ., ., TERN>): the operand is the 2K-bit pair of planes
// [+1 | -1], 2K + 1 bucket rows, two-byte cache entries -- so pass 2 is that of the 129..256-bit binary codes.  Before: the VALU kernels
// (one exact 0.0 among the code elements cost 2.7 x the binary evaluation).
inline bool tbits_shape(int K, bool ternary) { return ternary && K <= 256 && mfma_env_on() && g_r2_force != 0; }
inline int tbits_tiles(int K) { return K <= 64 ? 2 : (K <= 128 ? 4 : 8); }      // 64-bit tiles of the 2K-bit operand
inline bool mfma_shape(int K, bool ternary) { return r2_shape(K, ternary) || r2w_shape(K, ternary) || bits_shape(K, ternary) || tbits_shape(K, ternary); }
// k_scan_ap_r2 (round 5): pass 2 evaluates the pairs again on the MFMA from the packed words instead of reading a pair cache (binary codes
// of at most 64 bits whose pass 1 is k_scan_hist_r2).  Measured at Q 5000 x R 117 218 x 64 bit: pass 1 without the cache stores 0.157 ->
// 0.128 ms, pass 2 0.173 (k_scan_ap_c on the cache) -> 0.262 ms, step 0.361 -> 0.416 ms -- so it is the pass 2 of evaluations that HAVE no
// cache (XMH_SCAN_CACHE_MB exceeded or 0, a workspace without room for it), where it replaces the VALU re-evaluation of k_scan_ap_s.
// XMH_SCAN_AP_R2=1 drops the cache for every such shape (the A/B of DESIGN 3.1), =0 never launches it (read per call; a histogram / ap
// call pair must see the same value).
inline int ap_r2_mode() {
    const char* e = getenv("XMH_SCAN_AP_R2");
    return e ? (atoi(e) != 0 ? 1 : 0) : 2;
}
constexpr int kAp2Waves = 4, kAp2Groups = 2;           // k_scan_ap_r2: waves per block x query groups of 16 per wave (66 KB of counters at 65 bucket rows, two
                                                       // blocks per CU; 2 x 2, 1 x 2 and 1 x 1 measured the same 0.262-0.275 ms at the headline shape)
constexpr int kMfmaWaves = 4;                          // k_scan_hist_b: waves (16 queries each) per block
static_assert(xmh::kScanBitsWaves == kMfmaWaves, "k_scan_hist_b takes the plan's query tiles of 64");
// largest chunk of k_scan_hist_b's plan, a multiple of 64.  Round 6: 8064 -> 24192.  The [chunk][bucket][query] tables weigh 257 (513) rows per
// chunk here: on configs[4]'s shard 159 chunks made k_scan_below 0.5 ms and the counter start values 1.7 GB of pass 2's reads; 8064 / 16128 /
// 24192 / 32256: 6.6 / 5.97 / 5.93 / 5.99 ms per step (COCO shape: ternary 128 bit 0.83 -> 0.78, 256-bit binary 0.75 -> 0.72)
constexpr int kBitsMaxChunk = 24192;
// waves per block x query groups of 16 per wave of k_scan_hist_r2 / r2w, and the blocks per CU the chunk count is sized for: codes of at
// most 32 bits have so few buckets that one block of 256 queries per CU measured best (Q 5000 x R 117 218, blocks per CU x rounds: K=16
// 0.416 / 0.350 / 0.367 / 0.352 / 0.368 ms per step for 1 / 2 / 3 / 4 / 6); 33..64 bits: 66 KB of counters per block, two per CU;
// 65..128 bits: 129 bucket rows, two groups per wave (66 KB), two blocks per CU.
struct R2Geom { int nw, nq, blocks_per_cu; int queries() const { return nw * nq * 16; } };
inline R2Geom r2_geom(int K) { return K <= 32 ? R2Geom{4, 4, 1} : (K <= 64 ? R2Geom{4, 4, 2} : R2Geom{4, 2, 2}); }

// Pair cache: binary codes of at most 64 bits and of 65..128 bits (one byte per pair: distance << 1 | relevant) and 129..256 bits (two
// bytes), while it stays under XMH_SCAN_CACHE_MB (default 131072: an MI355X has 288 GB of HBM and up to there the cache still pays --
// the UNSHARDED configs[4] gallery, Q 5000 x R 10 M x 256 bit = 100 GB of entries, runs 89.6 -> 67.3 ms per step with it; 0 = off).
// The VALU kernels write and read the same layouts (k_scan_hist_s / k_scan_ap_s<.., CACHE>: 33..256 bits).
inline long long cache_cap_mb() {
    const char* cap_env = getenv("XMH_SCAN_CACHE_MB");              // read per call
    return cap_env ? atoll(cap_env) : 131072;
}
inline bool ap_c_on() {                                  // XMH_SCAN_AP_C=0: the integer-counter kernels take every pass 2 (tests: the path of galleries beyond 2^23 items)
    const char* e = getenv("XMH_SCAN_AP_C");
    return !(e && atoi(e) == 0);
}
size_t pair_cache_bytes(const xmh_scan_plan& p, int K, bool ternary) {
    const long long cap_mb = cache_cap_mb();
    const bool tb = tbits_shape(K, ternary);                        // ternary on the MFMA: two-byte entries, 8 slots (as 129..256-bit binary codes)
    if ((ternary && !tb) || K > 256 || cap_mb <= 0 || (!tb && K <= 32 && !r2_shape(K, ternary))) return 0;
    if (r2_shape(K, ternary) && ap_r2_mode() == 1) return 0;        // k_scan_ap_r2 evaluates the pairs itself
    const int S = tb ? 8 : (K <= 64 ? 4 : 8);                      // slots of the kernels that use it: 64 / S queries per wave
    // records of 64 lanes per (chunk, tile of 64 / S queries, batch of 64 items): 16 B per lane for one-byte entries (S = 4), 12 B per lane for the
    // 12-bit entries of longer and of ternary codes (S = 8; xmh_common.h) -- 65..128-bit codes whose one-byte entries k_scan_hist_r2w writes use
    // two thirds of that region
    const size_t bytes = (size_t)p.nchunk * (size_t)(p.nqtile * S) * (size_t)((p.chunk + 63) / 64) * (S == 4 ? 1024 : 64 * 4 * xmh::kCache12Dwords);
    return bytes <= (size_t)cap_mb << 20 ? bytes : 0;
}

WsLayout ws_layout(const xmh_scan_plan& p, size_t cache_bytes) {
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
    L.tick = take((size_t)p.nqtile * 4);     // k_scan_below tickets, one per 64-query tile; ends where `gate` starts: one memset clears both
    L.gate = take(256 + 8 * 4096);           // [0]: nrel_max gate word, [1]: finalize ticket, +256: per-block partial sums (<= 4096 blocks)
    L.ap_part = take((size_t)p.nchunk * p.qpad * 4);
    L.pair_cache = take(cache_bytes);        // LAST: a workspace that ends in front of it is a workspace without a cache (cache_for_workspace)
    L.total = o;
    return L;
}

// The cache a call may use given the workspace it was handed: the plan's, or none when the caller's buffer ends in front of it (a device
// that has no room for Q x R bytes next to a resident encoder: xmh_scan_ws_bytes_nocache).  Both calls of a pair see the same ws_bytes
// and so take the same decision -- no process-wide state (round 4 lowered XMH_SCAN_CACHE_MB in the environment for this: ADVICE r4).
size_t cache_for_workspace(const xmh_scan_plan& p, int K, bool ternary, size_t ws_bytes) {
    const size_t c = pair_cache_bytes(p, K, ternary);
    return c && ws_bytes < ws_layout(p, c).total ? 0 : c;
}

int make_plan(int64_t Q, int64_t R, int K, int ternary, xmh_scan_plan* p) {
    if (Q <= 0 || R <= 0 || K <= 0) return xmh::fail(XMH_EINVAL, "scan plan: bad shape Q=%lld R=%lld K=%d", (long long)Q, (long long)R, K);
    if (R >= (1ll << 24) || Q >= (1ll << 24)) return xmh::fail(XMH_ENOTSUP, "scan plan: shard too large (R=%lld, Q=%lld)", (long long)R, (long long)Q);
    const int64_t nb = ternary ? 2 * (int64_t)K + 1 : (int64_t)K + 1;
    const int Wd = (K + 31) / 32;
    if (Wd > 64 || (ternary && Wd > 8))
        return xmh::fail(XMH_ENOTSUP, "scan plan: K=%d%s (at most 2048 code bits, 256 with zero planes)", K, ternary ? " ternary" : "");
    const int Wc = Wd <= 1 ? 1 : (Wd <= 2 ? 2 : (Wd <= 4 ? 4 : (Wd <= 8 ? 8 : (Wd <= 16 ? 16 : (Wd <= 32 ? 32 : 64)))));
    const int S64 = slots_for(Wc, ternary != 0, 8);
    const int64_t lds_ap = nb * (64 / S64) * 8;
    if (lds_ap > 150 * 1024) return xmh::fail(XMH_ENOTSUP, "scan plan: %lld distance buckets need %lld B of LDS per wave; K=%d%s", (long long)nb, (long long)lds_ap, K, ternary ? " ternary" : "");
    const bool r2 = r2_shape(K, ternary != 0) || r2w_shape(K, ternary != 0);
    int64_t nqt = xmh::ceil_div(Q, 64);
    const int r2q = r2_geom(K).queries();
    if (r2) {                                     // whole blocks of k_scan_hist_r2 AND whole 64-query tiles
        int64_t l = r2q;
        while (l % 64) l += r2q;
        nqt = xmh::ceil_div(nqt * 64, l) * l / 64;
    }
    const int64_t wpc = 8;                        // resident waves per CU the VALU kernels are sized for (two per SIMD)
    // one chunk x 64-query tile per resident wave slot (`rounds` sets of them); slotted kernels run S waves per tile
    const int64_t slots = (int64_t)xmh::device_cu_count() * wpc;
    // two resident sets of waves balance the tail better, but the [chunk][bucket][query] tables double: pays up to 65 buckets
    // (K=64: 0.640 -> 0.624 ms, K=16: 0.502 -> 0.476), costs at 257 (K=256: 1.45 -> 1.54); with the pair cache (33..64 bits) the passes
    // are shorter and the tables weigh more: one set (0.543 -> 0.536 ms); codes of 32 bits and less: three sets
    const bool cache_shape = !ternary && K > 32 && K <= 64 && cache_cap_mb() > 0;
    const int64_t rounds = nb <= 33 ? 3 : (nb <= 65 && !cache_shape ? 2 : 1);
    int64_t nchunk = rounds * slots / nqt;
    if (S64 > 8) nchunk = nchunk * 8 / S64;       // long codes: S waves per tile already fill the slots; fewer chunks = smaller tables
    const bool mfma = mfma_shape(K, ternary != 0);
    if (mfma) {
        // k_scan_hist_b: 257 (513) bucket rows: three sets of blocks (COCO shape, sets 2 / 3 / 4 / 6: ternary 128 bit 0.93 / 0.78 / 0.83 / 0.83 ms, 256-bit
        // binary 0.85 / 0.73 / 0.75 / 0.76); ternary codes of at most 64 bits have 129 rows and smaller tables: six (0.65 -> 0.62 ms)
        int64_t bsets = nb <= 129 ? 6 : 3;
        if (const char* e = xmh_experiment_env("XMH_BITS_SETS")) bsets = atoi(e) > 0 ? atoi(e) : bsets;
        nchunk = bsets * xmh::device_cu_count() / nqt;
        if (r2) {                                 // blocks of r2q queries, blocks_per_cu of them per CU, `sets` sets of them
            int sets = K > 32 && K <= 64 ? 1 : 2;                   // k_scan_hist_r2 at 33..64 bits: one set (1 / 2 / 3: 0.178 / 0.187 / 0.199 ms)
            if (const char* e = xmh_experiment_env("XMH_R2_SETS")) sets = atoi(e) > 0 ? atoi(e) : sets;
            nchunk = (int64_t)sets * xmh::device_cu_count() * r2_geom(K).blocks_per_cu / (nqt * 64 / r2q);
        }
    }
    if (nchunk < 1) nchunk = 1;
    if (nchunk > 8) nchunk = (nchunk + 4) / 8 * 8;      // whole XCD groups: every XCD gets the same number of chunks
    int64_t chunk = xmh::ceil_div(R, nchunk);
    if (chunk < kMinChunk) chunk = kMinChunk;
    if (chunk > kMaxChunk) chunk = kMaxChunk;
    chunk = xmh::ceil_div(chunk, 8) * 8;
    if (mfma) {                                   // batches of 64 items
        chunk = xmh::ceil_div(chunk, 64) * 64;
        int64_t bits_max = kBitsMaxChunk;
        if (const char* e = xmh_experiment_env("XMH_BITS_MAX_CHUNK")) bits_max = atoll(e) / 64 * 64 > 0 ? atoll(e) / 64 * 64 : bits_max;
        if (!r2 && chunk > bits_max) {                              // capped: whole XCD groups of equal chunks again
            nchunk = xmh::ceil_div(xmh::ceil_div(R, bits_max), 8) * 8;
            chunk = xmh::ceil_div(xmh::ceil_div(R, nchunk), 64) * 64;
        }
    }
    nchunk = xmh::ceil_div(R, chunk);
    p->chunk = chunk;
    p->nchunk = nchunk;
    p->nqtile = nqt;
    p->qpad = nqt * 64;
    p->nbuckets = nb;
    p->ws_bytes = ws_layout(*p, pair_cache_bytes(*p, K, ternary != 0)).total;
    return XMH_OK;
}

// 1 if same-address returning LDS adds of one instruction come back in ascending lane order on this device (probed once
// per process with a 1-wave kernel and one blocking copy), 0 otherwise or when XMH_SCAN_MASKED is set
static thread_local bool g_force_masked = false;      // xmh_scan_verify: the re-derivation runs the kernels that do not rely on the lane order
int lane_order_ok(hipStream_t st) {
    static int cached[64];
    static bool have[64];
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (g_force_masked) return 0;
    if (getenv("XMH_SCAN_MASKED") != nullptr) return 0;           // read per call: the self-check test runs both paths in one process
    std::lock_guard<std::mutex> lock(mu);
    if (have[dev]) return cached[dev];
    int ok = 0;
    {
        // per DEVICE (not per process: a process may drive several), two probes: one wave alone, every counter geometry; then pass 2's own
        // geometry under load beside matrix waves (k_probe_lane_order_load).  Both must hold.
        const char* fe = getenv("XMH_SCAN_PROBE_FAULT");          // test hook, read once per device: expect an order no device serves
        const int fault = fe && atoi(fe) != 0;
        uint32_t* flag = nullptr;
        float* sink = nullptr;
        if (hipMalloc(&flag, 8) == hipSuccess && hipMalloc(&sink, 256) == hipSuccess) {
            uint32_t h[2] = {0, 1};
            int cus = xmh::device_cu_count();
            if (hipMemsetAsync(flag, 0, 8, st) == hipSuccess) {
                hipLaunchKernelGGL(k_probe_lane_order, dim3(1), dim3(64), 0, st, flag);
                hipLaunchKernelGGL(k_probe_lane_order_load, dim3((unsigned)(cus * 25)), dim3(64), 0, st, flag + 1, fault, sink);
                if (hipGetLastError() == hipSuccess && hipMemcpyAsync(h, flag, 8, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess)
                    ok = h[0] == 1u && h[1] == 0u;
            }
        }
        if (flag) (void)hipFree(flag);
        if (sink) (void)hipFree(sink);
        if (!ok)
            fprintf(stderr, "xmh: the lane-order probe FAILED on device %d%s: same-address returning LDS adds are not served in lane order here; "
                            "pass 2 runs the masked kernels (about 2x slower, same results).\n", dev, fault ? " (XMH_SCAN_PROBE_FAULT set: forced)" : "");
    }
    cached[dev] = ok;
    have[dev] = true;
    return ok;
}

namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    bool alloc(size_t n) { return hipMalloc(&p, n) == hipSuccess; }
    template <typename T> T* as() { return static_cast<T*>(p); }
};
// one (K, forced kernel choice) evaluation of the self-check scan into host vectors; false on any HIP / library error
bool selfcheck_eval(int K, int force, const std::vector<uint32_t>& qb, const std::vector<uint32_t>& ql, const std::vector<uint32_t>& rb,
                    const std::vector<uint32_t>& rl, int64_t Q, int64_t R, int C, std::vector<uint32_t>& hist, std::vector<double>& ap,
                    std::vector<int32_t>& cap) {
    g_r2_force = force;
    bool ok = false;
    do {
        xmh_scan_plan p;
        if (xmh_scan_plan_make(Q, R, K, 0, &p) != XMH_OK) break;
        const size_t nb = (size_t)p.nbuckets;
        DevBuf dq, dql, dr, drl, ws, dh, dap, dcap;
        if (!dq.alloc(qb.size() * 4) || !dql.alloc(ql.size() * 4) || !dr.alloc(rb.size() * 4) || !drl.alloc(rl.size() * 4) || !ws.alloc(p.ws_bytes) ||
            !dh.alloc(2 * Q * nb * 4) || !dap.alloc(Q * 8) || !dcap.alloc(Q * 4))
            break;
        if (hipMemcpy(dq.p, qb.data(), qb.size() * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dql.p, ql.data(), ql.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(dr.p, rb.data(), rb.size() * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(drl.p, rl.data(), rl.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
            break;
        uint32_t* h = dh.as<uint32_t>();
        if (xmh_hamming_hist(dq.as<uint32_t>(), nullptr, dql.as<uint32_t>(), dr.as<uint32_t>(), nullptr, drl.as<uint32_t>(), Q, R, K, C, ws.p, p.ws_bytes, h, h + Q * nb,
                             nullptr) != XMH_OK)
            break;
        if (xmh_hamming_ap(dq.as<uint32_t>(), nullptr, dql.as<uint32_t>(), dr.as<uint32_t>(), nullptr, drl.as<uint32_t>(), Q, R, K, C, ws.p, p.ws_bytes, nullptr, nullptr,
                           nullptr, 0, dap.as<double>(), dcap.as<int32_t>(), nullptr) != XMH_OK)
            break;
        hist.resize(2 * Q * nb); ap.resize(Q); cap.resize(Q);
        if (hipDeviceSynchronize() != hipSuccess) break;
        if (hipMemcpy(hist.data(), dh.p, hist.size() * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(ap.data(), dap.p, Q * 8, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(cap.data(), dcap.p, Q * 4, hipMemcpyDeviceToHost) != hipSuccess)
            break;
        ok = true;
    } while (false);
    g_r2_force = -1;
    return ok;
}
}  // namespace

bool r2_selfcheck_ok() {
    static int state[64];                                 // 0 unknown, 1 passed, 2 failed
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;     // no device: nothing can run, nothing to guard
    const char* e = getenv("XMH_SCAN_M2_SELFCHECK");
    if (e && atoi(e) == 0) return true;
    std::lock_guard<std::mutex> lock(mu);
    if (state[dev]) return state[dev] == 1;
    state[dev] = 1;                                       // the evaluations below re-enter through g_r2_force only
    const int64_t Q = 200, R = 9000;
    const int C = 80, LW = 3;
    bool same = true, ran = true;
    for (int K : {64, 40, 16, 100}) {                     // two code words, one + a partial one, <= 32 bits (all k_scan_hist_r2), 65..128 bits (k_scan_hist_r2w)
        const int W = (K + 31) / 32;
        std::vector<uint32_t> qb(Q * W), ql(Q * LW), rb(R * W), rl(R * LW);
        uint64_t x = 0x9E3779B97F4A7C15ull ^ (uint64_t)K;
        auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)(x >> 16); };
        const uint32_t last = K % 32 ? (1u << (K % 32)) - 1 : 0xffffffffu;
        for (auto& v : qb) v = rnd();
        for (int64_t i = 0; i < R; ++i)                   // a quarter of the gallery repeats earlier codes: ties inside and across chunks
            for (int w = 0; w < W; ++w) rb[i * W + w] = (i > 64 && (rnd() & 3) == 0) ? rb[(rnd() % 64) * W + w] : rnd();
        for (int64_t i = 0; i < Q; ++i) qb[i * W + W - 1] &= last;
        for (int64_t i = 0; i < R; ++i) rb[i * W + W - 1] &= last;
        for (int64_t i = 0; i < Q * LW; ++i) ql[i] = rnd() & rnd() & rnd() & (i % LW == LW - 1 ? 0xffffu : 0xffffffffu);
        for (int64_t i = 0; i < R * LW; ++i) rl[i] = rnd() & rnd() & rnd() & (i % LW == LW - 1 ? 0xffffu : 0xffffffffu);
        for (int64_t i = 0; i < Q; ++i) ql[i * LW] |= 1u;                            // every query has a relevant item
        for (int64_t i = 0; i < R; i += 7) rl[i * LW] |= 1u;
        std::vector<uint32_t> h1, h0;
        std::vector<double> a1, a0;
        std::vector<int32_t> c1, c0;
        if (!selfcheck_eval(K, 1, qb, ql, rb, rl, Q, R, C, h1, a1, c1) || !selfcheck_eval(K, 0, qb, ql, rb, rl, Q, R, C, h0, a0, c0)) { ran = false; break; }
        // histograms and divisors bit for bit; the two kernels chunk the gallery differently, so pass 2's per-chunk float partial sums
        // add in another order: the AP sums to float rounding (a wrong pair moves one by >= 1 / R relative, far above it)
        bool ap_close = true;
        for (int64_t i = 0; i < Q; ++i) ap_close = ap_close && fabs(a1[i] - a0[i]) <= 4e-6 * fabs(a0[i]) + 1e-9;
        if (h1 != h0 || c1 != c0 || !ap_close) { same = false; break; }
    }
    if (e && atoi(e) == 2) same = false;
    if (!ran) {                                           // out of memory or a launch failure: no verdict, do not cache one
        state[dev] = 0;
        return true;
    }
    if (!same) {
        state[dev] = 2;
        fprintf(stderr, "xmh: k_scan_hist_r2 self-check FAILED on device %d (results differ from the VALU kernels): falling back to the VALU scan "
                        "for this process.  Please report the hipcc / ROCm versions.\n", dev);
    }
    return state[dev] == 1;
}

template <bool TERN, typename F>
int dispatch_shape(int W, int LW, F&& f) {
#define XMH_CASE(WW, LL) \
    if (W == WW && LW == LL) return f(std::integral_constant<int, WW>{}, std::integral_constant<int, LL>{});
    XMH_CASE(1, 1) XMH_CASE(1, 2) XMH_CASE(1, 3) XMH_CASE(1, 4)
    XMH_CASE(2, 1) XMH_CASE(2, 2) XMH_CASE(2, 3) XMH_CASE(2, 4)
    XMH_CASE(4, 1) XMH_CASE(4, 2) XMH_CASE(4, 3) XMH_CASE(4, 4)
    XMH_CASE(8, 1) XMH_CASE(8, 2) XMH_CASE(8, 3) XMH_CASE(8, 4)
    if constexpr (!TERN) {                                 // 129 ... 256 classes (IAPR TC-12 has 255) as EIGHT label words: binary codes up to 256 bits,
        XMH_CASE(1, 8) XMH_CASE(2, 8) XMH_CASE(4, 8) XMH_CASE(8, 8)      // the VALU kernels; callers with 5 ... 7 words pad to 8 (Python: RankingScan does)
    }
    if constexpr (!TERN) {                                 // long binary codes
        XMH_CASE(16, 1) XMH_CASE(16, 2) XMH_CASE(16, 3) XMH_CASE(16, 4)
        XMH_CASE(32, 1) XMH_CASE(32, 2) XMH_CASE(32, 3) XMH_CASE(32, 4)
        XMH_CASE(64, 1) XMH_CASE(64, 2) XMH_CASE(64, 3) XMH_CASE(64, 4)
    }
#undef XMH_CASE
    return xmh::fail(XMH_ENOTSUP, "scan: unsupported shape W=%d code words (K in {<=32,64,128,256}, binary also 512,1024,2048), Lw=%d label words (C<=128; binary codes up to 256 bits also exactly 8 words: 129..256 classes)%s",
                     W, LW, TERN ? ", ternary" : "");
}

int check_common(const char* who, const uint32_t* qbits, const uint32_t* qzero, const uint32_t* qlab,
                 const uint32_t* rbits, const uint32_t* rzero, const uint32_t* rlab, int C, void* ws, size_t ws_bytes,
                 const xmh_scan_plan& p) {
    if (!qbits || !rbits || !qlab || !rlab || !ws) return xmh::fail(XMH_EINVAL, "%s: null pointer", who);
    if ((qzero == nullptr) != (rzero == nullptr)) return xmh::fail(XMH_EINVAL, "%s: zero masks must be given for both sides or neither", who);
    if (C <= 0) return xmh::fail(XMH_EINVAL, "%s: C=%d", who, C);
    const size_t need = ws_layout(p, 0).total;                       // at least the plan without its pair cache (xmh_scan_ws_bytes_nocache)
    if (ws_bytes < need) return xmh::fail(XMH_EINVAL, "%s: workspace too small (%zu < %zu; %zu with the pair cache)", who, ws_bytes, need, (size_t)p.ws_bytes);
    return XMH_OK;
}

ScanArgs make_args(const uint32_t* qbits, const uint32_t* qzero, const uint32_t* qlab, const uint32_t* rbits,
                   const uint32_t* rzero, const uint32_t* rlab, int64_t Q, int64_t R, int K, const xmh_scan_plan& p) {
    ScanArgs a;
    a.qbits = qbits; a.qzero = qzero; a.qlab = qlab;
    a.rbits = rbits; a.rzero = rzero; a.rlab = rlab;
    a.Q = (int)Q; a.R = (int)R; a.K = K;
    a.chunk = (int)p.chunk; a.nchunk = (int)p.nchunk; a.nqt = (int)p.nqtile; a.qpad = (int)p.qpad; a.nb = (int)p.nbuckets;
    a.pair_cache = nullptr;
    return a;
}

inline int scan_grid(const xmh_scan_plan& p) { return (int)(8 * p.nqtile * xmh::ceil_div(p.nchunk, 8)); }

template <typename KernT>
int raise_lds(KernT kern, size_t lds, const char* who) { return xmh::raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds, who); }

}  // namespace


namespace {
// the launch in front of pass 1: clears the control words of the call and reads every chunk's packed words on the XCD that will scan it
void launch_touch(const uint32_t* rbits, const uint32_t* rlab, int64_t R, int W, int LW, const xmh_scan_plan& p, char* base, const WsLayout& L, hipStream_t st) {
    hipLaunchKernelGGL(k_scan_touch, dim3((unsigned)(8 * kTouchPerChunk * xmh::ceil_div(p.nchunk, 8))), dim3(256), 0, st, rbits, rlab, R, W, LW, p.chunk,
                       (int)p.nchunk, reinterpret_cast<uint32_t*>(base + L.tick), (int)((L.gate + 256 - L.tick) / 4));
}

MfmaArgs r2_args(const uint32_t* qbits, const uint32_t* qlab, const uint32_t* rbits, const uint32_t* rlab, int64_t Q, int64_t R, int K, int W, int LW,
                 const xmh_scan_plan& p, int queries_per_block) {
    MfmaArgs a{qbits, (int)Q, (int)R, K, W, (int)p.chunk, (int)p.nchunk, (int)(p.qpad / queries_per_block), (int)p.nbuckets, (int)p.qpad};
    a.rbits = rbits; a.rlab = rlab; a.qlab = qlab; a.LW = LW;
    return a;
}

// binary codes of at most 64 bits: k_scan_hist_r2 in the plan's geometry (r2_geom)
template <int NML, int NW, int NQ>
int hist_r2_t(const MfmaArgs& a, const xmh_scan_plan& p, uint32_t* chunk_hist, uint4* cache, hipStream_t st) {
    const dim3 grid((unsigned)(8 * a.nqt * xmh::ceil_div(p.nchunk, 8)));
    const size_t lds = (size_t)NW * NQ * p.nbuckets * 16 * 4;
    xmh::ProfScope prof("scan_hist", st);
    auto go = [&](auto kern) {
        const int r2 = raise_lds(kern, lds, "xmh_hamming_hist");
        if (r2) return r2;
        hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, st, a, chunk_hist, cache);
        return (int)XMH_OK;
    };
    return cache ? go(k_scan_hist_r2<NML, NW, NQ, true>) : go(k_scan_hist_r2<NML, NW, NQ, false>);
}

// 65..128 bits: k_scan_hist_r2w (4 waves x 2 query groups)
template <int NML>
int hist_r2w_t(const MfmaArgs& a, const xmh_scan_plan& p, char* base, const WsLayout& L, uint32_t* chunk_hist, uint4* cache, hipStream_t st) {
    constexpr int NW = 4, NQ = 2;
    const dim3 grid((unsigned)(8 * a.nqt * xmh::ceil_div(p.nchunk, 8)));
    const size_t lds = (size_t)NW * NQ * p.nbuckets * 16 * 4;
    uint32_t* ovf = reinterpret_cast<uint32_t*>(base + L.gate) + kGateWrapped;
    xmh::ProfScope prof("scan_hist", st);
    auto go = [&](auto kern) {
        const int r2 = raise_lds(kern, lds, "xmh_hamming_hist");
        if (r2) return r2;
        hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, st, a, chunk_hist, cache, ovf);
        return (int)XMH_OK;
    };
    return cache ? go(k_scan_hist_r2w<NML, NW, NQ, true>) : go(k_scan_hist_r2w<NML, NW, NQ, false>);
}

int mfma_hist(const uint32_t* qbits, const uint32_t* qlab, const uint32_t* rbits, const uint32_t* rlab, int64_t Q, int64_t R, int K, int W, int LW,
              const xmh_scan_plan& p, char* base, const WsLayout& L, uint32_t* chunk_hist, uint4* cache, hipStream_t st,
              const uint32_t* qzero = nullptr, const uint32_t* rzero = nullptr) {
    if (qzero) {                                                     // tbits_shape: ternary codes, K <= 256
        XMH_HIP(hipMemsetAsync(base + L.tick, 0, L.gate + 256 - L.tick, st));       // the control words of the call
        xmh::ScanBitsArgs a{rbits, rlab, qbits, qlab, (int)Q, (int)R, K, W, LW, (int)p.chunk, (int)p.nchunk, (int)(p.qpad / (16 * kMfmaWaves)),
                            (int)p.nbuckets, (int)p.qpad};
        a.rzero = rzero;
        a.qzero = qzero;
        return xmh::launch_scan_hist_bits(a, tbits_tiles(K), chunk_hist, cache, st);
    }
    if (bits_shape(K, false)) {
        XMH_HIP(hipMemsetAsync(base + L.tick, 0, L.gate + 256 - L.tick, st));       // the control words of the call
        const xmh::ScanBitsArgs a{rbits, rlab, qbits, qlab, (int)Q, (int)R, K, W, LW, (int)p.chunk, (int)p.nchunk, (int)(p.qpad / (16 * kMfmaWaves)),
                                  (int)p.nbuckets, (int)p.qpad};
        return xmh::launch_scan_hist_bits(a, 4, chunk_hist, cache, st);
    }
    launch_touch(rbits, rlab, R, W, LW, p, base, L, st);
    XMH_LAUNCH_CHECK("xmh_hamming_hist control words");
    const R2Geom g = r2_geom(K);
    const MfmaArgs a = r2_args(qbits, qlab, rbits, rlab, Q, R, K, W, LW, p, g.queries());
    if (K > 64) return LW <= 2 ? hist_r2w_t<1>(a, p, base, L, chunk_hist, cache, st) : hist_r2w_t<2>(a, p, base, L, chunk_hist, cache, st);
    return LW <= 2 ? hist_r2_t<1, 4, 4>(a, p, chunk_hist, cache, st) : hist_r2_t<2, 4, 4>(a, p, chunk_hist, cache, st);
}

}  // namespace

extern "C" int xmh_scan_plan_make(int64_t Q, int64_t R, int K, int ternary, xmh_scan_plan* plan_host) {
    if (!plan_host) return xmh::fail(XMH_EINVAL, "xmh_scan_plan_make: null plan");
    return make_plan(Q, R, K, ternary, plan_host);
}

extern "C" size_t xmh_scan_pair_cache_bytes(int64_t Q, int64_t R, int K, int ternary) {
    xmh_scan_plan p;
    if (make_plan(Q, R, K, ternary, &p)) return 0;
    return pair_cache_bytes(p, K, ternary != 0);
}

extern "C" size_t xmh_scan_pair_cache_offset(int64_t Q, int64_t R, int K, int ternary) {
    xmh_scan_plan p;
    if (make_plan(Q, R, K, ternary, &p)) return (size_t)-1;
    return ws_layout(p, pair_cache_bytes(p, K, ternary != 0)).pair_cache;
}

// The workspace of the same plan WITHOUT the pair cache: a caller whose device has no room for xmh_scan_plan.ws_bytes (Q x R bytes of
// cache next to a resident encoder) hands xmh_hamming_hist / _ap a buffer of this size instead, and both passes run uncached (pass 2 of
// codes of at most 64 bits: k_scan_ap_r2).  The decision follows from the size handed in, per call: no process-wide state.
extern "C" size_t xmh_scan_ws_bytes_nocache(int64_t Q, int64_t R, int K, int ternary) {
    xmh_scan_plan p;
    if (make_plan(Q, R, K, ternary, &p)) return 0;
    return ws_layout(p, 0).total;
}

// Width of the rank field of the packed 32-bit pass-2 counters for an UNSHARDED evaluation of this shape, 0 = only the 64-bit kernels
// are launched.  One function for the launch path (hamming_ap_impl) and for xmh_scan_describe, so that the kernel name the bench looks
// up in a profile is the kernel that ran (the two had drifted apart on XMH_SCAN_NO_PACK32).
int packed_rank_bits(int K, int64_t R) {
    // XMH_SCAN_PACK32: "0" = never, "all" = every code length (tests); default: from 65 bits on (measured at Q 5000 x R 117 218, sparse
    // relevance, pass 2 packed / 64-bit: K=16 0.275 / 0.251 ms, K=64 0.199 / 0.195, K=128 0.274 / 0.286, K=256 0.347 / 0.412)
    const char* e = getenv("XMH_SCAN_PACK32");
    if (e && e[0] == '0') return 0;
    if (!(K > 64 || (e && e[0] == 'a'))) return 0;
    const int64_t rank_max = R + 2;
    int rank_bits = 0;
    while ((1ll << rank_bits) < rank_max) ++rank_bits;
    return rank_bits > 24 ? 0 : rank_bits;
}

// Which kernel instances an UNSHARDED mAP@all evaluation of this shape launches for the two passes (as rocprofv3 prints them, minus
// the namespace): bench_roofline.py picks the PMC rows of exactly these -- a prefix match once took the 128-bit kernel's row for
// the 64-bit headline.  Mirrors the dispatch of xmh_hamming_hist / hamming_ap_impl (lane order assumed to hold).
extern "C" int xmh_scan_describe(int64_t Q, int64_t R, int K, int C, int ternary, char* out, size_t out_bytes) {
    if (!out || out_bytes < 64) return xmh::fail(XMH_EINVAL, "xmh_scan_describe: buffer too small");
    xmh_scan_plan p;
    const int rc = make_plan(Q, R, K, ternary, &p);
    if (rc) return rc;
    const bool tern = ternary != 0;
    const int Wd = (K + 31) / 32, LW = (C + 31) / 32;
    const int Wc = Wd <= 1 ? 1 : (Wd <= 2 ? 2 : (Wd <= 4 ? 4 : (Wd <= 8 ? 8 : (Wd <= 16 ? 16 : (Wd <= 32 ? 32 : 64)))));
    const size_t cache = pair_cache_bytes(p, K, tern);
    const bool use_mfma = mfma_shape(K, tern) && LW <= 4;
    char p1[160], p2[200];
    const int NML = LW <= 2 ? 1 : 2;
    const int S4 = slots_for(Wc, tern, 4), S8 = slots_for(Wc, tern, 8);
    const R2Geom g = r2_geom(K);
    if (use_mfma && r2w_shape(K, tern)) snprintf(p1, sizeof(p1), "k_scan_hist_r2w<%d, %d, %d, %s>", NML, g.nw, g.nq, cache ? "true" : "false");
    else if (use_mfma && r2_shape(K, tern)) snprintf(p1, sizeof(p1), "k_scan_hist_r2<%d, %d, %d, %s>", NML, g.nw, g.nq, cache ? "true" : "false");
    else if (use_mfma && tbits_shape(K, tern)) {
        int nsh, nqt;
        xmh::scan_hist_bits_shape(tbits_tiles(K), true, &nsh, &nqt);
        snprintf(p1, sizeof(p1), "k_scan_hist_b<%d, %d, %d, %d, %s, true>", tbits_tiles(K), kMfmaWaves, nsh, nqt, cache ? "true" : "false");
    }
    else if (use_mfma) {
        int nsh, nqt;
        xmh::scan_hist_bits_shape(4, false, &nsh, &nqt);
        snprintf(p1, sizeof(p1), "k_scan_hist_b<4, %d, %d, %d, %s, false>", kMfmaWaves, nsh, nqt, cache ? "true" : "false");
    }
    else {
        const bool cached = cache && !tern && Wc <= 8;
        const int S = cached ? cache_slots(Wc) : S4;
        snprintf(p1, sizeof(p1), "k_scan_hist_s<%d, %d, %s, %d, %d, %s>", Wc, LW, tern ? "true" : "false", S, S == 64 ? waves_for(Wc, tern) : 1,
                 cached ? "true" : "false");
    }
    const bool packable = packed_rank_bits(K, R) > 0;                 // exactly what hamming_ap_impl launches for an unsharded evaluation
    const bool apc_on = ap_c_on();
    const bool byte128 = cache && use_mfma && r2w_shape(K, tern);     // one-byte entries of 65..128-bit codes
    const bool tb = tern && cache && use_mfma && tbits_shape(K, tern);  // ternary on the MFMA: the cached pass 2 of the 129..256-bit binary codes
    if (tb) {
        snprintf(p2, sizeof(p2), "%s%s", packable ? "k_scan_ap_s<8, 1, false, false, 8, true, false, 1, true>|" : "",
                 apc_on && R <= kFloatBitsMaxItems ? "k_scan_ap_c<false, 16, false>" : "k_scan_ap_s<8, 1, false, false, 8, false, false, 1, true>");
    } else if (cache && !tern && K <= 128 && apc_on && (byte128 || (K <= 64 && !packable)) && R <= kFloatBitsMaxItems) {
        snprintf(p2, sizeof(p2), "k_scan_ap_c<false, 8, %s>", byte128 ? "true" : "false");     // all three template arguments: the name a profile prints
    } else if (use_mfma && !cache && r2_shape(K, tern) && ap_r2_mode() != 0 && !packable && R <= kFloatBitsMaxItems) {
        snprintf(p2, sizeof(p2), "k_scan_ap_r2<%d, %d, %d, false>", NML, kAp2Waves, kAp2Groups);
    } else {
        const bool cached = cache && !tern && Wc <= 8;
        // codes of 65 bits and more launch both counter widths, a device word picks one: both names, packed first
        char a32[96] = "";
        if (packable) {
            const int S = cached ? cache_slots(Wc) : S4;
            snprintf(a32, sizeof(a32), "k_scan_ap_s<%d, %d, %s, %s, %d, true, false, %d, %s>|", Wc, cached ? 1 : LW, tern ? "true" : "false", cached ? "false" : "true", S,
                     S == 64 ? waves_for(Wc, tern) : 1, cached ? "true" : "false");
        }
        const int S = cached ? cache_slots(Wc) : S8;
        if (cache && apc_on && K > 128 && K <= 256 && !tern && R <= kFloatBitsMaxItems)      // round 6: float-bit counters on two-byte entries
            snprintf(p2, sizeof(p2), "%sk_scan_ap_c<false, 16, false>", a32);
        else
        snprintf(p2, sizeof(p2), "%sk_scan_ap_s<%d, %d, %s, %s, %d, false, false, %d, %s>", a32, Wc, cached ? 1 : LW, tern ? "true" : "false", cached ? "false" : "true", S,
                 S == 64 ? waves_for(Wc, tern) : 1, cached ? "true" : "false");     // (uncached kernels: the capped form serves both)
    }
    if ((size_t)snprintf(out, out_bytes, "pass1=%s;pass2=%s", p1, p2) >= out_bytes) return xmh::fail(XMH_EINVAL, "xmh_scan_describe: buffer too small");
    return XMH_OK;
}

extern "C" int xmh_hamming_hist(const uint32_t* qbits, const uint32_t* qzero, const uint32_t* qlab,
                                const uint32_t* rbits, const uint32_t* rzero, const uint32_t* rlab, int64_t Q, int64_t R,
                                int K, int C, void* ws, size_t ws_bytes, uint32_t* hist_all, uint32_t* hist_rel,
                                xmh_stream_t stream) {
    XMH_RANGE("xmh_hamming_hist (pass 1)");
    const bool tern = qzero != nullptr;
    xmh_scan_plan p;
    int rc = make_plan(Q, R, K, tern, &p);
    if (rc) return rc;
    rc = check_common("xmh_hamming_hist", qbits, qzero, qlab, rbits, rzero, rlab, C, ws, ws_bytes, p);
    if (rc) return rc;
    const ScanArgs a = make_args(qbits, qzero, qlab, rbits, rzero, rlab, Q, R, K, p);
    const size_t cache_bytes = cache_for_workspace(p, K, tern, ws_bytes);
    const bool mfma_plan = mfma_shape(K, tern);
    const WsLayout L = ws_layout(p, cache_bytes);
    char* base = static_cast<char*>(ws);
    uint32_t* chunk_hist = reinterpret_cast<uint32_t*>(base + L.chunk_hist);
    uint2* below = reinterpret_cast<uint2*>(base + L.below);
    uint2* tot = reinterpret_cast<uint2*>(base + L.tot);
    hipStream_t st = xmh::as_stream(stream);
    const int W = (K + 31) / 32, LW = (C + 31) / 32;
    const bool use_mfma = mfma_plan && LW <= 4 && lane_order_ok(st);
    // the tile tickets of k_scan_below, the nrel_max gate word and the finalize ticket of the evaluation calls on this workspace
    if (!use_mfma) XMH_HIP(hipMemsetAsync(base + L.tick, 0, L.gate + 256 - L.tick, st));   // (the MFMA path clears them in its first launch)
    if (use_mfma) {
        rc = mfma_hist(qbits, qlab, rbits, rlab, Q, R, K, W, LW, p, base, L, chunk_hist,
                       cache_bytes ? reinterpret_cast<uint4*>(base + L.pair_cache) : nullptr, st, qzero, rzero);
        if (rc) return rc;
    }
    auto launch = [&](auto tern_c) {
        constexpr bool T = decltype(tern_c)::value;
        return dispatch_shape<T>(W, LW, [&](auto w, auto l) {
            constexpr int WW = decltype(w)::value, LL = decltype(l)::value;
            constexpr int S = slots_for(WW, T, 4);
            constexpr int NW = S == 64 ? waves_for(WW, T) : 1;
            constexpr bool CAN = kPairCacheShape<WW, T>;           // the shape the pair cache is laid out for
            const size_t lds = (((size_t)p.nbuckets * (64 / S) + 3) & ~(size_t)3) * 4 * NW + aos_ring_bytes(WW, LL, T);
            ScanArgs as = a;
            as.nqt = a.nqt * S / NW;                              // tiles of (64/S) * NW queries
            xmh::ProfScope prof("scan_hist", st);
            if constexpr (CAN) {
                if (cache_bytes) {
                    constexpr int SC = cache_slots(WW);
                    auto kc = k_scan_hist_s<WW, LL, T, SC, 1, true>;
                    const size_t ldsc = (((size_t)p.nbuckets * (64 / SC) + 3) & ~(size_t)3) * 4 + aos_ring_bytes(WW, LL, T);
                    const int r3 = raise_lds(kc, ldsc, "xmh_hamming_hist");
                    if (r3) return r3;
                    as.nqt = a.nqt * SC;
                    as.pair_cache = reinterpret_cast<uint4*>(base + L.pair_cache);
                    hipLaunchKernelGGL(kc, dim3(scan_grid(p) * SC), dim3(64), ldsc, st, as, chunk_hist);
                    return (int)XMH_OK;
                }
            }
            auto kern = k_scan_hist_s<WW, LL, T, S, NW, false>;
            const int r2 = raise_lds(kern, lds, "xmh_hamming_hist");
            if (r2) return r2;
            hipLaunchKernelGGL(kern, dim3(scan_grid(p) * S / NW), dim3(64 * NW), lds, st, as, chunk_hist);
            return (int)XMH_OK;
        });
    };
    if (!use_mfma) {
        rc = tern ? launch(std::true_type{}) : launch(std::false_type{});
        if (rc) return rc;
    }
    XMH_LAUNCH_CHECK("xmh_hamming_hist");
    hipLaunchKernelGGL(k_scan_below, dim3((unsigned)p.nqtile, (unsigned)xmh::ceil_div(p.nbuckets, 4)), dim3(256), 0, st, chunk_hist,
                       (int)p.qpad, (int)p.nbuckets, (int)p.nchunk, use_mfma ? kRelHi16 : 0u, below, tot,
                       reinterpret_cast<uint32_t*>(base + L.tick), (int)Q, reinterpret_cast<uint2*>(base + L.dpre),
                       reinterpret_cast<uint32_t*>(base + L.cap), reinterpret_cast<uint32_t*>(base + L.gate), hist_all, hist_rel);
    XMH_LAUNCH_CHECK("xmh_hamming_hist below");
    return XMH_OK;
}

namespace {
// ---- xmh_scan_verify: re-derive an evaluation with the kernels that do not rely on the lane order (round 6) ----------------------------
// Debug mode, off by default.  While on, every UNSHARDED xmh_hamming_ap / xmh_hamming_map call is followed by a second evaluation of the
// same inputs in a scratch workspace with the masked VALU kernels (no returning-atomic lane order, no hand-scheduled MFMA statement),
// and the per-query AP sums and divisors are compared on the host: divisors bit for bit, sums to float rounding (the two families chunk
// the gallery differently, so pass 2's per-chunk fp32 partial sums add in another order; a wrong rank moves a sum by >= 1 / R relative).
// Synchronises the stream -- a diagnostic for a new device / compiler, not something to leave on.
static std::atomic<int> g_verify{0};
static thread_local bool g_in_verify = false;
constexpr int XMH_EVERIFY_CODE = -74;                 // EBADMSG: the two derivations disagree

int hamming_ap_impl(const uint32_t* qbits, const uint32_t* qzero, const uint32_t* qlab, const uint32_t* rbits,
                    const uint32_t* rzero, const uint32_t* rlab, int64_t Q, int64_t R, int K, int C, void* ws,
                    size_t ws_bytes, const uint32_t* base_all, const uint32_t* base_rel,
                    const uint32_t* nrel_total, int64_t k, double* ap_sum, int32_t* cap, double* map_out, xmh_stream_t stream,
                    const uint32_t* hist_g = nullptr, int world = 0, int rank = 0);

int verify_against_masked(const uint32_t* qbits, const uint32_t* qzero, const uint32_t* qlab, const uint32_t* rbits, const uint32_t* rzero,
                          const uint32_t* rlab, int64_t Q, int64_t R, int K, int C, int64_t k, const double* ap_sum, const int32_t* cap,
                          xmh_stream_t stream) {
    hipStream_t st = xmh::as_stream(stream);
    std::vector<double> a1((size_t)Q), a0((size_t)Q);
    std::vector<int32_t> c1((size_t)Q), c0((size_t)Q);
    XMH_HIP(hipMemcpyAsync(a1.data(), ap_sum, (size_t)Q * 8, hipMemcpyDeviceToHost, st));
    XMH_HIP(hipMemcpyAsync(c1.data(), cap, (size_t)Q * 4, hipMemcpyDeviceToHost, st));
    XMH_HIP(hipStreamSynchronize(st));
    struct Restore {
        int r2; bool fm;
        ~Restore() { g_r2_force = r2; g_force_masked = fm; g_in_verify = false; }
    } restore{g_r2_force, g_force_masked};
    g_in_verify = true;
    g_force_masked = true;
    g_r2_force = 0;                                       // VALU pass 1 too
    xmh_scan_plan p;
    if (const int rc = xmh_scan_plan_make(Q, R, K, qzero != nullptr, &p)) return rc;
    const size_t small = xmh_scan_ws_bytes_nocache(Q, R, K, qzero != nullptr);      // without a pair cache: the re-derivation evaluates the pairs from the codes
    DevBuf ws2, ap2, cap2;
    if (small == 0 || !ws2.alloc(small) || !ap2.alloc((size_t)Q * 8) || !cap2.alloc((size_t)Q * 4))
        return xmh::fail(XMH_ENOMEM, "xmh_scan_verify: no room for the second evaluation (%zu bytes)", small);
    if (const int rc = xmh_hamming_hist(qbits, qzero, qlab, rbits, rzero, rlab, Q, R, K, C, ws2.p, small, nullptr, nullptr, stream)) return rc;
    if (const int rc = hamming_ap_impl(qbits, qzero, qlab, rbits, rzero, rlab, Q, R, K, C, ws2.p, small, nullptr, nullptr, nullptr, k, ap2.as<double>(),
                                       cap2.as<int32_t>(), nullptr, stream))
        return rc;
    XMH_HIP(hipMemcpyAsync(a0.data(), ap2.p, (size_t)Q * 8, hipMemcpyDeviceToHost, st));
    XMH_HIP(hipMemcpyAsync(c0.data(), cap2.p, (size_t)Q * 4, hipMemcpyDeviceToHost, st));
    XMH_HIP(hipStreamSynchronize(st));
    if (const char* fe = getenv("XMH_SCAN_VERIFY_FAULT")) {   // test hook: pretend the fast derivation got one query wrong
        const int64_t at = atoll(fe) % Q;
        a1[(size_t)at] *= 1.0 + 1e-3;
    }
    int64_t bad = -1, nbad = 0;
    for (int64_t i = 0; i < Q; ++i) {
        const bool same = c1[(size_t)i] == c0[(size_t)i] && fabs(a1[(size_t)i] - a0[(size_t)i]) <= 4e-6 * fabs(a0[(size_t)i]) + 1e-9;
        if (!same) { if (bad < 0) bad = i; ++nbad; }
    }
    if (nbad)
        return xmh::fail(XMH_EVERIFY_CODE, "xmh_scan_verify: %lld of %lld queries differ between the lane-order kernels and the masked re-derivation "
                         "(first: query %lld, AP sum %.9g vs %.9g, divisor %d vs %d)", (long long)nbad, (long long)Q, (long long)bad, a1[(size_t)bad], a0[(size_t)bad],
                         c1[(size_t)bad], c0[(size_t)bad]);
    return XMH_OK;
}

int hamming_ap_impl(const uint32_t* qbits, const uint32_t* qzero, const uint32_t* qlab, const uint32_t* rbits,
                    const uint32_t* rzero, const uint32_t* rlab, int64_t Q, int64_t R, int K, int C, void* ws,
                    size_t ws_bytes, const uint32_t* base_all, const uint32_t* base_rel,
                    const uint32_t* nrel_total, int64_t k, double* ap_sum, int32_t* cap, double* map_out, xmh_stream_t stream,
                    const uint32_t* hist_g, int world, int rank) {
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
    const size_t cache_bytes = cache_for_workspace(p, K, tern, ws_bytes);
    const bool mfma_plan = mfma_shape(K, tern);
    const WsLayout L = ws_layout(p, cache_bytes);
    char* base = static_cast<char*>(ws);
    const uint2* below = reinterpret_cast<const uint2*>(base + L.below);
    const uint2* tot = reinterpret_cast<const uint2*>(base + L.tot);
    uint2* dpre = reinterpret_cast<uint2*>(base + L.dpre);
    uint32_t* cap_ws = reinterpret_cast<uint32_t*>(base + L.cap);
    float* ap_part = reinterpret_cast<float*>(base + L.ap_part);
    hipStream_t st = xmh::as_stream(stream);
    uint32_t* nrel_max = reinterpret_cast<uint32_t*>(base + L.gate);
    // packed 32-bit counters apply to a single shard when rank fits rank_bits and the largest relevant count fits the rest
    // Measured at Q 5000 x R 117 218, sparse relevance (the packed width applies), pass 2 packed / 64-bit: K=16 0.275 / 0.251 ms,
    // K=32 0.278 / 0.253, K=64 0.199 / 0.195, K=128 0.274 / 0.286, K=256 0.347 / 0.412 -- packed counters pay from 65 bits on;
    // below that only the 64-bit kernel is launched (and no gated launch returns at once); XMH_SCAN_PACK32_ALL=1 brings the packed
    // kernels back for every length (tests).
    const bool sharded = base_all != nullptr || hist_g != nullptr;
    const bool masked = !lane_order_ok(xmh::as_stream(stream));
    const int rank_bits = sharded || masked ? 0 : packed_rank_bits(K, R);      // (the masked fallback has 64-bit counters only)
    // unsharded: k_scan_below left dpre, nrel and the gate word behind (xmh_hamming_hist).  Sharded: the offsets come from the caller.
    if (hist_g && rank < 0) {                                        // the all-to-all form: this shard's offset rows, by slice owner
        const int S = (int)(p.qpad / world);
        hipLaunchKernelGGL(k_shard_scatter_offsets, dim3((unsigned)xmh::ceil_div(p.qpad, 256), (unsigned)p.nbuckets + 1), dim3(256), 0, st,
                           reinterpret_cast<const uint2*>(hist_g), (int)p.nbuckets, S, (int)Q, (int)p.qpad, k, dpre, cap_ws, cap, nrel_max + 2);
        XMH_LAUNCH_CHECK("xmh_hamming_map_sharded_offsets scatter");
    } else if (hist_g) {                                             // offsets straight from the gathered totals tables of the shards
        hipLaunchKernelGGL(k_shard_offsets_dpre, dim3((unsigned)p.nqtile), dim3(256), 0, st, reinterpret_cast<const uint2*>(hist_g), world, rank,
                           (int)Q, (int)p.qpad, (int)p.nbuckets, k, dpre, cap_ws, cap, nrel_max + 2);
        XMH_LAUNCH_CHECK("xmh_hamming_map_sharded offsets");
    } else if (base_all) {
        hipLaunchKernelGGL(k_scan_dpre, dim3((unsigned)p.nqtile), dim3(256), 0, st, tot, (int)Q, (int)p.qpad, (int)p.nbuckets, base_all,
                           base_rel, nrel_total, k, dpre, cap_ws, cap, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, nrel_max + 2);
        XMH_LAUNCH_CHECK("xmh_hamming_ap dpre");
    }
    const uint32_t kcap = k > 0 && k < (int64_t)0xffffffffll ? (uint32_t)k : 0xffffffffu;
    const int W = (K + 31) / 32, LW = (C + 31) / 32;
    const bool capped = k > 0;

    // k_scan_ap_c (float-bit counters): one-byte pair cache present, 64-bit counters, lane order holds, and the gallery over ALL
    // shards small enough -- known here for an unsharded call; for the totals-table form of the sharded call the offsets kernel leaves
    // the size in a device word and both kernels are launched, each returning at once when it is the other's turn (as for the counter
    // widths); the explicit-offsets form stays on k_scan_ap_s, except on one-byte entries
    // of 65..128-bit codes (which that kernel cannot read), where k_scan_dpre leaves the largest rank of the shard in the word instead.
    // XMH_SCAN_AP_C=0 turns it off.
    const bool apc_on = ap_c_on();
    // 65..128-bit codes whose pass 1 (k_scan_hist_r2w: the MFMA pass 1 ran iff the lane order holds and the labels fit -- the same predicate
    // as in xmh_hamming_hist) left ONE-byte entries: k_scan_ap_c reads them like those of shorter codes, 8 slots x 8 queries wide.  A
    // distance of 128 does not fit a byte and wrapped; pass 1 raised a control word then, k_scan_ap_c returns at once and the kernel that
    // evaluates the pairs from the codes (launched behind it, gated the other way) takes the call.  The cached k_scan_ap_s cannot read
    // these entries.
    const bool byte128 = cache_bytes && mfma_plan && LW <= 4 && !masked && r2w_shape(K, tern);
    const uint32_t* wrapped = byte128 ? (const uint32_t*)(nrel_max + kGateWrapped) : nullptr;
    const size_t cache_s = byte128 ? 0 : cache_bytes;                 // what the k_scan_ap_s launches below may read
    const bool apc = cache_bytes && apc_on && (byte128 || (K <= 64 && rank_bits == 0)) && !tern && !masked && (!base_all || byte128) &&
                     (sharded || R <= kFloatBitsMaxItems);
    const uint32_t* fb_gate = apc && sharded ? (const uint32_t*)(nrel_max + 2) : nullptr;      // (k_scan_ap_r2 sets it too, below)
    if (apc) {
        // one-byte entries are read 8 slots x 8 queries wide (k_scan_ap_c<., 8, HALF>) where the counter rows of 16 queries leave few waves per
        // CU -- 65..128-bit codes (129 rows: 16.5 KB per wave); 16 / 64 / 128 bit, 4 x 16 against 8 x 8: 0.154 / 0.185, 0.171 / 0.182, 0.220 / 0.189 ms
        const bool half = byte128;
        const int SC = half ? 8 : 4;                                     // slots of the geometry pass 2 runs: 64 / SC queries per wave
        ScanArgs as = a;
        as.nqt = a.nqt * SC;
        as.pair_cache = reinterpret_cast<uint4*>(base + L.pair_cache);
        const size_t lds = (size_t)p.nbuckets * (64 / SC) * 8;
        const dim3 grid((unsigned)(scan_grid(p) * SC));
        xmh::ProfScope prof("scan_ap", st);
        auto go = [&](auto kc) {
            const int r3 = raise_lds(kc, lds, "xmh_hamming_ap");
            if (r3) return r3;
            hipLaunchKernelGGL(kc, grid, dim3(64), lds, st, as, below, (const uint2*)dpre, (const uint32_t*)cap_ws, ap_part, fb_gate, kcap, wrapped,
                               (const uint32_t*)nullptr, 0);
            return (int)XMH_OK;
        };
        rc = half ? (capped ? go(k_scan_ap_c<true, 8, true>) : go(k_scan_ap_c<false, 8, true>)) : (capped ? go(k_scan_ap_c<true, 8>) : go(k_scan_ap_c<false, 8>));
        if (rc) return rc;
        XMH_LAUNCH_CHECK("xmh_hamming_ap (float-bit counters)");
    }
    // k_scan_ap_r2: no pair cache, the pairs evaluated again on the MFMA from the packed words (same gating by the size word as k_scan_ap_c)
    const bool apr2 = !cache_bytes && mfma_plan && K <= 64 && r2_shape(K, tern) && ap_r2_mode() != 0 && LW <= 4 && !tern && !masked && !base_all &&
                      rank_bits == 0 && (sharded || R <= kFloatBitsMaxItems);
    if (apr2) {
        fb_gate = sharded ? (const uint32_t*)(nrel_max + 2) : nullptr;
        constexpr int NW2 = kAp2Waves, NQ2 = kAp2Groups;
        const MfmaArgs ma = r2_args(qbits, qlab, rbits, rlab, Q, R, K, W, LW, p, NW2 * NQ2 * 16);
        const dim3 grid((unsigned)(8 * ma.nqt * xmh::ceil_div(p.nchunk, 8)));
        const size_t lds = (size_t)NW2 * NQ2 * p.nbuckets * 16 * 8;
        xmh::ProfScope prof("scan_ap", st);
        auto go = [&](auto kern) {
            const int r3 = raise_lds(kern, lds, "xmh_hamming_ap");
            if (r3) return r3;
            hipLaunchKernelGGL(kern, grid, dim3(64 * NW2), lds, st, ma, below, (const uint2*)dpre, (const uint32_t*)cap_ws, ap_part, fb_gate, kcap);
            return (int)XMH_OK;
        };
        rc = LW <= 2 ? (capped ? go(k_scan_ap_r2<1, NW2, NQ2, true>) : go(k_scan_ap_r2<1, NW2, NQ2, false>))
                     : (capped ? go(k_scan_ap_r2<2, NW2, NQ2, true>) : go(k_scan_ap_r2<2, NW2, NQ2, false>));
        if (rc) return rc;
        XMH_LAUNCH_CHECK("xmh_hamming_ap (MFMA from the packed words)");
    }
    // tb: ternary codes whose pass 1 ran on the MFMA (same predicate as in xmh_hamming_hist) and left two-byte entries: the cached pass-2
    // kernels read nothing but the entries and the tables, so the BINARY 256-bit instances serve (the bucket count comes from the plan)
    const bool tb = tern && cache_bytes && mfma_plan && LW <= 4 && !masked && tbits_shape(K, tern);
    // behind k_scan_ap_c on one-byte entries of 65..128-bit codes only the 64-bit stand-in is launched (always valid; it runs once in a blue moon)
    const int rb = (byte128 && apc) ? 0 : rank_bits;
    // both counter widths are launched; the device word nrel_max (written by k_scan_dpre) lets exactly one of them run
    auto launch = [&](auto tern_c, auto cap_c, auto p32_c, auto masked_c) {
        constexpr bool T = decltype(tern_c)::value;
        constexpr bool CP = decltype(cap_c)::value;
        constexpr bool P32 = decltype(p32_c)::value;
        constexpr bool MK = decltype(masked_c)::value;
        return dispatch_shape<T>(tb ? 8 : W, LW, [&](auto w, auto l) {
            constexpr int WW = decltype(w)::value, LL = decltype(l)::value;
            if constexpr (P32 && MK) return xmh::fail(XMH_ENOTSUP, "xmh_hamming_ap: the masked fallback runs 64-bit counters only");      // (never launched: rank_bits is 0 when masked)
            else {
            constexpr int S = slots_for(WW, T, P32 ? 4 : 8);
            constexpr int NW = S == 64 ? waves_for(WW, T) : 1;
            const size_t cells = (size_t)p.nbuckets * (64 / S);
            const size_t lds = (P32 ? ((cells + 3) & ~(size_t)3) * 4 : ((cells * 2 + 3) & ~(size_t)3) * 4) * NW + aos_ring_bytes(WW, LL, T);
            ScanArgs as = a;
            as.nqt = a.nqt * S / NW;
            xmh::ProfScope prof(P32 ? "scan_ap32" : "scan_ap", st);
            if constexpr (kPairCacheShape<WW, T> && !MK) {
                if (cache_s) {                                        // pass 1 of this call pair left the pairs in the workspace
                    constexpr int SC = cache_slots(WW);
                    auto kc = k_scan_ap_s<WW, 1, T, CP, SC, P32, MK, 1, true>;      // the cached kernel reads no label word: one instance for every class count
                    const size_t cellsc = (size_t)p.nbuckets * (64 / SC);
                    const size_t lds = P32 ? ((cellsc + 3) & ~(size_t)3) * 4 : ((cellsc * 2 + 3) & ~(size_t)3) * 4;   // counters only: no gallery ring
                    const int r3 = raise_lds(kc, lds, "xmh_hamming_ap");
                    if (r3) return r3;
                    as.nqt = a.nqt * SC;
                    as.pair_cache = reinterpret_cast<uint4*>(base + L.pair_cache);
                    hipLaunchKernelGGL(kc, dim3(scan_grid(p) * SC), dim3(64), lds, st, as, below, (const uint2*)dpre, (const uint32_t*)cap_ws,
                                       ap_part, (const uint32_t*)nrel_max, rb, kcap, fb_gate, (const uint32_t*)nullptr);
                    return (int)XMH_OK;
                }
            }
            // the kernels that evaluate the pairs from the codes exist in the capped form only: uncapped, kcap is 0xffffffff and the cap of a
            // query its relevant count, which no ordinal exceeds -- the same credits bit for bit, half the instances (round 5)
            (void)CP;
            auto kern = k_scan_ap_s<WW, LL, T, true, S, P32, MK, NW, false>;
            const int r2 = raise_lds(kern, lds, "xmh_hamming_ap");
            if (r2) return r2;
            // behind a launched k_scan_ap_c (one-byte entries of 65..128-bit codes): the stand-in, gated by the wrap word / the size word
            const bool stand_in = byte128 && apc;
            hipLaunchKernelGGL(kern, dim3(scan_grid(p) * S / NW), dim3(64 * NW), lds, st, as, below, (const uint2*)dpre, (const uint32_t*)cap_ws, ap_part,
                               (const uint32_t*)nrel_max, rb, kcap, stand_in || apr2 ? fb_gate : (const uint32_t*)nullptr, stand_in ? wrapped : (const uint32_t*)nullptr);
            return (int)XMH_OK;
            }
        });
    };
    // Round 6: 129..256-bit binary codes (two-byte entries).  Where the shard is too large or too relevant for packed 32-bit counters --
    // configs[4]: a shard of 1.25 M rows, 150 k relevant items per query -- the 64-bit pass 2 was the integer-counter k_scan_ap_s (15 VALU
    // instructions per pair); the float-bit counters of k_scan_ap_c (9) read the same entries 8 slots x 8 queries wide.  Unsharded calls.
    const bool apc16 = cache_bytes && apc_on && ((K > 128 && K <= 256 && !tern) || tb) && !masked && !sharded && R <= kFloatBitsMaxItems;
    if (apc16) {
        ScanArgs as = a;
        as.nqt = a.nqt * 8;
        as.pair_cache = reinterpret_cast<uint4*>(base + L.pair_cache);
        const size_t lds16 = (size_t)p.nbuckets * 8 * 8;
        xmh::ProfScope prof("scan_ap", st);
        auto go16 = [&](auto kc) {
            const int r3 = raise_lds(kc, lds16, "xmh_hamming_ap");
            if (r3) return r3;
            hipLaunchKernelGGL(kc, dim3((unsigned)(scan_grid(p) * 8)), dim3(64), lds16, st, as, below, (const uint2*)dpre, (const uint32_t*)cap_ws, ap_part,
                               (const uint32_t*)nullptr, kcap, (const uint32_t*)nullptr, (const uint32_t*)nrel_max, rank_bits);
            return (int)XMH_OK;
        };
        rc = capped ? go16(k_scan_ap_c<true, 16>) : go16(k_scan_ap_c<false, 16>);
        if (rc) return rc;
        XMH_LAUNCH_CHECK("xmh_hamming_ap (float-bit counters, two-byte entries)");
    }
    if ((apc || apr2) && !fb_gate && !byte128) {
        // k_scan_ap_c / k_scan_ap_r2 alone takes the call
    } else {
    using T1 = std::true_type;
    using T0 = std::false_type;
    auto launch_width = [&](auto p32_c) {
        auto by_mask = [&](auto tern_c, auto cap_c) { return masked ? launch(tern_c, cap_c, p32_c, T1{}) : launch(tern_c, cap_c, p32_c, T0{}); };
        if (tern && !tb) return capped ? by_mask(T1{}, T1{}) : by_mask(T1{}, T0{});
        return capped ? by_mask(T0{}, T1{}) : by_mask(T0{}, T0{});
    };
    if (rb) {
        rc = launch_width(T1{});
        if (rc) return rc;
        XMH_LAUNCH_CHECK("xmh_hamming_ap packed");
    }
    if (!apc16) {                                                   // (k_scan_ap_c<., 16> above is the 64-bit pass 2 of that shape)
        rc = launch_width(T0{});
        if (rc) return rc;
        XMH_LAUNCH_CHECK("xmh_hamming_ap");
    }
    }
    const unsigned nred = (unsigned)xmh::ceil_div(Q, 64);
    if (map_out && nred <= 4096) {                                   // reduce + mean in one launch (last-ticket block finalises)
        hipLaunchKernelGGL(k_ap_reduce_map, dim3(nred), dim3(256), 0, st, ap_part, (int)Q, (int)p.qpad, (int)p.nchunk, (const uint32_t*)cap_ws, kcap,
                           cap, ap_sum, reinterpret_cast<double*>(base + L.gate + 256), nrel_max + 1, map_out);
        XMH_LAUNCH_CHECK("xmh_hamming_map reduce+finalize");
    } else {
        hipLaunchKernelGGL(k_ap_reduce, dim3(nred), dim3(256), 0, st, ap_part, (int)Q, (int)p.qpad, (int)p.nchunk, ap_sum,
                           sharded ? (const uint32_t*)nullptr : (const uint32_t*)cap_ws, kcap, cap);
        XMH_LAUNCH_CHECK("xmh_hamming_ap reduce");
        if (map_out) {
            hipLaunchKernelGGL(k_map_finalize, dim3(1), dim3(256), 0, st, (const double*)ap_sum, (const int32_t*)cap, Q, map_out);
            XMH_LAUNCH_CHECK("xmh_hamming_map finalize");
        }
    }
    if (g_verify.load(std::memory_order_relaxed) && !g_in_verify && !sharded)
        return verify_against_masked(qbits, qzero, qlab, rbits, rzero, rlab, Q, R, K, C, k, ap_sum, cap, stream);
    return XMH_OK;
}
}  // namespace

extern "C" int xmh_scan_verify(int on) {
    g_verify.store(on != 0 ? 1 : 0, std::memory_order_relaxed);
    return XMH_OK;
}

extern "C" int xmh_hamming_ap(const uint32_t* qbits, const uint32_t* qzero, const uint32_t* qlab, const uint32_t* rbits,
                              const uint32_t* rzero, const uint32_t* rlab, int64_t Q, int64_t R, int K, int C, void* ws,
                              size_t ws_bytes, const uint32_t* base_all, const uint32_t* base_rel,
                              const uint32_t* nrel_total, int64_t k, double* ap_sum, int32_t* cap, xmh_stream_t stream) {
    XMH_RANGE("xmh_hamming_ap (pass 2)");
    return hamming_ap_impl(qbits, qzero, qlab, rbits, rzero, rlab, Q, R, K, C, ws, ws_bytes, base_all, base_rel, nrel_total, k, ap_sum, cap,
                           nullptr, stream);
}

extern "C" int xmh_hamming_map(const uint32_t* qbits, const uint32_t* qzero, const uint32_t* qlab, const uint32_t* rbits,
                               const uint32_t* rzero, const uint32_t* rlab, int64_t Q, int64_t R, int K, int C, void* ws,
                               size_t ws_bytes, int64_t k, double* ap_sum, int32_t* cap, double* map_out, xmh_stream_t stream) {
    XMH_RANGE("xmh_hamming_map (pass 2 + mean)");
    if (!map_out) return xmh::fail(XMH_EINVAL, "xmh_hamming_map: null output");
    return hamming_ap_impl(qbits, qzero, qlab, rbits, rzero, rlab, Q, R, K, C, ws, ws_bytes, nullptr, nullptr, nullptr, k, ap_sum, cap, map_out,
                           stream);
}

extern "C" size_t xmh_scan_totals_offset(int64_t Q, int64_t R, int K, int ternary, size_t* bytes) {
    xmh_scan_plan p;
    if (make_plan(Q, R, K, ternary, &p)) return (size_t)-1;
    const WsLayout L = ws_layout(p, pair_cache_bytes(p, K, ternary != 0));
    if (bytes) *bytes = (size_t)p.nbuckets * p.qpad * 8;
    return L.tot;
}

extern "C" int xmh_hamming_map_sharded(const uint32_t* qbits, const uint32_t* qzero, const uint32_t* qlab, const uint32_t* rbits,
                                       const uint32_t* rzero, const uint32_t* rlab, int64_t Q, int64_t R, int K, int C, void* ws,
                                       size_t ws_bytes, const uint32_t* hist_gathered, int world, int rank, int64_t k, double* ap_sum,
                                       int32_t* cap, double* map_partial, xmh_stream_t stream) {
    XMH_RANGE("xmh_hamming_map_sharded (pass 2)");
    if (!hist_gathered || !map_partial) return xmh::fail(XMH_EINVAL, "xmh_hamming_map_sharded: null pointer");
    if (world <= 0 || rank < 0 || rank >= world) return xmh::fail(XMH_EINVAL, "xmh_hamming_map_sharded: bad world=%d rank=%d", world, rank);
    return hamming_ap_impl(qbits, qzero, qlab, rbits, rzero, rlab, Q, R, K, C, ws, ws_bytes, nullptr, nullptr, nullptr, k, ap_sum, cap, map_partial,
                           stream, hist_gathered, world, rank);
}

extern "C" int xmh_shard_slice_offsets(const uint32_t* totals_slices, int world, int nbuckets, int slice, uint32_t* offsets_out, xmh_stream_t stream) {
    XMH_RANGE("xmh_shard_slice_offsets");
    if (!totals_slices || !offsets_out) return xmh::fail(XMH_EINVAL, "xmh_shard_slice_offsets: null pointer");
    if (world <= 0 || nbuckets <= 0 || slice <= 0) return xmh::fail(XMH_EINVAL, "xmh_shard_slice_offsets: bad arguments (world=%d nb=%d slice=%d)", world, nbuckets, slice);
    hipLaunchKernelGGL(k_shard_slice_offsets, dim3((unsigned)xmh::ceil_div(slice, 64)), dim3(256), 0, xmh::as_stream(stream),
                       reinterpret_cast<const uint2*>(totals_slices), world, nbuckets, slice, reinterpret_cast<uint2*>(offsets_out));
    XMH_LAUNCH_CHECK("xmh_shard_slice_offsets");
    return XMH_OK;
}

extern "C" int xmh_hamming_map_sharded_offsets(const uint32_t* qbits, const uint32_t* qzero, const uint32_t* qlab, const uint32_t* rbits,
                                               const uint32_t* rzero, const uint32_t* rlab, int64_t Q, int64_t R, int K, int C, void* ws,
                                               size_t ws_bytes, const uint32_t* offsets, int world, int64_t k, double* ap_sum, int32_t* cap,
                                               double* map_partial, xmh_stream_t stream) {
    XMH_RANGE("xmh_hamming_map_sharded_offsets (pass 2)");
    if (!offsets || !map_partial) return xmh::fail(XMH_EINVAL, "xmh_hamming_map_sharded_offsets: null pointer");
    xmh_scan_plan p;
    const int rc = make_plan(Q, R, K, qzero != nullptr, &p);
    if (rc) return rc;
    if (world <= 0 || p.qpad % world) return xmh::fail(XMH_EINVAL, "xmh_hamming_map_sharded_offsets: qpad=%lld is not a multiple of world=%d", (long long)p.qpad, world);
    return hamming_ap_impl(qbits, qzero, qlab, rbits, rzero, rlab, Q, R, K, C, ws, ws_bytes, nullptr, nullptr, nullptr, k, ap_sum, cap, map_partial,
                           stream, offsets, world, -1);
}

extern "C" int xmh_map_finalize(const double* ap_sum, const int32_t* cap, int64_t Q, double* map_out, xmh_stream_t stream) {
    XMH_RANGE("xmh_map_finalize");
    if (!ap_sum || !cap || !map_out || Q <= 0) return xmh::fail(XMH_EINVAL, "xmh_map_finalize: bad arguments");
    hipLaunchKernelGGL(k_map_finalize, dim3(1), dim3(256), 0, xmh::as_stream(stream), ap_sum, cap, Q, map_out);
    XMH_LAUNCH_CHECK("xmh_map_finalize");
    return XMH_OK;
}

extern "C" int xmh_shard_offsets(const uint32_t* hist_gathered, int world, int rank, int64_t Q, int nbuckets, uint32_t* base_all,
                                 uint32_t* base_rel, uint32_t* nrel_total, xmh_stream_t stream) {
    XMH_RANGE("xmh_shard_offsets");
    if (!hist_gathered || !base_all || !base_rel || !nrel_total) return xmh::fail(XMH_EINVAL, "xmh_shard_offsets: null pointer");
    if (world <= 0 || rank < 0 || rank >= world || Q <= 0 || nbuckets <= 0 || Q >= (1ll << 24))
        return xmh::fail(XMH_EINVAL, "xmh_shard_offsets: bad arguments (world=%d rank=%d Q=%lld nb=%d)", world, rank, (long long)Q, nbuckets);
    hipLaunchKernelGGL(k_shard_offsets, dim3((unsigned)Q), dim3(64), 0, xmh::as_stream(stream), hist_gathered, world, rank, (int)Q, nbuckets,
                       base_all, base_rel, nrel_total);
    XMH_LAUNCH_CHECK("xmh_shard_offsets");
    return XMH_OK;
}
