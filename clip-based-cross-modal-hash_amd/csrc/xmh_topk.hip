// Exact per-query top-k over bit-packed codes in ONE streaming pass over the gallery shard
// (north_star: "fused bit-packed XOR-popcount + per-query top-k kernel with wavefront reductions and
// coalesced HBM reads over the gallery").  Order = (distance asc, gallery index asc).
//
// Shape: one lane per gallery item (16-byte coalesced loads, next tile prefetched into a second register
// set), up to 8 queries per block held in SGPRs.  Each persistent block owns a CONTIGUOUS range of tiles and
// walks it in index order, keeping for every query a small candidate buffer in LDS plus a bucket histogram
// of the buffer.  t_run = current k-th smallest distance; an item is a candidate only if d < t_run, so after
// the first few tiles almost every tile costs: loads + XOR/popcount + one ballot + ONE barrier.  Candidates
// are appended in index order (wave-local ballot prefix + one cross-wave exchange), which makes "first n
// ties in buffer order" the exact index tie-break -- no sort in the streaming loop.  A second tiny kernel
// merges the per-block lists with the same machinery and bitonic-sorts the final k keys.
//
// Bound: HBM for few queries.  Algorithmic bytes per launch = R*W*4 (gallery read once)
// + Q*W*4 + nblocks*Q*k*6 (partial lists) ; per pair 2W lane-ops.
#include "xmh_common.h"
#include <algorithm>
#include <vector>

#include <stdlib.h>

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kQG = 8;             // queries per block
constexpr uint32_t kInf = 0xFFFFu;

// ---- per-query selection state in LDS -------------------------------------------------------------
struct Sel {
    uint32_t* hist;   // [nb]   counts of appended entries (superset of the live top-k)
    int32_t* bi;      // [cap]  item index (local row / global index)
    uint16_t* bd;     // [cap]  distance
    int* meta;        // [0]=n entries  [1]=t_run  [2]=cnt_lt (entries with d < t_run)
};

struct Shared {
    int* wave_tot;    // [kWaves]
    int* mask;        // [2]
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// k-th smallest bucket of hist (wave 0 only).  Writes meta[1] = t (kInf if fewer than k entries), meta[2] = #entries < t.
__device__ void find_threshold(const Sel& s, int nb, int k) {
    const int lane = lane_id();
    const int per = (nb + 63) / 64;
    const int lo = lane * per;
    const int hi = lo + per < nb ? lo + per : nb;
    int mine = 0;
    for (int d = lo; d < hi; ++d) mine += (int)s.hist[d];
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    const int total = __shfl(incl, 63);
    const int excl = incl - mine;
    if (total < k) {
        if (lane == 0) {
            s.meta[1] = (int)kInf;
            s.meta[2] = total;
        }
        return;
    }
    if (excl < k && k <= incl) {          // exactly one lane
        int run = excl;
        for (int d = lo; d < hi; ++d) {
            const int h = (int)s.hist[d];
            if (run + h >= k) {
                s.meta[1] = d;
                s.meta[2] = run;
                break;
            }
            run += h;
        }
    }
}

// Stable in-place compaction of the buffer to the live top-k: all d < t, then... no: keep ORDER (index order),
// drop entries with d > t and ties at t beyond the first (k - cnt_lt).  All threads; ends with a barrier.
__device__ void compact(const Sel& s, const Shared& sh, int cap, int k) {
    const int n = s.meta[0];
    const int t = s.meta[1];
    const int need = (t == (int)kInf) ? 0x7fffffff : k - s.meta[2];   // t == kInf: fewer than k entries, d < t keeps all
    const int per = (cap + kThreads - 1) / kThreads;     // contiguous segment per thread
    const int lo = threadIdx.x * per;
    const int hi = (lo + per < n) ? lo + per : n;
    // pass 1: ties per thread -> ordered block prefix
    int my_ties = 0;
    for (int p = lo; p < hi; ++p) my_ties += ((int)s.bd[p] == t);
    // block exclusive scan of my_ties (wave scan + wave totals)
    const int lane = lane_id(), w = wave_id();
    int incl = my_ties;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) sh.wave_tot[w] = incl;
    __syncthreads();
    int tie_before = incl - my_ties;
    for (int x = 0; x < w; ++x) tie_before += sh.wave_tot[x];
    __syncthreads();
    // pass 2: read the whole segment into registers (static indices -> VGPRs, not scratch) with a keep mask,
    // ordered prefix of the keep counts, then write back: positions only move down, and nobody writes before
    // everybody has read.
    constexpr int kMaxSeg = 24;
    int32_t ri[kMaxSeg];
    uint16_t rd[kMaxSeg];
    uint32_t keepm = 0;
    {
        int tr = tie_before;
#pragma unroll
        for (int u = 0; u < kMaxSeg; ++u) {
            const int p = lo + u;
            if (p < hi) {
                const int d = (int)s.bd[p];
                ri[u] = s.bi[p];
                rd[u] = (uint16_t)d;
                bool ok = d < t;
                if (d == t) {
                    ok = tr < need;
                    ++tr;
                }
                if (ok) keepm |= 1u << u;
            }
        }
    }
    const int keep = __popc(keepm);
    incl = keep;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) sh.wave_tot[w] = incl;
    __syncthreads();
    int pos = incl - keep;
    int total = 0;
    for (int x = 0; x < kWaves; ++x) {
        if (x < w) pos += sh.wave_tot[x];
        total += sh.wave_tot[x];
    }
#pragma unroll
    for (int u = 0; u < kMaxSeg; ++u) {
        if (keepm & (1u << u)) {
            s.bi[pos] = ri[u];
            s.bd[pos] = rd[u];
            ++pos;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) s.meta[0] = total;
    __syncthreads();
}

// Feed one tile (IPT items per lane, lane-strided inside the wave's contiguous sub-range, so that order is
// (wave, j, lane)) into the selection state of one query.  Called uniformly by all threads.
template <int IPT>
__device__ void feed_tile(const Sel& s, const Shared& sh, const int (&d)[IPT], const int32_t (&item)[IPT], int nb, int k,
                          int cap, int tile_items) {
    const int lane = lane_id(), w = wave_id();
    const int t_old = s.meta[1];
    // a. histogram of candidates
#pragma unroll
    for (int j = 0; j < IPT; ++j)
        if (d[j] < t_old) atomicAdd(&s.hist[d[j]], 1u);
    __syncthreads();
    // b. new threshold
    if (w == 0) find_threshold(s, nb, k);
    __syncthreads();
    const int t_new = s.meta[1];
    const int n0 = s.meta[0];
    // c. ordered append of candidates with d <= t_new
    int wcnt = 0;
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const bool acc = d[j] < t_old && d[j] <= t_new;
        wcnt += __popcll(__ballot(acc));
    }
    if (lane == 0) sh.wave_tot[w] = wcnt;
    __syncthreads();
    int off = n0, total = 0;
    for (int x = 0; x < kWaves; ++x) {
        if (x < w) off += sh.wave_tot[x];
        total += sh.wave_tot[x];
    }
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const bool acc = d[j] < t_old && d[j] <= t_new;
        const unsigned long long m = __ballot(acc);
        if (acc) {
            const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
            s.bi[pos] = item[j];
            s.bd[pos] = (uint16_t)d[j];
        }
        off += __popcll(m);
    }
    __syncthreads();
    if (threadIdx.x == 0) s.meta[0] = n0 + total;
    __syncthreads();
    if (cap - (n0 + total) < tile_items) compact(s, sh, cap, k);
}

struct Layout {
    int nb, cap, nq;        // buckets, candidate slots per query, query slots (kQG in the stream kernel; the merge kernel keeps one
                            // query and sorts in the slots behind it: 3 slots give the 8 KB the 1024-key sort needs)
    __host__ __device__ size_t hist_off(int q) const { return (size_t)q * nb * 4; }
    __host__ __device__ size_t bi_off(int q) const { return (size_t)nq * nb * 4 + (size_t)q * cap * 4; }
    __host__ __device__ size_t bd_off(int q) const { return (size_t)nq * nb * 4 + (size_t)nq * cap * 4 + (size_t)q * cap * 2; }
    __host__ __device__ size_t meta_off() const {
        size_t o = (size_t)nq * nb * 4 + (size_t)nq * cap * 6;
        return (o + 15) & ~(size_t)15;
    }
    __host__ __device__ size_t bytes() const { return meta_off() + (nq * 4 + kWaves + 2 + 2) * 4; }
};

__device__ __forceinline__ Sel sel_of(char* smem, const Layout& L, int q) {
    Sel s;
    s.hist = reinterpret_cast<uint32_t*>(smem + L.hist_off(q));
    s.bi = reinterpret_cast<int32_t*>(smem + L.bi_off(q));
    s.bd = reinterpret_cast<uint16_t*>(smem + L.bd_off(q));
    s.meta = reinterpret_cast<int*>(smem + L.meta_off()) + q * 4;
    return s;
}
__device__ __forceinline__ Shared shared_of(char* smem, const Layout& L) {
    Shared sh;
    int* base = reinterpret_cast<int*>(smem + L.meta_off()) + L.nq * 4;
    sh.wave_tot = base;
    sh.mask = base + kWaves;
    return sh;
}

__device__ void init_state(char* smem, const Layout& L) {
    for (int q = 0; q < L.nq; ++q) {
        Sel s = sel_of(smem, L, q);
        for (int d = threadIdx.x; d < L.nb; d += kThreads) s.hist[d] = 0u;
        if (threadIdx.x == 0) {
            s.meta[0] = 0;
            s.meta[1] = (int)kInf;
            s.meta[2] = 0;
        }
    }
    Shared sh = shared_of(smem, L);
    if (threadIdx.x < 2) sh.mask[threadIdx.x] = 0;
    __syncthreads();
}

template <int W>
struct Rec {
    uint32_t w[W];
};

template <int W>
__device__ __forceinline__ void load_rec(Rec<W>& r, const uint32_t* __restrict__ base, int64_t item, bool ok) {
    if (!ok) {
#pragma unroll
        for (int x = 0; x < W; ++x) r.w[x] = 0u;
        return;
    }
    const uint32_t* p = base + item * W;
    if constexpr (W % 4 == 0) {
#pragma unroll
        for (int x = 0; x < W / 4; ++x) {
            const uint4 v = reinterpret_cast<const uint4*>(p)[x];
            r.w[4 * x] = v.x; r.w[4 * x + 1] = v.y; r.w[4 * x + 2] = v.z; r.w[4 * x + 3] = v.w;
        }
    } else if constexpr (W == 2) {
        const uint2 v = *reinterpret_cast<const uint2*>(p);
        r.w[0] = v.x; r.w[1] = v.y;
    } else {
#pragma unroll
        for (int x = 0; x < W; ++x) r.w[x] = p[x];
    }
}

// ---- streaming kernel -----------------------------------------------------------------------------
// TERN (round 6): codes with exact zeros (sign(0) = 0, reference runners/base.py:407-410) carry a second plane (bit set <=> element is 0,
// padding bits set); the distance is in HALF units, 2 d = K - q.r = #(positions where either side is 0) + 2 #(both live and different)
// in [0, 2K] (nb = 2K + 1 buckets), `pad` = 32 W - K removes the padding bits from the first count.
template <int W>
__device__ __forceinline__ int dist2_words(const uint32_t (&rb)[W], const uint32_t (&rz)[W], const uint32_t* __restrict__ qb,
                                           const uint32_t* __restrict__ qz, int pad) {
    int dead = 0, diff = 0;
#pragma unroll
    for (int x = 0; x < W; ++x) {
        const uint32_t z = rz[x] | qz[x];
        dead += __popc(z);
        diff += __popc((rb[x] ^ qb[x]) & ~z);
    }
    return dead - pad + 2 * diff;
}

template <int W, int IPT, bool TERN>
__global__ __launch_bounds__(kThreads) void k_topk_stream(const uint32_t* __restrict__ qbits, const uint32_t* __restrict__ qzero,
                                                          const uint32_t* __restrict__ rbits, const uint32_t* __restrict__ rzero,
                                                          int pad, int Q, int64_t R, int k,
                                                          Layout L, int tiles_per_block, int nblocks,
                                                          uint16_t* __restrict__ part_d, int32_t* __restrict__ part_i,
                                                          const int* __restrict__ gate) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = kThreads * IPT;
    const int lane = lane_id(), w = wave_id();
    const int q0 = blockIdx.y * kQG;
    const int nq = (Q - q0 < kQG) ? Q - q0 : kQG;
    if (gate) {                                           // round 5: one flag per query -- a group whose queries all got their exact lists
        int any = 0;                                      // from the fast path returns, the others recompute (the merge takes only the
        for (int q = 0; q < nq; ++q) any |= gate[q0 + q]; // failed queries' lists: a failure costs its group's share, not the whole call)
        if (!any) return;
    }
    init_state(smem, L);
    Shared sh = shared_of(smem, L);

    const int64_t tile0 = (int64_t)blockIdx.x * tiles_per_block;
    const int64_t ntiles_all = (R + TILE - 1) / TILE;
    const int64_t tile1 = (tile0 + tiles_per_block < ntiles_all) ? tile0 + tiles_per_block : ntiles_all;

    auto item_of = [&](int64_t tile, int j) -> int64_t { return tile * TILE + (int64_t)w * (64 * IPT) + j * 64 + lane; };

    Rec<W> cur[IPT], nxt[IPT];
    Rec<W> curz[TERN ? IPT : 1], nxtz[TERN ? IPT : 1];
    if (tile0 < tile1) {
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const int64_t it = item_of(tile0, j);
            load_rec<W>(cur[j], rbits, it, it < R);
            if constexpr (TERN) load_rec<W>(curz[j], rzero, it, it < R);
        }
    }
    for (int64_t tile = tile0; tile < tile1; ++tile) {
        const bool more = tile + 1 < tile1;
        if (more) {
#pragma unroll
            for (int j = 0; j < IPT; ++j) {
                const int64_t it = item_of(tile + 1, j);
                load_rec<W>(nxt[j], rbits, it, it < R);
                if constexpr (TERN) load_rec<W>(nxtz[j], rzero, it, it < R);
            }
        }
        int d[kQG][IPT];
        int32_t item[IPT];
        int mask = 0;
#pragma unroll
        for (int j = 0; j < IPT; ++j) item[j] = (int32_t)item_of(tile, j);
#pragma unroll
        for (int q = 0; q < kQG; ++q) {
            if (q < nq) {
                const uint32_t* __restrict__ qw = qbits + (int64_t)(q0 + q) * W;     // uniform -> SGPRs
                const int t_run = sel_of(smem, L, q).meta[1];
                bool any = false;
#pragma unroll
                for (int j = 0; j < IPT; ++j) {
                    int acc = 0;
                    if constexpr (TERN) acc = dist2_words<W>(cur[j].w, curz[j].w, qw, qzero + (int64_t)(q0 + q) * W, pad);
                    else {
#pragma unroll
                        for (int x = 0; x < W; ++x) acc += __popc(cur[j].w[x] ^ qw[x]);
                    }
                    d[q][j] = ((int64_t)item[j] < R && item[j] >= 0) ? acc : (int)kInf;
                    any |= d[q][j] < t_run;
                }
                if (__ballot(any)) mask |= 1 << q;
            }
        }
        if (__syncthreads_or(mask)) {                      // rare after warm-up: somebody has a candidate
            if (threadIdx.x == 0) sh.mask[0] = 0;
            __syncthreads();
            if (mask && lane == 0) atomicOr(&sh.mask[0], mask);
            __syncthreads();
            const int m = sh.mask[0];
#pragma unroll
            for (int q = 0; q < kQG; ++q) {
                if (m & (1 << q)) feed_tile<IPT>(sel_of(smem, L, q), sh, d[q], item, L.nb, k, L.cap, TILE);
            }
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < IPT; ++j) {
                cur[j] = nxt[j];
                if constexpr (TERN) curz[j] = nxtz[j];
            }
        }
    }
    __syncthreads();
    // final: live top-k of this block's range, in index order
    for (int q = 0; q < nq; ++q) {
        Sel s = sel_of(smem, L, q);
        compact(s, sh, L.cap, k);
        const int n = s.meta[0];
        uint16_t* od = part_d + ((int64_t)(q0 + q) * nblocks + blockIdx.x) * k;
        int32_t* oi = part_i + ((int64_t)(q0 + q) * nblocks + blockIdx.x) * k;
        for (int p = threadIdx.x; p < k; p += kThreads) {
            od[p] = p < n ? s.bd[p] : (uint16_t)kInf;
            oi[p] = p < n ? s.bi[p] : -1;
        }
    }
}

// ---- merge kernel: one block per query, streams [nblocks][k] partial lists (already in index order) ----
template <int IPT>
__global__ __launch_bounds__(kThreads) void k_topk_merge(const uint16_t* __restrict__ part_d, const int32_t* __restrict__ part_i,
                                                         int nblocks, int k, Layout L, int64_t base_index,
                                                         uint16_t* __restrict__ out_d, int32_t* __restrict__ out_i,
                                                         const int* __restrict__ gate) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (gate && gate[blockIdx.x] == 0) return;            // this query's list came out of the fast path
    constexpr int TILE = kThreads * IPT;
    const int lane = lane_id(), w = wave_id();
    const int q = blockIdx.x;
    init_state(smem, L);
    Shared sh = shared_of(smem, L);
    Sel s = sel_of(smem, L, 0);
    const int64_t n_in = (int64_t)nblocks * k;
    const uint16_t* pd = part_d + (int64_t)q * n_in;
    const int32_t* pi = part_i + (int64_t)q * n_in;
    for (int64_t base = 0; base < n_in; base += TILE) {
        int d[IPT];
        int32_t item[IPT];
        bool any = false;
        const int t_run = s.meta[1];
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const int64_t e = base + (int64_t)w * (64 * IPT) + j * 64 + lane;
            int32_t ii = -1;
            int dd = (int)kInf;
            if (e < n_in) {
                ii = pi[e];
                dd = ii >= 0 ? (int)pd[e] : (int)kInf;
            }
            d[j] = dd;
            item[j] = ii;
            any |= dd < t_run;
        }
        if (__syncthreads_or(any ? 1 : 0)) feed_tile<IPT>(s, sh, d, item, L.nb, k, L.cap, TILE);
    }
    __syncthreads();
    compact(s, sh, L.cap, k);
    // sort the <= k survivors by (distance, index): bitonic on 64-bit keys in the (now free) tail of the buffer
    const int n = s.meta[0];
    int P = 1;
    while (P < k) P <<= 1;
    // keys live in the (unused) buffer slots of queries 1.. of the layout: (nq-1)*cap*4 B >= 8 KB = 1024 keys
    unsigned long long* key = reinterpret_cast<unsigned long long*>(smem + L.bi_off(1));
    for (int p = threadIdx.x; p < P; p += kThreads)
        key[p] = p < n ? (((unsigned long long)s.bd[p] << 32) | (unsigned int)s.bi[p]) : ~0ull;
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int p = threadIdx.x; p < P / 2; p += kThreads) {
                const int i = 2 * p - (p & (stride - 1));
                const int j2 = i + stride;
                const bool up = ((i & size) == 0);
                const unsigned long long a = key[i], b = key[j2];
                if ((a > b) == up) {
                    key[i] = b;
                    key[j2] = a;
                }
            }
            __syncthreads();
        }
    }
    for (int p = threadIdx.x; p < k; p += kThreads) {
        const unsigned long long v = key[p];
        const bool ok = v != ~0ull;
        out_d[(int64_t)q * k + p] = ok ? (uint16_t)(v >> 32) : (uint16_t)kInf;
        out_i[(int64_t)q * k + p] = ok ? (int32_t)(base_index + (int64_t)(uint32_t)v) : -1;
    }
}

// ===================================================================================================
// Fast path: sample -> per-query distance threshold -> ONE streaming filter pass -> exact select.
// The filter kernel has no LDS state and no barrier: lanes own gallery items (16-byte coalesced loads,
// next tile prefetched), queries sit in SGPRs, and an item is appended to its query's global candidate
// list only if d <= t_est[q] (a few hundred items out of millions).  k_topk_select then sorts the
// candidates of a query by the 64-bit key (distance, index) and emits the first k.
// Exactness is verified, not assumed: if a list overflowed or holds fewer than k items the select kernel
// raises `fail`, and the robust streaming kernels above (gated on that flag) recompute the call.
// ===================================================================================================
// candidate counters live kCntStride words apart: the appends of ALL blocks are device-scope atomics on these few words, and
// counters that share a cache line share one memory channel (64 queries on two lines: 45 us of a 160 us pass at Q = 64)
constexpr int kCntStride = 64;
constexpr int kCandCap = 8192;        // candidates kept per query (keys of 8 B)
// Round 5: a query's list is kSub sub-lists of kSubCap keys, each with its own counter (kCntStride words apart like the queries').  The
// filters flush their staged candidates when a wave ends, i.e. all at about the same time, and one counter per query serialised those
// atomics in the L2: 10 M x 256 bit with 380 candidates per launch lost ~1 us to it, 40 M x 64 bit with 1 600 lost 7 us of 54.  A wave
// takes the sub-list (its number + its flush count) mod kSub, so a run of equal codes that one wave meets still spreads.
constexpr int kSub = 8;
constexpr int kSubCap = kCandCap / kSub;
constexpr int kSampleBlocks = 256;
constexpr int kSamplePerBlock = 1024;
constexpr int kFoldPickQ = 16;        // up to this many queries the last sample block picks the thresholds (no pick launch)

struct FastWs {
    uint32_t* hist;            // [Q][nb]  sample histogram
    uint32_t* t_est;           // [Q]
    uint32_t* bound;           // [Q]      index bound of the threshold bucket (index_bound)
    uint32_t* cnt;             // [Q]      candidates appended
    int* fail;                 // [Q]: the fast path could not give this query its exact list
    unsigned long long* cand;  // [Q][kCandCap]
};

template <int W>
__device__ __forceinline__ int dist_words(const Rec<W>& r, const uint32_t* __restrict__ qw) {
    int acc = 0;
#pragma unroll
    for (int x = 0; x < W; ++x) acc += __popc(r.w[x] ^ qw[x]);
    return acc;
}

// t_est = smallest distance whose sampled cumulative count reaches `target` (nb-1 if it never does), by one wave: lanes take
// 64 consecutive buckets, wave prefix sum, first lane over the target wins (a thread per query walking the buckets one
// dependent load at a time took 13 us -- a quarter of the Q=1 filter pass).  The row is left ZEROED for the next call on this
// workspace.  COHERENT: the counts were added by other blocks of the SAME launch (agent-scope loads).
// below / at (round 5): the sampled count strictly below the bucket taken and the count in it (0 / 0 if the target was never reached).
template <bool COHERENT>
__device__ __forceinline__ int pick_row(uint32_t* __restrict__ row, int nb, uint32_t target, int lane, uint32_t* below, uint32_t* at) {
    uint32_t carry = 0;
    int t = nb - 1;
    bool found = false;
    *below = 0u;
    *at = 0u;
    for (int base = 0; base < nb; base += 512) {                    // 8 segments of 64 buckets per round, their loads issued together
        uint32_t v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = base + j * 64 + lane;
            v[j] = 0u;
            if (d < nb) v[j] = COHERENT ? __hip_atomic_load(row + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : row[d];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = base + j * 64 + lane;
            if (d < nb) row[d] = 0u;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d0 = base + j * 64;
            if (found || d0 >= nb) continue;
            uint32_t s = v[j];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t u = __shfl_up(s, o, 64);
                if (lane >= o) s += u;
            }
            const unsigned long long over = __ballot(d0 + lane < nb && carry + s >= target);
            if (over) {
                const int win = __ffsll((long long)over) - 1;
                t = d0 + win;
                found = true;
                *at = (uint32_t)__shfl((int)v[j], win, 64);
                *below = carry + (uint32_t)__shfl((int)s, win, 64) - *at;
            }
            carry += __shfl(s, 63, 64);
        }
    }
    return t;
}

// Round 5: an INDEX BOUND for the threshold bucket.  Candidates = every item below the threshold t plus the items AT t whose index is
// below bound[q].  Exact for any bound: a non-candidate has d > t, or d == t with an index not below the bound -- it follows every
// candidate in (distance, index) order, so whenever the list holds k keys the k smallest of them are the k smallest of the gallery (the
// select checks the count as before).  The bound only decides how often that holds: it is set from the sample so that the bucket's
// share below it covers what the buckets under t are expected to leave open, with 3-sigma / 2-sigma lower bounds on both sampled
// counts and a factor 2.  Fine buckets (128 bits and more) get bound = R, i.e. nothing changes; for coarse codes on large galleries --
// 16 bit over 40 M rows: 610 items at distance 0, 10 400 within 1 -- the list no longer overflows when the pick goes one bucket further.
struct PickParams { float inv_frac; uint32_t k, R; int exact; };
__device__ __forceinline__ uint32_t index_bound(uint32_t below, uint32_t at, PickParams pp) {
    if (pp.exact || at == 0u) return pp.R;                        // exact counts (small gallery) or no estimate: no bound
    const float b = (float)below, a = (float)at;
    const float below_lb = fmaxf(0.0f, b - 3.0f * sqrtf(b)) * pp.inv_frac;       // items strictly below t, at least
    const float need = fmaxf((float)pp.k - below_lb, 0.0f) + 8.0f;                 // wanted from the bucket t
    const float at_lb = fmaxf(1.0f, a - 2.0f * sqrtf(a)) * pp.inv_frac;          // items in the bucket t, at least
    const float rows = 2.0f * need * (float)pp.R / at_lb;
    return rows >= (float)pp.R ? pp.R : (uint32_t)rows;
}

// Control words at the head of the fast-path workspace.  Contract (xmh_topk_ws_init / xmh_hamming_topk_prepared): zero on entry,
// zero again on exit, like the sample histogram -- every kernel that consumes one of them puts it back.
struct TopkCtl {
    uint32_t sample_ticket;
    uint32_t filter_ticket;
    uint32_t robust_ticket;
};

// sample histogram: block b reads kSamplePerBlock consecutive rows starting at b*stride (whole gallery if small).
// FOLD (few queries): the block that finishes last (ticket) also picks the thresholds and resets the per-call state, so the
// call needs neither a memset nor a pick launch.
template <int W, bool TERN>
__global__ __launch_bounds__(kThreads) void k_topk_sample(const uint32_t* __restrict__ qbits, const uint32_t* __restrict__ qzero,
                                                          const uint32_t* __restrict__ rbits, const uint32_t* __restrict__ rzero, int pad,
                                                          int Q, int64_t R, int nb, int64_t stride, int per_block,
                                                          uint32_t* __restrict__ hist, int fold, uint32_t target,
                                                          TopkCtl* __restrict__ ctl, uint32_t* __restrict__ t_est,
                                                          uint32_t* __restrict__ cnt, int* __restrict__ fail, PickParams pp, uint32_t* __restrict__ bound,
                                                          const int QG) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sh[];     // [QG][nb]: QG = 16 queries per round unless the histograms of long ternary codes (2K + 1 buckets) leave room for fewer
    __shared__ int last;
    const int64_t lo = (int64_t)blockIdx.x * stride;
    const int64_t hi = (lo + per_block < R) ? lo + per_block : R;
    // many queries (no fold): the groups of 16 queries are spread over blockIdx.y -- 64 queries in one block were 52 us of a 200 us call
    for (int q0 = blockIdx.y * QG; q0 < Q; q0 += QG * gridDim.y) {
        const int nq = (Q - q0 < QG) ? Q - q0 : QG;
        for (int e = threadIdx.x; e < nq * nb; e += kThreads) sh[e] = 0u;
        __syncthreads();
        constexpr int NB = W <= 8 ? 4 : (W <= 16 ? 2 : 1);          // records in flight per thread (one dependent miss per record otherwise)
        for (int64_t it0 = lo + threadIdx.x; it0 < hi; it0 += (int64_t)NB * kThreads) {
            Rec<W> r[NB];
            Rec<W> rz[TERN ? NB : 1];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int64_t it = it0 + (int64_t)j * kThreads;
                load_rec<W>(r[j], rbits, it < hi ? it : lo, true);
                if constexpr (TERN) load_rec<W>(rz[j], rzero, it < hi ? it : lo, true);
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (it0 + (int64_t)j * kThreads < hi)
                    for (int q = 0; q < nq; ++q) {
                        int d;
                        if constexpr (TERN) d = dist2_words<W>(r[j].w, rz[j].w, qbits + (int64_t)(q0 + q) * W, qzero + (int64_t)(q0 + q) * W, pad);
                        else d = dist_words<W>(r[j], qbits + (int64_t)(q0 + q) * W);
                        atomicAdd(&sh[q * nb + d], 1u);
                    }
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < nq * nb; e += kThreads)
            if (sh[e]) atomicAdd(&hist[(int64_t)q0 * nb + e], sh[e]);
        __syncthreads();
    }
    if (!fold) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // this block's adds have been performed
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = __hip_atomic_fetch_add(&ctl->sample_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = t == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    const int lane = lane_id();
    for (int q = wave_id(); q < Q; q += kWaves) {
        uint32_t below, at;
        const int t = pick_row<true>(hist + (int64_t)q * nb, nb, target, lane, &below, &at);
        if (lane == 0) {
            t_est[q] = (uint32_t)t;
            bound[q] = index_bound(below, at, pp);
            fail[q] = 0;
        }
        if (lane < kSub) cnt[((int64_t)q * kSub + lane) * kCntStride] = 0u;
    }
    if (threadIdx.x == 0) __hip_atomic_store(&ctl->sample_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// many queries: one wave per query after the sample launch; also resets the per-call state (candidate counts, fail flag)
__global__ __launch_bounds__(64) void k_topk_pick(uint32_t* __restrict__ hist, int Q, int nb, uint32_t target,
                                                  uint32_t* __restrict__ t_est, uint32_t* __restrict__ cnt, int* __restrict__ fail, PickParams pp,
                                                  uint32_t* __restrict__ bound) {
    const int q = blockIdx.x, lane = threadIdx.x;
    if (q >= Q) return;
    uint32_t below, at;
    const int t = pick_row<false>(hist + (int64_t)q * nb, nb, target, lane, &below, &at);
    if (lane == 0) {
        t_est[q] = (uint32_t)t;
        bound[q] = index_bound(below, at, pp);
        fail[q] = 0;
    }
    if (lane < kSub) cnt[((int64_t)q * kSub + lane) * kCntStride] = 0u;
}

__device__ __forceinline__ int sub_of_wave(int turn) { return (int)((blockIdx.x * (kThreads / 64) + wave_id() + turn) & (kSub - 1)); }

__device__ __forceinline__ void append_one(int q, uint32_t d, uint32_t it, uint32_t* __restrict__ cnt, unsigned long long* __restrict__ cand, int sub) {
    const uint32_t pos = atomicAdd(cnt + ((int64_t)q * kSub + sub) * kCntStride, 1u);
    if (pos < (uint32_t)kSubCap) cand[(int64_t)q * kCandCap + sub * kSubCap + pos] = ((unsigned long long)d << 32) | it;
}

// the wave's staged candidates -> the per-query lists, 64 per round trip; *count (the wave's own LDS word) goes back to zero
__device__ __noinline__ void flush_staged(const uint2* stage, uint32_t* count, int cap, int q0, uint32_t* __restrict__ cnt,
                                          unsigned long long* __restrict__ cand) {
    __builtin_amdgcn_wave_barrier();
    const int lane = lane_id();
    int n = (int)*count;
    n = __builtin_amdgcn_readfirstlane(n < cap ? n : cap);      // entries past the capacity went out directly
    for (int b = 0; b < n; b += 64) {
        const int sub = sub_of_wave(b >> 6);
        const bool have = b + lane < n;
        const uint2 e = have ? stage[b + lane] : make_uint2(0u, 0u);
        const int ql = have ? (int)(e.y >> 16) : -1;
        // one global atomic per DISTINCT query of the batch (runs of equal codes in the gallery put dozens of candidates of one query
        // into a batch: the duplicate-heavy gallery of the bench went 0.113 -> 0.066 ms per pass at Q = 8 with this), at most 4 rounds, the rest one by one
        unsigned long long todo = __ballot(have);
        for (int round = 0; round < 4 && todo; ++round) {
            const int lead = __ffsll((long long)todo) - 1;
            const int qcur = __shfl(ql, lead);
            const unsigned long long m = __ballot(ql == qcur) & todo;
            uint32_t base = 0;
            if (lane == lead) base = atomicAdd(cnt + ((int64_t)(q0 + qcur) * kSub + sub) * kCntStride, (uint32_t)__popcll(m));
            base = (uint32_t)__shfl((int)base, lead);
            if (have && ql == qcur) {
                const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (pos < (uint32_t)kSubCap) cand[(int64_t)(q0 + qcur) * kCandCap + sub * kSubCap + pos] = ((unsigned long long)(e.y & 0xffffu) << 32) | e.x;
            }
            todo &= ~m;
        }
        if ((todo >> lane) & 1ull) append_one(q0 + ql, e.y & 0xffffu, e.x, cnt, cand, sub);
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) *count = 0;
    __builtin_amdgcn_wave_barrier();
}

// Candidates of the VALU filters go through the same wave-private LDS list as the matrix-core filter's (round 5).  The direct append --
// a global atomic whose return the wave waits for -- shares vmcnt with the prefetched tile: every candidate held its wave for a whole
// memory round trip with nothing of its own in flight behind it.  Measured on 10 M x 256 bit, per-piece filter, 512 blocks: 45.6 us
// with ~1 candidate per launch, 49.6 with 380, 58.9 with 2 900; staged: 46.3 with 380.
constexpr int kStageV = 128;                                // entries per wave; flushed from 64 on, the overflow goes out directly
__device__ __forceinline__ void stage_candidate(bool hit, uint32_t item, uint32_t d, int ql, uint2* stage, uint32_t* count, int q0,
                                                uint32_t* __restrict__ cnt, unsigned long long* __restrict__ cand) {
    if (hit) {
        const uint32_t pos = atomicAdd(count, 1u);
        if (pos < (uint32_t)kStageV) stage[pos] = make_uint2(item, d | ((uint32_t)ql << 16));
        else append_one(q0 + ql, d, item, cnt, cand, sub_of_wave((int)(pos >> 6)));
    }
}

// ---- the same filter for codes of whole 128-bit pieces (W % 4 == 0), round 5 ---------------------------------------------------
// k_topk_filter gives every lane an ITEM: at 256 bits two 16-byte loads per lane, 32 bytes apart between neighbouring lanes, so each
// load instruction of a wave touches 2 KB and uses half of it.  Here every lane takes a 16-byte PIECE and a wave's load instruction
// covers 1 KB contiguous; the W / 4 lanes of an item add their partial distances with DPP moves (no LDS), every lane of the group ends
// with the whole distance and its first lane reports.  On such loads the non-temporal hint pays (on the per-item form it costs):
// measured on 10 M x 256 bit, four galleries in rotation so that the Infinity Cache cannot help (tools/proto_stream_read.hip):
// per-item loads 6.1-6.5 TB/s, per-piece 6.1-6.3, per-piece + nt 6.7-7.0 = 0.84-0.87 of the 8 TB/s peak; 96 / 192 MB galleries (cache
// resident) 6.5 / 6.9 -> 7.4 / 7.2.  Two blocks per CU were best or within 2 % of it at every size.
// The query words a lane needs are those of ITS piece: 4 registers per query instead of W.
typedef uint32_t topk_u4 __attribute__((ext_vector_type(4)));

template <int LPI>
__device__ __forceinline__ int join_pieces(int h) {
    if constexpr (LPI >= 2) h += __builtin_amdgcn_mov_dpp(h, 0xB1, 0xf, 0xf, true);     // quad_perm [1,0,3,2]
    if constexpr (LPI >= 4) h += __builtin_amdgcn_mov_dpp(h, 0x4E, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
    if constexpr (LPI >= 8) h += __builtin_amdgcn_mov_dpp(h, 0x141, 0xf, 0xf, true);    // row_half_mirror: lane i <-> 7 - i, the other quad's sum
    if constexpr (LPI >= 16) h += __builtin_amdgcn_mov_dpp(h, 0x140, 0xf, 0xf, true);   // row_mirror: lane i <-> 15 - i, the other half's sum
    return h;
}

template <int W, int NLD, int QN, bool TERN>
__global__ __launch_bounds__(kThreads) void k_topk_filter_seq(const uint32_t* __restrict__ qbits, const uint32_t* __restrict__ qzero,
                                                              const uint32_t* __restrict__ rbits, const uint32_t* __restrict__ rzero, int pad,
                                                              int Q, int64_t R, const uint32_t* __restrict__ t_est, const uint32_t* __restrict__ bound,
                                                              uint32_t* __restrict__ cnt, unsigned long long* __restrict__ cand) {
    static_assert(W % 4 == 0 && W <= 64, "whole 16-byte pieces, at most 16 lanes per item");
    constexpr int LPI = W / 4;                              // lanes (pieces) per item
    constexpr int LOGL = LPI == 1 ? 0 : (LPI == 2 ? 1 : (LPI == 4 ? 2 : (LPI == 8 ? 3 : 4)));
    static_assert((1 << LOGL) == LPI, "a power of two");
    constexpr int TILE = kThreads * NLD;                    // pieces per tile
    constexpr int NZ = TERN ? NLD : 1, QZ = TERN ? QN : 1;
    const int part = threadIdx.x & (LPI - 1);
    const int q0 = blockIdx.y * QN;
    const int64_t npieces = R * LPI;
    const int64_t nfull = npieces / TILE, ntiles = (npieces + TILE - 1) / TILE;
    const topk_u4* __restrict__ g = reinterpret_cast<const topk_u4*>(rbits);
    const topk_u4* __restrict__ gz = reinterpret_cast<const topk_u4*>(rzero);    // TERN: the zero plane, read piece for piece like the bits
    topk_u4 qw[QN], qz[QZ];
    int thr[QN];
    int bnd[QN];                                            // items at the threshold count only below this index (index_bound; R < 2^31)
#pragma unroll
    for (int q = 0; q < QN; ++q) {
        const int qq = q0 + q < Q ? q0 + q : Q - 1;            // surplus slots repeat the last query and are ignored below
        qw[q] = *reinterpret_cast<const topk_u4*>(qbits + (int64_t)qq * W + 4 * part);
        if constexpr (TERN) qz[q] = *reinterpret_cast<const topk_u4*>(qzero + (int64_t)qq * W + 4 * part);
        thr[q] = q0 + q < Q ? (int)t_est[qq] : -1;
        bnd[q] = (int)bound[qq];
        if constexpr (TERN) thr[q] = q0 + q < Q ? thr[q] + pad : -1;     // the joined sums below still hold the padding bits: compare there
    }
    auto piece_of = [&](int64_t tile, int j) -> int64_t { return tile * TILE + (int64_t)j * kThreads + threadIdx.x; };
    auto load_tile = [&](topk_u4 (&dst)[NLD], topk_u4 (&dstz)[NZ], int64_t tile) {
        if (tile < nfull) {                                 // uniform: whole tiles load without a bounds check
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                dst[j] = __builtin_nontemporal_load(g + piece_of(tile, j));
                if constexpr (TERN) dstz[j] = __builtin_nontemporal_load(gz + piece_of(tile, j));
            }
        } else {
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                const int64_t pc = piece_of(tile, j);
                dst[j] = pc < npieces ? __builtin_nontemporal_load(g + pc) : topk_u4{0u, 0u, 0u, 0u};
                if constexpr (TERN) dstz[j] = pc < npieces ? __builtin_nontemporal_load(gz + pc) : topk_u4{0u, 0u, 0u, 0u};
            }
        }
    };
    __shared__ uint2 stage_all[kThreads / 64][kStageV];
    __shared__ uint32_t stage_n[kThreads / 64];
    uint2* mine_stage = stage_all[wave_id()];
    uint32_t* mine_n = stage_n + wave_id();
    if (lane_id() == 0) *mine_n = 0;
    __builtin_amdgcn_wave_barrier();
    topk_u4 cur[NLD], nxt[NLD], curz[NZ], nxtz[NZ];
    int64_t tile = blockIdx.x;
    if (tile < ntiles) load_tile(cur, curz, tile);
    for (; tile < ntiles; tile += gridDim.x) {
        const int64_t tn = tile + gridDim.x;
        if (tn < ntiles) load_tile(nxt, nxtz, tn);
        int dd[QN][NLD];
        bool hit_any = false;
        const int first_item = (int)((tile * TILE) >> LOGL);  // tiles are in index order: past the bound the threshold bucket no longer counts
#pragma unroll
        for (int q = 0; q < QN; ++q) {
            const int te = thr[q] - (first_item >= bnd[q] ? 1 : 0);       // uniform
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                int h;
                if constexpr (TERN) {                       // half units: positions dead on either side + 2 x (live and different)
                    const topk_u4 z = curz[j] | qz[q];
                    const topk_u4 x = (cur[j] ^ qw[q]) & ~z;
                    h = __popc(z.x) + __popc(z.y) + __popc(z.z) + __popc(z.w) + 2 * (__popc(x.x) + __popc(x.y) + __popc(x.z) + __popc(x.w));
                } else {
                    h = __popc(cur[j].x ^ qw[q].x) + __popc(cur[j].y ^ qw[q].y) + __popc(cur[j].z ^ qw[q].z) + __popc(cur[j].w ^ qw[q].w);
                }
                dd[q][j] = join_pieces<LPI>(h);
                hit_any |= dd[q][j] <= te;                  // every lane of the item sees it; pieces past the end are sorted out below
            }
        }
        if (__ballot(hit_any)) {                            // uncommon
#pragma unroll
            for (int q = 0; q < QN; ++q) {
#pragma unroll
                for (int j = 0; j < NLD; ++j) {
                    const int64_t pc = piece_of(tile, j);
                    const bool in = dd[q][j] < thr[q] || (dd[q][j] == thr[q] && (int)(pc >> LOGL) < bnd[q]);
                    stage_candidate(part == 0 && pc < npieces && in, (uint32_t)(pc >> LOGL), (uint32_t)(dd[q][j] - (TERN ? pad : 0)), q, mine_stage, mine_n, q0, cnt, cand);
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (__builtin_amdgcn_readfirstlane((int)*mine_n) >= 64) flush_staged(mine_stage, mine_n, kStageV, q0, cnt, cand);
        }
        if (tn < ntiles) {
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                cur[j] = nxt[j];
                if constexpr (TERN) curz[j] = nxtz[j];
            }
        }
    }
    flush_staged(mine_stage, mine_n, kStageV, q0, cnt, cand);
}

// ---- ternary codes of 32 / 64 bits (W = 1, 2): one lane per item, NLD items per lane and tile, both planes (round 6) -----------------
// MITH / DSPH quantise with sign_() and can emit exact zeros (reference runners/base.py:407-410, runners/MITH/runner.py:125-131); such
// code sets are rare and short, so this filter keeps the simple form: 4- / 8-byte loads from either plane, queries in scalar registers.
template <int W, int NLD, int QN>
__global__ __launch_bounds__(kThreads) void k_topk_filter_item_tern(const uint32_t* __restrict__ qbits, const uint32_t* __restrict__ qzero,
                                                                    const uint32_t* __restrict__ rbits, const uint32_t* __restrict__ rzero, int pad,
                                                                    int Q, int64_t R, const uint32_t* __restrict__ t_est, const uint32_t* __restrict__ bound,
                                                                    uint32_t* __restrict__ cnt, unsigned long long* __restrict__ cand) {
    constexpr int TILE = kThreads * NLD;                    // items per tile
    const int q0 = blockIdx.y * QN;
    const int64_t ntiles = (R + TILE - 1) / TILE;
    int thr[QN], bnd[QN];
#pragma unroll
    for (int q = 0; q < QN; ++q) {
        const int qq = q0 + q < Q ? q0 + q : Q - 1;
        thr[q] = q0 + q < Q ? (int)t_est[qq] : -1;
        bnd[q] = (int)bound[qq];
    }
    __shared__ uint2 stage_all[kThreads / 64][kStageV];
    __shared__ uint32_t stage_n[kThreads / 64];
    uint2* mine_stage = stage_all[wave_id()];
    uint32_t* mine_n = stage_n + wave_id();
    if (lane_id() == 0) *mine_n = 0;
    __builtin_amdgcn_wave_barrier();
    auto item_of = [&](int64_t tile, int j) -> int64_t { return tile * TILE + (int64_t)j * kThreads + threadIdx.x; };
    Rec<W> cur[NLD], curz[NLD], nxt[NLD], nxtz[NLD];
    auto load_tile = [&](Rec<W> (&b)[NLD], Rec<W> (&z)[NLD], int64_t tile) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int64_t it = item_of(tile, j);
            load_rec<W>(b[j], rbits, it, it < R);
            load_rec<W>(z[j], rzero, it, it < R);
        }
    };
    int64_t tile = blockIdx.x;
    if (tile < ntiles) load_tile(cur, curz, tile);
    for (; tile < ntiles; tile += gridDim.x) {
        const int64_t tn = tile + gridDim.x;
        if (tn < ntiles) load_tile(nxt, nxtz, tn);
#pragma unroll
        for (int q = 0; q < QN; ++q) {
            const int qq = q0 + q < Q ? q0 + q : Q - 1;    // uniform: scalar loads
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                const int64_t it = item_of(tile, j);
                const int d = dist2_words<W>(cur[j].w, curz[j].w, qbits + (int64_t)qq * W, qzero + (int64_t)qq * W, pad);
                const bool in = it < R && (d < thr[q] || (d == thr[q] && (int)it < bnd[q]));
                if (__ballot(in)) stage_candidate(in, (uint32_t)it, (uint32_t)d, q, mine_stage, mine_n, q0, cnt, cand);
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (__builtin_amdgcn_readfirstlane((int)*mine_n) >= 64) flush_staged(mine_stage, mine_n, kStageV, q0, cnt, cand);
        if (tn < ntiles) {
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                cur[j] = nxt[j];
                curz[j] = nxtz[j];
            }
        }
    }
    flush_staged(mine_stage, mine_n, kStageV, q0, cnt, cand);
}

// ---- short codes (32 / 64 bits per item: W = 1, 2) as 16-byte pieces, round 5 --------------------------------------------------------
// The per-item filter reads 4 or 8 bytes per lane: 256 / 512 bytes per wave instruction, and 40 M x 32-bit codes streamed at 3.4 TB/s,
// bound by the number of load instructions, not by HBM or the VALU.  Here a lane loads 16 bytes = 4 / 2 ITEMS (contiguous 1 KB per
// wave instruction, non-temporal).  The common path keeps only a running minimum per query (v_min3 takes two distances at a time) and
// compares once per tile; a tile with a candidate recomputes its distances in the rare path.  Query words and thresholds are uniform:
// scalar registers.  A gallery view that does not start on a 16-byte boundary (a shard cut at any row) is read from the boundary
// below it; the pieces at either end that are not whole are loaded word by word.
template <int W, int NLD, int QN>
__global__ __launch_bounds__(kThreads) void k_topk_filter_short(const uint32_t* __restrict__ qbits, const uint32_t* __restrict__ rbits,
                                                                int Q, int64_t R, const uint32_t* __restrict__ t_est, const uint32_t* __restrict__ bound,
                                                                uint32_t* __restrict__ cnt, unsigned long long* __restrict__ cand) {
    static_assert(W == 1 || W == 2, "4 or 2 items per 16-byte piece");
    constexpr int IPP = 4 / W;
    constexpr int TILE = kThreads * NLD;                    // pieces per tile
    const int q0 = blockIdx.y * QN;
    const int mis = (int)((reinterpret_cast<uintptr_t>(rbits) & 15) >> 2);           // words between the 16-byte boundary below and the first item
    const topk_u4* __restrict__ g = reinterpret_cast<const topk_u4*>(reinterpret_cast<uintptr_t>(rbits) & ~(uintptr_t)15);
    const uint32_t* __restrict__ gw = reinterpret_cast<const uint32_t*>(g);
    const int64_t nwords = (int64_t)mis + R * W;            // words from the boundary to the end of the gallery
    const int64_t npieces = (nwords + 3) >> 2;
    const int64_t ntiles = (npieces + TILE - 1) / TILE;
    uint32_t qw[QN][W];
    int thr[QN];
    int bnd[QN];                                            // items at the threshold count only below this index (index_bound; R < 2^31)
#pragma unroll
    for (int q = 0; q < QN; ++q) {
        const int qq = q0 + q < Q ? q0 + q : Q - 1;            // surplus slots repeat the last query and never hit
#pragma unroll
        for (int x = 0; x < W; ++x) qw[q][x] = qbits[(int64_t)qq * W + x];
        thr[q] = q0 + q < Q ? (int)t_est[qq] : -1;
        bnd[q] = (int)bound[qq];
    }
    auto piece_of = [&](int64_t tile, int j) -> int64_t { return tile * TILE + (int64_t)j * kThreads + threadIdx.x; };
    auto load_tile = [&](topk_u4 (&dst)[NLD], int64_t tile) {
        const bool whole = (tile > 0 || mis == 0) && (tile + 1) * (int64_t)TILE * 4 <= nwords;      // uniform
        if (whole) {
#pragma unroll
            for (int j = 0; j < NLD; ++j) dst[j] = __builtin_nontemporal_load(g + piece_of(tile, j));
        } else {
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                const int64_t w0 = piece_of(tile, j) * 4;
                uint32_t v[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) v[x] = (w0 + x >= mis && w0 + x < nwords) ? gw[w0 + x] : 0u;
                dst[j] = topk_u4{v[0], v[1], v[2], v[3]};
            }
        }
    };
    auto dist_of = [&](const topk_u4& pc, int s, int q) -> int {
        if constexpr (W == 1) return __popc(pc[s] ^ qw[q][0]);
        else return __popc(pc[2 * s] ^ qw[q][0]) + __popc(pc[2 * s + 1] ^ qw[q][1]);
    };
    __shared__ uint2 stage_all[kThreads / 64][kStageV];
    __shared__ uint32_t stage_n[kThreads / 64];
    uint2* mine_stage = stage_all[wave_id()];
    uint32_t* mine_n = stage_n + wave_id();
    if (lane_id() == 0) *mine_n = 0;
    __builtin_amdgcn_wave_barrier();
    topk_u4 cur[NLD], nxt[NLD];
    int64_t tile = blockIdx.x;
    if (tile < ntiles) load_tile(cur, tile);
    for (; tile < ntiles; tile += gridDim.x) {
        const int64_t tn = tile + gridDim.x;
        if (tn < ntiles) load_tile(nxt, tn);
        bool hit_any = false;
        unsigned qhit = 0;
        const int first_item = (int)((tile * TILE * 4 - mis) / W);   // (negative in the first tile of a view that starts inside a piece: below any bound)
#pragma unroll
        for (int q = 0; q < QN; ++q) {
            const int te = thr[q] - (first_item >= bnd[q] ? 1 : 0);  // uniform: past the bound the threshold bucket no longer counts
            int m = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                if constexpr (IPP == 4) {
                    m = min(m, min(dist_of(cur[j], 0, q), dist_of(cur[j], 1, q)));
                    m = min(m, min(dist_of(cur[j], 2, q), dist_of(cur[j], 3, q)));
                } else {
                    m = min(m, min(dist_of(cur[j], 0, q), dist_of(cur[j], 1, q)));
                }
            }
            const bool h = m <= te;                         // zero words of a ragged end may vote: sorted out below
            hit_any |= h;
            qhit |= (__ballot(h) != 0ull ? 1u : 0u) << q;
        }
        if (__ballot(hit_any)) {                            // uncommon: recompute the tile's distances for the queries that voted
#pragma unroll
            for (int q = 0; q < QN; ++q) {
                if (!((qhit >> q) & 1u)) continue;          // wave-uniform
#pragma unroll
                for (int j = 0; j < NLD; ++j) {
                    const int64_t w0 = piece_of(tile, j) * 4 - mis;      // word index of the piece's first word, relative to the gallery
#pragma unroll
                    for (int s = 0; s < IPP; ++s) {
                        const int64_t wi = w0 + s * W;
                        const int d = dist_of(cur[j], s, q);
                        const bool in = d < thr[q] || (d == thr[q] && (int)(wi / W) < bnd[q]);
                        stage_candidate(wi >= 0 && wi < R * W && in, (uint32_t)(wi / W), (uint32_t)d, q, mine_stage, mine_n, q0, cnt, cand);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (__builtin_amdgcn_readfirstlane((int)*mine_n) >= 64) flush_staged(mine_stage, mine_n, kStageV, q0, cnt, cand);
        }
        if (tn < ntiles) {
#pragma unroll
            for (int j = 0; j < NLD; ++j) cur[j] = nxt[j];
        }
    }
    flush_staged(mine_stage, mine_n, kStageV, q0, cnt, cand);
}

// ---- the filter for MANY queries on the matrix cores -------------------------------------------------------------------------
// From a handful of queries on the filter is bound by its integer work (17 VALU operations per query and item), not by the gallery
// stream.
// v_mfma_i32_16x16x64_i8 takes that work: the Hamming distance of query q and item x is popcount(q) + sum_i s_i x_i with
// s_i = 1 - 2 q_i.  The item's bits become bytes WITHOUT being moved: word & (0x01010101 << p) leaves bits p, p + 8, p + 16, p + 24 of
// a 32-bit word each alone in its byte, worth 2^p there (p = 7 goes through (word >> 1) & 0x40404040: +128 is not an int8) -- 9
// operations for 32 bits, independent of the number of queries.  The query side (B operand, built once per wave and kept in
// registers) carries the matching weight: its byte for that bit is s_i * 64 / 2^p, so every product is 64 s_i x_i, and with the
// accumulator started at 64 (popcount(q) - threshold(q) - 1) one chain of K/64 MFMAs leaves 64 (distance - threshold - 1) for 16
// items x 16 queries: lane l holds query l & 15 and the items 4 * (l >> 4) + r, r = 0..3.  A candidate is a NEGATIVE result, so ONE
// vote on the OR of a lane's 4 * QT results covers all of them.  Lane (row = l & 15, quarter = l >> 4) supplies the quarter
// `quarter` of item `row`; which of its bits sits in which k slot of which MFMA is the same on both operands and otherwise free (a
// sum over k does not care).
// Candidates go to the same per-query lists as in k_topk_filter, through a wave-private staging list (below).  W % 4 == 0
// (128-bit steps of the code length); QT = query tiles of 16 per pass over the gallery.
typedef int topk_v4i __attribute__((ext_vector_type(4)));

template <int W, int QT>
// four query tiles: 172 registers would leave two waves per SIMD; capped to three (4 spilled outside the loop): 0.149 -> 0.132 ms at Q = 64
__global__ __launch_bounds__(kThreads, (QT == 4 ? 3 : 1)) void k_topk_filter_mfma(const uint32_t* __restrict__ qbits, const uint32_t* __restrict__ rbits,
                                                               int Q, int64_t R, const uint32_t* __restrict__ t_est,
                                                               uint32_t* __restrict__ cnt, unsigned long long* __restrict__ cand) {
    static_assert(W % 4 == 0, "a lane owns a quarter of an item: whole words");
    constexpr int KT = W / 2;                               // MFMAs per distance (64 bits each)
    constexpr int LW = W / 4;                               // words per lane
    constexpr int U = 4;                                    // groups of 16 items per wave and step (2 and 8 measured the same) (2 and 8 measured the same)
    const int lane = lane_id(), row = lane & 15, quarter = lane >> 4;
    const int q0 = blockIdx.y * (16 * QT);
    topk_v4i bq[QT][KT];
    int bias[QT], thr[QT];                                  // accumulator start 64 (popcount(q) - threshold - 1)
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int q = q0 + 16 * t + row;                    // B operand: column = lane & 15
        const int qq = q < Q ? q : Q - 1;
        int pc = 0;
        for (int x = 0; x < W; ++x) pc += __popc(qbits[(int64_t)qq * W + x]);
        thr[t] = q < Q ? (int)t_est[qq] : -1;               // surplus columns: start at 64 * popcount >= 0, never negative
        bias[t] = 64 * (pc - thr[t] - 1);
#pragma unroll
        for (int m = 0; m < KT; ++m) {
            const uint32_t w = qbits[(int64_t)qq * W + quarter * LW + (m >> 1)];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int p = 4 * (m & 1) + x;              // the item side's mask number: bits p, p + 8, p + 16, p + 24 of the word
                const int mag = p < 7 ? (64 >> p) : 1;
                uint32_t b = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) b |= (uint32_t)(uint8_t)(((w >> (p + 8 * j)) & 1u) ? -mag : mag) << (8 * j);
                bq[t][m][x] = (int)b;
            }
        }
    }
    const int64_t nstep = (R + 16 * U - 1) / (16 * U);
    const int64_t wstride = (int64_t)gridDim.x * (kThreads / 64);
    int64_t step = (int64_t)blockIdx.x * (kThreads / 64) + wave_id();
    uint32_t cur[U][LW], nxt[U][LW];
    auto load = [&](uint32_t (&dst)[U][LW], int64_t st) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // the 64 lanes read the 16 items of a group as ONE contiguous run (lane j: quarter j & 3 of item j >> 2); the operand
            // layout wants quarter l >> 4 of item l & 15 in lane l: exchanged through ds_bpermute when the words are used
            int64_t it = (st * U + u) * 16 + (lane >> 2);
            if (it >= R) it = R - 1;                        // clamped rows repeat the last item: dropped where candidates are staged
            const uint32_t* p = rbits + it * W + (lane & 3) * LW;
            if constexpr (LW == 1) dst[u][0] = p[0];
            else if constexpr (LW == 2) {
                const uint2 v = *reinterpret_cast<const uint2*>(p);
                dst[u][0] = v.x; dst[u][1] = v.y;
            } else {
#pragma unroll
                for (int x = 0; x < LW / 4; ++x) {
                    const uint4 v = reinterpret_cast<const uint4*>(p)[x];
                    dst[u][4 * x] = v.x; dst[u][4 * x + 1] = v.y; dst[u][4 * x + 2] = v.z; dst[u][4 * x + 3] = v.w;
                }
            }
        }
    };
    // Candidates are staged in a wave-private LDS list (a lane with a candidate takes its slot with an LDS atomic on the wave's own
    // counter: only the lanes that hold one run that code) and go out 64 at a time: one global
    // atomic round trip per flush instead of one per candidate -- each used to hold its wave for the atomic's return AND for the
    // prefetched tile, because the two share vmcnt (Q = 64: 0.23 ms per pass with the direct append, 0.16 staged).
    constexpr int kStage = 192;
    __shared__ uint2 stage_all[kThreads / 64][kStage];
    __shared__ uint32_t stage_n[kThreads / 64];
    uint2* mine_stage = stage_all[wave_id()];
    uint32_t* mine_n = stage_n + wave_id();
    if (lane == 0) *mine_n = 0;
    bool dirty = false;                                     // wave-uniform: something was staged since the last look at the count
    const int src4 = 4 * (4 * row + quarter);               // ds_bpermute address: the lane that loaded this lane's quarter of its item
    if (step < nstep) load(cur, step);
    for (; step < nstep; step += wstride) {
        const bool more = step + wstride < nstep;
        if (more) load(nxt, step + wstride);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            topk_v4i a[KT];
#pragma unroll
            for (int v = 0; v < LW; ++v) {                  // one word = two k tiles: 9 operations for 32 bits
                const uint32_t w = (uint32_t)__builtin_amdgcn_ds_bpermute(src4, (int)cur[u][v]);
#pragma unroll
                for (int p = 0; p < 7; ++p) a[2 * v + (p >> 2)][p & 3] = (int)(w & (0x01010101u << p));
                a[2 * v + 1][3] = (int)((w >> 1) & 0x40404040u);
            }
            topk_v4i acc[QT];
#pragma unroll
            for (int t = 0; t < QT; ++t) acc[t] = topk_v4i{bias[t], bias[t], bias[t], bias[t]};
#pragma unroll
            for (int m = 0; m < KT; ++m)                    // k tile outermost: consecutive MFMAs belong to different chains
#pragma unroll
                for (int t = 0; t < QT; ++t) acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[m], bq[t][m], acc[t], 0, 0, 0);
            int tsign[QT], sign = 0;                        // sign bit set <=> a candidate among the 4 results of the tile / of the lane
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                tsign[t] = acc[t][0] | acc[t][1] | acc[t][2] | acc[t][3];
                sign |= tsign[t];
            }
            if (__ballot(sign < 0)) {                       // a candidate somewhere in these 16 items x 16 QT queries (about one group in six at Q = 64)
                dirty = true;
                if (sign < 0) {                             // divergent from here: usually one lane
                    const int64_t it0 = (step * U + u) * 16 + 4 * quarter;   // C rows of this lane: it0 + r
#pragma unroll
                    for (int t = 0; t < QT; ++t) {
                        if (tsign[t] >= 0) continue;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (acc[t][r] < 0 && it0 + r < R) {         // clamped rows of the last group repeat item R - 1: dropped here
                                const uint32_t d = (uint32_t)((acc[t][r] >> 6) + thr[t] + 1);
                                const uint32_t pos = atomicAdd(mine_n, 1u);
                                if (pos < (uint32_t)kStage) mine_stage[pos] = make_uint2((uint32_t)(it0 + r), d | ((uint32_t)(16 * t + row) << 16));
                                else append_one(q0 + 16 * t + row, d, (uint32_t)(it0 + r), cnt, cand, sub_of_wave((int)(pos >> 6)));
                            }
                        }
                    }
                }
            }
        }
        if (dirty) {                                        // once per step at most: is the list worth a round trip?
            dirty = false;
            __builtin_amdgcn_wave_barrier();
            if (__builtin_amdgcn_readfirstlane((int)*mine_n) >= 64) flush_staged(mine_stage, mine_n, kStage, q0, cnt, cand);
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int x = 0; x < LW; ++x) cur[u][x] = nxt[u][x];
        }
    }
    flush_staged(mine_stage, mine_n, kStage, q0, cnt, cand);
}

// block-wide search: first bin b of hist[0..n) whose cumulative count reaches `need` (1 <= need <= total) -> out[0] = b,
// out[1] = count below b.  Every thread sums a contiguous segment, wave scan, cross-wave offsets through LDS, the owning
// thread walks its segment (one wave stepping through 64 bins at a time was 16 dependent rounds for the 1024 index bins).
__device__ __forceinline__ void block_find(const uint32_t* hist, int n, uint32_t need, int* out, uint32_t* wtot) {
    const int per = (n + kThreads - 1) / kThreads;
    const int lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    uint32_t mine = 0;
    for (int d = lo; d < hi; ++d) mine += hist[d];
    uint32_t incl = mine;
    const int lane = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_up(incl, o, 64);
        if (lane >= o) incl += u;
    }
    if (lane == 63) wtot[wave_id()] = incl;
    __syncthreads();
    uint32_t excl = incl - mine;
    for (int w = 0; w < wave_id(); ++w) excl += wtot[w];
    if (excl < need && need <= excl + mine) {                       // exactly one thread
        uint32_t run = excl;
        for (int d = lo; d < hi; ++d) {
            const uint32_t h = hist[d];
            if (run + h >= need) {
                out[0] = d;
                out[1] = (int)run;
                break;
            }
            run += h;
        }
    }
    __syncthreads();
}

// one block per query: verify the candidate list, radix-select its k smallest (distance, index) keys, write them in order
__global__ __launch_bounds__(kThreads) void k_topk_select(const unsigned long long* __restrict__ cand, const uint32_t* __restrict__ cnt,
                                                          int64_t R, int k, int nb, int64_t base_index, uint16_t* __restrict__ out_d,
                                                          int32_t* __restrict__ out_i, int* __restrict__ fail) {
    // LDS: key[kCandCap] (64-bit (distance, index) keys), surv[1024], hist[max(nb, 1024)], a few scalars
    extern __shared__ __attribute__((aligned(16))) unsigned long long key[];
    unsigned long long* surv = key + kCandCap;
    uint32_t* hist = reinterpret_cast<uint32_t*>(surv + 1024);
    const int nh = nb > 2048 ? nb : 2048;                       // >= 8 KB: reused as a list of 1024 keys
    int* sc = reinterpret_cast<int*>(hist + nh);               // [0] d*, [1] count below d*, [2] bin*, [3] count below bin*, [4] survivors, [5] keys in the last bin, [8..11] wave totals
    uint32_t* wtot = reinterpret_cast<uint32_t*>(sc + 8);
    const int q = blockIdx.x;
    // the list is kSub sub-lists (see kSub).  The first 128 keys of each are requested together with the counts (lists are a few hundred
    // keys: one miss latency instead of two): 32 threads per sub-list, four keys each
    static_assert(kThreads == 32 * kSub, "32 threads per sub-list");
    const unsigned long long* cq = cand + (int64_t)q * kCandCap;
    const int sub = threadIdx.x >> 5, sl = threadIdx.x & 31;
    unsigned long long kspec[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) kspec[i] = cq[sub * kSubCap + sl + 32 * i];
    uint32_t nsub[kSub], n = 0, base = 0;
    bool over = false;
#pragma unroll
    for (int s_ = 0; s_ < kSub; ++s_) {
        nsub[s_] = cnt[((int64_t)q * kSub + s_) * kCntStride];
        over |= nsub[s_] > (uint32_t)kSubCap;
        if (s_ < sub) base += nsub[s_];
        n += nsub[s_];
    }
    uint32_t mine_n = 0;
#pragma unroll
    for (int s_ = 0; s_ < kSub; ++s_) mine_n = s_ == sub ? nsub[s_] : mine_n;
    const uint32_t want = (uint32_t)((int64_t)k < R ? (int64_t)k : R);
    if (over || n < want) {                                     // a sub-list overflowed (its counter ran on) or too few candidates
        if (threadIdx.x == 0) fail[q] = 1;
        return;
    }
    const int kk = (int)want;                                   // number of real results (<= k)
    // Radix selection instead of sorting all candidates: a distance histogram finds the bucket d* where the k-th result lies;
    // everything below it survives, inside it a histogram over the top 10 index bits finds the bin, and only the (few)
    // candidates of that last bin are ranked against each other.  The <= k survivors are then placed by counting.
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if ((uint32_t)(sl + 32 * i) < mine_n) key[base + sl + 32 * i] = kspec[i];
    for (uint32_t e = 128 + sl; e < mine_n; e += 32) key[base + e] = cq[sub * kSubCap + e];
    for (int e = threadIdx.x; e < nb; e += kThreads) hist[e] = 0u;
    if (threadIdx.x == 0) sc[4] = 0;
    __syncthreads();
    for (int p = threadIdx.x; p < (int)n; p += kThreads) atomicAdd(&hist[(uint32_t)(key[p] >> 32)], 1u);
    __syncthreads();
    block_find(hist, nb, (uint32_t)kk, sc, wtot);               // first bucket where the cumulative count reaches kk
    const uint32_t dstar = (uint32_t)sc[0];
    const int need = kk - sc[1];                               // results still to come from bucket d*
    int shift = 0;
    while ((R >> shift) > 1024) ++shift;                        // 1024 index bins
    for (int e = threadIdx.x; e < 1024; e += kThreads) hist[e] = 0u;
    __syncthreads();
    for (int p = threadIdx.x; p < (int)n; p += kThreads)
        if ((uint32_t)(key[p] >> 32) == dstar) atomicAdd(&hist[(uint32_t)key[p] >> shift], 1u);
    __syncthreads();
    block_find(hist, 1024, (uint32_t)need, sc + 2, wtot);
    const uint32_t bstar = (uint32_t)sc[2];
    const int need2 = need - sc[3];                            // results still to come from (d*, bin*)
    unsigned long long* grp = reinterpret_cast<unsigned long long*>(hist);      // the histogram is done: its space lists the last bin
    if (threadIdx.x == 0) sc[5] = 0;
    __syncthreads();
    for (int p = threadIdx.x; p < (int)n; p += kThreads) {
        const unsigned long long mine = key[p];
        const uint32_t d = (uint32_t)(mine >> 32), bin = (uint32_t)mine >> shift;
        if (d < dstar || (d == dstar && bin < bstar)) surv[atomicAdd(&sc[4], 1)] = mine;
        else if (d == dstar && bin == bstar) {
            const int g = atomicAdd(&sc[5], 1);
            if (g < 1024) grp[g] = mine;
        }
    }
    __syncthreads();
    const int ng = sc[5];
    if (ng > 1024) {                                           // thousands of equal distances inside one index bin: leave it to
        if (threadIdx.x == 0) fail[q] = 1;                     // the robust path
        return;
    }
    for (int p = threadIdx.x; p < ng; p += kThreads) {         // rank inside the last bin (usually a handful of keys)
        const unsigned long long mine = grp[p];
        int pos = 0;
        for (int j = 0; j < ng; ++j) pos += grp[j] < mine;
        if (pos < need2) surv[atomicAdd(&sc[4], 1)] = mine;
    }
    __syncthreads();
    const int ns = sc[4];                                       // == kk
    for (int p = threadIdx.x; p < ns; p += kThreads) {
        const unsigned long long mine = surv[p];
        int pos = 0;
        for (int j = 0; j < ns; ++j) pos += surv[j] < mine;
        out_d[(int64_t)q * k + pos] = (uint16_t)(mine >> 32);
        out_i[(int64_t)q * k + pos] = (int32_t)(base_index + (int64_t)(uint32_t)mine);
    }
    for (int p = kk + threadIdx.x; p < k; p += kThreads) {      // shard smaller than k: unused slots
        out_d[(int64_t)q * k + p] = (uint16_t)kInf;
        out_i[(int64_t)q * k + p] = -1;
    }
}

struct TopkPlan {
    int W, ipt, tile, nblocks, tiles_per_block, nqg, nb;
    Layout L, Lm;
    size_t robust_bytes, off_ctl, off_hist, off_test, off_bound, off_cnt, off_fail, off_cand;
    size_t ws_bytes;
};

int ipt_for(int W) { return W >= 16 ? 1 : (W >= 8 ? 4 : 8); }      // long codes: one 64..256-byte record per thread and tile

int plan_topk(int64_t Q, int64_t R, int K, int k, TopkPlan* p, bool tern = false) {
    if (Q <= 0 || R <= 0 || K <= 0) return xmh::fail(XMH_EINVAL, "topk: bad shape Q=%lld R=%lld K=%d", (long long)Q, (long long)R, K);
    if (k <= 0 || k > 1024) return xmh::fail(XMH_EINVAL, "topk: k=%d out of range (1..1024)", k);
    if (R >= (1ll << 31) - 65536) return xmh::fail(XMH_ENOTSUP, "topk: shard of %lld rows (max 2^31-1)", (long long)R);
    const int W = (K + 31) / 32;
    if (W != 1 && W != 2 && W != 4 && W != 8 && W != 16 && W != 32 && W != 64)
        return xmh::fail(XMH_ENOTSUP, "topk: K=%d unsupported (code words must be a power of two up to 64, i.e. K <= 2048)", K);
    p->W = W;
    p->ipt = ipt_for(W);
    p->tile = kThreads * p->ipt;
    const int nb = tern ? 2 * K + 1 : K + 1;               // ternary codes: half-unit distances 0 ... 2K
    p->nb = nb;
    p->L.nb = nb;
    p->L.nq = kQG;
    p->L.cap = k + p->tile + 64;
    if (p->L.bytes() > 160 * 1024) return xmh::fail(XMH_ENOTSUP, "topk: k=%d, K=%d%s needs %zu B of LDS (max 163840)", k, K, tern ? " (ternary)" : "", p->L.bytes());
    if ((p->L.cap + kThreads - 1) / kThreads > 24) return xmh::fail(XMH_ENOTSUP, "topk: candidate buffer too large for the compaction segment");
    const int64_t ntiles = xmh::ceil_div(R, p->tile);
    int bpc = (int)((160 * 1024) / p->L.bytes());
    if (bpc > 4) bpc = 4;
    if (bpc < 1) bpc = 1;
    int64_t nblocks = (int64_t)xmh::device_cu_count() * bpc;
    if (nblocks > ntiles) nblocks = ntiles;
    p->tiles_per_block = (int)xmh::ceil_div(ntiles, nblocks);
    p->nblocks = (int)xmh::ceil_div(ntiles, p->tiles_per_block);
    p->nqg = (int)xmh::ceil_div(Q, kQG);
    p->Lm.nb = nb;
    p->Lm.nq = 3;
    p->Lm.cap = k + kThreads * 4 + 64;
    if (p->Lm.bytes() > 160 * 1024) return xmh::fail(XMH_ENOTSUP, "topk: k=%d, K=%d needs %zu B of LDS in the merge (max 163840)", k, K, p->Lm.bytes());
    p->robust_bytes = ((size_t)Q * p->nblocks * k * 6 + 255) & ~(size_t)255;
    size_t o = p->robust_bytes;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o += (bytes + 255) & ~(size_t)255;
        return at;
    };
    p->off_ctl = take(256);                           // ctl, hist, t_est, cnt, fail are contiguous: one memset clears them
    p->off_hist = take((size_t)Q * nb * 4);
    p->off_test = take((size_t)Q * 4);
    p->off_bound = take((size_t)Q * 4);
    p->off_cnt = take((size_t)Q * kSub * kCntStride * 4);
    p->off_fail = take((size_t)Q * 4);
    p->off_cand = take((size_t)Q * kCandCap * 8);
    p->ws_bytes = o;
    return XMH_OK;
}

// opt a kernel in to more than 64 KB of dynamic LDS, once per (device, kernel) and size (xmh::raise_dynamic_lds)
template <typename KernT>
int raise_lds(KernT kern, size_t bytes, const char* who) { return xmh::raise_dynamic_lds(reinterpret_cast<const void*>(kern), bytes, who); }

}  // namespace

extern "C" size_t xmh_topk_ws_bytes(int64_t Q, int64_t R, int K, int k) {
    TopkPlan p;
    if (plan_topk(Q, R, K, k, &p) != XMH_OK) return 0;
    return p.ws_bytes;
}

extern "C" int xmh_topk_ws_init(int64_t Q, int64_t R, int K, int k, void* ws, size_t ws_bytes, xmh_stream_t stream) {
    TopkPlan p;
    int rc = plan_topk(Q, R, K, k, &p);
    if (rc) return rc;
    if (!ws) return xmh::fail(XMH_EINVAL, "xmh_topk_ws_init: null workspace");
    if (ws_bytes < p.ws_bytes) return xmh::fail(XMH_EINVAL, "xmh_topk_ws_init: workspace too small (%zu < %zu)", ws_bytes, p.ws_bytes);
    XMH_HIP(hipMemsetAsync(static_cast<char*>(ws) + p.off_ctl, 0, p.off_cand - p.off_ctl, xmh::as_stream(stream)));
    return XMH_OK;
}

namespace {
// Which instance of the streaming filter a (code words, queries) call launches -- one decision for the launch and for xmh_topk_describe:
//   k_topk_filter_mfma<W, qt>      distances on the matrix cores: >= 5 queries at 128 / 256 / 512 bits, qt tiles of 16 queries per pass
//                                  (XMH_TOPK_MFMA, experiments builds only: the smallest query count that takes it, 0 = never);
//   k_topk_filter_seq<W, 4, qn>    codes of whole 16-byte pieces (128 ... 2048 bits), every other case: qn queries per pass;
//   k_topk_filter_short<W, 4, qn>  32- and 64-bit codes, 4 / 2 items per 16-byte piece.
// Measured (10 M x 256 bit, filter launch): one matrix-core pass over 16 queries 62-64 us whatever their number; per-piece filter 49 / 50 /
// 51.5 / 52 us for 1 / 2 / 3 / 4 queries (3 run as 4; the matrix cores took 67) and 74-78 us for 5-8 as one group of 8 -- VALU-bound, so
// the matrix cores keep those.  The per-item filter of rounds 1-4 (k_topk_filter: one lane per item, 54 / 58 / 95 / 61 / 79 / 80 us for 1 /
// 2 / 3 / 4 / 6 / 8 queries; query groups sharing tiles through L1 for short codes) is gone: behind on every shape
// (tools/proto_stream_read.hip keeps its load form as "rec" for the record).
constexpr int kSeqLoads = 4;                                 // 16-byte loads in flight per lane and tile (8 measured 2-4 % behind)
struct FilterChoice { bool mfma; int qt, qn; bool shrt; };
FilterChoice topk_filter_choice(int W, int64_t Q) {
    FilterChoice c{false, 0, 1, W < 4};
    static const int mfma_min_q = [] { const char* e = xmh_experiment_env("XMH_TOPK_MFMA"); return e ? atoi(e) : -1; }();
    if ((W == 4 || W == 8 || W == 16) && (mfma_min_q < 0 ? Q >= 5 : (mfma_min_q > 0 && Q >= mfma_min_q))) {
        const int qtmax = W == 16 ? 2 : 4;
        c.mfma = true;
        c.qt = Q <= 16 ? 1 : (Q <= 32 || qtmax == 2 ? 2 : 4);
    }
    c.qn = Q >= 5 ? 8 : (Q >= 3 ? 4 : (Q >= 2 ? 2 : 1));    // 4 query registers per query (pieces) / scalar registers (short codes)
    return c;
}

// the ternary fast path's streaming pass: per-piece filter with both planes (128 bits and more), per-item for 32 / 64 bits
template <int WW>
void launch_filter_tern(const uint32_t* qbits, const uint32_t* qzero, const uint32_t* rbits, const uint32_t* rzero, int pad, int64_t Q, int64_t R,
                        const FastWs& f, hipStream_t st) {
    const int qn = Q >= 5 ? 8 : (Q >= 3 ? 4 : (Q >= 2 ? 2 : 1));
    const unsigned gy = (unsigned)xmh::ceil_div(Q, qn);
    int64_t fb = (int64_t)xmh::device_cu_count() * 2;
    xmh::ProfScope prof("topk_filter", st);
    auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3((unsigned)fb, gy), dim3(kThreads), 0, st, qbits, qzero, rbits, rzero, pad, (int)Q, R,
                                                  (const uint32_t*)f.t_est, (const uint32_t*)f.bound, f.cnt, f.cand); };
    if constexpr (WW % 4 == 0) {
        const int64_t ft = xmh::ceil_div(R * (WW / 4), (int64_t)kThreads * kSeqLoads);
        if (fb > ft) fb = ft;
        if (qn == 1) go(k_topk_filter_seq<WW, kSeqLoads, 1, true>);
        if (qn == 2) go(k_topk_filter_seq<WW, kSeqLoads, 2, true>);
        if (qn == 4) go(k_topk_filter_seq<WW, kSeqLoads, 4, true>);
        if (qn == 8) go(k_topk_filter_seq<WW, kSeqLoads, 8, true>);
    } else {
        const int64_t ft = xmh::ceil_div(R, (int64_t)kThreads * kSeqLoads);
        if (fb > ft) fb = ft;
        if (qn == 1) go(k_topk_filter_item_tern<WW, kSeqLoads, 1>);
        if (qn == 2) go(k_topk_filter_item_tern<WW, kSeqLoads, 2>);
        if (qn == 4) go(k_topk_filter_item_tern<WW, kSeqLoads, 4>);
        if (qn == 8) go(k_topk_filter_item_tern<WW, kSeqLoads, 8>);
    }
}

int topk_call(const uint32_t* qbits, const uint32_t* qzero, const uint32_t* rbits, const uint32_t* rzero, int64_t Q, int64_t R, int K, int k,
              int64_t base_index, void* ws, size_t ws_bytes, uint16_t* dist, int32_t* idx, xmh_stream_t stream, bool prepared) {
    if ((qzero == nullptr) != (rzero == nullptr)) return xmh::fail(XMH_EINVAL, "xmh_hamming_topk: zero planes for both sides or neither");
    const bool tern = qzero != nullptr;
    TopkPlan p;
    int rc = plan_topk(Q, R, K, k, &p, tern);
    if (rc) return rc;
    const int pad = 32 * p.W - K;                         // padding bits: set in both zero planes, taken out of the half-unit distance
    if (!qbits || !rbits || !ws || !dist || !idx) return xmh::fail(XMH_EINVAL, "xmh_hamming_topk: null pointer");
    if (ws_bytes < p.ws_bytes) return xmh::fail(XMH_EINVAL, "xmh_hamming_topk: workspace too small (%zu < %zu)", ws_bytes, p.ws_bytes);
    int32_t* part_i = static_cast<int32_t*>(ws);
    uint16_t* part_d = reinterpret_cast<uint16_t*>(static_cast<char*>(ws) + (size_t)Q * p.nblocks * k * 4);
    hipStream_t st = xmh::as_stream(stream);
    char* wsb = static_cast<char*>(ws);
    FastWs f;
    f.hist = reinterpret_cast<uint32_t*>(wsb + p.off_hist);
    f.t_est = reinterpret_cast<uint32_t*>(wsb + p.off_test);
    f.bound = reinterpret_cast<uint32_t*>(wsb + p.off_bound);
    f.cnt = reinterpret_cast<uint32_t*>(wsb + p.off_cnt);
    f.fail = reinterpret_cast<int*>(wsb + p.off_fail);
    f.cand = reinterpret_cast<unsigned long long*>(wsb + p.off_cand);
    TopkCtl* ctl = reinterpret_cast<TopkCtl*>(wsb + p.off_ctl);
    const bool robust_only = getenv("XMH_TOPK_ROBUST_ONLY") != nullptr;   // test hook: skip the fast path
    const int* gate = nullptr;
    if (!robust_only) {
        // ---- fast path: sample -> threshold -> filter -> select (all stream-ordered, no host sync) ----
        // The control words and the sample histogram are zero on entry and left zero on exit (each consumer puts its word back):
        // a prepared workspace needs no memset launch.
        if (!prepared) XMH_HIP(hipMemsetAsync(wsb + p.off_ctl, 0, p.off_cand - p.off_ctl, st));
        const int fold_pick = Q <= kFoldPickQ;            // few queries: the last sample block picks the thresholds itself
        const int nb = p.nb;
        int sblocks = kSampleBlocks;
        int64_t stride = R / sblocks;
        // short codes: the sample grows with the gallery (2.6 % of it up to 80 M rows): at a fixed 262 144 rows the safety margin of the pick (+8
        // sampled hits) asked for 8 / fraction = 1 200 estimated candidates at 40 M rows, one bucket too far for short codes, where a
        // bucket more is 4-5 x the candidates and overflows the list (40 M x 32 bit, Q = 16: the robust path in most calls)
        // (codes of 128 bits and more have fine buckets and keep the fixed sample: at 256 bit, Q = 64 the larger one cost 6 % of the call)
        int per_block = kSamplePerBlock * (p.W >= 4 ? 1 : (int)(R / 10000000 < 1 ? 1 : (R / 10000000 > 8 ? 8 : R / 10000000)));
        bool exact = false;
        if (stride <= per_block) {                      // small gallery: the "sample" is the whole gallery
            sblocks = (int)xmh::ceil_div(R, per_block);
            stride = per_block;
            exact = true;
        }
        const double frac = exact ? 1.0 : (double)((int64_t)sblocks * per_block) / (double)R;
        const uint32_t target = exact ? (uint32_t)((int64_t)k < R ? (int64_t)k : R) : (uint32_t)(2.0 * k * frac + 8.0);
        const int sqg = nb <= 2049 ? 16 : (32768 / nb > 0 ? 32768 / nb : 1);      // 128 KB of histograms per sample block at most (ternary 1024 / 2048 bits)
        const size_t slds = (size_t)sqg * nb * 4;
        const PickParams pp{(float)(1.0 / frac), (uint32_t)k, (uint32_t)R, exact ? 1 : 0};
#define XMH_FAST(WW)                                                                                                       \
        {                                                                                                                  \
            const dim3 sgrid_(sblocks, fold_pick ? 1u : (unsigned)(xmh::ceil_div(Q, 16) < 4096 ? xmh::ceil_div(Q, 16) : 4096));       \
            xmh::RangeScope rs_("topk: sample + threshold pick");                                                         \
            if (tern)                                                                                                      \
                hipLaunchKernelGGL((k_topk_sample<WW, true>), sgrid_, dim3(kThreads), slds, st, qbits, qzero, rbits, rzero, pad, (int)Q, R, nb, stride, \
                                   per_block, f.hist, fold_pick, target, ctl, f.t_est, f.cnt, f.fail, pp, f.bound, sqg);   \
            else                                                                                                           \
                hipLaunchKernelGGL((k_topk_sample<WW, false>), sgrid_, dim3(kThreads), slds, st, qbits, qzero, rbits, rzero, pad, (int)Q, R, nb, stride, \
                                   per_block, f.hist, fold_pick, target, ctl, f.t_est, f.cnt, f.fail, pp, f.bound, sqg);   \
            if (!fold_pick)                                                                                                \
                hipLaunchKernelGGL(k_topk_pick, dim3((unsigned)Q), dim3(64), 0, st, f.hist, (int)Q, nb, target, f.t_est, f.cnt, f.fail, pp, f.bound); \
            rs_.end();                                                                                                     \
            XMH_RANGE("topk: streaming filter");                                                                           \
            if (tern) {                                           /* both planes: the per-piece / per-item ternary filters */      \
                launch_filter_tern<WW>(qbits, qzero, rbits, rzero, pad, Q, R, f, st);                                      \
                break;                                                                                                     \
            }                                                                                                              \
            const FilterChoice fc_ = topk_filter_choice(WW, Q);      /* one decision for the launch and for xmh_topk_describe */        \
            const int qn = fc_.qn;                                                                                         \
            bool on_mfma = false;                                                                                          \
            if constexpr (WW == 4 || WW == 8 || WW == 16) {                                                                 \
                if (fc_.mfma) {                                                                                            \
                    constexpr int QTMAX_ = WW == 16 ? 2 : 4;          /* B operands: 16 * QT * W / 8 registers */                  \
                    const int qt_ = fc_.qt;                                                                                \
                    const unsigned gy_ = (unsigned)xmh::ceil_div(Q, 16 * qt_);                                             \
                    int64_t fb_ = (int64_t)xmh::device_cu_count() * 8 / gy_;                                               \
                    if (fb_ < xmh::device_cu_count()) fb_ = xmh::device_cu_count();                                        \
                    const int64_t steps_ = xmh::ceil_div(R, (int64_t)64 * (kThreads / 64));                                 \
                    if (fb_ > steps_) fb_ = steps_;                                                                        \
                    xmh::ProfScope prof("topk_filter", st);                                                                \
                    auto gom_ = [&](auto kern_) { hipLaunchKernelGGL(kern_, dim3((unsigned)fb_, gy_), dim3(kThreads), 0, st, qbits, rbits, (int)Q, R, \
                                                                     (const uint32_t*)f.t_est, f.cnt, f.cand); };           \
                    if (qt_ == 1) gom_(k_topk_filter_mfma<WW, 1>);                                                          \
                    else if (qt_ == 2) gom_(k_topk_filter_mfma<WW, 2>);                                                     \
                    else if constexpr (QTMAX_ == 4) gom_(k_topk_filter_mfma<WW, 4>);                                        \
                    on_mfma = true;                                                                                        \
                }                                                                                                          \
            }                                                                                                              \
            if constexpr (WW % 4 == 0) {                                                                                   \
                if (!on_mfma) {                                     /* 16-byte pieces, non-temporal, two blocks per CU */         \
                    const unsigned gy_ = (unsigned)xmh::ceil_div(Q, qn);                                                   \
                    int64_t fb_ = (int64_t)xmh::device_cu_count() * 2;                                                     \
                    const int64_t ft_ = xmh::ceil_div(R * (WW / 4), (int64_t)kThreads * kSeqLoads);                        \
                    if (fb_ > ft_) fb_ = ft_;                                                                              \
                    xmh::ProfScope prof("topk_filter", st);                                                                \
                    auto gos_ = [&](auto kern_) { hipLaunchKernelGGL(kern_, dim3((unsigned)fb_, gy_), dim3(kThreads), 0, st, qbits, qzero, rbits, rzero, 0, (int)Q, R, \
                                                                     (const uint32_t*)f.t_est, (const uint32_t*)f.bound, f.cnt, f.cand); }; \
                    if (qn == 1) gos_(k_topk_filter_seq<WW, kSeqLoads, 1, false>);                                          \
                    if (qn == 2) gos_(k_topk_filter_seq<WW, kSeqLoads, 2, false>);                                          \
                    if (qn == 4) gos_(k_topk_filter_seq<WW, kSeqLoads, 4, false>);                                          \
                    if (qn == 8) gos_(k_topk_filter_seq<WW, kSeqLoads, 8, false>);                                          \
                }                                                                                                          \
            }                                                                                                              \
            if constexpr (WW < 4) {                                 /* 32- and 64-bit codes: several items per 16 bytes */          \
            {                                                                                                              \
                const unsigned gy_ = (unsigned)xmh::ceil_div(Q, qn);                                                       \
                int64_t fb_ = (int64_t)xmh::device_cu_count() * 2;                                                         \
                const int64_t ft_ = xmh::ceil_div(xmh::ceil_div(R * WW + 3, (int64_t)4), (int64_t)kThreads * kSeqLoads);   \
                if (fb_ > ft_) fb_ = ft_;                                                                                  \
                xmh::ProfScope prof("topk_filter", st);                                                                    \
                auto gos_ = [&](auto kern_) { hipLaunchKernelGGL(kern_, dim3((unsigned)fb_, gy_), dim3(kThreads), 0, st, qbits, rbits, (int)Q, R, \
                                                                 (const uint32_t*)f.t_est, (const uint32_t*)f.bound, f.cnt, f.cand); }; \
                if (qn == 1) gos_(k_topk_filter_short<WW, kSeqLoads, 1>);                                                   \
                if (qn == 2) gos_(k_topk_filter_short<WW, kSeqLoads, 2>);                                                   \
                if (qn == 4) gos_(k_topk_filter_short<WW, kSeqLoads, 4>);                                                   \
                if (qn == 8) gos_(k_topk_filter_short<WW, kSeqLoads, 8>);                                                   \
            }                                                                                                              \
            }                                                                                                              \
        }
        switch (p.W) {
            case 1: XMH_FAST(1) break;
            case 2: XMH_FAST(2) break;
            case 4: XMH_FAST(4) break;
            case 8: XMH_FAST(8) break;
            case 16: XMH_FAST(16) break;
            case 32: XMH_FAST(32) break;
            default: XMH_FAST(64) break;
        }
#undef XMH_FAST
        XMH_LAUNCH_CHECK("xmh_hamming_topk fast path");
        {
            XMH_RANGE("topk: select");
            auto kern = k_topk_select;
            const size_t sel_lds = (size_t)kCandCap * 8 + 1024 * 8 + (size_t)(p.nb > 2048 ? p.nb : 2048) * 4 + 64;
            rc = raise_lds(kern, sel_lds, "xmh_hamming_topk select");
            if (rc) return rc;
            hipLaunchKernelGGL(kern, dim3((unsigned)Q), dim3(kThreads), sel_lds, st, (const unsigned long long*)f.cand,
                               (const uint32_t*)f.cnt, R, k, p.nb, base_index, dist, idx, f.fail);
        }
        XMH_LAUNCH_CHECK("xmh_hamming_topk select");
        gate = f.fail;                                   // the robust kernels below run only if a query failed
    }
    XMH_RANGE("topk: robust path (gated on the fail flags)");
    const dim3 grid(p.nblocks, p.nqg);
    const size_t lds = p.L.bytes();
#define XMH_TOPK_LAUNCH(WW, II)                                                                                        \
    {                                                                                                                  \
        auto go_ = [&](auto kern) -> int {                                                                             \
            if (const int rc_ = raise_lds(kern, lds, "xmh_hamming_topk")) return rc_;                                   \
            hipLaunchKernelGGL(kern, grid, dim3(kThreads), lds, st, qbits, qzero, rbits, rzero, pad, (int)Q, R, k, p.L, p.tiles_per_block, \
                               p.nblocks, part_d, part_i, gate);                                                       \
            return XMH_OK;                                                                                             \
        };                                                                                                             \
        rc = tern ? go_(k_topk_stream<WW, II, true>) : go_(k_topk_stream<WW, II, false>);                              \
        if (rc) return rc;                                                                                             \
    }
    switch (p.W) {
        case 1: XMH_TOPK_LAUNCH(1, 8) break;
        case 2: XMH_TOPK_LAUNCH(2, 8) break;
        case 4: XMH_TOPK_LAUNCH(4, 8) break;
        case 8: XMH_TOPK_LAUNCH(8, 4) break;
        case 16: XMH_TOPK_LAUNCH(16, 1) break;
        case 32: XMH_TOPK_LAUNCH(32, 1) break;
        default: XMH_TOPK_LAUNCH(64, 1) break;
    }
#undef XMH_TOPK_LAUNCH
    XMH_LAUNCH_CHECK("xmh_hamming_topk stream");
    {
        auto kern = k_topk_merge<4>;
        const size_t ldsm = p.Lm.bytes();
        rc = raise_lds(kern, ldsm, "xmh_hamming_topk merge");
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)Q), dim3(kThreads), ldsm, st, part_d, part_i, p.nblocks, k, p.Lm, base_index, dist, idx, gate);
    }
    XMH_LAUNCH_CHECK("xmh_hamming_topk merge");
    return XMH_OK;
}
}  // namespace

extern "C" int xmh_hamming_topk(const uint32_t* qbits, const uint32_t* rbits, int64_t Q, int64_t R, int K, int k, int64_t base_index, void* ws,
                                size_t ws_bytes, uint16_t* dist, int32_t* idx, xmh_stream_t stream) {
    XMH_RANGE("xmh_hamming_topk");
    return topk_call(qbits, nullptr, rbits, nullptr, Q, R, K, k, base_index, ws, ws_bytes, dist, idx, stream, false);
}

extern "C" int xmh_hamming_topk_prepared(const uint32_t* qbits, const uint32_t* rbits, int64_t Q, int64_t R, int K, int k, int64_t base_index,
                                         void* ws, size_t ws_bytes, uint16_t* dist, int32_t* idx, xmh_stream_t stream) {
    XMH_RANGE("xmh_hamming_topk_prepared");
    return topk_call(qbits, nullptr, rbits, nullptr, Q, R, K, k, base_index, ws, ws_bytes, dist, idx, stream, true);
}

// ---- ternary codes (round 6): the same call with the zero planes; distances come back in HALF units (2 d = K - q.r, 0 ... 2K) ----
extern "C" size_t xmh_topk_ternary_ws_bytes(int64_t Q, int64_t R, int K, int k) {
    TopkPlan p;
    if (plan_topk(Q, R, K, k, &p, true) != XMH_OK) return 0;
    return p.ws_bytes;
}

extern "C" int xmh_topk_ternary_ws_init(int64_t Q, int64_t R, int K, int k, void* ws, size_t ws_bytes, xmh_stream_t stream) {
    TopkPlan p;
    if (const int rc = plan_topk(Q, R, K, k, &p, true)) return rc;
    if (!ws) return xmh::fail(XMH_EINVAL, "xmh_topk_ternary_ws_init: null workspace");
    if (ws_bytes < p.ws_bytes) return xmh::fail(XMH_EINVAL, "xmh_topk_ternary_ws_init: workspace too small (%zu < %zu)", ws_bytes, p.ws_bytes);
    XMH_HIP(hipMemsetAsync(static_cast<char*>(ws) + p.off_ctl, 0, p.off_cand - p.off_ctl, xmh::as_stream(stream)));
    return XMH_OK;
}

extern "C" int xmh_hamming_topk_ternary(const uint32_t* qbits, const uint32_t* qzero, const uint32_t* rbits, const uint32_t* rzero, int64_t Q,
                                        int64_t R, int K, int k, int64_t base_index, void* ws, size_t ws_bytes, int prepared, uint16_t* dist2,
                                        int32_t* idx, xmh_stream_t stream) {
    XMH_RANGE("xmh_hamming_topk_ternary");
    if (!qzero || !rzero) return xmh::fail(XMH_EINVAL, "xmh_hamming_topk_ternary: needs both zero planes (binary codes: xmh_hamming_topk)");
    return topk_call(qbits, qzero, rbits, rzero, Q, R, K, k, base_index, ws, ws_bytes, dist2, idx, stream, prepared != 0);
}

// ---------------------------------------------------------------------------------------------------
// host merge of per-shard lists (sharded retrieval): plain host code, a k-way merge with one cursor per shard
// -- world <= 8 comparisons per output slot instead of an argsort of [Q][world * k] 64-bit keys
// ---------------------------------------------------------------------------------------------------
// Diagnostics for the measurement harness, like xmh_scan_describe: "filter=<kernel instance>" of the fast path's streaming pass for
// this shape, spelled as rocprofv3 prints it, so that a profile row is matched by its exact name.
extern "C" int xmh_topk_describe(int64_t Q, int64_t R, int K, int k, char* out, size_t out_bytes) {
    TopkPlan p;
    if (const int rc = plan_topk(Q, R, K, k, &p)) return rc;
    if (!out || out_bytes < 64) return xmh::fail(XMH_EINVAL, "xmh_topk_describe: buffer too small");
    const FilterChoice c = topk_filter_choice(p.W, Q);
    if (c.mfma) snprintf(out, out_bytes, "filter=k_topk_filter_mfma<%d, %d>", p.W, c.qt);
    else if (c.shrt) snprintf(out, out_bytes, "filter=k_topk_filter_short<%d, %d, %d>", p.W, kSeqLoads, c.qn);
    else snprintf(out, out_bytes, "filter=k_topk_filter_seq<%d, %d, %d>", p.W, kSeqLoads, c.qn);
    return XMH_OK;
}

extern "C" size_t xmh_topk_record_bytes(int64_t Q, int k) {
    if (Q < 0 || k <= 0) return 0;
    return ((size_t)Q * (size_t)k * 6 + 3) & ~(size_t)3;
}

extern "C" int xmh_topk_merge_host(const void* gathered_host, int world, int64_t Q, int k, int32_t* dist_out, int32_t* idx_out) {
    XMH_RANGE("xmh_topk_merge_host");
    if (world <= 0 || world > 4096 || Q < 0 || k <= 0) return xmh::fail(XMH_EINVAL, "xmh_topk_merge_host: bad shape world=%d Q=%lld k=%d", world, (long long)Q, k);
    if (Q == 0) return XMH_OK;
    if (!gathered_host || !dist_out || !idx_out) return xmh::fail(XMH_EINVAL, "xmh_topk_merge_host: null pointer");
    const size_t rec = xmh_topk_record_bytes(Q, k);
    const char* base = static_cast<const char*>(gathered_host);
    std::vector<int> cur((size_t)world);
    for (int64_t q = 0; q < Q; ++q) {
        std::fill(cur.begin(), cur.end(), 0);
        for (int o = 0; o < k; ++o) {
            int best = -1;
            uint64_t best_key = ~0ull;
            for (int w = 0; w < world; ++w) {
                if (cur[w] >= k) continue;
                const int32_t* idx = reinterpret_cast<const int32_t*>(base + (size_t)w * rec) + q * k;
                const int32_t id = idx[cur[w]];
                if (id < 0) { cur[w] = k; continue; }                                  // unused slots end a list
                const uint16_t* dist = reinterpret_cast<const uint16_t*>(base + (size_t)w * rec + (size_t)Q * k * 4) + q * k;
                const uint64_t key = ((uint64_t)dist[cur[w]] << 32) | (uint32_t)id;
                if (key < best_key) { best_key = key; best = w; }
            }
            if (best < 0) { dist_out[q * k + o] = 0xFFFF; idx_out[q * k + o] = -1; continue; }
            dist_out[q * k + o] = (int32_t)(best_key >> 32);
            idx_out[q * k + o] = (int32_t)(uint32_t)best_key;
            ++cur[best];
        }
    }
    return XMH_OK;
}
