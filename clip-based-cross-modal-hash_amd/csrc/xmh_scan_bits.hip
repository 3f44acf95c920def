// Pass 1 of the fused ranking scan for 128- and 256-bit binary codes with the MFMA operands built IN REGISTERS from the packed
// bits (round 3; the boundary, the tables and pass 2 are those of xmh_scan.hip -- this file only replaces k_scan_hist_m there).
//
// k_scan_hist_m stages a pre-expanded int8 image of the gallery (8 bytes per code bit and label bit, built by k_scan_expand, 1 KB
// pieces through a two-deep LDS ring by LDS-DMA, two barriers per 64-item batch) and at 129 / 257 bucket rows per wave its LDS
// footprint leaves one block per CU.  Here nothing is staged: as in k_topk_filter_mfma (xmh_topk.hip), `word & (0x01010101 << p)`
// leaves bits p, p + 8, p + 16, p + 24 of a packed word each alone in its byte, worth 2^p there (p = 7 through
// `(word >> 1) & 0x40404040`), 9 VALU operations for 32 bits; the query side (B operand, built once per wave, kept in registers)
// carries the matching weight s_i * 64 / 2^p with s_i = 1 - 2 q_i, so every product is 64 s_i x_i and an accumulator started at
// (LDS address of this lane's bucket-0 counter) + 64 * popcount(q) ends as the ADDRESS of counter [distance][query] -- counter rows
// are 64 bytes (16 queries x u32), exactly the layout of k_scan_hist_m.  Lane (row = l & 15, quarter = l >> 4) loads the quarter
// `quarter` of the code words of its item straight from the packed gallery (one or two words), and the label word `quarter` (labels:
// up to 128 classes = 4 words = two label tiles; their B bytes are 64 / 2^p where the query has the label, so the label chain ends
// as 0x10000 + 64 * (common labels) and min(., 0x10001) is the counter increment (all << 16 | relevant)).  No operand image, no
// LDS ring, no barrier in the loop; LDS holds the counters only (K = 256: 66 KB per block of 4 query tiles).
// The pair cache (12-bit entries since round 6, distance << 1 | relevant, three dwords per lane and batch: xmh_common.h) is written for the cached pass 2, and the
// items of a 16-item group sit in the same C rows (row r <-> item 16 g + 4 (r & 3) + (r >> 2)), so pass 2 is unchanged.
// MFMA results in VGPRs (round 6 correction of a round-3 note).  A round-3 build of this file with -mllvm -amdgpu-mfma-vgpr-form=1 gave wrong
// distances by a few units at K = 256, and the note here blamed a VALU write onto a quad three wait states behind the MFMA that reads it as
// srcC.  That pattern is legal (three wait states is what a four-pass MFMA needs) and is in the shipped 512-bit ternary instance, whose blocks
// of 512 threads make hipcc choose the VGPR form by itself: its histograms equal an exact integer restatement cell for cell
// (tools/diag_bits_ternary.py).  What broke the round-3 build was the `ds_add_u32` ASM statement reading the MFMA result with no wait states in
// front of it -- in the AGPR form a v_accvgpr_read sat in between and hid it.  The add is a builtin now (below), and tools/isa_hazards.py
// rule R1 checks the distance on every instance at build time.
#include "xmh_common.h"
#include "xmh_scan_bits.h"

#ifndef XMH_HIST_B_GI
#define XMH_HIST_B_GI 2
#endif

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

// word -> the 8 operand registers of its two k tiles (register p: bits p, p + 8, p + 16, p + 24, each worth 2^p; p = 7 worth 64)
__device__ __forceinline__ void word_to_bytes(uint32_t w, v4i& lo, v4i& hi) {
#pragma unroll
    for (int p = 0; p < 4; ++p) lo[p] = (int)(w & (0x01010101u << p));
#pragma unroll
    for (int p = 4; p < 7; ++p) hi[p - 4] = (int)(w & (0x01010101u << p));
    hi[3] = (int)((w >> 1) & 0x40404040u);
}

// the matching B registers: byte j of register p <- bit p + 8 j of the word; on = value where the bit is set, off = where it is not,
// both scaled by 64 / 2^p (p = 7: 1)
__device__ __forceinline__ void word_to_weights(uint32_t w, int on, int off, v4i& lo, v4i& hi) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int mag = p < 7 ? (64 >> p) : 1;
        uint32_t b = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) b |= (uint32_t)(uint8_t)((((w >> (p + 8 * j)) & 1u) ? on : off) * mag) << (8 * j);
        if (p < 4) lo[p] = (int)b;
        else hi[p - 4] = (int)b;
    }
}

// TERN (round 6): codes in {-1, 0, +1} (sign_() left exact zeros: reference runners/base.py:407-410).  The i8 dot product takes them as
// they are: the item operand is the 2K-bit pair of planes [pos | neg] (pos = bits & ~zero: element is +1; neg = ~bits & ~zero: element is
// -1), built from the two packed planes as the lane loads them; the query weights are -q on the pos half and +q on the neg half, so the
// chain started at (bucket-0 counter) + 64 K ends at the counter of K - q.r -- the reference's distance in half units, 0 ... 2K (2K + 1
// bucket rows).  A lane's quarter of the operand lies wholly in one half (quarters 0, 1: pos; 2, 3: neg), so the planes cost the lane two
// loads and two ANDs per word instead of one load.  K <= 64 runs as NMC = 2 (a 128-bit operand), K <= 128 as NMC = 4, K <= 256 as NMC = 8.  Entries of the
// pair cache: 2 (K - q.r) | relevant in 16 bits, the layout of the 129 ... 256-bit binary codes: pass 2 reads them with the same kernels.
// Round 6 -- NQT: query tiles per wave: the item operand, the nine operations per word that build it and the loads behind them serve NQT
// tiles (NQT x GI independent MFMA chains in flight).  NSH: waves that SHARE a counter table -- wave (tiles, share) takes the batches share,
// share + NSH, ... of the chunk and adds into the same LDS counters (the adds are atomic).  The counters bound the residency of this kernel
// (257 rows x 64 B per tile: 66 KB per block of four tiles), so sharing them is what keeps two waves on a SIMD when a wave takes two tiles.
template <int NMC, int NW, int NSH, int NQT, bool CACHE, bool TERN>
__global__ __launch_bounds__(64 * NW * NSH / NQT) void k_scan_hist_b(xmh::ScanBitsArgs a, uint32_t* __restrict__ chunk_hist, uint4* __restrict__ pair_cache) {
    constexpr int LWC = NMC / 2;                                    // code words per lane (a quarter of the padded code)
    constexpr int WH = TERN ? NMC : 2 * NMC;                        // words of one plane half of the operand (TERN: 2 WH words in all)
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];   // NW x [nb][16] u32 counters
    const int b = blockIdx.x;
    const int qtile = (b >> 3) % a.nqt, chunk_id = (b & 7) + 8 * ((b >> 3) / a.nqt);        // as mfma_map_block (xmh_scan.hip)
    if (chunk_id >= a.nchunk) return;
    constexpr int NWV = NW / NQT;                                    // waves of one share; each takes NQT query tiles
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) % NWV, share = (threadIdx.x >> 6) / NWV;
    const int ql = lane & 15, slot = lane >> 4;
    const int ncell = a.nb * 16;
    uint32_t* cnt0 = lds + wave * NQT * ncell;                      // this wave's NQT counter tables, one after the other
    for (int e = lane + 64 * share; e < NQT * ncell; e += 64 * NSH) cnt0[e] = 0u;
    const bool negq = TERN && slot * LWC >= WH;                      // TERN: this lane's quarter lies in the neg half of the operand
    v4i bq[NQT][NMC], bl[NQT][2];
    int pcn[NQT], lanebase[NQT], cinit[NQT], qn[NQT];
    bool validn[NQT];
#pragma unroll
    for (int n = 0; n < NQT; ++n) {
        const int q = ((qtile * NW + wave * NQT + n) * 16) + ql;
        const bool valid = q < a.Q;
        qn[n] = q;
        validn[n] = valid;
        int pc = 0;
        if (TERN) pc = a.K;                                          // the chain starts at the row of K - 0
        else if (valid)
            for (int w = 0; w < a.W; ++w) pc += __popc(a.qbits[(int64_t)q * a.W + w]);
        pcn[n] = pc;
#pragma unroll
        for (int v = 0; v < LWC; ++v) {
            if constexpr (TERN) {
                const int wi = (slot * LWC + v) % WH;                // word of the plane
                const bool have = valid && wi < a.W;
                const uint32_t b = have ? a.qbits[(int64_t)q * a.W + wi] : 0u, z = have ? a.qzero[(int64_t)q * a.W + wi] : 0xffffffffu;
                const uint32_t qpos = b & ~z, qneg = ~b & ~z;        // (padding bits are set in the zero plane: neither)
                v4i p0, p1, n0, n1;
                word_to_weights(qpos, negq ? 1 : -1, 0, p0, p1);     // -q on the pos half, +q on the neg half
                word_to_weights(qneg, negq ? -1 : 1, 0, n0, n1);
                bq[n][2 * v] = p0 | n0;                              // disjoint bytes
                bq[n][2 * v + 1] = p1 | n1;
            } else {
                const int wi = slot * LWC + v;
                const uint32_t w = valid && wi < a.W ? a.qbits[(int64_t)q * a.W + wi] : 0u;
                word_to_weights(w, -1, valid && wi < a.W ? 1 : 0, bq[n][2 * v], bq[n][2 * v + 1]);      // no such word: zero weights, whatever the item lane loads
            }
        }
        word_to_weights(valid && slot < a.LW ? a.qlab[(int64_t)q * a.LW + slot] : 0u, 1, 0, bl[n][0], bl[n][1]);
        lanebase[n] = (int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)(cnt0 + n * ncell) + ql * 4;     // this lane's bucket-0 counter
        cinit[n] = lanebase[n] + 64 * pc;
    }
    const int64_t lo = (int64_t)chunk_id * a.chunk;
    const int64_t hi = (lo + a.chunk < a.R) ? lo + a.chunk : a.R;
    const int nbat = (int)((hi - lo + 63) >> 6);
    // A rows: row r of group g is item 16 g + 4 (r & 3) + (r >> 2) (k_scan_hist_m's image order: C register j of lane (slot, query)
    // is item 16 g + 4 j + slot, which is what the pair-cache layout below and the cached pass 2 count on)
    const int rowitem = 4 * (ql & 3) + (ql >> 2);
    uint32_t* crow[NQT];                                             // this lane's first 12-bit record (xmh_common.h) of each tile; the other one 32 records on
#pragma unroll
    for (int n = 0; n < NQT; ++n)
        crow[n] = CACHE ? reinterpret_cast<uint32_t*>(pair_cache) + (((int64_t)chunk_id * (a.qpad >> 3) + (qn[n] >> 3)) * ((a.chunk + 63) >> 6) * 64 + slot * 8 + (qn[n] & 7)) * xmh::kCache12Dwords : nullptr;
    uint32_t cur[4][LWC + 1], nxt[4][LWC + 1];
    // a word index past the end of the record is clamped to the last word: the query operand of that lane quarter is zero (word_to_weights of
    // a zero word with off = 0 for the labels; for the code see wq below), so what is loaded in its place counts for nothing
    int wic[LWC];
#pragma unroll
    for (int v = 0; v < LWC; ++v) {
        const int wi = TERN ? (slot * LWC + v) % WH : slot * LWC + v;
        wic[v] = wi < a.W ? wi : a.W - 1;
    }
    // TERN: bits and zero plane -> this lane's half of the operand
    auto plane = [&](uint32_t b, uint32_t z) -> uint32_t { return (negq ? ~b : b) & ~z; };
    const int wil = slot < a.LW ? slot : (a.LW > 0 ? a.LW - 1 : 0);
    auto load = [&](uint32_t (&dst)[4][LWC + 1], int i) {
        const int64_t first = lo + (int64_t)i * 64;
        if (first + 64 <= hi) {                                      // whole batch inside the chunk (wave-uniform): unconditional loads, constant strides
            const uint32_t* __restrict__ pc = a.rbits + (first + rowitem) * a.W;
            const uint32_t* __restrict__ pl = a.rlab + (first + rowitem) * a.LW + wil;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int v = 0; v < LWC; ++v) {
                    if constexpr (TERN) dst[g][v] = plane(pc[g * 16 * a.W + wic[v]], a.rzero[(first + rowitem + g * 16) * a.W + wic[v]]);
                    else dst[g][v] = pc[g * 16 * a.W + wic[v]];
                }
                dst[g][LWC] = pl[g * 16 * a.LW];
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t item = first + 16 * g + rowitem;
                const int64_t it = item < hi ? item : hi - 1;
                const uint32_t ok = item < hi ? 0xffffffffu : 0u;    // beyond the chunk: all-zero code, no labels (corrected below)
#pragma unroll
                for (int v = 0; v < LWC; ++v) {
                    if constexpr (TERN) dst[g][v] = plane(a.rbits[it * a.W + wic[v]], a.rzero[it * a.W + wic[v]]) & ok;      // beyond the chunk: both halves empty
                    else dst[g][v] = a.rbits[it * a.W + wic[v]] & ok;
                }
                dst[g][LWC] = a.rlab[it * a.LW + wil] & ok;
            }
        }
    };
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the zeroed counters are in place (this wave's own cells)
    if (NSH > 1) __syncthreads();                                  // ... and those the tile's other waves zeroed
    if (share < nbat) load(cur, share);
    for (int i = share; i < nbat; i += NSH) {
        if (i + NSH < nbat) load(nxt, i + NSH);
        uint32_t cw[NQT][3], cw2[NQT][3];                              // the two records this lane fills per tile and batch: 8 entries of 12 bits each
        // Round 6: the groups of 16 items go through the matrix pipe GI at a time, their chains interleaved MFMA by MFMA.  With 66 KB of
        // counters per block the kernel runs two waves per SIMD, and one group's chain (four dependent MFMAs + two for the labels) left
        // the pipe waiting on its own results; independent chains of the other groups fill those slots.
        constexpr int GI = XMH_HIST_B_GI;
#pragma unroll
        for (int g0 = 0; g0 < 4; g0 += GI) {
            v4i am[GI][NMC], al[GI][2];
#pragma unroll
            for (int u = 0; u < GI; ++u) {
#pragma unroll
                for (int v = 0; v < LWC; ++v) word_to_bytes(cur[g0 + u][v], am[u][2 * v], am[u][2 * v + 1]);
                word_to_bytes(cur[g0 + u][LWC], al[u][0], al[u][1]);
            }
            v4i acc[NQT][GI], lab[NQT][GI];
#pragma unroll
            for (int n = 0; n < NQT; ++n)
#pragma unroll
                for (int u = 0; u < GI; ++u) {
                    acc[n][u] = v4i{cinit[n], cinit[n], cinit[n], cinit[n]};
                    lab[n][u] = v4i{0x10000, 0x10000, 0x10000, 0x10000};
                }
#pragma unroll
            for (int m = 0; m < NMC; ++m) {
#pragma unroll
                for (int n = 0; n < NQT; ++n)
#pragma unroll
                    for (int u = 0; u < GI; ++u) acc[n][u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(am[u][m], bq[n][m], acc[n][u], 0, 0, 0);
                if (m < 2) {
#pragma unroll
                    for (int n = 0; n < NQT; ++n)
#pragma unroll
                        for (int u = 0; u < GI; ++u) lab[n][u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(al[u][m], bl[n][m], lab[n][u], 0, 0, 0);
                }
            }
#pragma unroll
            for (int n = 0; n < NQT; ++n)
#pragma unroll
                for (int u = 0; u < GI; ++u) {
                    const int g = g0 + u;
                    uint32_t e[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t inc = min((uint32_t)lab[n][u][j], 0x10001u);                    // all << 16 | relevant
                        // a builtin, not an asm statement: with 512 threads and more per block hipcc keeps the MFMA results in VGPRs, and only for an
                        // instruction it placed itself does it keep the wait states between the MFMA and this read of its result
                        __hip_atomic_fetch_add((__attribute__((address_space(3))) uint32_t*)(uintptr_t)(uint32_t)acc[n][u][j], inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (CACHE) e[j] = (inc & 1u) | ((uint32_t)(acc[n][u][j] - lanebase[n]) >> 5);    // entry: distance << 1 | relevant (16 bits)
                    }
                    if (CACHE) {                                     // steps 2g, 2g+1 of the two 8-slot lanes this lane feeds (k_scan_hist_m): entries 2g, 2g+1
                        auto put = [&](uint32_t (&d)[3], uint32_t x, uint32_t y) {      // of their records, appended 12 bits each (g is a constant after unrolling)
                            if (g == 0) d[0] = x | (y << 12);
                            else if (g == 1) { d[0] |= x << 24; d[1] = (x >> 8) | (y << 4); }
                            else if (g == 2) { d[1] |= (x << 16) | (y << 28); d[2] = y >> 4; }
                            else d[2] |= (x << 8) | (y << 20);
                        };
                        put(cw[n], e[0], e[2]);
                        put(cw2[n], e[1], e[3]);
                    }
                }
        }
        if (CACHE) {
#pragma unroll
            for (int n = 0; n < NQT; ++n) {
                uint32_t* dst = crow[n] + (int64_t)i * 64 * xmh::kCache12Dwords;
                __builtin_nontemporal_store(cw[n][0], dst);
                __builtin_nontemporal_store(cw[n][1], dst + 1);
                __builtin_nontemporal_store(cw[n][2], dst + 2);
                __builtin_nontemporal_store(cw2[n][0], dst + 32 * xmh::kCache12Dwords);
                __builtin_nontemporal_store(cw2[n][1], dst + 32 * xmh::kCache12Dwords + 1);
                __builtin_nontemporal_store(cw2[n][2], dst + 32 * xmh::kCache12Dwords + 2);
            }
        }
        if (i + NSH < nbat) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int v = 0; v <= LWC; ++v) cur[g][v] = nxt[g][v];
        }
    }
    // the padding items of a ragged last batch are all-zero-bit codes without labels: distance popcount(query) (TERN: empty planes, K - 0), never relevant
    const int npad = nbat * 64 - (int)(hi - lo);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (NSH > 1) __syncthreads();                                  // every wave of the tile has added its batches
#pragma unroll
    for (int n = 0; n < NQT; ++n)
        if (npad > 0 && slot == 0 && validn[n] && share == 0) cnt0[n * ncell + pcn[n] * 16 + ql] -= (uint32_t)npad << 16;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (NSH > 1) __syncthreads();
    uint32_t* __restrict__ out = chunk_hist + ((int64_t)chunk_id * a.nb) * a.qpad + (qtile * NW + wave * NQT) * 16;
    for (int e = lane + 64 * share; e < NQT * ncell; e += 64 * NSH) {
        const int n = e / ncell, c = e - n * ncell;
        out[(int64_t)(c >> 4) * a.qpad + n * 16 + (c & 15)] = cnt0[e];
    }
}

// (waves per query tile, query tiles per wave) of each instance, measured on configs[4]'s shard and at the COCO shape, A/B in one process
// on one box (profiles/r06_hist_b_share_ab.txt).  A block stays 4 query tiles = the plan's 64 queries; its waves = 4 NSH / NQT.
//   256-bit binary      (2, 2): pass 1 of the shard 5.01 -> 3.51 ms, the COCO shape 0.87 -> 0.77 ms per step
//   ternary <= 64 bits  (2, 2): 0.77 -> 0.62 ms per step;  ternary <= 128 bits (2, 2): 1.01 -> 0.83
//   ternary <= 256 bits (2, 1): 1.57 -> 1.39 (eight code tiles x two query tiles of B operands spill)
// Two tiles per wave halve the operand work per pair, two waves per table give the residency back.  One tile per wave with three waves
// per table gave 6 %, two tiles per wave alone LOST 6 % (one wave per SIMD).
template <int NMC, bool TERN>
struct HistBShape {
    static constexpr int NSH = 2;
    static constexpr int NQT = TERN && NMC >= 8 ? 1 : 2;
};

template <int NMC, int NW, int NSH, int NQT, bool TERN>
int launch_s(const xmh::ScanBitsArgs& a, uint32_t* chunk_hist, uint4* cache, hipStream_t st) {
    const size_t lds = (size_t)NW * a.nb * 16 * 4;
    const dim3 grid((unsigned)(8 * a.nqt * xmh::ceil_div(a.nchunk, 8)));
    xmh::ProfScope prof("scan_hist", st);
    if (cache) {
        auto kern = k_scan_hist_b<NMC, NW, NSH, NQT, true, TERN>;
        if (const int rc = xmh::raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds, "xmh_hamming_hist")) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(64 * NW * NSH / NQT), lds, st, a, chunk_hist, cache);
    } else {
        auto kern = k_scan_hist_b<NMC, NW, NSH, NQT, false, TERN>;
        if (const int rc = xmh::raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds, "xmh_hamming_hist")) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(64 * NW * NSH / NQT), lds, st, a, chunk_hist, cache);
    }
    return XMH_OK;
}

template <int NMC, int NW, bool TERN>
int launch_t(const xmh::ScanBitsArgs& a, uint32_t* chunk_hist, uint4* cache, hipStream_t st) {
#ifdef XMH_EXPERIMENTS                                             // XMH_HIST_B_NSH = 1 .. 4, XMH_HIST_B_NQT = 1, 2: every combination, for the A/B
    const char *e1 = xmh_experiment_env("XMH_HIST_B_NSH"), *e2 = xmh_experiment_env("XMH_HIST_B_NQT");
    if (e1 || e2) {
        const int nsh = e1 ? atoi(e1) : 1, nqt = e2 ? atoi(e2) : 1;
        switch (nsh * 10 + nqt) {
            case 11: return launch_s<NMC, NW, 1, 1, TERN>(a, chunk_hist, cache, st);
            case 21: return launch_s<NMC, NW, 2, 1, TERN>(a, chunk_hist, cache, st);
            case 31: return launch_s<NMC, NW, 3, 1, TERN>(a, chunk_hist, cache, st);
            case 12: return launch_s<NMC, NW, 1, 2, TERN>(a, chunk_hist, cache, st);
            case 22: return launch_s<NMC, NW, 2, 2, TERN>(a, chunk_hist, cache, st);
            case 42: return launch_s<NMC, NW, 4, 2, TERN>(a, chunk_hist, cache, st);
            case 32: return launch_s<NMC, NW, 3, 2, TERN>(a, chunk_hist, cache, st);
            default: return xmh::fail(XMH_EINVAL, "XMH_HIST_B_NSH=%d XMH_HIST_B_NQT=%d: no such instance", nsh, nqt);
        }
    }
#endif
    return launch_s<NMC, NW, HistBShape<NMC, TERN>::NSH, HistBShape<NMC, TERN>::NQT, TERN>(a, chunk_hist, cache, st);
}

}  // namespace

namespace xmh {

void scan_hist_bits_shape(int nmc, bool ternary, int* nsh, int* nqt) {
    *nsh = 2;
    *nqt = ternary && nmc >= 8 ? HistBShape<8, true>::NQT : HistBShape<4, false>::NQT;
    static_assert(HistBShape<2, true>::NQT == HistBShape<4, false>::NQT && HistBShape<4, true>::NQT == HistBShape<4, false>::NQT && HistBShape<8, true>::NSH == 2, "");
}

int launch_scan_hist_bits(const ScanBitsArgs& a, int nmc, uint32_t* chunk_hist, uint4* cache, hipStream_t st) {
    if (a.rzero) {
        if (!a.qzero) return fail(XMH_EINVAL, "xmh_hamming_hist: zero planes of both sides or neither");
        if (nmc == 2 && a.K <= 64) return launch_t<2, kScanBitsWaves, true>(a, chunk_hist, cache, st);
        if (nmc == 4 && a.K <= 128) return launch_t<4, kScanBitsWaves, true>(a, chunk_hist, cache, st);
        if (nmc == 8 && a.K <= 256) return launch_t<8, kScanBitsWaves, true>(a, chunk_hist, cache, st);      // 513 bucket rows: 131 KB of counters, one block per CU
        return fail(XMH_ENOTSUP, "xmh_hamming_hist: no ternary k_scan_hist_b instance for %d code tiles at K=%d", nmc, a.K);
    }
    if (nmc == 4) return launch_t<4, kScanBitsWaves, false>(a, chunk_hist, cache, st);
    return fail(XMH_ENOTSUP, "xmh_hamming_hist: no k_scan_hist_b instance for %d code tiles", nmc);
}

}  // namespace xmh
