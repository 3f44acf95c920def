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
// LDS ring, no barrier; LDS holds the counters only (K = 256: 66 KB per block of 4 waves, two blocks per CU).
// The pair cache (16-bit entries, distance << 1 | relevant) is written in k_scan_hist_m's layout for the cached pass 2, and the
// items of a 16-item group sit in the same C rows (row r <-> item 16 g + 4 (r & 3) + (r >> 2)), so pass 2 is unchanged.
// NOT compiled with -mllvm -amdgpu-mfma-vgpr-form=1 (no file of the library is): with the results in VGPRs hipcc re-materialises the label
// chain's start value (v_mov_b64 into the quad) three instructions behind the MFMA that still reads that quad as srcC, and the
// hardware takes the new value -- wrong distances by a few units for K = 256 (found with tools/diag_bits.py).  In the default
// AGPR form the start values go through v_accvgpr_write and the hazard table covers them.
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
template <int NMC, int NW, bool CACHE, bool TERN>
__global__ __launch_bounds__(64 * NW) void k_scan_hist_b(xmh::ScanBitsArgs a, uint32_t* __restrict__ chunk_hist, uint4* __restrict__ pair_cache) {
    constexpr int LWC = NMC / 2;                                    // code words per lane (a quarter of the padded code)
    constexpr int WH = TERN ? NMC : 2 * NMC;                        // words of one plane half of the operand (TERN: 2 WH words in all)
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];   // NW x [nb][16] u32 counters
    const int b = blockIdx.x;
    const int qtile = (b >> 3) % a.nqt, chunk_id = (b & 7) + 8 * ((b >> 3) / a.nqt);        // as mfma_map_block (xmh_scan.hip)
    if (chunk_id >= a.nchunk) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ql = lane & 15, slot = lane >> 4;
    const int q0 = (qtile * NW + wave) * 16;
    const int q = q0 + ql;
    const int ncell = a.nb * 16;
    uint32_t* cnt = lds + wave * ncell;
    for (int e = lane; e < ncell; e += 64) cnt[e] = 0u;
    const bool valid = q < a.Q;
    // B operands: this lane is column ql (its query) and k quarter `slot`
    v4i bq[NMC], bl[2];
    int pc = 0;
    const bool negq = TERN && slot * LWC >= WH;                      // TERN: this lane's quarter lies in the neg half of the operand
    if (TERN) pc = a.K;                                              // the chain starts at the row of K - 0
    else if (valid)
        for (int w = 0; w < a.W; ++w) pc += __popc(a.qbits[(int64_t)q * a.W + w]);
#pragma unroll
    for (int v = 0; v < LWC; ++v) {
        if constexpr (TERN) {
            const int wi = (slot * LWC + v) % WH;                    // word of the plane
            const bool have = valid && wi < a.W;
            const uint32_t b = have ? a.qbits[(int64_t)q * a.W + wi] : 0u, z = have ? a.qzero[(int64_t)q * a.W + wi] : 0xffffffffu;
            const uint32_t qpos = b & ~z, qneg = ~b & ~z;            // (padding bits are set in the zero plane: neither)
            v4i p0, p1, n0, n1;
            word_to_weights(qpos, negq ? 1 : -1, 0, p0, p1);         // -q on the pos half, +q on the neg half
            word_to_weights(qneg, negq ? -1 : 1, 0, n0, n1);
            bq[2 * v] = p0 | n0;                                     // disjoint bytes
            bq[2 * v + 1] = p1 | n1;
        } else {
            const int wi = slot * LWC + v;
            const uint32_t w = valid && wi < a.W ? a.qbits[(int64_t)q * a.W + wi] : 0u;
            word_to_weights(w, -1, valid && wi < a.W ? 1 : 0, bq[2 * v], bq[2 * v + 1]);      // no such word: zero weights, whatever the item lane loads
        }
    }
    word_to_weights(valid && slot < a.LW ? a.qlab[(int64_t)q * a.LW + slot] : 0u, 1, 0, bl[0], bl[1]);
    const int lanebase = (int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)cnt + ql * 4;     // this lane's bucket-0 counter
    const int cinit = lanebase + 64 * pc;
    const int64_t lo = (int64_t)chunk_id * a.chunk;
    const int64_t hi = (lo + a.chunk < a.R) ? lo + a.chunk : a.R;
    const int nbat = (int)((hi - lo + 63) >> 6);
    // A rows: row r of group g is item 16 g + 4 (r & 3) + (r >> 2) (k_scan_hist_m's image order: C register j of lane (slot, query)
    // is item 16 g + 4 j + slot, which is what the pair-cache layout below and the cached pass 2 count on)
    const int rowitem = 4 * (ql & 3) + (ql >> 2);
    uint4* crow = nullptr;
    if (CACHE) crow = pair_cache + ((int64_t)chunk_id * (a.qpad >> 3) + (q >> 3)) * ((a.chunk + 63) >> 6) * 64 + slot * 8 + (q & 7);
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
    if (nbat > 0) load(cur, 0);
    for (int i = 0; i < nbat; ++i) {
        if (i + 1 < nbat) load(nxt, i + 1);
        uint32_t cw[4], cw2[4];
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
            v4i acc[GI], lab[GI];
#pragma unroll
            for (int u = 0; u < GI; ++u) {
                acc[u] = v4i{cinit, cinit, cinit, cinit};
                lab[u] = v4i{0x10000, 0x10000, 0x10000, 0x10000};
            }
#pragma unroll
            for (int m = 0; m < NMC; ++m) {
#pragma unroll
                for (int u = 0; u < GI; ++u) acc[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(am[u][m], bq[m], acc[u], 0, 0, 0);
                if (m < 2) {
#pragma unroll
                    for (int u = 0; u < GI; ++u) lab[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(al[u][m], bl[m], lab[u], 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < GI; ++u) {
                const int g = g0 + u;
                uint32_t e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t inc = min((uint32_t)lab[u][j], 0x10001u);                    // all << 16 | relevant
                    asm volatile("ds_add_u32 %0, %1" ::"v"(acc[u][j]), "v"(inc) : "memory");
                    if (CACHE) e[j] = (inc & 1u) | ((uint32_t)(acc[u][j] - lanebase) >> 5);       // entry: distance << 1 | relevant (16 bits)
                }
                if (CACHE) {                                         // steps 2g, 2g+1 of the two 8-slot lanes this lane feeds (k_scan_hist_m)
                    cw[g] = e[0] | (e[2] << 16);
                    cw2[g] = e[1] | (e[3] << 16);
                }
            }
        }
        if (CACHE) {
            uint4* dst = crow + (int64_t)i * 64;
            __builtin_nontemporal_store(cw[0], &dst->x);
            __builtin_nontemporal_store(cw[1], &dst->y);
            __builtin_nontemporal_store(cw[2], &dst->z);
            __builtin_nontemporal_store(cw[3], &dst->w);
            __builtin_nontemporal_store(cw2[0], &dst[32].x);
            __builtin_nontemporal_store(cw2[1], &dst[32].y);
            __builtin_nontemporal_store(cw2[2], &dst[32].z);
            __builtin_nontemporal_store(cw2[3], &dst[32].w);
        }
        if (i + 1 < nbat) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int v = 0; v <= LWC; ++v) cur[g][v] = nxt[g][v];
        }
    }
    // the padding items of a ragged last batch are all-zero-bit codes without labels: distance popcount(query) (TERN: empty planes, K - 0), never relevant
    const int npad = nbat * 64 - (int)(hi - lo);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (npad > 0 && slot == 0 && valid) cnt[pc * 16 + ql] -= (uint32_t)npad << 16;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t* __restrict__ out = chunk_hist + ((int64_t)chunk_id * a.nb) * a.qpad + q0;
    for (int e = lane; e < ncell; e += 64) out[(int64_t)(e >> 4) * a.qpad + (e & 15)] = cnt[e];
}

template <int NMC, int NW, bool TERN>
int launch_t(const xmh::ScanBitsArgs& a, uint32_t* chunk_hist, uint4* cache, hipStream_t st) {
    const size_t lds = (size_t)NW * a.nb * 16 * 4;
    const dim3 grid((unsigned)(8 * a.nqt * xmh::ceil_div(a.nchunk, 8)));
    xmh::ProfScope prof("scan_hist", st);
    if (cache) {
        auto kern = k_scan_hist_b<NMC, NW, true, TERN>;
        if (const int rc = xmh::raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds, "xmh_hamming_hist")) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, st, a, chunk_hist, cache);
    } else {
        auto kern = k_scan_hist_b<NMC, NW, false, TERN>;
        if (const int rc = xmh::raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds, "xmh_hamming_hist")) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, st, a, chunk_hist, cache);
    }
    return XMH_OK;
}

}  // namespace

namespace xmh {

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
