// The MFMA-evaluated scan kernels of xmh_scan.hip and the few definitions they share with the VALU kernels (ScanArgs, the block -> (chunk,
// query tile) maps, the float-bit constants): k_scan_touch, k_scan_hist_r2, k_scan_hist_r2w, k_scan_ap_c, k_scan_ap_r2.  A header so that
// tools/proto_scan_ablate.hip can compile exactly this code, with its ablation macros, in seconds (xmh_scan.hip instantiates several
// hundred VALU kernels: two minutes); included by xmh_scan.hip inside its anonymous namespace scope -- nothing else includes it.
#pragma once
#include "xmh_common.h"
#include <type_traits>

namespace {

constexpr int kMaxChunk = 32768;   // u16 halves of the packed pass-1 counters must not overflow
constexpr uint32_t kF23 = 0x4B000000u;                  // bits of 2^23 as a float (k_scan_ap_c)
constexpr int64_t kFloatBitsMaxItems = (1ll << 23) - 3;   // largest gallery (all shards) whose ranks k_scan_ap_c's float-bit counters hold
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
    uint4* pair_cache;      // pass 1 -> pass 2: (distance << 1 | relevant) of every pair, see k_scan_hist_s; null = recompute
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

// ===================================================================================================
// MFMA-evaluated scan (binary codes of 33..128 bits in pass 1, at most 64 in pass 2; at most 128 classes).
//
// Hamming distance and label overlap of 16 gallery items x 16 queries are two i8 dot-product tiles -- literally what the
// reference computes, B1 @ B2^T and query_L @ retrieval_L^T (common/calc_utils.py:51-56, :72) -- so they go to
// v_mfma_i32_16x16x64_i8 instead of 8 VALU instructions per pair:
//   * code tile: item bytes +-1, query bytes -+SCALE (SCALE = bytes of one bucket row of the LDS counters); started from the lane's
//     counter base, the accumulator IS the LDS byte address of counter [distance][query]:  base + SCALE*K/2 - (SCALE/2)*dot;
//   * label tile: item bytes 127, query bytes 64; pass 1 starts it at 1 so that min(acc, 1 + 8128) is the add operand
//     1 + relevant * 8128 (counters hold all + relevant * 8128; a chunk has at most 8064 items), pass 2 takes min(acc, 1).
// Per pair the VALU does ONE instruction in pass 1 (v_min) and the credit arithmetic in pass 2.
// Geometry = the slotted scheme with S = 4: lane = slot * 16 + query, and MFMA row 4*slot + j of a 16-item group holds item
// 4*j + slot, so accumulator register j of a lane is its step j and same-query lanes of one LDS instruction are consecutive
// items in lane order -- exactly what pass 2's returning adds need (lane_order_ok).
// Operand images (built per call by two small kernels, in the workspace): gallery [64-item batch][16-item group][tile m][lane][16 B],
// i.e. every MFMA A operand is one contiguous KB = one global_load_lds piece and one conflict-free ds_read_b128 per lane; the
// NW waves of a block (NW * 16 queries) share each staged batch (2-deep ring, LDS-DMA issued one batch ahead).
// The LDS atomics are inline asm: hipcc would drain the LDS-DMA (vmcnt(0)) before any LDS atomic it cannot prove disjoint from
// the ring.  Returning adds are waited for with counted lgkmcnt statements naming their destinations (LDS returns in order).
// ===================================================================================================
typedef int v4i __attribute__((ext_vector_type(4)));

struct MfmaArgs {
    const uint32_t* qbits;
    int Q, R, K, W;
    int chunk, nchunk, nqt, nb, qpad;
    // the packed words the operands are built from
    const uint32_t* rbits = nullptr;
    const uint32_t* rlab = nullptr;
    const uint32_t* qlab = nullptr;
    int LW = 0;
};

__device__ __forceinline__ bool mfma_map_block(const MfmaArgs& a, int& chunk_id, int& qtile) {
    const int b = blockIdx.x;
    const int xcd = b & 7, t = b >> 3;
    qtile = t % a.nqt;
    chunk_id = xcd + 8 * (t / a.nqt);
    return chunk_id < a.nchunk;
}

// ---------------------------------------------------------------------------------------------------
// k_scan_hist_r2: pass 1 for binary codes of at most 64 bits (rounds 3-4).  The i8 MFMA emits the LDS address of counter
// [distance][query]; everything around it is arranged so that the VALU does two instructions per pair:
//   * VGPR-form MFMAs (inline asm): results are consumed where they land, no v_accvgpr_read.  Every instruction that touches
//     an MFMA result is asm volatile in program order, software-pipelined one (item group, query group) behind its MFMAs, so
//     the MFMA -> VALU / DS read hazard (8 wait states, hipcc inserts them only for its own instructions) is covered by the 12
//     consumer instructions of the previous group that sit in between.
//   * the pair-cache byte comes out of the matrix pipe too: a second code chain with query bytes -+2 IS 2 * distance, and the label
//     chain counts into (all << 16 | relevant) counters -- started at 0x10000, min(acc, 0x10001) is the add operand and its low byte the
//     relevance bit -- so one SDWA OR per pair (byte0(2d) | byte0(inc) written to byte j of the cache word) assembles the entry.
//   * the A operands are built in registers from the packed gallery words (round 4): no operand image, no LDS-DMA, no ring, no
//     barrier.  Lane (row, slot) loads the word of its item that holds its 16 bits (fetched one batch ahead) and spreads it with 5-6
//     VALU operations per tile; waves are independent, LDS holds the counters only.
//   * NQ query groups of 16 per wave: every tile built feeds NQ MFMA groups.
// Counter rows are 64 bytes (16 queries x u32); chunks hold up to 32768 items (16-bit halves).  Round 3's form of the same statements
// (k_scan_hist_m2: operand images through a 3-deep LDS ring) and round 2's k_scan_hist_m were removed in round 5 (DESIGN 3.1 keeps their
// measurements).
// ---------------------------------------------------------------------------------------------------
template <int NML, int NW, int NQ, bool CACHE>
__device__ __forceinline__ void scan_hist_r2_body(const MfmaArgs& a, uint32_t* __restrict__ chunk_hist, uint4* __restrict__ pair_cache) {
    constexpr int NMI = 1 + NML, NMQ = 2 + NML;
    constexpr bool REGS = true;                                      // (the operands come from the packed words; kept as a name in the expressions below)
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];   // NW x NQ x [nb][16] u32 counters
    int chunk_id, qtile;
    if (!mfma_map_block(a, chunk_id, qtile)) return;                 // a.nqt counts tiles of NW * NQ * 16 queries here
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ql = lane & 15, slot = lane >> 4;
    const int t16 = (qtile * NW + wave) * NQ;                        // first 16-query tile of this wave
    const int ncell = a.nb * 16;
    uint32_t* cnt = lds + (wave * NQ) * ncell;
    for (int e = lane; e < NQ * ncell; e += 64) cnt[e] = 0u;
    v4i bq[NQ][NMQ], cq[NQ], kqv[REGS ? NQ : 1];                    // the image kernels start every 2 * distance chain at K: one quad
    bool valid[NQ];
    // REGS: byte b of operand register j of lane (row, slot) stands for bit 4 (slot & 1) + j + 8 b of word slot >> 1 (of the code, or of the
    // 64 label bits of a tile): (word >> (4 (slot & 1) + j)) & 0x01010101 leaves those four bits each alone in its byte, worth 1.  Item bytes
    // are therefore 0 / 1 (not -+1): distance = popcount(q) + sum x_i (1 - 2 q_i), so the query bytes are +-64 (counter rows of 64 bytes) for
    // the address chain, +-2 for the 2 * distance chain, and the chains start at 64 popcount(q) / 2 popcount(q) more.
    const int rsh = 4 * (slot & 1), rwi = slot >> 1;
#pragma unroll
    for (int h = 0; h < NQ; ++h) {
        valid[h] = (t16 + h) * 16 + ql < a.Q;
        int pcq = 0;
        {
            const int64_t q = (int64_t)(t16 + h) * 16 + ql;
            uint32_t qw = 0u;
            if (valid[h]) {
                for (int w = 0; w < a.W; ++w) pcq += __popc(a.qbits[q * a.W + w]);
                if (rwi < a.W) qw = a.qbits[q * a.W + rwi];
            }
            qw >>= rsh;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t t = (qw >> j) & 0x01010101u;
                const bool on = valid[h] && rwi < a.W;                 // no such code word (codes of at most 32 bits): zero operand, whatever the item lane holds
                // registers 1 and 3 of an item tile hold their bits worth 2 (see build()): their query bytes are halved, the products stay +-64 / +-2
                if (j & 1) {
                    bq[h][0][j] = on ? (int)(0x20202020u ^ (t * 0xc0u)) : 0;     // +32 / -32 (0xe0)
                    bq[h][1][j] = on ? (int)(0x01010101u ^ (t * 0xfeu)) : 0;     // +1 / -1
                } else {
                    bq[h][0][j] = on ? (int)(0x40404040u ^ (t << 7)) : 0;        // +64 / -64 (0xc0)
                    bq[h][1][j] = on ? (int)(0x02020202u ^ (t * 0xfcu)) : 0;     // +2 / -2 (0xfe)
                }
            }
#pragma unroll
            for (int m = 0; m < NML; ++m) {
                uint32_t lw = 0u;
                if (valid[h] && 2 * m + rwi < a.LW) lw = a.qlab[q * a.LW + 2 * m + rwi];
                lw >>= rsh;
#pragma unroll
                for (int j = 0; j < 4; ++j) bq[h][2 + m][j] = (int)((lw >> j) & 0x01010101u);
            }
            const int k2 = 2 * pcq;
            kqv[h] = v4i{k2, k2, k2, k2};
        }
        const int c0 = (int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)(cnt + h * ncell) + ql * 4 + (valid[h] ? 64 * pcq : 0);
        cq[h] = v4i{c0, c0, c0, c0};
    }
    v4i lab0 = {0x10000, 0x10000, 0x10000, 0x10000};
    asm volatile("" : "+v"(lab0));                                  // opaque: kept in VGPRs, not re-materialised from SGPRs inside the loop
#pragma unroll
    for (int h = 0; h < NQ; ++h) asm volatile("" : "+v"(cq[h]));
#pragma unroll
    for (int h = 0; h < (REGS ? NQ : 1); ++h) asm volatile("" : "+v"(kqv[h]));
    const int64_t lo = (int64_t)chunk_id * a.chunk;
    const int64_t hi = (lo + a.chunk < a.R) ? lo + a.chunk : a.R;
    const int nbat = (int)((hi - lo + 63) >> 6);
    const int64_t bat0 = lo >> 6;                                    // chunks start on 64-item boundaries
    uint4* crow[NQ];
#pragma unroll
    for (int h = 0; h < NQ; ++h)
        crow[h] = CACHE ? pair_cache + ((int64_t)chunk_id * (a.qpad >> 4) + (t16 + h)) * ((a.chunk + 63) >> 6) * 64 + lane : nullptr;
    // REGS: the packed words of this lane's four items of a batch (row r of group g is item 16 g + 4 (r & 3) + (r >> 2): accumulator register j
    // of a lane is its step j, as in the image), fetched one batch ahead with unconditional loads (clamped item, word index clamped to
    // the last one and masked: no predication, so hipcc waits with counted vmcnt only where the words are used; two batches ahead measured
    // the same 0.185 ms)
    uint32_t wcur[4][NMI], wnxt[4][NMI];
    const int ritem = 4 * (lane & 3) + ((lane & 15) >> 2);
    // a word index past the end of the record is clamped to the last word: the query operand of that lane group is zero (above), so the
    // bits loaded in its place count for nothing
    const int wi_c = rwi < a.W ? rwi : a.W - 1;
    int wi_l[NML];
#pragma unroll
    for (int m = 0; m < NML; ++m) wi_l[m] = 2 * m + rwi < a.LW ? 2 * m + rwi : (a.LW > 0 ? a.LW - 1 : 0);
    auto load_words = [&](int64_t batch, uint32_t (&w)[4][NMI]) {
        const int64_t first = batch * 64;
        if (first + 64 <= (int64_t)a.R) {                             // whole batch inside the gallery (wave-uniform): one address per array, constant strides
            const uint32_t* __restrict__ pc = a.rbits + (first + ritem) * a.W + wi_c;
            const uint32_t* __restrict__ pl = a.rlab + (first + ritem) * a.LW;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                w[g][0] = pc[g * 16 * a.W];
#pragma unroll
                for (int m = 0; m < NML; ++m) w[g][1 + m] = pl[g * 16 * a.LW + wi_l[m]];
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t item = first + g * 16 + ritem;
                const int64_t it = item < a.R ? item : (int64_t)a.R - 1;
                const uint32_t ok = item < a.R ? 0xffffffffu : 0u;   // items past the end: all-zero codes, no labels (the epilogue takes them out again)
                w[g][0] = a.rbits[it * a.W + wi_c] & ok;
#pragma unroll
                for (int m = 0; m < NML; ++m) w[g][1 + m] = a.rlab[it * a.LW + wi_l[m]] & ok;
            }
        }
    };
    // 6 operations per code tile: two shifts bring bits 0, 1 and bits 2, 3 of the lane's nibble to the bottom of their bytes, four masks
    // pick them -- the odd ones worth 2 where they stand, which the halved query bytes above make up for (the products of both chains have
    // to be exact: they are an address and a cache byte).  The label tiles only have to tell an overlap from none, so their bits stay where
    // one shift leaves them: y & (0x01010101 << j) is worth 2^j (1, 2, 4, 8) in its byte, the sum over the common labels is positive
    // exactly when there is one, and min(0x10000 + sum, 0x10001) is the add operand as before -- 5 operations.
    auto build = [&](v4i (&At)[NMI], const uint32_t (&w)[NMI]) {
        const uint32_t xa = w[0] >> rsh, xb = w[0] >> (rsh + 2);
        At[0][0] = (int)(xa & 0x01010101u);
        At[0][1] = (int)(xa & 0x02020202u);
        At[0][2] = (int)(xb & 0x01010101u);
        At[0][3] = (int)(xb & 0x02020202u);
#pragma unroll
        for (int m = 1; m < NMI; ++m) {
            const uint32_t y = w[m] >> rsh;
#pragma unroll
            for (int j = 0; j < 4; ++j) At[m][j] = (int)(y & (0x01010101u << j));
        }
    };
    load_words(bat0, wcur);
#ifdef XMH_ABL_NOBUILD
    v4i A[4][NMI];
#endif
    for (int i = 0; i < nbat; ++i) {
#ifdef XMH_ABL_NOLOADW
        if (i == 0) load_words(bat0, wnxt);
#else
        load_words(bat0 + (i + 1 < nbat ? i + 1 : i), wnxt);
#endif
        // A tiles of item group g live in set g & 1; read (asm: hipcc would wait for ALL outstanding reads at the first use) one group
        // ahead of the MFMAs that use them, waited for with counted lgkmcnt (LDS operations of a wave complete in order):
        //   R(0) R(1) | group 0 | R(2) | group 1 | R(3) | group 2 | group 3      with 4 adds per statement behind the first
#ifndef XMH_ABL_NOBUILD
        v4i A[REGS ? 4 : 2][NMI];                                    // REGS: one set per group, each kept alive one statement past its last MFMA
#endif
        uint32_t cw[NQ][4];
        v4i addr_p, d2_p, lab_p;                                     // results of the previous (group, query group), consumed by the next statement
        // One asm statement = the MFMAs of (group g, query group h) with the consumer instructions of the PREVIOUS pair between them:
        // an MFMA occupies the matrix pipe for 16 cycles, the three VALU / DS instructions behind it issue meanwhile, so a wave
        // keeps the pipe busy by itself (four MFMAs then twelve consumers left it idle half the time: 1830 cycles per batch measured
        // against 512 of MFMA work per wave).  Inside a statement the hazards are handled by hand (hipcc does not see them):
        //   * MFMA result -> VALU / DS read needs 8 wait states: the consumers read the PREVIOUS statement's results;
        //   * the SDWA byte inserts into w have a dst_sel forwarding hazard (one wait state): an add or an MFMA sits between them;
        //   * a VALU write directly in front of an MFMA was read stale as srcC (seen on hardware with a v_mov_b64 hipcc had placed
        //     there): every statement opens with s_nop 3, and the constant accumulator quads are opaque to hipcc (no re-materialising);
        //   * a VALU write landing on the A / B registers of an MFMA issued two or three instructions earlier corrupted its operand:
        //     all results are early-clobber outputs of the statement that also names the A tiles as inputs;
        //   * MFMA -> MFMA srcC dependencies are interlocked in hardware.
        // The consumers: inc = min(label overlap chain, 0x10001) in place; cache byte j = byte0(2 * distance) | byte0(inc); the add.
// ablation switches of tools/proto_scan_ablate.hip (-DXMH_ABL_NOADD: the LDS adds become s_nop; -DXMH_ABL_NOMFMA: the MFMAs do; results are
// wrong then, only the time means something): never set in the library build
// wait states the statements carry by hand (XMH_R2_NOPS: tools/proto_scan_ablate.hip prices the alternatives; tools/isa_hazards.py R1-R3 gates them)
#ifndef XMH_R2_NOPS
#define XMH_R2_NOPS 0
#endif
#if XMH_R2_NOPS == 0
#define XMH_R2_OPEN "s_nop 3\n\t"
#define XMH_R2_EVAL_CLOSE "s_nop 7"
#define XMH_R2_TAIL "s_nop 7\n\ts_nop 3"
#elif XMH_R2_NOPS == 1
#define XMH_R2_OPEN "s_nop 1\n\t"
#define XMH_R2_EVAL_CLOSE "s_nop 7"
#define XMH_R2_TAIL "s_nop 7\n\ts_nop 3"
#elif XMH_R2_NOPS == 2
#define XMH_R2_OPEN ""
#define XMH_R2_EVAL_CLOSE "s_nop 7"
#define XMH_R2_TAIL "s_nop 7\n\ts_nop 3"
#elif XMH_R2_NOPS == 3
#define XMH_R2_OPEN "s_nop 1\n\t"
#define XMH_R2_EVAL_CLOSE "s_nop 3"
#define XMH_R2_TAIL "s_nop 1"
#else
#define XMH_R2_OPEN ""
#define XMH_R2_EVAL_CLOSE "s_nop 3"
#define XMH_R2_TAIL "s_nop 1"
#endif
#ifdef XMH_ABL_NOMIN
#define XMH_VMIN "; "                                                // (ablation: the instruction becomes an assembler comment)
#else
#define XMH_VMIN "v_min_u32 "
#endif
#ifdef XMH_ABL_MINFULL
#undef XMH_VMIN
#define XMH_VMIN "v_and_b32 "                                        // (ablation: a full-rate instruction of the same shape)
#endif
#ifdef XMH_ABL_NOADD
#define XMH_ADD(A, D) "s_nop 0\n\t"
#else
#define XMH_ADD(A, D) "ds_add_u32 " A ", " D "\n\t"
#endif
#ifdef XMH_ABL_NOMFMA
#define XMH_MFMA(D, A, B, C) "s_nop 0\n\t"
#else
#define XMH_MFMA(D, A, B, C) "v_mfma_i32_16x16x64_i8 " D ", " A ", " B ", " C "\n\t"
#endif
#define XMH_SDWA(J) "dst_sel:BYTE_" #J " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0\n\t"
        auto fused = [&](const v4i (&At)[NMI], int h, v4i& addr, v4i& d2, v4i& lab, uint32_t& w) {
            uint32_t i0, i1, i2, i3;                                  // min results: fresh registers (in-place on the MFMA's result tuple made hipcc copy them)
            if (NML == 2 && CACHE) {
                asm volatile(
                    XMH_R2_OPEN XMH_MFMA("%0", "%8", "%9", "%10")
                    XMH_VMIN "%4, 0x10001, %26\n\t" XMH_VMIN "%5, 0x10001, %27\n\t" XMH_VMIN "%6, 0x10001, %28\n\t"
                    XMH_MFMA("%0", "%11", "%12", "%0")
                    XMH_VMIN "%7, 0x10001, %29\n\t"
                    "v_or_b32_sdwa %3, %18, %4 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0\n\t"
                    XMH_ADD("%22", "%4")
                    XMH_MFMA("%1", "%13", "%14", "%15")
                    "v_or_b32_sdwa %3, %19, %5 " XMH_SDWA(1) XMH_ADD("%23", "%5") "v_or_b32_sdwa %3, %20, %6 " XMH_SDWA(2)
                    XMH_MFMA("%2", "%13", "%16", "%17")
                    XMH_ADD("%24", "%6") "v_or_b32_sdwa %3, %21, %7 " XMH_SDWA(3) XMH_ADD("%25", "%7")
                    : "=&v"(lab), "=&v"(addr), "=&v"(d2), "=&v"(w), "=&v"(i0), "=&v"(i1), "=&v"(i2), "=&v"(i3)
                    : "v"(At[1]), "v"(bq[h][2]), "v"(lab0), "v"(At[NMI - 1]), "v"(bq[h][NMQ - 1]), "v"(At[0]), "v"(bq[h][0]), "v"(cq[h]), "v"(bq[h][1]), "v"(kqv[REGS ? h : 0]),
                      "v"(d2_p[0]), "v"(d2_p[1]), "v"(d2_p[2]), "v"(d2_p[3]), "v"(addr_p[0]), "v"(addr_p[1]), "v"(addr_p[2]), "v"(addr_p[3]),
                      "v"(lab_p[0]), "v"(lab_p[1]), "v"(lab_p[2]), "v"(lab_p[3])
                    : "memory");
            } else if (NML == 2) {
                asm volatile(
                    XMH_R2_OPEN XMH_MFMA("%0", "%6", "%7", "%8")
                    XMH_VMIN "%2, 0x10001, %18\n\t" XMH_VMIN "%3, 0x10001, %19\n\t" XMH_VMIN "%4, 0x10001, %20\n\t"
                    XMH_MFMA("%0", "%9", "%10", "%0")
                    XMH_VMIN "%5, 0x10001, %21\n\t" XMH_ADD("%14", "%2") XMH_ADD("%15", "%3")
                    XMH_MFMA("%1", "%11", "%12", "%13")
                    XMH_ADD("%16", "%4") XMH_ADD("%17", "%5")
                    : "=&v"(lab), "=&v"(addr), "=&v"(i0), "=&v"(i1), "=&v"(i2), "=&v"(i3)
                    : "v"(At[1]), "v"(bq[h][2]), "v"(lab0), "v"(At[NMI - 1]), "v"(bq[h][NMQ - 1]), "v"(At[0]), "v"(bq[h][0]), "v"(cq[h]),
                      "v"(addr_p[0]), "v"(addr_p[1]), "v"(addr_p[2]), "v"(addr_p[3]), "v"(lab_p[0]), "v"(lab_p[1]), "v"(lab_p[2]), "v"(lab_p[3])
                    : "memory");
            } else if (CACHE) {
                asm volatile(
                    XMH_R2_OPEN XMH_MFMA("%0", "%8", "%9", "%10")
                    XMH_VMIN "%4, 0x10001, %24\n\t" XMH_VMIN "%5, 0x10001, %25\n\t" XMH_VMIN "%6, 0x10001, %26\n\t" XMH_VMIN "%7, 0x10001, %27\n\t"
                    "v_or_b32_sdwa %3, %16, %4 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0\n\t"
                    XMH_ADD("%20", "%4")
                    XMH_MFMA("%1", "%11", "%12", "%13")
                    "v_or_b32_sdwa %3, %17, %5 " XMH_SDWA(1) XMH_ADD("%21", "%5") "v_or_b32_sdwa %3, %18, %6 " XMH_SDWA(2)
                    XMH_MFMA("%2", "%11", "%14", "%15")
                    XMH_ADD("%22", "%6") "v_or_b32_sdwa %3, %19, %7 " XMH_SDWA(3) XMH_ADD("%23", "%7")
                    : "=&v"(lab), "=&v"(addr), "=&v"(d2), "=&v"(w), "=&v"(i0), "=&v"(i1), "=&v"(i2), "=&v"(i3)
                    : "v"(At[1]), "v"(bq[h][2]), "v"(lab0), "v"(At[0]), "v"(bq[h][0]), "v"(cq[h]), "v"(bq[h][1]), "v"(kqv[REGS ? h : 0]),
                      "v"(d2_p[0]), "v"(d2_p[1]), "v"(d2_p[2]), "v"(d2_p[3]), "v"(addr_p[0]), "v"(addr_p[1]), "v"(addr_p[2]), "v"(addr_p[3]),
                      "v"(lab_p[0]), "v"(lab_p[1]), "v"(lab_p[2]), "v"(lab_p[3])
                    : "memory");
            } else {
                asm volatile(
                    XMH_R2_OPEN XMH_MFMA("%0", "%6", "%7", "%8")
                    XMH_VMIN "%2, 0x10001, %16\n\t" XMH_VMIN "%3, 0x10001, %17\n\t" XMH_VMIN "%4, 0x10001, %18\n\t" XMH_VMIN "%5, 0x10001, %19\n\t"
                    XMH_MFMA("%1", "%9", "%10", "%11")
                    XMH_ADD("%12", "%2") XMH_ADD("%13", "%3") XMH_ADD("%14", "%4") XMH_ADD("%15", "%5")
                    : "=&v"(lab), "=&v"(addr), "=&v"(i0), "=&v"(i1), "=&v"(i2), "=&v"(i3)
                    : "v"(At[1]), "v"(bq[h][2]), "v"(lab0), "v"(At[0]), "v"(bq[h][0]), "v"(cq[h]),
                      "v"(addr_p[0]), "v"(addr_p[1]), "v"(addr_p[2]), "v"(addr_p[3]), "v"(lab_p[0]), "v"(lab_p[1]), "v"(lab_p[2]), "v"(lab_p[3])
                    : "memory");
            }
        };
        // the first statement of a batch: nothing to consume yet.  Closed by 8 wait states: the next statement's consumers (and any
        // copy hipcc places in front of it) read these results, and its own MFMA is only one slot away
        auto evaluate = [&](const v4i (&At)[NMI], int h, v4i& addr, v4i& d2, v4i& lab) {
            if (NML == 2 && CACHE) {
                asm volatile(XMH_R2_OPEN XMH_MFMA("%0", "%3", "%4", "%5") XMH_MFMA("%0", "%6", "%7", "%0")
                             XMH_MFMA("%1", "%8", "%9", "%10") XMH_MFMA("%2", "%8", "%11", "%12") XMH_R2_EVAL_CLOSE
                             : "=&v"(lab), "=&v"(addr), "=&v"(d2)
                             : "v"(At[1]), "v"(bq[h][2]), "v"(lab0), "v"(At[NMI - 1]), "v"(bq[h][NMQ - 1]), "v"(At[0]), "v"(bq[h][0]), "v"(cq[h]),
                               "v"(bq[h][1]), "v"(kqv[REGS ? h : 0]));
            } else if (NML == 2) {
                asm volatile(XMH_R2_OPEN XMH_MFMA("%0", "%2", "%3", "%4") XMH_MFMA("%0", "%5", "%6", "%0")
                             XMH_MFMA("%1", "%7", "%8", "%9") XMH_R2_EVAL_CLOSE
                             : "=&v"(lab), "=&v"(addr)
                             : "v"(At[1]), "v"(bq[h][2]), "v"(lab0), "v"(At[NMI - 1]), "v"(bq[h][NMQ - 1]), "v"(At[0]), "v"(bq[h][0]), "v"(cq[h]));
            } else if (CACHE) {
                asm volatile(XMH_R2_OPEN XMH_MFMA("%0", "%3", "%4", "%5") XMH_MFMA("%1", "%6", "%7", "%8")
                             XMH_MFMA("%2", "%6", "%9", "%10") XMH_R2_EVAL_CLOSE
                             : "=&v"(lab), "=&v"(addr), "=&v"(d2)
                             : "v"(At[1]), "v"(bq[h][2]), "v"(lab0), "v"(At[0]), "v"(bq[h][0]), "v"(cq[h]), "v"(bq[h][1]), "v"(kqv[REGS ? h : 0]));
            } else {
                asm volatile(XMH_R2_OPEN XMH_MFMA("%0", "%2", "%3", "%4") XMH_MFMA("%1", "%5", "%6", "%7") XMH_R2_EVAL_CLOSE
                             : "=&v"(lab), "=&v"(addr)
                             : "v"(At[1]), "v"(bq[h][2]), "v"(lab0), "v"(At[0]), "v"(bq[h][0]), "v"(cq[h]));
            }
        };
        auto consume = [&](uint32_t& w, const v4i (&live)[NMI]) {        // the last pair of a batch (live: see the hazard list above)
            uint32_t i0, i1, i2, i3;
            if (CACHE) {
                asm volatile(
                    XMH_VMIN "%1, 0x10001, %5\n\t" XMH_VMIN "%2, 0x10001, %6\n\t" XMH_VMIN "%3, 0x10001, %7\n\t" XMH_VMIN "%4, 0x10001, %8\n\t"
                    "v_or_b32_sdwa %0, %9, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0\n\t"
                    XMH_ADD("%13", "%1")
                    "v_or_b32_sdwa %0, %10, %2 " XMH_SDWA(1) XMH_ADD("%14", "%2") "v_or_b32_sdwa %0, %11, %3 " XMH_SDWA(2)
                    XMH_ADD("%15", "%3") "v_or_b32_sdwa %0, %12, %4 " XMH_SDWA(3) XMH_ADD("%16", "%4")
                    : "=&v"(w), "=&v"(i0), "=&v"(i1), "=&v"(i2), "=&v"(i3)
                    : "v"(lab_p[0]), "v"(lab_p[1]), "v"(lab_p[2]), "v"(lab_p[3]), "v"(d2_p[0]), "v"(d2_p[1]), "v"(d2_p[2]), "v"(d2_p[3]), "v"(addr_p[0]),
                      "v"(addr_p[1]), "v"(addr_p[2]), "v"(addr_p[3]), "v"(live[0]), "v"(live[1]), "v"(live[NMI - 1])
                    : "memory");
            } else {
                asm volatile(
                    XMH_VMIN "%0, 0x10001, %4\n\t" XMH_VMIN "%1, 0x10001, %5\n\t" XMH_VMIN "%2, 0x10001, %6\n\t" XMH_VMIN "%3, 0x10001, %7\n\t"
                    XMH_ADD("%8", "%0") XMH_ADD("%9", "%1") XMH_ADD("%10", "%2") XMH_ADD("%11", "%3")
                    : "=&v"(i0), "=&v"(i1), "=&v"(i2), "=&v"(i3)
                    : "v"(lab_p[0]), "v"(lab_p[1]), "v"(lab_p[2]), "v"(lab_p[3]), "v"(addr_p[0]), "v"(addr_p[1]), "v"(addr_p[2]), "v"(addr_p[3]), "v"(live[0]),
                      "v"(live[1]), "v"(live[NMI - 1])
                    : "memory");
            }
        };
#undef XMH_SDWA
#undef XMH_ADD
#undef XMH_MFMA
        auto group = [&](auto gc) {
            constexpr int G = decltype(gc)::value;
            constexpr int SET = REGS ? G : (G & 1);
#pragma unroll
            for (int h = 0; h < NQ; ++h) {
                v4i addr, d2, lab;
                if (G == 0 && h == 0) evaluate(A[SET], h, addr, d2, lab);
                else fused(A[SET], h, addr, d2, lab, cw[(h + NQ - 1) % NQ][h == 0 ? G - 1 : G]);
                addr_p = addr; d2_p = d2; lab_p = lab;
                if (REGS && G > 0 && h == 0) {                         // the previous group's tiles may be reused from here on, not earlier
                    if constexpr (NMI == 2) asm volatile("" ::"v"(A[REGS ? G - 1 : 0][0]), "v"(A[REGS ? G - 1 : 0][1]));
                    else asm volatile("" ::"v"(A[REGS ? G - 1 : 0][0]), "v"(A[REGS ? G - 1 : 0][1]), "v"(A[REGS ? G - 1 : 0][NMI - 1]));
                }
            }
        };
#ifdef XMH_ABL_NOBUILD
        if (i == 0) {
#endif
        build(A[0], wcur[0]);
        build(A[1], wcur[1]);
        build(A[2], wcur[2]);
        build(A[3], wcur[3]);
#ifdef XMH_ABL_NOBUILD
        }
#endif
        group(std::integral_constant<int, 0>{});
        group(std::integral_constant<int, 1>{});
        group(std::integral_constant<int, 2>{});
        group(std::integral_constant<int, 3>{});
        asm volatile(XMH_R2_TAIL ::: "memory");              // the last MFMAs' results: 8 wait states before a VALU / DS read
        consume(cw[NQ - 1][3], A[3]);
#ifdef XMH_ABL_NOSTORE
        if (CACHE && a.Q < 0) {
#else
        if (CACHE) {                                                 // streamed once each way: non-temporal (see k_scan_hist_s)
#endif
#pragma unroll
            for (int h = 0; h < NQ; ++h) {
                uint4* dst = crow[h] + (int64_t)i * 64;
                __builtin_nontemporal_store(cw[h][0], &dst->x);
                __builtin_nontemporal_store(cw[h][1], &dst->y);
                __builtin_nontemporal_store(cw[h][2], &dst->z);
                __builtin_nontemporal_store(cw[h][3], &dst->w);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int m = 0; m < NMI; ++m) wcur[g][m] = wnxt[g][m];
    }
    // the padding items of a ragged last batch are all-zero-bit codes without labels: distance popcount(query), never relevant
    const int npad = nbat * 64 - (int)(hi - lo);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (npad > 0 && slot == 0) {
#pragma unroll
        for (int h = 0; h < NQ; ++h) {
            if (!valid[h]) continue;
            const int64_t q = (int64_t)(t16 + h) * 16 + ql;
            int dpad = 0;
            for (int w = 0; w < a.W; ++w) dpad += __popc(a.qbits[q * a.W + w]);
            cnt[h * ncell + dpad * 16 + ql] -= (uint32_t)npad << 16;
        }
    }
#pragma unroll
    for (int h = 0; h < NQ; ++h) {
        uint32_t* __restrict__ out = chunk_hist + ((int64_t)chunk_id * a.nb) * a.qpad + (t16 + h) * 16;
        for (int e = lane; e < ncell; e += 64) out[(int64_t)(e >> 4) * a.qpad + (e & 15)] = cnt[h * ncell + e];
    }
}

// What is left of k_scan_expand2 for k_scan_hist_r2: the control words of the call are cleared, and the packed words of every chunk are
// read once by blocks that land on the XCD whose k_scan_hist_r2 blocks will read them (block b runs on XCD b & 7, chunk c is scanned on XCD
// c & 7: mfma_map_block).  k_scan_hist_r2 fetches its words one batch ahead, which covers an L2 hit but not a miss to HBM behind the
// 600 MB the previous evaluation's pass 2 streamed through: without this launch pass 1 measured 0.204 ms instead of 0.184 (the image
// kernel it replaces had been warming the caches by accident).
constexpr int kTouchPerChunk = 8;
__global__ __launch_bounds__(256) void k_scan_touch(const uint32_t* __restrict__ rbits, const uint32_t* __restrict__ rlab, int64_t R, int W, int LW, int64_t chunk,
                                                    int nchunk, uint32_t* __restrict__ ctl, int ctl_words) {
    if (blockIdx.x == 0)
        for (int e = threadIdx.x; e < ctl_words; e += 256) ctl[e] = 0u;
    const int b = blockIdx.x;
    const int c = (b & 7) + 8 * ((b >> 3) / kTouchPerChunk), sub = (b >> 3) % kTouchPerChunk;
    if (c >= nchunk) return;
    const int64_t lo = (int64_t)c * chunk, hi = lo + chunk < R ? lo + chunk : R;
    uint32_t acc = 0u;
    auto sweep = [&](const uint32_t* base, int64_t w0, int64_t w1) {      // one load per 128-byte line is enough
        for (int64_t w = w0 + ((int64_t)sub * 256 + threadIdx.x) * 32; w < w1; w += (int64_t)kTouchPerChunk * 256 * 32) acc ^= base[w];
    };
    sweep(rbits, lo * W, hi * W);
    sweep(rlab, lo * LW, hi * LW);
    asm volatile("" ::"v"(acc));
}

// k_scan_hist_r2 (round 4): k_scan_hist_m2's statements fed from registers.  The skeleton around the MFMA statements of k_scan_hist_m2 --
// LDS-DMA issue, waiting for pieces, the barrier, the A-tile reads: a third of a wave's cycles (tools/stamp_m2.hip) -- and the 36 KB ring
// that holds a block to two per CU exist only to bring 16 bytes per lane and tile that are a function of ONE packed word: here each lane
// loads that word (4 bytes per tile; a wave's 16 items x 24 bytes per group, L2-resident) one batch ahead and spreads it with 8 VALU
// operations per tile.  Waves are independent (no barrier, no shared staging); LDS holds the counters only.
#ifndef XMH_R2_ATTR
#define XMH_R2_ATTR                                                  // (tools/proto_scan_ablate.hip sets occupancy attributes here)
#endif
template <int NML, int NW, int NQ, bool CACHE>
__global__ __launch_bounds__(64 * NW) XMH_R2_ATTR void k_scan_hist_r2(MfmaArgs a, uint32_t* __restrict__ chunk_hist, uint4* __restrict__ pair_cache) {
    scan_hist_r2_body<NML, NW, NQ, CACHE>(a, chunk_hist, pair_cache);
}

// ---------------------------------------------------------------------------------------------------
// k_scan_hist_r2w (round 4): k_scan_hist_r2 for codes of 65..128 bits -- TWO code tiles per chain (six MFMAs per (16 items x 16 queries):
// label, label, address, address, 2 * distance, 2 * distance), 129 bucket rows, 2 query groups per wave (66 KB of counters per block of four
// waves, two blocks per CU), one-byte pair-cache entries in the layout of the shorter codes.  Same operand construction, same pipeline of
// statements one (item group, query group) behind their MFMAs, same hazards (see k_scan_hist_m2).  Two differences in form: the operands
// are named (a statement has 28 of the 30 an asm may take), and the cache word is assembled by a second, small statement -- byte j of the
// word = byte0(2 * distance) | byte0(increment), the even bytes into one register and the odd ones into another so that no two SDWA
// inserts into one register follow each other (the dst_sel hazard), OR-ed at the end.  A distance of 128 makes 2 * distance = 256, whose
// byte wraps to 0: the same statement keeps the largest 2 * distance seen, and the kernel raises *ovf as k_scan_hist_m<.., BYTE> does.
// ---------------------------------------------------------------------------------------------------
#define XMH_W_MFMA(D, A, B, C) "v_mfma_i32_16x16x64_i8 %[" D "], %[" A "], %[" B "], %[" C "]\n\t"
#define XMH_W_MIN(I, L) "v_min_u32 %[" I "], 0x10001, %[" L "]\n\t"
#define XMH_W_ADD(A, D) "ds_add_u32 %[" A "], %[" D "]\n\t"
template <int NML, int NW, int NQ, bool CACHE>
__global__ __launch_bounds__(64 * NW) void k_scan_hist_r2w(MfmaArgs a, uint32_t* __restrict__ chunk_hist, uint4* __restrict__ pair_cache,
                                                           uint32_t* __restrict__ ovf) {
    constexpr int NMC = 2, NMI = NMC + NML;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];   // NW x NQ x [nb][16] u32 counters (all << 16 | relevant)
    int chunk_id, qtile;
    if (!mfma_map_block(a, chunk_id, qtile)) return;                 // a.nqt counts tiles of NW * NQ * 16 queries here
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ql = lane & 15, slot = lane >> 4;
    const int t16 = (qtile * NW + wave) * NQ;
    const int ncell = a.nb * 16;
    uint32_t* cnt = lds + (wave * NQ) * ncell;
    for (int e = lane; e < NQ * ncell; e += 64) cnt[e] = 0u;
    const int rsh = 4 * (slot & 1), rwi = slot >> 1;                 // this lane's nibble of the 16 bits it owns; its word inside a 64-bit tile
    v4i bA[NQ][NMC], bD[NQ][NMC], bL[NQ][NML], cq[NQ], kq[NQ];
    bool valid[NQ];
#pragma unroll
    for (int h = 0; h < NQ; ++h) {
        const int64_t q = (int64_t)(t16 + h) * 16 + ql;
        valid[h] = q < a.Q;
        int pcq = 0;
        if (valid[h])
            for (int w = 0; w < a.W; ++w) pcq += __popc(a.qbits[q * a.W + w]);
#pragma unroll
        for (int c = 0; c < NMC; ++c) {
            const int wi = 2 * c + rwi;
            const bool on = valid[h] && wi < a.W;                    // no such code word: zero operand, whatever the item lane holds
            const uint32_t qw = (on ? a.qbits[q * a.W + wi] : 0u) >> rsh;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t t = (qw >> j) & 0x01010101u;
                if (j & 1) {                                         // registers 1 and 3 of an item tile hold their bits worth 2: halved query bytes
                    bA[h][c][j] = on ? (int)(0x20202020u ^ (t * 0xc0u)) : 0;
                    bD[h][c][j] = on ? (int)(0x01010101u ^ (t * 0xfeu)) : 0;
                } else {
                    bA[h][c][j] = on ? (int)(0x40404040u ^ (t << 7)) : 0;
                    bD[h][c][j] = on ? (int)(0x02020202u ^ (t * 0xfcu)) : 0;
                }
            }
        }
#pragma unroll
        for (int m = 0; m < NML; ++m) {
            uint32_t lw = 0u;
            if (valid[h] && 2 * m + rwi < a.LW) lw = a.qlab[q * a.LW + 2 * m + rwi];
            lw >>= rsh;
#pragma unroll
            for (int j = 0; j < 4; ++j) bL[h][m][j] = (int)((lw >> j) & 0x01010101u);
        }
        const int c0 = (int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)(cnt + h * ncell) + ql * 4 + (valid[h] ? 64 * pcq : 0);
        const int k2 = valid[h] ? 2 * pcq : 0;
        cq[h] = v4i{c0, c0, c0, c0};
        kq[h] = v4i{k2, k2, k2, k2};
    }
    v4i lab0 = {0x10000, 0x10000, 0x10000, 0x10000};
    asm volatile("" : "+v"(lab0));                                  // opaque: kept in VGPRs, not re-materialised in front of an MFMA
#pragma unroll
    for (int h = 0; h < NQ; ++h) asm volatile("" : "+v"(cq[h]), "+v"(kq[h]));
    const int64_t lo = (int64_t)chunk_id * a.chunk;
    const int64_t hi = (lo + a.chunk < a.R) ? lo + a.chunk : a.R;
    const int nbat = (int)((hi - lo + 63) >> 6);
    const int64_t bat0 = lo >> 6;                                    // chunks start on 64-item boundaries
    uint4* crow[NQ];
#pragma unroll
    for (int h = 0; h < NQ; ++h)
        crow[h] = CACHE ? pair_cache + ((int64_t)chunk_id * (a.qpad >> 4) + (t16 + h)) * ((a.chunk + 63) >> 6) * 64 + lane : nullptr;
    uint32_t wcur[4][NMI], wnxt[4][NMI];
    const int ritem = 4 * (lane & 3) + ((lane & 15) >> 2);           // row r of group g is item 16 g + 4 (r & 3) + (r >> 2)
    int wi_c[NMC], wi_l[NML];
#pragma unroll
    for (int c = 0; c < NMC; ++c) wi_c[c] = 2 * c + rwi < a.W ? 2 * c + rwi : a.W - 1;
#pragma unroll
    for (int m = 0; m < NML; ++m) wi_l[m] = 2 * m + rwi < a.LW ? 2 * m + rwi : (a.LW > 0 ? a.LW - 1 : 0);
    auto load_words = [&](int64_t batch, uint32_t (&w)[4][NMI]) {
        const int64_t first = batch * 64;
        if (first + 64 <= (int64_t)a.R) {                             // whole batch inside the gallery (wave-uniform)
            const uint32_t* __restrict__ pc = a.rbits + (first + ritem) * a.W;
            const uint32_t* __restrict__ pl = a.rlab + (first + ritem) * a.LW;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int c = 0; c < NMC; ++c) w[g][c] = pc[g * 16 * a.W + wi_c[c]];
#pragma unroll
                for (int m = 0; m < NML; ++m) w[g][NMC + m] = pl[g * 16 * a.LW + wi_l[m]];
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t item = first + g * 16 + ritem;
                const int64_t it = item < a.R ? item : (int64_t)a.R - 1;
                const uint32_t ok = item < a.R ? 0xffffffffu : 0u;   // items past the end: all-zero codes, no labels (taken out again below)
#pragma unroll
                for (int c = 0; c < NMC; ++c) w[g][c] = a.rbits[it * a.W + wi_c[c]] & ok;
#pragma unroll
                for (int m = 0; m < NML; ++m) w[g][NMC + m] = a.rlab[it * a.LW + wi_l[m]] & ok;
            }
        }
    };
    auto build = [&](v4i (&At)[NMI], const uint32_t (&w)[NMI]) {     // see k_scan_hist_r2: 6 operations per code tile, 5 per label tile
#pragma unroll
        for (int c = 0; c < NMC; ++c) {
            const uint32_t xa = w[c] >> rsh, xb = w[c] >> (rsh + 2);
            At[c][0] = (int)(xa & 0x01010101u);
            At[c][1] = (int)(xa & 0x02020202u);
            At[c][2] = (int)(xb & 0x01010101u);
            At[c][3] = (int)(xb & 0x02020202u);
        }
#pragma unroll
        for (int m = 0; m < NML; ++m) {
            const uint32_t y = w[NMC + m] >> rsh;
#pragma unroll
            for (int j = 0; j < 4; ++j) At[NMC + m][j] = (int)(y & (0x01010101u << j));
        }
    };
    uint32_t dmax = 0u;                                              // largest 2 * distance this lane packed
    load_words(bat0, wcur);
    for (int i = 0; i < nbat; ++i) {
        load_words(bat0 + (i + 1 < nbat ? i + 1 : i), wnxt);
        v4i A[4][NMI];
        uint32_t cw[NQ][4];
        v4i addr_p = {0, 0, 0, 0}, d2_p = {0, 0, 0, 0}, lab_p = {0, 0, 0, 0};
        // the MFMAs of (group g, query group h) with the increments and adds of the PREVIOUS pair between them
        auto fused = [&](const v4i (&At)[NMI], int h, v4i& addr, v4i& d2, v4i& lab, uint32_t& i0, uint32_t& i1, uint32_t& i2, uint32_t& i3) {
            if (NML == 2 && CACHE) {
                asm volatile("s_nop 3\n\t" XMH_W_MFMA("lab", "al0", "bl0", "lab0") XMH_W_MIN("i0", "lp0") XMH_W_MIN("i1", "lp1") XMH_W_MIN("i2", "lp2")
                             XMH_W_MFMA("lab", "al1", "bl1", "lab") XMH_W_MIN("i3", "lp3") XMH_W_ADD("ap0", "i0") XMH_W_MFMA("addr", "a0", "ba0", "cq")
                             XMH_W_ADD("ap1", "i1") XMH_W_MFMA("addr", "a1", "ba1", "addr") XMH_W_ADD("ap2", "i2") XMH_W_MFMA("d2", "a0", "bd0", "kq")
                             XMH_W_ADD("ap3", "i3") XMH_W_MFMA("d2", "a1", "bd1", "d2")
                             : [lab] "=&v"(lab), [addr] "=&v"(addr), [d2] "=&v"(d2), [i0] "=&v"(i0), [i1] "=&v"(i1), [i2] "=&v"(i2), [i3] "=&v"(i3)
                             : [a0] "v"(At[0]), [a1] "v"(At[1]), [al0] "v"(At[2]), [al1] "v"(At[NMI - 1]), [ba0] "v"(bA[h][0]), [ba1] "v"(bA[h][1]),
                               [bd0] "v"(bD[h][0]), [bd1] "v"(bD[h][1]), [bl0] "v"(bL[h][0]), [bl1] "v"(bL[h][NML - 1]), [lab0] "v"(lab0), [cq] "v"(cq[h]),
                               [kq] "v"(kq[h]), [lp0] "v"(lab_p[0]), [lp1] "v"(lab_p[1]), [lp2] "v"(lab_p[2]), [lp3] "v"(lab_p[3]), [ap0] "v"(addr_p[0]),
                               [ap1] "v"(addr_p[1]), [ap2] "v"(addr_p[2]), [ap3] "v"(addr_p[3])
                             : "memory");
            } else if (NML == 2) {
                asm volatile("s_nop 3\n\t" XMH_W_MFMA("lab", "al0", "bl0", "lab0") XMH_W_MIN("i0", "lp0") XMH_W_MIN("i1", "lp1") XMH_W_MIN("i2", "lp2")
                             XMH_W_MFMA("lab", "al1", "bl1", "lab") XMH_W_MIN("i3", "lp3") XMH_W_ADD("ap0", "i0") XMH_W_ADD("ap1", "i1")
                             XMH_W_MFMA("addr", "a0", "ba0", "cq") XMH_W_ADD("ap2", "i2") XMH_W_ADD("ap3", "i3") XMH_W_MFMA("addr", "a1", "ba1", "addr")
                             : [lab] "=&v"(lab), [addr] "=&v"(addr), [i0] "=&v"(i0), [i1] "=&v"(i1), [i2] "=&v"(i2), [i3] "=&v"(i3)
                             : [a0] "v"(At[0]), [a1] "v"(At[1]), [al0] "v"(At[2]), [al1] "v"(At[NMI - 1]), [ba0] "v"(bA[h][0]), [ba1] "v"(bA[h][1]),
                               [bl0] "v"(bL[h][0]), [bl1] "v"(bL[h][NML - 1]), [lab0] "v"(lab0), [cq] "v"(cq[h]), [lp0] "v"(lab_p[0]), [lp1] "v"(lab_p[1]),
                               [lp2] "v"(lab_p[2]), [lp3] "v"(lab_p[3]), [ap0] "v"(addr_p[0]), [ap1] "v"(addr_p[1]), [ap2] "v"(addr_p[2]), [ap3] "v"(addr_p[3])
                             : "memory");
            } else if (CACHE) {
                asm volatile("s_nop 3\n\t" XMH_W_MFMA("lab", "al0", "bl0", "lab0") XMH_W_MIN("i0", "lp0") XMH_W_MIN("i1", "lp1") XMH_W_MIN("i2", "lp2")
                             XMH_W_MIN("i3", "lp3") XMH_W_ADD("ap0", "i0") XMH_W_MFMA("addr", "a0", "ba0", "cq") XMH_W_ADD("ap1", "i1")
                             XMH_W_MFMA("addr", "a1", "ba1", "addr") XMH_W_ADD("ap2", "i2") XMH_W_MFMA("d2", "a0", "bd0", "kq") XMH_W_ADD("ap3", "i3")
                             XMH_W_MFMA("d2", "a1", "bd1", "d2")
                             : [lab] "=&v"(lab), [addr] "=&v"(addr), [d2] "=&v"(d2), [i0] "=&v"(i0), [i1] "=&v"(i1), [i2] "=&v"(i2), [i3] "=&v"(i3)
                             : [a0] "v"(At[0]), [a1] "v"(At[1]), [al0] "v"(At[2]), [ba0] "v"(bA[h][0]), [ba1] "v"(bA[h][1]), [bd0] "v"(bD[h][0]),
                               [bd1] "v"(bD[h][1]), [bl0] "v"(bL[h][0]), [lab0] "v"(lab0), [cq] "v"(cq[h]), [kq] "v"(kq[h]), [lp0] "v"(lab_p[0]),
                               [lp1] "v"(lab_p[1]), [lp2] "v"(lab_p[2]), [lp3] "v"(lab_p[3]), [ap0] "v"(addr_p[0]), [ap1] "v"(addr_p[1]), [ap2] "v"(addr_p[2]),
                               [ap3] "v"(addr_p[3])
                             : "memory");
            } else {
                asm volatile("s_nop 3\n\t" XMH_W_MFMA("lab", "al0", "bl0", "lab0") XMH_W_MIN("i0", "lp0") XMH_W_MIN("i1", "lp1") XMH_W_MIN("i2", "lp2")
                             XMH_W_MIN("i3", "lp3") XMH_W_ADD("ap0", "i0") XMH_W_ADD("ap1", "i1") XMH_W_MFMA("addr", "a0", "ba0", "cq") XMH_W_ADD("ap2", "i2")
                             XMH_W_ADD("ap3", "i3") XMH_W_MFMA("addr", "a1", "ba1", "addr")
                             : [lab] "=&v"(lab), [addr] "=&v"(addr), [i0] "=&v"(i0), [i1] "=&v"(i1), [i2] "=&v"(i2), [i3] "=&v"(i3)
                             : [a0] "v"(At[0]), [a1] "v"(At[1]), [al0] "v"(At[2]), [ba0] "v"(bA[h][0]), [ba1] "v"(bA[h][1]), [bl0] "v"(bL[h][0]), [lab0] "v"(lab0),
                               [cq] "v"(cq[h]), [lp0] "v"(lab_p[0]), [lp1] "v"(lab_p[1]), [lp2] "v"(lab_p[2]), [lp3] "v"(lab_p[3]), [ap0] "v"(addr_p[0]),
                               [ap1] "v"(addr_p[1]), [ap2] "v"(addr_p[2]), [ap3] "v"(addr_p[3])
                             : "memory");
            }
        };
        // the cache word of the pair whose increments the statement above just made (d: that pair's 2 * distance results, a statement old)
        auto pack = [&](uint32_t& w, const v4i& d, uint32_t i0, uint32_t i1, uint32_t i2, uint32_t i3) {
            uint32_t wb;
            asm volatile("v_or_b32_sdwa %[wa], %[d0], %[i0] dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0\n\t"
                         "v_or_b32_sdwa %[wb], %[d1], %[i1] dst_sel:BYTE_1 dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0\n\t"
                         "v_max3_u32 %[mx], %[d0], %[d1], %[mx]\n\t"
                         "v_or_b32_sdwa %[wa], %[d2], %[i2] dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0\n\t"
                         "v_or_b32_sdwa %[wb], %[d3], %[i3] dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0\n\t"
                         "v_max3_u32 %[mx], %[d2], %[d3], %[mx]\n\t"
                         "v_or_b32 %[wa], %[wa], %[wb]"
                         : [wa] "=&v"(w), [wb] "=&v"(wb), [mx] "+v"(dmax)
                         : [d0] "v"(d[0]), [d1] "v"(d[1]), [d2] "v"(d[2]), [d3] "v"(d[3]), [i0] "v"(i0), [i1] "v"(i1), [i2] "v"(i2), [i3] "v"(i3));
        };
        // the first statement of a batch: nothing to consume yet; closed by 8 wait states (the next statement's consumers read these results)
        auto evaluate = [&](const v4i (&At)[NMI], int h, v4i& addr, v4i& d2, v4i& lab) {
            if (NML == 2) {
                asm volatile("s_nop 3\n\t" XMH_W_MFMA("lab", "al0", "bl0", "lab0") XMH_W_MFMA("lab", "al1", "bl1", "lab")
                             : [lab] "=&v"(lab)
                             : [al0] "v"(At[2]), [al1] "v"(At[NMI - 1]), [bl0] "v"(bL[h][0]), [bl1] "v"(bL[h][NML - 1]), [lab0] "v"(lab0));
            } else {
                asm volatile("s_nop 3\n\t" XMH_W_MFMA("lab", "al0", "bl0", "lab0") : [lab] "=&v"(lab) : [al0] "v"(At[2]), [bl0] "v"(bL[h][0]), [lab0] "v"(lab0));
            }
            if (CACHE) {
                asm volatile("s_nop 3\n\t" XMH_W_MFMA("addr", "a0", "ba0", "cq") XMH_W_MFMA("addr", "a1", "ba1", "addr") XMH_W_MFMA("d2", "a0", "bd0", "kq")
                             XMH_W_MFMA("d2", "a1", "bd1", "d2") "s_nop 7"
                             : [addr] "=&v"(addr), [d2] "=&v"(d2)
                             : [a0] "v"(At[0]), [a1] "v"(At[1]), [ba0] "v"(bA[h][0]), [ba1] "v"(bA[h][1]), [bd0] "v"(bD[h][0]), [bd1] "v"(bD[h][1]), [cq] "v"(cq[h]),
                               [kq] "v"(kq[h]));
            } else {
                asm volatile("s_nop 3\n\t" XMH_W_MFMA("addr", "a0", "ba0", "cq") XMH_W_MFMA("addr", "a1", "ba1", "addr") "s_nop 7"
                             : [addr] "=&v"(addr)
                             : [a0] "v"(At[0]), [a1] "v"(At[1]), [ba0] "v"(bA[h][0]), [ba1] "v"(bA[h][1]), [cq] "v"(cq[h]));
            }
        };
        // the last pair of a batch: its increments and adds (live: the tiles of the last statements stay untouched until here)
        auto consume = [&](uint32_t& i0, uint32_t& i1, uint32_t& i2, uint32_t& i3, const v4i (&live)[NMI]) {
            if constexpr (NMI == 4) {
                asm volatile(XMH_W_MIN("i0", "lp0") XMH_W_MIN("i1", "lp1") XMH_W_MIN("i2", "lp2") XMH_W_MIN("i3", "lp3") XMH_W_ADD("ap0", "i0")
                             XMH_W_ADD("ap1", "i1") XMH_W_ADD("ap2", "i2") XMH_W_ADD("ap3", "i3")
                             : [i0] "=&v"(i0), [i1] "=&v"(i1), [i2] "=&v"(i2), [i3] "=&v"(i3)
                             : [lp0] "v"(lab_p[0]), [lp1] "v"(lab_p[1]), [lp2] "v"(lab_p[2]), [lp3] "v"(lab_p[3]), [ap0] "v"(addr_p[0]), [ap1] "v"(addr_p[1]),
                               [ap2] "v"(addr_p[2]), [ap3] "v"(addr_p[3]), "v"(live[0]), "v"(live[1]), "v"(live[2]), "v"(live[3])
                             : "memory");
            } else {
                asm volatile(XMH_W_MIN("i0", "lp0") XMH_W_MIN("i1", "lp1") XMH_W_MIN("i2", "lp2") XMH_W_MIN("i3", "lp3") XMH_W_ADD("ap0", "i0")
                             XMH_W_ADD("ap1", "i1") XMH_W_ADD("ap2", "i2") XMH_W_ADD("ap3", "i3")
                             : [i0] "=&v"(i0), [i1] "=&v"(i1), [i2] "=&v"(i2), [i3] "=&v"(i3)
                             : [lp0] "v"(lab_p[0]), [lp1] "v"(lab_p[1]), [lp2] "v"(lab_p[2]), [lp3] "v"(lab_p[3]), [ap0] "v"(addr_p[0]), [ap1] "v"(addr_p[1]),
                               [ap2] "v"(addr_p[2]), [ap3] "v"(addr_p[3]), "v"(live[0]), "v"(live[1]), "v"(live[2])
                             : "memory");
            }
        };
        auto group = [&](auto gc) {
            constexpr int G = decltype(gc)::value;
#pragma unroll
            for (int h = 0; h < NQ; ++h) {
                v4i addr, d2 = d2_p, lab;
                if (G == 0 && h == 0) {
                    evaluate(A[G], h, addr, d2, lab);
                } else {
                    uint32_t i0, i1, i2, i3;
                    fused(A[G], h, addr, d2, lab, i0, i1, i2, i3);
                    if (CACHE) pack(cw[(h + NQ - 1) % NQ][h == 0 ? G - 1 : G], d2_p, i0, i1, i2, i3);
                }
                addr_p = addr; d2_p = d2; lab_p = lab;
                if (G > 0 && h == 0) {                                 // the previous group's tiles may be reused from here on, not earlier
                    if constexpr (NMI == 4) asm volatile("" ::"v"(A[G > 0 ? G - 1 : 0][0]), "v"(A[G > 0 ? G - 1 : 0][1]), "v"(A[G > 0 ? G - 1 : 0][2]), "v"(A[G > 0 ? G - 1 : 0][3]));
                    else asm volatile("" ::"v"(A[G > 0 ? G - 1 : 0][0]), "v"(A[G > 0 ? G - 1 : 0][1]), "v"(A[G > 0 ? G - 1 : 0][2]));
                }
            }
        };
        build(A[0], wcur[0]);
        build(A[1], wcur[1]);
        build(A[2], wcur[2]);
        build(A[3], wcur[3]);
        group(std::integral_constant<int, 0>{});
        group(std::integral_constant<int, 1>{});
        group(std::integral_constant<int, 2>{});
        group(std::integral_constant<int, 3>{});
        asm volatile("s_nop 7\n\ts_nop 3" ::: "memory");              // the last MFMAs' results: 8 wait states before a VALU / DS read
        {
            uint32_t i0, i1, i2, i3;
            consume(i0, i1, i2, i3, A[3]);
            if (CACHE) pack(cw[NQ - 1][3], d2_p, i0, i1, i2, i3);
        }
#ifdef XMH_ABL_NOSTORE
        if (CACHE && a.Q < 0) {
#else
        if (CACHE) {                                                 // streamed once each way: non-temporal (see k_scan_hist_s)
#endif
#pragma unroll
            for (int h = 0; h < NQ; ++h) {
                uint4* dst = crow[h] + (int64_t)i * 64;
                __builtin_nontemporal_store(cw[h][0], &dst->x);
                __builtin_nontemporal_store(cw[h][1], &dst->y);
                __builtin_nontemporal_store(cw[h][2], &dst->z);
                __builtin_nontemporal_store(cw[h][3], &dst->w);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int m = 0; m < NMI; ++m) wcur[g][m] = wnxt[g][m];
    }
    // the padding items of a ragged last batch are all-zero-bit codes without labels: distance popcount(query), never relevant
    const int npad = nbat * 64 - (int)(hi - lo);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (npad > 0 && slot == 0) {
#pragma unroll
        for (int h = 0; h < NQ; ++h) {
            if (!valid[h]) continue;
            const int64_t q = (int64_t)(t16 + h) * 16 + ql;
            int dpad = 0;
            for (int w = 0; w < a.W; ++w) dpad += __popc(a.qbits[q * a.W + w]);
            cnt[h * ncell + dpad * 16 + ql] -= (uint32_t)npad << 16;
        }
    }
#pragma unroll
    for (int h = 0; h < NQ; ++h) {
        uint32_t* __restrict__ out = chunk_hist + ((int64_t)chunk_id * a.nb) * a.qpad + (t16 + h) * 16;
        for (int e = lane; e < ncell; e += 64) out[(int64_t)(e >> 4) * a.qpad + (e & 15)] = cnt[h * ncell + e];
    }
    if (CACHE && dmax >= 256u) *ovf = 1u;                            // every writer stores the same 1; read by the next launch
}
#undef XMH_W_MFMA
#undef XMH_W_MIN
#undef XMH_W_ADD

// ---------------------------------------------------------------------------------------------------
// k_scan_ap_c: pass 2 from the one-byte pair cache (binary codes of at most 64 bits), round 3.  Same counters-in-LDS scheme and
// the same arithmetic as the cached k_scan_ap_s, so the results are bit-identical; what changed is the VALU work per pair, which
// bounds this pass (PMC, round 2: VALU busy 86 % of the launch at 10.5 instructions per pair, most of them half-rate):
//   * the counters hold FLOAT BIT PATTERNS: rank counter = bits(2^23 + rank), ordinal counter = bits(2^24 - 1 - ordinal).  An integer
//     add of 1 to the bits of a float in [2^23, 2^24) is an add of 1.0 to its value, so the 64-bit add {1, -relevant} still steps
//     both, and what comes back needs no v_cvt_f32_u32 (half rate) nor the ordinal * relevant product: rank = lo - 2^23 and
//     ordinal = (2^24 - 1) - hi are full-rate float subtractions, and the relevance mask (0 / ~0, sign-extended straight out of
//     the cache byte) zeroes the reciprocal with a full-rate AND.  Needs every rank and ordinal below 2^23: galleries of up to
//     8 388 605 items over all shards (larger ones take k_scan_ap_s);
//   * the counter address is one v_lshl_add_u32 on a precomputed LDS address (hipcc emitted shift, and, three-operand add);
//   * the atomics are inline asm on that address, their returns waited for with counted lgkmcnt one group of 8 later.
// Per pair: 2 bit-field extracts, the address, the pair {1, mask}, ds_add_rtn_u64; sub, rcp, and, sub, fmac.
// ---------------------------------------------------------------------------------------------------
// EB = entry bits of the pair cache: 8 (codes of at most 64 bits: 4 slots x 16 queries, 16 steps per lane and batch) or 16 (65..256 bits:
// 8 slots x 8 queries, 8 steps)
// HALF (round 4, one-byte entries only): 8 slots x 8 queries on the cache pass 1 wrote for 4 slots x 16 queries -- half the counter rows per
// wave (129 bucket rows of 65..128-bit codes: 16.5 -> 8.3 KB, twice the waves per CU).  Lane (slot8, query8) of half h reads the 16-byte
// record of the writer's lane (slot8 & 3, 8 h + query8) and takes every other byte of it: the writer's step t holds items 4 t + slot4,
// so step t' here = the writer's step 2 t' + (slot8 >> 2), items 8 t' + slot8 -- ascending with the lane, as the order of the returning
// adds requires.  The byte offsets (8 (slot8 >> 2) and 16 more) go into v_bfe as register operands: no instruction more per pair.
// (Round 6: the same reading of TWO-byte entries, 16 slots x 4 queries, was measured and dropped: twice the waves per CU, but the two
// 4-query tiles of a record row are different blocks and each fetched the row's lines -- configs[4]'s shard 2.56 -> 5.65 ms.  That pass is
// bound by the stream of its pair cache; fetching the words four batches ahead instead of two is worth 4 % there, below.  A batch-major record order -- the records
// of one batch of all query tiles contiguous, so that the blocks running side by side stream one region -- changed nothing at any shape.)
template <bool CAPPED, int EB, bool HALF = false>
__global__ __launch_bounds__(64) void k_scan_ap_c(ScanArgs a, const uint2* __restrict__ below, const uint2* __restrict__ dpre,
                                                  const uint32_t* __restrict__ cap_ws, float* __restrict__ ap_part,
                                                  const uint32_t* __restrict__ items_total, uint32_t kcap, const uint32_t* __restrict__ skip_if = nullptr,
                                                  const uint32_t* __restrict__ nrel_max = nullptr, int rank_bits = 0) {
    static_assert(!HALF || EB == 8, "the 8 x 8 geometry on one-byte entries");
    constexpr int QW = HALF ? 8 : 128 / EB, LOG_QW = QW == 16 ? 4 : 3, S = 64 / QW, EPW = HALF ? 2 : 32 / EB;      // queries per tile, slots, entries a lane takes from a cache word
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];   // [nb][QW] 64-bit counters
    int chunk_id, qtile;
    if (!map_block(a, chunk_id, qtile)) return;                      // a.nqt counts QW-query tiles here
    if (items_total && (int64_t)*items_total > kFloatBitsMaxItems) return;      // sharded call: the integer-counter kernel takes it
    if (skip_if && *skip_if != 0u) return;                           // a distance wrapped in the one-byte cache of 65..128-bit codes: see k_scan_hist_m
    // two-byte entries (129..256 bits, round 6): the packed 32-bit k_scan_ap_s is launched beside this kernel and takes the call when the
    // shard's ranks and relevant counts fit its counters (the same test, on the same device word, as in k_scan_ap_s)
    if (nrel_max && rank_bits > 0 && (uint64_t)(*nrel_max) + 2 < (1ull << (32 - rank_bits))) return;
    const int lane = threadIdx.x & 63;
    const int ql = lane & (QW - 1), slot = lane >> LOG_QW;
    const int q0 = qtile * QW, q = q0 + ql;
    const int ncell = a.nb * QW;
    unsigned long long* cnt = reinterpret_cast<unsigned long long*>(lds);
    {
        const uint2* __restrict__ pb = below + ((int64_t)chunk_id * a.nb) * a.qpad + q0;
        const uint2* __restrict__ pd = dpre + q0;
        auto pack = [](uint2 x, uint2 y) {
            return (unsigned long long)(kF23 + x.x + y.x + 1u) | ((unsigned long long)(kF23 + (0x7fffffu - (x.y + y.y + 1u))) << 32);
        };
        int e = lane;
        for (; e + 7 * 64 < ncell; e += 8 * 64) {
            uint2 x[8], y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ee = e + j * 64;
                const int64_t at = (int64_t)(ee >> LOG_QW) * a.qpad + (ee & (QW - 1));
                x[j] = pb[at];
                y[j] = pd[at];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) cnt[e + j * 64] = pack(x[j], y[j]);
        }
        for (; e < ncell; e += 64) {
            const int64_t at = (int64_t)(e >> LOG_QW) * a.qpad + (e & (QW - 1));
            cnt[e] = pack(pb[at], pd[at]);
        }
    }
    const uint32_t cntbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)lds + ql * 8;
    const float capf = CAPPED ? (float)min(cap_ws[q], kcap) : 0.0f;   // exact: below 2^23
    const int64_t lo = (int64_t)chunk_id * a.chunk;
    const int64_t hi = (lo + a.chunk < a.R) ? lo + a.chunk : a.R;
    float acc = 0.0f;
    auto credit = [&](unsigned long long old, uint32_t m) {
        // (one v_pk_add_f32 with neg_hi does both subtractions on the returned register pair -- measured: 0.182 ms either way, the pass
        // is not bound by its VALU instruction count alone; left as two plain operations)
        const float rank = __uint_as_float((uint32_t)old) - 8388608.0f;
        const float ord = 16777215.0f - __uint_as_float((uint32_t)(old >> 32));
        if (CAPPED) m = ord <= capf ? m : 0u;
#ifdef XMH_ABL_AP_NORCP
        acc = fmaf(ord, __uint_as_float(__float_as_uint(rank) & m), acc);
#else
        acc = fmaf(ord, __uint_as_float(__float_as_uint(__builtin_amdgcn_rcpf(rank)) & m), acc);
#endif
    };
    const uint32_t hoff = HALF ? 8u * (uint32_t)(slot >> 2) : 0u;    // HALF: this lane's bytes of a word are hoff / 8 and hoff / 8 + 2
    const uint32_t hb0 = hoff, hb1 = hoff + 16u, hd0 = hoff + 1u, hd1 = hoff + 17u;
    // j: the entry's place in the word (one-byte entries), or -- EB == 16: entries stored 12 bits each, xmh_common.h -- its bit offset there
    auto issue1 = [&](uint32_t w, int j, unsigned long long& old, uint32_t& m) {
        uint32_t d;
        if (HALF) {
            m = (uint32_t)__builtin_amdgcn_sbfe((int)w, j ? hb1 : hb0, 1);
            d = __builtin_amdgcn_ubfe(w, j ? hd1 : hd0, EB - 1);
        } else if (EB == 16) {
            m = (uint32_t)__builtin_amdgcn_sbfe((int)w, j, 1);
            d = __builtin_amdgcn_ubfe(w, j + 1, 11);
        } else {
            m = (uint32_t)__builtin_amdgcn_sbfe((int)w, EB * j, 1);               // 0 / ~0
            d = __builtin_amdgcn_ubfe(w, EB * j + 1, EB - 1);
        }
        uint32_t addr;
        if (QW == 16) asm("v_lshl_add_u32 %0, %1, 7, %2" : "=v"(addr) : "v"(d), "v"(cntbase));      // rows of 16 queries x 8 bytes
        else asm("v_lshl_add_u32 %0, %1, 6, %2" : "=v"(addr) : "v"(d), "v"(cntbase));
        const unsigned long long inc = 1ull | ((unsigned long long)m << 32);
#ifdef XMH_ABL_AP_NOATOMIC
        old = ((unsigned long long)addr << 3) | inc;                  // (ablation: no LDS operation; the waits below then wait for nothing)
        asm volatile("" : "+v"(old));
#else
        asm volatile("ds_add_rtn_u64 %0, %1, %2" : "=v"(old) : "v"(addr), "v"(inc) : "memory");     // same-address lanes resolve in lane = item order
#endif
    };
    // Two sets of result registers, A and B, used in turn.  The returns of a group are only ever named by the asm statement that issued them
    // and by the `s_waitcnt` statement that later covers them (as in-out operands): between the two hipcc must not see a reason to touch
    // them -- it believes an asm's result is there when the statement ends.  Round 4: the earlier form handed a group's results to "the
    // previous group" by assignment (oldp = oldn); on the path of a chunk with exactly one whole batch hipcc made that eight v_mov_b64 in
    // front of the wait, i.e. copies of registers the LDS had not written yet -- wrong APs a few evaluations in a thousand, only in the
    // one-group-per-batch variants (tools/isa_hazards.py rule R5 now checks every pass-2 kernel for it at build time).
    unsigned long long oldA[8], oldB[8];
    uint32_t mA[8], mB[8];
    constexpr bool kFourBatchIterations = true;
    // 8 steps = two cache words of one-byte entries, or all four words of a batch of two-byte entries / of the 8-query-wide reading
    auto issue = [&](unsigned long long (&old)[8], uint32_t (&m)[8], uint32_t wa, uint32_t wb, uint32_t wc, uint32_t wd) {
        if constexpr (EB == 16) {                                    // a 12-bit record: wa, wb, wc; entries 2 and 5 straddle a dword (wd is not used)
            const uint32_t sx = __builtin_amdgcn_alignbit(wb, wa, 24), sy = __builtin_amdgcn_alignbit(wc, wb, 28);
#pragma unroll
            for (int u = 0; u < 8; ++u) issue1(XMH_CACHE12_WORD(u, wa, wb, wc, sx, sy), XMH_CACHE12_BIT(u), old[u], m[u]);
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int wi = u / EPW;
                issue1(wi == 0 ? wa : (wi == 1 ? wb : (wi == 2 ? wc : wd)), u % EPW, old[u], m[u]);
            }
        }
    };
    auto drain8 = [&](unsigned long long (&old)[8], uint32_t (&m)[8]) {      // this set's returns are in: 8 newer LDS operations are in flight
        asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(old[0]), "+v"(old[1]), "+v"(old[2]), "+v"(old[3]), "+v"(old[4]), "+v"(old[5]), "+v"(old[6]), "+v"(old[7])::"memory");
#pragma unroll
        for (int u = 0; u < 8; ++u) credit(old[u], m[u]);
    };
    auto drain0 = [&](unsigned long long (&old)[8], uint32_t (&m)[8]) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(old[0]), "+v"(old[1]), "+v"(old[2]), "+v"(old[3]), "+v"(old[4]), "+v"(old[5]), "+v"(old[6]), "+v"(old[7])::"memory");
#pragma unroll
        for (int u = 0; u < 8; ++u) credit(old[u], m[u]);
    };
    const int nbatch = (a.chunk + 63) >> 6;
    const uint4* crow = HALF ? a.pair_cache + ((int64_t)chunk_id * (a.nqt >> 1) + (qtile >> 1)) * nbatch * 64 + (slot & 3) * 16 + (qtile & 1) * 8 + ql
                             : a.pair_cache + ((int64_t)chunk_id * a.nqt + qtile) * nbatch * 64 + lane;
    const int nfull = (int)((hi - lo) >> 6);                         // whole batches of this chunk
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the counters are in place before the first asm atomic is counted
    // the cache words are fetched TWO batches ahead: a batch is ~1.2 us of this wave's time at 5 waves per SIMD, an HBM miss under load
    // takes longer than that (Q 5000 x R 117 218, 64 bit, pass 2 with the words one / two / three batches ahead: 0.187 / 0.178 / 0.181 ms)
    // the cache is read once, 1 KB contiguous per wave instruction: the non-temporal hint pays on exactly this form (round 5, see
    // k_topk_filter_seq): pass 2 at the headline shape 0.192-0.194 -> 0.185-0.187 ms in the ablation harness (XMH_ABL_AP_PLAIN = without)
    const uint32_t* crow12 = reinterpret_cast<const uint32_t*>(a.pair_cache) + (((int64_t)chunk_id * a.nqt + qtile) * nbatch * 64 + lane) * xmh::kCache12Dwords;
    auto cache_words = [&](int64_t batch) -> uint4 {
        if constexpr (EB == 16) {                                    // 12-bit records: three dwords per lane, 768 B per wave instruction
            typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));
            const u32x3_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x3_t*>(crow12 + batch * 64 * xmh::kCache12Dwords));
            return make_uint4(v.x, v.y, v.z, 0u);
        } else {
#ifdef XMH_ABL_AP_PLAIN
            return crow[batch * 64];
#else
            typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
            const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(crow + batch * 64));
            return make_uint4(v.x, v.y, v.z, v.w);
#endif
        }
    };
    uint4 cw = cache_words(0);
    uint4 nw = cache_words(1 < nbatch ? 1 : 0);
    auto next_words = [&](int bi) {                                  // unconditional: counted vmcnt, no predication
#ifdef XMH_ABL_AP_NOLOAD
        const uint4 nw2 = make_uint4(0x10203040u + bi + lane, 0x18283848u ^ lane, 0x11223344u + 3 * lane, 0x21314151u + bi);      // (ablation: no cache stream)
#else
        const uint4 nw2 = cache_words(bi + 2 < nbatch ? bi + 2 : nbatch - 1);
#endif
        return nw2;
    };
    // the steps of ONE batch whose first `count` items exist, through compiler-placed atomics (the ragged last batch)
    auto slow_batch = [&](const uint4& w4, int count) {
        const uint32_t sx = EB == 16 ? __builtin_amdgcn_alignbit(w4.y, w4.x, 24) : 0u, sy = EB == 16 ? __builtin_amdgcn_alignbit(w4.z, w4.y, 28) : 0u;
#pragma unroll
        for (int t = 0; t < QW; ++t) {
            if (t * S + slot < count) {
                const uint32_t w = EB == 16 ? XMH_CACHE12_WORD(t, w4.x, w4.y, w4.z, sx, sy) : (t < EPW ? w4.x : (t < 2 * EPW ? w4.y : (t < 3 * EPW ? w4.z : w4.w)));
                const uint32_t bit = EB == 16 ? (uint32_t)XMH_CACHE12_BIT(t) : (HALF ? hoff + 16u * (t % EPW) : (uint32_t)(EB * (t % EPW)));
                const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)w, bit, 1), d = __builtin_amdgcn_ubfe(w, bit + 1u, EB == 16 ? 11 : EB - 1);
                const unsigned long long o = atomicAdd(&cnt[d * QW + ql], 1ull | ((unsigned long long)m << 32));
                credit(o, m);
            }
        }
    };
    if (EB == 8 && !HALF) {
        // two groups per batch (A = its first two words, B = the other two); an iteration issues and credits two batches and ends drained,
        // like the one-group form below
        int bi = 0;
        if (kFourBatchIterations) {
            for (; bi + 3 < nfull; bi += 4) {
                const uint4 n0 = next_words(bi);
                issue(oldA, mA, cw.x, cw.y, 0u, 0u);
                issue(oldB, mB, cw.z, cw.w, 0u, 0u);
                drain8(oldA, mA);
                cw = nw;
                nw = n0;
#pragma unroll
                for (int t = 1; t < 4; ++t) {
                    const uint4 nt = next_words(bi + t);
                    issue(oldA, mA, cw.x, cw.y, 0u, 0u);
                    drain8(oldB, mB);
                    issue(oldB, mB, cw.z, cw.w, 0u, 0u);
                    drain8(oldA, mA);
                    cw = nw;
                    nw = nt;
                }
                drain0(oldB, mB);
            }
        }
        for (; bi + 1 < nfull; bi += 2) {
            const uint4 nw2 = next_words(bi);
            issue(oldA, mA, cw.x, cw.y, 0u, 0u);
            issue(oldB, mB, cw.z, cw.w, 0u, 0u);
            drain8(oldA, mA);
            cw = nw;
            nw = nw2;
            const uint4 nw3 = next_words(bi + 1);
            issue(oldA, mA, cw.x, cw.y, 0u, 0u);
            drain8(oldB, mB);
            issue(oldB, mB, cw.z, cw.w, 0u, 0u);
            drain8(oldA, mA);
            drain0(oldB, mB);
            cw = nw;
            nw = nw3;
        }
        if (bi < nfull) {
            const uint4 nw2 = next_words(bi);
            issue(oldA, mA, cw.x, cw.y, 0u, 0u);
            issue(oldB, mB, cw.z, cw.w, 0u, 0u);
            drain8(oldA, mA);
            drain0(oldB, mB);
            cw = nw;
            nw = nw2;
        }
    } else {
        // one group per batch (two-byte entries, or the 8-query-wide reading).  No result crosses a loop edge or a branch: an iteration
        // issues and credits four batches (A B A B, each credited while the next one's atomics are in flight) and ends drained; the
        // remainder does the same with two batches, then one.  One LDS round trip exposed per four batches is the price of results that
        // hipcc cannot be tempted to copy early.
        // Round 6: the cache words are fetched FOUR batches ahead here.  A batch of this form is 8 steps (half the wave time of the one-byte
        // form's 16) and with 257 bucket rows of 64-bit counters only nine waves fit a CU.  configs[4]'s shard (12.5 GB of pair cache per
        // pass), words two / four / eight batches ahead: 2.58 / 2.47 / 3.2 ms.  An iteration takes four batches, so the four word registers
        // keep their roles and no copy is needed: r_j holds batch bi + j and is refilled with batch bi + j + 4 once its atomics are issued.
        auto words_at = [&](int b) { return cache_words(b < nbatch ? b : nbatch - 1); };
        auto issue4 = [&](unsigned long long (&old)[8], uint32_t (&m)[8], const uint4& r) { issue(old, m, r.x, r.y, r.z, r.w); };
        uint4 r0 = cw, r1 = nw, r2 = words_at(2), r3 = words_at(3);
        int bi = 0;
        for (; bi + 3 < nfull; bi += 4) {
            issue4(oldA, mA, r0);
            r0 = words_at(bi + 4);
            issue4(oldB, mB, r1);
            r1 = words_at(bi + 5);
            drain8(oldA, mA);
            issue4(oldA, mA, r2);
            r2 = words_at(bi + 6);
            drain8(oldB, mB);
            issue4(oldB, mB, r3);
            r3 = words_at(bi + 7);
            drain8(oldA, mA);
            drain0(oldB, mB);
        }
        const int rem = nfull - bi;                                  // 0 .. 3 whole batches left, in r0, r1, r2; the ragged one's words follow them
        if (rem >= 2) {
            issue4(oldA, mA, r0);
            issue4(oldB, mB, r1);
            drain8(oldA, mA);
            drain0(oldB, mB);
            if (rem == 3) {
                issue4(oldA, mA, r2);
                drain0(oldA, mA);
            }
        } else if (rem == 1) {
            issue4(oldA, mA, r0);
            drain0(oldA, mA);
        }
        cw = rem == 0 ? r0 : (rem == 1 ? r1 : (rem == 2 ? r2 : r3));
    }
    const int cntb = (int)(hi - lo) - nfull * 64;                    // ragged last batch (cw holds its words)
    slow_batch(cw, cntb);
#pragma unroll
    for (int o = QW; o < 64; o <<= 1) acc += __shfl_xor(acc, o, 64);
    if (slot == 0) ap_part[(int64_t)chunk_id * a.qpad + q] = acc;
}

// ---------------------------------------------------------------------------------------------------
// k_scan_ap_r2 (round 5): pass 2 WITHOUT a pair cache -- the operands of the i8 MFMAs built in registers from the packed words exactly as
// in k_scan_hist_r2 (same load_words / build, same query bytes for the address chain), the MFMA emits the LDS address of counter
// [distance][query], and the pair's relevance comes out of the label chain: query label bytes are 0 / -1, so the chain ends at minus
// the (weighted) number of common labels and v_max_i32(acc, -1) is the mask 0 / ~0 -- which is both the high half of the 64-bit
// increment {1, -relevant} of k_scan_ap_c's float-bit counters and the AND mask of its credit.  One statement per (16 items x 16
// queries) holds that pair's MFMAs with the consumers of the PREVIOUS pair between them (four v_max, four ds_add_rtn_u64), as in
// k_scan_hist_m2 (same hazards, same spacing: see there); the returns are credited two statements later behind a counted lgkmcnt
// (k_scan_ap_c's arithmetic in k_scan_ap_c's order per lane: lane (slot, query) still owns the items = slot mod 4, ascending, so the
// per-chunk sums are bit-identical to the cached path's).  Nothing crosses a batch: an iteration ends drained (tools/isa_hazards.py R5).
//   * the increment pairs {1, mask} live in PINNED registers (v[112:127], two sets used in turn): the statement writes the high halves by
//     name -- inline asm has no way to address half of a 64-bit operand -- and the low halves keep their 1 for the whole kernel;
//   * counter rows are k_scan_ap_c's: 16 queries x 8 B = 128 bytes, so that the 16 lanes of one slot cover all 32 banks once (rows of 64
//     bytes measured 49 % of the LDS cycles as bank conflicts).  A product of +128 does not fit an i8 operand: the item bytes are worth 2 and
//     4 instead of 1 and 2 against the query bytes +-64 / +-32 of pass 1.
// ---------------------------------------------------------------------------------------------------
#define XMH_AP2_SETA "v113", "v115", "v117", "v119", "v[112:113]", "v[114:115]", "v[116:117]", "v[118:119]"
#define XMH_AP2_SETB "v121", "v123", "v125", "v127", "v[120:121]", "v[122:123]", "v[124:125]", "v[126:127]"
#define XMH_AP2_MFMA(D, A, B, C) "v_mfma_i32_16x16x64_i8 %[" D "], %[" A "], %[" B "], " C "\n\t"
#define XMH_AP2_MAX(H, L) "v_max_i32 " H ", -1, %[" L "]\n\t"
#define XMH_AP2_ADD(O, A, P) "ds_add_rtn_u64 %[" O "], %[" A "], " P "\n\t"
// the consumers of the previous pair between the MFMAs of this one (two label tiles / one label tile), and alone (the last pair of a batch)
#define XMH_AP2_FUSED2_(H0, H1, H2, H3, P0, P1, P2, P3)                                                                                   \
    "s_nop 3\n\t" XMH_AP2_MFMA("lab", "a1", "q1", "0") XMH_AP2_MAX(H0, "l0") XMH_AP2_MAX(H1, "l1") XMH_AP2_MAX(H2, "l2")                \
    XMH_AP2_MFMA("lab", "a2", "q2", "%[lab]") XMH_AP2_MAX(H3, "l3") XMH_AP2_ADD("o0", "p0", P0) XMH_AP2_ADD("o1", "p1", P1)            \
    XMH_AP2_MFMA("addr", "a0", "q0", "%[c0]") XMH_AP2_ADD("o2", "p2", P2) XMH_AP2_ADD("o3", "p3", P3)
#define XMH_AP2_FUSED1_(H0, H1, H2, H3, P0, P1, P2, P3)                                                                                   \
    "s_nop 3\n\t" XMH_AP2_MFMA("lab", "a1", "q1", "0") XMH_AP2_MAX(H0, "l0") XMH_AP2_MAX(H1, "l1") XMH_AP2_MAX(H2, "l2") XMH_AP2_MAX(H3, "l3") \
    XMH_AP2_MFMA("addr", "a0", "q0", "%[c0]") XMH_AP2_ADD("o0", "p0", P0) XMH_AP2_ADD("o1", "p1", P1) XMH_AP2_ADD("o2", "p2", P2)       \
    XMH_AP2_ADD("o3", "p3", P3)
#define XMH_AP2_TAIL_(H0, H1, H2, H3, P0, P1, P2, P3)                                                                                     \
    XMH_AP2_MAX(H0, "l0") XMH_AP2_MAX(H1, "l1") XMH_AP2_MAX(H2, "l2") XMH_AP2_MAX(H3, "l3") XMH_AP2_ADD("o0", "p0", P0)                   \
    XMH_AP2_ADD("o1", "p1", P1) XMH_AP2_ADD("o2", "p2", P2) XMH_AP2_ADD("o3", "p3", P3)
#define XMH_AP2_FUSED2(...) XMH_AP2_FUSED2_(__VA_ARGS__)
#define XMH_AP2_FUSED1(...) XMH_AP2_FUSED1_(__VA_ARGS__)
#define XMH_AP2_TAIL(...) XMH_AP2_TAIL_(__VA_ARGS__)
template <int NML, int NW, int NQ, bool CAPPED>
__global__ __launch_bounds__(64 * NW) void k_scan_ap_r2(MfmaArgs a, const uint2* __restrict__ below, const uint2* __restrict__ dpre,
                                                        const uint32_t* __restrict__ cap_ws, float* __restrict__ ap_part,
                                                        const uint32_t* __restrict__ items_total, uint32_t kcap) {
    using u64 = unsigned long long;
    constexpr int NMI = 1 + NML;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];   // NW x NQ x [nb][16] 64-bit float-bit counters
    int chunk_id, qtile;
    if (!mfma_map_block(a, chunk_id, qtile)) return;                 // a.nqt counts tiles of NW * NQ * 16 queries here
    if (items_total && (int64_t)*items_total > kFloatBitsMaxItems) return;      // sharded call: the integer-counter kernel takes it
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ql = lane & 15, slot = lane >> 4;
    const int t16 = (qtile * NW + wave) * NQ;                        // first 16-query group of this wave
    const int ncell = a.nb * 16;
    u64* cnt = reinterpret_cast<u64*>(lds) + (wave * NQ) * ncell;
    auto pack = [](uint2 x, uint2 y) {                               // k_scan_ap_c's counters: bits(2^23 + rank), bits(2^24 - 1 - ordinal)
        return (u64)(kF23 + x.x + y.x + 1u) | ((u64)(kF23 + (0x7fffffu - (x.y + y.y + 1u))) << 32);
    };
#pragma unroll
    for (int h = 0; h < NQ; ++h) {
        const uint2* __restrict__ pb = below + ((int64_t)chunk_id * a.nb) * a.qpad + (t16 + h) * 16;
        const uint2* __restrict__ pd = dpre + (t16 + h) * 16;
        u64* c = cnt + h * ncell;
        auto cell = [&](int e) { return e; };                          // [bucket][16 queries]: rows of 128 bytes
        int e = lane;
        for (; e + 3 * 64 < ncell; e += 4 * 64) {
            uint2 x[4], y[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ee = e + j * 64;
                const int64_t at = (int64_t)(ee >> 4) * a.qpad + (ee & 15);
                x[j] = pb[at];
                y[j] = pd[at];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) c[cell(e + j * 64)] = pack(x[j], y[j]);
        }
        for (; e < ncell; e += 64) {
            const int64_t at = (int64_t)(e >> 4) * a.qpad + (e & 15);
            c[cell(e)] = pack(pb[at], pd[at]);
        }
    }
    // query operands: the address chain with k_scan_hist_r2's query bytes (+-64 even registers, +-32 odd ones; the item bytes are doubled, see
    // build), started 128 popcount(q) above the lane's counter; the label tiles 0 / -1
    v4i bq[NQ], bl[NQ][NML], cq[NQ];
    float capf[NQ], acc[NQ];
    const int rsh = 4 * (slot & 1), rwi = slot >> 1;
#pragma unroll
    for (int h = 0; h < NQ; ++h) {
        const int64_t q = (int64_t)(t16 + h) * 16 + ql;
        const bool valid = q < a.Q;
        int pcq = 0;
        uint32_t qw = 0u;
        if (valid) {
            for (int w = 0; w < a.W; ++w) pcq += __popc(a.qbits[q * a.W + w]);
            if (rwi < a.W) qw = a.qbits[q * a.W + rwi];
        }
        qw >>= rsh;
        const bool on = valid && rwi < a.W;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t t = (qw >> j) & 0x01010101u;
            if (j & 1) bq[h][j] = on ? (int)(0x20202020u ^ (t * 0xc0u)) : 0;       // +32 / -32
            else bq[h][j] = on ? (int)(0x40404040u ^ (t << 7)) : 0;                // +64 / -64
        }
#pragma unroll
        for (int m = 0; m < NML; ++m) {
            uint32_t lw = 0u;
            if (valid && 2 * m + rwi < a.LW) lw = a.qlab[q * a.LW + 2 * m + rwi];
            lw >>= rsh;
#pragma unroll
            for (int j = 0; j < 4; ++j) bl[h][m][j] = (int)(((lw >> j) & 0x01010101u) * 0xffu);
        }
        const int c0 = (int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) u64*)(cnt + h * ncell) + ql * 8 + (valid ? 128 * pcq : 0);
        cq[h] = v4i{c0, c0, c0, c0};
        asm volatile("" : "+v"(cq[h]));                                // opaque: kept in VGPRs (see k_scan_hist_m2, hazard iii)
        capf[h] = CAPPED ? (float)min(cap_ws[q], kcap) : 0.0f;         // exact: below 2^23
        acc[h] = 0.0f;
    }
    const int64_t lo = (int64_t)chunk_id * a.chunk;
    const int64_t hi = (lo + a.chunk < a.R) ? lo + a.chunk : a.R;
    const int nbat = (int)((hi - lo + 63) >> 6);
    const int64_t bat0 = lo >> 6;                                    // chunks start on 64-item boundaries
    uint32_t wcur[4][NMI], wnxt[4][NMI];
    const int ritem = 4 * (lane & 3) + ((lane & 15) >> 2);
    const int wi_c = rwi < a.W ? rwi : a.W - 1;
    int wi_l[NML];
#pragma unroll
    for (int m = 0; m < NML; ++m) wi_l[m] = 2 * m + rwi < a.LW ? 2 * m + rwi : (a.LW > 0 ? a.LW - 1 : 0);
    auto load_words = [&](int64_t batch, uint32_t (&w)[4][NMI]) {       // k_scan_hist_r2's: whole batches without clamps or masks
        const int64_t first = batch * 64;
        if (first + 64 <= (int64_t)a.R) {
            const uint32_t* __restrict__ pc = a.rbits + (first + ritem) * a.W + wi_c;
            const uint32_t* __restrict__ pl = a.rlab + (first + ritem) * a.LW;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                w[g][0] = pc[g * 16 * a.W];
#pragma unroll
                for (int m = 0; m < NML; ++m) w[g][1 + m] = pl[g * 16 * a.LW + wi_l[m]];
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t item = first + g * 16 + ritem;
                const int64_t it = item < a.R ? item : (int64_t)a.R - 1;
                const uint32_t ok = item < a.R ? 0xffffffffu : 0u;   // items past the end: all-zero codes, no labels -- they come after every real item
                w[g][0] = a.rbits[it * a.W + wi_c] & ok;             // of their bucket and are never relevant, so they change no credit
#pragma unroll
                for (int m = 0; m < NML; ++m) w[g][1 + m] = a.rlab[it * a.LW + wi_l[m]] & ok;
            }
        }
    };
    // code tile: item bytes worth 2 (even registers) and 4 (odd ones) against the query bytes +-64 / +-32: products of +-128 = one counter row.
    // Two rotations bring bits (0, 1) and (2, 3) of the lane's nibble to bits 1, 2 of their bytes (what wraps around lands outside the masks).
    const uint32_t rot_a = (uint32_t)(rsh + 31) & 31u, rot_b = (uint32_t)rsh + 1u;
    auto build = [&](v4i (&At)[NMI], const uint32_t (&w)[NMI]) {
        const uint32_t xa = __builtin_amdgcn_alignbit(w[0], w[0], rot_a), xb = __builtin_amdgcn_alignbit(w[0], w[0], rot_b);
        At[0][0] = (int)(xa & 0x02020202u);
        At[0][1] = (int)(xa & 0x04040404u);
        At[0][2] = (int)(xb & 0x02020202u);
        At[0][3] = (int)(xb & 0x04040404u);
#pragma unroll
        for (int m = 1; m < NMI; ++m) {
            const uint32_t y = w[m] >> rsh;
#pragma unroll
            for (int j = 0; j < 4; ++j) At[m][j] = (int)(y & (0x01010101u << j));
        }
    };
    auto credit = [&](u64 old, uint32_t m, int h) {                  // k_scan_ap_c's, operation for operation
        const float rank = __uint_as_float((uint32_t)old) - 8388608.0f;
        const float ord = 16777215.0f - __uint_as_float((uint32_t)(old >> 32));
        if (CAPPED) m = ord <= capf[h] ? m : 0u;
        acc[h] = fmaf(ord, __uint_as_float(__float_as_uint(__builtin_amdgcn_rcpf(rank)) & m), acc[h]);
    };
    // increment pairs {1, mask}: two sets of four, pinned (see the header); set = pair index & 1
    v4i incA01 = {1, 0, 1, 0}, incA23 = {1, 0, 1, 0}, incB01 = {1, 0, 1, 0}, incB23 = {1, 0, 1, 0};
    asm volatile("" : "+{v[112:115]}"(incA01), "+{v[116:119]}"(incA23), "+{v[120:123]}"(incB01), "+{v[124:127]}"(incB23));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the counters are in place (own wave's region only) before the first asm atomic
    load_words(bat0, wcur);
    for (int i = 0; i < nbat; ++i) {
        load_words(bat0 + (i + 1 < nbat ? i + 1 : i), wnxt);
        v4i A[4][NMI];                                               // one tile set per item group, each kept alive one statement past its last MFMA
        build(A[0], wcur[0]);
        build(A[1], wcur[1]);
        build(A[2], wcur[2]);
        build(A[3], wcur[3]);
        u64 oldA[4], oldB[4];
        v4i addr_p, lab_p;
        auto evaluate = [&](const v4i (&At)[NMI], int h, v4i& addr, v4i& lab) {      // first pair of a batch: nothing to consume; closed by 8 wait states
            if constexpr (NML == 2)
                asm volatile("s_nop 3\n\t" XMH_AP2_MFMA("lab", "a1", "q1", "0") XMH_AP2_MFMA("lab", "a2", "q2", "%[lab]") XMH_AP2_MFMA("addr", "a0", "q0", "%[c0]") "s_nop 7"
                             : [lab] "=&v"(lab), [addr] "=&v"(addr)
                             : [a1] "v"(At[1]), [q1] "v"(bl[h][0]), [a2] "v"(At[NMI - 1]), [q2] "v"(bl[h][NML - 1]), [a0] "v"(At[0]), [q0] "v"(bq[h]), [c0] "v"(cq[h]));
            else
                asm volatile("s_nop 3\n\t" XMH_AP2_MFMA("lab", "a1", "q1", "0") XMH_AP2_MFMA("addr", "a0", "q0", "%[c0]") "s_nop 7"
                             : [lab] "=&v"(lab), [addr] "=&v"(addr)
                             : [a1] "v"(At[1]), [q1] "v"(bl[h][0]), [a0] "v"(At[0]), [q0] "v"(bq[h]), [c0] "v"(cq[h]));
        };
#define XMH_AP2_OUTS(OLD) [lab] "=&v"(lab), [addr] "=&v"(addr), [o0] "=&v"(OLD[0]), [o1] "=&v"(OLD[1]), [o2] "=&v"(OLD[2]), [o3] "=&v"(OLD[3])
#define XMH_AP2_PREV [p0] "v"(addr_p[0]), [p1] "v"(addr_p[1]), [p2] "v"(addr_p[2]), [p3] "v"(addr_p[3]), [l0] "v"(lab_p[0]), [l1] "v"(lab_p[1]), [l2] "v"(lab_p[2]), [l3] "v"(lab_p[3])
        auto fusedA = [&](const v4i (&At)[NMI], int h, v4i& addr, v4i& lab) {          // consumers into set A
            if constexpr (NML == 2)
                asm volatile(XMH_AP2_FUSED2(XMH_AP2_SETA)
                             : XMH_AP2_OUTS(oldA), "+{v[112:115]}"(incA01), "+{v[116:119]}"(incA23)
                             : [a1] "v"(At[1]), [q1] "v"(bl[h][0]), [a2] "v"(At[NMI - 1]), [q2] "v"(bl[h][NML - 1]), [a0] "v"(At[0]), [q0] "v"(bq[h]), [c0] "v"(cq[h]), XMH_AP2_PREV
                             : "memory");
            else
                asm volatile(XMH_AP2_FUSED1(XMH_AP2_SETA)
                             : XMH_AP2_OUTS(oldA), "+{v[112:115]}"(incA01), "+{v[116:119]}"(incA23)
                             : [a1] "v"(At[1]), [q1] "v"(bl[h][0]), [a0] "v"(At[0]), [q0] "v"(bq[h]), [c0] "v"(cq[h]), XMH_AP2_PREV
                             : "memory");
        };
        auto fusedB = [&](const v4i (&At)[NMI], int h, v4i& addr, v4i& lab) {          // consumers into set B
            if constexpr (NML == 2)
                asm volatile(XMH_AP2_FUSED2(XMH_AP2_SETB)
                             : XMH_AP2_OUTS(oldB), "+{v[120:123]}"(incB01), "+{v[124:127]}"(incB23)
                             : [a1] "v"(At[1]), [q1] "v"(bl[h][0]), [a2] "v"(At[NMI - 1]), [q2] "v"(bl[h][NML - 1]), [a0] "v"(At[0]), [q0] "v"(bq[h]), [c0] "v"(cq[h]), XMH_AP2_PREV
                             : "memory");
            else
                asm volatile(XMH_AP2_FUSED1(XMH_AP2_SETB)
                             : XMH_AP2_OUTS(oldB), "+{v[120:123]}"(incB01), "+{v[124:127]}"(incB23)
                             : [a1] "v"(At[1]), [q1] "v"(bl[h][0]), [a0] "v"(At[0]), [q0] "v"(bq[h]), [c0] "v"(cq[h]), XMH_AP2_PREV
                             : "memory");
        };
        // the returns of set A / B are in (4 newer LDS operations in flight at most): credit them to query group h
        auto drainA = [&](int h, auto newer) {
            if constexpr (decltype(newer)::value == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(oldA[0]), "+v"(oldA[1]), "+v"(oldA[2]), "+v"(oldA[3])::"memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(oldA[0]), "+v"(oldA[1]), "+v"(oldA[2]), "+v"(oldA[3])::"memory");
            credit(oldA[0], (uint32_t)incA01[1], h);
            credit(oldA[1], (uint32_t)incA01[3], h);
            credit(oldA[2], (uint32_t)incA23[1], h);
            credit(oldA[3], (uint32_t)incA23[3], h);
        };
        auto drainB = [&](int h, auto newer) {
            if constexpr (decltype(newer)::value == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(oldB[0]), "+v"(oldB[1]), "+v"(oldB[2]), "+v"(oldB[3])::"memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(oldB[0]), "+v"(oldB[1]), "+v"(oldB[2]), "+v"(oldB[3])::"memory");
            credit(oldB[0], (uint32_t)incB01[1], h);
            credit(oldB[1], (uint32_t)incB01[3], h);
            credit(oldB[2], (uint32_t)incB23[1], h);
            credit(oldB[3], (uint32_t)incB23[3], h);
        };
        using N4 = std::integral_constant<int, 4>;
        using N0 = std::integral_constant<int, 0>;
        // pair P = g * NQ + h.  Statement P evaluates pair P and issues the atomics of pair P - 1 into set (P - 1) & 1; behind it the returns of
        // pair P - 2 (the same set as the NEXT statement writes) are credited.
        auto step = [&](auto pc) {
            constexpr int P = decltype(pc)::value, G = P / NQ, H = P % NQ;
            v4i addr, lab;
            if constexpr (P == 0) evaluate(A[0], 0, addr, lab);
            else if constexpr ((P - 1) & 1) fusedB(A[G], H, addr, lab);
            else fusedA(A[G], H, addr, lab);
            addr_p = addr; lab_p = lab;
            if constexpr (G > 0 && H == 0) {                          // the previous group's tiles may be reused from here on, not earlier
                if constexpr (NMI == 2) asm volatile("" ::"v"(A[G - 1][0]), "v"(A[G - 1][1]));
                else asm volatile("" ::"v"(A[G - 1][0]), "v"(A[G - 1][1]), "v"(A[G - 1][NMI - 1]));
            }
            if constexpr (P >= 2) {
                if constexpr ((P - 2) & 1) drainB((P - 2) % NQ, N4{});
                else drainA((P - 2) % NQ, N4{});
            }
        };
        auto run = [&](auto self, auto pc) -> void {
            step(pc);
            if constexpr (decltype(pc)::value + 1 < 4 * NQ) self(self, std::integral_constant<int, decltype(pc)::value + 1>{});
        };
        run(run, std::integral_constant<int, 0>{});
        asm volatile("s_nop 7\n\ts_nop 3" ::: "memory");              // the last MFMAs' results: 8 wait states before a VALU / DS read
        {
            constexpr int PL = 4 * NQ - 1;                            // the last pair's atomics, then everything drains
            static_assert(PL & 1, "the last pair of a batch goes to set B");
            asm volatile(XMH_AP2_TAIL(XMH_AP2_SETB)
                         : [o0] "=&v"(oldB[0]), [o1] "=&v"(oldB[1]), [o2] "=&v"(oldB[2]), [o3] "=&v"(oldB[3]), "+{v[120:123]}"(incB01), "+{v[124:127]}"(incB23)
                         : XMH_AP2_PREV, "v"(A[3][0]), "v"(A[3][1]), "v"(A[3][NMI - 1])
                         : "memory");
            drainA((PL - 1) % NQ, N4{});
            drainB(PL % NQ, N0{});
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int m = 0; m < NMI; ++m) wcur[g][m] = wnxt[g][m];
    }
#undef XMH_AP2_OUTS
#undef XMH_AP2_PREV
#pragma unroll
    for (int h = 0; h < NQ; ++h) {
        float s = acc[h];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (slot == 0) ap_part[(int64_t)chunk_id * a.qpad + (t16 + h) * 16 + ql] = s;
    }
}


}  // namespace
