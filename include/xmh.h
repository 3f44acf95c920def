/*
 * xmh.h -- C ABI of libxmh.so, the MI355X (gfx950) encode-and-retrieve library that sits behind the
 * plugin surface of kalenforn/clip-based-cross-modal-hash.
 *
 * Conventions:
 *   - every entry point is extern "C", takes plain pointers and sizes, returns 0 on success or a
 *     negative errno-style code; nothing throws, nothing is owned by the library, no global state
 *     except a thread-local last-error string (xmh_last_error).
 *   - pointers are DEVICE pointers unless the parameter name ends in _host.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, stream-ordered, and the
 *     call returns without synchronising.
 *   - codes are bit-packed: word w bit j of `bits[n][W]` (W = ceil(K/32)) is 1 iff code[n][32w+j] > 0;
 *     `zero[n][W]` marks code == 0 (sign(0) = 0, reference runners/base.py:410) and has its padding
 *     bits (positions >= K) SET so that padded positions never count.
 *   - labels are bit-packed multi-hot masks `lab[n][Lw]`, Lw = ceil(C/32).
 *
 * Each declaration cites the reference interface it replaces (file:line under the reference repo).
 */
#ifndef XMH_H
#define XMH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XMH_OK 0
#define XMH_EINVAL (-22)
#define XMH_ENOMEM (-12)
#define XMH_ENOTSUP (-95)
#define XMH_EHIP (-5)

/* element types accepted for label / mask inputs */
#define XMH_DT_F32 0
#define XMH_DT_I64 1
#define XMH_DT_I32 2
#define XMH_DT_U8 3

typedef void* xmh_stream_t;

int xmh_version(void);
/* identity of the sources this library was built from: the first 16 hex digits of sha256 over the .hip and .h files of csrc/ and include/xmh.h
 * (the Makefile's SRCID).  bench.py recomputes it from the files beside the library and reports whether they match. */
const char* xmh_build_id(void);
/* thread-local description of the last failure on this thread ("" if none) */
const char* xmh_last_error(void);

/* Per-kernel timing for bench.py: when enabled, the library brackets its dominant kernels with HIP events
 * on the launch stream (names: "scan_hist", "scan_ap", "topk_filter", "gemm", ...).  xmh_prof_read blocks
 * on the last recorded event of that name and returns the mean launch duration since xmh_prof_enable(1). */
int xmh_prof_enable(int on);
int xmh_prof_read(const char* name_host, double* avg_ms_host, int64_t* launches_host);
/* roctx ranges (SURVEY 5 "Build adds"; round 6).  `on` of xmh_prof_enable is a bit set: 1 = the HIP events above, 2 = roctx ranges
 * around every phase of the path -- tower forward, head, pack, pass 1, pass 2, the top-k phases, the float route --, so that a
 * `rocprofv3 --marker-trace --kernel-trace` timeline of valid() (reference runners/base.py:307-357) carries phase markers.  The roctx
 * library (librocprofiler-sdk-roctx.so, else libroctx64.so) is looked up with dlopen when bit 2 is first asked for (XMH_ENOTSUP if
 * absent); libxmh.so does not link it.  xmh_range_push / _pop: the same ranges for the host layer above the C ABI (the collectives of
 * the sharded evaluation, the encode loop); no-ops while ranges are off. */
int xmh_range_push(const char* name_host);
int xmh_range_pop(void);
/* Debug mode for a new device / compiler (round 6).  The fast ranking kernels rely on two things no ISA document promises: same-address
 * lanes of one returning LDS add are served in lane order (probed per device: one wave alone, then pass 2's geometry under load beside
 * matrix waves), and hand-kept hazards around inline MFMA statements (a self-check scan per device).  While xmh_scan_verify(1) is on,
 * every UNSHARDED xmh_hamming_ap / xmh_hamming_map / xmh_calc_map_k is followed by a second derivation of the same evaluation with the
 * masked VALU kernels, which rely on neither; divisors must agree bit for bit and the per-query AP sums to float rounding, else the
 * call returns -74 with the first differing query in xmh_last_error().  Costs about three evaluations and a stream synchronisation
 * per call. */
int xmh_scan_verify(int on);

/* ---------------------------------------------------------------------------------------------
 * Quantisers (a-6).
 * xmh_pack_sign        replaces BaseTrainer.make_hash_code  (runners/base.py:407-410): sign -> -1/0/+1.
 *                      `flags` (device int32, nullable, caller zeroes it) gets bit0 OR-ed in if any element was
 *                      exactly 0 and bit1 if any element was not in {-1,0,+1} (un-quantised input: the
 *                      packed path does not apply).
 * xmh_pack_pair_argmax replaces DCMHTTrainer.make_hash_code (runners/DCMHT/runner.py:82-95):
 *                      probs[n][2K] -> bit j = (probs[2j+1] > probs[2j]) strictly (ties -> -1).
 * xmh_unpack_pm1       packed -> +-1/0 float32 [n][K] (the buffers get_code returns, runners/base.py:245-257,
 *                      and the .mat writer, :386-405).  `zero` may be NULL.
 * `row_index` (device int64[n], nullable): when given, row i of the input is written to packed row
 * row_index[i] (the scatter `buffer[index,:] = code` of runners/base.py:256-257).
 * ------------------------------------------------------------------------------------------- */
int xmh_pack_sign(const float* codes, int64_t n, int K, const int64_t* row_index,
                  uint32_t* bits, uint32_t* zero, int32_t* flags, xmh_stream_t stream);
int xmh_pack_pair_argmax(const float* probs, int64_t n, int K, const int64_t* row_index,
                         uint32_t* bits, xmh_stream_t stream);
int xmh_unpack_pm1(const uint32_t* bits, const uint32_t* zero, int64_t n, int K, float* out,
                   xmh_stream_t stream);
/* multi-hot labels [n][C] of dtype `dt` (XMH_DT_*) -> packed masks [n][ceil(C/32)] (label > 0).
 * Replaces the int64 label matmul operands of calc_map_k (common/calc_utils.py:72) and calc_label_sim (:8-10). */
int xmh_pack_labels(const void* labels, int dt, int64_t n, int C, uint32_t* lab, xmh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Materialised distances (a-1): calc_hammingDist (common/calc_utils.py:51-56).
 *   out_f32[Q][R] = 0.5 * (K - q.r)   (exact for -1/0/+1 codes; pass zero masks as NULL for binary)
 *   out_u16[Q][R] = popcount(q xor r) (binary codes only)
 * Exactly one of out_f32 / out_u16 must be non-NULL.
 * ------------------------------------------------------------------------------------------- */
int xmh_hamming_dist(const uint32_t* qbits, const uint32_t* qzero, const uint32_t* rbits, const uint32_t* rzero,
                     int64_t Q, int64_t R, int K, float* out_f32, uint16_t* out_u16, xmh_stream_t stream);
/* calc_label_sim (common/calc_utils.py:8-10): out[Q][R] = 1.0f iff the two items share a label. */
int xmh_label_sim(const uint32_t* qlab, const uint32_t* rlab, int64_t Q, int64_t R, int C, float* out,
                  xmh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused ranking scan (a-2): calc_map_k (common/calc_utils.py:58-92) without the [Q,R] intermediates.
 *
 * Canonical order = (distance asc, gallery index asc)  == torch.sort(stable=True).
 * Two streaming passes over the gallery shard (a wave owns 64/S queries x S item slots, per-query bucket counters in LDS):
 *   pass 1 (xmh_hamming_hist): per (query, gallery chunk) bucket counts of all / relevant items;
 *   pass 2 (xmh_hamming_ap):   rank of every relevant item = bucket base + running in-bucket count,
 *                              accumulates sum_j j/rank_j for ordinals j <= min(n_rel, k).
 * `xmh_scan_plan` reports the chunking and the workspace both passes share.
 * ------------------------------------------------------------------------------------------- */
typedef struct xmh_scan_plan {
    int64_t chunk;      /* gallery items per chunk                                  */
    int64_t nchunk;     /* chunks in this shard                                     */
    int64_t nqtile;     /* 64-query tiles                                           */
    int64_t qpad;       /* Q rounded up to 64                                       */
    int64_t nbuckets;   /* K+1 (binary) or 2K+1 (ternary, distances in half units)  */
    size_t ws_bytes;    /* workspace the caller must provide to hist/ap             */
} xmh_scan_plan;

int xmh_scan_plan_make(int64_t Q, int64_t R, int K, int ternary, xmh_scan_plan* plan_host);
/* Bytes of the workspace (included in plan.ws_bytes) that hold the PAIR CACHE: xmh_hamming_hist leaves one byte per (query,
 * gallery item) pair -- distance << 1 | relevant; two bytes for codes of 129..256 bits (and the region is sized for two bytes from
 * 65 bits on: codes of 65..128 bits use its first half with one-byte entries, where a distance of 128 wraps to 0 and a control word
 * tells pass 2 to evaluate the pairs itself; XMH_SCAN_BYTE128=0 brings their two-byte entries back) -- and xmh_hamming_ap reads it
 * instead of evaluating the pair again.  Used for binary codes of 33..256 bits while it stays under XMH_SCAN_CACHE_MB (default 131072 MB;
 * 0 = off); returns 0 when it is not used. */
size_t xmh_scan_pair_cache_bytes(int64_t Q, int64_t R, int K, int ternary);
/* Byte offset of that pair cache inside the workspace ((size_t)-1 on a bad shape) -- for tests and diagnostics, which decode it
 * and compare every entry with the oracle's distance and relevance.  Layout for codes of at most 64 bits:
 * [chunk][16-query tile][64-item batch of the chunk][lane 0..63][16 bytes]; lane = slot * 16 + query-in-tile, byte t of a lane
 * is the pair (that query, item 64 * batch + 4 * t + slot of the chunk); entry = distance << 1 | relevant (mod 256).  Longer and ternary codes:
 * [chunk][8-query tile][batch][lane][3 x u32], lane = slot * 8 + query-in-tile, entry t (12 bits, round 6: bits [12 t, 12 t + 12) of the
 * 96) = item 64 * batch + 8 * t + slot. */
size_t xmh_scan_pair_cache_offset(int64_t Q, int64_t R, int K, int ternary);
/* The workspace of the same plan WITHOUT its pair cache.  xmh_hamming_hist / xmh_hamming_ap / xmh_hamming_map take the size they are handed
 * as the decision: a buffer of at least xmh_scan_plan.ws_bytes runs with the cache, one of at least this size (and smaller than that) runs
 * the same plan uncached -- for devices that have no room for Q x R bytes next to a resident encoder.  Both calls of a pair must be given
 * the same size.  (The reference keeps a [Q, R] fp32 matrix and its sort on the host: common/calc_utils.py:76-77.) */
size_t xmh_scan_ws_bytes_nocache(int64_t Q, int64_t R, int K, int ternary);
/* Diagnostics for the measurement harness: writes "pass1=<kernel instance>;pass2=<kernel instance>[|<second width>]" -- the kernels
 * an unsharded mAP@all evaluation of this shape launches, spelled as rocprofv3 prints them -- so that a profile row is matched by
 * its exact name.  out_bytes >= 64. */
int xmh_scan_describe(int64_t Q, int64_t R, int K, int C, int ternary, char* out, size_t out_bytes);

/* pass 1.  hist_all / hist_rel: [Q][nbuckets] u32 totals over this shard (either may be NULL).
 * qzero / rzero NULL => binary codes.
 * Labels: [n][ceil(C / 32)] u32 masks.  C <= 128 (1..4 words) on every path; binary codes up to 256 bits also take exactly EIGHT words
 * (C in 225..256): a caller with 129..224 classes pads its masks to eight words with zero bits and passes C = 256 (Python:
 * RankingScan does).  More classes: XMH_ENOTSUP.
 * Also leaves in the workspace what a single-shard pass 2 needs from the totals (per-bucket rank offsets, the relevant count
 * of every query), so that xmh_hamming_ap / xmh_hamming_map with NULL offsets launch pass 2 and the reduction only. */
int xmh_hamming_hist(const uint32_t* qbits, const uint32_t* qzero, const uint32_t* qlab,
                     const uint32_t* rbits, const uint32_t* rzero, const uint32_t* rlab,
                     int64_t Q, int64_t R, int K, int C, void* ws, size_t ws_bytes,
                     uint32_t* hist_all, uint32_t* hist_rel, xmh_stream_t stream);

/* pass 2 (needs the workspace pass 1 filled for the same inputs).
 *   base_all / base_rel [Q][nbuckets] u32 and nrel_total [Q] u32: rank offsets contributed by items
 *   OUTSIDE this shard (all lower buckets anywhere + same bucket on lower-ranked shards) and the global
 *   relevant count; all three NULL => single shard, the offsets xmh_hamming_hist left in the workspace are used.
 *   (A call WITH offsets overwrites them: run xmh_hamming_hist again before a single-shard evaluation on that workspace.)
 *   Any number of evaluations (different k) may follow one xmh_hamming_hist.
 *   k <= 0 means "all" (k = None in the reference).
 *   ap_sum[Q] (f64) = sum over this shard's relevant items of ordinal/rank, for ordinal <= cap;
 *   cap[Q] (i32)    = min(n_rel, k)  (the divisor, common/calc_utils.py:81). */
int xmh_hamming_ap(const uint32_t* qbits, const uint32_t* qzero, const uint32_t* qlab,
                   const uint32_t* rbits, const uint32_t* rzero, const uint32_t* rlab,
                   int64_t Q, int64_t R, int K, int C, void* ws, size_t ws_bytes,
                   const uint32_t* base_all, const uint32_t* base_rel, const uint32_t* nrel_total,
                   int64_t k, double* ap_sum, int32_t* cap, xmh_stream_t stream);

/* xmh_hamming_ap of an UNSHARDED gallery followed by xmh_map_finalize in one call: ap_sum / cap as xmh_hamming_ap writes
 * them, map_out[0] = mean_q ap_sum[q] / cap[q]. */
int xmh_hamming_map(const uint32_t* qbits, const uint32_t* qzero, const uint32_t* qlab, const uint32_t* rbits,
                    const uint32_t* rzero, const uint32_t* rlab, int64_t Q, int64_t R, int K, int C, void* ws,
                    size_t ws_bytes, int64_t k, double* ap_sum, int32_t* cap, double* map_out, xmh_stream_t stream);
/* calc_map_k (reference common/calc_utils.py:58-92) in ONE call, on the operands the reference's callers hold
 * (runners/base.py:259-264: float code matrices on the device): qB [Q][K] / rB [R][K] float32 codes (-1 / 0 / +1), qlab / rlab packed
 * label masks (xmh_pack_labels: the label matrices are the same for every call of an evaluation, pack them once), k <= 0 = all.
 * Packs both code matrices into the workspace, reads their value flags (first synchronisation), runs both passes with the binary or --
 * an exact zero among the codes -- the ternary kernels, and copies the mAP to *map_host (second synchronisation: the reference
 * returns a host scalar too).  *flags_host: bit 0 = an exact zero was seen, bit 1 = a value outside {-1, 0, +1}: then NOTHING is
 * evaluated (the codes are not quantised: common/calc_utils.py would rank the float inner products) and *map_host is left alone.
 * K must be a code length the kernels take as it is (<= 32, 64, 128, 256, 512, 1024, 2048 bits); C as for xmh_hamming_hist.
 * The same kernels and the same bits as pack + xmh_hamming_hist + xmh_hamming_map. */
size_t xmh_calc_map_k_ws_bytes(int64_t Q, int64_t R, int K, int C);
int xmh_calc_map_k(const float* qB, const float* rB, const uint32_t* qlab, const uint32_t* rlab, int64_t Q, int64_t R, int K, int C,
                   int64_t k, void* ws, size_t ws_bytes, double* map_host, int32_t* flags_host, xmh_stream_t stream);
/* mean over queries of ap_sum/cap -> map_out[0] (f64, device).  A query with cap == 0 makes the result
 * NaN, as torch.mean of an empty tensor does in the reference (common/calc_utils.py:87-89). */
int xmh_map_finalize(const double* ap_sum, const int32_t* cap, int64_t Q, double* map_out, xmh_stream_t stream);
/* Sharded evaluation (replaces the dense all_reduce gather of runners/base.py:259-264 on the retrieval side):
 * hist_gathered = the all-gathered shard totals of xmh_hamming_hist, [world][2][Q][nbuckets] u32 (plane 0 all items,
 * plane 1 relevant).  Writes the three inputs xmh_hamming_ap takes for shard `rank`: base_all/base_rel [Q][nbuckets] =
 * items in lower buckets on any shard + items of the same bucket on lower ranks; nrel_total [Q]. */
int xmh_shard_offsets(const uint32_t* hist_gathered, int world, int rank, int64_t Q, int nbuckets, uint32_t* base_all,
                      uint32_t* base_rel, uint32_t* nrel_total, xmh_stream_t stream);
/* Sharded evaluation without an export pass.  xmh_hamming_hist leaves this shard's totals table in the workspace:
 * [nbuckets][qpad] pairs of u32 {all items, relevant items} at byte offset xmh_scan_totals_offset() (*bytes = its size; the
 * offset depends on Q, K and ternary only through the plan of THIS shard, the table's shape on Q and K alone).  All-gather those
 * tables as they are -> totals_gathered[world][nbuckets][qpad][2], then xmh_hamming_map_sharded = offsets + pass 2 + this
 * shard's share of the mean (three launches): ap_sum[Q] = this shard's credits, cap[Q] = the global divisor,
 * map_partial[0] = (1/Q) sum_q ap_sum[q] / cap[q].  The mAP is the SUM of map_partial over the shards (one 8-byte all-reduce
 * instead of a [Q] f64 one and a finalize launch). */
size_t xmh_scan_totals_offset(int64_t Q, int64_t R, int K, int ternary, size_t* bytes);
int xmh_hamming_map_sharded(const uint32_t* qbits, const uint32_t* qzero, const uint32_t* qlab, const uint32_t* rbits,
                            const uint32_t* rzero, const uint32_t* rlab, int64_t Q, int64_t R, int K, int C, void* ws,
                            size_t ws_bytes, const uint32_t* totals_gathered, int world, int rank, int64_t k, double* ap_sum,
                            int32_t* cap, double* map_partial, xmh_stream_t stream);

/* The same evaluation with an all-to-all exchange (2 x the table size per rank instead of world x): rank j receives from every
 * shard the columns of ITS slice of `slice` = qpad / world queries -> totals_slices[world][nbuckets][slice][2];
 * xmh_shard_slice_offsets resolves those queries for EVERY shard -> offsets_out[world][nbuckets + 1][slice][2]
 * ({lower buckets anywhere + same bucket on lower shards} per bucket; last row {relevant items, items} over all shards); a second
 * all-to-all returns shard w its rows, ordered by slice owner -> offsets[world][nbuckets + 1][slice][2], which
 * xmh_hamming_map_sharded_offsets turns into pass 2 + this shard's share of the mean exactly like xmh_hamming_map_sharded.
 * Needs qpad % world == 0 (qpad is a multiple of 64; 128 for codes of at most 64 bits). */
int xmh_shard_slice_offsets(const uint32_t* totals_slices, int world, int nbuckets, int slice, uint32_t* offsets_out, xmh_stream_t stream);
int xmh_hamming_map_sharded_offsets(const uint32_t* qbits, const uint32_t* qzero, const uint32_t* qlab, const uint32_t* rbits,
                                    const uint32_t* rzero, const uint32_t* rlab, int64_t Q, int64_t R, int K, int C, void* ws,
                                    size_t ws_bytes, const uint32_t* offsets, int world, int64_t k, double* ap_sum, int32_t* cap,
                                    double* map_partial, xmh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Per-query top-k (north_star: "fused bit-packed XOR-popcount + per-query top-k kernel").
 * Exact top-k under the canonical order.  dist[Q][k] u16, idx[Q][k] i32 hold GLOBAL indices
 * (base_index + local row).  k <= 1024.  Workspace: xmh_topk_ws_bytes(Q, R, K, k).
 * ------------------------------------------------------------------------------------------- */
size_t xmh_topk_ws_bytes(int64_t Q, int64_t R, int K, int k);
int xmh_hamming_topk(const uint32_t* qbits, const uint32_t* rbits, int64_t Q, int64_t R, int K, int k,
                     int64_t base_index, void* ws, size_t ws_bytes, uint16_t* dist, int32_t* idx,
                     xmh_stream_t stream);
/* The same call on a PREPARED workspace: xmh_topk_ws_init zeroes the control words and the sample histogram at the head of the
 * workspace once, every xmh_hamming_topk_prepared call on it (same Q, R, K, k; one stream at a time) finds them zero and leaves
 * them zero, so a repeated query loop pays no memset launch.  xmh_hamming_topk == init + prepared call. */
int xmh_topk_ws_init(int64_t Q, int64_t R, int K, int k, void* ws, size_t ws_bytes, xmh_stream_t stream);
int xmh_hamming_topk_prepared(const uint32_t* qbits, const uint32_t* rbits, int64_t Q, int64_t R, int K, int k,
                              int64_t base_index, void* ws, size_t ws_bytes, uint16_t* dist, int32_t* idx,
                              xmh_stream_t stream);
/* Ternary codes (round 6).  BaseTrainer.make_hash_code is an in-place sign_() (reference runners/base.py:407-410): an activation of
 * exactly 0 stays 0, and MITH sums two saturating tanh terms before it (runners/MITH/runner.py:125-131), so the code sets of the
 * MITH / DSPH configs can hold -1 / 0 / +1.  Same call with the zero planes of both sides (bit set <=> the element is 0, padding
 * bits set: what xmh_pack_sign writes); the order is the same canonical (distance, index) order over the reference's
 * 0.5 * (K - q.r) (common/calc_utils.py:51-56), and dist2[Q][k] holds that distance in HALF units (K - q.r, 0 ... 2K; 0xFFFF =
 * unused slot) since it is a multiple of 1/2.  Workspace: xmh_topk_ternary_ws_bytes; `prepared` != 0: the workspace went through
 * xmh_topk_ternary_ws_init (or an earlier call) and needs no memset. */
size_t xmh_topk_ternary_ws_bytes(int64_t Q, int64_t R, int K, int k);
int xmh_topk_ternary_ws_init(int64_t Q, int64_t R, int K, int k, void* ws, size_t ws_bytes, xmh_stream_t stream);
int xmh_hamming_topk_ternary(const uint32_t* qbits, const uint32_t* qzero, const uint32_t* rbits, const uint32_t* rzero,
                             int64_t Q, int64_t R, int K, int k, int64_t base_index, void* ws, size_t ws_bytes, int prepared,
                             uint16_t* dist2, int32_t* idx, xmh_stream_t stream);
/* Diagnostics for the measurement harness: writes "filter=<kernel instance>" -- the streaming kernel the fast path of
 * xmh_hamming_topk launches for this shape, spelled as rocprofv3 prints it (the counterpart of xmh_scan_describe).  out_bytes >= 64. */
int xmh_topk_describe(int64_t Q, int64_t R, int K, int k, char* out, size_t out_bytes);
/* Host-side k-way merge of the shards' exact top-k lists (north_star: "partial top-k lists are merged on the host"; replaces the
 * reference's gather of the whole code matrix, runners/base.py:259-264, on the retrieval side).  HOST pointers, no GPU work:
 * gathered_host = `world` records as the ranks all-gathered them, each [Q][k] i32 global indices followed by [Q][k] u16
 * distances (xmh_topk_record_bytes(Q, k) bytes per record, 4-byte aligned); every list ascending in (distance, index) with unused
 * slots (index -1) at its end, which is what xmh_hamming_topk writes.  dist_out / idx_out [Q][k] i32: the k smallest (distance,
 * index) pairs over all shards; slots past the total number of rows carry distance 0xFFFF and index -1. */
size_t xmh_topk_record_bytes(int64_t Q, int k);
int xmh_topk_merge_host(const void* gathered_host, int world, int64_t Q, int k, int32_t* dist_out_host, int32_t* idx_out_host);

/* ---------------------------------------------------------------------------------------------
 * Encoder primitives (a-9 .. a-12).  fp32 activations, token-major [B, L, D].  The Python model classes
 * (xmh/models) chain them exactly like models/CLIP/model.py chains torch ops.
 * ------------------------------------------------------------------------------------------- */
#define XMH_ACT_NONE 0
#define XMH_ACT_QUICKGELU 1   /* x * sigmoid(1.702 x), models/CLIP/model.py:162-164 */
#define XMH_ACT_GELU_ERF 2    /* nn.GELU() of the MITH ResidualMLPs, models/MITH/hash/hash.py:22 */
#define XMH_ACT_TANH 3
#define XMH_ACT_RELU 4
#define XMH_PREC_F32 0        /* v_mfma_f32_32x32x2_f32: exact fp32 products */
#define XMH_PREC_F16 1        /* operands rounded to fp16, fp32 accumulate ("fast mode")     */

/* C[M,N] = act(A[M,K] . W[N,K]^T + bias[N]) (+ residual[M,N]).  W in nn.Linear layout.  bias / residual may
 * be NULL.  Replaces every nn.Linear / MultiheadAttention projection / x @ proj of the reference's encoder. */
int xmh_gemm_nt_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                    const float* residual, int64_t ldr, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                    int act, int precision, xmh_stream_t stream);
/* Fast mode proper: the same contraction with A and W already stored as IEEE fp16 (K % 32 == 0, 16-byte aligned rows);
 * fp32 accumulate, fp32 bias/residual/output.  xmh_cast_f32_to_f16 is the HBM-bound conversion pass (n % 8 == 0). */
int xmh_gemm_nt_h16(const void* A_half, int64_t lda, const void* W_half, int64_t ldw, const float* bias,
                    const float* residual, int64_t ldr, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                    int act, xmh_stream_t stream);
int xmh_cast_f32_to_f16(const float* x, void* y_half, int64_t n, xmh_stream_t stream);
/* Parity-grade contraction at fp16-MFMA rate: fp32 A is split into two fp16 planes (a = hi + lo, 22 mantissa bits) by a pass
 * into stream-ordered scratch (hipMallocAsync) and multiplied by the LDS-DMA staged plane kernel; every fp16 x fp16 product is
 * exact in fp32, accumulation is fp32.  (The whole-tower entry points below never take this route: there the producing kernel
 * writes the planes.)
 *   W_lo_half == NULL: W_half must hold fp16-EXACT weights (true for CLIP weights after convert_weights,
 *                      models/CLIP/model.py:415-436): two MFMAs per product, relative error 2^-22;
 *   W_lo_half != NULL: any fp32 weight, given as w = W_half + W_lo_half (host: hi = half(w), lo = half(w - hi)): three MFMAs
 *                      per product (the lo*lo term, 2^-22 relative, is dropped).
 * Requires |a| < 65504 (larger values saturate).  Same shape rules as xmh_gemm_nt_h16 (16-byte aligned rows, K % 32 == 0). */
int xmh_gemm_nt_split16(const float* A, int64_t lda, const void* W_half, const void* W_lo_half, int64_t ldw, const float* bias,
                        const float* residual, int64_t ldr, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                        int act, xmh_stream_t stream);
/* models/CLIP/model.py:153-159 (fp32 LayerNorm) */
int xmh_layernorm_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, float* y,
                      int64_t ldy, int64_t rows, int D, xmh_stream_t stream);
/* nn.MultiheadAttention core inside ResidualAttentionBlock (models/CLIP/model.py:167-197):
 * qkv [B, L, 3*H*dh] -> out [B, L, H*dh]; causal != 0 applies build_attention_mask (:358-364);
 * key_padding_mask [B, L] bytes (non-zero = ignore key) or NULL. */
int xmh_attention_f32(const float* qkv, int64_t B, int L, int H, int dh, int causal, const uint8_t* key_padding_mask,
                      float* out, xmh_stream_t stream);
/* The same attention with the two products on the fp16 MFMA and hi/lo split operands (x = hi + lo, 22 mantissa bits; the scheme of
 * xmh_gemm_nt_split16): product error 2^-22, 5x less matrix time.  What parity and fast mode use; xmh_attention_f32 (exact fp32
 * products) is exact mode.  L > 64 runs the same fp32 kernel as xmh_attention_f32. */
int xmh_attention_split16(const float* qkv, int64_t B, int L, int H, int dh, int causal, const uint8_t* key_padding_mask,
                          float* out, xmh_stream_t stream);
/* Evaluation image transform (reference dataset/transformer_dataset.py:38-42: torchvision Resize((r, r), BICUBIC) +
 * ToTensor + Normalize on a PIL image), bit-exact with Pillow's two-pass 8-bit resample.  images [B][H][W][3] u8 RGB.
 * bounds_x [out][2] = (first input index, count), kk_x [out][ksize_x] = 22-bit fixed-point coefficients (device; built
 * once per image size by the host).  The horizontal pass (W != out_w) writes tmp [B][H][out_w][3] u8 (caller-owned).
 * Outputs (either may be NULL): resized_u8 [B][out_h][out_w][3]; out_chw [B][3][out_h][out_w] f32 =
 * ((u8 / 255) - mean) / std with mean3_host / std3_host three HOST floats each. */
int xmh_image_preprocess_u8(const uint8_t* images, int64_t B, int H, int W, int out_h, int out_w,
                            const int32_t* bounds_w, const int32_t* kk_w, int ksize_w,
                            const int32_t* bounds_h, const int32_t* kk_h, int ksize_h,
                            const float* mean3_host, const float* std3_host, uint8_t* tmp, uint8_t* resized_u8,
                            float* out_chw, xmh_stream_t stream);
/* VisionTransformer.conv1 input gather (models/CLIP/model.py:219,235): image [B,C,res,res] -> cols [B*G*G, C*P*P] */
int xmh_im2col_patch(const float* image, int64_t B, int channels, int resolution, int patch, float* cols,
                     xmh_stream_t stream);
/* class token + positional embedding + ln_pre (models/CLIP/model.py:237-243) -> x [B, n_patches+1, D] */
int xmh_vit_assemble(const float* patch_out, const float* cls, const float* pos, const float* gamma, const float* beta,
                     float eps, float* x, int64_t B, int n_patches, int D, xmh_stream_t stream);
/* token + positional embedding and EOS position = argmax(ids) (models/CLIP/model.py:374-379) */
int xmh_text_embed(const int64_t* ids, const float* tok_emb, const float* pos, float* x, int32_t* eos_index, int64_t B,
                   int L, int D, int vocab, xmh_stream_t stream);
/* out[r] = x[r*group + (idx ? idx[r] : offset)]   (cls row: offset 0, group L; EOS row: idx) */
int xmh_gather_rows(const float* x, int64_t ldx, const int32_t* idx, int offset, int group, float* out, int64_t rows,
                    int D, xmh_stream_t stream);
/* eval BatchNorm1d of the DCMHT image head (models/DCMHT/hash/hash.py:22,40) */
int xmh_affine_cols(const float* x, const float* mean, const float* var, const float* gamma, const float* beta,
                    float eps, float* y, int64_t rows, int D, xmh_stream_t stream);
/* softmax over each consecutive pair: softmax_hash (models/common/hash.py:21-31); x, y [rows, 2K] */
int xmh_pair_softmax(const float* x, float* y, int64_t rows, int K, xmh_stream_t stream);

/* MITH LocalizedTokenAggregation (models/MITH/hash/hash.py:109-169) + sin/cos positional encoding (:41-65):
 * scores [B,L,K] (tanh concept scores of every token), tokens [B,L,D] (raw CLIP tokens), token_mask [B,L] bytes
 * (non-zero = padded / EOS token, may be NULL), pos_enc [K,D] or NULL -> out [B,K,D]. */
int xmh_lta_aggregate(const float* scores, const float* tokens, const uint8_t* token_mask, const float* pos_enc,
                      float* out, int64_t B, int L, int K, int D, int top_k, xmh_stream_t stream);
/* MITH BitwiseHashing (models/MITH/hash/hash.py:68-85): out[b,k] = tanh(w[k,:] . z[b,k,:] + bias[k]) (+ addend[b,k],
 * the cls_hash + tokens_hash sum of runners/MITH/runner.py:129-130 when given). */
int xmh_bitwise_hash(const float* z, const float* w, const float* bias, const float* addend, float* out, int64_t B,
                     int K, int D, xmh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Whole-tower entry points: the CLIP ViT / text forward of models/CLIP/model.py as ONE call each (the kernel chain of
 * the primitives above, enqueued from C++: ~150 launches per tower without a host round trip per primitive).
 * All pointers are device pointers except the descriptor structs themselves and `blocks`, which live on the host.
 * ------------------------------------------------------------------------------------------- */
/* nn.Linear / MHA projection / `x @ proj` (W [N, K], row-major).  The three precisions of xmh_gemm_nt_* read different
 * members: exact (2) w_f32; parity (0) w_hi (+ w_lo for weights that are not fp16-exact), else w_f32 when w_hi is NULL or the
 * shape is unaligned; fast (1) w_hi. */
typedef struct xmh_linear {
    const float* w_f32;
    const void* w_hi;      /* IEEE fp16 [N, K] = half(W), or NULL */
    const void* w_lo;      /* IEEE fp16 [N, K] = half(W - w_hi) when W is not fp16-exact, else NULL */
    const float* bias;     /* [N] or NULL */
    int64_t n, k;
} xmh_linear;
/* ResidualAttentionBlock (models/CLIP/model.py:167-197) */
typedef struct xmh_clip_block {
    const float *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    xmh_linear qkv, out, fc, proj;
} xmh_clip_block;
/* VisionTransformer (models/CLIP/model.py:206-268) */
typedef struct xmh_vit_weights {
    int resolution, patch, width, heads, layers, out_dim;
    xmh_linear conv1;                                  /* [width, 3*patch*patch], no bias */
    const float *cls, *pos, *ln_pre_w, *ln_pre_b, *ln_post_w, *ln_post_b;
    xmh_linear proj;                                   /* visual.proj TRANSPOSED: [out_dim, width] */
    const xmh_clip_block* blocks;                      /* host array [layers] */
} xmh_vit_weights;
/* text tower of CLIP (models/CLIP/model.py:373-396) */
typedef struct xmh_text_weights {
    int vocab, context, width, heads, layers, out_dim;
    const float *tok_emb, *pos, *ln_final_w, *ln_final_b;
    xmh_linear proj;                                   /* text_projection TRANSPOSED: [out_dim, width] */
    const xmh_clip_block* blocks;
} xmh_text_weights;
/* bytes of caller-owned scratch the calls below need for B items of L tokens at `width` (conv_k = 3*patch*patch for the image
 * tower, 0 for text; out_dim > 0 only when all tokens are projected) */
size_t xmh_clip_workspace_bytes(int64_t B, int L, int width, int conv_k, int out_dim, int precision);
/* Transformer.forward (models/CLIP/model.py:200-211) on token-major x [B, L, width], in place. */
int xmh_clip_blocks_forward(const xmh_clip_block* blocks, int layers, int width, int heads, float* x, int64_t B, int L,
                            int causal, const uint8_t* key_padding_mask, int precision, void* workspace,
                            size_t workspace_bytes, xmh_stream_t stream);
/* The same forward with every activation a backward pass of ResidualAttentionBlock (models/CLIP/model.py:167-197) reads kept per
 * layer instead of in the shared scratch ("forward kernels with saved activations"; no backward is built here).
 * `saved` holds `layers` records of 16 * B*L*width floats (xmh_clip_saved_bytes), each record the row-major fp32 fields
 *     x_in [M, D] | ln1 [M, D] | qkv [M, 3D] | attn [M, D] | x_mid [M, D] | ln2 [M, D] | fc_pre [M, 4D] | fc_act [M, 4D]      (M = B*L, D = width)
 * x_in = the residual stream entering the block, ln1 = ln_1(x_in), qkv = in_proj(ln1), attn = the heads' outputs before out_proj,
 * x_mid = x_in + out_proj(attn), ln2 = ln_2(x_mid), fc_pre = c_fc(ln2), fc_act = QuickGELU(fc_pre); the block's output is the
 * next record's x_in, the last block's output is x (in place, bit-identical to xmh_clip_blocks_forward in every precision). */
size_t xmh_clip_saved_bytes(int64_t B, int L, int width, int layers);
int xmh_clip_blocks_forward_saved(const xmh_clip_block* blocks, int layers, int width, int heads, float* x, int64_t B, int L,
                                  int causal, const uint8_t* key_padding_mask, int precision, void* workspace,
                                  size_t workspace_bytes, float* saved, size_t saved_bytes, xmh_stream_t stream);
/* VisionTransformer.forward: image [B, 3, r, r] f32 -> out_cls [B, out_dim]; out_tokens [B, L, out_dim] (L = patches + 1,
 * every token through ln_post and proj: the return_patches mode, row 0 of each item = the cls feature) or NULL.
 * Exactly one of out_cls / out_tokens may be NULL. */
int xmh_vit_b32_forward(const xmh_vit_weights* w, const float* image, int64_t B, int precision, float* out_cls,
                        float* out_tokens, void* workspace, size_t workspace_bytes, xmh_stream_t stream);
/* CLIP.encode_text: ids [B, L] i64 (+ key_padding_mask [B, L] bytes or NULL) -> out_eos [B, out_dim] (feature of the EOS
 * token, argmax(ids)); out_tokens [B, L, out_dim] or NULL; eos_index [B] i32 or NULL (the EOS positions). */
int xmh_text_forward(const xmh_text_weights* w, const int64_t* ids, const uint8_t* key_padding_mask, int64_t B, int L,
                     int precision, float* out_eos, float* out_tokens, int32_t* eos_index, void* workspace,
                     size_t workspace_bytes, xmh_stream_t stream);
/* The same tower when only out_eos is wanted, on PACKED rows: caption b takes part with its tokens up to and including EOS only
 * (row_offsets [B + 1] i32 on the device, row_offsets[b + 1] - row_offsets[b] = argmax(ids[b]) + 1; total_rows = row_offsets[B], a
 * HOST value that sizes the launches).  Under the causal mask of CLIP.encode_text (models/CLIP/model.py:358-364, :373-396) nothing
 * behind a caption's EOS token can reach the row :392 selects, so out_eos is bit-identical to xmh_text_forward's -- at
 * sum(lengths) / (B L) of the work.  No key padding mask, no token outputs (callers that need those use xmh_text_forward); L <= 64;
 * workspace as for xmh_text_forward (xmh_clip_workspace_bytes(B, L, width, 0, 0, precision)). */
int xmh_text_forward_packed(const xmh_text_weights* w, const int64_t* ids, const int32_t* row_offsets, int64_t total_rows, int64_t B,
                            int L, int precision, float* out_eos, void* workspace, size_t workspace_bytes, xmh_stream_t stream);
/* The packed tower with the caption lengths counted ON THE DEVICE: no host value sizes anything, so the call never synchronises (the
 * image and text towers' streams do not wait on a host thread; the call can be captured in a hipGraph).  Launches are sized for B * L
 * rows and return on the rows behind the real count.
 *   key_padding_mask [B, L] bytes or NULL (MITH: models/MITH/MITH.py:59-66 -> models/CLIP/model.py:378): applied to the keys as in
 *     xmh_text_forward.  Caption b keeps its rows up to EOS, or up to the last position the mask leaves visible if that lies further
 *     back: every row a consumer may read unmasked is computed exactly as in xmh_text_forward.
 *   out_tokens [B, L, out_dim] or NULL (return_patches, models/CLIP/model.py:391-395): kept rows bit-identical to xmh_text_forward's;
 *     the dropped rows -- all of them hidden by key_padding_mask, or behind EOS when there is no mask -- are returned as ZERO, where the
 *     reference returns what attention made of padding.  A caller that reads those rows (the reference's own consumers mask them:
 *     models/MITH/hash/hash.py:142-148) must use xmh_text_forward.
 *   out_eos [B, out_dim] or NULL: bit-identical to xmh_text_forward's.
 * Parity / fast mode only (-ENOTSUP in exact mode); L <= 64; workspace = xmh_clip_workspace_bytes(B, L, width, 0, out_tokens ? out_dim : 0, precision). */
int xmh_text_forward_packed_dev(const xmh_text_weights* w, const int64_t* ids, const uint8_t* key_padding_mask, int64_t B, int L,
                                int precision, float* out_eos, float* out_tokens, void* workspace, size_t workspace_bytes,
                                xmh_stream_t stream);

/* One modality of the DCMHT head in eval mode (models/DCMHT/hash/hash.py:15-82): MultiheadAttention over a length-1
 * sequence == out_proj(v_proj(x)) (softmax over one key is 1), BatchNorm1d with running statistics (image) or LayerNorm
 * (text), fc2 + relu, softmax over each (off, on) pair. */
typedef struct xmh_dcmht_head {
    xmh_linear v_proj;                                 /* rows [2E, 3E) of atten.in_proj_weight / in_proj_bias */
    xmh_linear out_proj;                               /* atten.out_proj */
    int norm_is_batchnorm;                             /* 1: bn_mean / bn_var are used; 0: LayerNorm */
    float norm_eps;
    const float *norm_w, *norm_b, *bn_mean, *bn_var;
    xmh_linear fc2;                                    /* [2K, E] */
} xmh_dcmht_head;
size_t xmh_head_workspace_bytes(int64_t B, int E, int precision);
/* emb [B, E] -> probs [B, 2K] (what DCMHT.encode_image / encode_text return) and/or the packed K-bit code of
 * DCMHTTrainer.make_hash_code (runners/DCMHT/runner.py:82-95: bit = p_on > p_off) scattered to rows row_index[i] (NULL:
 * row i).  probs or bits may be NULL, not both. */
int xmh_head_dcmht(const xmh_dcmht_head* h, const float* emb, int64_t B, int precision, float* probs, uint32_t* bits,
                   const int64_t* row_index, void* workspace, size_t workspace_bytes, xmh_stream_t stream);
/* DSPH head in eval mode (models/DSPH/hash/hash.py:6-45): out = tanh(fc(emb)) [B, K]; and/or the sign-quantised packed
 * code of BaseTrainer.make_hash_code (runners/base.py:407-410) with its zero plane and value flags (see xmh_pack_sign). */
int xmh_head_dsph(const xmh_linear* fc, const float* emb, int64_t B, int precision, float* out, uint32_t* bits,
                  uint32_t* zero, int32_t* flags, const int64_t* row_index, void* workspace, size_t workspace_bytes,
                  xmh_stream_t stream);

/* One modality of the MITH head in eval mode (models/MITH/hash/hash.py:172-254; dataflow SURVEY 2.4):
 *   GlobalConceptLearning: y = ResidualMLPs(x) (per layer x += fc2(gelu(fc1(LN(x))))), concept scores tanh(E y), on the cls /
 *   EOS feature -> cls_hash [B, K] and on every token -> scores [B, L, K];
 *   LocalConceptTransforming: localized token aggregation of the RAW tokens by those scores (+ positional encoding)
 *   -> [B, K, D], transformer blocks over the K concept tokens, bitwise hashing -> tokens_hash [B, K]. */
typedef struct xmh_mith_mlp {
    const float *ln_w, *ln_b;
    float ln_eps;
    xmh_linear fc1, fc2;                               /* [4D, D], [D, 4D] */
} xmh_mith_mlp;
typedef struct xmh_mith_head {
    int width, k_bits, top_k, res_layers, layers, heads;
    const xmh_mith_mlp* mlps;                          /* host array [res_layers] */
    xmh_linear concept;                                /* common_concept_embedding [K, D], no bias */
    const float* pos_enc;                              /* [K, D] (already divided by sqrt(D)) */
    const xmh_clip_block* blocks;                      /* host array [layers] */
    const float *hash_w, *hash_b;                      /* the K one-row linears stacked: [K, D], [K] */
} xmh_mith_head;
size_t xmh_head_mith_workspace_bytes(int64_t B, int L, int width, int k_bits, int precision);
/* cls [B, D], tokens [B, L, D] (contiguous), token_mask [B, L] bytes (non-zero = ignore) or NULL
 * -> cls_hash [B, K], tokens_hash [B, K] (what MITH.encode_image / encode_text return as elements 1 and 2). */
int xmh_head_mith(const xmh_mith_head* h, const float* cls, const float* tokens, const uint8_t* token_mask, int64_t B, int L,
                  int precision, float* cls_hash, float* tokens_hash, void* workspace, size_t workspace_bytes,
                  xmh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Float similarities of common/calc_utils.py on un-quantised inputs (a-3, a-4).
 * ------------------------------------------------------------------------------------------- */
/* y = x / ||x|| row-wise (cosine_similarity :38-49, no eps) and/or sqnorm[r] = ||x_r||^2; y or sqnorm may be NULL */
int xmh_row_l2normalize(const float* x, int64_t rows, int D, float* y, float* sqnorm, xmh_stream_t stream);
/* in place: gram[i][j] = sqrt(max(|a_i|^2 + |b_j|^2 - 2 gram[i][j], 0))   (euclidean_similarity :28-36) */
int xmh_pairwise_l2_from_gram(float* gram_inout, const float* sqnorm_a, const float* sqnorm_b, int64_t M, int64_t N,
                              xmh_stream_t stream);
/* in place: x = alpha * x + beta   (0.5 * (K - q.r) on float codes, calc_hammingDist :51-56) */
int xmh_affine_inplace(float* x, int64_t n, float alpha, float beta, xmh_stream_t stream);
/* calc_map_k on FLOAT "codes" (values outside {-1, 0, +1}: UMoED-style raw tanh outputs, reference runners/UMoED/runner.py:162-186) the
 * way the reference computes it (common/calc_utils.py:72-89): distances 0.5 * (K - qB . rB^T) by exact-fp32 GEMM (:51-56), ONE stable
 * sort per query (:76-77; a segmented LSD radix sort, ties in gallery-index order = torch.sort(stable=True)), the first min(n_rel, k)
 * relevant ranks (:81-89).  SURVEY 8(b)'s xmh_gemm_f32_sort_map.  qB [Q][K], rB [R][K] fp32 row-major, packed label masks as
 * xmh_pack_labels writes them; k <= 0: mAP@all.  Outputs like xmh_hamming_ap (ap_sum[Q] f64, cap[Q] i32) and, when map_out is not
 * NULL, the mean over the queries like xmh_map_finalize.  Also the route of every code set the bit-packed kernels have no instance for
 * (ternary codes above 256 bits, more than 2048 bits, more than 256 classes): the drop-in never refuses what the reference evaluates.
 * Workspace: xmh_gemm_f32_sort_ws_bytes(Q, R) (a query tile of at most 1.5 GB: 20 bytes per pair); any size from one row
 * (20 R + 1280 bytes) up is accepted and sets the tile.  R < 2^31. */
size_t xmh_gemm_f32_sort_ws_bytes(int64_t Q, int64_t R);
int xmh_gemm_f32_sort_map(const float* qB, const float* rB, const uint32_t* qlab, const uint32_t* rlab, int64_t Q, int64_t R,
                          int K, int C, int64_t k, void* ws, size_t ws_bytes, double* ap_sum, int32_t* cap, double* map_out,
                          xmh_stream_t stream);
/* The ranking half alone, for a distance matrix the caller holds: dist[Q][R] any floats, same order, same outputs.
 * Workspace: 16 Q R + 1280 bytes. */
int xmh_float_sort_ap(const float* dist, const uint32_t* qlab, const uint32_t* rlab, int64_t Q, int64_t R, int C, int64_t k,
                      void* ws, size_t ws_bytes, double* ap_sum, int32_t* cap, xmh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Loss of the DCMHT objective: forward, and its gradient with respect to the two code matrices (what
 * loss.backward() of runners/DCMHT/runner.py:124 hands to the hash heads).  The backward of the encoders is out of scope.
 * ------------------------------------------------------------------------------------------- */
/* similarity_loss (models/DCMHT/DCMHT.py:72-98) of one pair of code matrices a, b [B, D] with packed multi-hot labels lab
 * [B][ceil(C/32)] (label_sim = calc_label_sim(labels, labels), common/calc_utils.py:8-10).  cosine == 0: euclidean branch with
 * max_value = sqrt(2 K vartheta) (:81-88); cosine != 0: the cosine branch with `threshold` (:92-95).
 * out2 (device, 2 doubles) = (positive_loss, negative_loss). */
int xmh_pair_similarity_loss(const float* a, const float* b, int64_t B, int D, const uint32_t* lab, int C, int cosine,
                             float max_value, float threshold, double* out2, xmh_stream_t stream);
/* soft_argmax_hash_loss (models/DCMHT/DCMHT.py:100-105): out (device, 1 double) = 1 - mean((2 code - 1)^2) over n elements */
int xmh_quant_loss(const float* code, int64_t n, double* out, xmh_stream_t stream);
/* d(positive_loss + negative_loss)/da of xmh_pair_similarity_loss, as autograd derives it from models/DCMHT/DCMHT.py:72-98
 * (torch.cdist's backward: zero distances contribute nothing; clip passes the gradient on the closed interval).  grad_a [B, D]
 * (device) is written, or added to when accumulate != 0, with scale * upstream[0] (upstream: device float, NULL = 1) folded in.
 * The gradient with respect to b is the same call with a and b exchanged; for a term with a == b pass scale = 2. */
int xmh_pair_similarity_loss_grad(const float* a, const float* b, int64_t B, int D, const uint32_t* lab, int C, int cosine,
                                  float max_value, float threshold, float scale, const float* upstream, float* grad_a,
                                  int accumulate, xmh_stream_t stream);
/* d xmh_quant_loss / d code = -4 (2 code - 1) / n, times scale * upstream[0], written or accumulated like above */
int xmh_quant_loss_grad(const float* code, int64_t n, float scale, const float* upstream, float* grad, int accumulate,
                        xmh_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* XMH_H */
