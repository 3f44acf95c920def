/*
 * xmh_oracle.c -- plain-C restatement of the integer retrieval maths.  TEST INFRASTRUCTURE ONLY.
 *
 * Used by tests/ as a fast checker at sizes where the numpy oracle (oracle/retrieval.py) is too slow.
 * It is NOT on the product path and nothing under clip-based-cross-modal-hash_amd/ links or loads it.
 *
 * What it restates (reference file:line):
 *   orc_hamming      calc_hammingDist, common/calc_utils.py:51-56, for +-1 codes: popcount(q xor r)
 *   orc_hist/orc_ap  calc_map_k, common/calc_utils.py:58-92: relevance (qL.rL^T > 0) as mask AND (:72),
 *                    ranking by (distance, gallery index) == torch.sort(stable=True) (:77), the first
 *                    min(n_rel,k) relevant ranks (:81,:86-88), sum of ordinal/rank (:89).
 *   orc_topk         exact per-query top-k under the same order (north_star retrieval mode).
 *   orc_topk_ternary the same for codes in {-1, 0, +1} (BaseTrainer.make_hash_code is sign_(), runners/base.py:407-410, sign(0) = 0):
 *                    calc_hammingDist's 0.5 * (K - q.r) (:51-56) in half units, K - q.r = #(either side 0) + 2 #(both live, different).
 * Pinned through tests/test_oracle_c.py against oracle/retrieval.py, which is pinned against the
 * golden vectors generated from the reference.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int dist_of(const uint32_t* q, const uint32_t* r, int W) {
    int d = 0;
    for (int w = 0; w < W; ++w) d += __builtin_popcount(q[w] ^ r[w]);
    return d;
}

static inline int rel_of(const uint32_t* q, const uint32_t* r, int Lw) {
    uint32_t hit = 0;
    for (int w = 0; w < Lw; ++w) hit |= q[w] & r[w];
    return hit != 0;
}

void orc_hamming(const uint32_t* qbits, const uint32_t* rbits, int64_t Q, int64_t R, int W, uint16_t* out) {
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < Q; ++q)
        for (int64_t r = 0; r < R; ++r) out[q * R + r] = (uint16_t)dist_of(qbits + q * W, rbits + r * W, W);
}

void orc_hist(const uint32_t* qbits, const uint32_t* qlab, const uint32_t* rbits, const uint32_t* rlab, int64_t Q,
              int64_t R, int W, int Lw, int nb, uint32_t* hist_all, uint32_t* hist_rel) {
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < Q; ++q) {
        uint32_t* ha = hist_all + q * nb;
        uint32_t* hr = hist_rel + q * nb;
        memset(ha, 0, sizeof(uint32_t) * nb);
        memset(hr, 0, sizeof(uint32_t) * nb);
        for (int64_t r = 0; r < R; ++r) {
            const int d = dist_of(qbits + q * W, rbits + r * W, W);
            ha[d]++;
            hr[d] += rel_of(qlab + q * Lw, rlab + r * Lw, Lw);
        }
    }
}

/* base_all/base_rel/nrel_total may be NULL (single shard).  ap_sum[q] = sum ordinal/rank, cap[q] = min(n_rel,k). */
void orc_ap(const uint32_t* qbits, const uint32_t* qlab, const uint32_t* rbits, const uint32_t* rlab, int64_t Q, int64_t R,
            int W, int Lw, int nb, const uint32_t* base_all, const uint32_t* base_rel, const uint32_t* nrel_total,
            int64_t k, double* ap_sum, int32_t* cap) {
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < Q; ++q) {
        uint64_t* ca = (uint64_t*)calloc(nb, sizeof(uint64_t));
        uint64_t* cr = (uint64_t*)calloc(nb, sizeof(uint64_t));
        uint64_t nrel = 0;
        if (base_all) {
            for (int d = 0; d < nb; ++d) {
                ca[d] = base_all[q * nb + d];
                cr[d] = base_rel[q * nb + d];
            }
            nrel = nrel_total[q];
        } else {
            for (int64_t r = 0; r < R; ++r) {
                const int d = dist_of(qbits + q * W, rbits + r * W, W);
                ca[d]++;
                cr[d] += rel_of(qlab + q * Lw, rlab + r * Lw, Lw);
            }
            uint64_t sa = 0, sr = 0;
            for (int d = 0; d < nb; ++d) {
                const uint64_t a = ca[d], b = cr[d];
                ca[d] = sa;
                cr[d] = sr;
                sa += a;
                sr += b;
            }
            nrel = sr;
        }
        const uint64_t c = (k > 0 && (uint64_t)k < nrel) ? (uint64_t)k : nrel;
        double s = 0.0;
        for (int64_t r = 0; r < R; ++r) {
            const int d = dist_of(qbits + q * W, rbits + r * W, W);
            const int rel = rel_of(qlab + q * Lw, rlab + r * Lw, Lw);
            const uint64_t rank = ++ca[d];
            if (rel) {
                const uint64_t ord = ++cr[d];
                if (ord <= c) s += (double)ord / (double)rank;
            }
        }
        ap_sum[q] = s;
        cap[q] = (int32_t)c;
        free(ca);
        free(cr);
    }
}

/* exact top-k by (distance, index); out arrays [Q][k]; if R < k the tail is filled with dist 0xFFFF / idx -1 */
void orc_topk(const uint32_t* qbits, const uint32_t* rbits, int64_t Q, int64_t R, int W, int nb, int k, int64_t base_index,
              uint16_t* dist, int32_t* idx) {
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < Q; ++q) {
        uint32_t* hist = (uint32_t*)calloc(nb + 1, sizeof(uint32_t));
        for (int64_t r = 0; r < R; ++r) hist[dist_of(qbits + q * W, rbits + r * W, W)]++;
        /* write position of each bucket inside the first k slots */
        uint32_t* start = (uint32_t*)calloc(nb + 1, sizeof(uint32_t));
        uint32_t run = 0;
        for (int d = 0; d < nb; ++d) {
            start[d] = run;
            run += hist[d];
        }
        for (int i = 0; i < k; ++i) {
            dist[q * k + i] = 0xFFFF;
            idx[q * k + i] = -1;
        }
        for (int64_t r = 0; r < R; ++r) {
            const int d = dist_of(qbits + q * W, rbits + r * W, W);
            const uint32_t pos = start[d]++;
            if (pos < (uint32_t)k) {
                dist[q * k + pos] = (uint16_t)d;
                idx[q * k + pos] = (int32_t)(base_index + r);
            }
        }
        free(hist);
        free(start);
    }
}

/* ternary codes: bits (1 <=> element > 0) and zero planes (1 <=> element == 0; padding bits of the last word set on both sides).
 * dist2 = K - q.r in 0 ... 2K (half units of the reference's distance); order (dist2, index). */
static inline int dist2_of(const uint32_t* qb, const uint32_t* qz, const uint32_t* rb, const uint32_t* rz, int W, int K) {
    int dead = 0, diff = 0;
    for (int w = 0; w < W; ++w) {
        const uint32_t z = qz[w] | rz[w];
        dead += __builtin_popcount(z);
        diff += __builtin_popcount((qb[w] ^ rb[w]) & ~z);
    }
    return dead - (32 * W - K) + 2 * diff;
}

void orc_topk_ternary(const uint32_t* qbits, const uint32_t* qzero, const uint32_t* rbits, const uint32_t* rzero, int64_t Q, int64_t R, int W,
                      int K, int k, int64_t base_index, uint16_t* dist2, int32_t* idx) {
    const int nb = 2 * K + 1;
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < Q; ++q) {
        uint32_t* start = (uint32_t*)calloc(nb + 1, sizeof(uint32_t));
        for (int64_t r = 0; r < R; ++r) start[dist2_of(qbits + q * W, qzero + q * W, rbits + r * W, rzero + r * W, W, K) + 1]++;
        for (int d = 0; d < nb; ++d) start[d + 1] += start[d];
        for (int i = 0; i < k; ++i) {
            dist2[q * k + i] = 0xFFFF;
            idx[q * k + i] = -1;
        }
        for (int64_t r = 0; r < R; ++r) {
            const int d = dist2_of(qbits + q * W, qzero + q * W, rbits + r * W, rzero + r * W, W, K);
            const uint32_t pos = start[d]++;
            if (pos < (uint32_t)k) {
                dist2[q * k + pos] = (uint16_t)d;
                idx[q * k + pos] = (int32_t)(base_index + r);
            }
        }
        free(start);
    }
}
