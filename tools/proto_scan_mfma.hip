// Prototype of the MFMA-evaluated ranking scan (pass 1 = bucket histograms), standalone: validates against a CPU count on a
// small ragged case and times the configs[1] shape.  hipcc --offload-arch=gfx950 -O3 tools/proto_scan_mfma.hip -o tools/proto_scan_mfma.bin
//
// Idea: Hamming distance and label overlap of 16 gallery items x 16 queries are two i8 dot-product tiles -- exactly what the
// reference computes (B1 @ B2^T, query_L @ retrieval_L^T, common/calc_utils.py:51-56,72) -- so they go to
// v_mfma_i32_16x16x64_i8.  Scaling the query operand by the counter stride and starting the accumulator at the lane's
// counter base makes the MFMA emit the LDS BYTE ADDRESS of the (bucket, query) counter directly; the label tile is scaled so
// that min(acc, 1 + S) is the add operand (1 | relevant * S).  Per pair the VALU does ONE instruction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) uint32_t lds_u32;
constexpr int kRelScale = 127 * 64;        // label tile: item byte 127 x query byte 64

// gallery image: [batch of 64 items][group g of 16][m][lane][16 B]; lane = c*16 + rho supplies A row rho, k-chunk c of MFMA m;
// row rho of a group is item 4*(rho&3) + (rho>>2): the 4 accumulator registers of a lane are then 4 consecutive "steps"
template <int NMC, int NML>
__global__ void k_expand_gallery(const uint32_t* __restrict__ rbits, const uint32_t* __restrict__ rlab, int R, int W, int LW, int K,
                                 uint4* __restrict__ out, int64_t npieces) {
    constexpr int NM = NMC + NML;
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npieces) return;
    const int lane = p & 63;
    const int m = (p >> 6) % NM;
    const int64_t grp = (p >> 6) / NM;                 // batch * 4 + g
    const int rho = lane & 15, c = lane >> 4;
    const int64_t item = grp * 16 + 4 * (rho & 3) + (rho >> 2);
    uint32_t by[4] = {0, 0, 0, 0};
    if (m < NMC) {
        const int k0 = m * 64 + c * 16;
        uint32_t bits = 0;
        if (item < R && k0 < W * 32) bits = (rbits[item * W + (k0 >> 5)] >> (k0 & 31)) & 0xffffu;
        for (int t = 0; t < 16; ++t) {
            const uint32_t b = (k0 + t < K) ? (((bits >> t) & 1u) ? 0x01u : 0xffu) : 0u;
            by[t >> 2] |= b << (8 * (t & 3));
        }
    } else {
        const int k0 = (m - NMC) * 64 + c * 16;
        uint32_t bits = 0;
        if (item < R && k0 < LW * 32) bits = (rlab[item * LW + (k0 >> 5)] >> (k0 & 31)) & 0xffffu;
        for (int t = 0; t < 16; ++t) by[t >> 2] |= (((bits >> t) & 1u) ? 127u : 0u) << (8 * (t & 3));
    }
    out[p] = make_uint4(by[0], by[1], by[2], by[3]);
}

// query image: [tile of 16 queries][m][lane][16 B]; code bytes -/+ scale (so that acc = base - scale * dot), label bytes 64
template <int NMC, int NML>
__global__ void k_expand_queries(const uint32_t* __restrict__ qbits, const uint32_t* __restrict__ qlab, int Q, int W, int LW, int K, int scale,
                                 uint4* __restrict__ out, int64_t npieces) {
    constexpr int NM = NMC + NML;
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npieces) return;
    const int lane = p & 63;
    const int m = (p >> 6) % NM;
    const int64_t tile = (p >> 6) / NM;
    const int ql = lane & 15, c = lane >> 4;
    const int64_t q = tile * 16 + ql;
    uint32_t by[4] = {0, 0, 0, 0};
    if (q < Q) {
        if (m < NMC) {
            const int k0 = m * 64 + c * 16;
            uint32_t bits = 0;
            if (k0 < W * 32) bits = (qbits[q * W + (k0 >> 5)] >> (k0 & 31)) & 0xffffu;
            for (int t = 0; t < 16; ++t) {
                const uint32_t b = (k0 + t < K) ? (((bits >> t) & 1u) ? (uint32_t)(-scale) & 0xffu : (uint32_t)scale) : 0u;
                by[t >> 2] |= b << (8 * (t & 3));
            }
        } else {
            const int k0 = (m - NMC) * 64 + c * 16;
            uint32_t bits = 0;
            if (k0 < LW * 32) bits = (qlab[q * LW + (k0 >> 5)] >> (k0 & 31)) & 0xffffu;
            for (int t = 0; t < 16; ++t) by[t >> 2] |= (((bits >> t) & 1u) ? 64u : 0u) << (8 * (t & 3));
        }
    }
    out[p] = make_uint4(by[0], by[1], by[2], by[3]);
}

struct Args {
    const uint4* gimg;
    const uint4* qimg;
    const uint32_t* qbits;
    int Q, R, K, W;
    int chunk, nchunk, nqt, nb, qpad;
};

template <int NMC, int NML, int NW, int SUB, int ABL>
__global__ __launch_bounds__(64 * NW) void k_hist_mfma(Args a, uint32_t* __restrict__ chunk_hist) {
    constexpr int NM = NMC + NML;
    constexpr int PIECES = 4 * NM;                 // 1 KB pieces per 64-item batch
    constexpr int PPW = PIECES / NW;
    static_assert(PIECES % NW == 0, "pieces per wave");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int b = blockIdx.x;
    const int xcd = b & 7, t = b >> 3;
    const int qtile = t % a.nqt;
    const int chunk_id = xcd + 8 * (t / a.nqt);
    if (chunk_id >= a.nchunk) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ql = lane & 15, slot = lane >> 4;
    const int q0 = (qtile * NW + wave) * 16;
    const int q = q0 + ql;
    const int ncell = a.nb * 16 * SUB, cstride = ncell;          // dwords per wave
    uint32_t* cnt = lds + wave * cstride;
    for (int e = lane; e < ncell; e += 64) cnt[e] = 0u;
    char* ring = reinterpret_cast<char*>(lds + NW * cstride);   // 2 buffers of PIECES KB
    v4i bq[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) bq[m] = *reinterpret_cast<const v4i*>(a.qimg + ((int64_t)(q0 >> 4) * NM + m) * 64 + lane);
    const bool valid = q < a.Q;
    const int cinit = (int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)cnt + ql * 4 + (SUB == 2 ? (slot & 1) * 64 : 0) + (valid ? 32 * SUB * a.K : 0);
    const int64_t lo = (int64_t)chunk_id * a.chunk;
    const int64_t hi = (lo + a.chunk < a.R) ? lo + a.chunk : a.R;
    const int nbat = (int)((hi - lo + 63) >> 6);
    const int64_t bat0 = lo >> 6;
    auto stage = [&](int buf, int64_t batch) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int p = j * NW + wave;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.gimg + (batch * PIECES + p) * 64 + lane),
                                             (__attribute__((address_space(3))) void*)(ring + buf * (PIECES * 1024) + p * 1024), 16, 0, 0);
        }
    };
    stage(0, bat0);
    for (int i = 0; i < nbat; ++i) {
        if (ABL == 2) {
        } else if (i + 1 < nbat) {
            stage((i + 1) & 1, bat0 + i + 1);
            if (PPW == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (PPW == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (ABL != 2) __builtin_amdgcn_s_barrier();
        const char* base = ring + (i & 1) * (PIECES * 1024) + lane * 16;
        v4i am[4][NM];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int m = 0; m < NM; ++m) am[g][m] = *reinterpret_cast<const v4i*>(base + (g * NM + m) * 1024);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            v4i acc = {cinit, cinit, cinit, cinit};
#pragma unroll
            for (int m = 0; m < NMC; ++m) {
                if (ABL == 3) { acc[0] += (am[g][m][0] & 0xfc0); acc[1] += (am[g][m][1] & 0xfc0); acc[2] += (am[g][m][2] & 0xfc0); acc[3] += (am[g][m][3] & 0xfc0); }
                else acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(am[g][m], bq[m], acc, 0, 0, 0);
            }
            v4i lab = {1, 1, 1, 1};
#pragma unroll
            for (int m = NMC; m < NM; ++m) {
                if (ABL == 3) lab += am[g][m];
                else lab = __builtin_amdgcn_mfma_i32_16x16x64_i8(am[g][m], bq[m], lab, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t inc = min((uint32_t)lab[j], (uint32_t)(1 + kRelScale));
                if (ABL == 1) asm volatile("" :: "v"(acc[j]), "v"(inc));
                else asm volatile("ds_add_u32 %0, %1" :: "v"(acc[j]), "v"(inc) : "memory");   // hidden from hipcc: it would drain the LDS-DMA (vmcnt(0)) first
            }
        }
        if (ABL != 2) __builtin_amdgcn_s_barrier();
    }
    // padding items of the ragged last batch are all-zero-bit codes without labels: distance popcount(query), never relevant
    const int npad = nbat * 64 - (int)(hi - lo);
    if (npad > 0 && slot == 0 && valid) {
        int dpad = 0;
        for (int w = 0; w < a.W; ++w) dpad += __popc(a.qbits[(int64_t)q * a.W + w]);
        cnt[dpad * 16 * SUB + ql] -= (uint32_t)npad;
    }
    uint32_t* __restrict__ out = chunk_hist + ((int64_t)chunk_id * a.nb) * a.qpad + q0;
    __builtin_amdgcn_s_waitcnt(0);
    for (int e = lane; e < a.nb * 16; e += 64) {
        uint32_t v = cnt[(e >> 4) * 16 * SUB + (e & 15)];
        if (SUB == 2) v += cnt[(e >> 4) * 32 + 16 + (e & 15)];
        out[(int64_t)(e >> 4) * a.qpad + (e & 15)] = v;
    }
}

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state; }

template <int SUB, int ABL = 0>
int run_case(int Q, int R, int K, int C, int nchunk_req, bool check, int iters) {
    constexpr int NMC = 1, NML = 2, NW = 4, NM = NMC + NML;
    const int W = (K + 31) / 32, LW = (C + 31) / 32, nb = K + 1;
    std::vector<uint32_t> qb((size_t)Q * W), rb((size_t)R * W), qlv((size_t)Q * LW), rlv((size_t)R * LW);
    for (auto& x : qb) x = rnd() ^ (rnd() << 16);
    for (auto& x : rb) x = rnd() ^ (rnd() << 16);
    auto lab = [&](std::vector<uint32_t>& v, int n) {
        for (int i = 0; i < n; ++i)
            for (int w = 0; w < LW; ++w) {
                uint32_t m = rnd() & rnd() & rnd() & rnd() & (rnd() | rnd());      // sparse
                const int bits = C - 32 * w >= 32 ? 32 : C - 32 * w;
                if (bits < 32) m &= (1u << bits) - 1u;
                v[(size_t)i * LW + w] = m;
            }
    };
    lab(qlv, Q); lab(rlv, R);
    if (K % 32) for (int i = 0; i < Q; ++i) qb[(size_t)i * W + W - 1] &= (1u << (K % 32)) - 1u;
    if (K % 32) for (int i = 0; i < R; ++i) rb[(size_t)i * W + W - 1] &= (1u << (K % 32)) - 1u;
    int chunk = (R + nchunk_req - 1) / nchunk_req;
    chunk = (chunk + 63) / 64 * 64;
    if (chunk > 8064) chunk = 8064;
    const int nchunk = (R + chunk - 1) / chunk;
    const int nqt = (Q + 16 * NW - 1) / (16 * NW);
    const int qpad = nqt * 16 * NW;
    const int64_t nbatch = (R + 63) / 64 + nchunk;      // slack: every chunk may end in a ragged batch
    uint32_t *d_qb, *d_rb, *d_ql, *d_rl, *d_hist;
    uint4 *d_gimg, *d_qimg;
    CHECK(hipMalloc(&d_qb, qb.size() * 4)); CHECK(hipMalloc(&d_rb, rb.size() * 4));
    CHECK(hipMalloc(&d_ql, qlv.size() * 4)); CHECK(hipMalloc(&d_rl, rlv.size() * 4));
    CHECK(hipMemcpy(d_qb, qb.data(), qb.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_rb, rb.data(), rb.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_ql, qlv.data(), qlv.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_rl, rlv.data(), rlv.size() * 4, hipMemcpyHostToDevice));
    // NOTE: chunks start on 64-item boundaries of the GLOBAL index (chunk % 64 == 0), so batch b = items [64 b, 64 b + 64)
    const int64_t gpieces = (int64_t)((R + 63) / 64) * 4 * NM * 64;
    const int64_t qpieces = (int64_t)(qpad / 16) * NM * 64;
    CHECK(hipMalloc(&d_gimg, (size_t)(gpieces + 64 * 4 * NM) * 16)); CHECK(hipMalloc(&d_qimg, (size_t)qpieces * 16));
    CHECK(hipMalloc(&d_hist, (size_t)nchunk * nb * qpad * 4));
    (void)nbatch;
    Args a{d_gimg, d_qimg, d_qb, Q, R, K, W, chunk, nchunk, nqt, nb, qpad};
    const size_t ldsb = (size_t)NW * nb * 16 * 4 * SUB + 2 * 4 * NM * 1024;
    const int grid = 8 * nqt * ((nchunk + 7) / 8);
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hist_mfma<NMC, NML, NW, SUB, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipEvent_t e0, e1, e2;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2));
    float ms_exp = 0, ms_hist = 0;
    for (int it = 0; it < iters; ++it) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_expand_gallery<NMC, NML>), dim3((unsigned)((gpieces + 255) / 256)), dim3(256), 0, 0, d_rb, d_rl, R, W, LW, K, d_gimg, gpieces);
        hipLaunchKernelGGL((k_expand_queries<NMC, NML>), dim3((unsigned)((qpieces + 255) / 256)), dim3(256), 0, 0, d_qb, d_ql, Q, W, LW, K, 32 * SUB, d_qimg, qpieces);
        CHECK(hipEventRecord(e1));
        hipLaunchKernelGGL((k_hist_mfma<NMC, NML, NW, SUB, ABL>), dim3(grid), dim3(64 * NW), ldsb, 0, a, d_hist);
        CHECK(hipEventRecord(e2));
        CHECK(hipEventSynchronize(e2));
        float x, y;
        CHECK(hipEventElapsedTime(&x, e0, e1)); CHECK(hipEventElapsedTime(&y, e1, e2));
        if (it == 0 || x < ms_exp) ms_exp = x;
        if (it == 0 || y < ms_hist) ms_hist = y;
    }
    CHECK(hipGetLastError());
    printf("ABL=%d SUB=%d Q=%d R=%d K=%d C=%d chunk=%d nchunk=%d grid=%d lds=%zu: expand %.3f ms, hist %.3f ms (best of %d) -> %.3e pairs/s\n", ABL, SUB, Q, R, K, C, chunk, nchunk,
           grid, ldsb, ms_exp, ms_hist, iters, (double)Q * R / (ms_hist * 1e-3));
    int bad = 0;
    if (check) {
        std::vector<uint32_t> h((size_t)nchunk * nb * qpad);
        CHECK(hipMemcpy(h.data(), d_hist, h.size() * 4, hipMemcpyDeviceToHost));
        std::vector<uint32_t> wa((size_t)nb), wr((size_t)nb);
        for (int q = 0; q < Q && bad < 10; ++q) {
            for (int c = 0; c < nchunk && bad < 10; ++c) {
                std::fill(wa.begin(), wa.end(), 0u); std::fill(wr.begin(), wr.end(), 0u);
                const int64_t lo = (int64_t)c * chunk, hi = std::min<int64_t>(lo + chunk, R);
                for (int64_t i = lo; i < hi; ++i) {
                    int d = 0; uint32_t hit = 0;
                    for (int w = 0; w < W; ++w) d += __builtin_popcount(qb[(size_t)q * W + w] ^ rb[(size_t)i * W + w]);
                    for (int w = 0; w < LW; ++w) hit |= qlv[(size_t)q * LW + w] & rlv[(size_t)i * LW + w];
                    wa[d]++; wr[d] += hit != 0;
                }
                for (int d = 0; d < nb; ++d) {
                    const uint32_t got = h[((size_t)c * nb + d) * qpad + q];
                    const uint32_t ga = got % kRelScale, gr = got / kRelScale;
                    if (ga != wa[d] || gr != wr[d]) {
                        if (bad < 10) printf("  MISMATCH q=%d chunk=%d d=%d: got (%u,%u) want (%u,%u)\n", q, c, d, ga, gr, wa[d], wr[d]);
                        ++bad;
                    }
                }
            }
        }
        printf("  check: %s\n", bad ? "FAILED" : "ok");
    }
    hipFree(d_qb); hipFree(d_rb); hipFree(d_ql); hipFree(d_rl); hipFree(d_hist); hipFree(d_gimg); hipFree(d_qimg);
    return bad;
}

int main() {
    int bad = 0;
    bad += run_case<1>(100, 5000, 64, 80, 3, true, 2);
    bad += run_case<2>(100, 5000, 64, 80, 3, true, 2);
    bad += run_case<1>(37, 1501, 64, 80, 2, true, 2);
    bad += run_case<2>(37, 1501, 64, 80, 2, true, 2);
    bad += run_case<2>(64, 64 * 40, 64, 33, 1, true, 2);
    for (int nc : {24, 32}) {
        run_case<1, 0>(5000, 117218, 64, 80, nc, false, 20);
        run_case<1, 1>(5000, 117218, 64, 80, nc, false, 20);
        run_case<1, 2>(5000, 117218, 64, 80, nc, false, 20);
        run_case<1, 3>(5000, 117218, 64, 80, nc, false, 20);
    }
    return bad != 0;
}
