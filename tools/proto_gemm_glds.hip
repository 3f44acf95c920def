// Prototype bench of the LDS-DMA staged fp16 GEMM (k_gemm_g16) before it moves into xmh_gemm.hip.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/proto_gemm_glds.hip -o /tmp/pg && /tmp/pg
// C[M,N] = sum over NA planes of A_p[M,K] . W[N,K]^T  (NT, fp16 operands, fp32 accumulate / output).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct GArgs {
    const _Float16* A;      // plane 0 (lo when NA == 2)
    const _Float16* A2;     // plane 1 (hi)
    const _Float16* W;
    float* C;
    int64_t lda, ldw, ldc;
    int M, N, K;
    unsigned long long* tim;
};

__device__ __forceinline__ void tile_of_block(int nbm, int nbn, int& tm, int& tn) {
    const int nwg = nbm * nbn;
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    tn = id / nbm;
    tm = id % nbm;
}

// LDS image of an operand tile: [rows][BK halves]; CH = BK/8 16-byte chunks per row, RPB = 16/CH rows per 256-byte bank row.
// Chunk c of row r is stored at chunk c ^ ((r / RPB) & (CH-1)): the 16 lanes of a ds_read_b128 group (16 distinct rows, one k
// chunk) then cover all 16 slots of the bank row.  global_load_lds writes lane-linear (base + lane * 16), so the permutation is
// applied to the SOURCE address: lane l of a 1 KB piece fills row l / CH, stored chunk l % CH.
template <int WM, int WN, int MI, int NJ, int NA, int BK, int NBUF, int DEPTH, int MINB, int ABL, int GM>
__global__ __launch_bounds__(64 * WM * WN, MINB) void k_gemm_g16(GArgs g) {
    constexpr int NWAVE = WM * WN;
    constexpr int TBM = 32 * MI * WM, TBN = 32 * NJ * WN;
    constexpr int CH = BK / 8, RPB = 16 / CH, RPP = 64 / CH;        // chunks per row, rows per bank row, rows per 1 KB piece
    constexpr int ROWB = BK * 2;
    constexpr int PA = TBM / RPP, PW = TBN / RPP;                   // 1 KB pieces per operand tile
    constexpr int NPIECE = NA * PA + PW;
    static_assert(NPIECE % NWAVE == 0, "pieces per wave");
    constexpr int PPW = NPIECE / NWAVE;
    constexpr int BUFB = NPIECE * 1024;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int nbm = (g.M + TBM - 1) / TBM, nbn = (g.N + TBN - 1) / TBN;
    int tm, tn;
    tile_of_block(nbm, nbn, tm, tn);
    if (GM > 0) {      // grouped order inside the XCD's contiguous id range: GM tile rows x all tile columns, column-major inside
        const int id = tn * nbm + tm;
        const int per = GM * nbn;
        const int grp = id / per, rem = id % per;
        const int gm0 = grp * GM;
        const int gsz = nbm - gm0 < GM ? nbm - gm0 : GM;
        tm = gm0 + rem % gsz;
        tn = rem / gsz;
    }
    const int m0 = tm * TBM, n0 = tn * TBN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave / WN) * 32 * MI, wn = (wave % WN) * 32 * NJ;
    const int fr = lane & 31, fh = lane >> 5;

    const _Float16* src[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int p = j * NWAVE + wave;
        const int prow = lane / CH;
        int r;
        const _Float16* base;
        if (p < NA * PA) {
            const int pl = p / PA;
            r = (p % PA) * RPP + prow;
            const int rg = m0 + r < g.M ? m0 + r : g.M - 1;
            base = (pl == 0 ? g.A : g.A2) + (int64_t)rg * g.lda;
        } else {
            r = (p - NA * PA) * RPP + prow;
            const int rg = n0 + r < g.N ? n0 + r : g.N - 1;
            base = g.W + (int64_t)rg * g.ldw;
        }
        const int chunk = (lane % CH) ^ ((r / RPB) & (CH - 1));
        src[j] = base + chunk * 8;
    }
    auto stage = [&](int buf, int k0) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int p = j * NWAVE + wave;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + k0),
                                             (__attribute__((address_space(3))) void*)(lds + buf * BUFB + p * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    if (ABL >= 11 && ABL <= 14 && ((blockIdx.x >> 8) & 1)) {
#pragma unroll
        for (int d = 0; d < ABL - 10; ++d) __builtin_amdgcn_s_sleep(127);
    }
    unsigned long long t0 = 0, t1 = 0, t2 = 0, acc_wait = 0, acc_bar = 0, acc_stage = 0;
    if (ABL == 20 || ABL == 21) t0 = __builtin_readcyclecounter();
    const int swz = (fr / RPB) & (CH - 1);      // wm, wn, i*32 are multiples of 32: the swizzle depends on fr only
    const int nk = g.K / BK;
    // DEPTH = tiles in flight beyond the one being computed (1: wait for everything each step; 2 needs NBUF >= 3)
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (d < nk) stage(d % NBUF, d * BK);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt % NBUF;
        unsigned long long ta = 0, tb = 0, tc = 0, td = 0;
        if (ABL == 21) ta = __builtin_readcyclecounter();
        if (DEPTH == 1 || kt + 1 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (PPW == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (PPW == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (PPW == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (PPW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (PPW == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (PPW == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (PPW == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (PPW == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else if (PPW == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (PPW == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ABL == 21) tb = __builtin_readcyclecounter();
        __builtin_amdgcn_s_barrier();
        if (ABL == 21) tc = __builtin_readcyclecounter();
        if (kt + DEPTH < nk && ABL != 2) stage((kt + DEPTH) % NBUF, (kt + DEPTH) * BK);
        if (ABL == 21) { td = __builtin_readcyclecounter(); acc_wait += tb - ta; acc_bar += tc - tb; acc_stage += td - tc; }
        const char* bA = lds + buf * BUFB;
        const char* bW = bA + NA * PA * 1024;
        if (ABL == 30) {      // explicit fragment double-buffering: slab s+1 is read while the MFMAs of slab s run
            constexpr int NS = BK / 16;
            f16x8 fa[2][NA][MI], fb[2][NJ];
            auto rd = [&](int s, int w) {
                const int coff = ((2 * s + fh) ^ swz) * 16;
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[w][j] = *reinterpret_cast<const f16x8*>(bW + (wn + j * 32 + fr) * ROWB + coff);
#pragma unroll
                for (int pl = 0; pl < NA; ++pl)
#pragma unroll
                    for (int i = 0; i < MI; ++i) fa[w][pl][i] = *reinterpret_cast<const f16x8*>(bA + pl * PA * 1024 + (wm + i * 32 + fr) * ROWB + coff);
            };
            rd(0, 0);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (s + 1 < NS) rd(s + 1, (s + 1) & 1);
#pragma unroll
                for (int pl = 0; pl < NA; ++pl)
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s & 1][pl][i], fb[s & 1][j], acc[i][j], 0, 0, 0);
            }
            // pin the order: reads of slab s+1 are issued before the MFMAs of slab s
            constexpr int NR = NA * MI + NJ, NM = NA * MI * NJ;
            __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (s + 1 < NS) __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
            }
        } else
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {                        // k slabs of 16: lane half fh takes chunk 2s + fh
            const int coff = ((2 * s + fh) ^ swz) * 16;
            f16x8 b[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[j] = *reinterpret_cast<const f16x8*>((ABL == 3 ? lds : bW) + (wn + j * 32 + fr) * ROWB + (ABL == 3 ? 0 : coff));
#pragma unroll
            for (int pl = 0; pl < NA; ++pl) {
                f16x8 a[MI];
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const f16x8*>((ABL == 3 ? lds : bA) + pl * PA * 1024 + (wm + i * 32 + fr) * ROWB + (ABL == 3 ? 0 : coff));
                if (ABL == 31) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if (ABL == 4) { asm volatile("" ::"v"(a[i]), "v"(b[j])); }
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
                    }
                if (ABL == 31) __builtin_amdgcn_s_setprio(0);
            }
        }
    }
    if (ABL == 20 || ABL == 21) t1 = __builtin_readcyclecounter();
    if (ABL == 21 && (tid & 63) == 0) {
        unsigned long long* o = g.tim + (blockIdx.x * 4 + wave) * 4;
        o[0] = t1 - t0; o[1] = acc_wait; o[2] = acc_bar; o[3] = acc_stage;
    }
    if (ABL == 7 || ABL == 8) {
        // transposed epilogue: the wave's (32 MI) x (32 NJ) fp32 tile goes through its own LDS region, rows come back as 16-byte
        // pieces: 16 lanes cover 256 bytes of one row (fp32) / 8 lanes cover 128 bytes (fp16)
        __builtin_amdgcn_s_barrier();
        constexpr int TW = 32 * NJ;                           // tile width in floats
        float* reg = reinterpret_cast<float*>(lds) + wave * (32 * MI * TW);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) reg[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh) * TW + j * 32 + fr] = acc[i][j][e];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (ABL == 7) {
            constexpr int LPR = TW / 4;                       // lanes per row
            constexpr int RPI = 64 / LPR;                     // rows per instruction
#pragma unroll
            for (int it = 0; it < 32 * MI / RPI; ++it) {
                const int r = it * RPI + lane / LPR, c = (lane % LPR) * 4;
                const float4 v = *reinterpret_cast<const float4*>(reg + r * TW + c);
                const int row = m0 + wm + r, col = n0 + wn + c;
                if (row < g.M && col < g.N) *reinterpret_cast<float4*>(g.C + (int64_t)row * g.ldc + col) = v;
            }
        } else {
            constexpr int LPR = TW / 8;
            constexpr int RPI = 64 / LPR;
#pragma unroll
            for (int it = 0; it < 32 * MI / RPI; ++it) {
                const int r = it * RPI + lane / LPR, c = (lane % LPR) * 8;
                const float4 v0 = *reinterpret_cast<const float4*>(reg + r * TW + c);
                const float4 v1 = *reinterpret_cast<const float4*>(reg + r * TW + c + 4);
                f16x8 h;
                h[0] = (_Float16)v0.x; h[1] = (_Float16)v0.y; h[2] = (_Float16)v0.z; h[3] = (_Float16)v0.w;
                h[4] = (_Float16)v1.x; h[5] = (_Float16)v1.y; h[6] = (_Float16)v1.z; h[7] = (_Float16)v1.w;
                const int row = m0 + wm + r, col = n0 + wn + c;
                if (row < g.M && col < g.N) *reinterpret_cast<f16x8*>(reinterpret_cast<_Float16*>(g.C) + (int64_t)row * g.ldc + col) = h;
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = n0 + wn + j * 32 + fr;
            if (col >= g.N) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                if (ABL == 1) { if (acc[i][j][e] == 12345.678f) g.C[0] = 1.0f; }
                else if (ABL == 5) { if (row < g.M) reinterpret_cast<_Float16*>(g.C)[(int64_t)row * g.ldc + col] = (_Float16)acc[i][j][e]; }
                else if (ABL == 6) { if (row < g.M) __builtin_nontemporal_store(acc[i][j][e], &g.C[(int64_t)row * g.ldc + col]); }
                else if (row < g.M) g.C[(int64_t)row * g.ldc + col] = acc[i][j][e];
            }
        }
    }
    if (ABL == 20) {
        t2 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t3 = __builtin_readcyclecounter();
        if (tid == 0) {
            g.tim[blockIdx.x * 4 + 0] = t0; g.tim[blockIdx.x * 4 + 1] = t1; g.tim[blockIdx.x * 4 + 2] = t2; g.tim[blockIdx.x * 4 + 3] = t3;
        }
    }
}

// producer / consumer specialisation: WM*WN compute waves (ds_read + MFMA only) and LW loader waves (LDS-DMA only), one barrier
// per k-step for everyone.  The DMA issue (~90 cycles per instruction under load) no longer blocks the waves that feed the MFMA pipe.
template <int WM, int WN, int MI, int NJ, int NA, int BK, int LW, int MINB, int GM>
__global__ __launch_bounds__(64 * (WM * WN + LW), MINB) void k_gemm_spec(GArgs g) {
    constexpr int NCW = WM * WN;
    constexpr int TBM = 32 * MI * WM, TBN = 32 * NJ * WN;
    constexpr int CH = BK / 8, RPB = 16 / CH, RPP = 64 / CH;
    constexpr int ROWB = BK * 2;
    constexpr int PA = TBM / RPP, PW = TBN / RPP;
    constexpr int NPIECE = NA * PA + PW;
    static_assert(NPIECE % LW == 0, "pieces per loader wave");
    constexpr int PPL = NPIECE / LW;
    constexpr int BUFB = NPIECE * 1024;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int nbm = (g.M + TBM - 1) / TBM, nbn = (g.N + TBN - 1) / TBN;
    int tm, tn;
    tile_of_block(nbm, nbn, tm, tn);
    if (GM > 0) {
        const int id = tn * nbm + tm;
        const int per = GM * nbn;
        const int grp = id / per, rem = id % per;
        const int gm0 = grp * GM;
        const int gsz = nbm - gm0 < GM ? nbm - gm0 : GM;
        tm = gm0 + rem % gsz;
        tn = rem / gsz;
    }
    const int m0 = tm * TBM, n0 = tn * TBN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nk = g.K / BK;
    if (wave >= NCW) {                                                  // ---- loader waves
        const int lw = wave - NCW;
        const _Float16* src[PPL];
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const int p = j * LW + lw;
            const int prow = lane / CH;
            int r;
            const _Float16* base;
            if (p < NA * PA) {
                const int pl = p / PA;
                r = (p % PA) * RPP + prow;
                const int rg = m0 + r < g.M ? m0 + r : g.M - 1;
                base = (pl == 0 ? g.A : g.A2) + (int64_t)rg * g.lda;
            } else {
                r = (p - NA * PA) * RPP + prow;
                const int rg = n0 + r < g.N ? n0 + r : g.N - 1;
                base = g.W + (int64_t)rg * g.ldw;
            }
            src[j] = base + ((lane % CH) ^ ((r / RPB) & (CH - 1))) * 8;
        }
        auto stage = [&](int buf, int k0) {
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
                const int p = j * LW + lw;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + k0),
                                                 (__attribute__((address_space(3))) void*)(lds + buf * BUFB + p * 1024), 16, 0, 0);
            }
        };
        stage(0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 1 < nk) stage((kt + 1) & 1, (kt + 1) * BK);
        }
        return;
    }
    const int wm = (wave / WN) * 32 * MI, wn = (wave % WN) * 32 * NJ;
    const int fr = lane & 31, fh = lane >> 5;
    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    const int swz = (fr / RPB) & (CH - 1);
    for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_s_barrier();
        const char* bA = lds + (kt & 1) * BUFB;
        const char* bW = bA + NA * PA * 1024;
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            const int coff = ((2 * s + fh) ^ swz) * 16;
            f16x8 b[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[j] = *reinterpret_cast<const f16x8*>(bW + (wn + j * 32 + fr) * ROWB + coff);
#pragma unroll
            for (int pl = 0; pl < NA; ++pl) {
                f16x8 a[MI];
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const f16x8*>(bA + pl * PA * 1024 + (wm + i * 32 + fr) * ROWB + coff);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = n0 + wn + j * 32 + fr;
            if (col >= g.N) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                if (row < g.M) g.C[(int64_t)row * g.ldc + col] = acc[i][j][e];
            }
        }
    }
}

__global__ void k_ref(GArgs g, int na, float* out, int rows) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)rows * g.N) return;
    const int m = (int)(idx / g.N) * (g.M / rows), n = (int)(idx % g.N);
    double s = 0.0;
    for (int k = 0; k < g.K; ++k) {
        double a = (double)(float)g.A[(int64_t)m * g.lda + k];
        if (na == 2) a += (double)(float)g.A2[(int64_t)m * g.lda + k];
        s += a * (double)(float)g.W[(int64_t)n * g.ldw + k];
    }
    out[idx] = (float)s;
}

__global__ void k_fill(_Float16* p, int64_t n, uint32_t seed, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        uint32_t x = (uint32_t)i * 2654435761u ^ seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = (_Float16)(((float)(x & 0xffffff) / 8388608.0f - 1.0f) * scale);
    }
}

template <int WM, int WN, int MI, int NJ, int NA, int BK, int NBUF, int DEPTH, int MINB, int ABL = 0, int GM = 0>
float run(const GArgs& g, int iters, float* ref, int refrows) {
    constexpr int TBM = 32 * MI * WM, TBN = 32 * NJ * WN;
    constexpr size_t ldsb = (size_t)NBUF * (NA * TBM + TBN) * BK * 2;
    auto kern = k_gemm_g16<WM, WN, MI, NJ, NA, BK, NBUF, DEPTH, MINB, ABL, GM>;
    char name[128];
    snprintf(name, sizeof name, "%dx%d w%dx%d(%dx%d) NA%d BK%d buf%d d%d mb%d abl%d gm%d", TBM, TBN, WM, WN, MI, NJ, NA, BK, NBUF, DEPTH, MINB, ABL, GM);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    const int nblk = ((g.M + TBM - 1) / TBM) * ((g.N + TBN - 1) / TBN);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemset(g.C, 0, (size_t)g.M * g.ldc * 4));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(nblk), dim3(64 * WM * WN), ldsb, 0, g);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(nblk), dim3(64 * WM * WN), ldsb, 0, g);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    // check the sampled rows
    std::vector<float> hc((size_t)g.M * g.N), hr((size_t)refrows * g.N);
    CK(hipMemcpy(hc.data(), g.C, hc.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), ref, hr.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    for (int r = 0; r < refrows; ++r) {
        const int m = r * (g.M / refrows);
        for (int n = 0; n < g.N; ++n) {
            maxerr = std::max(maxerr, (double)fabsf(hc[(size_t)m * g.N + n] - hr[(size_t)r * g.N + n]));
            maxref = std::max(maxref, (double)fabsf(hr[(size_t)r * g.N + n]));
        }
    }
    const double tf = 2.0 * g.M * g.N * g.K * NA / ms / 1e9;
    printf("  %-40s %4d blocks  %8.4f ms  %8.1f TF (MFMA work; useful %.1f)  rel err %.2e\n", name, nblk, ms, tf, tf / NA, maxerr / maxref);
    return ms;
}

template <int WM, int WN, int MI, int NJ, int NA, int BK, int LW, int MINB, int GM>
float run_spec(const GArgs& g, int iters, float* ref, int refrows) {
    constexpr int TBM = 32 * MI * WM, TBN = 32 * NJ * WN;
    constexpr size_t ldsb = (size_t)2 * (NA * TBM + TBN) * BK * 2;
    auto kern = k_gemm_spec<WM, WN, MI, NJ, NA, BK, LW, MINB, GM>;
    char name[128];
    snprintf(name, sizeof name, "SPEC %dx%d w%dx%d(%dx%d) NA%d BK%d lw%d mb%d gm%d", TBM, TBN, WM, WN, MI, NJ, NA, BK, LW, MINB, GM);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    const int nblk = ((g.M + TBM - 1) / TBM) * ((g.N + TBN - 1) / TBN);
    const int nthr = 64 * (WM * WN + LW);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemset(g.C, 0, (size_t)g.M * g.ldc * 4));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(nblk), dim3(nthr), ldsb, 0, g);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(nblk), dim3(nthr), ldsb, 0, g);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    std::vector<float> hc((size_t)g.M * g.N), hr((size_t)refrows * g.N);
    CK(hipMemcpy(hc.data(), g.C, hc.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), ref, hr.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    for (int r = 0; r < refrows; ++r) {
        const int m = r * (g.M / refrows);
        for (int n = 0; n < g.N; ++n) {
            maxerr = std::max(maxerr, (double)fabsf(hc[(size_t)m * g.N + n] - hr[(size_t)r * g.N + n]));
            maxref = std::max(maxref, (double)fabsf(hr[(size_t)r * g.N + n]));
        }
    }
    const double tf = 2.0 * g.M * g.N * g.K * NA / ms / 1e9;
    printf("  %-40s %4d blocks  %8.4f ms  %8.1f TF (MFMA work; useful %.1f)  rel err %.2e\n", name, nblk, ms, tf, tf / NA, maxerr / maxref);
    return ms;
}

int main() {
    const int shapes[][3] = {{5000, 2304, 768}, {5000, 768, 768}, {5000, 3072, 768}, {5000, 768, 3072}, {20000, 2304, 768}, {20000, 768, 768},
                             {20000, 3072, 768}, {20000, 768, 3072}, {3200, 1536, 512}, {3200, 512, 2048}, {4096, 4096, 4096}, {8192, 8192, 8192}};
    for (auto& s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        _Float16 *A, *A2, *W;
        float *C, *ref;
        CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&A2, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2));
        CK(hipMalloc(&C, (size_t)M * N * 4));
        const int refrows = 16;
        CK(hipMalloc(&ref, (size_t)refrows * N * 4));
        hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, A, (int64_t)M * K, 1u, 1.0f);
        hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, A2, (int64_t)M * K, 2u, 0.001f);
        hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, W, (int64_t)N * K, 3u, 1.0f);
        unsigned long long* tim;
        CK(hipMalloc(&tim, 8192 * 16 * 8));
        GArgs g{A, A2, W, C, K, K, N, M, N, K, tim};
        printf("M=%d N=%d K=%d\n", M, N, K);
        const int iters = 20;
        hipLaunchKernelGGL(k_ref, dim3((refrows * N + 255) / 256), dim3(256), 0, 0, g, 1, ref, refrows);
        run<4, 2, 1, 2, 1, 64, 2, 1, 2, 0, 8>(g, iters, ref, refrows);
        run<4, 2, 1, 2, 1, 64, 2, 1, 2, 31, 8>(g, iters, ref, refrows);
        hipLaunchKernelGGL(k_ref, dim3((refrows * N + 255) / 256), dim3(256), 0, 0, g, 2, ref, refrows);
        run<2, 4, 2, 2, 2, 32, 2, 1, 1, 0, 8>(g, iters, ref, refrows);
        run<2, 4, 2, 2, 2, 32, 2, 1, 1, 31, 8>(g, iters, ref, refrows);
        run<2, 2, 2, 3, 2, 32, 2, 1, 2, 0, 8>(g, iters, ref, refrows);
        run<2, 2, 2, 3, 2, 32, 2, 1, 2, 31, 8>(g, iters, ref, refrows);
        CK(hipFree(A)); CK(hipFree(A2)); CK(hipFree(W)); CK(hipFree(C)); CK(hipFree(ref));
    }
    return 0;
}
