// k_scan_hist_b (xmh_scan_bits.hip): pass 1 of the ranking scan for 128 / 256-bit codes, MFMA operands from the packed bits
#pragma once
#include "xmh_common.h"

namespace xmh {

constexpr int kScanBitsWaves = 4;                    // waves (16 queries each) per block; must equal kMfmaWaves of xmh_scan.hip

struct ScanBitsArgs {
    const uint32_t* rbits;
    const uint32_t* rlab;
    const uint32_t* qbits;
    const uint32_t* qlab;
    int Q, R, K, W, LW;
    int chunk, nchunk, nqt, nb, qpad;
    const uint32_t* rzero = nullptr;                 // ternary codes (round 6): the zero planes of both sides, else null
    const uint32_t* qzero = nullptr;
};

// nmc = code tiles of 64 bits (2: up to 128 bits, 4: up to 256); counters come out as (all << 16 | relevant) per (chunk, bucket, query).
// Ternary codes (a.rzero != null): K <= 64 with nmc = 2, K <= 128 with nmc = 4, K <= 256 with nmc = 8 -- the operand is the 2K-bit pair of planes
// [element is +1 | element is -1], the distance K - q.r in half units (2K + 1 bucket rows).
// (waves per counter table, query tiles per wave) of the instance launch_scan_hist_bits picks; for xmh_scan_describe
void scan_hist_bits_shape(int nmc, bool ternary, int* nsh, int* nqt);
int launch_scan_hist_bits(const ScanBitsArgs& a, int nmc, uint32_t* chunk_hist, uint4* cache, hipStream_t st);

}  // namespace xmh
