// Evaluation image transform on the GPU: the reference's
//     Compose([Resize((224, 224), interpolation=Image.BICUBIC), ToTensor(), Normalize(mean, std)])
// (dataset/transformer_dataset.py:38-42) on raw RGB uint8 images, bit-exact with Pillow's ImagingResample:
// two passes (horizontal, then vertical) of an 8-bit fixed-point convolution -- int32 coefficients with 22 fractional
// bits, accumulator started at 2^21, `>> 22`, clip to [0, 255] after EACH pass -- followed by float32
// ((u8 / 255) - mean) / std written channel-major.  The per-output-pixel coefficient tables depend only on
// (input size, output size); the host builds them once per image size (xmh/dataset/preprocess.py) and passes them in.
// Both passes are byte-streaming (HBM-bound: H*W*3 + H*Wo*3 (+ Ho*Wo*3 and Ho*Wo*12) bytes per image); one thread per
// output pixel, the three channels of a pixel together.
#include "xmh_common.h"

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= kPrecisionBits;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal: in [B][H][W][3] -> out [B][H][Wo][3]
__global__ __launch_bounds__(256) void k_resample_h(const uint8_t* __restrict__ in, int64_t rows, int W, int Wo,
                                                    const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk, int ksize,
                                                    uint8_t* __restrict__ out) {
    const int64_t total = rows * Wo;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / Wo;
        const int xo = (int)(e % Wo);
        const int xmin = bounds[2 * xo], cnt = bounds[2 * xo + 1];
        const int32_t* k = kk + (int64_t)xo * ksize;
        const uint8_t* p = in + (row * W + xmin) * 3;
        int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
        for (int x = 0; x < cnt; ++x) {
            const int w = k[x];
            s0 += (int)p[3 * x] * w;
            s1 += (int)p[3 * x + 1] * w;
            s2 += (int)p[3 * x + 2] * w;
        }
        uint8_t* o = out + e * 3;
        o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
    }
}

// vertical (+ optional normalise): in [B][H][Wo][3] -> u8 [B][Ho][Wo][3] and/or float [B][3][Ho][Wo]
__global__ __launch_bounds__(256) void k_resample_v(const uint8_t* __restrict__ in, int64_t B, int H, int Ho, int Wo,
                                                    const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk, int ksize,
                                                    float m0, float m1, float m2, float d0, float d1, float d2,
                                                    uint8_t* __restrict__ out_u8, float* __restrict__ out_f) {
    const int64_t total = B * Ho * Wo;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int xo = (int)(e % Wo);
        const int yo = (int)((e / Wo) % Ho);
        const int64_t b = e / ((int64_t)Wo * Ho);
        uint8_t r0, r1, r2;
        if (bounds) {
            const int ymin = bounds[2 * yo], cnt = bounds[2 * yo + 1];
            const int32_t* k = kk + (int64_t)yo * ksize;
            const uint8_t* p = in + ((b * H + ymin) * Wo + xo) * 3;
            int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
            for (int y = 0; y < cnt; ++y) {
                const int w = k[y];
                s0 += (int)p[0] * w;
                s1 += (int)p[1] * w;
                s2 += (int)p[2] * w;
                p += (int64_t)Wo * 3;
            }
            r0 = clip8(s0); r1 = clip8(s1); r2 = clip8(s2);
        } else {                                                       // H == Ho: no vertical pass in Pillow either
            const uint8_t* p = in + ((b * H + yo) * Wo + xo) * 3;
            r0 = p[0]; r1 = p[1]; r2 = p[2];
        }
        if (out_u8) {
            uint8_t* o = out_u8 + e * 3;
            o[0] = r0; o[1] = r1; o[2] = r2;
        }
        if (out_f) {
            // ToTensor: float32(u8) / 255; Normalize: (x - mean) / std -- IEEE divisions, no reciprocal shortcuts (bit parity)
            const int64_t plane = (int64_t)Ho * Wo;
            float* o = out_f + b * 3 * plane + (int64_t)yo * Wo + xo;
            o[0] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)r0, 255.0f), m0), d0);
            o[plane] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)r1, 255.0f), m1), d1);
            o[2 * plane] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)r2, 255.0f), m2), d2);
        }
    }
}

inline unsigned grid_for(int64_t work) {
    int64_t g = xmh::ceil_div(work, 256);
    const int64_t cap = (int64_t)xmh::device_cu_count() * 16;
    return (unsigned)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int xmh_image_preprocess_u8(const uint8_t* images, int64_t B, int H, int W, int out_h, int out_w,
                                       const int32_t* bounds_w, const int32_t* kk_w, int ksize_w,
                                       const int32_t* bounds_h, const int32_t* kk_h, int ksize_h,
                                       const float* mean3_host, const float* std3_host, uint8_t* tmp, uint8_t* resized_u8,
                                       float* out_chw, xmh_stream_t stream) {
    XMH_RANGE("xmh_image_preprocess_u8");
    if (B < 0 || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0) return xmh::fail(XMH_EINVAL, "xmh_image_preprocess_u8: bad shape");
    if (B == 0) return XMH_OK;
    if (!images || (!resized_u8 && !out_chw)) return xmh::fail(XMH_EINVAL, "xmh_image_preprocess_u8: null pointer");
    const bool need_h = W != out_w, need_v = H != out_h;
    if (need_h && (!bounds_w || !kk_w || ksize_w <= 0 || !tmp)) return xmh::fail(XMH_EINVAL, "xmh_image_preprocess_u8: horizontal tables / tmp buffer missing (W=%d -> %d)", W, out_w);
    if (need_v && (!bounds_h || !kk_h || ksize_h <= 0)) return xmh::fail(XMH_EINVAL, "xmh_image_preprocess_u8: vertical tables missing (H=%d -> %d)", H, out_h);
    if (out_chw && (!mean3_host || !std3_host)) return xmh::fail(XMH_EINVAL, "xmh_image_preprocess_u8: mean/std missing");
    hipStream_t st = xmh::as_stream(stream);
    const uint8_t* mid = images;
    if (need_h) {
        hipLaunchKernelGGL(k_resample_h, dim3(grid_for(B * H * out_w)), dim3(256), 0, st, images, B * H, W, out_w, bounds_w, kk_w, ksize_w, tmp);
        XMH_LAUNCH_CHECK("xmh_image_preprocess_u8 horizontal");
        mid = tmp;
    }
    const float m0 = out_chw ? mean3_host[0] : 0.f, m1 = out_chw ? mean3_host[1] : 0.f, m2 = out_chw ? mean3_host[2] : 0.f;
    const float d0 = out_chw ? std3_host[0] : 1.f, d1 = out_chw ? std3_host[1] : 1.f, d2 = out_chw ? std3_host[2] : 1.f;
    hipLaunchKernelGGL(k_resample_v, dim3(grid_for(B * out_h * out_w)), dim3(256), 0, st, mid, B, H, out_h, out_w,
                       need_v ? bounds_h : (const int32_t*)nullptr, kk_h, ksize_h, m0, m1, m2, d0, d1, d2, resized_u8, out_chw);
    XMH_LAUNCH_CHECK("xmh_image_preprocess_u8 vertical");
    return XMH_OK;
}
