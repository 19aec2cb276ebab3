// The step BEFORE the training-step hot path: the reference's image transform chain
//   float32(decoded bytes) / 255 -> ToTensor -> ToPILImage -> Resize((h, w)) -> ToTensor -> Normalize     (dataloaders.py:32-49, trainer.py:97-103)
// on the GPU, for a batch of decoded uint8 HWC images, BIT-EXACT against Pillow's resize (ImagingResample, Resample.c): antialiased
// separable triangle filter in 22-bit fixed point, horizontal pass into an 8-bit intermediate, then the vertical pass -- fused here
// with byte / 255, the ImageNet normalisation and the HWC -> CHW transposition.  The coefficient tables are computed on the HOST in
// double exactly as Pillow does (mcav_resample_coeffs), so the device side is pure integer arithmetic.
// (ToPILImage's float -> mul(255).byte() truncation is the identity on float32(v)/255 for every byte v, so it needs no kernel.)
#include <math.h>

#include "mcav_common.h"

namespace mcav {

constexpr int PP_BITS = 32 - 8 - 2;

// src [B][H0][W0][3] -> tmp [B][H0][w][3]
__global__ __launch_bounds__(256) void pp_horizontal_kernel(const uint8_t* __restrict__ src, int B, int H0, int W0, int w, const int* __restrict__ bounds,
                                                            const int* __restrict__ kk, int ksize, uint8_t* __restrict__ tmp) {
    const size_t total = (size_t)B * H0 * w;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int xx = (int)(i % w);
        const size_t row = i / w;                               // b * H0 + y
        const int xmin = bounds[2 * xx], xmax = bounds[2 * xx + 1];
        const int* k = kk + (size_t)xx * ksize;
        const uint8_t* s = src + (row * W0 + xmin) * 3;
        int a0 = 1 << (PP_BITS - 1), a1 = a0, a2 = a0;
        for (int x = 0; x < xmax; ++x) {
            const int kv = k[x];
            a0 += (int)s[3 * x + 0] * kv; a1 += (int)s[3 * x + 1] * kv; a2 += (int)s[3 * x + 2] * kv;
        }
        uint8_t* o = tmp + i * 3;
        o[0] = (uint8_t)min(max(a0 >> PP_BITS, 0), 255); o[1] = (uint8_t)min(max(a1 >> PP_BITS, 0), 255); o[2] = (uint8_t)min(max(a2 >> PP_BITS, 0), 255);
    }
}

// tmp [B][H0][w][3] -> dst [B][3][h][w] = ((vertical pass) / 255 - mean) / std
__global__ __launch_bounds__(256) void pp_vertical_kernel(const uint8_t* __restrict__ tmp, int B, int H0, int w, int h, const int* __restrict__ bounds,
                                                          const int* __restrict__ kk, int ksize, float m0, float m1, float m2, float s0, float s1,
                                                          float s2, float* __restrict__ dst) {
    const size_t total = (size_t)B * h * w;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % w);
        const int yy = (int)((i / w) % h);
        const size_t b = i / ((size_t)w * h);
        const int ymin = bounds[2 * yy], ymax = bounds[2 * yy + 1];
        const int* k = kk + (size_t)yy * ksize;
        const uint8_t* s = tmp + ((b * H0 + ymin) * w + x) * 3;
        int a0 = 1 << (PP_BITS - 1), a1 = a0, a2 = a0;
        for (int y = 0; y < ymax; ++y) {
            const int kv = k[y];
            const uint8_t* q = s + (size_t)y * w * 3;
            a0 += (int)q[0] * kv; a1 += (int)q[1] * kv; a2 += (int)q[2] * kv;
        }
        const float v0 = (float)min(max(a0 >> PP_BITS, 0), 255) / 255.0f, v1 = (float)min(max(a1 >> PP_BITS, 0), 255) / 255.0f,
                    v2 = (float)min(max(a2 >> PP_BITS, 0), 255) / 255.0f;
        const size_t plane = (size_t)h * w, o = b * 3 * plane + (size_t)yy * w + x;
        dst[o] = (v0 - m0) / s0;
        dst[o + plane] = (v1 - m1) / s1;
        dst[o + 2 * plane] = (v2 - m2) / s2;
    }
}

}  // namespace mcav

using namespace mcav;

// Pillow's precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter over the whole axis (host side, double arithmetic in
// Pillow's operation order).  bounds: [out_size][2] = (first source index, count); kk: [out_size][*ksize] fixed-point weights.
// Returns the required kk capacity (out_size * ksize) when kk is NULL or kk_capacity is too small (nothing written then).
MCAV_EXPORT int mcav_resample_coeffs(int in_size, int out_size, int* ksize_out, int* bounds, int* kk, int kk_capacity) {
    if (in_size <= 0 || out_size <= 0 || !ksize_out) return MCAV_E_INVALID;
    const double scale = (double)((float)in_size - 0.0f) / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    *ksize_out = ksize;
    if (!bounds || !kk || kk_capacity < out_size * ksize) return out_size * ksize;
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = 0.0 + (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        int* k = kk + (size_t)xx * ksize;
        double wbuf[64];
        if (xmax > 64) return MCAV_E_INVALID;                   // scale factors above ~30: not an image-loader case
        for (int x = 0; x < xmax; ++x) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            const double w = a < 1.0 ? 1.0 - a : 0.0;
            wbuf[x] = w;
            ww += w;
        }
        int x = 0;
        for (; x < xmax; ++x) {
            const double v = ww != 0.0 ? wbuf[x] / ww : wbuf[x];
            k[x] = v < 0 ? (int)(-0.5 + v * (1 << PP_BITS)) : (int)(0.5 + v * (1 << PP_BITS));
        }
        for (; x < ksize; ++x) k[x] = 0;
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    return MCAV_OK;
}

MCAV_EXPORT size_t mcav_image_preprocess_workspace_bytes(int B, int H0, int w) {
    return (B > 0 && H0 > 0 && w > 0) ? align_up((size_t)B * H0 * w * 3, 256) : 0;
}

// src: device uint8 [B][H0][W0][3] (decoded RGB).  hbounds/hkk (width axis, W0 -> w) and vbounds/vkk (height axis, H0 -> h): DEVICE copies of
// mcav_resample_coeffs' tables.  mean3 / std3: host.  dst: device float [B][3][h][w].
MCAV_EXPORT int mcav_image_preprocess(const uint8_t* src, int B, int H0, int W0, int h, int w, const int* hbounds, const int* hkk, int hksize,
                                      const int* vbounds, const int* vkk, int vksize, const float* mean3, const float* std3, float* dst,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    if (!src || !dst || !hbounds || !hkk || !vbounds || !vkk || !mean3 || !std3 || !workspace || B <= 0 || H0 <= 0 || W0 <= 0 || h <= 0 || w <= 0)
        return MCAV_E_INVALID;
    if (workspace_bytes < mcav_image_preprocess_workspace_bytes(B, H0, w)) return MCAV_E_WORKSPACE;
    hipStream_t s = as_stream(stream);
    uint8_t* tmp = reinterpret_cast<uint8_t*>(workspace);
    const size_t n1 = (size_t)B * H0 * w, n2 = (size_t)B * h * w;
    const int g1 = (int)((n1 + 255) / 256 < 8192 ? (n1 + 255) / 256 : 8192), g2 = (int)((n2 + 255) / 256 < 8192 ? (n2 + 255) / 256 : 8192);
    pp_horizontal_kernel<<<g1, 256, 0, s>>>(src, B, H0, W0, w, hbounds, hkk, hksize, tmp);
    pp_vertical_kernel<<<g2, 256, 0, s>>>(tmp, B, H0, w, h, vbounds, vkk, vksize, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], dst);
    return launch_status();
}
