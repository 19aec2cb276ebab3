// Source-side addressing of the implicit-GEMM convolution kernels (shared by the forward/dgrad kernel and the
// wgrad kernel): how a destination pixel and a filter tap select the source pixel(s), with zero / reflection
// padding, the fused nearest-upsample + channel-concat of the decoder, and the two adjoint gathers.
#pragma once
#include "mcav_common.h"
#include "../../include/mcav_conv.h"

namespace mcav {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CK = 16;         // K-tile depth: 16 input channels of one filter tap
constexpr int LDK = CK + 4;    // LDS row stride (floats) of a [rows][CK] operand tile: conflict-free ds_read_b128

struct GatherSrc {
    const float* x1;
    const float* x2;
    int B, Hs, Ws, C1, C2, up1;
    int mode, stride, sign, offset, pad_mode;
};

__device__ __forceinline__ int reflect_idx(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}

// 4 consecutive channels [c, c+4) of the logical (concatenated, possibly upsampled) source at pixel (n, sy, sx);
// the caller guarantees 0 <= sy < Hs, 0 <= sx < Ws.  Channels past C1 + C2 read as zero.
__device__ __forceinline__ f32x4 load_src4(const GatherSrc& g, int n, int sy, int sx, int c) {
    const float* p;
    int C, cc;
    size_t pix;
    if (c < g.C1) {
        p = g.x1; C = g.C1; cc = c;
        pix = g.up1 ? ((size_t)(n * (g.Hs >> 1) + (sy >> 1)) * (g.Ws >> 1) + (sx >> 1)) : ((size_t)(n * g.Hs + sy) * g.Ws + sx);
    } else {
        p = g.x2; C = g.C2; cc = c - g.C1;
        pix = (size_t)(n * g.Hs + sy) * g.Ws + sx;
    }
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (cc >= C) return v;                      // K padding beyond the real channels
    const float* q = p + pix * C + cc;
    if (((C & 3) == 0) && cc + 4 <= C) {
        v = *reinterpret_cast<const f32x4*>(q);
    } else {                                    // narrow tensors (C = 1 disparity maps): scalar, masked
        if (cc + 0 < C) v.x = q[0];
        if (cc + 1 < C) v.y = q[1];
        if (cc + 2 < C) v.z = q[2];
        if (cc + 3 < C) v.w = q[3];
    }
    return v;
}

// One element group of the A operand: destination pixel (n, dy, dx), filter tap (ky, kx), channels [c, c+4).
__device__ __forceinline__ f32x4 gather4(const GatherSrc& g, int n, int dy, int dx, int ky, int kx, int c) {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (g.mode == MCAV_G_DIRECT || g.mode == MCAV_G_SMALLC) {
        int sy = dy * g.stride + g.sign * ky + g.offset;
        int sx = dx * g.stride + g.sign * kx + g.offset;
        if (g.pad_mode == MCAV_PAD_REFLECT) {
            sy = reflect_idx(sy, g.Hs);
            sx = reflect_idx(sx, g.Ws);
        } else if ((unsigned)sy >= (unsigned)g.Hs || (unsigned)sx >= (unsigned)g.Ws) {
            return z;
        }
        return load_src4(g, n, sy, sx, c);
    }
    if (g.mode == MCAV_G_ADJ_REFLECT) {
        // forward: y[o] = sum_k w[k] x[reflect(o + k - 1)], 3x3; so x[d] collects dy[d + 1 - k] plus, on the rows/cols next
        // to the border, the output whose padded tap reflected onto d.
        int ys[2], xs[2], ny = 0, nx = 0;
        const int sy = dy + 1 - ky, sx = dx + 1 - kx;
        if ((unsigned)sy < (unsigned)g.Hs) ys[ny++] = sy;
        if (dy == 1 && ky == 0) ys[ny++] = 0;
        if (dy == g.Hs - 2 && ky == 2) ys[ny++] = g.Hs - 1;
        if ((unsigned)sx < (unsigned)g.Ws) xs[nx++] = sx;
        if (dx == 1 && kx == 0) xs[nx++] = 0;
        if (dx == g.Ws - 2 && kx == 2) xs[nx++] = g.Ws - 1;
        for (int a = 0; a < ny; ++a)
            for (int b = 0; b < nx; ++b) z += load_src4(g, n, ys[a], xs[b], c);
        return z;
    }
    // MCAV_G_ADJ_STRIDE2: forward y[o] = sum_k w[k] x[2 o + k - pad]; x[d] collects dy[(d + pad - k) / 2] when that is whole
    const int ty = dy + g.offset - ky, tx = dx + g.offset - kx;
    if (ty < 0 || tx < 0 || ((ty | tx) & 1)) return z;
    const int sy = ty >> 1, sx = tx >> 1;
    if (sy >= g.Hs || sx >= g.Ws) return z;
    return load_src4(g, n, sy, sx, c);
}

// XCD-aware bijective remap of a linear workgroup id: consecutive logical ids share an XCD (and its L2).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// ------------------------------------------------------------------------------------------------ shared by the conv kernels
struct IgemmParams {
    GatherSrc g;
    const float* w;
    int kh, kw, Kp, Kstride, taps;
    float* y;
    int Hd, Wd, Cd, n_begin, n_count, y_choff;
    const float* bias;
    int act;
    const float* dact_aux;
    int dact;
    const float* addend;
    int pool;
    float* stats;
    const float *stats_x, *stats_mean, *stats_invstd;      // BatchNorm-backward form of the statistics (mcav_igemm_desc.stats_x)
    int M;           // rows of the GEMM (incl. class / group padding)
    int Mc, McP;     // ADJ_STRIDE2: pixels per parity class and its BM-padded size; groups > 1: rows per group and padded size
    int groups;
    int mtiles, ntiles;
    int no_tab;      // desc.tile bit 8: force the general kernel (A/B timing and parity of both paths)
    const float* wm; // merged-tap filter copy (mcav_pack_weights_upmerge)
    int Np_all;      // packed filter rows (desc.Np)
    int bm;          // rows per tile of the chosen config (UPM row order: tiles of the four classes of one region are adjacent)
    int upm;         // 1: rows are grouped by output parity class and the x1 part of K runs as 4 merged taps on the low-resolution source
    int ksplit;      // > 1: the K loop of a tile is cut into ksplit workgroups (launches of a handful of tiles: PoseNet's 2x5 .. 6x20 maps);
    float* kslab;    //      each writes its raw partial tile to kslab[split] (y-shaped), splitk_finish_kernel sums them and applies the epilogue
};

__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == MCAV_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == MCAV_ACT_ELU) return v > 0.f ? v : expm1f(v);
    if (act == MCAV_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}

// derivative of the activation expressed through its OUTPUT y
__device__ __forceinline__ float act_bwd(float y, int act) {
    if (act == MCAV_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == MCAV_ACT_ELU) return y > 0.f ? 1.f : y + 1.f;
    if (act == MCAV_ACT_SIGMOID) return y * (1.f - y);
    return 1.f;
}

// Raw buffer loads: an offset at or beyond num_records reads as zero in hardware, so padding / out-of-image rows need
// neither a branch nor a select after the load (either would force an s_waitcnt right behind it).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;      // "reads as zero": every tensor here is < 2 GiB (checked on the host)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* ptr, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, bytes, 0x00020000);
}

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}

__device__ __forceinline__ f32x4 buf_load4s(__amdgpu_buffer_rsrc_t r, unsigned byte_off, int sbyte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, sbyte_off, 0));
}

__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}

// (a store at or beyond num_records is dropped in hardware: rows past the tensor need no branch)
__device__ __forceinline__ void buf_store1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)byte_off, 0, 0);
}

}  // namespace mcav
