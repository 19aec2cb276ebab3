// 3x3 reflection-padded convolution with ONE output channel: the disparity heads of the depth decoder
// (reference models/depth/resnet_dispnet.py:66-68 `dispconv`, layers.py:42-58 Conv3x3; forward at :93-94 with a sigmoid).
//
// An implicit GEMM with N = 1 wastes 15/16 of an MFMA tile and, at 192x640, made this head the slowest layer of the
// step although it is 2 % of the FLOPs.  It is an HBM-bound stencil: 64 B of input per pixel (C = 16) against 4 B out.
// Mapping: C/4 adjacent lanes own one pixel, each lane 4 channels, so every load / store of the [B,H,W,C] tensors is a
// contiguous 16 B per lane and 1 KB per wavefront; the 9 x 4 filter taps of a lane live in registers.
//   forward : y = act(bias + sum_tap <x[reflect(p + tap)], w[tap]>), lanes of a pixel reduced with DPP shuffles
//   backward: ONE pass over x produces all three gradients.  With dpre = dy * act'(y) and the adjoint gather
//             G[p][tap] = sum of dpre[q] over the outputs q whose (reflected) tap landed on p:
//               dx[p][c]   = (sum_tap G[p][tap] w[c][tap]) * act_x'(x[p][c]) + addend[p][c]
//               dw[c][tap] = sum_p G[p][tap] x[p][c]          dbias = sum_p dpre[p]
//             (x is read once, 64 B per pixel; the generic path read it 9 times for wgrad and again for dgrad).
// Reductions: lane accumulators -> wavefront shuffles -> LDS -> per-block slab -> fixed-order finalize (no atomics).
#include "conv_gather.h"
#include "kernel_timer.h"

namespace mcav {

__device__ __forceinline__ float nc_act_fwd(float v, int act) {
    if (act == MCAV_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == MCAV_ACT_ELU) return v > 0.f ? v : expm1f(v);
    if (act == MCAV_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}

// derivative of the activation expressed through its OUTPUT
__device__ __forceinline__ float nc_act_bwd(float y, int act) {
    if (act == MCAV_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == MCAV_ACT_ELU) return y > 0.f ? 1.f : y + 1.f;
    if (act == MCAV_ACT_SIGMOID) return y * (1.f - y);
    return 1.f;
}

template <int LPP>
__global__ __launch_bounds__(256) void conv3x3r_c1_fwd_kernel(const float* __restrict__ x, int B, int H, int W, const float* __restrict__ w,
                                                              const float* __restrict__ bias, int act, float* __restrict__ y) {
    constexpr int C = 4 * LPP;
    const int c4 = threadIdx.x % LPP;
    f32x4 wt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        wt[t].x = w[(c4 * 4 + 0) * 9 + t]; wt[t].y = w[(c4 * 4 + 1) * 9 + t];
        wt[t].z = w[(c4 * 4 + 2) * 9 + t]; wt[t].w = w[(c4 * 4 + 3) * 9 + t];
    }
    const float bv = bias ? bias[0] : 0.f;
    const long npix = (long)B * H * W;
    // the LPP lanes of a pixel share p, so they leave the loop together: the shuffles below only ever pair active lanes
    for (long p = (long)blockIdx.x * (256 / LPP) + threadIdx.x / LPP; p < npix; p += (long)gridDim.x * (256 / LPP)) {
        const int n = (int)(p / ((long)H * W));
        const int r = (int)(p - (long)n * H * W);
        const int py = r / W, px = r - py * W;
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int sy = reflect_idx(py + ky - 1, H);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int sx = reflect_idx(px + kx - 1, W);
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((size_t)(n * H + sy) * W + sx) * C + c4 * 4);
                const f32x4 q = wt[ky * 3 + kx];
                acc += (v.x * q.x + v.y * q.y) + (v.z * q.z + v.w * q.w);
            }
        }
#pragma unroll
        for (int off = 1; off < LPP; off <<= 1) acc += __shfl_xor(acc, off, 64);
        if (c4 == 0) y[p] = nc_act_fwd(acc + bv, act);
    }
}

// slab row layout of the backward partials: [9 * C] filter gradients as (c * 9 + tap), then the bias gradient
template <int LPP>
__global__ __launch_bounds__(256) void conv3x3r_c1_bwd_kernel(const float* __restrict__ x, int B, int H, int W, const float* __restrict__ w,
                                                              const float* __restrict__ dy, const float* __restrict__ yout, int act, int x_act,
                                                              const float* __restrict__ addend, float* __restrict__ dx, float* __restrict__ slab) {
    constexpr int C = 4 * LPP;
    constexpr int PPB = 256 / LPP;                    // pixels per block pass
    __shared__ float red[4][9 * C + 1];
    const int c4 = threadIdx.x % LPP, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 wt[9], dwv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        wt[t].x = w[(c4 * 4 + 0) * 9 + t]; wt[t].y = w[(c4 * 4 + 1) * 9 + t];
        wt[t].z = w[(c4 * 4 + 2) * 9 + t]; wt[t].w = w[(c4 * 4 + 3) * 9 + t];
        dwv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float db = 0.f;
    const long npix = (long)B * H * W;
    auto dpre_at = [&](int n, int yy, int xx) -> float {      // yout == NULL: dy already is the pre-activation gradient
        const size_t o = (size_t)(n * H + yy) * W + xx;
        return yout ? dy[o] * nc_act_bwd(yout[o], act) : dy[o];
    };
    for (long p = (long)blockIdx.x * PPB + threadIdx.x / LPP; p < npix; p += (long)gridDim.x * PPB) {
        const int n = (int)(p / ((long)H * W));
        const int r = (int)(p - (long)n * H * W);
        const int py = r / W, px = r - py * W;
        float G[9];
        const bool interior = py >= 2 && py <= H - 3 && px >= 2 && px <= W - 3;
        if (__all(interior)) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) G[ky * 3 + kx] = dpre_at(n, py + 1 - ky, px + 1 - kx);
        } else {
            // outputs whose tap (ky, kx) reads this pixel: the plain one (py + 1 - ky) and, next to the border, the output
            // on the border line whose padded tap reflected onto this pixel
            int ya[3], yb[3], xa[3], xb[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int sy = py + 1 - k, sx = px + 1 - k;
                ya[k] = (unsigned)sy < (unsigned)H ? sy : -1;
                yb[k] = (py == 1 && k == 0) ? 0 : ((py == H - 2 && k == 2) ? H - 1 : -1);
                xa[k] = (unsigned)sx < (unsigned)W ? sx : -1;
                xb[k] = (px == 1 && k == 0) ? 0 : ((px == W - 2 && k == 2) ? W - 1 : -1);
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    float s = 0.f;
                    if (ya[ky] >= 0 && xa[kx] >= 0) s += dpre_at(n, ya[ky], xa[kx]);
                    if (ya[ky] >= 0 && xb[kx] >= 0) s += dpre_at(n, ya[ky], xb[kx]);
                    if (yb[ky] >= 0 && xa[kx] >= 0) s += dpre_at(n, yb[ky], xa[kx]);
                    if (yb[ky] >= 0 && xb[kx] >= 0) s += dpre_at(n, yb[ky], xb[kx]);
                    G[ky * 3 + kx] = s;
                }
        }
        const size_t off = (size_t)p * C + c4 * 4;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + off);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            dwv[t] += xv * G[t];
            g += wt[t] * G[t];
        }
        if (c4 == 0) db += dpre_at(n, py, px);
        g.x *= nc_act_bwd(xv.x, x_act); g.y *= nc_act_bwd(xv.y, x_act); g.z *= nc_act_bwd(xv.z, x_act); g.w *= nc_act_bwd(xv.w, x_act);
        if (addend) g += *reinterpret_cast<const f32x4*>(addend + off);
        *reinterpret_cast<f32x4*>(dx + off) = g;
    }
    // lanes holding the same channel group, then the 4 wavefronts
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int off = LPP; off < 64; off <<= 1) {
            dwv[t].x += __shfl_xor(dwv[t].x, off, 64); dwv[t].y += __shfl_xor(dwv[t].y, off, 64);
            dwv[t].z += __shfl_xor(dwv[t].z, off, 64); dwv[t].w += __shfl_xor(dwv[t].w, off, 64);
        }
    }
#pragma unroll
    for (int off = LPP; off < 64; off <<= 1) db += __shfl_xor(db, off, 64);
    if (lane < LPP) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            red[wave][(c4 * 4 + 0) * 9 + t] = dwv[t].x; red[wave][(c4 * 4 + 1) * 9 + t] = dwv[t].y;
            red[wave][(c4 * 4 + 2) * 9 + t] = dwv[t].z; red[wave][(c4 * 4 + 3) * 9 + t] = dwv[t].w;
        }
        if (lane == 0) red[wave][9 * C] = db;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 9 * C + 1; e += 256)
        slab[(size_t)blockIdx.x * (9 * C + 1) + e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

__global__ __launch_bounds__(256) void conv3x3r_c1_finalize_kernel(const float* slab, int blocks, int C, float* dw, float* dbias, int accumulate) {
    __shared__ float part[256];
    const int e = blockIdx.x, n = 9 * C + 1;          // one block per output element, 256 lanes over the per-block partials
    float s = 0.f;
    for (int k = threadIdx.x; k < blocks; k += 256) s += slab[(size_t)k * n + e];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) part[threadIdx.x] += part[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (e < 9 * C) dw[e] = accumulate ? dw[e] + part[0] : part[0];
        else if (dbias) dbias[0] = accumulate ? dbias[0] + part[0] : part[0];
    }
}

constexpr int NC_BWD_BLOCKS = 1024;

inline bool nc_ok(int C) { return C == 16 || C == 32 || C == 64 || C == 128; }

}  // namespace mcav

using namespace mcav;

MCAV_EXPORT int mcav_conv3x3r_c1_fwd(const float* x, int B, int H, int W, int C, const float* w_oihw, const float* bias, int act, float* y,
                                     void* stream) {
    if (!x || !w_oihw || !y || B <= 0 || H < 2 || W < 2 || !nc_ok(C)) return MCAV_E_INVALID;
    const long npix = (long)B * H * W;
    if (npix >= (1L << 31)) return MCAV_E_INVALID;
    hipStream_t s = as_stream(stream);
    const int ppb = 256 / (C / 4);
    long blocks = (npix + ppb - 1) / ppb;
    if (blocks > 8192) blocks = 8192;
    switch (C) {
        case 16: timed_launch(conv3x3r_c1_fwd_kernel<4>, dim3((int)blocks), dim3(256), 0, s, x, B, H, W, w_oihw, bias, act, y); break;
        case 32: timed_launch(conv3x3r_c1_fwd_kernel<8>, dim3((int)blocks), dim3(256), 0, s, x, B, H, W, w_oihw, bias, act, y); break;
        case 64: timed_launch(conv3x3r_c1_fwd_kernel<16>, dim3((int)blocks), dim3(256), 0, s, x, B, H, W, w_oihw, bias, act, y); break;
        default: timed_launch(conv3x3r_c1_fwd_kernel<32>, dim3((int)blocks), dim3(256), 0, s, x, B, H, W, w_oihw, bias, act, y); break;
    }
    return launch_status();
}

MCAV_EXPORT size_t mcav_conv3x3r_c1_bwd_workspace_bytes(int C) {
    return nc_ok(C) ? align_up(sizeof(float) * (size_t)NC_BWD_BLOCKS * (9 * C + 1), 256) : 0;
}

MCAV_EXPORT int mcav_conv3x3r_c1_bwd(const float* x, int B, int H, int W, int C, const float* w_oihw, const float* dy, const float* y, int act,
                                     int x_act, const float* addend, float* dx, float* dw_oihw, float* dbias, int accumulate, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    if (!x || !w_oihw || !dy || !dx || !dw_oihw || !workspace || B <= 0 || H < 2 || W < 2 || !nc_ok(C)) return MCAV_E_INVALID;
    if (!y && act != MCAV_ACT_NONE) return MCAV_E_INVALID;
    if (workspace_bytes < mcav_conv3x3r_c1_bwd_workspace_bytes(C)) return MCAV_E_WORKSPACE;
    const long npix = (long)B * H * W;
    if (npix >= (1L << 31)) return MCAV_E_INVALID;
    hipStream_t s = as_stream(stream);
    const int ppb = 256 / (C / 4);
    long blocks = (npix + ppb - 1) / ppb;
    if (blocks > NC_BWD_BLOCKS) blocks = NC_BWD_BLOCKS;
    float* slab = reinterpret_cast<float*>(workspace);
    switch (C) {
        case 16: timed_launch(conv3x3r_c1_bwd_kernel<4>, dim3((int)blocks), dim3(256), 0, s, x, B, H, W, w_oihw, dy, y, act, x_act, addend, dx, slab); break;
        case 32: timed_launch(conv3x3r_c1_bwd_kernel<8>, dim3((int)blocks), dim3(256), 0, s, x, B, H, W, w_oihw, dy, y, act, x_act, addend, dx, slab); break;
        case 64: timed_launch(conv3x3r_c1_bwd_kernel<16>, dim3((int)blocks), dim3(256), 0, s, x, B, H, W, w_oihw, dy, y, act, x_act, addend, dx, slab); break;
        default: timed_launch(conv3x3r_c1_bwd_kernel<32>, dim3((int)blocks), dim3(256), 0, s, x, B, H, W, w_oihw, dy, y, act, x_act, addend, dx, slab); break;
    }
    conv3x3r_c1_finalize_kernel<<<9 * C + 1, 256, 0, s>>>(slab, (int)blocks, C, dw_oihw, dbias, accumulate);
    return launch_status();
}
