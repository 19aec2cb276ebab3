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

// Tiling of both kernels: a block owns NC_TH image rows x PXB = 256 / LPP columns of one image and walks down the rows, so that every
// byte of x is fetched once (plus the tile's one-row / one-column halo: 18/16 x 66/64 at C = 16).  The loads of the next NC_PF rows are in
// flight while a row is computed (the walk has no other source of memory-level parallelism).
constexpr int NC_TH = 16, NC_PF = 4;
static_assert(NC_PF % 2 == 0, "the forward kernel's two-row ring is indexed by the unrolled position");

__device__ __forceinline__ int nc_reflect(int i, int n) { return min(max(reflect_idx(i, n), 0), n - 1); }      // (columns past a ragged tile clamp)

// forward: the row being read goes through LDS once (two-row ring), each lane takes its pixel's two neighbours from there and adds the
// row's contribution to the three output rows it touches (accumulators rotate; a row is finished after its ky = 2).  The tile's two halo
// columns (2 x (NC_TH + 2) pixels) are fetched once, up front, by the first threads.
template <int LPP>
__global__ __launch_bounds__(256, 4) void conv3x3r_c1_fwd_kernel(const float* __restrict__ x, int B, int H, int W, const float* __restrict__ w,
                                                              const float* __restrict__ bias, int act, float* __restrict__ y, int tiles_x, int tiles_y) {
    constexpr int C = 4 * LPP, PXB = 256 / LPP, NR = NC_TH + 2;
    __shared__ __attribute__((aligned(16))) float rows[2][PXB * C];
    __shared__ __attribute__((aligned(16))) float halo[NR][2][C];
    const int tid = threadIdx.x, c4 = tid % LPP, tx = tid / LPP;
    f32x4 wt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        wt[t].x = w[(c4 * 4 + 0) * 9 + t]; wt[t].y = w[(c4 * 4 + 1) * 9 + t];
        wt[t].z = w[(c4 * 4 + 2) * 9 + t]; wt[t].w = w[(c4 * 4 + 3) * 9 + t];
    }
    const float bv = bias ? bias[0] : 0.f;
    const int total = B * tiles_y * tiles_x;
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int n = tile / (tiles_y * tiles_x), tr = tile - n * tiles_y * tiles_x;
        const int y0 = (tr / tiles_x) * NC_TH, x0 = (tr % tiles_x) * PXB;
        const int nrows = min(NC_TH, H - y0) + 2;                        // input rows y0 - 1 .. y0 + nrows - 2 (reflected)
        const int sx = nc_reflect(x0 + tx, W);                           // (the column past the image's last is its reflection; further ones clamp)
        auto fetch = [&](int i) {
            const int sy = nc_reflect(y0 - 1 + i, H);
            return *reinterpret_cast<const f32x4*>(x + ((size_t)(n * H + sy) * W + sx) * C + c4 * 4);
        };
        f32x4 pre[NC_PF];
#pragma unroll
        for (int i = 0; i < NC_PF; ++i)
            if (i < nrows) pre[i] = fetch(i);
        for (int e = tid; e < nrows * 2 * LPP; e += 256) {               // halo columns x0 - 1 and x0 + PXB of every row of the tile
            const int i = e / (2 * LPP), side = (e / LPP) & 1, c = e % LPP;
            const int sy = nc_reflect(y0 - 1 + i, H), hx = nc_reflect(side ? x0 + PXB : x0 - 1, W);
            *reinterpret_cast<f32x4*>(&halo[i][side][c * 4]) = *reinterpret_cast<const f32x4*>(x + ((size_t)(n * H + sy) * W + hx) * C + c * 4);
        }
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;                               // partial sums of output rows i, i - 1, i - 2 (relative to y0 - 1)
#pragma unroll 1
        for (int i0 = 0; i0 < nrows; i0 += NC_PF)                         // (a rolled outer loop: fully unrolled, the compiler hoists every row's load)
#pragma unroll
        for (int j = 0; j < NC_PF; ++j) {
            const int i = i0 + j;
            if (i < nrows) {                                              // block-uniform
                float* buf = rows[j & 1];
                const f32x4 vm = pre[j];
                *reinterpret_cast<f32x4*>(buf + tid * 4) = vm;
                if (i + NC_PF < nrows) pre[j] = fetch(i + NC_PF);
                __syncthreads();
                const f32x4 vl = *reinterpret_cast<const f32x4*>(tx > 0 ? buf + (tid - LPP) * 4 : &halo[i][0][c4 * 4]);
                const f32x4 vr = *reinterpret_cast<const f32x4*>(tx < PXB - 1 ? buf + (tid + LPP) * 4 : &halo[i][1][c4 * 4]);
                auto dot3 = [&](int ky) {
                    const f32x4 q0 = wt[ky * 3], q1 = wt[ky * 3 + 1], q2 = wt[ky * 3 + 2];
                    return ((vl.x * q0.x + vl.y * q0.y) + (vl.z * q0.z + vl.w * q0.w)) + ((vm.x * q1.x + vm.y * q1.y) + (vm.z * q1.z + vm.w * q1.w)) +
                           ((vr.x * q2.x + vr.y * q2.y) + (vr.z * q2.z + vr.w * q2.w));
                };
                a2 = a1 + dot3(2);                                        // input row r is tap ky = 2 of output row r - 1, ky = 1 of r, ky = 0 of r + 1
                a1 = a0 + dot3(1);
                a0 = dot3(0);
                if (i >= 2) {
                    float acc = a2;
#pragma unroll
                    for (int off = 1; off < LPP; off <<= 1) acc += __shfl_xor(acc, off, 64);
                    const int oy = y0 + i - 2, ox = x0 + tx;
                    if (c4 == 0 && ox < W) y[(size_t)(n * H + oy) * W + ox] = nc_act_fwd(acc + bv, act);
                }
            }
        }
        __syncthreads();                                                  // the ring and the halo are reused by the next tile
    }
}

// slab row layout of the backward partials: [9 * C] filter gradients as (c * 9 + tap), then the bias gradient.
// The tile's dpre = dy * act'(y) (one row / column of halo, zero outside the image) is staged in LDS once; a lane reads the 3x3
// neighbourhood of its pixel from there.  G[ky][kx] = dpre[py + 1 - ky][px + 1 - kx], plus -- on the lines next to the border -- the border
// line's output whose padded tap reflected onto this pixel (py == 1: ky = 0 also collects row 0; py == H - 2: ky = 2 also row H - 1).
template <int LPP>
__global__ __launch_bounds__(256) void conv3x3r_c1_bwd_kernel(const float* __restrict__ x, int B, int H, int W, const float* __restrict__ w,
                                                              const float* __restrict__ dy, const float* __restrict__ yout, int act, int x_act,
                                                              const float* __restrict__ addend, float* __restrict__ dx, float* __restrict__ slab,
                                                              int tiles_x, int tiles_y) {
    constexpr int C = 4 * LPP, PXB = 256 / LPP, DW = PXB + 2;
    __shared__ float dpre[(NC_TH + 2) * DW];
    __shared__ float red[4][9 * C + 1];
    const int tid = threadIdx.x, c4 = tid % LPP, tx = tid / LPP, lane = tid & 63, wave = tid >> 6;
    f32x4 wt[9], dwv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        wt[t].x = w[(c4 * 4 + 0) * 9 + t]; wt[t].y = w[(c4 * 4 + 1) * 9 + t];
        wt[t].z = w[(c4 * 4 + 2) * 9 + t]; wt[t].w = w[(c4 * 4 + 3) * 9 + t];
        dwv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float db = 0.f;
    const int total = B * tiles_y * tiles_x;
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int n = tile / (tiles_y * tiles_x), tr = tile - n * tiles_y * tiles_x;
        const int y0 = (tr / tiles_x) * NC_TH, x0 = (tr % tiles_x) * PXB;
        const int nrows = min(NC_TH, H - y0);
        const int px = x0 + tx;
        const bool live = px < W;
        const int pxc = live ? px : W - 1;
        f32x4 prex[NC_PF], prea[NC_PF];
        auto fetch = [&](int i, f32x4& a, f32x4& b) {
            const size_t off = ((size_t)(n * H + y0 + i) * W + pxc) * C + c4 * 4;
            a = *reinterpret_cast<const f32x4*>(x + off);
            if (addend) b = *reinterpret_cast<const f32x4*>(addend + off);
        };
#pragma unroll
        for (int i = 0; i < NC_PF; ++i)
            if (i < nrows) fetch(i, prex[i], prea[i]);
        for (int e = tid; e < (nrows + 2) * DW; e += 256) {
            const int ry = e / DW, rx = e - ry * DW;
            const int yy = y0 - 1 + ry, xx = x0 - 1 + rx;
            float v = 0.f;
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
                const size_t o = (size_t)(n * H + yy) * W + xx;
                v = yout ? dy[o] * nc_act_bwd(yout[o], act) : dy[o];
            }
            dpre[e] = v;
        }
        __syncthreads();
        const bool colb = live && (px <= 1 || px >= W - 2);
#pragma unroll                                                            // (fully unrolled the compiler issues the rows' loads earlier still: 0.100 ms against 0.110 at 12x192x640)
        for (int i0 = 0; i0 < NC_TH; i0 += NC_PF)
#pragma unroll
        for (int j = 0; j < NC_PF; ++j) {
            const int i = i0 + j;
            if (i < nrows) {                                              // block-uniform
                const int py = y0 + i;
                const f32x4 xv = prex[j];
                f32x4 g = addend ? prea[j] : f32x4{0.f, 0.f, 0.f, 0.f};
                if (i + NC_PF < nrows) fetch(i + NC_PF, prex[j], prea[j]);
                float N[3][3];                                            // N[a][b] = dpre[py - 1 + a][px - 1 + b]
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) N[a][b] = dpre[(i + a) * DW + tx + b];
                float G[9];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) G[ky * 3 + kx] = N[2 - ky][2 - kx];
                const bool rowb = py <= 1 || py >= H - 2;
                if (rowb || __any(colb)) {
                    const bool yt = py == 1, yb = py == H - 2, xl = px == 1, xr = px == W - 2;      // extra sources: rows 0 / H - 1, columns 0 / W - 1
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        if (yt) G[0 * 3 + kx] += N[0][2 - kx];
                        if (yb) G[2 * 3 + kx] += N[2][2 - kx];
                    }
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        if (xl) G[ky * 3 + 0] += N[2 - ky][0];
                        if (xr) G[ky * 3 + 2] += N[2 - ky][2];
                    }
                    if (yt && xl) G[0] += N[0][0];
                    if (yt && xr) G[2] += N[0][2];
                    if (yb && xl) G[6] += N[2][0];
                    if (yb && xr) G[8] += N[2][2];
                }
                if (live) {
                    f32x4 gs = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        dwv[t] += xv * G[t];
                        gs += wt[t] * G[t];
                    }
                    if (c4 == 0) db += N[1][1];
                    gs.x *= nc_act_bwd(xv.x, x_act); gs.y *= nc_act_bwd(xv.y, x_act); gs.z *= nc_act_bwd(xv.z, x_act); gs.w *= nc_act_bwd(xv.w, x_act);
                    *reinterpret_cast<f32x4*>(dx + ((size_t)(n * H + py) * W + px) * C + c4 * 4) = gs + g;
                }
            }
        }
        __syncthreads();                                                  // dpre is rebuilt for the next tile
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

constexpr int NC_BWD_BLOCKS = 2048;      // slab rows = the most blocks a backward launch may use (more tiles: grid-stride)

inline bool nc_ok(int C) { return C == 16 || C == 32 || C == 64 || C == 128; }

}  // namespace mcav

using namespace mcav;

MCAV_EXPORT int mcav_conv3x3r_c1_fwd(const float* x, int B, int H, int W, int C, const float* w_oihw, const float* bias, int act, float* y,
                                     void* stream) {
    if (!x || !w_oihw || !y || B <= 0 || H < 2 || W < 2 || !nc_ok(C)) return MCAV_E_INVALID;
    const long npix = (long)B * H * W;
    if (npix >= (1L << 31)) return MCAV_E_INVALID;
    hipStream_t s = as_stream(stream);
    const int pxb = 256 / (C / 4), tiles_x = (W + pxb - 1) / pxb, tiles_y = (H + NC_TH - 1) / NC_TH;
    const long tiles = (long)B * tiles_y * tiles_x;
    const int blocks = (int)(tiles < 16384 ? tiles : 16384);
    switch (C) {
        case 16: timed_launch(conv3x3r_c1_fwd_kernel<4>, dim3(blocks), dim3(256), 0, s, x, B, H, W, w_oihw, bias, act, y, tiles_x, tiles_y); break;
        case 32: timed_launch(conv3x3r_c1_fwd_kernel<8>, dim3(blocks), dim3(256), 0, s, x, B, H, W, w_oihw, bias, act, y, tiles_x, tiles_y); break;
        case 64: timed_launch(conv3x3r_c1_fwd_kernel<16>, dim3(blocks), dim3(256), 0, s, x, B, H, W, w_oihw, bias, act, y, tiles_x, tiles_y); break;
        default: timed_launch(conv3x3r_c1_fwd_kernel<32>, dim3(blocks), dim3(256), 0, s, x, B, H, W, w_oihw, bias, act, y, tiles_x, tiles_y); break;
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
    const int pxb = 256 / (C / 4), tiles_x = (W + pxb - 1) / pxb, tiles_y = (H + NC_TH - 1) / NC_TH;
    const long tiles = (long)B * tiles_y * tiles_x;
    const int blocks = (int)(tiles < NC_BWD_BLOCKS ? tiles : NC_BWD_BLOCKS);
    float* slab = reinterpret_cast<float*>(workspace);
    switch (C) {
        case 16: timed_launch(conv3x3r_c1_bwd_kernel<4>, dim3(blocks), dim3(256), 0, s, x, B, H, W, w_oihw, dy, y, act, x_act, addend, dx, slab, tiles_x, tiles_y); break;
        case 32: timed_launch(conv3x3r_c1_bwd_kernel<8>, dim3(blocks), dim3(256), 0, s, x, B, H, W, w_oihw, dy, y, act, x_act, addend, dx, slab, tiles_x, tiles_y); break;
        case 64: timed_launch(conv3x3r_c1_bwd_kernel<16>, dim3(blocks), dim3(256), 0, s, x, B, H, W, w_oihw, dy, y, act, x_act, addend, dx, slab, tiles_x, tiles_y); break;
        default: timed_launch(conv3x3r_c1_bwd_kernel<32>, dim3(blocks), dim3(256), 0, s, x, B, H, W, w_oihw, dy, y, act, x_act, addend, dx, slab, tiles_x, tiles_y); break;
    }
    conv3x3r_c1_finalize_kernel<<<9 * C + 1, 256, 0, s>>>(slab, (int)blocks, C, dw_oihw, dbias, accumulate);
    return launch_status();
}
