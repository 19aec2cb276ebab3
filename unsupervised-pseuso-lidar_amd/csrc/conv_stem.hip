// The depth network's image stem on the fp32 MFMA: conv 7x7, stride 2, pad 3, 3 -> 64 channels (reference models/depth/resnet_dispnet.py:38,
// torchvision resnet conv1) on the NHWC4 image, forward and weight gradient.
//
// The general implicit-GEMM kernel runs this layer at 54 (forward) / 49 (weight gradient) TFLOP/s: K = 49 taps x 4 channels with one
// 16-byte gather per (pixel, tap) and the address arithmetic in the loop, on a datapath the fp32 MFMA shares with the VALU.  Here a
// workgroup owns an 8 x 32 tile of OUTPUT pixels and stages its 21 x 70 input patch ONCE in LDS, split into channel planes and, within a
// plane, into even and odd source columns: an output pixel x reads source column 2 x + kx, so for a fixed tap the 32 lanes of a fragment
// read 32 consecutive floats of one parity plane (conflict-free ds_read_b32), and the two lane halves of v_mfma_f32_32x32x2_f32 (k and
// k + 1) take the even / odd column of a tap PAIR (kx = 2 j, 2 j + 1; the 8th column of a filter row has zero weights).  The K order is
// (channel, ky, column pair): 3 x 7 x 4 = 84 k-steps (168 k for 147 real taps).  With the K loop fully unrolled every LDS address is a
// lane-constant base plus an immediate: the loop body is ds_read_b32 + MFMA only.  The 168 x 64 filter slice sits in LDS too
// (mcav_pack_stem_weights: [k][64], zero rows for the padding column).  Epilogue: the shared one (BatchNorm statistics per stacked pass).
#include "conv_shared.h"
#include "kernel_timer.h"

namespace mcav {

constexpr int ST_TH = 8, ST_TW = 32;              // output tile: rows x columns
constexpr int ST_PR = 2 * ST_TH + 5;              // 21 source rows
constexpr int ST_PC = 36;                         // floats per parity plane row: indices px + kxp <= 31 + 3, padded
constexpr int ST_KS = 3 * 7 * 4;                  // 84 k-steps (channel, ky, column pair)
constexpr int ST_N = 64;
using StemTile = Tile<256, 64, 64, 64, 32, 16>;   // for the shared epilogue: four wavefronts stacked along M, 2 x 2 accumulator tiles each

__global__ __launch_bounds__(256, 2) void stem7x7s2_fwd_kernel(IgemmParams p, const float* __restrict__ wk, int tiles_x, int tiles_y) {
    __shared__ __attribute__((aligned(16))) float sP[3][ST_PR][2][ST_PC];       // 18144 B
    __shared__ __attribute__((aligned(16))) float sW[2 * ST_KS][ST_N];          // 43008 B
    __shared__ int s_out[256];
    float (*const s_stat)[2][ST_N] = reinterpret_cast<float (*)[2][ST_N]>(&sP[0][0][0][0]);      // free once the K loop is done
    static_assert(sizeof(float) * 3 * ST_PR * 2 * ST_PC >= sizeof(float) * 4 * 2 * ST_N, "statistics scratch fits the patch");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per_img = tiles_x * tiles_y, ntiles = p.g.B * per_img;
    const GatherSrc& g = p.g;
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(g.x1, (unsigned)((size_t)g.B * g.Hs * g.Ws * 4 * 4));

    // ---- the filter slice [168][64], once per (persistent) workgroup; all its loads are issued before the first LDS store
    {
        constexpr int WN4 = (2 * ST_KS * ST_N / 4 + 255) / 256;
        f32x4 wv[WN4];
        const __amdgpu_buffer_rsrc_t rsw = make_rsrc(wk, 2 * ST_KS * ST_N * 4);
#pragma unroll
        for (int j = 0; j < WN4; ++j) wv[j] = buf_load4(rsw, (unsigned)((tid + 256 * j) * 16));      // past the slice: reads zero, not stored
#pragma unroll
        for (int j = 0; j < WN4; ++j)
            if (tid + 256 * j < 2 * ST_KS * ST_N / 4) reinterpret_cast<f32x4*>(&sW[0][0])[tid + 256 * j] = wv[j];
    }
    // ---- a tile's input patch: source rows 2 oy0 - 3 .., columns 2 ox0 - 3 ..; out-of-image pixels read as zero (zero padding).  The loads
    // of tile t + 1 are issued before the MFMA loop of tile t and stored when its readers are done.
    constexpr int PN = (ST_PR * 2 * ST_PC + 255) / 256;
    f32x4 pv[PN];
    auto issue = [&](int t) {
        const int b = t / per_img, tr = t - b * per_img;
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int sy0 = 2 * ty * ST_TH - 3, sx0 = 2 * tx * ST_TW - 3;
#pragma unroll
        for (int j = 0; j < PN; ++j) {
            const int i = tid + 256 * j;
            const int row = i / (2 * ST_PC), col = i - row * (2 * ST_PC);          // col = 2 * index + parity
            const int sy = sy0 + row, sx = sx0 + col;
            const bool ok = t < ntiles && i < ST_PR * 2 * ST_PC && (unsigned)sy < (unsigned)g.Hs && (unsigned)sx < (unsigned)g.Ws;
            pv[j] = buf_load4(rsx, ok ? (unsigned)(((b * g.Hs + sy) * g.Ws + sx) * 16) : OOB);
        }
    };
    const int px = lane & 31, h = lane >> 5;
    // lane-constant bases; everything else is an immediate after unrolling
    const float* pa = &sP[0][4 * wave][h][px];            // output row 2 wave (source row 4 wave + ky); row 2 wave + 1 is two source rows on
    const float* pb = &sW[h][px];

    issue(blockIdx.x);
    for (int mt = blockIdx.x; mt < ntiles; mt += gridDim.x) {
        const int b = mt / per_img, tr = mt - b * per_img;
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int oy0 = ty * ST_TH, ox0 = tx * ST_TW;
        __syncthreads();                                  // the previous tile's epilogue is done with s_out and the statistics scratch
        {   // destination pixel of every tile row (row = (2 wave + i) * 32 + px)
            const int r = tid, oy = oy0 + (r >> 5), ox = ox0 + (r & 31);
            s_out[r] = (oy < p.Hd && ox < p.Wd) ? (b * p.Hd + oy) * p.Wd + ox : -1;
        }
#pragma unroll
        for (int j = 0; j < PN; ++j) {
            const int i = tid + 256 * j;
            const int row = i / (2 * ST_PC), col = i - row * (2 * ST_PC);
            if (i < ST_PR * 2 * ST_PC) {
                sP[0][row][col & 1][col >> 1] = pv[j].x;
                sP[1][row][col & 1][col >> 1] = pv[j].y;
                sP[2][row][col & 1][col >> 1] = pv[j].z;
            }
        }
        __syncthreads();
        issue(mt + gridDim.x);                            // (past the last tile: out-of-range offsets, no memory traffic)

        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // operands of k-step s + 1 are read from LDS BEFORE the four MFMAs of step s are issued (a wavefront issues in order: a read placed
        // right in front of its use leaves the matrix pipe idle for the LDS latency, once per step); the scheduler is fenced so that it
        // keeps that order
        float fa0[2], fa1[2], fb0[2], fb1[2];
        auto rd = [&](auto sc) {
            constexpr int ks = decltype(sc)::value, kxp = ks & 3, cky = ks >> 2, ky = cky % 7, c = cky / 7, sl = ks & 1;
            fa0[sl] = pa[(c * ST_PR + ky) * 2 * ST_PC + kxp];
            fa1[sl] = pa[(c * ST_PR + ky + 2) * 2 * ST_PC + kxp];
            fb0[sl] = pb[ks * 2 * ST_N];
            fb1[sl] = pb[ks * 2 * ST_N + 32];
        };
        rd(std::integral_constant<int, 0>{});
        static_for<ST_KS>([&](auto sc) {
            constexpr int ks = decltype(sc)::value, sl = ks & 1;
            if constexpr (ks + 1 < ST_KS) rd(std::integral_constant<int, ks + 1>{});
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[sl], fb0[sl], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[sl], fb1[sl], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[sl], fb0[sl], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[sl], fb1[sl], acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
        __syncthreads();                                  // every wavefront is done with the patch: its memory becomes the statistics scratch
        igemm_epilogue<StemTile>(p, acc, s_out, s_stat, tid, wave * 64, 0, 0, mt);
    }
}

// OIHW [64][3][7][7] -> [168][64]: row ((c * 7 + ky) * 4 + j) * 2 + h holds w[:, c, ky, 2 j + h] (zero for kx = 7)
__global__ void pack_stem_weights_kernel(const float* __restrict__ w, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * ST_KS * ST_N) return;
    const int n = i % ST_N, k = i / ST_N;
    const int hh = k & 1, j = (k >> 1) & 3, cky = k >> 3, ky = cky % 7, c = cky / 7;
    const int kx = 2 * j + hh;
    out[i] = kx < 7 ? w[((n * 3 + c) * 7 + ky) * 7 + kx] : 0.f;
}

// ------------------------------------------------------------------------------------------------ weight gradient of the stem
// dW[k][n] = sum over output pixels of patch(pixel, k) * dy[pixel, n]: the reduction runs over pixels, two per v_mfma_f32_32x32x2_f32 (the lane
// halves take the two pixels of a horizontal pair), GEMM rows = the 168 k of the forward K order, columns = the 64 output channels.
// A persistent workgroup walks 4 x 32 output tiles: the tile's 13 x 70 input patch (same planes as the forward kernel) and its 128 x 64
// dy tile go to LDS; wavefront w owns output channels 32 (w & 1) .. + 31 and three of the six 32-row k tiles.  A lane's k row fixes a
// lane-constant LDS offset (channel plane, source row ky, parity plane, column pair) -- 32 consecutive k land on 32 different banks -- and the
// pixel adds an immediate, so the loop body is again ds_read_b32 + MFMA.  Each workgroup writes one slab partial in the layout the shared
// reduction expects (row = tap * 4 + channel).
constexpr int SW_TH = 4;                          // output rows per tile
constexpr int SW_PR = 2 * SW_TH + 5;              // 13 source rows

__global__ __launch_bounds__(256) void stem7x7s2_wgrad_kernel(const float* __restrict__ x4, const float* __restrict__ dy, int B, int Hs, int Ws,
                                                              int Hd, int Wd, int Cdy, int dy_choff, float* __restrict__ slab, int Ktot, int slabN) {
    __shared__ __attribute__((aligned(16))) float sP[3][SW_PR][2][ST_PC];       // 11232 B
    __shared__ __attribute__((aligned(16))) float sDY[SW_TH * ST_TW][ST_N];     // 32768 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (Wd + ST_TW - 1) / ST_TW, tiles_y = (Hd + SW_TH - 1) / SW_TH;
    const int ntiles = B * tiles_x * tiles_y;
    const int i = lane & 31, h = lane >> 5;
    const int nt = wave & 1, mt0 = 3 * (wave >> 1);
    // lane-constant A offsets (floats) of its k row in each of its three k tiles; k >= 168 reads offset 0 and is never stored
    int koff[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const int k = (mt0 + m) * 32 + i;
        const int par = k & 1, kxp = (k >> 1) & 3, cky = k >> 3, ky = cky % 7, c = cky / 7;
        koff[m] = k < 2 * ST_KS ? ((c * SW_PR + ky) * 2 + par) * ST_PC + kxp + h : 0;
    }
    const float* const pP = &sP[0][0][0][0];
    const float* const pD = &sDY[h][nt * 32 + i];
    f32x16 acc[3];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(x4, (unsigned)((size_t)B * Hs * Ws * 16));
    const __amdgpu_buffer_rsrc_t rsy = make_rsrc(dy, (unsigned)((size_t)B * Hd * Wd * Cdy * 4));

    // every load of a tile is issued before the first LDS store (one memory round trip per tile), and the loads of tile t + 1 are issued
    // BEFORE the MFMA loop of tile t: they fly while the matrix pipe works and are stored once the tile's readers are done
    constexpr int PN = (SW_PR * 2 * ST_PC + 255) / 256, DN = SW_TH * ST_TW * (ST_N / 4) / 256;
    f32x4 pv[PN], dv[DN];
    const int per_img = tiles_x * tiles_y;
    auto issue = [&](int t) {
        const int b = t / per_img, tr = t - b * per_img;
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int oy0 = ty * SW_TH, ox0 = tx * ST_TW;
        const int sy0 = 2 * oy0 - 3, sx0 = 2 * ox0 - 3;
        const bool live = t < ntiles;
#pragma unroll
        for (int j = 0; j < PN; ++j) {
            const int e = tid + 256 * j;
            const int row = e / (2 * ST_PC), col = e - row * (2 * ST_PC);
            const int sy = sy0 + row, sx = sx0 + col;
            const bool ok = live && e < SW_PR * 2 * ST_PC && (unsigned)sy < (unsigned)Hs && (unsigned)sx < (unsigned)Ws;
            pv[j] = buf_load4(rsx, ok ? (unsigned)(((b * Hs + sy) * Ws + sx) * 16) : OOB);
        }
#pragma unroll
        for (int j = 0; j < DN; ++j) {                                          // dy tile: pixels outside the image contribute zero
            const int e = tid + 256 * j;
            const int pix = e >> 4, c4 = e & 15;
            const int oy = oy0 + (pix >> 5), ox = ox0 + (pix & 31);
            const bool ok = live && oy < Hd && ox < Wd;
            dv[j] = buf_load4(rsy, ok ? (unsigned)((((b * Hd + oy) * Wd + ox) * Cdy + dy_choff + c4 * 4) * 4) : OOB);
        }
    };
    issue(blockIdx.x);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        __syncthreads();                                  // the previous tile's readers are done
#pragma unroll
        for (int j = 0; j < PN; ++j) {
            const int e = tid + 256 * j;
            const int row = e / (2 * ST_PC), col = e - row * (2 * ST_PC);
            if (e < SW_PR * 2 * ST_PC) {
                sP[0][row][col & 1][col >> 1] = pv[j].x;
                sP[1][row][col & 1][col >> 1] = pv[j].y;
                sP[2][row][col & 1][col >> 1] = pv[j].z;
            }
        }
#pragma unroll
        for (int j = 0; j < DN; ++j) {
            const int e = tid + 256 * j;
            *reinterpret_cast<f32x4*>(&sDY[e >> 4][(e & 15) * 4]) = dv[j];
        }
        __syncthreads();
        issue(t + gridDim.x);                             // (past the last tile: out-of-range offsets, no memory traffic)
        // (operands of pixel pair s + 1 are read before the three MFMAs of pair s are issued, as in the forward kernel)
        float fa[2][3], fb[2];
        auto rd = [&](auto sc) {
            constexpr int st = decltype(sc)::value, r = st / (ST_TW / 2), xp = st % (ST_TW / 2), sl = st & 1;
            constexpr int po = (2 * r) * 2 * ST_PC + 2 * xp;                   // pixel (r, 2 xp + h): source row + 2 r, index + 2 xp (+ h in koff)
            fb[sl] = pD[(r * ST_TW + 2 * xp) * ST_N];
#pragma unroll
            for (int m = 0; m < 3; ++m) fa[sl][m] = pP[koff[m] + po];
        };
        constexpr int NST = SW_TH * ST_TW / 2;
        rd(std::integral_constant<int, 0>{});
        static_for<NST>([&](auto sc) {
            constexpr int st = decltype(sc)::value, sl = st & 1;
            if constexpr (st + 1 < NST) rd(std::integral_constant<int, st + 1>{});
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 3; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sl][m], fb[sl], acc[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    // ---- one slab partial per workgroup: row = tap * 4 + channel, column = output channel
    float* out = slab + (size_t)blockIdx.x * (Ktot + 1) * slabN;
    const int n = nt * 32 + i;
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = (mt0 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int par = k & 1, kxp = (k >> 1) & 3, cky = k >> 3, ky = cky % 7, c = cky / 7, kx = 2 * kxp + par;
            if (k < 2 * ST_KS && kx < 7) out[(size_t)((ky * 7 + kx) * 4 + c) * slabN + n] = acc[m][r];
        }
}

static bool stem_ok(const mcav_igemm_desc* d) {
    return d && d->mode == MCAV_G_SMALLC && d->kh == 7 && d->kw == 7 && d->stride == 2 && d->sign == 1 && d->offset == -3 && d->C1 == 4 &&
           d->C2 == 0 && d->n_begin == 0 && d->n_count == ST_N && d->Np == ST_N && d->w_stem && !d->pool && !d->dact_aux && !d->addend &&
           d->y_choff == 0 && !((d->tile >> 9) & 1) && d->Hd == (d->Hs - 1) / 2 + 1 && d->Wd == (d->Ws - 1) / 2 + 1;
}

}  // namespace mcav

using namespace mcav;

int mcav_stem_mtiles(const mcav_igemm_desc* d) {      // 0 = not this kernel's launch
    if (!stem_ok(d)) return 0;
    return d->B * ((d->Hd + ST_TH - 1) / ST_TH) * ((d->Wd + ST_TW - 1) / ST_TW);
}

bool mcav_try_stem(const mcav_igemm_desc* d, const IgemmParams& p, hipStream_t s) {
    if (!stem_ok(d)) return false;
    const int tiles_x = (d->Wd + ST_TW - 1) / ST_TW, tiles_y = (d->Hd + ST_TH - 1) / ST_TH;
    IgemmParams q = p;
    q.groups = 1;                                         // (the statistics rows of a tile are image-major: groups need nothing else)
    const int ntiles = d->B * tiles_x * tiles_y;
    timed_launch(stem7x7s2_fwd_kernel, dim3(ntiles < 512 ? ntiles : 512), dim3(256), 0, s, q, d->w_stem, tiles_x, tiles_y);      // persistent: 2 per CU
    return true;
}

MCAV_EXPORT int mcav_pack_stem_weights(const float* w_oihw, float* packed168x64, void* stream) {
    if (!w_oihw || !packed168x64) return MCAV_E_INVALID;
    pack_stem_weights_kernel<<<(2 * ST_KS * ST_N + 255) / 256, 256, 0, as_stream(stream)>>>(w_oihw, packed168x64);
    return launch_status();
}

// ---- weight gradient hooks (called by the planner / launcher of conv_igemm.hip, as the halo kernels' are)
int mcav_stem_wgrad_splits(const mcav_wgrad_desc* d) {      // 0 = not applicable, else the number of slab partials (= persistent workgroups)
    if (!d || d->mode != MCAV_G_SMALLC || d->kh != 7 || d->kw != 7 || d->stride != 2 || d->sign != 1 || d->offset != -3) return 0;
    if (d->C1 != 4 || d->C2 != 0 || d->Kp != 4 || d->Cout != ST_N || d->Cin != 3 || d->dbias || d->upm || ((d->tile >> 9) & 1)) return 0;
    if (d->Hd != (d->Hs - 1) / 2 + 1 || d->Wd != (d->Ws - 1) / 2 + 1 || (d->Cdy & 3) || (d->dy_choff & 3) || d->Cdy - d->dy_choff < ST_N) return 0;
    const long tiles = (long)d->B * ((d->Hd + SW_TH - 1) / SW_TH) * ((d->Wd + ST_TW - 1) / ST_TW);
    return (int)(tiles < 512 ? tiles : 512);
}

void mcav_stem_wgrad_launch(const mcav_wgrad_desc* d, float* slab, int Ktot, int slabN, int splits, hipStream_t s) {
    timed_launch(stem7x7s2_wgrad_kernel, dim3(splits), dim3(256), 0, s, d->x1, d->dy, d->B, d->Hs, d->Ws, d->Hd, d->Wd, d->Cdy, d->dy_choff, slab,
                 Ktot, slabN);
}
