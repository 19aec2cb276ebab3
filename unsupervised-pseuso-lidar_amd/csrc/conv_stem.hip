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

constexpr int ST_TW = 32;                         // output tile width (one MFMA row block)
constexpr int ST_PC = 36;                         // floats per parity plane row: indices px + kxp <= 31 + 3, padded

// C real input channels stored in pixels of CP floats; N output channels (64: two 32-wide column tiles per wavefront; 16: one, the upper
// 16 lanes duplicate the lower ones and are masked by the epilogue); TH output rows per tile (a wavefront owns TH / 4 of them).
template <int C_, int CP_, int N_, int TH_>
struct StemCfg {
    static constexpr int C = C_, CP = CP_, N = N_, TH = TH_;
    static constexpr int PR = 2 * TH + 5;                 // source rows of the patch
    static constexpr int KS = C * 7 * 4;                  // k-steps (channel, ky, column pair)
    static constexpr int TM = TH / 4, TN = N >= 32 ? N / 32 : 1;
    static constexpr int NW = N;                          // floats per filter-slice row
    static constexpr int LD4 = (C + 3) / 4;               // 16-byte loads per source pixel
    using Epi = Tile<TH * 32, (N >= 32 ? N : 32), TM * 32, (N >= 32 ? N : 32), 32, 16>;      // for the shared epilogue: 4 wavefronts stacked along M
};
using StemDepth = StemCfg<3, 4, 64, 8>;           // torchvision conv1 on the NHWC4 image
using StemPose = StemCfg<9, 16, 16, 4>;           // PoseNet conv1 on the 16-channel (9 real) input pack

template <class S>
__global__ __launch_bounds__(256, 2) void stem7x7s2_fwd_kernel(IgemmParams p, const float* __restrict__ wk, int tiles_x, int tiles_y) {
    constexpr int C = S::C, PR = S::PR, KS = S::KS, TM = S::TM, TN = S::TN, NW = S::NW, TH = S::TH;
    __shared__ __attribute__((aligned(16))) float sP[C][PR][2][ST_PC];
    __shared__ __attribute__((aligned(16))) float sW[2 * KS][NW];
    __shared__ __attribute__((aligned(16))) unsigned s_out[TH * 32];      // byte offset of each tile row's output pixel (OOB: none): the lean epilogue's form
    using E = typename S::Epi;
    float (*const s_stat)[2][E::BN] = reinterpret_cast<float (*)[2][E::BN]>(&sP[0][0][0][0]);      // free once the K loop is done
    static_assert(sizeof(float) * C * PR * 2 * ST_PC >= sizeof(float) * 4 * 2 * E::BN, "statistics scratch fits the patch");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per_img = tiles_x * tiles_y, ntiles = p.g.B * per_img;
    const GatherSrc& g = p.g;
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(g.x1, (unsigned)((size_t)g.B * g.Hs * g.Ws * S::CP * 4));

    // ---- the filter slice [2 KS][NW], once per (persistent) workgroup; all its loads are issued before the first LDS store
    {
        constexpr int W4 = 2 * KS * NW / 4, WN4 = (W4 + 255) / 256;
        f32x4 wv[WN4];
        const __amdgpu_buffer_rsrc_t rsw = make_rsrc(wk, 2 * KS * NW * 4);
#pragma unroll
        for (int j = 0; j < WN4; ++j) wv[j] = buf_load4(rsw, (unsigned)((tid + 256 * j) * 16));      // past the slice: reads zero, not stored
#pragma unroll
        for (int j = 0; j < WN4; ++j)
            if (tid + 256 * j < W4) reinterpret_cast<f32x4*>(&sW[0][0])[tid + 256 * j] = wv[j];
    }
    // ---- a tile's input patch: source rows 2 oy0 - 3 .., columns 2 ox0 - 3 ..; out-of-image pixels read as zero (zero padding).  The loads
    // of tile t + 1 are issued before the MFMA loop of tile t and stored when its readers are done.
    constexpr int PN = (PR * 2 * ST_PC + 255) / 256;
    f32x4 pv[PN][S::LD4];
    auto issue = [&](int t) {
        const int b = t / per_img, tr = t - b * per_img;
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int sy0 = 2 * ty * TH - 3, sx0 = 2 * tx * ST_TW - 3;
#pragma unroll
        for (int j = 0; j < PN; ++j) {
            const int i = tid + 256 * j;
            const int row = i / (2 * ST_PC), col = i - row * (2 * ST_PC);          // col = 2 * index + parity
            const int sy = sy0 + row, sx = sx0 + col;
            const bool ok = t < ntiles && i < PR * 2 * ST_PC && (unsigned)sy < (unsigned)g.Hs && (unsigned)sx < (unsigned)g.Ws;
            const unsigned off = ok ? (unsigned)(((b * g.Hs + sy) * g.Ws + sx) * S::CP * 4) : OOB;
#pragma unroll
            for (int q = 0; q < S::LD4; ++q) pv[j][q] = buf_load4(rsx, off == OOB ? OOB : off + 16u * q);
        }
    };
    const int px = lane & 31, h = lane >> 5;
    // lane-constant bases; everything else is an immediate after unrolling
    const float* pa = &sP[0][2 * TM * wave][h][px];       // this wavefront's first output row TM * wave: source row 2 TM wave + ky
    const float* pb = &sW[h][(NW >= 32 ? px : px & (NW - 1))];

    issue(blockIdx.x);
    for (int mt = blockIdx.x; mt < ntiles; mt += gridDim.x) {
        const int b = mt / per_img, tr = mt - b * per_img;
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int oy0 = ty * TH, ox0 = tx * ST_TW;
        __syncthreads();                                  // the previous tile's epilogue is done with s_out and the statistics scratch
        if (tid < TH * 32) {   // destination pixel of every tile row (row = (TM wave + i) * 32 + px)
            const int r = tid, oy = oy0 + (r >> 5), ox = ox0 + (r & 31);
            s_out[r] = (oy < p.Hd && ox < p.Wd) ? (unsigned)((b * p.Hd + oy) * p.Wd + ox) * (unsigned)(p.Cd * 4) : OOB;
        }
#pragma unroll
        for (int j = 0; j < PN; ++j) {
            const int i = tid + 256 * j;
            const int row = i / (2 * ST_PC), col = i - row * (2 * ST_PC);
            if (i < PR * 2 * ST_PC) {
#pragma unroll
                for (int c = 0; c < C; ++c) sP[c][row][col & 1][col >> 1] = pv[j][c >> 2][c & 3];
            }
        }
        __syncthreads();
        issue(mt + gridDim.x);                            // (past the last tile: out-of-range offsets, no memory traffic)

        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // operands of k-step s + 1 are read from LDS BEFORE the MFMAs of step s are issued (a wavefront issues in order: a read placed
        // right in front of its use leaves the matrix pipe idle for the LDS latency, once per step); the scheduler is fenced so that it
        // keeps that order
        float fa[2][TM], fb[2][TN];
        auto rd = [&](auto sc) {
            constexpr int ks = decltype(sc)::value, kxp = ks & 3, cky = ks >> 2, ky = cky % 7, c = cky / 7, sl = ks & 1;
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[sl][i] = pa[(c * PR + ky + 2 * i) * 2 * ST_PC + kxp];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[sl][j] = pb[ks * 2 * NW + 32 * j];
        };
        rd(std::integral_constant<int, 0>{});
        static_for<KS>([&](auto sc) {
            constexpr int ks = decltype(sc)::value, sl = ks & 1;
            if constexpr (ks + 1 < KS) rd(std::integral_constant<int, ks + 1>{});
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sl][i], fb[sl][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
        __syncthreads();                                  // every wavefront is done with the patch: its memory becomes the statistics scratch
        // (the lean epilogue since round 4: raw buffer stores, no per-element branch or 64-bit address arithmetic -- the stem kernels were bound by
        // their epilogue, not by the MFMAs: 0.18 of the depth stem's 0.23 ms remained with the K loop removed)
        igemm_epilogue_lean<E>(p, acc, s_out, s_stat, tid, wave * TM * 32, 0, 0, mt);
    }
}

// ------------------------------------------------------------------------------------------------ the depth stem as an fp32 contraction on split operands (round 4)
// The fp32 kernel above needs 84 v_mfma_f32_32x32x2_f32 per 32 x 32 tile (100 us of MFMA issue for the step's 24 images) and runs at 232 us.  The
// same contraction on three bf16 planes per operand (six plane products, fp32 accumulation; mcav_igemm_desc.mma >= 2) is 14
// k-steps of v_mfma_f32_32x32x16_bf16: with the K order (ky, kx 0..7, channel 0..3) a lane's 8 consecutive k are two source pixels x four
// channels = 16 contiguous bytes of the NHWC4 image row, and because the stride is 2 and a pixel is 4 channels, output pixel px starts
// 16 px bytes into the row: every A fragment is ONE aligned ds_read_b128 of the staged patch (no im2col), conflict-free across the 32
// pixels of a fragment.  The filter slice (64 outputs x 224 k x 3 planes, 87 KB) is split once per persistent workgroup from the packed
// fp32 copy the fp32 kernel uses.  One workgroup per CU (125 KB of LDS), four wavefronts of 2 x 2 tiles, ONE accumulator per tile.
typedef __bf16 st_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 st_bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short st_u16;
typedef unsigned int st_u32x2 __attribute__((ext_vector_type(2)));
constexpr int SS_TH = 8, SS_PR = 2 * SS_TH + 5, SS_PCOL = 72, SS_PROW = SS_PCOL * 4;      // patch: 21 rows x 72 pixels x 4 channels (288 bf16 per row)
constexpr int SS_K = 224, SS_WROW = 232;                      // filter row of an output channel, padded: rows 116 dwords apart (conflict-free ds_read_b128)
constexpr int SS_PPL = SS_PR * SS_PROW, SS_WPL = 64 * SS_WROW;      // plane strides (elements)
#ifndef MCAV_SS_DIAG
#define MCAV_SS_DIAG 0                                    // timing experiments only, WRONG results: 1 no MFMAs, 2 no epilogue
#endif
constexpr int SS_DIAG = MCAV_SS_DIAG;
constexpr size_t stem_split_lds_bytes() { return sizeof(st_u16) * 3 * (size_t)(SS_PPL + SS_WPL) + sizeof(unsigned) * SS_TH * 32; }

__device__ __forceinline__ void st_split3(f32x4 v, st_u32x2& h, st_u32x2& m, st_u32x2& l) {
    const st_bf16x4 hb = __builtin_convertvector(v, st_bf16x4);
    const f32x4 r1 = v - __builtin_convertvector(hb, f32x4);                  // exact
    const st_bf16x4 mb = __builtin_convertvector(r1, st_bf16x4);
    const f32x4 r2 = r1 - __builtin_convertvector(mb, f32x4);                 // exact
    const st_bf16x4 lb = __builtin_convertvector(r2, st_bf16x4);
    h = __builtin_bit_cast(st_u32x2, hb); m = __builtin_bit_cast(st_u32x2, mb); l = __builtin_bit_cast(st_u32x2, lb);
}

__global__ __launch_bounds__(256, 1) void stem7x7s2_split_fwd_kernel(IgemmParams p, const float* __restrict__ wk, int tiles_x, int tiles_y) {
    using S = StemDepth;
    using E = typename S::Epi;
    constexpr int TM = S::TM, TN = S::TN, TH = S::TH;
    static_assert(TH == SS_TH && TM == 2 && TN == 2, "8 output rows per tile, 2 x 2 tiles per wavefront");
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    st_u16* const sP = reinterpret_cast<st_u16*>(s_raw);                       // [3][SS_PR][SS_PROW]
    st_u16* const sW = sP + 3 * SS_PPL;                                        // [3][64][SS_WROW]
    unsigned* const s_out = reinterpret_cast<unsigned*>(sW + 3 * SS_WPL);     // [TH * 32] byte offset of each tile row's output pixel (OOB: none)
    float (*const s_stat)[2][E::BN] = reinterpret_cast<float (*)[2][E::BN]>(sP);      // free once the K loop is done

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per_img = tiles_x * tiles_y, ntiles = p.g.B * per_img;
    const GatherSrc& g = p.g;
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(g.x1, (unsigned)((size_t)g.B * g.Hs * g.Ws * S::CP * 4));

    // ---- the filter slice, once per (persistent) workgroup: packed fp32 copy [(c * 7 + ky) * 8 + kx][64] -> three planes [64][(ky, kx, c)]
    constexpr int WN = 64 * SS_K / 256;                   // 56 elements per thread: all loads in flight before the first use
    float wv[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int e = tid + 256 * j, co = e & 63, k = e >> 6, ky = k >> 5, kx = (k >> 2) & 7, c = k & 3;
        wv[j] = c < 3 ? wk[((c * 7 + ky) * 8 + kx) * 64 + co] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int e = tid + 256 * j, co = e & 63, k = e >> 6;
        const float v = wv[j];
        const __bf16 hb = (__bf16)v;
        const float r1 = v - (float)hb;
        const __bf16 mb = (__bf16)r1;
        const __bf16 lb = (__bf16)(r1 - (float)mb);
        sW[co * SS_WROW + k] = __builtin_bit_cast(st_u16, hb);
        sW[SS_WPL + co * SS_WROW + k] = __builtin_bit_cast(st_u16, mb);
        sW[2 * SS_WPL + co * SS_WROW + k] = __builtin_bit_cast(st_u16, lb);
    }
    // ---- a tile's input patch: source rows 2 oy0 - 3 .., pixels 2 ox0 - 3 ..; out-of-image pixels read as zero (zero padding)
    constexpr int PN = (SS_PR * SS_PCOL + 255) / 256;
    f32x4 pv[PN];
    auto issue = [&](int t) {
        const int b = t / per_img, tr = t - b * per_img;
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int sy0 = 2 * ty * TH - 3, sx0 = 2 * tx * ST_TW - 3;
#pragma unroll
        for (int j = 0; j < PN; ++j) {
            const int i = tid + 256 * j;
            const int row = i / SS_PCOL, col = i - row * SS_PCOL;
            const int sy = sy0 + row, sx = sx0 + col;
            const bool ok = t < ntiles && i < SS_PR * SS_PCOL && (unsigned)sy < (unsigned)g.Hs && (unsigned)sx < (unsigned)g.Ws;
            pv[j] = buf_load4(rsx, ok ? (unsigned)(((b * g.Hs + sy) * g.Ws + sx) * S::CP * 4) : OOB);
        }
    };
    const int px = lane & 31, h = lane >> 5;
    // lane-constant bases (elements): output row 2 wave + i reads patch row 2 (2 wave + i) + ky; pixel px starts 8 px elements into the row
    const st_u16* const pa = sP + (4 * wave) * SS_PROW + 8 * px + 8 * h;
    const st_u16* const pb = sW + px * SS_WROW + 8 * h;

    issue(blockIdx.x);
    for (int mt = blockIdx.x; mt < ntiles; mt += gridDim.x) {
        const int b = mt / per_img, tr = mt - b * per_img;
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int oy0 = ty * TH, ox0 = tx * ST_TW;
        __syncthreads();                                  // the previous tile's epilogue is done with s_out and the statistics scratch (and the filter slice is stored)
        if (tid < TH * 32) {
            const int r = tid, oy = oy0 + (r >> 5), ox = ox0 + (r & 31);
            s_out[r] = (oy < p.Hd && ox < p.Wd) ? (unsigned)((b * p.Hd + oy) * p.Wd + ox) * (unsigned)(p.Cd * 4) : OOB;
        }
#pragma unroll
        for (int j = 0; j < PN; ++j) {
            const int i = tid + 256 * j;
            if (i < SS_PR * SS_PCOL) {
                st_u32x2 hh, mm, ll;
                st_split3(pv[j], hh, mm, ll);
                *reinterpret_cast<st_u32x2*>(sP + i * 4) = hh;
                *reinterpret_cast<st_u32x2*>(sP + SS_PPL + i * 4) = mm;
                *reinterpret_cast<st_u32x2*>(sP + 2 * SS_PPL + i * 4) = ll;
            }
        }
        __syncthreads();
        issue(mt + gridDim.x);                            // (past the last tile: out-of-range offsets, no memory traffic)

        f32x16 acc[TM][TN], mid[TM][TN], low[TM][TN];     // (one wavefront per SIMD: the register file has room for the small terms' own accumulators)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; mid[i][j][r] = 0.f; low[i][j][r] = 0.f; }
        // 14 k-steps (ky, half of the row's eight pixels); fragments of step s + 1 are read before the MFMAs of step s are issued
        st_bf16x8 fa[2][TM][3], fb[2][TN][3];
        auto rd = [&](auto sc) {
            constexpr int ks = decltype(sc)::value, ky = ks >> 1, kh = ks & 1, sl = ks & 1;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int q = 0; q < 3; ++q) fa[sl][i][q] = *reinterpret_cast<const st_bf16x8*>(pa + q * SS_PPL + (2 * i + ky) * SS_PROW + 16 * kh);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 3; ++q) fb[sl][j][q] = *reinterpret_cast<const st_bf16x8*>(pb + q * SS_WPL + 32 * j * SS_WROW + 32 * ky + 16 * kh);
        };
        rd(std::integral_constant<int, 0>{});
        static_for<14>([&](auto sc) {
            constexpr int ks = decltype(sc)::value, sl = ks & 1;
            if constexpr (ks + 1 < 14) rd(std::integral_constant<int, ks + 1>{});
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const st_bf16x8(&a)[3] = fa[sl][i];
                    const st_bf16x8(&bq)[3] = fb[sl][j];
                    if (SS_DIAG & 1) continue;
                    low[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], bq[0], low[i][j], 0, 0, 0);
                    mid[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bq[0], mid[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bq[0], acc[i][j], 0, 0, 0);
                    low[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bq[2], low[i][j], 0, 0, 0);
                    mid[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bq[1], mid[i][j], 0, 0, 0);
                    low[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bq[1], low[i][j], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] += mid[i][j] + low[i][j];
        __syncthreads();                                  // every wavefront is done with the patch: its memory becomes the statistics scratch
        if (!(SS_DIAG & 2)) igemm_epilogue_lean<E>(p, acc, s_out, s_stat, tid, wave * TM * 32, 0, 0, mt);
    }
}

// OIHW [N][C][7][7] -> [2 KS][N]: row ((c * 7 + ky) * 4 + j) * 2 + h holds w[:, c, ky, 2 j + h] (zero for kx = 7)
__global__ void pack_stem_weights_kernel(const float* __restrict__ w, float* __restrict__ out, int C, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * 56 * N) return;
    const int n = i % N, k = i / N;
    const int hh = k & 1, j = (k >> 1) & 3, cky = k >> 3, ky = cky % 7, c = cky / 7;
    const int kx = 2 * j + hh;
    out[i] = kx < 7 ? w[((n * C + c) * 7 + ky) * 7 + kx] : 0.f;
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

// C / CP / N as in the forward kernel; KP = the filter's K padding in the slab layout (row = tap * KP + channel).  The 2 KS k rows are
// split over the wavefronts as MT tiles of 32 each; N = 64: wavefront w takes output channels 32 (w & 1) .. and half of the k tiles;
// N = 16: every wavefront takes all 16 (lanes 16-31 duplicate and are dropped) and a quarter of the k tiles.
template <int C, int CP, int N, int KP>
__global__ __launch_bounds__(256) void stem7x7s2_wgrad_kernel(const float* __restrict__ x4, const float* __restrict__ dy, int B, int Hs, int Ws,
                                                              int Hd, int Wd, int Cdy, int dy_choff, float* __restrict__ slab, int Ktot, int slabN,
                                                              int want_bias) {
    constexpr int KS = C * 7 * 4, KT = (2 * KS + 31) / 32;                      // k rows, 32-row k tiles
    constexpr int NSPLIT = N >= 64 ? 2 : 1, WPN = 4 / NSPLIT, MT = (KT + WPN - 1) / WPN;      // wavefronts per column group, k tiles per wavefront
    constexpr int LD4 = (C + 3) / 4, N4 = N / 4;
    __shared__ __attribute__((aligned(16))) float sP[C][SW_PR][2][ST_PC];
    __shared__ __attribute__((aligned(16))) float sDY[SW_TH * ST_TW][N];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (Wd + ST_TW - 1) / ST_TW, tiles_y = (Hd + SW_TH - 1) / SW_TH;
    const int ntiles = B * tiles_x * tiles_y;
    const int i = lane & 31, h = lane >> 5;
    const int nt = NSPLIT == 2 ? wave & 1 : 0, mt0 = MT * (NSPLIT == 2 ? wave >> 1 : wave);
    // lane-constant A offsets (floats) of its k row in each of its k tiles; k >= 2 KS reads offset 0 and is never stored
    int koff[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int k = (mt0 + m) * 32 + i;
        const int par = k & 1, kxp = (k >> 1) & 3, cky = k >> 3, ky = cky % 7, c = cky / 7;
        koff[m] = k < 2 * KS ? ((c * SW_PR + ky) * 2 + par) * ST_PC + kxp + h : 0;
    }
    const float* const pP = &sP[0][0][0][0];
    const float* const pD = &sDY[h][nt * 32 + (N >= 32 ? i : i & (N - 1))];
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(x4, (unsigned)((size_t)B * Hs * Ws * CP * 4));
    const __amdgpu_buffer_rsrc_t rsy = make_rsrc(dy, (unsigned)((size_t)B * Hd * Wd * Cdy * 4));

    // every load of a tile is issued before the first LDS store (one memory round trip per tile), and the loads of tile t + 1 are issued
    // BEFORE the MFMA loop of tile t: they fly while the matrix pipe works and are stored once the tile's readers are done
    constexpr int PN = (SW_PR * 2 * ST_PC + 255) / 256, DN = (SW_TH * ST_TW * N4 + 255) / 256;
    f32x4 pv[PN][LD4], dv[DN];
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};                   // bias gradient: this thread's 4 channels (tid % N4 is the same in every pass: 256 % N4 == 0)
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
            const unsigned off = ok ? (unsigned)(((b * Hs + sy) * Ws + sx) * CP * 4) : OOB;
#pragma unroll
            for (int q = 0; q < LD4; ++q) pv[j][q] = buf_load4(rsx, off == OOB ? OOB : off + 16u * q);
        }
#pragma unroll
        for (int j = 0; j < DN; ++j) {                                          // dy tile: pixels outside the image contribute zero
            const int e = tid + 256 * j;
            const int pix = e / N4, c4 = e - pix * N4;
            const int oy = oy0 + (pix >> 5), ox = ox0 + (pix & 31);
            const bool ok = live && e < SW_TH * ST_TW * N4 && oy < Hd && ox < Wd;
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
#pragma unroll
                for (int c = 0; c < C; ++c) sP[c][row][col & 1][col >> 1] = pv[j][c >> 2][c & 3];
            }
        }
#pragma unroll
        for (int j = 0; j < DN; ++j) {
            const int e = tid + 256 * j;
            if (e < SW_TH * ST_TW * N4) {
                *reinterpret_cast<f32x4*>(&sDY[e / N4][(e % N4) * 4]) = dv[j];
                bsum += dv[j];
            }
        }
        __syncthreads();
        issue(t + gridDim.x);                             // (past the last tile: out-of-range offsets, no memory traffic)
        // (operands of pixel pair s + 1 are read before the MFMAs of pair s are issued, as in the forward kernel)
        float fa[2][MT], fb[2];
        auto rd = [&](auto sc) {
            constexpr int st = decltype(sc)::value, r = st / (ST_TW / 2), xp = st % (ST_TW / 2), sl = st & 1;
            constexpr int po = (2 * r) * 2 * ST_PC + 2 * xp;                   // pixel (r, 2 xp + h): source row + 2 r, index + 2 xp (+ h in koff)
            fb[sl] = pD[(r * ST_TW + 2 * xp) * N];
#pragma unroll
            for (int m = 0; m < MT; ++m) fa[sl][m] = pP[koff[m] + po];
        };
        constexpr int NST = SW_TH * ST_TW / 2;
        rd(std::integral_constant<int, 0>{});
        static_for<NST>([&](auto sc) {
            constexpr int st = decltype(sc)::value, sl = st & 1;
            if constexpr (st + 1 < NST) rd(std::integral_constant<int, st + 1>{});
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sl][m], fb[sl], acc[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    // ---- one slab partial per workgroup: row = tap * KP + channel, column = output channel; row Ktot = column sums of dy (bias gradient)
    float* out = slab + (size_t)blockIdx.x * (Ktot + 1) * slabN;
    const int n = nt * 32 + i;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = (mt0 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int par = k & 1, kxp = (k >> 1) & 3, cky = k >> 3, ky = cky % 7, c = cky / 7, kx = 2 * kxp + par;
            if (k < 2 * KS && kx < 7 && n < N) out[(size_t)((ky * 7 + kx) * KP + c) * slabN + n] = acc[m][r];
        }
    if (want_bias) {
        __syncthreads();
        float (*red)[4] = reinterpret_cast<float (*)[4]>(&sDY[0][0]);          // [256][4]
        *reinterpret_cast<f32x4*>(&red[tid][0]) = bsum;
        __syncthreads();
        if (tid < N) {                                    // column tid = group tid / 4, element tid % 4: threads g, g + N4, g + 2 N4, ... hold it
            float t = 0.f;
            for (int q = tid >> 2; q < 256; q += N4) t += red[q][tid & 3];
            out[(size_t)Ktot * slabN + tid] = t;
        }
    }
}

// which stem this forward launch is: 1 = the depth net's image stem (SMALLC, 3 -> 64), 2 = PoseNet conv1 (9 of 16 channels -> 16), 0 = neither
static int stem_kind(const mcav_igemm_desc* d) {
    if (!d || !d->w_stem || d->kh != 7 || d->kw != 7 || d->stride != 2 || d->sign != 1 || d->offset != -3 || d->C2 != 0 || d->n_begin != 0 ||
        d->pool || d->dact_aux || d->addend || d->y_choff != 0 || ((d->tile >> 9) & 1) || d->pad_mode != MCAV_PAD_ZERO ||
        d->Hd != (d->Hs - 1) / 2 + 1 || d->Wd != (d->Ws - 1) / 2 + 1)
        return 0;
    if (d->mode == MCAV_G_SMALLC && d->C1 == 4 && d->n_count == 64 && d->Np == 64) return 1;
    if (d->mode == MCAV_G_DIRECT && d->C1 == 16 && d->Kp == 16 && d->n_count == 16 && d->Np == 16 && !d->stats && !d->up1) return 2;
    return 0;
}

}  // namespace mcav

using namespace mcav;

int mcav_stem_mtiles(const mcav_igemm_desc* d) {      // 0 = not this kernel's launch
    const int kind = stem_kind(d);
    if (!kind) return 0;
    const int th = kind == 1 ? StemDepth::TH : StemPose::TH;
    return d->B * ((d->Hd + th - 1) / th) * ((d->Wd + ST_TW - 1) / ST_TW);
}

bool mcav_try_stem(const mcav_igemm_desc* d, const IgemmParams& p, hipStream_t s) {
    const int kind = stem_kind(d);
    if (!kind) return false;
    const int th = kind == 1 ? StemDepth::TH : StemPose::TH;
    const int tiles_x = (d->Wd + ST_TW - 1) / ST_TW, tiles_y = (d->Hd + th - 1) / th;
    IgemmParams q = p;
    q.groups = 1;                                         // (the statistics rows of a tile are image-major: groups need nothing else)
    const int ntiles = d->B * tiles_x * tiles_y;
    // The split form of the depth stem: measured 0.189 ms against the fp32 kernel's 0.179 (both with the lean epilogue; 24 x 192 x 640): the stem is
    // bound by its patch staging and its 189 MB of output stores, which the split kernel's ONE workgroup per CU (125 KB of LDS) has nothing to
    // overlap with, while its MFMAs take 0.06 instead of 0.10 ms.  OFF under the default mode (mma = 2); mma = 3 (the parity tests) runs it.
    static const int split_on = MCAV_KNOB_INT("MCAV_STEM_SPLIT", 0);
    if (kind == 1 && (d->mma == 3 || (d->mma == 2 && split_on))) {
        static const bool allowed = hipFuncSetAttribute(reinterpret_cast<const void*>(stem7x7s2_split_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        (int)stem_split_lds_bytes()) == hipSuccess;
        if (allowed) {
            timed_launch(stem7x7s2_split_fwd_kernel, dim3(ntiles < 256 ? ntiles : 256), dim3(256), stem_split_lds_bytes(), s, q, d->w_stem, tiles_x, tiles_y);
            return true;
        }
    }
    const dim3 grid(ntiles < 512 ? ntiles : 512);         // persistent: 2 per CU
    if (kind == 1) timed_launch(stem7x7s2_fwd_kernel<StemDepth>, grid, dim3(256), 0, s, q, d->w_stem, tiles_x, tiles_y);
    else timed_launch(stem7x7s2_fwd_kernel<StemPose>, grid, dim3(256), 0, s, q, d->w_stem, tiles_x, tiles_y);
    return true;
}

MCAV_EXPORT int mcav_pack_stem_weights(const float* w_oihw, int Cout, int Cin, float* packed, void* stream) {
    if (!w_oihw || !packed || !((Cout == 64 && Cin == 3) || (Cout == 16 && Cin == 9))) return MCAV_E_INVALID;
    const int total = Cin * 56 * Cout;
    pack_stem_weights_kernel<<<(total + 255) / 256, 256, 0, as_stream(stream)>>>(w_oihw, packed, Cin, Cout);
    return launch_status();
}

// ---- weight gradient hooks (called by the planner / launcher of conv_igemm.hip, as the halo kernels' are)
static int stem_wgrad_kind(const mcav_wgrad_desc* d) {
    if (!d || d->kh != 7 || d->kw != 7 || d->stride != 2 || d->sign != 1 || d->offset != -3 || d->C2 != 0 || d->upm || ((d->tile >> 9) & 1)) return 0;
    if (d->pad_mode != MCAV_PAD_ZERO || d->up1) return 0;
    if (d->Hd != (d->Hs - 1) / 2 + 1 || d->Wd != (d->Ws - 1) / 2 + 1 || (d->Cdy & 3) || (d->dy_choff & 3)) return 0;
    if (d->mode == MCAV_G_SMALLC && d->C1 == 4 && d->Kp == 4 && d->Cout == 64 && d->Cin == 3 && d->Cdy - d->dy_choff >= 64) return 1;
    if (d->mode == MCAV_G_DIRECT && d->C1 == 16 && d->Kp == 16 && d->Cout == 16 && d->Cin == 9 && d->Cdy - d->dy_choff >= 16) return 2;
    return 0;
}

int mcav_stem_wgrad_splits(const mcav_wgrad_desc* d) {      // 0 = not applicable, else the number of slab partials (= persistent workgroups)
    if (!stem_wgrad_kind(d)) return 0;
    const long tiles = (long)d->B * ((d->Hd + SW_TH - 1) / SW_TH) * ((d->Wd + ST_TW - 1) / ST_TW);
    return (int)(tiles < 512 ? tiles : 512);
}

void mcav_stem_wgrad_launch(const mcav_wgrad_desc* d, float* slab, int Ktot, int slabN, int splits, hipStream_t s) {
    const int wb = d->dbias != nullptr;
    if (stem_wgrad_kind(d) == 1)
        timed_launch(stem7x7s2_wgrad_kernel<3, 4, 64, 4>, dim3(splits), dim3(256), 0, s, d->x1, d->dy, d->B, d->Hs, d->Ws, d->Hd, d->Wd, d->Cdy, d->dy_choff,
                     slab, Ktot, slabN, wb);
    else
        timed_launch(stem7x7s2_wgrad_kernel<9, 16, 16, 16>, dim3(splits), dim3(256), 0, s, d->x1, d->dy, d->B, d->Hs, d->Ws, d->Hd, d->Wd, d->Cdy, d->dy_choff,
                     slab, Ktot, slabN, wb);
}
