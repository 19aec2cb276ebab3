// Pieces shared by the fp32 (conv_igemm.hip) and bf16 (conv_bf16.hip) implicit-GEMM kernels: tile shapes, the row decoder, the epilogue,
// the weight-gradient parameter block and the host-side planners.
#pragma once
#include <stdlib.h>
#include <type_traits>
#include <utility>

#include "conv_gather.h"

namespace mcav {

template <int BM_, int BN_, int WM_, int WN_, int MF_, int CK_ = 16>
struct Tile {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, MF = MF_;
    static constexpr int KD = CK_;                            // K-tile depth of the forward/dgrad kernel (channels of one tap)
    static constexpr int LD = CK_ + 4;                        // LDS row stride: 20 or 36 floats, both conflict-free for ds_read_b128
    static constexpr int LPR = CK_ / 4;                       // lanes (16-byte columns) per operand row
    static constexpr int RPP = 256 / LPR;                     // rows loaded per pass of the 256 threads
    static constexpr int WAVES_M = BM / WM, WAVES_N = BN / WN;
    static constexpr int TM = WM / MF, TN = WN / MF;         // MFMA tiles per wavefront
    static constexpr int ACC = MF == 32 ? 16 : 4;             // accumulator registers per MFMA tile
    using AccT = typename std::conditional<MF_ == 32, f32x16, f32x4>::type;
    static constexpr int AROWS = BM / RPP;                    // A rows per thread per K-tile
    static constexpr int BVECS = (BN * LPR + 255) / 256;      // B float4 per thread per K-tile
    static_assert(WAVES_M * WAVES_N == 4, "4 wavefronts per workgroup");
    static_assert(BM % RPP == 0, "BM multiple of the rows per pass");
};

using Tile128x64 = Tile<128, 64, 64, 32, 32>;
using Tile64x64 = Tile<64, 64, 32, 32, 32>;
using Tile256x32 = Tile<256, 32, 64, 32, 32>;
using Tile256x16 = Tile<256, 16, 64, 16, 16>;
using Tile128x128 = Tile<128, 128, 64, 64, 32>;
using Tile64x16 = Tile<64, 16, 16, 16, 16>;
using Tile128x32 = Tile<128, 32, 32, 32, 32>;
using Tile128x64k32 = Tile<128, 64, 64, 32, 32, 32>;      // 32-deep K-tiles: twice the MFMA work per barrier / per load round trip
using Tile128x128k32 = Tile<128, 128, 64, 64, 32, 32>;
using Tile64x64k32 = Tile<64, 64, 32, 32, 32, 32>;
using Tile64x64k64 = Tile<64, 64, 32, 32, 32, 64>;       // 64-deep: 32 MFMAs per wavefront between barriers
using Tile32x64k32 = Tile<32, 64, 32, 16, 16, 32>;       // short tiles for the 6x20 maps of layer4 (M = 2880): twice the workgroups

// f(integral_constant<int, 0>{}), ..., f(integral_constant<int, N - 1>{}): a loop whose index is a compile-time constant in the body
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// destination pixel of GEMM row m: returns false for padding rows.
__device__ __forceinline__ bool decode_row(const IgemmParams& p, int m, int& n, int& dy, int& dx) {
    if (p.g.mode == MCAV_G_ADJ_STRIDE2) {
        // row blocks hold the parity classes in the order 3, 2, 1, 0: class (1,1) visits four taps of a 3x3 filter, class (0,0) one --
        // the long tiles are dispatched first and the short ones fill the tail
        const int blk = m / p.McP, r = m - blk * p.McP, cls = 3 - blk;
        if (r >= p.Mc) return false;
        const int Hc = (p.Hd + 1) >> 1, Wc = (p.Wd + 1) >> 1;
        n = r / (Hc * Wc);
        const int q = r - n * (Hc * Wc);
        const int y2 = q / Wc, x2 = q - y2 * Wc;
        dy = 2 * y2 + (cls >> 1);
        dx = 2 * x2 + (cls & 1);
        return dy < p.Hd && dx < p.Wd;
    }
    if (m >= p.M) return false;
    if (p.upm) {             // merged-tap upsample: a tile holds one output parity class (py, px); the four classes of one image region are
                             // consecutive tiles (same XCD, same time: the skip tensor's pixels they all read stay in L2)
        const int blk = m / p.bm, cls = blk & 3, r = (blk >> 2) * p.bm + (m - blk * p.bm);
        if (r >= p.Mc) return false;
        const int Hc = p.Hd >> 1, Wc = p.Wd >> 1;
        n = r / (Hc * Wc);
        const int q = r - n * (Hc * Wc);
        const int y2 = q / Wc, x2 = q - y2 * Wc;
        dy = 2 * y2 + (cls >> 1);
        dx = 2 * x2 + (cls & 1);
        return true;
    }
    if (p.groups > 1) {      // group-major rows, each group padded to whole tiles
        const int grp = m / p.McP, r = m - grp * p.McP;
        if (r >= p.Mc) return false;
        const int hw = p.Hd * p.Wd, ng = r / hw, q = r - ng * hw;
        n = grp * (p.g.B / p.groups) + ng;
        dy = q / p.Wd;
        dx = q - dy * p.Wd;
        return true;
    }
    if (p.pool) {
        const int blk = m >> 2, q = m & 3;
        const int Hh = p.Hd >> 1, Wh = p.Wd >> 1;
        n = blk / (Hh * Wh);
        const int r = blk - n * (Hh * Wh);
        const int y2 = r / Wh, x2 = r - y2 * Wh;
        dy = 2 * y2 + (q >> 1);
        dx = 2 * x2 + (q & 1);
        return true;
    }
    n = m / (p.Hd * p.Wd);
    const int r = m - n * (p.Hd * p.Wd);
    dy = r / p.Wd;
    dx = r - dy * p.Wd;
    return true;
}

// Kernel kinds (compile-time specialisations of the A-operand gather; the generic one handles everything):
//   K_FAST    DIRECT gather, every source tensor has a multiple of 4 channels: one source pixel per (row, tap), offsets cached
//             per tap, BRANCH-FREE 16-byte loads (invalid rows load from offset 0 and are zeroed by a select) so that the
//             compiler keeps all loads of a tile in flight behind one counted s_waitcnt
//   K_REFLADJ adjoint of the 3x3 reflection-padded conv: same as K_FAST away from the image border; wavefronts that touch the
//             border (wave-uniform test) take 4 predicated loads per row instead of 1
//   K_GENERIC any mode / any channel count (image stem, stride-2 adjoint, 1-channel disparity maps)
enum { K_FAST = 0, K_REFLADJ = 1, K_GENERIC = 2 };

// Epilogue shared by the forward/dgrad kernels: bias, activation, optional act'(aux) factor and addend, 2x2 sum-pool, channel-range
// store, BatchNorm column sums.  C/D layout: 32x32: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5);
//                                            16x16: col = lane & 15, row = 4 (lane >> 4) + r.   Registers 4q..4q+3 are 4 consecutive rows.
template <class T>
__device__ __forceinline__ void igemm_epilogue(const IgemmParams& p, typename T::AccT (&acc)[T::TM][T::TN], const int* s_out,
                                               float (*s_stat)[2][T::BN], int tid, int wm0, int wn0, int n0, int mt) {
    constexpr int BN = T::BN;
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int MF = T::MF;
    const int ccol = lane & (MF - 1);
    float ssum[T::TN], ssq[T::TN];
#pragma unroll
    for (int j = 0; j < T::TN; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j) {
            const int nl = n0 + wn0 + j * MF + ccol;          // column within this launch
            const bool ncol = nl < p.n_count;
            const float bv = (p.bias && ncol) ? p.bias[p.n_begin + nl] : 0.f;
#pragma unroll
            for (int q = 0; q < T::ACC / 4; ++q) {
                const int rbase = wm0 + i * MF + (MF == 32 ? 8 * q + 4 * (lane >> 5) : 4 * (lane >> 4));
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act_fwd(acc[i][j][4 * q + e] + bv, p.act);
                if (p.pool) {
                    const int o = s_out[rbase];
                    if (o >= 0 && ncol) {
                        float s = (v[0] + v[1]) + (v[2] + v[3]);
                        const size_t off = (size_t)o * p.Cd + p.y_choff + nl;
                        if (p.dact_aux && !(p.dact & MCAV_DACT_AFTER_ADDEND)) s *= act_bwd(p.dact_aux[off], p.dact & 0xff);
                        if (p.addend) s += p.addend[off];
                        if (p.dact_aux && (p.dact & MCAV_DACT_AFTER_ADDEND)) s *= act_bwd(p.dact_aux[off], p.dact & 0xff);
                        p.y[off] = s;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int o = s_out[rbase + e];
                        if (o >= 0 && ncol) {
                            float s = v[e];
                            const size_t off = (size_t)o * p.Cd + p.y_choff + nl;
                            if (p.dact_aux && !(p.dact & MCAV_DACT_AFTER_ADDEND)) s *= act_bwd(p.dact_aux[off], p.dact & 0xff);
                            if (p.addend) s += p.addend[off];
                            if (p.dact_aux && (p.dact & MCAV_DACT_AFTER_ADDEND)) s *= act_bwd(p.dact_aux[off], p.dact & 0xff);
                            p.y[off] = s;
                            ssum[j] += s;
                            if (p.stats_x) {
                                const int gi = (mt / (p.mtiles / p.groups)) * p.n_count + nl;
                                ssq[j] += s * ((p.stats_x[off] - p.stats_mean[gi]) * p.stats_invstd[gi]);
                            } else {
                                ssq[j] += s * s;
                            }
                        }
                    }
                }
            }
        }
    if (p.stats) {
        // column sums over this workgroup's rows: lanes holding the same column, then the wavefronts stacked along M
#pragma unroll
        for (int j = 0; j < T::TN; ++j) {
            if (MF == 32) {
                ssum[j] += __shfl_xor(ssum[j], 32, 64);
                ssq[j] += __shfl_xor(ssq[j], 32, 64);
            } else {
                ssum[j] += __shfl_xor(ssum[j], 16, 64); ssq[j] += __shfl_xor(ssq[j], 16, 64);
                ssum[j] += __shfl_xor(ssum[j], 32, 64); ssq[j] += __shfl_xor(ssq[j], 32, 64);
            }
            if (lane < MF) {
                s_stat[wave / T::WAVES_N][0][wn0 + j * MF + lane] = ssum[j];
                s_stat[wave / T::WAVES_N][1][wn0 + j * MF + lane] = ssq[j];
            }
        }
        __syncthreads();
        for (int e = tid; e < 2 * BN; e += 256) {
            const int which = e / BN, col = e - which * BN;
            if (n0 + col < p.n_count) {
                float s = 0.f;
#pragma unroll
                for (int wmi = 0; wmi < T::WAVES_M; ++wmi) s += s_stat[wmi][which][col];
                p.stats[((size_t)mt * 2 + which) * p.n_count + n0 + col] = s;
            }
        }
    }
}

// The same epilogue for the table-driven kernels, written for few vector instructions (fp32 MFMAs and the VALU share the SIMD's FMA lanes,
// so every vector instruction outside the K loop is taken from the other resident workgroups' MFMA time): s_outb holds each tile row's
// BYTE offset into y (OOB for rows past M), a 4-row group is one 16-byte LDS read, stores / aux loads are raw buffer accesses whose
// out-of-range offsets are dropped in hardware, and the aux / addend values of a 32x32 block are all loaded before the first use.
template <class T>
__device__ __forceinline__ void igemm_epilogue_lean(const IgemmParams& p, typename T::AccT (&acc)[T::TM][T::TN], const unsigned* s_outb,
                                                    float (*s_stat)[2][T::BN], int tid, int wm0, int wn0, int n0, int mt) {
    constexpr int BN = T::BN, MF = T::MF, NQ = T::ACC / 4;
    const int lane = tid & 63, wave = tid >> 6;
    const int ccol = lane & (MF - 1);
    const unsigned ybytes = (unsigned)((size_t)p.g.B * (p.pool ? (p.Hd >> 1) * (p.Wd >> 1) : p.Hd * p.Wd) * p.Cd * 4);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.y, ybytes);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.dact_aux ? p.dact_aux : p.y, ybytes);
    const __amdgpu_buffer_rsrc_t rad = make_rsrc(p.addend ? p.addend : p.y, ybytes);
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(p.stats_x ? p.stats_x : p.y, ybytes);
    const int sgrp = p.stats_x ? (mt / (p.mtiles / p.groups)) * p.n_count : 0;      // this tile's group in stats_mean / stats_invstd
    float ssum[T::TN], ssq[T::TN];
#pragma unroll
    for (int j = 0; j < T::TN; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j) {
            __builtin_amdgcn_sched_barrier(0);                // one accumulator block at a time (hoisted across blocks, the aux loads cost 40 registers)
            const int nl = n0 + wn0 + j * MF + ccol;          // column within this launch
            if (nl < p.n_count) {
                const float bv = p.bias ? p.bias[p.n_begin + nl] : 0.f;
                const unsigned colb = (unsigned)(p.y_choff + nl) * 4u;
                u32x4 ro[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int rbase = wm0 + i * MF + (MF == 32 ? 8 * q + 4 * (lane >> 5) : 4 * (lane >> 4));
                    ro[q] = *reinterpret_cast<const u32x4*>(&s_outb[rbase]);
                }
                if (p.pool) {
                    float aux[NQ], add[NQ];
                    if (p.dact_aux) {
#pragma unroll
                        for (int q = 0; q < NQ; ++q) aux[q] = buf_load1(rx, ro[q].x + colb);
                    }
                    if (p.addend) {
#pragma unroll
                        for (int q = 0; q < NQ; ++q) add[q] = buf_load1(rad, ro[q].x + colb);
                    }
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = act_fwd(acc[i][j][4 * q + e] + bv, p.act);
                        float s = (v[0] + v[1]) + (v[2] + v[3]);
                        if (p.dact_aux && !(p.dact & MCAV_DACT_AFTER_ADDEND)) s *= act_bwd(aux[q], p.dact & 0xff);
                        if (p.addend) s += add[q];
                        if (p.dact_aux && (p.dact & MCAV_DACT_AFTER_ADDEND)) s *= act_bwd(aux[q], p.dact & 0xff);
                        buf_store1(ry, ro[q].x + colb, s);
                    }
                } else {
                    // half a block (8 rows per lane) at a time: its aux / addend loads are all in flight before the first use
                    constexpr int QH = NQ >= 2 ? NQ / 2 : 1;
#pragma unroll
                    for (int q0 = 0; q0 < NQ; q0 += QH) {
                        float aux[4 * QH], add[4 * QH], xs[4 * QH];
                        float smu = 0.f, sis = 0.f;
                        if (p.stats_x) {      // BatchNorm-backward statistics: the raw conv output of the layer being differentiated, its mean / invstd
                            smu = p.stats_mean[sgrp + nl];
                            sis = p.stats_invstd[sgrp + nl];
#pragma unroll
                            for (int q = 0; q < QH; ++q)
#pragma unroll
                                for (int e = 0; e < 4; ++e) xs[4 * q + e] = buf_load1(rsx, ro[q0 + q][e] + colb);
                        }
                        if (p.dact_aux) {
#pragma unroll
                            for (int q = 0; q < QH; ++q)
#pragma unroll
                                for (int e = 0; e < 4; ++e) aux[4 * q + e] = buf_load1(rx, ro[q0 + q][e] + colb);
                        }
                        if (p.addend) {
#pragma unroll
                            for (int q = 0; q < QH; ++q)
#pragma unroll
                                for (int e = 0; e < 4; ++e) add[4 * q + e] = buf_load1(rad, ro[q0 + q][e] + colb);
                        }
#pragma unroll
                        for (int q = 0; q < QH; ++q)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float s = acc[i][j][4 * (q0 + q) + e] + bv;
                                if (p.act == MCAV_ACT_RELU) s = fmaxf(s, 0.f);
                                else if (p.act != MCAV_ACT_NONE) s = act_fwd(s, p.act);
                                if (p.dact_aux && !(p.dact & MCAV_DACT_AFTER_ADDEND)) s *= act_bwd(aux[4 * q + e], p.dact & 0xff);
                                if (p.addend) s += add[4 * q + e];
                                if (p.dact_aux && (p.dact & MCAV_DACT_AFTER_ADDEND)) s *= act_bwd(aux[4 * q + e], p.dact & 0xff);
                                buf_store1(ry, ro[q0 + q][e] + colb, s);
                                if (p.stats) {
                                    const float sv = ro[q0 + q][e] != OOB ? s : 0.f;
                                    ssum[j] += sv;
                                    ssq[j] += p.stats_x ? sv * ((xs[4 * q + e] - smu) * sis) : sv * sv;
                                }
                            }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    if (p.stats) {
#pragma unroll
        for (int j = 0; j < T::TN; ++j) {
            if (MF == 32) {
                ssum[j] += __shfl_xor(ssum[j], 32, 64);
                ssq[j] += __shfl_xor(ssq[j], 32, 64);
            } else {
                ssum[j] += __shfl_xor(ssum[j], 16, 64); ssq[j] += __shfl_xor(ssq[j], 16, 64);
                ssum[j] += __shfl_xor(ssum[j], 32, 64); ssq[j] += __shfl_xor(ssq[j], 32, 64);
            }
            if (lane < MF) {
                s_stat[wave / T::WAVES_N][0][wn0 + j * MF + lane] = ssum[j];
                s_stat[wave / T::WAVES_N][1][wn0 + j * MF + lane] = ssq[j];
            }
        }
        __syncthreads();
        for (int e = tid; e < 2 * BN; e += 256) {
            const int which = e / BN, col = e - which * BN;
            if (n0 + col < p.n_count) {
                float s = 0.f;
#pragma unroll
                for (int wmi = 0; wmi < T::WAVES_M; ++wmi) s += s_stat[wmi][which][col];
                p.stats[((size_t)mt * 2 + which) * p.n_count + n0 + col] = s;
            }
        }
    }
}

constexpr int TAB_TAPS = 32;     // 3x3 filters, the 4x4 stride-2 form of the pooled upsample adjoint, PoseNet's 5x5
constexpr int WG_TABCAP = 2048;

struct WgradParams {
    GatherSrc g;
    int kh, kw, Kp, taps, Ktot;   // Ktot = taps * Kp (GEMM rows)
    const float* dy;
    int Hd, Wd, Cdy, dy_choff, Cout;
    int CoutLoad;                  // Cout rounded up to 4 when dy physically has those (zero) channels
    int Mpix;                      // B * Hd * Wd
    int splits, pix_per_split;     // pixel ranges per workgroup (multiple of KP)
    int mtiles, ntiles;
    float* slab;                   // [splits][Ktot + 1][slabN]; row Ktot holds the per-split column sums of dy (bias gradient)
    int slabN;                     // row stride of the slab (Cout rounded up to 16)
    int want_bias;
    int tab_cht_log2;              // wgrad_tab_kernel: log2 of the tiles per table chunk (>= 20: the whole split is one chunk)
    int upm;                       // merged-tap upsample (mcav_wgrad_desc.upm): rows = 16 (class, merged tap) x Kp, pixels = LOW-resolution ones
    int Hf, Wf;                    // upm: full-resolution size of dy (Hd, Wd hold the low-resolution one)
    int split_planes;              // conv_bf16.hip: 1 = the fp32 contraction on three bf16 planes per operand (mcav_wgrad_desc.mma = 2)
    // conv_bf16.hip, wgrad3x3_patch_kernel (patch = 1): pixel blocks of TH x TW (<= 64 pixels) of one image, `bps` consecutive blocks per split,
    // workgroup = (split, 64 input x 64 output channels)
    int patch, pTH, pTW, ptiles_y, ptiles_x, prefl, pnblocks, pbps, pct_co, pnarrow;
};

constexpr int KP = 32;   // pixels per K-tile of the wgrad GEMM

struct WgradPlan {
    WgradParams p;
    int tile;
    bool use_tab;
    bool use_halo;
    bool use_stem;                 // conv_stem.hip: the 7x7 stride-2 image stem's own weight-gradient kernel
    size_t slab_bytes, pre_bytes;
    int ci_t, groups, per_group;
};


// host-side planners (conv_igemm.hip)
bool fill_params(const mcav_igemm_desc* d, IgemmParams& p, int& tile);
bool plan_wgrad(const mcav_wgrad_desc* d, WgradPlan& pl);
void tile_dims(int id, int& BM, int& BN);

}  // namespace mcav
