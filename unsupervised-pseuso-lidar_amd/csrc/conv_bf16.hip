// bf16 MFMA tiles of the implicit-GEMM convolution (BASELINE.json configs[2] / [4]: "bf16 MFMA conv tiles").
//
// Same GEMMs, tables, epilogues and slab reductions as conv_igemm.hip, with the contraction on v_mfma_f32_32x32x16_bf16 /
// v_mfma_f32_16x16x32_bf16 (16x the fp32 MFMA rate) and fp32 accumulation:
//   * activations and gradients stay fp32 NHWC in HBM (every other kernel of the step is unchanged); a K-tile is loaded with the same
//     16-byte table-driven buffer loads and rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) on its way into LDS, so an LDS
//     panel holds twice the K depth per byte;
//   * filters come from bf16 packed copies (mcav_pack_weights_multi with the bf16 flag) of the fp32 master weights;
//   * BatchNorm statistics, bias, activations, slabs and the OIHW gradients are fp32 exactly as in the fp32 path.
// At this rate the kernels are no longer MFMA-bound but L2/HBM-bound (the fp32 A operand is 64 B per pixel per 16 channels), so the
// K loop is the plain "write early" pipeline -- no per-MFMA piece placement -- at four workgroups per CU.
// Weight gradient: the reduction runs over pixels, and a bf16 MFMA wants 8 consecutive k per lane, so the panels are stored
// k-contiguous ([row][64 pixels]): a thread loads 4 channels of 8 consecutive pixels and packs, per channel, the 8 pixels into one
// 16-byte LDS store (a register transpose; the cvt_pk pairs consecutive pixels), XOR-swizzled by (row >> 1) & 7 so that the
// ds_read_b128 fragment reads are conflict-free.
#include "conv_shared.h"
#include "kernel_timer.h"

namespace mcav {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int BM_, int BN_, int WM_, int WN_, int MF_, int CK_>
struct BTile {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, MF = MF_;
    static constexpr int KD = CK_;                            // channels per K-tile
    static constexpr int LDH = CK_ + 8;                       // LDS row stride in bf16 elements: 2 CK + 16 bytes (80 / 144: conflict-free ds_read_b128)
    static constexpr int LPRA = CK_ / 4;                      // lanes per A row (each loads 4 fp32 channels = 16 bytes)
    static constexpr int RPPA = 256 / LPRA;
    static constexpr int AROWS = BM / RPPA;
    static constexpr int LPRB = CK_ / 8;                      // 16-byte pieces (8 bf16) per B row
    static constexpr int BVECS = (BN * LPRB + 255) / 256;
    static constexpr int WAVES_M = BM / WM, WAVES_N = BN / WN;
    static constexpr int TM = WM / MF, TN = WN / MF;
    static constexpr int ACC = MF == 32 ? 16 : 4;
    using AccT = typename std::conditional<MF_ == 32, f32x16, f32x4>::type;
    static_assert(WAVES_M * WAVES_N == 4, "4 wavefronts per workgroup");
    static_assert(BM % RPPA == 0, "BM multiple of the rows per pass");
    static_assert(MF_ == 32 ? CK_ % 16 == 0 : CK_ % 32 == 0, "K-tile holds whole MFMA k-steps");
};

using BT64x64k64 = BTile<64, 64, 32, 32, 32, 64>;
using BT64x64k32 = BTile<64, 64, 32, 32, 32, 32>;
using BT128x64k64 = BTile<128, 64, 64, 32, 32, 64>;
using BT128x64k32 = BTile<128, 64, 64, 32, 32, 32>;
using BT32x64k64 = BTile<32, 64, 32, 16, 16, 64>;
using BT32x64k32 = BTile<32, 64, 32, 16, 16, 32>;

__device__ __forceinline__ u32x2 pack_bf16x4(f32x4 v) {
    const bf16x4 h = __builtin_convertvector(v, bf16x4);
    return __builtin_bit_cast(u32x2, h);
}

// ------------------------------------------------------------------------------------------------ forward / data gradient
// TK = 0: DIRECT gathers (zero / reflection padding, stride 1 / 2, fused upsample + concat) and the stride-2 adjoint;
// TK = 1: the adjoint of the 3x3 reflection-padded conv (border wavefronts add up to three reflected sources per row, in fp32, before the
//         rounding to bf16).  Prologue (row decode, tap list, offset tables) as igemm_tab_kernel.
//
// NS = 1: operands rounded to bf16 (mma = 1).  NS = 3: an fp32 contraction on the bf16 MFMA (mma = 2): each operand element is split into
// three bf16 planes a = h + m + l (h = bf16(a), m = bf16(a - h), l = bf16(a - h - m), round to nearest: what is left is below 2^-26 |a|),
// the filter planes come pre-split from the packed copy, and the six plane products hh, hm, mh, hl, lh, mm are accumulated in fp32 (dropped:
// ml, lm, ll <= 2^-26 of a product, under the rounding of one fp32 multiply-add); the products below hh go into accumulators of their own
// and are added once at the end.  Measured against float64 (tools/mfma_split_test.hip, profiles/r03_mfma_split_exactness.txt): the result
// is at least as exact as v_mfma_f32_32x32x2_f32 on the unsplit operands, at 6 / 16 of its MFMA time.  The planes triple the LDS panel, so
// the split form keeps ONE panel (two barriers per K-tile; the other workgroups of the CU cover them).
template <class T, int TK, int NS>
__global__ __launch_bounds__(256) void igemm_bf16_kernel(IgemmParams p, const u16* __restrict__ w16) {
    constexpr bool REFL = TK == 1;
    constexpr int BM = T::BM, BN = T::BN, CKT = T::KD;
    constexpr int NB = NS == 1 ? 2 : 1;                           // LDS panels; indexed [NB * NS]: buffer b of the plain form = plane 0 of panel b
    __shared__ __attribute__((aligned(16))) u16 As[NB * NS][BM][T::LDH];
    __shared__ __attribute__((aligned(16))) u16 Bs[NB * NS][BN][T::LDH];
    __shared__ __attribute__((aligned(16))) unsigned s_out[BM];      // byte offset of each row's output pixel (OOB: none): the lean epilogue's form
    float (*const s_stat)[2][BN] = reinterpret_cast<float (*)[2][BN]>(&As[1][0][0]);
    static_assert(sizeof(u16) * BM * T::LDH >= sizeof(float) * T::WAVES_M * 2 * BN, "statistics scratch fits one A buffer");
    extern __shared__ unsigned s_dyn[];
    unsigned* const s_o1 = s_dyn;
    unsigned* const s_o2 = s_dyn + (p.g.C2 > 0 ? p.taps * BM : 0);
    __shared__ int s_rows[REFL ? 3 * BM : 1];
    int* const s_rn = REFL ? s_rows : reinterpret_cast<int*>(&As[0][0][0]);
    int* const s_ry = s_rn + BM;
    int* const s_rx = s_rn + 2 * BM;
    static_assert(sizeof(u16) * 2 * BM * T::LDH >= sizeof(int) * 3 * BM, "row coordinates fit the A panel");
    __shared__ int s_tl[TAB_TAPS + 1];
    __shared__ int s_nt;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lid = p.g.mode == MCAV_G_ADJ_STRIDE2 ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    const int nt = lid % p.ntiles, mt = lid / p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const GatherSrc& g = p.g;

    for (int r = tid; r < BM; r += 256) {
        int n, dy, dx;
        const bool ok = decode_row(p, m0 + r, n, dy, dx);
        int o = -1;
        if (ok) o = p.pool ? ((n * (p.Hd >> 1) + (dy >> 1)) * (p.Wd >> 1) + (dx >> 1)) : ((n * p.Hd + dy) * p.Wd + dx);
        s_out[r] = ok ? (unsigned)o * (unsigned)(p.Cd * 4) : OOB;
        s_rn[r] = ok ? n : -1; s_ry[r] = dy; s_rx[r] = dx;
    }
    if (tid == 0) {
        int nv = 0;
        const bool adj = g.mode == MCAV_G_ADJ_STRIDE2;
        const int cls = adj ? 3 - m0 / p.McP : 0, cpy = cls >> 1, cpx = cls & 1;
        for (int t = 0; t < p.taps; ++t) {
            const int ky = t / p.kw, kx = t - ky * p.kw;
            if (!adj || (((cpy + g.offset - ky) | (cpx + g.offset - kx)) & 1) == 0) s_tl[nv++] = t;
        }
        s_tl[nv] = 0;
        s_nt = nv;
    }
    __syncthreads();
    for (int e = tid; e < p.taps * BM; e += 256) {
        const int tp = e / BM, r = e - tp * BM;
        const int ky = tp / p.kw, kx = tp - ky * p.kw;
        const int n = s_rn[r], dy = s_ry[r], dx = s_rx[r];
        int sy, sx;
        bool ok = n >= 0;
        if (REFL) {
            sy = dy + 1 - ky;
            sx = dx + 1 - kx;
        } else if (g.mode == MCAV_G_ADJ_STRIDE2) {
            const int ty = dy + g.offset - ky, tx = dx + g.offset - kx;
            ok = ok && ty >= 0 && tx >= 0 && (((ty | tx) & 1) == 0);
            sy = ty >> 1; sx = tx >> 1;
        } else {
            sy = dy * g.stride + g.sign * ky + g.offset;
            sx = dx * g.stride + g.sign * kx + g.offset;
            if (g.pad_mode == MCAV_PAD_REFLECT) {
                sy = reflect_idx(sy, g.Hs);
                sx = reflect_idx(sx, g.Ws);
            }
        }
        ok = ok && (unsigned)sy < (unsigned)g.Hs && (unsigned)sx < (unsigned)g.Ws;
        const int pix = (n * g.Hs + sy) * g.Ws + sx;
        const int pix1 = g.up1 ? ((n * (g.Hs >> 1) + (sy >> 1)) * (g.Ws >> 1) + (sx >> 1)) : pix;
        s_o1[tp * BM + r] = ok ? (unsigned)(pix1 * g.C1) * 4u : OOB;
        if (g.C2 > 0) s_o2[tp * BM + r] = ok ? (unsigned)(pix * g.C2) * 4u : OOB;
    }
    __syncthreads();

    const int ntaps = s_nt;
    const int nchunks = p.Kp / CKT;
    const int T_total = ntaps * nchunks;
    const unsigned bytes1 = (unsigned)((size_t)g.B * (g.up1 ? (g.Hs >> 1) * (g.Ws >> 1) : g.Hs * g.Ws) * g.C1 * 4);
    const unsigned bytes2 = (unsigned)((size_t)g.B * g.Hs * g.Ws * g.C2 * 4);
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(g.x1, bytes1);
    const __amdgpu_buffer_rsrc_t rs2 = make_rsrc(g.C2 > 0 ? g.x2 : g.x1, g.C2 > 0 ? bytes2 : 0u);
    const unsigned plane_bytes = (unsigned)((size_t)p.Np_all * p.Kstride * 2);      // NS = 3: plane q of the filter starts q * plane_bytes further
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(w16, (unsigned)((size_t)(p.n_begin + p.n_count) * p.Kstride * 2) + (NS - 1) * plane_bytes);

    const int c4 = tid % T::LPRA, r0 = tid / T::LPRA;
    unsigned boff[T::BVECS];
#pragma unroll
    for (int j = 0; j < T::BVECS; ++j) {
        const int e = tid + 256 * j, nn = e / T::LPRB, cb = e % T::LPRB;
        const bool ok = nn < BN && n0 + nn < p.n_count;
        boff[j] = ok ? (unsigned)(((p.n_begin + n0 + nn) * p.Kstride + cb * 8) * 2) : OOB;
    }
    bool wave_border = false;
    if constexpr (REFL) {
        bool bd = false;
#pragma unroll
        for (int j = 0; j < T::AROWS; ++j) {
            const int r = r0 + T::RPPA * j;
            const int y = s_ry[r], x = s_rx[r];
            bd = bd || (s_rn[r] >= 0 && (y <= 1 || y >= g.Hs - 2 || x <= 1 || x >= g.Ws - 2));
        }
        wave_border = __any(bd);
    }
    constexpr int EXR = REFL ? T::AROWS : 1;                     // a register stage's reflected border sources
    int ti = 0, chunk = 0;
    int tap = __builtin_amdgcn_readfirstlane(s_tl[0]);
    unsigned oa[T::AROWS], ob[T::AROWS];
    auto refresh = [&]() {
#pragma unroll
        for (int j = 0; j < T::AROWS; ++j) {
            oa[j] = s_o1[tap * BM + r0 + T::RPPA * j] + (unsigned)c4 * 16u;
            ob[j] = s_o2[tap * BM + r0 + T::RPPA * j] + (unsigned)c4 * 16u;
        }
    };
    refresh();
    auto issue = [&](f32x4 (&ra)[T::AROWS], f32x4 (&rb)[NS][T::BVECS], f32x4 (&ex)[3][EXR]) {
        const int cbase = chunk * CKT;
        if (cbase < g.C1) {
#pragma unroll
            for (int j = 0; j < T::AROWS; ++j) ra[j] = buf_load4s(rs1, oa[j], cbase * 4);
        } else {
#pragma unroll
            for (int j = 0; j < T::AROWS; ++j) ra[j] = buf_load4s(rs2, ob[j], (cbase - g.C1) * 4);
        }
        if constexpr (REFL) {
            if (wave_border) {
                const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                for (int j = 0; j < T::AROWS; ++j) {
                    const int r = r0 + T::RPPA * j;
                    const int n = s_rn[r], dy = s_ry[r], dx = s_rx[r];
                    const int sy = dy + 1 - ky, sx = dx + 1 - kx;
                    const int ey = (dy == 1 && ky == 0) ? 0 : ((dy == g.Hs - 2 && ky == 2) ? g.Hs - 1 : -1);
                    const int exx = (dx == 1 && kx == 0) ? 0 : ((dx == g.Ws - 2 && kx == 2) ? g.Ws - 1 : -1);
                    const bool syok = (unsigned)sy < (unsigned)g.Hs, sxok = (unsigned)sx < (unsigned)g.Ws;
                    const int rowb = n * g.Hs, cb = c4 * 4;
                    const unsigned a0 = (unsigned)((((rowb + ey) * g.Ws + sx) * g.C1 + cb) * 4);
                    const unsigned a1 = (unsigned)((((rowb + sy) * g.Ws + exx) * g.C1 + cb) * 4);
                    const unsigned a2 = (unsigned)((((rowb + ey) * g.Ws + exx) * g.C1 + cb) * 4);
                    ex[0][j] = buf_load4s(rs1, (n >= 0 && ey >= 0 && sxok) ? a0 : OOB, cbase * 4);
                    ex[1][j] = buf_load4s(rs1, (n >= 0 && exx >= 0 && syok) ? a1 : OOB, cbase * 4);
                    ex[2][j] = buf_load4s(rs1, (n >= 0 && ey >= 0 && exx >= 0) ? a2 : OOB, cbase * 4);
                }
            }
        }
        const int kb = (tap * p.Kp + cbase) * 2;
#pragma unroll
        for (int q = 0; q < NS; ++q)
#pragma unroll
            for (int j = 0; j < T::BVECS; ++j) rb[q][j] = buf_load4s(rsw, boff[j], kb + q * (int)plane_bytes);
        if (++chunk == nchunks) {
            chunk = 0;
            ++ti;
            tap = __builtin_amdgcn_readfirstlane(s_tl[ti]);
            refresh();
        }
    };
    constexpr bool BFULL = (BN * T::LPRB) % 256 == 0;
    auto store = [&](f32x4 (&ra)[T::AROWS], const f32x4 (&rb)[NS][T::BVECS], const f32x4 (&ex)[3][EXR], auto bufc) {
        constexpr int buf = decltype(bufc)::value * NS;
#pragma unroll
        for (int j = 0; j < T::AROWS; ++j) {
            if constexpr (REFL) {
                if (wave_border) ra[j] += (ex[0][j] + ex[1][j]) + ex[2][j];
            }
            if constexpr (NS == 1) {
                *reinterpret_cast<u32x2*>(&As[buf][r0 + T::RPPA * j][c4 * 4]) = pack_bf16x4(ra[j]);
            } else {
                const bf16x4 h = __builtin_convertvector(ra[j], bf16x4);
                const f32x4 r1 = ra[j] - __builtin_convertvector(h, f32x4);            // exact
                const bf16x4 m = __builtin_convertvector(r1, bf16x4);
                const f32x4 r2 = r1 - __builtin_convertvector(m, f32x4);               // exact
                *reinterpret_cast<u32x2*>(&As[buf][r0 + T::RPPA * j][c4 * 4]) = __builtin_bit_cast(u32x2, h);
                *reinterpret_cast<u32x2*>(&As[buf + 1][r0 + T::RPPA * j][c4 * 4]) = __builtin_bit_cast(u32x2, m);
                *reinterpret_cast<u32x2*>(&As[buf + 2][r0 + T::RPPA * j][c4 * 4]) = pack_bf16x4(r2);
            }
        }
#pragma unroll
        for (int q = 0; q < NS; ++q)
#pragma unroll
            for (int j = 0; j < T::BVECS; ++j) {
                const int e = tid + 256 * j, nn = e / T::LPRB, cb = e % T::LPRB;
                if (BFULL || nn < BN) *reinterpret_cast<f32x4*>(&Bs[buf + q][nn][cb * 8]) = rb[q][j];
            }
    };

    const int wm0 = (wave / T::WAVES_N) * T::WM, wn0 = (wave % T::WAVES_N) * T::WN;
    typename T::AccT acc[T::TM][T::TN];
    typename T::AccT mid[NS == 1 ? 1 : T::TM][NS == 1 ? 1 : T::TN], low[NS == 1 ? 1 : T::TM][NS == 1 ? 1 : T::TN];      // mh + hm; lh + hl + mm
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < T::ACC; ++r) {
                acc[i][j][r] = 0.f;
                if constexpr (NS > 1) { mid[i][j][r] = 0.f; low[i][j][r] = 0.f; }
            }

    constexpr int MFR = T::MF;
    constexpr int KSTEP = MFR == 32 ? 16 : 32;                     // k per MFMA
    const int frow = lane & (MFR - 1), fk = (lane / MFR) * 8;      // lane group h holds k = 8 h .. 8 h + 7 of a step
    auto mfma = [&](const bf16x8& a, const bf16x8& b, typename T::AccT& c) {
        if constexpr (MFR == 32) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
        else c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    };
    auto compute = [&](auto bufc) {
        constexpr int buf = decltype(bufc)::value * NS;
#pragma unroll
        for (int ks = 0; ks < CKT / KSTEP; ++ks) {
            bf16x8 a[NS][T::TM], b[NS][T::TN];
#pragma unroll
            for (int q = 0; q < NS; ++q) {
#pragma unroll
                for (int i = 0; i < T::TM; ++i) a[q][i] = *reinterpret_cast<const bf16x8*>(&As[buf + q][wm0 + i * MFR + frow][ks * KSTEP + fk]);
#pragma unroll
                for (int j = 0; j < T::TN; ++j) b[q][j] = *reinterpret_cast<const bf16x8*>(&Bs[buf + q][wn0 + j * MFR + frow][ks * KSTEP + fk]);
            }
#pragma unroll
            for (int i = 0; i < T::TM; ++i)
#pragma unroll
                for (int j = 0; j < T::TN; ++j) {
                    if constexpr (NS == 1) {
                        mfma(a[0][i], b[0][j], acc[i][j]);
                    } else {      // (three independent accumulation chains)
                        mfma(a[2][i], b[0][j], low[i][j]);
                        mfma(a[1][i], b[0][j], mid[i][j]);
                        mfma(a[0][i], b[0][j], acc[i][j]);
                        mfma(a[0][i], b[2][j], low[i][j]);
                        mfma(a[0][i], b[1][j], mid[i][j]);
                        mfma(a[1][i], b[1][j], low[i][j]);
                    }
                }
        }
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;

    // "write early": at the top of step t buffer t & 1 holds tile t and the registers hold the in-flight loads of tile t + 1; the step
    // writes them to the other buffer, issues tile t + 2 into the same registers, multiplies tile t, one barrier.  The MFMA phase is
    // short here (32..64 cycles per 32x32x16), so the flight of the loads is covered by the other workgroups of the CU.
    f32x4 ra[T::AROWS], rb[NS][T::BVECS];
    f32x4 ra2[NS == 1 ? 1 : T::AROWS], rb2[NS][NS == 1 ? 1 : T::BVECS];      // the split form's second register stage
    f32x4 ex[3][EXR], ex2[3][NS == 1 ? 1 : EXR];
    if constexpr (NS == 1) {
        if (T_total > 0) {
            issue(ra, rb, ex);
            store(ra, rb, ex, B0{});
            if (T_total > 1) issue(ra, rb, ex);
        }
    } else {
        // tiles t + 1 and t + 2 are in flight while tile t is multiplied (a K-tile's six MFMAs per step are over in a few hundred cycles: one
        // tile of look-ahead does not cover a load's round trip)
        if (T_total > 0) {
            issue(ra, rb, ex);
            if (T_total > 1) issue(ra2, rb2, ex2);
            store(ra, rb, ex, B0{});
            if (T_total > 2) issue(ra, rb, ex);
        }
    }
    __syncthreads();
    if constexpr (NS == 1) {
        int t = 0;
        for (; t + 1 < T_total; t += 2) {
            store(ra, rb, ex, B1{});
            if (t + 2 < T_total) issue(ra, rb, ex);
            compute(B0{});
            __syncthreads();
            if (t + 2 < T_total) {
                store(ra, rb, ex, B0{});
                if (t + 3 < T_total) issue(ra, rb, ex);
            }
            compute(B1{});
            __syncthreads();
        }
        if (t < T_total) {
            compute(B0{});
            __syncthreads();
        }
    } else {
        // one panel: multiply tile t, barrier, write tile t + 1 from its register stage, issue tile t + 3 into that stage, barrier
        for (int t = 0; t < T_total; t += 2) {
            compute(B0{});
            __syncthreads();
            if (t + 1 < T_total) {
                store(ra2, rb2, ex2, B0{});
                if (t + 3 < T_total) issue(ra2, rb2, ex2);
                __syncthreads();
                compute(B0{});
                __syncthreads();
                if (t + 2 < T_total) {
                    store(ra, rb, ex, B0{});
                    if (t + 4 < T_total) issue(ra, rb, ex);
                    __syncthreads();
                }
            }
        }
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int j = 0; j < T::TN; ++j) acc[i][j] += mid[i][j] + low[i][j];
    }
    igemm_epilogue_lean<T>(p, acc, s_out, s_stat, tid, wm0, wn0, n0, mt);      // (round 4: raw buffer stores instead of a branch and a 64-bit address per element)
}

// ------------------------------------------------------------------------------------------------ 3x3 stride-1 zero-padded conv, patch in LDS
// The implicit-GEMM kernels above fetch (and round or split) every source pixel once per filter tap.  For the 3x3 stride-1 zero-padded
// convolutions of the ResNet trunk (and their data gradients: the same contraction with mirrored taps) a workgroup instead owns a TH x TW
// block of output pixels (TH TW <= 64 rows of the GEMM) and keeps the (TH + 2) x (TW + 2) source patch of one 32-channel chunk in LDS, as
// bf16 (NS = 1) or as the three planes of the split fp32 form (NS = 3): a source pixel is fetched and converted ONCE per chunk (1.56x halo
// at 8 x 8 instead of 9x), the nine taps are nine shifted views of the patch -- a tap is a constant added to each lane's patch row --, and
// only the filter tile of the tap (64 outputs x 32 channels per plane, L2-resident) is streamed.  The next chunk's patch is in flight
// during the nine taps of the current one, the next tap's filter tile during the current tap's MFMAs.
// GEMM row R (0..127) of a block -> linear pixel index: the 16 lanes of a ds_read_b128 group ({0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} of each
// 32-row half) get 16 consecutive pixels.
__device__ __forceinline__ int patch_pixel(int R) {
    const int r = R & 31;
    const bool g0 = r < 4 || (r >= 12 && r < 16) || (r >= 20 && r < 28);
    const int idx = g0 ? (r < 4 ? r : (r < 16 ? r - 8 : r - 12)) : (r < 12 ? r - 4 : (r < 20 ? r - 8 : r - 16));
    return (R & ~31) + (g0 ? 0 : 16) + idx;
}

// n / d for the small non-negative integers of the patch kernels' set-up (n < 2^20, d < 2^10), exact: (n + 0.5) / d is at least 0.5 / d away
// from an integer, far more than the rounding of the float product.  An integer division by a run-time value is ~40 vector instructions on
// this ISA, and a patch-kernel workgroup of the 48x160 maps lives for only 6 stages.
__device__ __forceinline__ int small_div(int n, float rcp_d) { return (int)(((float)n + 0.5f) * rcp_d); }

struct PatchGeo {
    int TH, TW, tiles_y, tiles_x;
    int tmb;      // 32-row blocks per wavefront: 1 = 64-pixel blocks, 2 = 128-pixel blocks
    int refl;     // 0 zero padding; 1 reflection padding, forward (the patch's ring is the mirrored row / column); 2 the ADJOINT of reflection
                  // padding (the zero-padded data gradient plus the ring's contributions folded onto rows / columns 1 and H - 2 / W - 2)
};

// Timing experiments only, WRONG results: a compile-time value of experiment builds (make variant NAME=pd8 FLAGS=-DMCAV_PATCH_DIAG=8), zero
// -- every branch on it folded away -- in the shipped library.  1 no filter staging in the loop, 2 no barriers, 4 no patch staging, 8 no
// MFMAs, 16 no epilogue.
#ifndef MCAV_PATCH_DIAG
#define MCAV_PATCH_DIAG 0
#endif

template <int TMB>
struct PatchCfg {                       // TMB 32-row blocks per wavefront: 64- or 128-pixel blocks
    static constexpr int BM = 64 * TMB;
    static constexpr int PIX = TMB == 1 ? 110 : 180;      // patch pixels: (TH + 2)(TW + 2) <= PIX: 4 x 16 -> 108, 3 x 20 -> 110 (8 x 8 -> 100, 6 x 10 -> 96); 8 x 16 -> 180, 6 x 20 -> 176
    static constexpr int NJ = (PIX + 31) / 32;            // patch pixels staged per thread
    using T = typename std::conditional<TMB == 1, BT64x64k32, BT128x64k32>::type;
};

template <int NS, int TMB, bool ADJ = false>                      // ADJ: the adjoint of reflection padding (geo.refl == 2), a form of its own
__global__ __launch_bounds__(256, 2) void conv3x3_patch_kernel(IgemmParams p, const u16* __restrict__ w16, PatchGeo geo) {
    using C = PatchCfg<TMB>;
    using T = typename C::T;
    // LDS rows are 64 bytes (32 channels: four rows per 256-byte bank row) with the four 16-byte slots of row r XOR-ed by (r >> 2) & 3: a
    // ds_read_b128 lane group (16 lanes: MI355X_MICROARCH.md section LDS) is conflict-free when its 16 rows are distinct mod 16.  The
    // filter rows of a group are (row = lane); for the patch, GEMM row R of the block stands for pixel patch_pixel(R), which gives every
    // lane group 16 consecutive pixels -- one pixel row of a 16-wide block, whose patch rows are consecutive whatever the tap.  The staging
    // stores fill whole 64-byte rows (128 bytes per ds_write group).  (Padded 80-byte rows: half of the LDS cycles were bank conflicts.)
    constexpr int BM = C::BM, BN = 64, CKT = 32, LDH = 32, PATCH_PIX = C::PIX;
    constexpr int APL = PATCH_PIX * LDH, BPL = BN * LDH;         // plane strides in elements
    // dynamic LDS (58 / 72 KB in the split form: two workgroups per CU): the patch planes, the filter tiles of ONE FILTER ROW (three taps x
    // planes), the rows' output offsets.  A stage = one filter row of one chunk: 36 TMB MFMAs per wavefront between two barriers; the filter
    // tiles are the kernel's L2 traffic (37 KB per stage), so 128-pixel blocks halve it per FLOP.
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    u16* const Ap = reinterpret_cast<u16*>(s_raw);                                     // [NS][PATCH_PIX][LDH]
    u16* const Bs = Ap + NS * APL;                                                     // [3 taps][NS][BN][LDH]
    unsigned* const s_out = reinterpret_cast<unsigned*>(Bs + 3 * NS * BPL);            // [BM] byte offset of each row's output pixel (OOB: none)
    float (*const s_stat)[2][BN] = reinterpret_cast<float (*)[2][BN]>(Bs);
    static_assert(sizeof(u16) * BPL >= sizeof(float) * T::WAVES_M * 2 * BN, "statistics scratch fits a filter plane");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = lid % p.ntiles, mt = lid / p.ntiles;
    const int n0 = nt * BN;
    const GatherSrc& g = p.g;
    const int TH = geo.TH, TW = geo.TW, PW = TW + 2, PH = TH + 2;
    const int per_img = geo.tiles_y * geo.tiles_x;
    const float rcp_pw = 1.0f / (float)PW, rcp_tw = 1.0f / (float)TW;
    const int img = small_div(mt, 1.0f / (float)per_img), tr = mt - img * per_img;
    const int tyi = small_div(tr, 1.0f / (float)geo.tiles_x);
    const int ty0 = tyi * TH, tx0 = (tr - tyi * geo.tiles_x) * TW;

    if (tid < BM) {
        const int pix = patch_pixel(tid);
        const int py = small_div(pix, rcp_tw), px = pix - py * TW;
        const int y = ty0 + py, x = tx0 + px;
        s_out[tid] = (py < TH && y < p.Hd && x < p.Wd) ? (unsigned)((img * p.Hd + y) * p.Wd + x) * (unsigned)(p.Cd * 4) : OOB;
    }
    // patch staging: thread -> (patch pixel pp0 + 32 j, 4 channels c4); a pixel outside the image reads as zero (the padding)
    const int c4 = tid & 7, pp0 = tid >> 3;
    unsigned aoff[C::NJ];
#pragma unroll
    for (int j = 0; j < C::NJ; ++j) {
        const int pp = pp0 + 32 * j;
        const int ppy = small_div(pp, rcp_pw), ppx = pp - ppy * PW;
        int y = ty0 - 1 + ppy, x = tx0 - 1 + ppx;
        if (geo.refl == 1) {                                      // reflection padding: the ring of the MAP is its row / column 1, H - 2 / W - 2
            y = y < 0 ? -y : (y >= g.Hs ? 2 * g.Hs - 2 - y : y);  // (rows further out belong to dropped outputs of a ragged block: zero if past the map)
            x = x < 0 ? -x : (x >= g.Ws ? 2 * g.Ws - 2 - x : x);
        }
        const bool ok = pp < PH * PW && (unsigned)y < (unsigned)g.Hs && (unsigned)x < (unsigned)g.Ws;
        aoff[j] = ok ? (unsigned)((((img * g.Hs + y) * g.Ws + x) * g.C1 + c4 * 4) * 4) : OOB;
    }
    const unsigned bytes1 = (unsigned)((size_t)g.B * g.Hs * g.Ws * g.C1 * 4);
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(g.x1, bytes1);
    const unsigned plane_bytes = (unsigned)((size_t)p.Np_all * p.Kstride * 2);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(w16, (unsigned)((size_t)(p.n_begin + p.n_count) * p.Kstride * 2) + (NS - 1) * plane_bytes);
    const int bn = tid >> 2, bcb = tid & 3;                       // filter tile: row (output) and 16-byte piece (8 channels)
    const unsigned boff = (n0 + bn < p.n_count) ? (unsigned)(((p.n_begin + n0 + bn) * p.Kstride + bcb * 8) * 2) : OOB;

    const int nchunks = p.Kp / CKT;
    auto issueA = [&](f32x4 (&ra)[C::NJ], int chunk) {
#pragma unroll
        for (int j = 0; j < C::NJ; ++j) ra[j] = buf_load4s(rs1, aoff[j], chunk * CKT * 4);
    };
    auto storeA = [&](const f32x4 (&ra)[C::NJ]) {
#pragma unroll
        for (int j = 0; j < C::NJ; ++j) {
            const int pp = pp0 + 32 * j;
            if (pp < PATCH_PIX) {
                u16* const dst = Ap + pp * LDH + (((c4 >> 1) ^ ((pp >> 2) & 3)) * 8) + (c4 & 1) * 4;
                if constexpr (NS == 1) {
                    *reinterpret_cast<u32x2*>(dst) = pack_bf16x4(ra[j]);
                } else {
                    const bf16x4 h = __builtin_convertvector(ra[j], bf16x4);
                    const f32x4 r1 = ra[j] - __builtin_convertvector(h, f32x4);            // exact
                    const bf16x4 m = __builtin_convertvector(r1, bf16x4);
                    const f32x4 r2 = r1 - __builtin_convertvector(m, f32x4);               // exact
                    *reinterpret_cast<u32x2*>(dst) = __builtin_bit_cast(u32x2, h);
                    *reinterpret_cast<u32x2*>(dst + APL) = __builtin_bit_cast(u32x2, m);
                    *reinterpret_cast<u32x2*>(dst + 2 * APL) = pack_bf16x4(r2);
                }
            }
        }
    };
    // stage s = chunk * 3 + ky: the filter tiles of taps (ky, 0..2) x planes
    auto issueB = [&](f32x4 (&rb)[3][NS], int s) {
        const int chunk = s / 3, ky = s - chunk * 3;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int kb = ((ky * 3 + kx) * p.Kp + chunk * CKT) * 2;
#pragma unroll
            for (int q = 0; q < NS; ++q) rb[kx][q] = buf_load4s(rsw, boff, kb + q * (int)plane_bytes);
        }
    };
    auto storeB = [&](const f32x4 (&rb)[3][NS]) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int q = 0; q < NS; ++q) *reinterpret_cast<f32x4*>(Bs + (kx * NS + q) * BPL + bn * LDH + ((bcb ^ ((bn >> 2) & 3)) * 8)) = rb[kx][q];
    };

    const int wm0 = (wave >> 1) * (32 * TMB), wn0 = (wave & 1) * 32;
    typename T::AccT acc[TMB][1], mid[TMB], low[TMB];
#pragma unroll
    for (int i = 0; i < TMB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][0][r] = 0.f; mid[i][r] = 0.f; low[i][r] = 0.f; }
    const int frow = lane & 31;
    // this lane's A rows = output pixels (py, px) of the block -> patch pixel (py + dy, px + dx), dy / dx = the tap (mirrored for the data gradient)
    int prow[TMB];
    unsigned rmask[TMB];                                         // reflection adjoint: bit 0 / 1 this row's pixel is in map row 1 / H - 2, bit 2 / 3 in column 1 / W - 2
#pragma unroll
    for (int i = 0; i < TMB; ++i) {
        const int pix = patch_pixel(wm0 + 32 * i + frow);
        const int py = small_div(pix, rcp_tw), px = pix - py * TW;
        prow[i] = py < TH ? py * PW + px : 0;                    // rows past the block multiply pixel 0 (their results are dropped)
        const int y = ty0 + py, x = tx0 + px;
        rmask[i] = (ADJ && py < TH) ? (unsigned)(y == 1) | ((unsigned)(y == p.Hd - 2) << 1) | ((unsigned)(x == 1) << 2) | ((unsigned)(x == p.Wd - 2) << 3) : 0u;
    }
    const int fslot = lane >> 5;                                  // which 16-byte half of a k-step this lane holds
    const int brow = wn0 + frow;
    const u16* const b_lane0 = Bs + brow * LDH + ((fslot ^ ((brow >> 2) & 3)) * 8);            // k-step 0 / 1 of this lane's filter row
    const u16* const b_lane1 = Bs + brow * LDH + (((2 + fslot) ^ ((brow >> 2) & 3)) * 8);
    const bool fwd = g.sign > 0;
    // The adjoint of reflection padding (geo.refl == 2; the decoder's 3x3 convolutions, reference layers.py Conv3x3 / ReflectionPad2d(1)).
    // Forward, y[q] = sum_k W_k x[refl(q + k - 1)]: the ring pixel -1 IS pixel 1, H is H - 2.  So dx[p] = sum_k W_k^T (sum over the q with
    // refl(q + k - 1) = p of dy[q]): the zero-padded data gradient (q = p + 1 - k where that is inside the map) plus, for p in row 1, the
    // q = -k of filter row k = 0 (q = 0: "source one row UP" = the fragment the mirrored filter row 2 reads), for p in row H - 2 filter row 2
    // with the source one row down, the same per column, and the four products of both for the pixels (1 | H - 2, 1 | W - 2).  Each extra
    // term is an MFMA group of a filter tap with the patch view of the OPPOSITE tap, its A rows zeroed except in the lanes of that row /
    // column: only the blocks that touch rows 1 / H - 2 or columns 1 / W - 2 run them (block-uniform branches), after the stage's own steps.
    const bool has_y1 = ADJ && ty0 <= 1 && ty0 + TH > 1, has_yl = ADJ && ty0 <= p.Hd - 2 && ty0 + TH > p.Hd - 2;
    const bool has_x1 = ADJ && tx0 <= 1 && tx0 + TW > 1, has_xl = ADJ && tx0 <= p.Wd - 2 && tx0 + TW > p.Wd - 2;
    auto extras = [&](int ky) {
        auto afr = [&](int t, int sky, int skx, int ks, bf16x8 (&a)[NS], bool keep) {          // the view tap (sky, skx) of the data gradient reads
            const int row = prow[t] + (2 - sky) * PW + 2 - skx;
            const u16* const ap = Ap + row * LDH + (((2 * ks + fslot) ^ ((row >> 2) & 3)) * 8);
            const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(ap + q * APL);
                a[q] = keep ? v : zero;
            }
        };
        auto mma = [&](int t, const bf16x8 (&a)[NS], const bf16x8 (&b)[NS]) {
            if constexpr (NS == 1) {
                acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[t][0], 0, 0, 0);
            } else {
                low[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], low[t], 0, 0, 0);
                mid[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], mid[t], 0, 0, 0);
                acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[t][0], 0, 0, 0);
                low[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], low[t], 0, 0, 0);
                mid[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], mid[t], 0, 0, 0);
                low[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], low[t], 0, 0, 0);
            }
        };
        const bool yrow = (ky == 0 && has_y1) || (ky == 2 && has_yl);
        const unsigned ybit = ky == 0 ? 1u : 2u;
        const int sky = 2 - ky;                                                               // the opposite filter row's view
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 b[3][NS];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const u16* const bp = (ks ? b_lane1 : b_lane0) + kx * NS * BPL;
#pragma unroll
                for (int q = 0; q < NS; ++q) b[kx][q] = *reinterpret_cast<const bf16x8*>(bp + q * BPL);
            }
#pragma unroll
            for (int t = 0; t < TMB; ++t) {
                const unsigned m = rmask[t];
                bf16x8 a[NS];
                if (has_x1) { afr(t, ky, 2, ks, a, (m & 4u) != 0); mma(t, a, b[0]); }
                if (has_xl) { afr(t, ky, 0, ks, a, (m & 8u) != 0); mma(t, a, b[2]); }
                if (yrow) {
                    const bool my = (m & ybit) != 0;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) { afr(t, sky, kx, ks, a, my); mma(t, a, b[kx]); }
                    if (has_x1) { afr(t, sky, 2, ks, a, my && (m & 4u)); mma(t, a, b[0]); }
                    if (has_xl) { afr(t, sky, 0, ks, a, my && (m & 8u)); mma(t, a, b[2]); }
                }
            }
        }
    };
    // A stage's six steps (three taps x two 16-deep k-steps), software-pipelined by hand: the fragments of step i + 1 are read from LDS while
    // the MFMAs of step i run (left to itself the compiler waits for each ds_read right in front of the MFMA that uses it).
    auto compute = [&](int ky) {
        const int rsh = (fwd ? ky : 2 - ky) * PW + (fwd ? 0 : 2);                          // patch row shift of tap (ky, 0); tap kx is dxr rows further
        const int dxr = fwd ? 1 : -1;
        bf16x8 fa[2][TMB][NS], fb[2][NS];
        auto load = [&](int i, bf16x8 (&a)[TMB][NS], bf16x8 (&b)[NS]) {
            const int kx = i >> 1, ks = i & 1;
            const u16* const bp = (ks ? b_lane1 : b_lane0) + kx * NS * BPL;
#pragma unroll
            for (int q = 0; q < NS; ++q) b[q] = *reinterpret_cast<const bf16x8*>(bp + q * BPL);
#pragma unroll
            for (int t = 0; t < TMB; ++t) {
                const int row = prow[t] + rsh + kx * dxr;
                const u16* const ap = Ap + row * LDH + (((2 * ks + fslot) ^ ((row >> 2) & 3)) * 8);
#pragma unroll
                for (int q = 0; q < NS; ++q) a[t][q] = *reinterpret_cast<const bf16x8*>(ap + q * APL);
            }
        };
        load(0, fa[0], fb[0]);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i + 1 < 6) load(i + 1, fa[(i + 1) & 1], fb[(i + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8(&b)[NS] = fb[i & 1];
#pragma unroll
            for (int t = 0; t < TMB; ++t) {
                const bf16x8(&a)[NS] = fa[i & 1][t];
                if constexpr (NS == 1) {
                    acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[t][0], 0, 0, 0);
                } else {
                    low[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], low[t], 0, 0, 0);
                    mid[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], mid[t], 0, 0, 0);
                    acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[t][0], 0, 0, 0);
                    low[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], low[t], 0, 0, 0);
                    mid[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], mid[t], 0, 0, 0);
                    low[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], low[t], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // The filter tiles are the kernel's traffic, all of it from L2: the tiles of stage s + 1 (and, in the 64-pixel form, s + 2) are in flight
    // in registers while stage s is multiplied.
    constexpr bool TWO = TMB == 1;                               // (the 128-pixel form has no registers for a second filter stage)
    f32x4 ra[C::NJ], rb0[3][NS], rb1[TWO ? 3 : 1][NS];
    const int total = nchunks * 3;
    issueA(ra, 0);
    issueB(rb0, 0);
    storeA(ra);
    storeB(rb0);
    if (total > 1) issueB(rb0, 1);
    if constexpr (TWO) {
        if (total > 2) issueB(rb1, 2);
    }
    __syncthreads();
    int ky = 0, chunk = 0;
    constexpr int diag = MCAV_PATCH_DIAG;
    auto stage = [&](int s, auto& rb_next) {                      // rb_next holds stage s + 1 and is re-issued for the stage after those in flight
        const bool more = chunk + 1 < nchunks;
        if (ky == 0 && more && !(diag & 4)) issueA(ra, chunk + 1);               // lands during the chunk's three stages
        if (!(diag & 8)) compute(ky);
        if constexpr (ADJ) {
            if (has_y1 | has_yl | has_x1 | has_xl) extras(ky);
        }
        if (!(diag & 2)) __syncthreads();
        if (ky == 2 && more && !(diag & 4)) storeA(ra);
        if (s + 1 < total) {
            if (!(diag & 1)) {
                storeB(rb_next);
                if (s + (TWO ? 3 : 2) < total) issueB(rb_next, s + (TWO ? 3 : 2));
            }
            if (!(diag & 2)) __syncthreads();
        }
        if (++ky == 3) { ky = 0; ++chunk; }
    };
    if constexpr (TWO) {
        for (int s = 0; s < total; s += 2) {
            stage(s, rb0);
            if (s + 1 < total) stage(s + 1, rb1);
        }
    } else {
        for (int s = 0; s < total; ++s) stage(s, rb0);
    }
    if constexpr (NS > 1) {
#pragma unroll
        for (int t = 0; t < TMB; ++t) acc[t][0] += mid[t] + low[t];
    }
    if (diag & 16) return;                                        // (timing only: no epilogue)
    igemm_epilogue_lean<T>(p, acc, s_out, s_stat, tid, wm0, wn0, n0, mt);
}

template <int NS, int TMB>
constexpr size_t patch_lds_bytes() { return sizeof(u16) * (size_t)NS * (PatchCfg<TMB>::PIX + 3 * 64) * 32 + sizeof(unsigned) * 64 * TMB; }

// ------------------------------------------------------------------------------------------------ patch kernel, twelve-wavefront form (round 4)
// What the weight-gradient kernel below showed: three wavefronts per SIMD on ONE workgroup per CU hide each other's waits where two workgroups of
// four did not.  The same block structure as conv3x3_patch_kernel -- a 64-pixel block's patch in LDS per 32-channel chunk, a stage = one filter
// row -- with THREE blocks per workgroup (192 GEMM rows x 64 outputs): twelve wavefronts, each one 32 x 32 tile of one block; the filter tiles
// of a stage (37 KB in the split form) serve three blocks (a third of the L2 stream per FLOP of the 64-pixel form) and are staged by 768 threads
// (12 registers per thread and stage in flight instead of 36) into a TWO-stage ring, so a stage is one barrier (two where the next chunk's
// patch replaces the current one).  Same epilogue (row offsets in s_out, statistics slab: one row per workgroup = three blocks of one group).
struct P3Tile {
    static constexpr int BM = 192, BN = 64, MF = 32, ACC = 16, TM = 1, TN = 1, WAVES_M = 6, WAVES_N = 2;
    using AccT = f32x16;
};
constexpr int P3_THREADS = 768, P3_SUBS = 3, P3_PIX = 110;
template <int NS> constexpr size_t patch3_lds_bytes() { return sizeof(u16) * (size_t)NS * (P3_SUBS * P3_PIX + 2 * 3 * 64) * 32 + sizeof(unsigned) * 192; }
#ifndef MCAV_PATCH3_DIAG
#define MCAV_PATCH3_DIAG 0
#endif

template <int NS, bool ADJ>
__global__ __launch_bounds__(P3_THREADS, 1) void conv3x3_patch3_kernel(IgemmParams p, const u16* __restrict__ w16, PatchGeo geo) {
    using T = P3Tile;
    constexpr int BN = 64, CKT = 32, LDH = 32, PIX = P3_PIX;
    constexpr int APL = PIX * LDH, BPL = BN * LDH;               // plane strides in elements
    constexpr int ASUB = NS * APL, BSTAGE = 3 * NS * BPL;        // one block's patch planes; one ring stage of filter tiles
    constexpr int diag = MCAV_PATCH3_DIAG;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    u16* const Ap = reinterpret_cast<u16*>(s_raw);                                     // [3 blocks][NS][PIX][LDH]
    u16* const Bs = Ap + P3_SUBS * ASUB;                                               // [2 stages][3 taps][NS][BN][LDH]
    unsigned* const s_out = reinterpret_cast<unsigned*>(Bs + 2 * BSTAGE);              // [192] byte offset of each row's output pixel
    float (*const s_stat)[2][BN] = reinterpret_cast<float (*)[2][BN]>(Bs);
    static_assert(sizeof(u16) * BSTAGE >= sizeof(float) * T::WAVES_M * 2 * BN, "statistics scratch fits a ring stage");

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = wave >> 2, wm0 = sub * 64 + ((wave >> 1) & 1) * 32, wn0 = (wave & 1) * 32;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = lid % p.ntiles, mt = lid / p.ntiles;
    const int n0 = nt * BN;
    const GatherSrc& g = p.g;
    const int TH = geo.TH, TW = geo.TW, PW = TW + 2, PH = TH + 2;
    const int per_img = geo.tiles_y * geo.tiles_x, nblocks = g.B * per_img;
    const float rcp_pw = 1.0f / (float)PW, rcp_tw = 1.0f / (float)TW;
    // the workgroup's three blocks (scalars); a block past the last one reads and writes nothing
    int b_img[P3_SUBS], b_ty0[P3_SUBS], b_tx0[P3_SUBS];
#pragma unroll
    for (int sb = 0; sb < P3_SUBS; ++sb) {
        const int blk = mt * P3_SUBS + sb;
        const int img = blk / per_img, tr = blk - img * per_img, tyi = tr / geo.tiles_x;
        b_img[sb] = blk < nblocks ? img : -1;
        b_ty0[sb] = tyi * TH;
        b_tx0[sb] = (tr - tyi * geo.tiles_x) * TW;
    }
    auto pick = [&](const int (&v)[P3_SUBS], int sb) { return sb == 0 ? v[0] : (sb == 1 ? v[1] : v[2]); };

    if (tid < 192) {
        const int sb = tid >> 6;
        const int pix = patch_pixel(tid & 63);
        const int py = small_div(pix, rcp_tw), px = pix - py * TW;
        const int img = pick(b_img, sb), y = pick(b_ty0, sb) + py, x = pick(b_tx0, sb) + px;
        s_out[tid] = (img >= 0 && py < TH && y < p.Hd && x < p.Wd) ? (unsigned)((img * p.Hd + y) * p.Wd + x) * (unsigned)(p.Cd * 4) : OOB;
    }
    // patch staging: item = (block, patch pixel, 4 channels): 3 x 110 x 8 items, four per thread
    constexpr int NJ = (P3_SUBS * PIX * 8 + P3_THREADS - 1) / P3_THREADS;
    unsigned aoff[NJ], adst[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int item = tid + P3_THREADS * j;
        const int sb = item / (PIX * 8), rem = item - sb * (PIX * 8), pp = rem >> 3, c4 = rem & 7;
        const int ppy = small_div(pp, rcp_pw), ppx = pp - ppy * PW;
        const int sbc = sb < P3_SUBS ? sb : 0;
        int y = pick(b_ty0, sbc) - 1 + ppy, x = pick(b_tx0, sbc) - 1 + ppx;
        if (geo.refl == 1) {
            y = y < 0 ? -y : (y >= g.Hs ? 2 * g.Hs - 2 - y : y);
            x = x < 0 ? -x : (x >= g.Ws ? 2 * g.Ws - 2 - x : x);
        }
        const int img = pick(b_img, sbc);
        const bool ok = sb < P3_SUBS && img >= 0 && pp < PH * PW && (unsigned)y < (unsigned)g.Hs && (unsigned)x < (unsigned)g.Ws;
        aoff[j] = ok ? (unsigned)((((img * g.Hs + y) * g.Ws + x) * g.C1 + c4 * 4) * 4) : OOB;
        adst[j] = sb < P3_SUBS ? (unsigned)(sb * ASUB + pp * LDH + (((c4 >> 1) ^ ((pp >> 2) & 3)) * 8) + (c4 & 1) * 4) : 0xffffffffu;
    }
    const unsigned bytes1 = (unsigned)((size_t)g.B * g.Hs * g.Ws * g.C1 * 4);
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(g.x1, bytes1);
    const unsigned plane_bytes = (unsigned)((size_t)p.Np_all * p.Kstride * 2);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(w16, (unsigned)((size_t)(p.n_begin + p.n_count) * p.Kstride * 2) + (NS - 1) * plane_bytes);
    // filter staging: item = (tap of the row, plane, output row, 16-byte piece): 3 NS x 256 items, NS per thread
    const int bn = (tid & 255) >> 2, bcb = tid & 3, bcombo = tid >> 8;        // this thread's first item: combo = bcombo + 3 j = kx * NS + q
    const unsigned boff = (n0 + bn < p.n_count) ? (unsigned)(((p.n_begin + n0 + bn) * p.Kstride + bcb * 8) * 2) : OOB;
    const unsigned bdst = (unsigned)(bn * LDH + ((bcb ^ ((bn >> 2) & 3)) * 8));

    const int nchunks = p.Kp / CKT;
    auto issueA = [&](f32x4 (&ra)[NJ], int chunk) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) ra[j] = buf_load4s(rs1, aoff[j], chunk * CKT * 4);
    };
    auto storeA = [&](const f32x4 (&ra)[NJ]) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (adst[j] != 0xffffffffu) {
                u16* const dst = Ap + adst[j];
                if constexpr (NS == 1) {
                    *reinterpret_cast<u32x2*>(dst) = pack_bf16x4(ra[j]);
                } else {
                    const bf16x4 h = __builtin_convertvector(ra[j], bf16x4);
                    const f32x4 r1 = ra[j] - __builtin_convertvector(h, f32x4);            // exact
                    const bf16x4 m = __builtin_convertvector(r1, bf16x4);
                    const f32x4 r2 = r1 - __builtin_convertvector(m, f32x4);               // exact
                    *reinterpret_cast<u32x2*>(dst) = __builtin_bit_cast(u32x2, h);
                    *reinterpret_cast<u32x2*>(dst + APL) = __builtin_bit_cast(u32x2, m);
                    *reinterpret_cast<u32x2*>(dst + 2 * APL) = pack_bf16x4(r2);
                }
            }
        }
    };
    // stage s = chunk * 3 + ky: the filter tiles of taps (ky, 0..2) x planes
    auto issueB = [&](f32x4 (&rb)[NS], int s) {
        const int chunk = s / 3, ky = s - chunk * 3;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int combo = bcombo + 3 * j, kx = combo / NS, q = combo - kx * NS;
            rb[j] = buf_load4s(rsw, boff, ((ky * 3 + kx) * p.Kp + chunk * CKT) * 2 + q * (int)plane_bytes);
        }
    };
    auto storeB = [&](const f32x4 (&rb)[NS], int ring) {
#pragma unroll
        for (int j = 0; j < NS; ++j) *reinterpret_cast<f32x4*>(Bs + ring * BSTAGE + (bcombo + 3 * j) * BPL + bdst) = rb[j];
    };

    typename T::AccT acc[1][1], mid, low;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][0][r] = 0.f; mid[r] = 0.f; low[r] = 0.f; }
    const int frow = lane & 31;
    int prow;
    unsigned rmask = 0;
    int my_ty0 = pick(b_ty0, sub), my_tx0 = pick(b_tx0, sub);
    {
        const int pix = patch_pixel((wm0 & 63) + frow);
        const int py = small_div(pix, rcp_tw), px = pix - py * TW;
        prow = py < TH ? py * PW + px : 0;
        const int y = my_ty0 + py, x = my_tx0 + px;
        rmask = (ADJ && py < TH) ? (unsigned)(y == 1) | ((unsigned)(y == p.Hd - 2) << 1) | ((unsigned)(x == 1) << 2) | ((unsigned)(x == p.Wd - 2) << 3) : 0u;
    }
    const u16* const Asub = Ap + sub * ASUB;
    const int fslot = lane >> 5;
    const int brow = wn0 + frow;
    const unsigned b_lane0 = (unsigned)(brow * LDH + ((fslot ^ ((brow >> 2) & 3)) * 8));
    const unsigned b_lane1 = (unsigned)(brow * LDH + (((2 + fslot) ^ ((brow >> 2) & 3)) * 8));
    const bool fwd = g.sign > 0;
    const bool has_y1 = ADJ && my_ty0 <= 1 && my_ty0 + TH > 1, has_yl = ADJ && my_ty0 <= p.Hd - 2 && my_ty0 + TH > p.Hd - 2;
    const bool has_x1 = ADJ && my_tx0 <= 1 && my_tx0 + TW > 1, has_xl = ADJ && my_tx0 <= p.Wd - 2 && my_tx0 + TW > p.Wd - 2;
    auto mma = [&](const bf16x8 (&a)[NS], const bf16x8 (&b)[NS]) {
        if constexpr (NS == 1) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[0][0], 0, 0, 0);
        } else {
            low = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], low, 0, 0, 0);
            mid = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], mid, 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[0][0], 0, 0, 0);
            low = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], low, 0, 0, 0);
            mid = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], mid, 0, 0, 0);
            low = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], low, 0, 0, 0);
        }
    };
    // the reflection adjoint's extra terms (conv3x3_patch_kernel): border blocks only, after the stage's own steps
    auto extras = [&](int ky, const u16* Bst) {
        auto afr = [&](int sky, int skx, int ks, bf16x8 (&a)[NS], bool keep) {
            const int row = prow + (2 - sky) * PW + 2 - skx;
            const u16* const ap = Asub + row * LDH + (((2 * ks + fslot) ^ ((row >> 2) & 3)) * 8);
            const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(ap + q * APL);
                a[q] = keep ? v : zero;
            }
        };
        const bool yrow = (ky == 0 && has_y1) || (ky == 2 && has_yl);
        const unsigned ybit = ky == 0 ? 1u : 2u;
        const int sky = 2 - ky;
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 b[3][NS];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int q = 0; q < NS; ++q) b[kx][q] = *reinterpret_cast<const bf16x8*>(Bst + (ks ? b_lane1 : b_lane0) + (kx * NS + q) * BPL);
            const unsigned m = rmask;
            bf16x8 a[NS];
            if (has_x1) { afr(ky, 2, ks, a, (m & 4u) != 0); mma(a, b[0]); }
            if (has_xl) { afr(ky, 0, ks, a, (m & 8u) != 0); mma(a, b[2]); }
            if (yrow) {
                const bool my = (m & ybit) != 0;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) { afr(sky, kx, ks, a, my); mma(a, b[kx]); }
                if (has_x1) { afr(sky, 2, ks, a, my && (m & 4u)); mma(a, b[0]); }
                if (has_xl) { afr(sky, 0, ks, a, my && (m & 8u)); mma(a, b[2]); }
            }
        }
    };
    // a stage's six steps (three taps x two 16-deep k-steps), fragments of step i + 1 read while the MFMAs of step i run
    auto compute = [&](int ky, const u16* Bst) {
        const int rsh = (fwd ? ky : 2 - ky) * PW + (fwd ? 0 : 2);
        const int dxr = fwd ? 1 : -1;
        bf16x8 fa[2][NS], fb[2][NS];
        auto load = [&](int i, bf16x8 (&a)[NS], bf16x8 (&b)[NS]) {
            const int kx = i >> 1, ks = i & 1;
            const u16* const bp = Bst + (ks ? b_lane1 : b_lane0) + kx * NS * BPL;
#pragma unroll
            for (int q = 0; q < NS; ++q) b[q] = *reinterpret_cast<const bf16x8*>(bp + q * BPL);
            const int row = prow + rsh + kx * dxr;
            const u16* const ap = Asub + row * LDH + (((2 * ks + fslot) ^ ((row >> 2) & 3)) * 8);
#pragma unroll
            for (int q = 0; q < NS; ++q) a[q] = *reinterpret_cast<const bf16x8*>(ap + q * APL);
        };
        load(0, fa[0], fb[0]);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i + 1 < 6) load(i + 1, fa[(i + 1) & 1], fb[(i + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (!(diag & 8)) mma(fa[i & 1], fb[i & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    f32x4 ra[NJ], rbn[NS];
    const int total = nchunks * 3;
    issueA(ra, 0);
    issueB(rbn, 0);
    storeA(ra);
    storeB(rbn, 0);
    if (total > 1) issueB(rbn, 1);
    __syncthreads();
    int ky = 0, chunk = 0;
    for (int s = 0; s < total; ++s) {
        const bool more = chunk + 1 < nchunks;
        const u16* const Bst = Bs + (s & 1) * BSTAGE;
        if (ky == 0 && more && !(diag & 4)) issueA(ra, chunk + 1);               // lands during the chunk's three stages
        compute(ky, Bst);
        if constexpr (ADJ) {
            if (has_y1 | has_yl | has_x1 | has_xl) extras(ky, Bst);
        }
        if (s + 1 < total && !(diag & 1)) {                                      // the next stage's filter tiles into the other ring stage (last read before the previous barrier)
            storeB(rbn, (s + 1) & 1);
            if (s + 2 < total) issueB(rbn, s + 2);
        }
        if (ky == 2 && more && !(diag & 4)) {                                    // the next chunk's patch replaces this one: everybody has to be done reading it
            __syncthreads();
            storeA(ra);
        }
        __syncthreads();
        if (++ky == 3) { ky = 0; ++chunk; }
    }
    if constexpr (NS > 1) acc[0][0] += mid + low;
    if (diag & 16) return;
    igemm_epilogue_lean<T>(p, acc, s_out, s_stat, tid, wm0, wn0, n0, mt);
}

// ------------------------------------------------------------------------------------------------ patch kernel, second form (round 4)
// What the counters said about conv3x3_patch_kernel (profiles/r03_pmc_patch_kernel.txt): its MFMA pipe is busy a third of
// the time because (a) every 64- / 128-pixel block streams the whole filter from L2 (20 B / clk / CU at most, latency-bound by the bytes a
// wavefront's registers keep in flight), (b) a stage is 36-72 MFMAs between two barriers, (c) set-up and epilogue of a 10 us workgroup.
// This form gives a workgroup SB sub-blocks of 64 output pixels (each a TH x TW block of one image: any of the 4 x 16 / 3 x 20 / 6 x 10 /
// 8 x 8 shapes the planner picks per map) x 64 outputs = a 128- or 256-row tile -- the filter stream per FLOP falls 2-4x and the 6x20 / 12x40
// maps (60-pixel blocks) fill whole MFMA tiles with blocks of several images --, runs ONE workgroup per CU (one wavefront per SIMD, the whole
// register file: 64 x 64 per wavefront at SB = 4, twelve 32x32 accumulators) and moves the filter tiles by LDS-DMA (global_load_lds_dwordx4:
// no registers, no ds_write pass) into a two-stage ring, so that stage s + 1 lands while stage s is multiplied and a stage is ONE barrier.
// LDS image of a filter tile = the first kernel's (64-byte rows, 16-byte slots XOR-ed by (row >> 2) & 3), obtained by permuting the SOURCE
// slot each lane fetches (an LDS-DMA instruction writes its 64 x 16 bytes linearly).  The patch still goes through registers (it is
// converted / split on the way), once per 32-channel chunk.  Same epilogue, same statistics slab (rows = workgroup rows).
// Timing experiments only, WRONG results (make variant NAME=p2d1 FLAGS=-DMCAV_PATCH2_DIAG=1): 1 no filter DMA in the loop, 2 no MFMAs, 4 no
// epilogue, 8 no patch staging in the loop.  Zero in the shipped library.
#ifndef MCAV_PATCH2_DIAG
#define MCAV_PATCH2_DIAG 0
#endif

template <int SB, int BN_>
struct Patch2Cfg {
    static constexpr int BM = 64 * SB, BN = BN_;
    static constexpr int PIX = 110;                       // patch pixels of one sub-block (PatchCfg<1>::PIX)
    static constexpr int VP = SB * PIX;                   // patch pixels of the workgroup
    static constexpr int NJ = (VP + 31) / 32;             // ... staged per thread
    // four wavefronts stacked along M: 64 x 64 each (SB = 4: a wavefront = a sub-block) or 32 x 64 (SB = 2: two wavefronts per sub-block)
    // BN = 32 (a 128 x 32 tile, 79 KB of LDS): TWO workgroups per CU, so that one's set-up, epilogue and DMA waits run beside the other's MFMAs
    using T = typename std::conditional<SB == 4, BTile<256, BN_, 64, BN_, 32, 32>, BTile<128, BN_, 32, BN_, 32, 32>>::type;
};

template <int NS, int SB, int BN>
constexpr size_t patch2_lds_bytes() { return sizeof(u16) * (size_t)NS * (SB * 110 + 2 * 3 * BN) * 32 + sizeof(unsigned) * 64 * SB; }

template <int NS, int SB, int BN>
__global__ __launch_bounds__(256, BN == 32 ? 2 : 1) void conv3x3_patch2_kernel(IgemmParams p, const u16* __restrict__ w16, PatchGeo geo) {
    using C = Patch2Cfg<SB, BN>;
    using T = typename C::T;
    constexpr int BM = C::BM, CKT = 32, LDH = 32, PIX = C::PIX, TM = T::TM, TN = T::TN;
    constexpr int APL = PIX * LDH;                               // one plane of one sub-block's patch, in elements
    constexpr int BPL = BN * LDH;                                // one (tap, plane) filter tile
    constexpr int BSTAGE = 3 * NS * BPL;                         // a stage: three taps x planes
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    u16* const Ap = reinterpret_cast<u16*>(s_raw);                                     // [SB][NS][PIX][LDH]
    u16* const Bs = Ap + SB * NS * APL;                                                // [2][3 taps][NS][BN][LDH]
    unsigned* const s_out = reinterpret_cast<unsigned*>(Bs + 2 * BSTAGE);              // [BM] byte offset of each row's output pixel (OOB: none)
    float (*const s_stat)[2][BN] = reinterpret_cast<float (*)[2][BN]>(Bs);
    static_assert(sizeof(u16) * BSTAGE >= sizeof(float) * T::WAVES_M * 2 * BN, "statistics scratch fits a filter stage");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = lid % p.ntiles, mt = lid / p.ntiles;
    const int n0 = nt * BN;
    const GatherSrc& g = p.g;
    const int TH = geo.TH, TW = geo.TW, PW = TW + 2, PH = TH + 2;
    const int per_img = geo.tiles_y * geo.tiles_x, nblk = g.B * per_img;

    // Set-up with few integer divisions (they are ~40 instructions each, and one wavefront per SIMD has nothing to hide them behind): the
    // sub-blocks' origins once per thread, the (row, column) of a patch pixel from a 110-entry LDS table.
    __shared__ int s_pp[C::PIX];
    if (tid < PIX) {
        const int ppy = tid / PW;
        s_pp[tid] = tid < PH * PW ? (ppy << 8) | (tid - ppy * PW) : -1;
    }
    int sb_base[SB], sb_y0[SB], sb_x0[SB];                       // image's first pixel index, block origin; base < 0: no such block
#pragma unroll
    for (int sb = 0; sb < SB; ++sb) {
        const int blk = mt * SB + sb;
        const int img = blk / per_img, tr = blk - img * per_img, tyi = tr / geo.tiles_x;
        sb_base[sb] = blk < nblk ? img : -1;
        sb_y0[sb] = tyi * TH;
        sb_x0[sb] = (tr - tyi * geo.tiles_x) * TW;
    }
    {   // rows -> output pixels (row = 64 sub-block + r; r -> pixel patch_pixel(r) of the sub-block)
        const int sbw = tid >> 6;
        if (sbw < SB) {
            unsigned o = OOB;
            int img = -1, y0 = 0, x0 = 0;
#pragma unroll
            for (int sb = 0; sb < SB; ++sb)
                if (sb == sbw) { img = sb_base[sb]; y0 = sb_y0[sb]; x0 = sb_x0[sb]; }
            if (img >= 0) {
                const int pix = patch_pixel(tid & 63);
                const int py = pix / TW, px = pix - py * TW;
                const int y = y0 + py, x = x0 + px;
                if (py < TH && y < p.Hd && x < p.Wd) o = (unsigned)((img * p.Hd + y) * p.Wd + x) * (unsigned)(p.Cd * 4);
            }
            s_out[tid] = o;
        }
    }
    __syncthreads();
    // patch staging: thread -> (virtual patch pixel pp0 + 32 j = sub-block x patch pixel, 4 channels c4); outside the image reads as zero
    const int c4 = tid & 7, pp0 = tid >> 3;
    unsigned aoff[C::NJ];
#pragma unroll
    for (int j = 0; j < C::NJ; ++j) {
        const int vp = pp0 + 32 * j;
        const int sbj = vp / PIX, pp = vp - sbj * PIX;           // (division by a constant)
        unsigned off = OOB;
        if (vp < C::VP) {
            int img = -1, y0 = 0, x0 = 0;
#pragma unroll
            for (int sb = 0; sb < SB; ++sb)
                if (sb == sbj) { img = sb_base[sb]; y0 = sb_y0[sb]; x0 = sb_x0[sb]; }
            const int yx = s_pp[pp];
            if (img >= 0 && yx >= 0) {
                const int y = y0 - 1 + (yx >> 8), x = x0 - 1 + (yx & 255);
                if ((unsigned)y < (unsigned)g.Hs && (unsigned)x < (unsigned)g.Ws) off = (unsigned)((((img * g.Hs + y) * g.Ws + x) * g.C1 + c4 * 4) * 4);
            }
        }
        aoff[j] = off;
    }
    const unsigned bytes1 = (unsigned)((size_t)g.B * g.Hs * g.Ws * g.C1 * 4);
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(g.x1, bytes1);
    const size_t plane_bytes = (size_t)p.Np_all * p.Kstride * 2;

    const int nchunks = p.Kp / CKT;
    auto issueA = [&](f32x4 (&ra)[C::NJ], int chunk) {
#pragma unroll
        for (int j = 0; j < C::NJ; ++j) ra[j] = buf_load4s(rs1, aoff[j], chunk * CKT * 4);
    };
    auto storeA = [&](const f32x4 (&ra)[C::NJ]) {
#pragma unroll
        for (int j = 0; j < C::NJ; ++j) {
            const int vp = pp0 + 32 * j;
            if (vp < C::VP) {
                const int sb = vp / PIX, pp = vp - sb * PIX;
                u16* const dst = Ap + sb * NS * APL + pp * LDH + (((c4 >> 1) ^ ((pp >> 2) & 3)) * 8) + (c4 & 1) * 4;
                if constexpr (NS == 1) {
                    *reinterpret_cast<u32x2*>(dst) = pack_bf16x4(ra[j]);
                } else {
                    const bf16x4 h = __builtin_convertvector(ra[j], bf16x4);
                    const f32x4 r1 = ra[j] - __builtin_convertvector(h, f32x4);            // exact
                    const bf16x4 m = __builtin_convertvector(r1, bf16x4);
                    const f32x4 r2 = r1 - __builtin_convertvector(m, f32x4);               // exact
                    *reinterpret_cast<u32x2*>(dst) = __builtin_bit_cast(u32x2, h);
                    *reinterpret_cast<u32x2*>(dst + APL) = __builtin_bit_cast(u32x2, m);
                    *reinterpret_cast<u32x2*>(dst + 2 * APL) = pack_bf16x4(r2);
                }
            }
        }
    };
    // Filter tiles of stage s = (chunk, ky) by LDS-DMA: 3 NS (tap, plane) tiles of BN rows x 64 bytes = 3 NS BN / 16 pieces of 1 KiB, wavefront
    // w issues pieces w, w + 4, ...  Piece q: tile q / (BN / 16), rows 16 (q % (BN / 16)) ..; lane -> row + (lane >> 2), LDS slot lane & 3, which
    // holds the logical slot (lane & 3) ^ ((row >> 2) & 3) of that row.  Rows past the launch's outputs re-read its last row (results dropped).
    const int lrow = lane >> 2, lsp = lane & 3;
    auto issueB = [&](int s) {
        const int chunk = s / 3, ky = s - chunk * 3;
        u16* const ring = Bs + (s & 1) * BSTAGE;
        constexpr int PPT = BN / 16, NPIECE = 3 * NS * PPT;          // pieces per tile, per stage
#pragma unroll
        for (int i = 0; i < (NPIECE + 3) / 4; ++i) {
            const int q = wave + 4 * i;
            if (NPIECE % 4 != 0 && q >= NPIECE) break;
            const int tp = q / PPT, kx = tp / NS, pl = tp - kx * NS;
            const int row = 16 * (q % PPT) + lrow;
            const int slot = lsp ^ ((row >> 2) & 3);
            int n = n0 + row;
            if (n >= p.n_count) n = p.n_count - 1;
            const char* src = reinterpret_cast<const char*>(w16) + pl * plane_bytes +
                              ((size_t)(p.n_begin + n) * p.Kstride + (size_t)((ky * 3 + kx) * p.Kp + chunk * CKT + slot * 8)) * 2;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(ring + tp * BPL + 16 * (q % PPT) * LDH), 16, 0, 0);
        }
    };

    const int wm0 = wave * (BM / 4), wn0 = 0;
    typename T::AccT acc[TM][TN], mid[NS > 1 ? TM : 1][NS > 1 ? TN : 1], low[NS > 1 ? TM : 1][NS > 1 ? TN : 1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.f;
                if constexpr (NS > 1) { mid[i][j][r] = 0.f; low[i][j][r] = 0.f; }
            }
    const int frow = lane & 31;
    // this lane's A rows = output pixels of its sub-block -> patch pixel (py + dy, px + dx), dy / dx = the tap (mirrored for the data gradient)
    const u16* abase[TM];
    int prow[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int R = wm0 + 32 * i + frow;
        const int sb = R >> 6;
        const int pix = patch_pixel(R & 63);
        const int py = pix / TW, px = pix - py * TW;
        prow[i] = py < TH ? py * PW + px : 0;                    // rows past the block multiply pixel 0 (their results are dropped)
        abase[i] = Ap + sb * NS * APL;
    }
    const int fslot = lane >> 5;                                  // which 16-byte half of a k-step this lane holds
    int boffs[TN][2];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int brow = 32 * j + frow;
        boffs[j][0] = brow * LDH + ((fslot ^ ((brow >> 2) & 3)) * 8);
        boffs[j][1] = brow * LDH + (((2 + fslot) ^ ((brow >> 2) & 3)) * 8);
    }
    const bool fwd = g.sign > 0;
    auto compute = [&](int ky, const u16* ring) {
        const int rsh = (fwd ? ky : 2 - ky) * PW + (fwd ? 0 : 2);                          // patch row shift of tap (ky, 0); tap kx is dxr rows further
        const int dxr = fwd ? 1 : -1;
        bf16x8 fa[2][TM][NS], fb[2][TN][NS];
        auto load = [&](int i, bf16x8 (&a)[TM][NS], bf16x8 (&b)[TN][NS]) {
            const int kx = i >> 1, ks = i & 1;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const u16* const bp = ring + kx * NS * BPL + boffs[j][ks];
#pragma unroll
                for (int q = 0; q < NS; ++q) b[j][q] = *reinterpret_cast<const bf16x8*>(bp + q * BPL);
            }
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const int row = prow[t] + rsh + kx * dxr;
                const u16* const ap = abase[t] + row * LDH + (((2 * ks + fslot) ^ ((row >> 2) & 3)) * 8);
#pragma unroll
                for (int q = 0; q < NS; ++q) a[t][q] = *reinterpret_cast<const bf16x8*>(ap + q * APL);
            }
        };
        load(0, fa[0], fb[0]);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i + 1 < 6) load(i + 1, fa[(i + 1) & 1], fb[(i + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const bf16x8(&a)[NS] = fa[i & 1][t];
                    const bf16x8(&b)[NS] = fb[i & 1][j];
                    if constexpr (NS == 1) {
                        acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[t][j], 0, 0, 0);
                    } else {
                        low[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], low[t][j], 0, 0, 0);
                        mid[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], mid[t][j], 0, 0, 0);
                        acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[t][j], 0, 0, 0);
                        low[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], low[t][j], 0, 0, 0);
                        mid[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], mid[t][j], 0, 0, 0);
                        low[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], low[t][j], 0, 0, 0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    f32x4 ra[C::NJ];
    const int total = nchunks * 3;
    issueA(ra, 0);
    issueB(0);
    storeA(ra);
    __syncthreads();                                             // (with an LDS-DMA in flight the compiler drains vmcnt here: stage 0 has landed)
    int ky = 0, chunk = 0;
    constexpr int diag = MCAV_PATCH2_DIAG;
    for (int s = 0; s < total; ++s) {
        const bool more = chunk + 1 < nchunks;
        if (s + 1 < total && !(diag & 1)) issueB(s + 1);         // into the ring slot stage s - 1 was read from (every wavefront is past its barrier)
        if (ky == 0 && more && !(diag & 8)) issueA(ra, chunk + 1);
        if (!(diag & 2)) compute(ky, Bs + (s & 1) * BSTAGE);
        __syncthreads();                                         // stage s + 1 has landed; everyone is done with stage s and (ky == 2) with the patch
        if (ky == 2 && more && !(diag & 8)) {
            storeA(ra);
            __syncthreads();
        }
        if (++ky == 3) { ky = 0; ++chunk; }
    }
    if (diag & 4) return;
    if constexpr (NS > 1) {
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[t][j] += mid[t][j] + low[t][j];
    }
    igemm_epilogue_lean<T>(p, acc, s_out, s_stat, tid, wm0, wn0, n0, mt);
}

// ------------------------------------------------------------------------------------------------ the same on the fp32 MFMA (mma = 0)
// conv3x3_patch_kernel's structure with fp32 operands and v_mfma_f32_32x32x2_f32: the default (fp32) path of the trunk's 3x3 stride-1
// zero-padded convolutions and their data gradients.  The table-driven kernel fetches the A operand once per tap through L2; with those
// fetches redirected to a few hot lines it ran 8-10 % faster (22 % on the 6x20 maps) -- here the patch of a
// 32-channel chunk is fetched once.  LDS: 128-byte rows (32 floats), the eight 16-byte slots of row r XOR-ed by (r >> 1) & 7 (a
// ds_read_b128 lane group of 16 rows distinct mod 16 covers all sixteen slots of the 256-byte bank row); a fragment read = four consecutive
// channels = the operands of four MFMAs (lane half h holds channels 8 g + 4 h + e of group g, e = the MFMA).  A stage = one filter row of a
// chunk: 48 TMB MFMAs (3072 TMB cycles) per wavefront between two barriers; 39 / 48 KB of LDS.
template <int TMB>
__global__ __launch_bounds__(256, TMB == 1 ? 4 : 3) void conv3x3_patch_f32_kernel(IgemmParams p, PatchGeo geo) {
    using C = PatchCfg<TMB>;
    using T = typename C::T;
    constexpr int BM = C::BM, BN = 64, CKT = 32, LDF = 32, PATCH_PIX = C::PIX;
    constexpr int BPL = BN * LDF;                                 // one tap's filter tile in floats
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    float* const Ap = reinterpret_cast<float*>(s_raw);                                  // [PATCH_PIX][LDF]
    float* const Bs = Ap + PATCH_PIX * LDF;                                             // [3 taps][BN][LDF]
    unsigned* const s_out = reinterpret_cast<unsigned*>(Bs + 3 * BPL);                 // [BM]
    float (*const s_stat)[2][BN] = reinterpret_cast<float (*)[2][BN]>(Bs);
    static_assert(sizeof(float) * BPL >= sizeof(float) * T::WAVES_M * 2 * BN, "statistics scratch fits a filter tile");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = lid % p.ntiles, mt = lid / p.ntiles;
    const int n0 = nt * BN;
    const GatherSrc& g = p.g;
    const int TH = geo.TH, TW = geo.TW, PW = TW + 2, PH = TH + 2;
    const int per_img = geo.tiles_y * geo.tiles_x;
    const int img = mt / per_img, tr = mt - img * per_img;
    const int ty0 = (tr / geo.tiles_x) * TH, tx0 = (tr % geo.tiles_x) * TW;
    if (tid < BM) {
        const int pix = patch_pixel(tid);
        const int py = pix / TW, px = pix - py * TW;
        const int y = ty0 + py, x = tx0 + px;
        s_out[tid] = (py < TH && y < p.Hd && x < p.Wd) ? (unsigned)((img * p.Hd + y) * p.Wd + x) * (unsigned)(p.Cd * 4) : OOB;
    }
    // patch staging: thread -> (patch pixel pp0 + 32 j, 4 channels c4)
    const int c4 = tid & 7, pp0 = tid >> 3;
    unsigned aoff[C::NJ];
#pragma unroll
    for (int j = 0; j < C::NJ; ++j) {
        const int pp = pp0 + 32 * j;
        const int ppy = pp / PW, ppx = pp - ppy * PW;
        const int y = ty0 - 1 + ppy, x = tx0 - 1 + ppx;
        const bool ok = pp < PH * PW && (unsigned)y < (unsigned)g.Hs && (unsigned)x < (unsigned)g.Ws;
        aoff[j] = ok ? (unsigned)((((img * g.Hs + y) * g.Ws + x) * g.C1 + c4 * 4) * 4) : OOB;
    }
    const unsigned bytes1 = (unsigned)((size_t)g.B * g.Hs * g.Ws * g.C1 * 4);
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(g.x1, bytes1);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(p.w, (unsigned)((size_t)(p.n_begin + p.n_count) * p.Kstride * 4));
    // filter tile of a tap: 64 rows x 8 slots of 16 bytes; thread -> pieces tid and tid + 256
    const int bn0 = tid >> 3, bsl = tid & 7;                      // rows bn0 and bn0 + 32, slot bsl
    unsigned boff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int bn = bn0 + 32 * h;
        boff[h] = (n0 + bn < p.n_count) ? (unsigned)(((p.n_begin + n0 + bn) * p.Kstride + bsl * 4) * 4) : OOB;
    }
    const int nchunks = p.Kp / CKT;
    auto issueA = [&](f32x4 (&ra)[C::NJ], int chunk) {
#pragma unroll
        for (int j = 0; j < C::NJ; ++j) ra[j] = buf_load4s(rs1, aoff[j], chunk * CKT * 4);
    };
    auto storeA = [&](const f32x4 (&ra)[C::NJ]) {
#pragma unroll
        for (int j = 0; j < C::NJ; ++j) {
            const int pp = pp0 + 32 * j;
            if (pp < PATCH_PIX) *reinterpret_cast<f32x4*>(Ap + pp * LDF + ((c4 ^ ((pp >> 1) & 7)) * 4)) = ra[j];
        }
    };
    auto issueB = [&](f32x4 (&rb)[3][2], int s) {
        const int chunk = s / 3, ky = s - chunk * 3;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int kb = ((ky * 3 + kx) * p.Kp + chunk * CKT) * 4;
#pragma unroll
            for (int h = 0; h < 2; ++h) rb[kx][h] = buf_load4s(rsw, boff[h], kb);
        }
    };
    auto storeB = [&](const f32x4 (&rb)[3][2]) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int bn = bn0 + 32 * h;
                *reinterpret_cast<f32x4*>(Bs + kx * BPL + bn * LDF + ((bsl ^ ((bn >> 1) & 7)) * 4)) = rb[kx][h];
            }
    };

    const int wm0 = (wave >> 1) * (32 * TMB), wn0 = (wave & 1) * 32;
    typename T::AccT acc[TMB][1];
#pragma unroll
    for (int i = 0; i < TMB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
    const int frow = lane & 31, fh = lane >> 5;
    int prow[TMB];
#pragma unroll
    for (int i = 0; i < TMB; ++i) {
        const int pix = patch_pixel(wm0 + 32 * i + frow);
        const int py = pix / TW, px = pix - py * TW;
        prow[i] = py < TH ? py * PW + px : 0;
    }
    const int brow = wn0 + frow;
    const bool fwd = g.sign > 0;
    // twelve steps per stage (three taps x four 8-channel groups); the fragments of step i + 1 are read while the four MFMAs of step i run
    auto compute = [&](int ky) {
        const int rsh = (fwd ? ky : 2 - ky) * PW + (fwd ? 0 : 2);
        const int dxr = fwd ? 1 : -1;
        f32x4 fa[2][TMB], fb[2];
        auto load = [&](int i, f32x4 (&a)[TMB], f32x4& b) {
            const int kx = i >> 2, kg = i & 3;
            const int slot = 2 * kg + fh;
            b = *reinterpret_cast<const f32x4*>(Bs + kx * BPL + brow * LDF + ((slot ^ ((brow >> 1) & 7)) * 4));
#pragma unroll
            for (int t = 0; t < TMB; ++t) {
                const int row = prow[t] + rsh + kx * dxr;
                a[t] = *reinterpret_cast<const f32x4*>(Ap + row * LDF + ((slot ^ ((row >> 1) & 7)) * 4));
            }
        };
        load(0, fa[0], fb[0]);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            if (i + 1 < 12) load(i + 1, fa[(i + 1) & 1], fb[(i + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < TMB; ++t)
                    acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i & 1][t][e], fb[i & 1][e], acc[t][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    f32x4 ra[C::NJ], rb[3][2];
    const int total = nchunks * 3;
    issueA(ra, 0);
    issueB(rb, 0);
    storeA(ra);
    storeB(rb);
    if (total > 1) issueB(rb, 1);
    __syncthreads();
    int ky = 0, chunk = 0;
    for (int s = 0; s < total; ++s) {
        const bool more = chunk + 1 < nchunks;
        if (ky == 0 && more) issueA(ra, chunk + 1);               // lands during the chunk's three stages
        compute(ky);
        __syncthreads();
        if (ky == 2 && more) storeA(ra);
        if (s + 1 < total) {
            storeB(rb);
            if (s + 2 < total) issueB(rb, s + 2);
            __syncthreads();
        }
        if (++ky == 3) { ky = 0; ++chunk; }
    }
    igemm_epilogue_lean<T>(p, acc, s_out, s_stat, tid, wm0, wn0, n0, mt);
}

template <int TMB>
constexpr size_t patch_f32_lds_bytes() { return sizeof(float) * (size_t)(PatchCfg<TMB>::PIX + 3 * 64) * 32 + sizeof(unsigned) * 64 * TMB; }

// The block shape for an H x W map: TH TW <= rows, (TH + 2)(TW + 2) <= pix, fewest wasted rows, then the smallest halo; 16-wide blocks read
// the patch without bank conflicts (worth some wasted rows).  Returns the cost (rows computed, weighted).
static double patch_block(int H, int W, int rows, int pix, int& TH, int& TW) {
    double best = 1e30;
    TH = 4; TW = 16;
    for (int th = 1; th <= 16; ++th)
        for (int tw = 4; tw <= 64; ++tw) {
            if (th * tw > rows || (th + 2) * (tw + 2) > pix) continue;
            const double tiles = (double)((H + th - 1) / th) * ((W + tw - 1) / tw);
            // a lane group = 16 consecutive pixels of the block: one pixel row of a 16-wide block (conflict-free); of a wider block with
            // tw % 4 == 0 at most two runs of a row each (two lanes collide); anything else scatters
            const double bank = tw % 16 == 0 ? 1.0 : (tw > 16 && tw % 4 == 0 ? 1.05 : 1.25);
            const double cost = tiles * rows * (1.0 + 0.05 * (double)((th + 2) * (tw + 2)) / (th * tw)) * bank;
            if (cost < best) { best = cost; TH = th; TW = tw; }
        }
    return best;
}

// 1 = the descriptor runs on conv3x3_patch_kernel (fills geo): a 3x3 stride-1 zero-padded convolution or its data gradient, one source
static bool patch_plan(const mcav_igemm_desc* d, PatchGeo& geo, bool f32 = false) {
    static const int enabled = MCAV_KNOB_INT("MCAV_PATCH", 1);
    if (!enabled || !d) return false;
    if (f32 ? (d->mma != 0 || !d->w) : (!d->w16 || d->mma < 1 || d->mma > 3)) return false;
    if (d->kh != 3 || d->kw != 3 || d->stride != 1) return false;
    geo.refl = 0;
    if (d->mode == MCAV_G_DIRECT && d->pad_mode == MCAV_PAD_REFLECT && d->sign == 1 && d->offset == -1) geo.refl = 1;
    else if (d->mode == MCAV_G_ADJ_REFLECT && d->sign == -1 && d->offset == 1) geo.refl = 2;
    else if (d->mode != MCAV_G_DIRECT || d->pad_mode != MCAV_PAD_ZERO) return false;
    else if (!((d->sign == 1 && d->offset == -1) || (d->sign == -1 && d->offset == 1))) return false;
    // reflection padding and its adjoint (the decoder's single-source 3x3 layers; round 4): the split form only
    static const int refl_on = MCAV_KNOB_INT("MCAV_PATCH_REFLECT", 1);
    if (geo.refl && (f32 || d->mma < 2 || !refl_on || d->Hs < 2 || d->Ws < 2)) return false;
    // ... and where it was measured ahead of the fp32 kernels (batch 12, profiles/r04_reflect_patch.txt): 64 outputs or more (half of a 64-wide
    // tile idle: 0.086 against 0.075 ms at 48x160, 64 -> 32), and for the adjoint the 24x80 maps and larger (12x40: level; 6x20, where every block
    // is a border block and runs up to three times the MFMAs: 0.097 against 0.076 ms); mma = 3 (the parity tests) takes it on every shape
    static const int refl_minpix = MCAV_KNOB_INT("MCAV_PATCH_REFLECT_MINPIX", 1000);
    if (geo.refl && d->mma != 3 && (d->n_count < 64 || (geo.refl == 2 && d->Hd * d->Wd < refl_minpix))) return false;
    if (d->C2 != 0 || d->up1 || d->pool || d->w_upmerge) return false;
    if (d->C1 % 32 != 0 || d->Kp != d->C1 || d->Hd != d->Hs || d->Wd != d->Ws || d->n_count < 32) return false;
    if (d->groups > 1 && d->B % d->groups != 0) return false;
    // 128-pixel blocks (half the filter traffic per FLOP) where they waste no more rows than 64-pixel ones and still fill the chip
    static const int force_tmb = MCAV_KNOB_INT("MCAV_PATCH_TMB", 0);
    int th1, tw1, th2, tw2;
    const double c1 = patch_block(d->Hd, d->Wd, 64, PatchCfg<1>::PIX, th1, tw1);
    const double c2 = patch_block(d->Hd, d->Wd, 128, PatchCfg<2>::PIX, th2, tw2);
    const long wg2 = (long)d->B * ((d->Hd + th2 - 1) / th2) * ((d->Wd + tw2 - 1) / tw2) * ((d->n_count + 63) / 64);
    geo.tmb = (c2 <= c1 * 1.02 && wg2 >= 1024) ? 2 : 1;
    if (force_tmb == 1 || force_tmb == 2) geo.tmb = force_tmb;
    geo.TH = geo.tmb == 2 ? th2 : th1;
    geo.TW = geo.tmb == 2 ? tw2 : tw1;
    geo.tiles_y = (d->Hd + geo.TH - 1) / geo.TH;
    geo.tiles_x = (d->Wd + geo.TW - 1) / geo.TW;
    // plain bf16 on few blocks (the 6x20 maps): the table-driven 32x64 tiles are faster (0.043 against 0.055 ms on 512 -> 512)
    if (d->mma == 1 && (long)d->B * geo.tiles_y * geo.tiles_x * ((d->n_count + 63) / 64) < 1024) return false;
    return true;
}

// Twelve-wavefront form (conv3x3_patch3_kernel): three 64-pixel blocks per workgroup, one workgroup per CU.  The split form only; the three
// blocks of a workgroup belong to one statistics group.  MCAV_PATCH3 (tune builds) / the low byte 0x33 of mcav_igemm_desc.tile (tests) select it.
static int patch3_plan(const mcav_igemm_desc* d, PatchGeo& geo) {
    static const int enabled = MCAV_KNOB_INT("MCAV_PATCH3", 0);       // OFF: ahead per launch on the 12x40 / 24x80 maps, level in the step (profiles/r04_patch3.txt)
    static const int min_wgs = MCAV_KNOB_INT("MCAV_PATCH3_MIN_WGS", 192);
    static const int min_k = MCAV_KNOB_INT("MCAV_PATCH3_MIN_K", 128);
    if (!d || d->mma < 2) return 0;
    const bool forced = (d->tile & 0xff) == 0x33;
    if (!enabled && !forced) return 0;
    PatchGeo g1;
    if (!patch_plan(d, g1)) return 0;
    int th, tw;
    patch_block(d->Hd, d->Wd, 64, P3_PIX, th, tw);
    geo = g1;
    geo.TH = th; geo.TW = tw; geo.tmb = 1;
    geo.tiles_y = (d->Hd + th - 1) / th;
    geo.tiles_x = (d->Wd + tw - 1) / tw;
    const long nblk = (long)d->B * geo.tiles_y * geo.tiles_x;
    if (d->groups > 1 && (nblk / d->groups) % P3_SUBS != 0) return 0;
    const long wgs = ((nblk + P3_SUBS - 1) / P3_SUBS) * ((d->n_count + 63) / 64);
    // Where it was measured ahead of the first kernel (batch 12, profiles/r04_patch3.txt): 128 or more channels per tap (four or more chunks: a
    // workgroup of two chunks lives for six stages and its set-up and epilogue run alone -- the 48x160 maps: 0.083 -> 0.096 ms) and enough
    // workgroups for one per CU (the 6x20 maps give 128: 0.110 -> 0.117); the 12x40 / 24x80 trunk maps: 0.093 -> 0.079 / 0.085 ms.
    if (!forced && (wgs < min_wgs || d->Kp < min_k)) return 0;
    return 1;
}

// Second form (conv3x3_patch2_kernel): SB = 4 or 2 sub-blocks of 64 pixels per workgroup, one workgroup per CU; 0 = not this form.  The
// choice is a count of rounds: a launch runs ceil(workgroups / CUs) rounds of SB units of work each, and the fewer sub-blocks win a tie only
// when they save a round (the filter stream per FLOP doubles with them).  A workgroup's sub-blocks must belong to one statistics group.
static int patch2_plan(const mcav_igemm_desc* d, PatchGeo& geo, int& bn) {
    // OFF by default: measured level with or behind the first kernel on every trunk shape (profiles/r04_patch2_forms.txt).
    // mcav_igemm_desc.tile bit 14 selects it (bit 15: its 128 x 32 two-workgroups-per-CU configuration): the parity tests; MCAV_PATCH2=1 in a
    // -DMCAV_TUNE_ENV build: experiments.
    static const int enabled = MCAV_KNOB_INT("MCAV_PATCH2", 0);
    static const int force_sb = MCAV_KNOB_INT("MCAV_PATCH2_SB", 0);
    static const int force_bn = MCAV_KNOB_INT("MCAV_PATCH2_BN", 0);
    if (!d || d->mma < 2) return 0;                                   // (the split form; plain bf16 keeps the first kernel)
    if (!enabled && !((d->tile >> 14) & 3)) return 0;
    PatchGeo g1;
    if (!patch_plan(d, g1) || g1.refl) return 0;                     // (reflection / its adjoint: the first kernel only)
    int th, tw;
    patch_block(d->Hd, d->Wd, 64, PatchCfg<1>::PIX, th, tw);
    geo.TH = th; geo.TW = tw; geo.tmb = 1; geo.refl = 0;
    geo.tiles_y = (d->Hd + th - 1) / th;
    geo.tiles_x = (d->Wd + tw - 1) / tw;
    const long per_img = (long)geo.tiles_y * geo.tiles_x, nblk = d->B * per_img, ntiles = (d->n_count + 63) / 64;
    const int groups = (d->stats && d->groups > 1) ? d->groups : 1;
    static const int cus = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        const int n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        (void)hipGetLastError();
        return n;
    }();
    auto fits = [&](int sb) { return d->B % groups == 0 && ((d->B / groups) * per_img) % sb == 0; };
    auto cost = [&](int sb) { const long wgs = ((nblk + sb - 1) / sb) * ntiles; return ((wgs + cus - 1) / cus) * sb; };
    int sb = 0;
    if (fits(4)) sb = 4;
    if (fits(2) && (sb == 0 || cost(2) < cost(4))) sb = 2;
    if ((force_sb == 2 || force_sb == 4) && fits(force_sb)) sb = force_sb;
    bn = 64;
    if ((force_bn == 32 || ((d->tile >> 15) & 1)) && fits(2)) { sb = 2; bn = 32; }
    return sb;
}

// ------------------------------------------------------------------------------------------------ weight gradient
// out[kflat, n] = sum_pix x(pix, tap)[c] * dy[pix, n] with both operands rounded to bf16 and the reduction over pixels on the MFMA's K axis.
// K-tile = KPB = 64 pixels.  Threads 0..127 stage the A operand (x through the per-(pixel, tap) offset table), threads 128..255 the B
// operand (dy rows): thread (col, pg) loads 4 channels (16 bytes) of the 8 consecutive pixels 8 pg .. 8 pg + 7 and writes, per channel, the
// 8 pixels as one 16-byte LDS store into the k-contiguous panel row of that channel.
constexpr int KPB = 64;

__device__ __forceinline__ int wsw(int row, int kb) { return ((kb ^ ((row >> 1) & 7)) << 3); }      // element offset of k-block kb in a panel row

// NS = 3: the fp32 contraction on split operands (see igemm_bf16_kernel): both operands are split on their way into LDS, one panel of
// three planes each, half the offset-table capacity (52 KB of LDS: three workgroups per CU).
template <int NS>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(WgradParams p) {
    constexpr int BM = 64, BN = 64;
    constexpr int NB = NS == 1 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) u16 Xs[NB * NS][BM][KPB];      // [kflat row][64 pixels], 128-byte rows, 16-byte slots XOR-swizzled
    __shared__ __attribute__((aligned(16))) u16 Ys[NB * NS][BN][KPB];
    __shared__ unsigned s_tab[NS == 1 ? WG_TABCAP : WG_TABCAP / 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per_split = p.mtiles * p.ntiles;
    // all row / column tiles of one pixel split read the same pixels (each tap a shifted view of x, every tile the same dy): consecutive
    // logical ids share an XCD, so the split's pixels are fetched into ONE L2 and re-read from there (the hardware deals workgroups
    // round-robin over the 8 XCDs; unlike the MFMA-bound fp32 kernel this one is bound by exactly that traffic)
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int split = lid / per_split, rem = lid - split * per_split;
    const int nt = rem % p.ntiles, mt = rem / p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const GatherSrc& g = p.g;

    const int pix_begin = split * p.pix_per_split;
    const int pix_end = min(p.Mpix, pix_begin + p.pix_per_split);
    const int T_total = pix_end > pix_begin ? (pix_end - pix_begin + KPB - 1) / KPB : 0;
    const int tap_lo = m0 / p.Kp;
    const int tap_hi = min(p.taps - 1, (m0 + BM - 1) / p.Kp);
    const int NT = tap_hi - tap_lo + 1;
    const bool two = g.C2 > 0;
    const int cht = p.tab_cht_log2;                              // log2 of the 64-pixel tiles per table chunk
    const int npc = min(T_total, 1 << min(cht, 20)) * KPB;
    const int half = NT * (two ? 2 : 1) * npc;

    auto build_chunk = [&](int c) {
        unsigned* tb = s_tab + (c & 1) * half;
        for (int pl = tid; pl < npc; pl += 256) {
            const int m = pix_begin + c * npc + pl;
            const bool live = m < pix_end;
            const int n = m / (p.Hd * p.Wd);
            const int r = m - n * (p.Hd * p.Wd);
            const int dy = r / p.Wd, dx = r - dy * p.Wd;
            for (int tl = 0; tl < NT; ++tl) {
                const int tap = tap_lo + tl;
                const int ky = tap / p.kw, kx = tap - ky * p.kw;
                int sy = dy * g.stride + ky + g.offset, sx = dx * g.stride + kx + g.offset;
                if (g.pad_mode == MCAV_PAD_REFLECT) {
                    sy = reflect_idx(sy, g.Hs);
                    sx = reflect_idx(sx, g.Ws);
                }
                const bool ok = live && (unsigned)sy < (unsigned)g.Hs && (unsigned)sx < (unsigned)g.Ws;
                const int pix = (n * g.Hs + sy) * g.Ws + sx;
                const int pix1 = g.up1 ? ((n * (g.Hs >> 1) + (sy >> 1)) * (g.Ws >> 1) + (sx >> 1)) : pix;
                tb[tl * npc + pl] = ok ? (unsigned)(pix1 * g.C1) * 4u : OOB;
                if (two) tb[(NT + tl) * npc + pl] = ok ? (unsigned)(pix * g.C2) * 4u : OOB;
            }
        }
    };
    const int nchunks = npc > 0 ? (T_total * KPB + npc - 1) / npc : 0;
    if (nchunks > 0) build_chunk(0);
    if (nchunks > 1) build_chunk(1);

    // ---- staging roles
    const bool isA = tid < 128;
    const int st = tid & 127;
    const int col = st & 15, pg = st >> 4;                       // 16-byte column (4 channels) and pixel group (8 pixels)
    // A: kflat = m0 + 4 col .. + 3
    const int kflat = m0 + col * 4;
    int tl_own = 0, ac = 0;
    bool a_ok = false;
    if (kflat < p.Ktot) {
        const int tap = kflat / p.Kp;
        ac = kflat - tap * p.Kp;
        tl_own = tap - tap_lo;
        a_ok = tap < p.taps;
    }
    const bool use2v = two && ac >= g.C1;                        // 16 channels (4 columns) never straddle the sources: C1 % 16 == 0 with a second source
    const int acc_ = use2v ? ac - g.C1 : ac;
    a_ok = a_ok && acc_ < (use2v ? g.C2 : g.C1);
    const unsigned chan = a_ok ? (unsigned)acc_ * 4u : OOB;
    const int trow = ((use2v ? NT : 0) + (a_ok ? tl_own : 0)) * npc + pg * 8;
    const unsigned bytes1 = (unsigned)((size_t)g.B * (g.up1 ? (g.Hs >> 1) * (g.Ws >> 1) : g.Hs * g.Ws) * g.C1 * 4);
    const unsigned bytes2 = (unsigned)((size_t)g.B * g.Hs * g.Ws * g.C2 * 4);
    const __amdgpu_buffer_rsrc_t rsx1 = make_rsrc(g.x1, bytes1);
    const __amdgpu_buffer_rsrc_t rsx2 = make_rsrc(two ? g.x2 : g.x1, two ? bytes2 : 0u);
    const __amdgpu_buffer_rsrc_t rsy = make_rsrc(p.dy, (unsigned)((size_t)pix_end * p.Cdy * 4));      // ends at this split's last pixel
    // B: channels n0 + 4 col .. + 3 of dy
    const int bc = n0 + col * 4;
    const bool b_ok = bc + 4 <= p.CoutLoad;
    const unsigned boff0 = b_ok ? (unsigned)(((pix_begin + pg * 8) * p.Cdy + p.dy_choff + bc) * 4) : OOB;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    const bool do_bias = p.want_bias && mt == 0;
    __syncthreads();

    int u = 0;                                                    // the issue pointer (tile index)
    // Both roles run the same instruction stream: 8 loads through a per-lane offset + a resource chosen per HALF workgroup (wave-uniform:
    // waves 0-1 stage A, waves 2-3 stage B), so there is no divergence and no waterfall around the descriptors.
    const bool waveA = __builtin_amdgcn_readfirstlane((int)isA) != 0;
    auto issue = [&](f32x4 (&rv)[8]) {
        if (waveA) {
            const int uc = u >> cht, ul = u - (uc << cht);
            const unsigned* tr = s_tab + (uc & 1) * half + trow + ul * KPB;
            unsigned to[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) to[j] = tr[j];
            // a wavefront's 16 columns may lie in x1 or in x2 (per 4-column group); the choice is per lane, so the two sources are issued as
            // two predicated load sets with scalar descriptors (lanes of the other source read with an out-of-range offset: zero)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned off = to[j] + chan;
                f32x4 v = buf_load4(rsx1, use2v ? OOB : off);
                if (two) v += buf_load4(rsx2, use2v ? off : OOB);
                rv[j] = v;
            }
        } else {
            const int sb = u * KPB * p.Cdy * 4;
#pragma unroll
            for (int j = 0; j < 8; ++j) rv[j] = buf_load4s(rsy, boff0 == OOB ? OOB : boff0 + (unsigned)(j * p.Cdy * 4), sb);
        }
        ++u;
    };
    auto store = [&](const f32x4 (&rv)[8], auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        if (!waveA && do_bias) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bsum += rv[j];
        }
        u16 (*dst)[BM][KPB] = waveA ? &Xs[buf * NS] : &Ys[buf * NS];
        static_assert(BM == BN, "one panel shape");
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int row = col * 4 + c;
            bf16x8 h;
            if constexpr (NS == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) h[j] = (__bf16)rv[j][c];
                *reinterpret_cast<bf16x8*>(&dst[0][row][wsw(row, pg)]) = h;
            } else {
                bf16x8 m, l;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = rv[j][c];
                    h[j] = (__bf16)v;
                    const float r1 = v - (float)h[j];                       // exact
                    m[j] = (__bf16)r1;
                    l[j] = (__bf16)(r1 - (float)m[j]);
                }
                *reinterpret_cast<bf16x8*>(&dst[0][row][wsw(row, pg)]) = h;
                *reinterpret_cast<bf16x8*>(&dst[1][row][wsw(row, pg)]) = m;
                *reinterpret_cast<bf16x8*>(&dst[2][row][wsw(row, pg)]) = l;
            }
        }
    };

    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    f32x16 acc, mid, low;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = mid[r] = low[r] = 0.f;
    const int fr = lane & 31, fh = lane >> 5;
    auto compute = [&](auto bufc) {
        constexpr int buf = decltype(bufc)::value * NS;
        if constexpr (NS == 1) {
            bf16x8 a[KPB / 16], b[KPB / 16];
#pragma unroll
            for (int ks = 0; ks < KPB / 16; ++ks) {
                a[ks] = *reinterpret_cast<const bf16x8*>(&Xs[buf][wm0 + fr][wsw(wm0 + fr, 2 * ks + fh)]);
                b[ks] = *reinterpret_cast<const bf16x8*>(&Ys[buf][wn0 + fr][wsw(wn0 + fr, 2 * ks + fh)]);
            }
#pragma unroll
            for (int ks = 0; ks < KPB / 16; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], b[ks], acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int ks = 0; ks < KPB / 16; ++ks) {
                bf16x8 a[3], b[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    a[q] = *reinterpret_cast<const bf16x8*>(&Xs[buf + q][wm0 + fr][wsw(wm0 + fr, 2 * ks + fh)]);
                    b[q] = *reinterpret_cast<const bf16x8*>(&Ys[buf + q][wn0 + fr][wsw(wn0 + fr, 2 * ks + fh)]);
                }
                low = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], low, 0, 0, 0);
                mid = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], mid, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
                low = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], low, 0, 0, 0);
                mid = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], mid, 0, 0, 0);
                low = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], low, 0, 0, 0);
            }
        }
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    auto maybe_build = [&](int tt) {
        if (((tt + 1) & ((1 << cht) - 1)) == 0 && ((tt + 1) >> cht) + 1 < nchunks) build_chunk(((tt + 1) >> cht) + 1);
    };

    f32x4 rv[8];
    if (T_total > 0) {
        issue(rv);
        store(rv, B0{});
        if (T_total > 1) issue(rv);
    }
    __syncthreads();
    if constexpr (NS == 1) {
        int t = 0;
        for (; t + 1 < T_total; t += 2) {
            maybe_build(t);
            store(rv, B1{});
            if (t + 2 < T_total) issue(rv);
            compute(B0{});
            __syncthreads();
            maybe_build(t + 1);
            if (t + 2 < T_total) {
                store(rv, B0{});
                if (t + 3 < T_total) issue(rv);
            }
            compute(B1{});
            __syncthreads();
        }
        if (t < T_total) {
            compute(B0{});
            __syncthreads();
        }
    } else {
        // one panel: multiply tile t, barrier, write tile t + 1 (its loads were in flight during the multiply), issue tile t + 2, barrier
        // (a second register stage was measured: 176 registers, two wavefronts per SIMD, slower)
        for (int t = 0; t < T_total; ++t) {
            maybe_build(t);
            compute(B0{});
            __syncthreads();
            if (t + 1 < T_total) {
                store(rv, B0{});
                if (t + 2 < T_total) issue(rv);
                __syncthreads();
            }
        }
        acc += mid + low;
    }

    float* slab = p.slab + (size_t)split * (p.Ktot + 1) * p.slabN;
    if (do_bias) {
        // column sums of dy over this split: each B-staging thread holds the sums of its 4 channels over its pixel group
        float (*red)[BN] = reinterpret_cast<float (*)[BN]>(&Xs[0][0][0]);      // 8 x 64 floats = 2 KB (the K loop has ended)
        if (!waveA) *reinterpret_cast<f32x4*>(&red[pg][col * 4]) = bsum;
        __syncthreads();
        if (tid < BN) {
            float tsum = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) tsum += red[k][tid];
            if (n0 + tid < p.slabN) slab[(size_t)p.Ktot * p.slabN + n0 + tid] = tsum;
        }
    }
    const int ccol = lane & 31;
    const int n = n0 + wn0 + ccol;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < p.Ktot && n < p.slabN) slab[(size_t)row * p.slabN + n] = acc[r];
    }
}

// ------------------------------------------------------------------------------------------------ host side
template <class T, int NS = 1>
inline void launch_igemm_bf16(const IgemmParams& p, const void* w16, bool refl, hipStream_t s) {
    const int grid = p.mtiles * p.ntiles;
    const size_t tab_bytes = sizeof(unsigned) * (size_t)p.taps * T::BM * (p.g.C2 > 0 ? 2 : 1);
    if (refl) timed_launch(igemm_bf16_kernel<T, 1, NS>, grid, dim3(256), tab_bytes, s, p, reinterpret_cast<const u16*>(w16));
    else timed_launch(igemm_bf16_kernel<T, 0, NS>, grid, dim3(256), tab_bytes, s, p, reinterpret_cast<const u16*>(w16));
}

// Which bf16 tile (0 = the launch is not one the bf16 kernels cover: the caller runs the fp32 path).  Sets *refl.
static int bf16_tile_for(const mcav_igemm_desc* d, bool* refl) {
    if (!d || !d->w16 || (d->mma < 1 || d->mma > 3)) return 0;
    if (d->mma == 2) {
        // The split form pays where the contraction dominates and the operands are converted once: the patch kernel.  On the table-driven
        // kernels (a conversion per tap) it was measured level with or behind the fp32 MFMA kernels (mma = 3 runs those too: tests).
        PatchGeo geo;
        if (!patch_plan(d, geo)) return 0;
    }
    if (d->pool || d->w_upmerge) return 0;                            // pooled / merged-tap forms stay on the fp32 kernels
    if (d->kh * d->kw > TAB_TAPS || d->Kp % 32 != 0 || d->C1 + d->C2 != d->Kp) return 0;
    if ((d->C1 & 3) || (d->C2 & 3) || (d->C2 > 0 && d->C1 % 32 != 0)) return 0;
    if (d->n_count < 32) return 0;                                    // narrow outputs: the halo / stencil kernels
    if (d->C2 == 0 && d->C1 <= 32 && d->n_count <= 32 && d->kh == 3 && d->stride == 1) return 0;      // conv_halo.hip's shapes (mcav_try_halo runs first)
    const bool direct = d->mode == MCAV_G_DIRECT || (d->mode == MCAV_G_ADJ_STRIDE2 && d->C2 == 0);
    const bool radj = d->mode == MCAV_G_ADJ_REFLECT && d->C2 == 0;
    if (!direct && !radj) return 0;
    *refl = radj;
    const bool k64 = d->mma == 1 && d->Kp % 64 == 0 && (d->C2 == 0 || d->C1 % 64 == 0);      // (the split form keeps 32-deep K-tiles: three planes per panel)
    const long M = (long)d->B * d->Hd * d->Wd;
    const long wg64 = ((M + 63) / 64) * ((d->n_count + 63) / 64);
    int shape = 10;                                                   // 64 x 64
    if (wg64 >= 4096 && !radj) shape = 8;                             // many rows: 128 x 64 (half the filter re-reads)
    else if (wg64 < 512) shape = 12;                                  // few rows (6x20 maps): 32 x 64
    if (d->mma >= 2) {
        static const int forced = MCAV_KNOB_INT("MCAV_SPLIT_TILE", 0);      // tuning knob: 10 / 8 / 12
        if ((forced == 10 || forced == 12 || (forced == 8 && !radj))) shape = forced;
    }
    return shape * 2 + (k64 ? 1 : 0);
}

}  // namespace mcav

using namespace mcav;

// fp32 MFMA patch kernel (mma = 0): which descriptors take it.  Measured level with the table-driven kernels on the 48x160 / 24x80 maps
// (0.129 against 0.133 ms, 0.115 against 0.112) and behind them on the 12x40 / 6x20 maps (ragged 16-wide blocks), so it is OFF by default:
// MCAV_PATCH_F32=1, or bit 13 of mcav_igemm_desc.tile (the parity test), selects it.
static bool patch_f32_plan(const mcav_igemm_desc* d, PatchGeo& geo) {
    static const int enabled = MCAV_KNOB_INT("MCAV_PATCH_F32", 0);
    if (!d || (!enabled && !((d->tile >> 13) & 1))) return false;
    return patch_plan(d, geo, true);
}

int mcav_patch_f32_mtiles(const mcav_igemm_desc* d) {
    PatchGeo geo;
    return patch_f32_plan(d, geo) ? d->B * geo.tiles_y * geo.tiles_x : 0;
}

bool mcav_try_patch_f32(const mcav_igemm_desc* d, hipStream_t s) {
    PatchGeo geo;
    if (!patch_f32_plan(d, geo)) return false;
    mcav_igemm_desc dd = *d;
    dd.tile = 2;
    dd.w_upmerge = nullptr;
    IgemmParams p;
    int tile;
    if (!fill_params(&dd, p, tile) || p.upm) return false;
    p.mtiles = d->B * geo.tiles_y * geo.tiles_x;
    p.ntiles = (p.n_count + 63) / 64;
    const int grid = p.mtiles * p.ntiles;
    if (geo.tmb == 2) timed_launch(conv3x3_patch_f32_kernel<2>, grid, dim3(256), patch_f32_lds_bytes<2>(), s, p, geo);
    else timed_launch(conv3x3_patch_f32_kernel<1>, grid, dim3(256), patch_f32_lds_bytes<1>(), s, p, geo);
    return true;
}

// returns 1 when the launch is not eligible (run the fp32 path), MCAV_OK / MCAV_E_* otherwise
int mcav_bf16_igemm(const mcav_igemm_desc* d, hipStream_t s) {
    bool refl = false;
    const int bt = bf16_tile_for(d, &refl);
    if (!bt) return 1;
    mcav_igemm_desc dd = *d;
    dd.tile = bt >> 1;                                                // the fp32 planner lays the rows out for these tile dimensions
    dd.w_upmerge = nullptr;
    IgemmParams p;
    int tile;
    // (past this point the caller may have put the bf16 copy into d->w as well: never fall through to the fp32 kernels)
 PatchGeo geo;
    int bn2 = 64;
    if (patch3_plan(d, geo)) {
        dd.tile = 10;
        if (!fill_params(&dd, p, tile) || p.upm) return MCAV_E_INVALID;
        if ((long)d->Np * p.Kstride * 2 * 3 >= 0x7fffffffL) return MCAV_E_INVALID;
        const long nblk = (long)d->B * geo.tiles_y * geo.tiles_x;
        p.mtiles = (int)((nblk + P3_SUBS - 1) / P3_SUBS);            // rows of the statistics slab = workgroups (three blocks of one group each)
        p.ntiles = (p.n_count + 63) / 64;
        static const bool allowed = [] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_patch3_kernel<3, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)patch3_lds_bytes<3>()) == hipSuccess &&
                   hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_patch3_kernel<3, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)patch3_lds_bytes<3>()) == hipSuccess;
        }();
        if (!allowed) return MCAV_E_LAUNCH;
        const int grid = p.mtiles * p.ntiles;
        const u16* w16 = reinterpret_cast<const u16*>(d->w16);
        if (geo.refl == 2) timed_launch(conv3x3_patch3_kernel<3, true>, grid, dim3(P3_THREADS), patch3_lds_bytes<3>(), s, p, w16, geo);
        else timed_launch(conv3x3_patch3_kernel<3, false>, grid, dim3(P3_THREADS), patch3_lds_bytes<3>(), s, p, w16, geo);
        return launch_status();
    }
    if (const int sb = patch2_plan(d, geo, bn2)) {
        dd.tile = 10;
        if (!fill_params(&dd, p, tile) || p.upm) return MCAV_E_INVALID;
        if ((long)d->Np * p.Kstride * 2 * 3 >= 0x7fffffffL) return MCAV_E_INVALID;
        const long nblk = (long)d->B * geo.tiles_y * geo.tiles_x;
        p.mtiles = (int)((nblk + sb - 1) / sb);                      // rows of the statistics slab = workgroup rows, image-major
        p.ntiles = (p.n_count + bn2 - 1) / bn2;
        static const bool allowed = [] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_patch2_kernel<3, 4, 64>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)patch2_lds_bytes<3, 4, 64>()) == hipSuccess &&
                   hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_patch2_kernel<3, 2, 64>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)patch2_lds_bytes<3, 2, 64>()) == hipSuccess &&
                   hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_patch2_kernel<3, 2, 32>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)patch2_lds_bytes<3, 2, 32>()) == hipSuccess;
        }();
        if (!allowed) return MCAV_E_LAUNCH;
        const int grid = p.mtiles * p.ntiles;
        const u16* w16 = reinterpret_cast<const u16*>(d->w16);
        if (bn2 == 32) timed_launch(conv3x3_patch2_kernel<3, 2, 32>, grid, dim3(256), patch2_lds_bytes<3, 2, 32>(), s, p, w16, geo);
        else if (sb == 4) timed_launch(conv3x3_patch2_kernel<3, 4, 64>, grid, dim3(256), patch2_lds_bytes<3, 4, 64>(), s, p, w16, geo);
        else timed_launch(conv3x3_patch2_kernel<3, 2, 64>, grid, dim3(256), patch2_lds_bytes<3, 2, 64>(), s, p, w16, geo);
        return launch_status();
    }
    if (patch_plan(d, geo)) {
        dd.tile = 10;
        if (!fill_params(&dd, p, tile) || p.upm) return MCAV_E_INVALID;
        if ((long)d->Np * p.Kstride * 2 * (d->mma >= 2 ? 3 : 1) >= 0x7fffffffL) return MCAV_E_INVALID;
        p.mtiles = d->B * geo.tiles_y * geo.tiles_x;                  // rows of the statistics slab = blocks, image-major (groups = runs of images)
        p.ntiles = (p.n_count + 63) / 64;
        // (more than 64 KB of dynamic LDS has to be allowed once per kernel)
        static const bool allowed = [] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_patch_kernel<3, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)patch_lds_bytes<3, 2>()) == hipSuccess &&
                   hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_patch_kernel<3, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)patch_lds_bytes<3, 2>()) == hipSuccess;
        }();
        const int grid = p.mtiles * p.ntiles;
        const u16* w16 = reinterpret_cast<const u16*>(d->w16);
        if (geo.refl == 2) {                                          // (patch_plan: the split form only)
            if (geo.tmb == 2) {
                if (!allowed) return MCAV_E_LAUNCH;
                timed_launch(conv3x3_patch_kernel<3, 2, true>, grid, dim3(256), patch_lds_bytes<3, 2>(), s, p, w16, geo);
            } else {
                timed_launch(conv3x3_patch_kernel<3, 1, true>, grid, dim3(256), patch_lds_bytes<3, 1>(), s, p, w16, geo);
            }
        } else if (d->mma >= 2) {
            if (geo.tmb == 2) {
                if (!allowed) return MCAV_E_LAUNCH;
                timed_launch(conv3x3_patch_kernel<3, 2>, grid, dim3(256), patch_lds_bytes<3, 2>(), s, p, w16, geo);
            } else {
                timed_launch(conv3x3_patch_kernel<3, 1>, grid, dim3(256), patch_lds_bytes<3, 1>(), s, p, w16, geo);
            }
        } else if (geo.tmb == 2) {
            timed_launch(conv3x3_patch_kernel<1, 2>, grid, dim3(256), patch_lds_bytes<1, 2>(), s, p, w16, geo);
        } else {
            timed_launch(conv3x3_patch_kernel<1, 1>, grid, dim3(256), patch_lds_bytes<1, 1>(), s, p, w16, geo);
        }
        return launch_status();
    }
    if (!fill_params(&dd, p, tile) || p.upm) return MCAV_E_INVALID;
    if ((long)d->Np * p.Kstride * 2 >= 0x7fffffffL) return MCAV_E_INVALID;
    const bool k64 = bt & 1;
    if (d->mma >= 2) {
        if ((long)d->Np * p.Kstride * 2 * 3 >= 0x7fffffffL) return MCAV_E_INVALID;
        switch (tile) {
            case 10: launch_igemm_bf16<BT64x64k32, 3>(p, d->w16, refl, s); break;
            case 8: launch_igemm_bf16<BT128x64k32, 3>(p, d->w16, refl, s); break;
            case 12: launch_igemm_bf16<BT32x64k32, 3>(p, d->w16, refl, s); break;
            default: return MCAV_E_INVALID;
        }
        return launch_status();
    }
    switch (tile) {
        case 10: if (k64) launch_igemm_bf16<BT64x64k64>(p, d->w16, refl, s); else launch_igemm_bf16<BT64x64k32>(p, d->w16, refl, s); break;
        case 8: if (k64) launch_igemm_bf16<BT128x64k64>(p, d->w16, refl, s); else launch_igemm_bf16<BT128x64k32>(p, d->w16, refl, s); break;
        case 12: if (k64) launch_igemm_bf16<BT32x64k64>(p, d->w16, refl, s); else launch_igemm_bf16<BT32x64k32>(p, d->w16, refl, s); break;
        default: return MCAV_E_INVALID;
    }
    return launch_status();
}

// The M-tile count (rows of the BatchNorm statistics slab) of the bf16 launch, whose tile shape is chosen independently of the fp32
// planner's: 0 when the descriptor does not run on the bf16 kernels.
int mcav_bf16_igemm_mtiles(const mcav_igemm_desc* d) {
    bool refl = false;
    const int bt = bf16_tile_for(d, &refl);
    if (!bt) return 0;
    PatchGeo geo;
    int bn2 = 64;
    if (patch3_plan(d, geo)) return (int)(((long)d->B * geo.tiles_y * geo.tiles_x + P3_SUBS - 1) / P3_SUBS);
    if (const int sb = patch2_plan(d, geo, bn2)) return (int)(((long)d->B * geo.tiles_y * geo.tiles_x + sb - 1) / sb);
    if (patch_plan(d, geo)) return d->B * geo.tiles_y * geo.tiles_x;
    mcav_igemm_desc dd = *d;
    dd.tile = bt >> 1;
    dd.w_upmerge = nullptr;
    IgemmParams p;
    int tile;
    if (!fill_params(&dd, p, tile) || p.upm) return 0;
    return p.mtiles;
}

MCAV_EXPORT int mcav_igemm_uses_bf16(const mcav_igemm_desc* d) {
    bool refl = false;
    return bf16_tile_for(d, &refl) != 0;
}

// ------------------------------------------------------------------------------------------------ weight gradient on the patch (round 4)
// dW[tap][ci][co] = sum over pixels p of x[p + tap - 1][ci] dy[p][co] for the 3x3 stride-1 layers, as an fp32 contraction on split operands
// (section 4c's six plane products; ONE fp32 accumulator per output tile: the rounding pattern of the fp32-MFMA kernel it replaces, measured
// level with it by tools/mfma_split_test.hip).  The reduction runs over PIXELS, so both MFMA operands want 8 consecutive pixels of one channel
// per lane while NHWC memory -- and the patch image of the forward kernel -- keeps a pixel's channels together: gfx950's transposing LDS read
// (ds_read_b64_tr_b16: a 16-lane group reads 4 rows x 16 columns of 16-bit elements and gets them column-major; each lane supplies the address
// of one row piece) turns [pixel][32 channels] rows into exactly those operands, whatever the pixels are -- so the nine taps are nine row
// offsets into ONE staged patch, as in conv3x3_patch_kernel, and a source element is fetched and split once per block instead of once per tap.
// A workgroup = (pixel split, 64 input x 64 output channels), twelve wavefronts (three per SIMD, one workgroup per CU): wavefront (ky, ci half,
// co half) owns the three 32 x 32 tiles of filter row ky.  Per block of <= 64 pixels (the planner's TH x TW shapes): the (TH + 2) x (TW + 2)
// source patch and the dy block, both as three bf16 planes, in a two-stage LDS ring (133 KB): block b + 1 is converted and stored while block
// b is multiplied (72 MFMAs per wavefront, one barrier per block), block b + 2 is in flight in registers.  The partial filter of a workgroup
// goes to the slab the fp32 kernels use ([split][Ktot + 1][slabN]; row Ktot = column sums of dy, summed by the staging threads), reduced by the same
// batched presum / reduce launches in the same fixed order.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
constexpr int WGP_PIX = 110;                       // patch pixels (PatchCfg<1>::PIX)
constexpr int WGP_THREADS = 768;
template <int NS> struct WgpCfg {                  // NS planes per operand: 3 = the split fp32 form, 1 = plain bf16 (mcav_wgrad_desc.mma = 1)
    static constexpr int XS = NS * WGP_PIX * 32;   // u16 elements: one 32-channel half of the patch
    static constexpr int DS = NS * 64 * 32;        // ... of the dy block
    static constexpr int STAGE = 2 * XS + 2 * DS;  // one ring stage: 33 408 elements = 66 816 bytes in the split form
};
template <int NS> constexpr size_t wgrad_patch_lds_bytes() { return sizeof(u16) * 2 * (size_t)WgpCfg<NS>::STAGE; }

// timing experiments only, WRONG results (make variant FLAGS=-DMCAV_WGP_DIAG=n): 1 no MFMAs, 2 no staging in the loop, 4 no slab stores, 8 no
// barrier in the loop, 16 no conversion arithmetic in the staging (raw halves stored)
#ifndef MCAV_WGP_DIAG
#define MCAV_WGP_DIAG 0
#endif

__device__ __forceinline__ bf16x8 tr_frag(const u16* lo4, const u16* hi4) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lo4);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)hi4);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int NS, bool BIAS>                       // BIAS: the launch also wants the column sums of dy
__global__ __launch_bounds__(WGP_THREADS, 1) void wgrad3x3_patch_kernel(WgradParams p) {
    constexpr int WGP_XS = WgpCfg<NS>::XS, WGP_DS = WgpCfg<NS>::DS, WGP_STAGE = WgpCfg<NS>::STAGE;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    u16* const lds = reinterpret_cast<u16*>(s_raw);
    constexpr int diag = MCAV_WGP_DIAG;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ky = wave >> 2, ci_sub = (wave >> 1) & 1, half = wave & 1;
    // 32 outputs or fewer (pnarrow): one 32-wide output tile, and the wavefront pairs share a block's four 16-pixel steps instead -- each half a
    // slab split of its own (the slab then holds 2 x the pixel splits)
    const bool narrow = p.pnarrow != 0;
    const int co_sub = narrow ? 0 : half, ks0 = narrow ? 2 * half : 0, nks = narrow ? 2 : 4;
    const int ct_ci = p.Kp >> 6, ctiles = ct_ci * p.pct_co;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int ct = lid % ctiles, split = lid / ctiles;
    const int ci0 = (ct / p.pct_co) * 64, co0 = (ct % p.pct_co) * 64;
    const int b_begin = split * p.pbps, b_end = min(p.pnblocks, b_begin + p.pbps);
    const GatherSrc& g = p.g;
    const int TH = p.pTH, TW = p.pTW, PW = TW + 2, PH = TH + 2, npix = TH * TW;
    const int per_img = p.ptiles_y * p.ptiles_x;
    const float rcp_pw = 1.0f / (float)PW, rcp_tw = 1.0f / (float)TW;

    // staging: item = (pixel, 4 channels); the patch has 110 x 16 of them, the dy block 64 x 16
    int xpy[3], xpx[3];                             // patch pixel -> (row, column) of the patch; -1: no such item / pixel
    unsigned xdst[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int item = tid + WGP_THREADS * j, pp = item >> 4, c4 = item & 15;
        const int ppy = small_div(pp, rcp_pw);
        const bool ok = pp < PH * PW && pp < WGP_PIX;
        xpy[j] = ok ? ppy : -1;
        xpx[j] = pp - ppy * PW;
        xdst[j] = (unsigned)((c4 >> 3) * WGP_XS + pp * 32 + (c4 & 7) * 4);
    }
    int dpy[2], dpx[2];
    unsigned ddst[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int item = tid + WGP_THREADS * j, k = item >> 4, c4 = item & 15;
        const int py = small_div(k, rcp_tw);
        const bool ok = item < 64 * 16 && k < npix && co0 + c4 * 4 < p.CoutLoad;
        dpy[j] = ok ? py : -1;
        dpx[j] = k - py * TW;
        ddst[j] = (unsigned)(2 * WGP_XS + (c4 >> 3) * WGP_DS + k * 32 + (c4 & 7) * 4);
    }
    const unsigned xch = (unsigned)((ci0 + (tid & 15) * 4) * 4), dch = (unsigned)((p.dy_choff + co0 + (tid & 15) * 4) * 4);
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(g.x1, (unsigned)((size_t)g.B * g.Hs * g.Ws * g.C1 * 4));
    const __amdgpu_buffer_rsrc_t rsd = make_rsrc(p.dy, (unsigned)((size_t)g.B * p.Hd * p.Wd * p.Cdy * 4));

    f32x4 rx[3], rd[2], bsum = {0.f, 0.f, 0.f, 0.f};
    auto load_x = [&](int blk) {
        const int img = blk / per_img, tr = blk - img * per_img;
        const int tyi = tr / p.ptiles_x;
        const int ty0 = tyi * TH, tx0 = (tr - tyi * p.ptiles_x) * TW;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            int y = ty0 - 1 + xpy[j], x = tx0 - 1 + xpx[j];
            if (p.prefl) {                          // reflection padding: the ring of the map is its row / column 1, H - 2 / W - 2
                y = y < 0 ? -y : (y >= g.Hs ? 2 * g.Hs - 2 - y : y);
                x = x < 0 ? -x : (x >= g.Ws ? 2 * g.Ws - 2 - x : x);
            }
            const bool ok = xpy[j] >= 0 && (unsigned)y < (unsigned)g.Hs && (unsigned)x < (unsigned)g.Ws;
            rx[j] = buf_load4s(rsx, ok ? (unsigned)(((img * g.Hs + y) * g.Ws + x) * g.C1 * 4) + xch : OOB, 0);
        }
    };
    auto load_d = [&](int blk) {
        const int img = blk / per_img, tr = blk - img * per_img;
        const int tyi = tr / p.ptiles_x;
        const int ty0 = tyi * TH, tx0 = (tr - tyi * p.ptiles_x) * TW;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int y = ty0 + dpy[j], x = tx0 + dpx[j];
            const bool ok = dpy[j] >= 0 && y < p.Hd && x < p.Wd;
            rd[j] = buf_load4s(rsd, ok ? (unsigned)(((img * p.Hd + y) * p.Wd + x) * p.Cdy * 4) + dch : OOB, 0);
        }
    };
    auto load_block = [&](int blk) { load_x(blk); load_d(blk); };
    auto split_store = [&](u16* dst, int plane_stride, const f32x4& v) {
        if (diag & 16) {
            const u32x2 raw = {__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[2])};
            *reinterpret_cast<u32x2*>(dst) = raw;
            *reinterpret_cast<u32x2*>(dst + plane_stride) = raw;
            *reinterpret_cast<u32x2*>(dst + 2 * plane_stride) = raw;
            return;
        }
        if constexpr (NS == 1) {
            *reinterpret_cast<u32x2*>(dst) = pack_bf16x4(v);
        } else {
            const bf16x4 h = __builtin_convertvector(v, bf16x4);
            const f32x4 r1 = v - __builtin_convertvector(h, f32x4);           // exact
            const bf16x4 m = __builtin_convertvector(r1, bf16x4);
            const f32x4 r2 = r1 - __builtin_convertvector(m, f32x4);          // exact
            *reinterpret_cast<u32x2*>(dst) = __builtin_bit_cast(u32x2, h);
            *reinterpret_cast<u32x2*>(dst + plane_stride) = __builtin_bit_cast(u32x2, m);
            *reinterpret_cast<u32x2*>(dst + 2 * plane_stride) = pack_bf16x4(r2);
        }
    };
    // the staged block goes to LDS in five pieces (three patch items, two dy items per thread)
    auto store_piece = [&](int stage, auto piece) {
        constexpr int i = decltype(piece)::value;
        u16* const base = lds + stage * WGP_STAGE;
        if constexpr (i < 3) {
            if (tid + WGP_THREADS * i < WGP_PIX * 16) split_store(base + xdst[i], WGP_PIX * 32, rx[i]);
        } else {
            if (tid + WGP_THREADS * (i - 3) < 64 * 16) split_store(base + ddst[i - 3], 64 * 32, rd[i - 3]);
            if constexpr (BIAS) bsum += rd[i - 3];    // column sums of dy (the bias gradient): this thread's pixels of its four channels (pixels past the map are zero)
        }
    };
    auto store_piece_n = [&](int stage, int i) {       // (i is a constant after unrolling)
        switch (i) {
            case 0: store_piece(stage, std::integral_constant<int, 0>()); break;
            case 1: store_piece(stage, std::integral_constant<int, 1>()); break;
            case 2: store_piece(stage, std::integral_constant<int, 2>()); break;
            case 3: store_piece(stage, std::integral_constant<int, 3>()); break;
            default: store_piece(stage, std::integral_constant<int, 4>()); break;
        }
    };
    auto store_block = [&](int stage) {
        store_piece(stage, std::integral_constant<int, 0>());
        store_piece(stage, std::integral_constant<int, 1>());
        store_piece(stage, std::integral_constant<int, 2>());
        store_piece(stage, std::integral_constant<int, 3>());
        store_piece(stage, std::integral_constant<int, 4>());
    };

    // operand addresses of this lane (elements, stage 0): lane 4q + c of a 16-lane group supplies row q, columns 4c .. 4c + 3 of the group's
    // 4 x 16 block and receives column (lane & 15), rows 0 .. 3: for the 32x32x16 operands group g covers channels 16 (g & 1) .. + 15 and
    // pixels 8 (g >> 1) + 4 h .. + 3 of the 16-pixel step (h = the first / second read of a fragment)
    const int colel = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    unsigned xa[4][2];                              // step i of this wavefront = the block's 16-pixel step ks0 + i
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = 16 * (ks0 + ks) + 8 * (lane >> 5) + 4 * h + ((lane & 15) >> 2);
            const int kk = k < npix ? k : 0;        // (pixels past the block: dy is zero there, any patch row will do)
            const int py = small_div(kk, rcp_tw), px = kk - py * TW;
            xa[ks][h] = (unsigned)(ci_sub * WGP_XS + ((py + ky) * PW + px) * 32 + colel);
        }
    const unsigned da = (unsigned)(2 * WGP_XS + co_sub * WGP_DS + (16 * ks0 + 8 * (lane >> 5) + ((lane & 15) >> 2)) * 32 + colel);

    f32x16 acc[3];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; acc[2][r] = 0.f; }

    // A block's steps (four 16-pixel steps x three taps of the wavefront's filter row; the narrow form: two x three), software-pipelined by hand:
    // the fragments of step s + 1 are read while the six MFMAs of step s run, and the NEXT block's conversion and LDS stores (store_piece: the
    // other ring stage) are spread over the steps instead of following them -- all twelve wavefronts run in phase between two barriers, so
    // staging done after the multiply would find the MFMA pipe idle (measured: 5.7 us per block against 4.5 without any staging).
    const int nsteps = 3 * nks;
    auto compute = [&](int stage, bool more, int next2) {      // next2: block b + 2, or -1
        const u16* const base = lds + stage * WGP_STAGE;
        bf16x8 a[2][NS], b[2][NS];
        auto loadA = [&](int st, bf16x8 (&f)[NS]) {
            const int ks = st / 3, kx = st - 3 * ks;
#pragma unroll
            for (int q = 0; q < NS; ++q) f[q] = tr_frag(base + xa[ks][0] + q * (WGP_PIX * 32) + kx * 32, base + xa[ks][1] + q * (WGP_PIX * 32) + kx * 32);
        };
        auto loadB = [&](int ks, bf16x8 (&f)[NS]) {
#pragma unroll
            for (int q = 0; q < NS; ++q) f[q] = tr_frag(base + da + q * (64 * 32) + (16 * ks) * 32, base + da + q * (64 * 32) + (16 * ks + 4) * 32);
        };
        loadB(0, b[0]);
        loadA(0, a[0]);
#pragma unroll
        for (int st = 0; st < 12; ++st) {
            if (st >= nsteps) break;
            const int ks = st / 3, kx = st - 3 * ks;
            if (st + 1 < nsteps) {
                if ((st + 1) % 3 == 0) loadB((st + 1) / 3, b[((st + 1) / 3) & 1]);
                loadA(st + 1, a[(st + 1) & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8(&fa)[NS] = a[st & 1];
            const bf16x8(&fb)[NS] = b[ks & 1];
            if (!(diag & 1)) {
                if constexpr (NS == 1) {
                    acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], acc[kx], 0, 0, 0);
                } else {
                    acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fb[0], acc[kx], 0, 0, 0);
                    acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[2], acc[kx], 0, 0, 0);
                    acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[1], acc[kx], 0, 0, 0);
                    acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[0], acc[kx], 0, 0, 0);
                    acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], acc[kx], 0, 0, 0);
                    acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], acc[kx], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (more && !(diag & 2)) {              // a piece of block b + 1's staging behind this step's MFMAs: steps 1, 3, .. 9 (narrow: 0 .. 4);
                const int pc = narrow ? (st < 5 ? st : -1) : (((st & 1) && st < 10) ? (st - 1) / 2 : -1);
                if (pc >= 0) store_piece_n(stage ^ 1, pc);
                // ... and block b + 2's loads as soon as their registers are free (half a block and more before their use: a load issued at the
                // end of the block would be waited for at its first piece)
                if (pc == 2 && next2 >= 0) load_x(next2);
                if (pc == 4 && next2 >= 0) load_d(next2);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (b_begin < b_end) {
        load_block(b_begin);
        store_block(0);
        if (b_begin + 1 < b_end) load_block(b_begin + 1);
        __syncthreads();
        int stage = 0;
        for (int b = b_begin; b < b_end; ++b) {
            compute(stage, b + 1 < b_end, b + 2 < b_end ? b + 2 : -1);      // (stores block b + 1 from the registers into the other stage, last read before the previous barrier)
            if (!(diag & 8)) __syncthreads();
            stage ^= 1;
        }
    }
    if (diag & 4) return;
    // the workgroup's partial filter -> slab [split][Ktot + 1][slabN]: accumulator register r of lane l = row 8 (r >> 2) + 4 (l >> 5) + (r & 3), column l & 31
    float* const slab = p.slab + (size_t)(narrow ? 2 * split + half : split) * (p.Ktot + 1) * p.slabN;
    const int co = co0 + co_sub * 32 + (lane & 31);
    if (co < p.slabN) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + ci_sub * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                slab[(size_t)((ky * 3 + kx) * p.Kp + ci) * p.slabN + co] = acc[kx][r];
            }
    }
    if constexpr (BIAS) {
        // slab row Ktot = the split's column sums of dy: the staging threads' sums (48 threads per group of four channels) through LDS -- the ring
        // is free after the loop's last barrier.  One input-channel tile writes the row; the narrow form's second slab split gets zeros.
        if (ct / p.pct_co != 0) return;
        f32x4* const sb = reinterpret_cast<f32x4*>(s_raw);
        sb[tid] = bsum;
        __syncthreads();
        if (tid < 64 && co0 + tid < p.slabN) {
            float sum = 0.f;
            for (int r = 0; r < WGP_THREADS / 16; ++r) sum += sb[r * 16 + (tid >> 2)][tid & 3];
            float* const s0 = p.slab + (size_t)(narrow ? 2 * split : split) * (p.Ktot + 1) * p.slabN + (size_t)p.Ktot * p.slabN;
            s0[co0 + tid] = sum;
            if (narrow) s0[(size_t)(p.Ktot + 1) * p.slabN + co0 + tid] = 0.f;
        }
    }
}

// Which weight gradients take wgrad3x3_patch_kernel: 3x3 stride 1, zero or reflection padding, one source of 64-channel multiples, >= 64 outputs.
static bool wgrad_patch_plan(const mcav_wgrad_desc* d, WgradPlan& pl) {
    static const int enabled = MCAV_KNOB_INT("MCAV_WGRAD_PATCH", 1);
    static const int target = MCAV_KNOB_INT("MCAV_WGRAD_PATCH_WGS", 256);
    if (!enabled || !d || d->mma < 1 || d->mma > 3 || d->upm || d->up1 || d->C2 != 0) return false;
    if (d->mode != MCAV_G_DIRECT || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->sign != 1 || d->offset != -1) return false;
    if (d->pad_mode != MCAV_PAD_ZERO && d->pad_mode != MCAV_PAD_REFLECT) return false;
    if (d->pad_mode == MCAV_PAD_REFLECT && (d->Hs < 2 || d->Ws < 2)) return false;
    if (d->C1 != d->Kp || d->Cin != d->Kp || d->Kp % 64 != 0 || d->Cout < 32 || d->Hd != d->Hs || d->Wd != d->Ws) return false;
    if (d->Cout < 64 && d->Cout != 32) return false;                  // (32 outputs: the narrow form; 33 .. 63 would leave half of a 64-wide tile idle)
    if ((d->Cdy & 3) || (d->dy_choff & 3)) return false;
    mcav_wgrad_desc dd = *d;
    dd.tile = 2;
    if (!plan_wgrad(&dd, pl) || pl.use_halo || pl.use_stem) return false;
    WgradParams& p = pl.p;
    if ((p.CoutLoad & 3) != 0) return false;
    int th, tw;
    patch_block(d->Hd, d->Wd, 64, WGP_PIX, th, tw);
    p.patch = 1; p.split_planes = d->mma >= 2;
    p.pTH = th; p.pTW = tw;
    p.ptiles_y = (d->Hd + th - 1) / th; p.ptiles_x = (d->Wd + tw - 1) / tw;
    p.prefl = d->pad_mode == MCAV_PAD_REFLECT;
    p.pnblocks = d->B * p.ptiles_y * p.ptiles_x;
    p.pct_co = (d->Cout + 63) / 64;
    p.pnarrow = d->Cout <= 32;
    const int ctiles = (d->Kp / 64) * p.pct_co;
    int splits = target / ctiles;                                     // one workgroup per CU
    if (splits < 1) splits = 1;
    if (splits > p.pnblocks) splits = p.pnblocks;
    p.pbps = (p.pnblocks + splits - 1) / splits;
    p.splits = (p.pnblocks + p.pbps - 1) / p.pbps * (p.pnarrow ? 2 : 1);      // slab splits
    pl.use_tab = false;
    pl.slab_bytes = align_up(sizeof(float) * (size_t)p.splits * (p.Ktot + 1) * p.slabN, 256);
    pl.groups = p.splits > 8 ? 8 : 0;
    pl.per_group = pl.groups ? (p.splits + pl.groups - 1) / pl.groups : 0;
    if (pl.groups) pl.groups = (p.splits + pl.per_group - 1) / pl.per_group;
    pl.pre_bytes = align_up(sizeof(float) * (size_t)pl.groups * (p.Ktot + 1) * p.slabN, 256);
    return true;
}

// Plans the bf16 weight-gradient launch into pl (splits over 64-pixel K-tiles, table chunking); false = not eligible.
bool mcav_bf16_wgrad_plan(const mcav_wgrad_desc* d, WgradPlan& pl) {
    if (!d || (d->mma < 1 || d->mma > 3) || d->upm) return false;
    if (wgrad_patch_plan(d, pl)) return true;
    if (d->mma == 2) return false;                                    // mma = 2: the split form where it is ahead (the patch kernel), else fp32 MFMA
    if (d->mode != MCAV_G_DIRECT || d->Kp % 16 != 0 || d->C1 + d->C2 != d->Kp || d->Cin != d->Kp) return false;
    if ((d->C1 & 15) || (d->C2 & 15) || d->Cout < 32 || (d->Cdy & 3) || (d->dy_choff & 3)) return false;
    mcav_wgrad_desc dd = *d;
    dd.tile = 2;                                                      // 64 x 64 slab tiles
    if (!plan_wgrad(&dd, pl) || pl.use_halo) return false;
    WgradParams& p = pl.p;
    if ((p.CoutLoad & 3) != 0) return false;
    const int out_tiles = p.mtiles * p.ntiles;
    // whole generations of the 1024 resident workgroups (four per CU), as plan_wgrad does: the fewest splits whose last generation is 95 % full
    int max_splits = (p.Mpix + 4 * KPB - 1) / (4 * KPB);
    if (max_splits > 512) max_splits = 512;
    if (max_splits < 1) max_splits = 1;
    const int slots = 1024;
    auto pps_of = [&](int sp) { return ((p.Mpix + sp - 1) / sp + KPB - 1) / KPB * KPB; };
    int lo = slots / out_tiles, hi = 3 * slots / out_tiles + 1;
    if (lo < 1) lo = 1;
    if (lo > max_splits) lo = max_splits;
    if (hi > max_splits) hi = max_splits;
    int splits = lo;
    double best = -1.0;
    for (int sp = lo; sp <= hi; ++sp) {
        const int w = out_tiles * ((p.Mpix + pps_of(sp) - 1) / pps_of(sp)), gens = (w + slots - 1) / slots;
        const double fill = (double)w / ((double)gens * slots);
        if (fill > best + 1e-9) { best = fill; splits = sp; }
        if (fill >= 0.95) { splits = sp; break; }
    }
    p.pix_per_split = pps_of(splits);
    p.splits = (p.Mpix + p.pix_per_split - 1) / p.pix_per_split;
    int ntmax = 1;
    for (int mt = 0; mt < p.mtiles; ++mt) {
        const int lo = mt * 64 / d->Kp, hi = (mt * 64 + 63) / d->Kp < p.taps - 1 ? (mt * 64 + 63) / d->Kp : p.taps - 1;
        if (hi - lo + 1 > ntmax) ntmax = hi - lo + 1;
    }
    const int epp = ntmax * (d->C2 > 0 ? 2 : 1);
    p.tab_cht_log2 = 20;
    const int tabcap = d->mma >= 2 ? WG_TABCAP / 2 : WG_TABCAP;       // (the split form gives half of the table's LDS to its planes)
    if ((long)p.pix_per_split * epp > tabcap) {
        if (2 * 2 * KPB * epp > tabcap) return false;                 // not even two 2-tile chunks fit
        int lg = 1;
        while ((2 << lg) * KPB * epp <= tabcap / 2) ++lg;
        p.tab_cht_log2 = lg;
    }
    p.split_planes = d->mma >= 2;
    pl.use_tab = true;
    pl.slab_bytes = align_up(sizeof(float) * (size_t)p.splits * (p.Ktot + 1) * p.slabN, 256);
    pl.groups = p.splits > 8 ? 8 : 0;
    pl.per_group = pl.groups ? (p.splits + pl.groups - 1) / pl.groups : 0;
    if (pl.groups) pl.groups = (p.splits + pl.per_group - 1) / pl.per_group;
    pl.pre_bytes = align_up(sizeof(float) * (size_t)pl.groups * (p.Ktot + 1) * p.slabN, 256);
    return true;
}

namespace mcav {
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = (__bf16)src[i];
}
}  // namespace mcav

namespace mcav {
__global__ __launch_bounds__(256) void f32_to_bf16_planes_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = src[i];
        const __bf16 h = (__bf16)v;
        const float r1 = v - (float)h;
        const __bf16 m = (__bf16)r1;
        dst[i] = h;
        dst[n + i] = m;
        dst[2 * n + i] = (__bf16)(r1 - (float)m);
    }
}
}  // namespace mcav

MCAV_EXPORT int mcav_f32_to_bf16_planes(const float* src, void* dst_bf16, size_t n, void* stream) {
    if (!src || !dst_bf16) return MCAV_E_INVALID;
    if (n == 0) return MCAV_OK;
    const size_t b = (n + 255) / 256;
    f32_to_bf16_planes_kernel<<<(unsigned)(b < 4096 ? b : 4096), 256, 0, as_stream(stream)>>>(src, reinterpret_cast<__bf16*>(dst_bf16), n);
    return launch_status();
}

MCAV_EXPORT int mcav_f32_to_bf16(const float* src, void* dst_bf16, size_t n, void* stream) {
    if (!src || !dst_bf16) return MCAV_E_INVALID;
    if (n == 0) return MCAV_OK;
    const size_t b = (n + 255) / 256;
    f32_to_bf16_kernel<<<(unsigned)(b < 4096 ? b : 4096), 256, 0, as_stream(stream)>>>(src, reinterpret_cast<__bf16*>(dst_bf16), n);
    return launch_status();
}

void mcav_bf16_wgrad_launch(const WgradParams& p, hipStream_t s) {
    if (p.patch) {
        static const bool allowed = [] {
            bool ok = true;
            ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad3x3_patch_kernel<3, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wgrad_patch_lds_bytes<3>()) == hipSuccess;
            ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad3x3_patch_kernel<3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wgrad_patch_lds_bytes<3>()) == hipSuccess;
            ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad3x3_patch_kernel<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wgrad_patch_lds_bytes<1>()) == hipSuccess;
            ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad3x3_patch_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wgrad_patch_lds_bytes<1>()) == hipSuccess;
            return ok;
        }();
        (void)allowed;                                                // (refused: the launch itself fails and launch_status() reports it)
        const int grid = (p.pnarrow ? p.splits / 2 : p.splits) * (p.Kp / 64) * p.pct_co;
        if (p.split_planes) {
            if (p.want_bias) timed_launch(wgrad3x3_patch_kernel<3, true>, grid, dim3(WGP_THREADS), wgrad_patch_lds_bytes<3>(), s, p);
            else timed_launch(wgrad3x3_patch_kernel<3, false>, grid, dim3(WGP_THREADS), wgrad_patch_lds_bytes<3>(), s, p);
        } else {
            if (p.want_bias) timed_launch(wgrad3x3_patch_kernel<1, true>, grid, dim3(WGP_THREADS), wgrad_patch_lds_bytes<1>(), s, p);
            else timed_launch(wgrad3x3_patch_kernel<1, false>, grid, dim3(WGP_THREADS), wgrad_patch_lds_bytes<1>(), s, p);
        }
        return;
    }
    if (p.split_planes) timed_launch(wgrad_bf16_kernel<3>, p.splits * p.mtiles * p.ntiles, dim3(256), 0, s, p);
    else timed_launch(wgrad_bf16_kernel<1>, p.splits * p.mtiles * p.ntiles, dim3(256), 0, s, p);
}
