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
template <class T, int TK>
__global__ __launch_bounds__(256) void igemm_bf16_kernel(IgemmParams p, const u16* __restrict__ w16) {
    constexpr bool REFL = TK == 1;
    constexpr int BM = T::BM, BN = T::BN, CKT = T::KD;
    __shared__ __attribute__((aligned(16))) u16 As[2][BM][T::LDH];
    __shared__ __attribute__((aligned(16))) u16 Bs[2][BN][T::LDH];
    __shared__ int s_out[BM];
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
        s_out[r] = o;
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
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(w16, (unsigned)((size_t)(p.n_begin + p.n_count) * p.Kstride * 2));

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
    f32x4 ex0[REFL ? T::AROWS : 1], ex1[REFL ? T::AROWS : 1], ex2[REFL ? T::AROWS : 1];
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
    auto issue = [&](f32x4 (&ra)[T::AROWS], f32x4 (&rb)[T::BVECS]) {
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
                    const int ex = (dx == 1 && kx == 0) ? 0 : ((dx == g.Ws - 2 && kx == 2) ? g.Ws - 1 : -1);
                    const bool syok = (unsigned)sy < (unsigned)g.Hs, sxok = (unsigned)sx < (unsigned)g.Ws;
                    const int rowb = n * g.Hs, cb = c4 * 4;
                    const unsigned a0 = (unsigned)((((rowb + ey) * g.Ws + sx) * g.C1 + cb) * 4);
                    const unsigned a1 = (unsigned)((((rowb + sy) * g.Ws + ex) * g.C1 + cb) * 4);
                    const unsigned a2 = (unsigned)((((rowb + ey) * g.Ws + ex) * g.C1 + cb) * 4);
                    ex0[j] = buf_load4s(rs1, (n >= 0 && ey >= 0 && sxok) ? a0 : OOB, cbase * 4);
                    ex1[j] = buf_load4s(rs1, (n >= 0 && ex >= 0 && syok) ? a1 : OOB, cbase * 4);
                    ex2[j] = buf_load4s(rs1, (n >= 0 && ey >= 0 && ex >= 0) ? a2 : OOB, cbase * 4);
                }
            }
        }
        const int kb = (tap * p.Kp + cbase) * 2;
#pragma unroll
        for (int j = 0; j < T::BVECS; ++j) rb[j] = buf_load4s(rsw, boff[j], kb);
        if (++chunk == nchunks) {
            chunk = 0;
            ++ti;
            tap = __builtin_amdgcn_readfirstlane(s_tl[ti]);
            refresh();
        }
    };
    constexpr bool BFULL = (BN * T::LPRB) % 256 == 0;
    auto store = [&](f32x4 (&ra)[T::AROWS], const f32x4 (&rb)[T::BVECS], auto bufc) {
        constexpr int buf = decltype(bufc)::value;
#pragma unroll
        for (int j = 0; j < T::AROWS; ++j) {
            if constexpr (REFL) {
                if (wave_border) ra[j] += (ex0[j] + ex1[j]) + ex2[j];
            }
            *reinterpret_cast<u32x2*>(&As[buf][r0 + T::RPPA * j][c4 * 4]) = pack_bf16x4(ra[j]);
        }
#pragma unroll
        for (int j = 0; j < T::BVECS; ++j) {
            const int e = tid + 256 * j, nn = e / T::LPRB, cb = e % T::LPRB;
            if (BFULL || nn < BN) *reinterpret_cast<f32x4*>(&Bs[buf][nn][cb * 8]) = rb[j];
        }
    };

    const int wm0 = (wave / T::WAVES_N) * T::WM, wn0 = (wave % T::WAVES_N) * T::WN;
    typename T::AccT acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < T::ACC; ++r) acc[i][j][r] = 0.f;

    constexpr int MFR = T::MF;
    constexpr int KSTEP = MFR == 32 ? 16 : 32;                     // k per MFMA
    const int frow = lane & (MFR - 1), fk = (lane / MFR) * 8;      // lane group h holds k = 8 h .. 8 h + 7 of a step
    auto compute = [&](auto bufc) {
        constexpr int buf = decltype(bufc)::value;
#pragma unroll
        for (int ks = 0; ks < CKT / KSTEP; ++ks) {
            bf16x8 a[T::TM], b[T::TN];
#pragma unroll
            for (int i = 0; i < T::TM; ++i) a[i] = *reinterpret_cast<const bf16x8*>(&As[buf][wm0 + i * MFR + frow][ks * KSTEP + fk]);
#pragma unroll
            for (int j = 0; j < T::TN; ++j) b[j] = *reinterpret_cast<const bf16x8*>(&Bs[buf][wn0 + j * MFR + frow][ks * KSTEP + fk]);
#pragma unroll
            for (int i = 0; i < T::TM; ++i)
#pragma unroll
                for (int j = 0; j < T::TN; ++j) {
                    if constexpr (MFR == 32) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                }
        }
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;

    // "write early": at the top of step t buffer t & 1 holds tile t and the registers hold the in-flight loads of tile t + 1; the step
    // writes them to the other buffer, issues tile t + 2 into the same registers, multiplies tile t, one barrier.  The MFMA phase is
    // short here (32..64 cycles per 32x32x16), so the flight of the loads is covered by the other workgroups of the CU.
    f32x4 ra[T::AROWS], rb[T::BVECS];
    if (T_total > 0) {
        issue(ra, rb);
        store(ra, rb, B0{});
        if (T_total > 1) issue(ra, rb);
    }
    __syncthreads();
    int t = 0;
    for (; t + 1 < T_total; t += 2) {
        store(ra, rb, B1{});
        if (t + 2 < T_total) issue(ra, rb);
        compute(B0{});
        __syncthreads();
        if (t + 2 < T_total) {
            store(ra, rb, B0{});
            if (t + 3 < T_total) issue(ra, rb);
        }
        compute(B1{});
        __syncthreads();
    }
    if (t < T_total) {
        compute(B0{});
        __syncthreads();
    }
    igemm_epilogue<T>(p, acc, s_out, s_stat, tid, wm0, wn0, n0, mt);
}

// ------------------------------------------------------------------------------------------------ weight gradient
// out[kflat, n] = sum_pix x(pix, tap)[c] * dy[pix, n] with both operands rounded to bf16 and the reduction over pixels on the MFMA's K axis.
// K-tile = KPB = 64 pixels.  Threads 0..127 stage the A operand (x through the per-(pixel, tap) offset table), threads 128..255 the B
// operand (dy rows): thread (col, pg) loads 4 channels (16 bytes) of the 8 consecutive pixels 8 pg .. 8 pg + 7 and writes, per channel, the
// 8 pixels as one 16-byte LDS store into the k-contiguous panel row of that channel.
constexpr int KPB = 64;

__device__ __forceinline__ int wsw(int row, int kb) { return ((kb ^ ((row >> 1) & 7)) << 3); }      // element offset of k-block kb in a panel row

template <int DUMMY>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(WgradParams p) {
    constexpr int BM = 64, BN = 64;
    __shared__ __attribute__((aligned(16))) u16 Xs[2][BM][KPB];      // [kflat row][64 pixels], 128-byte rows, 16-byte slots XOR-swizzled
    __shared__ __attribute__((aligned(16))) u16 Ys[2][BN][KPB];
    __shared__ unsigned s_tab[WG_TABCAP];
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
        u16 (*dst)[KPB] = waveA ? Xs[buf] : Ys[buf];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bf16x8 h;
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = (__bf16)rv[j][c];
            const int row = col * 4 + c;
            *reinterpret_cast<bf16x8*>(&dst[row][wsw(row, pg)]) = h;
        }
    };

    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int fr = lane & 31, fh = lane >> 5;
    auto compute = [&](auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        bf16x8 a[KPB / 16], b[KPB / 16];
#pragma unroll
        for (int ks = 0; ks < KPB / 16; ++ks) {
            a[ks] = *reinterpret_cast<const bf16x8*>(&Xs[buf][wm0 + fr][wsw(wm0 + fr, 2 * ks + fh)]);
            b[ks] = *reinterpret_cast<const bf16x8*>(&Ys[buf][wn0 + fr][wsw(wn0 + fr, 2 * ks + fh)]);
        }
#pragma unroll
        for (int ks = 0; ks < KPB / 16; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], b[ks], acc, 0, 0, 0);
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
template <class T>
inline void launch_igemm_bf16(const IgemmParams& p, const void* w16, bool refl, hipStream_t s) {
    const int grid = p.mtiles * p.ntiles;
    const size_t tab_bytes = sizeof(unsigned) * (size_t)p.taps * T::BM * (p.g.C2 > 0 ? 2 : 1);
    if (refl) timed_launch(igemm_bf16_kernel<T, 1>, grid, dim3(256), tab_bytes, s, p, reinterpret_cast<const u16*>(w16));
    else timed_launch(igemm_bf16_kernel<T, 0>, grid, dim3(256), tab_bytes, s, p, reinterpret_cast<const u16*>(w16));
}

// Which bf16 tile (0 = the launch is not one the bf16 kernels cover: the caller runs the fp32 path).  Sets *refl.
static int bf16_tile_for(const mcav_igemm_desc* d, bool* refl) {
    if (!d || !d->w16 || d->mma != 1) return 0;
    if (d->pool || d->w_upmerge) return 0;                            // pooled / merged-tap forms stay on the fp32 kernels
    if (d->kh * d->kw > TAB_TAPS || d->Kp % 32 != 0 || d->C1 + d->C2 != d->Kp) return 0;
    if ((d->C1 & 3) || (d->C2 & 3) || (d->C2 > 0 && d->C1 % 32 != 0)) return 0;
    if (d->n_count < 32) return 0;                                    // narrow outputs: the halo / stencil kernels
    if (d->C2 == 0 && d->C1 <= 32 && d->n_count <= 32 && d->kh == 3 && d->stride == 1) return 0;      // conv_halo.hip's shapes (mcav_try_halo runs first)
    const bool direct = d->mode == MCAV_G_DIRECT || (d->mode == MCAV_G_ADJ_STRIDE2 && d->C2 == 0);
    const bool radj = d->mode == MCAV_G_ADJ_REFLECT && d->C2 == 0;
    if (!direct && !radj) return 0;
    *refl = radj;
    const bool k64 = d->Kp % 64 == 0 && (d->C2 == 0 || d->C1 % 64 == 0);
    const long M = (long)d->B * d->Hd * d->Wd;
    const long wg64 = ((M + 63) / 64) * ((d->n_count + 63) / 64);
    int shape = 10;                                                   // 64 x 64
    if (wg64 >= 4096 && !radj) shape = 8;                             // many rows: 128 x 64 (half the filter re-reads)
    else if (wg64 < 512) shape = 12;                                  // few rows (6x20 maps): 32 x 64
    return shape * 2 + (k64 ? 1 : 0);
}

}  // namespace mcav

using namespace mcav;

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
    if (!fill_params(&dd, p, tile) || p.upm) return MCAV_E_INVALID;
    if ((long)d->Np * p.Kstride * 2 >= 0x7fffffffL) return MCAV_E_INVALID;
    const bool k64 = bt & 1;
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

// Plans the bf16 weight-gradient launch into pl (splits over 64-pixel K-tiles, table chunking); false = not eligible.
bool mcav_bf16_wgrad_plan(const mcav_wgrad_desc* d, WgradPlan& pl) {
    if (!d || d->mma != 1 || d->upm) return false;
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
    if ((long)p.pix_per_split * epp > WG_TABCAP) {
        if (2 * 2 * KPB * epp > WG_TABCAP) return false;              // not even two 2-tile chunks fit
        int lg = 1;
        while ((2 << lg) * KPB * epp <= WG_TABCAP / 2) ++lg;
        p.tab_cht_log2 = lg;
    }
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

MCAV_EXPORT int mcav_f32_to_bf16(const float* src, void* dst_bf16, size_t n, void* stream) {
    if (!src || !dst_bf16) return MCAV_E_INVALID;
    if (n == 0) return MCAV_OK;
    const size_t b = (n + 255) / 256;
    f32_to_bf16_kernel<<<(unsigned)(b < 4096 ? b : 4096), 256, 0, as_stream(stream)>>>(src, reinterpret_cast<__bf16*>(dst_bf16), n);
    return launch_status();
}

void mcav_bf16_wgrad_launch(const WgradParams& p, hipStream_t s) {
    timed_launch(wgrad_bf16_kernel<0>, p.splits * p.mtiles * p.ntiles, dim3(256), 0, s, p);
}
