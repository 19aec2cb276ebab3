// Implicit-GEMM convolution for gfx950 on the fp32 MFMA (exact fp32: v_mfma_f32_32x32x2_f32 / 16x16x4_f32).
//
//   y[pix, n] = epilogue( sum_{tap, c} src(pix, tap)[c] * w[n][tap][c] )        (mcav_igemm: forward and dgrad)
//   dw[n][tap][c] = sum_pix dy[pix, n] * src(pix, tap)[c]                         (mcav_wgrad)
//
// NHWC activations, packed [Np][taps][Kp] weights.  A 256-thread workgroup (4 wavefronts of 64) owns a BM x BN output
// tile; the K loop walks (tap, 16-channel chunk) tiles.  Each tile's A panel [BM][16] is gathered from HBM/L2 with
// 16-byte loads (4 lanes cover one pixel's 64 contiguous bytes), the B panel [BN][16] comes from the packed weights;
// both are register-staged into double-buffered LDS (rows padded to 20 floats so the ds_read_b128 fragment reads
// are bank-conflict free), the loads of tile t+1 are issued before the MFMAs of tile t, one barrier per tile.
// Fragments: one ds_read_b128 gives a lane 4 consecutive k of its row; lane-group g takes k = 4g..4g+3 of each
// 8- (32x32x2) or 16-deep (16x16x4) sub-step, A and B permuted identically, so the products are exact fp32 fmas.
#include <stdlib.h>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

#include "conv_shared.h"
#include "kernel_timer.h"

namespace mcav {

template <class T, int KIND>
__global__ __launch_bounds__(256) void igemm_kernel(IgemmParams p) {
    constexpr int BM = T::BM, BN = T::BN;
    constexpr int CK = T::KD;
    __shared__ __attribute__((aligned(16))) float As[2][BM][T::LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN][T::LD];
    __shared__ int s_out[BM];
    __shared__ float s_stat[T::WAVES_M][2][BN];
    // per-(tap, row) source byte offsets, computed once per workgroup (all the padding / reflection / stride / upsample logic
    // lives here, outside the K loop).  fp32 MFMA shares the SIMD's FMA datapath with the VALU (equal rates), so every
    // address instruction inside the loop is paid for in MFMA time: the loop keeps one v_add per load.
    constexpr int MAXTAB = 9;
    __shared__ unsigned s_o1[KIND == K_GENERIC ? 1 : MAXTAB][KIND == K_GENERIC ? 1 : BM];
    __shared__ unsigned s_o2[KIND == K_GENERIC ? 1 : MAXTAB][KIND == K_GENERIC ? 1 : BM];
    __shared__ int s_rn[KIND == K_GENERIC ? 1 : BM], s_ry[KIND == K_GENERIC ? 1 : BM], s_rx[KIND == K_GENERIC ? 1 : BM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // The XCD remap gives each XCD one contiguous run of tiles.  Parity-class tiles differ 4x in work per class, and a run is
    // (mostly) one class: there the hardware's round-robin over XCDs is kept, so every XCD gets the same mix, long tiles first.
    const int lid = p.g.mode == MCAV_G_ADJ_STRIDE2 ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    const int nt = lid % p.ntiles, mt = lid / p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const GatherSrc& g = p.g;

    // ---- per-thread A rows (LPR lanes per row: c4 = 16-byte column of the CK-float K chunk)
    const int c4 = tid % T::LPR, r0 = tid / T::LPR;
    int rn[T::AROWS], ry[T::AROWS], rx[T::AROWS];
#pragma unroll
    for (int j = 0; j < T::AROWS; ++j) {
        int n, dy, dx;
        const bool ok = decode_row(p, m0 + r0 + T::RPP * j, n, dy, dx);
        rn[j] = ok ? n : -1; ry[j] = dy; rx[j] = dx;
    }
    // ---- epilogue table: destination pixel index of every tile row
    for (int r = tid; r < BM; r += 256) {
        int n, dy, dx;
        const bool ok = decode_row(p, m0 + r, n, dy, dx);
        int o = -1;
        if (ok) o = p.pool ? ((n * (p.Hd >> 1) + (dy >> 1)) * (p.Wd >> 1) + (dx >> 1)) : ((n * p.Hd + dy) * p.Wd + dx);
        s_out[r] = o;
        if (KIND != K_GENERIC) { s_rn[r] = ok ? n : -1; s_ry[r] = dy; s_rx[r] = dx; }
    }

    // ---- K-tile enumeration: (tap, chunk); ADJ_STRIDE2 tiles only visit the taps of their parity class
    const int nchunks = g.mode == MCAV_G_SMALLC ? 1 : p.Kp / CK;
    int cls_py = 0, cls_px = 0;
    if (KIND != K_REFLADJ && g.mode == MCAV_G_ADJ_STRIDE2) { const int cls = 3 - m0 / p.McP; cls_py = cls >> 1; cls_px = cls & 1; }
    auto tap_ok = [&](int tap) -> bool {
        if (KIND == K_REFLADJ || g.mode != MCAV_G_ADJ_STRIDE2) return true;
        const int ky = tap / p.kw, kx = tap - ky * p.kw;
        return (((cls_py + g.offset - ky) | (cls_px + g.offset - kx)) & 1) == 0;
    };
    // valid taps of this workgroup, listed once (the K loop just walks the list)
    __shared__ int s_tl[64];
    __shared__ int s_nt;
    if (tid == 0) {
        int nv = 0;
        for (int t = 0; t < p.taps && t < 64; ++t)
            if (tap_ok(t)) s_tl[nv++] = t;
        s_nt = nv;
    }
    __syncthreads();
    const bool smallc = KIND != K_REFLADJ && g.mode == MCAV_G_SMALLC;
    const int T_total = smallc ? (p.taps * 4 + CK - 1) / CK : s_nt * nchunks;
    int ti = 0, chunk = 0;
    int tap = smallc ? 0 : __builtin_amdgcn_readfirstlane(s_tl[0]);

    // ---- cached per-row source BYTE offsets of the current tap (K_FAST / K_REFLADJ); OOB = reads as zero
    unsigned o1[T::AROWS], o2[T::AROWS];
    bool wave_border = false;          // K_REFLADJ: some row of this wavefront lies within 2 pixels of the image border
    if (KIND == K_REFLADJ) {
        bool b = false;
#pragma unroll
        for (int j = 0; j < T::AROWS; ++j) b = b || (rn[j] >= 0 && (ry[j] <= 1 || ry[j] >= g.Hs - 2 || rx[j] <= 1 || rx[j] >= g.Ws - 2));
        wave_border = __any(b);
    }
    const unsigned bytes1 = (unsigned)((size_t)g.B * (g.up1 ? (g.Hs >> 1) * (g.Ws >> 1) : g.Hs * g.Ws) * g.C1 * 4);
    const unsigned bytes2 = (unsigned)((size_t)g.B * g.Hs * g.Ws * g.C2 * 4);
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(g.x1, bytes1);
    const __amdgpu_buffer_rsrc_t rs2 = make_rsrc(g.C2 > 0 ? g.x2 : g.x1, g.C2 > 0 ? bytes2 : 0u);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(p.w, (unsigned)((size_t)(p.n_begin + p.n_count) * p.Kstride * 4));
    // source byte offsets (x1, x2) of destination pixel (n, dy, dx) under tap (ky, kx); OOB when the tap falls outside
    auto src_offsets = [&](int n, int dy, int dx, int ky, int kx, unsigned& oa, unsigned& ob) {
        int sy, sx;
        bool ok = n >= 0;
        if (KIND == K_FAST && g.mode == MCAV_G_ADJ_STRIDE2) {
            const int ty = dy + g.offset - ky, tx = dx + g.offset - kx;      // x[d] collects dy[(d + pad - k) / 2] when whole
            ok = ok && ty >= 0 && tx >= 0 && (((ty | tx) & 1) == 0);
            sy = ty >> 1; sx = tx >> 1;
        } else if (KIND == K_FAST) {
            sy = dy * g.stride + g.sign * ky + g.offset;
            sx = dx * g.stride + g.sign * kx + g.offset;
            if (g.pad_mode == MCAV_PAD_REFLECT) {
                sy = reflect_idx(sy, g.Hs);
                sx = reflect_idx(sx, g.Ws);
            }
        } else {
            sy = dy + 1 - ky;
            sx = dx + 1 - kx;
        }
        ok = ok && (unsigned)sy < (unsigned)g.Hs && (unsigned)sx < (unsigned)g.Ws;
        const int pix = (n * g.Hs + sy) * g.Ws + sx;
        const int pix1 = g.up1 ? ((n * (g.Hs >> 1) + (sy >> 1)) * (g.Ws >> 1) + (sx >> 1)) : pix;
        oa = ok ? (unsigned)(pix1 * g.C1) * 4u : OOB;
        ob = ok ? (unsigned)(pix * g.C2) * 4u : OOB;
    };
    const bool use_tab = KIND != K_GENERIC && g.mode != MCAV_G_SMALLC && p.taps <= MAXTAB;
    if (use_tab) {
        __syncthreads();                                   // s_rn / s_ry / s_rx are complete
        for (int e = tid; e < p.taps * BM; e += 256) {
            const int tp = e / BM, r = e - tp * BM;
            const int ky = tp / p.kw, kx = tp - ky * p.kw;
            unsigned oa, ob;
            src_offsets(s_rn[r], s_ry[r], s_rx[r], ky, kx, oa, ob);
            s_o1[tp][r] = oa;
            s_o2[tp][r] = ob;
        }
        __syncthreads();
    }
    auto refresh_offsets = [&]() {
        if (KIND == K_GENERIC || g.mode == MCAV_G_SMALLC) return;
        if (use_tab) {
#pragma unroll
            for (int j = 0; j < T::AROWS; ++j) { o1[j] = s_o1[tap][r0 + T::RPP * j]; o2[j] = s_o2[tap][r0 + T::RPP * j]; }
            return;
        }
        const int ky = tap / p.kw, kx = tap - ky * p.kw;
#pragma unroll
        for (int j = 0; j < T::AROWS; ++j) src_offsets(rn[j], ry[j], rx[j], ky, kx, o1[j], o2[j]);
    };
    refresh_offsets();

    unsigned boff[T::BVECS];       // byte offset of this thread's weight columns at kflat = 0 (OOB outside the launch's rows)
#pragma unroll
    for (int j = 0; j < T::BVECS; ++j) {
        const int e = tid + 256 * j, nn = e / T::LPR, cb = e % T::LPR;
        const bool ok = nn < BN && n0 + nn < p.n_count;
        boff[j] = ok ? (unsigned)(((p.n_begin + n0 + nn) * p.Kstride + cb * 4) * 4) : OOB;
    }
    // `live` = false issues the same loads with out-of-range offsets (they read zero and touch no memory): the K_FAST loop
    // issues loads unconditionally so that no value merge (and therefore no register copy) sits between an asm load and its wait
    auto load_tile = [&](f32x4 (&ra)[T::AROWS], f32x4 (&rb)[T::BVECS], bool live) {      // global -> registers for the tile at (tap, chunk)
        int ky, kx, c, kflat;
        if (KIND != K_REFLADJ && g.mode == MCAV_G_SMALLC) {
            const int t4 = chunk * 4 + c4;             // every 16-byte column is its own tap
            ky = t4 / p.kw; kx = t4 - ky * p.kw; c = 0; kflat = chunk * CK;
            if (t4 >= p.taps) ky = -1;
        } else {
            ky = tap / p.kw; kx = tap - ky * p.kw; c = chunk * CK + c4 * 4; kflat = tap * p.Kp + chunk * CK;
        }
        if constexpr (KIND == K_GENERIC) {
#pragma unroll
            for (int j = 0; j < T::AROWS; ++j) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (rn[j] >= 0 && ky >= 0) v = gather4(g, rn[j], ry[j], rx[j], ky, kx, c);
                ra[j] = v;
            }
        } else if (KIND == K_FAST && g.mode == MCAV_G_SMALLC) {
            // 4-channel image: one 16-byte load per (row, tap); zero padding = out-of-range offset
#pragma unroll
            for (int j = 0; j < T::AROWS; ++j) {
                const int sy = ry[j] * g.stride + ky + g.offset, sx = rx[j] * g.stride + kx + g.offset;
                const bool ok = live && rn[j] >= 0 && ky >= 0 && (unsigned)sy < (unsigned)g.Hs && (unsigned)sx < (unsigned)g.Ws;
                ra[j] = buf_load4(rs1, ok ? (unsigned)(((rn[j] * g.Hs + sy) * g.Ws + sx) * 4) * 4u : OOB);
            }
        } else if (KIND == K_FAST) {
            const bool use2 = chunk * CK >= g.C1;
            const int cc = use2 ? c - g.C1 : c;
            const bool cok = live && (use2 ? cc < g.C2 : cc < g.C1);
            if (use2) {
#pragma unroll
                for (int j = 0; j < T::AROWS; ++j) ra[j] = buf_load4(rs2, cok ? o2[j] + (unsigned)cc * 4u : OOB);
            } else {
#pragma unroll
                for (int j = 0; j < T::AROWS; ++j) ra[j] = buf_load4(rs1, cok ? o1[j] + (unsigned)cc * 4u : OOB);
            }
        } else {
            // the whole 16-channel chunk lies in x1 or in x2 (C1 % 16 == 0 whenever there is an x2): wave-uniform choice
            const bool use2 = chunk * CK >= g.C1;
            const int cc = use2 ? c - g.C1 : c;
            const bool cok = use2 ? cc < g.C2 : cc < g.C1;     // K padding beyond the real channels reads as zero
            if (use2) {
#pragma unroll
                for (int j = 0; j < T::AROWS; ++j) ra[j] = buf_load4(rs2, cok ? o2[j] + (unsigned)cc * 4u : OOB);
            } else {
#pragma unroll
                for (int j = 0; j < T::AROWS; ++j) ra[j] = buf_load4(rs1, cok ? o1[j] + (unsigned)cc * 4u : OOB);
            }
            if (KIND == K_REFLADJ && wave_border) {
                // border rows: add the output(s) whose reflected tap landed on this pixel (at most one extra row, one extra column)
#pragma unroll
                for (int j = 0; j < T::AROWS; ++j) {
                    const int dy = ry[j], dx = rx[j];
                    const int sy = dy + 1 - ky, sx = dx + 1 - kx;
                    const int ey = (dy == 1 && ky == 0) ? 0 : ((dy == g.Hs - 2 && ky == 2) ? g.Hs - 1 : -1);
                    const int ex = (dx == 1 && kx == 0) ? 0 : ((dx == g.Ws - 2 && kx == 2) ? g.Ws - 1 : -1);
                    const bool rok = rn[j] >= 0 && cok;
                    const bool syok = (unsigned)sy < (unsigned)g.Hs, sxok = (unsigned)sx < (unsigned)g.Ws;
                    const int rowb = rn[j] * g.Hs;
                    const unsigned a0 = (unsigned)((((rowb + ey) * g.Ws + sx) * g.C1 + c) * 4);
                    const unsigned a1 = (unsigned)((((rowb + sy) * g.Ws + ex) * g.C1 + c) * 4);
                    const unsigned a2 = (unsigned)((((rowb + ey) * g.Ws + ex) * g.C1 + c) * 4);
                    const f32x4 e0 = buf_load4(rs1, (rok && ey >= 0 && sxok) ? a0 : OOB);
                    const f32x4 e1 = buf_load4(rs1, (rok && ex >= 0 && syok) ? a1 : OOB);
                    const f32x4 e2 = buf_load4(rs1, (rok && ey >= 0 && ex >= 0) ? a2 : OOB);
                    ra[j] += (e0 + e1) + e2;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < T::BVECS; ++j) {
            const unsigned off = live ? boff[j] + (unsigned)kflat * 4u : OOB;
            rb[j] = buf_load4(rsw, off);
        }
    };
    auto store_tile = [&](const f32x4 (&ra)[T::AROWS], const f32x4 (&rb)[T::BVECS], int buf) {
#pragma unroll
        for (int j = 0; j < T::AROWS; ++j) *reinterpret_cast<f32x4*>(&As[buf][r0 + T::RPP * j][c4 * 4]) = ra[j];
#pragma unroll
        for (int j = 0; j < T::BVECS; ++j) {
            const int e = tid + 256 * j, nn = e / T::LPR, cb = e % T::LPR;
            if (nn < BN) *reinterpret_cast<f32x4*>(&Bs[buf][nn][cb * 4]) = rb[j];
        }
    };
    auto advance = [&]() {       // only called while another tile exists
        if (smallc) { ++chunk; return; }
        if (++chunk == nchunks) {
            chunk = 0;
            tap = __builtin_amdgcn_readfirstlane(s_tl[++ti]);
            refresh_offsets();
        }
    };

    // ---- accumulators
    const int wm0 = (wave / T::WAVES_N) * T::WM, wn0 = (wave % T::WAVES_N) * T::WN;
    typename T::AccT acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < T::ACC; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int buf) {
        if constexpr (T::MF == 32) {
            const int frow = lane & 31, fk = (lane >> 5) * 4;
#pragma unroll
            for (int ks = 0; ks < CK / 8; ++ks) {
                f32x4 a[T::TM], b[T::TN];
#pragma unroll
                for (int i = 0; i < T::TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(&As[buf][wm0 + i * 32 + frow][ks * 8 + fk]);
#pragma unroll
                for (int j = 0; j < T::TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(&Bs[buf][wn0 + j * 32 + frow][ks * 8 + fk]);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < T::TM; ++i)
#pragma unroll
                        for (int j = 0; j < T::TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[j][q], acc[i][j], 0, 0, 0);
            }
        } else {
            const int frow = lane & 15, fk = (lane >> 4) * 4;
#pragma unroll
            for (int ks = 0; ks < CK / 16; ++ks) {
                f32x4 a[T::TM], b[T::TN];
#pragma unroll
                for (int i = 0; i < T::TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(&As[buf][wm0 + i * 16 + frow][ks * 16 + fk]);
#pragma unroll
                for (int j = 0; j < T::TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(&Bs[buf][wn0 + j * 16 + frow][ks * 16 + fk]);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < T::TM; ++i)
#pragma unroll
                        for (int j = 0; j < T::TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][q], b[j][q], acc[i][j], 0, 0, 0);
            }
        }
    };

    // ---- main loop: tile t is computed from LDS buffer t & 1 while the loads of tiles t + 1 and t + 2 are in flight
    // (two register stages), so a load has a full iteration plus an MFMA phase to land before it is written to LDS.
    // ---- main loop ("write early"): iteration t
    //   1. the loads of tile t+1 (issued one iteration ago) have landed: write them to LDS buffer (t+1)&1
    //   2. issue the loads of tile t+2 into the same registers
    //   3. MFMAs of tile t from buffer t&1 -- they cover the LDS-write latency of (1) and the flight of (2)
    //   4. one barrier
    // One register stage, exact s_waitcnt (everything outstanding at (1) is tile t+1), no LDS-write latency on the critical path.
    f32x4 ra[T::AROWS], rb[T::BVECS];
    if (T_total > 0) {
        load_tile(ra, rb, true);
        store_tile(ra, rb, 0);
        if (T_total > 1) { advance(); load_tile(ra, rb, true); }
    }
    __syncthreads();
    for (int t = 0; t < T_total; ++t) {
        const int buf = t & 1;
        if (t + 1 < T_total) {
            store_tile(ra, rb, buf ^ 1);
            if (t + 2 < T_total) { advance(); load_tile(ra, rb, true); }
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the loads ABOVE the MFMAs (the scheduler otherwise sinks them below the barrier)
        compute(buf);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    igemm_epilogue<T>(p, acc, s_out, s_stat, tid, wm0, wn0, n0, mt);
}

// ------------------------------------------------------------------------------------------------ table-driven kernel
// The kernel the step spends its time in (every 1x1 / 3x3 layer whose channel counts are whole K-tiles: DIRECT gathers with
// zero / reflection padding, stride 1 / 2, fused upsample + concat, and the stride-2 adjoint).  Same tiling, LDS layout and
// epilogue as igemm_kernel, but NOTHING besides loads, LDS traffic and MFMAs is left inside the K loop:
//   * all source addressing is done once per workgroup into per-(tap, row) byte-offset tables in LDS (out-of-image taps
//     and padding rows hold an out-of-range offset, which a raw buffer load returns as zero);
//   * a K-tile's loads are `buffer_load_dwordx4 v, v_off, s[rsrc], s_chunk offen`: the per-row offset register changes
//     only when the tap changes, the channel-chunk offset is the instruction's SCALAR offset -- zero VALU per load;
//   * the loop is unrolled by two so that every LDS address is a base register plus an immediate;
//   * the accumulators never leave their registers (no conditional around the MFMAs).
// fp32 MFMA and the VALU share the SIMD's FMA datapath on gfx950 (equal peak rates): an address instruction in the loop costs
// MFMA time, and the previous loop spent about a third of its issue slots on them.

// REFL = the adjoint of the 3x3 reflection-padded conv (decoder dgrad): away from the image border it is the plain
// correlation (sy = dy + 1 - ky); a wavefront that owns rows within two pixels of the border (wave-uniform test) issues up
// to three more loads per row -- the outputs whose reflected tap landed on this pixel -- and adds them when the tile is
// written to LDS, one iteration later, so they cost no extra wait.
// TK = 2 (UPM): forward conv of cat(nearest-up2(x1), x2) with reflection padding, the x1 part as FOUR merged taps on the low-resolution
// x1 (see mcav_igemm_desc.w_upmerge): rows are grouped by output parity class, the K loop runs 4 * C1/CK tiles from x1 with the
// class's pre-summed filters and then the usual 9 * C2/CK tiles from x2.  Reflection on the upsampled grid = clamping the source index.
#ifndef MCAV_DIAG
#define MCAV_DIAG 0      // > 0: timing-only builds (results are wrong; `make diag D=n`): 1 no global loads in the steady-state loop, 2 no LDS stores either,
                         // 3 no barrier, 4 = correct results + per-phase cycle stamps, 5 no set-up arithmetic (tables point at a few hot lines), 6 and no
                         // first loads, 8 the real set-up, then the tables redirected as in 5
#endif
#if MCAV_DIAG == 4      // per-phase shader cycles of the table-driven kernel, summed over workgroups (thread 0): [1] tables | [2] first loads | [3] loop | [4] tail | [5] epilogue; [0] = workgroups
__device__ unsigned long long g_diag_stamps[8];
#define MCAV_STAMP_BEGIN() unsigned long long st_prev = __builtin_readcyclecounter()
#define MCAV_STAMP(i) do { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long st_now = __builtin_readcyclecounter(); \
        atomicAdd(&g_diag_stamps[i], st_now - st_prev); if (i == 5) atomicAdd(&g_diag_stamps[0], 1ull); st_prev = st_now; } } while (0)
#else
#define MCAV_STAMP_BEGIN()
#define MCAV_STAMP(i)
#endif
#ifndef MCAV_IG_OCC
#define MCAV_IG_OCC 1      // workgroups per SIMD asked of the register allocator for the 64x64 (32-deep) forward / adjoint kernel (`make variant`)
#endif
template <class T, int TK>
__global__ __launch_bounds__(256, (T::BM == 128 && T::BN == 64 && T::KD == 16 && TK != 1) ? 4
                                  : ((T::BM == 64 && T::BN == 64 && T::KD == 32 && TK == 0) ? MCAV_IG_OCC : 1)) void igemm_tab_kernel(IgemmParams p) {
    constexpr bool REFL = TK == 1, UPM = TK == 2;
    constexpr int BM = T::BM, BN = T::BN, CKT = T::KD;
    __shared__ __attribute__((aligned(16))) float As[2][BM][T::LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN][T::LD];
    __shared__ __attribute__((aligned(16))) unsigned s_out[BM];          // byte offset of each tile row's output pixel in y (OOB: no such row)
    // the BatchNorm column-sum scratch of the epilogue borrows the A panel (free once the K loop's last barrier has passed): with it the
    // REFL kind's 64x64 tile fits 40 KB too
    float (*const s_stat)[2][BN] = reinterpret_cast<float (*)[2][BN]>(&As[1][0][0]);
    static_assert(sizeof(float) * BM * T::LD >= sizeof(float) * T::WAVES_M * 2 * BN, "statistics scratch fits one A buffer");
    // LDS budget: with 32-deep 64x64 tiles everything below fits 40 KB, i.e. FOUR workgroups per CU.  The offset tables are dynamic
    // shared memory sized by the launch (taps x BM x {1, 2} sources); the row coordinates are only needed while the tables are built
    // and borrow the (not yet used) A panel (the REFL kind's reflected sources are table rows too: no address arithmetic in the loop).
    extern __shared__ unsigned s_dyn[];
    unsigned* const s_o1 = s_dyn;
    unsigned* const s_o2 = s_dyn + (UPM ? 4 * BM : (p.g.C2 > 0 ? p.taps * BM : 0));      // UPM: 4 merged-tap rows, then the 9 taps of x2
    int* const s_rn = reinterpret_cast<int*>(&As[0][0][0]);
    // REFL: three more tables behind s_o1 -- the reflected sources of the rows next to the border (OOB elsewhere)
    unsigned* const s_e0 = s_dyn + p.taps * BM;
    unsigned* const s_e1 = s_dyn + 2 * p.taps * BM;
    unsigned* const s_e2 = s_dyn + 3 * p.taps * BM;
    int* const s_ry = s_rn + BM;
    int* const s_rx = s_rn + 2 * BM;
    static_assert(2 * BM * T::LD >= 3 * BM, "row coordinates fit the A panel");
    __shared__ int s_tl[TAB_TAPS + 1];
    __shared__ int s_nt;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    MCAV_STAMP_BEGIN();
    // The XCD remap gives each XCD one contiguous run of tiles.  Parity-class tiles differ 4x in work per class, and a run is
    // (mostly) one class: there the hardware's round-robin over XCDs is kept, so every XCD gets the same mix, long tiles first.
    const int lid = p.g.mode == MCAV_G_ADJ_STRIDE2 ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    const int per_split = p.mtiles * p.ntiles;
    const int ks = p.ksplit > 1 ? lid / per_split : 0, lrem = lid - ks * per_split;      // split of the K loop (ksplit > 1), tile within it
    const int nt = lrem % p.ntiles, mt = lrem / p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const GatherSrc& g = p.g;

    // ---- once per workgroup: destination pixel of every tile row, the tap list, the offset tables
    for (int r = tid; r < (MCAV_DIAG >= 5 ? 0 : BM); r += 256) {
        int n, dy, dx;
        const bool ok = decode_row(p, m0 + r, n, dy, dx);
        int o = -1;
        if (ok) o = p.pool ? ((n * (p.Hd >> 1) + (dy >> 1)) * (p.Wd >> 1) + (dx >> 1)) : ((n * p.Hd + dy) * p.Wd + dx);
        s_out[r] = o >= 0 ? (unsigned)o * (unsigned)(p.Cd * 4) : OOB;
        s_rn[r] = ok ? n : -1; s_ry[r] = dy; s_rx[r] = dx;
    }
    if constexpr (MCAV_DIAG >= 5) {      // timing only: no row decode, no table arithmetic (every entry points at pixel 0), outputs go to row r
        for (int r = tid; r < BM; r += 256) s_out[r] = (unsigned)(m0 + r) < (unsigned)p.M ? (unsigned)(m0 + r) * (unsigned)(p.Cd * 4) : OOB;
        for (int e = tid; e < p.taps * BM; e += 256) s_o1[e] = (unsigned)(e & 63) * 64u;
        if (tid == 0) { for (int t = 0; t <= p.taps; ++t) s_tl[t] = t < p.taps ? t : 0; s_nt = p.taps; }
    } else
    if (tid == 0) {      // ADJ_STRIDE2 tiles hold one parity class of destination pixels and visit only that class's taps
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
    if constexpr (UPM) {
        const int Hl = g.Hs >> 1, Wl = g.Ws >> 1;
        for (int e = tid; e < 4 * BM; e += 256) {          // merged tap (a, b) of the row's parity class: low-resolution source, clamped
            const int tp = e / BM, r = e - tp * BM;
            const int n = s_rn[r], dy = s_ry[r], dx = s_rx[r];
            const int sy = min(max((dy >> 1) - 1 + (dy & 1) + (tp >> 1), 0), Hl - 1);
            const int sx = min(max((dx >> 1) - 1 + (dx & 1) + (tp & 1), 0), Wl - 1);
            s_o1[e] = n >= 0 ? (unsigned)(((n * Hl + sy) * Wl + sx) * g.C1) * 4u : OOB;
        }
        for (int e = tid; e < 9 * BM; e += 256) {          // the skip tensor x2: nine taps, reflection padding at full resolution
            const int tp = e / BM, r = e - tp * BM;
            const int ky = tp / 3, kx = tp - ky * 3;
            const int n = s_rn[r];
            const int sy = reflect_idx(s_ry[r] + ky - 1, g.Hs), sx = reflect_idx(s_rx[r] + kx - 1, g.Ws);
            s_o2[e] = n >= 0 ? (unsigned)(((n * g.Hs + sy) * g.Ws + sx) * g.C2) * 4u : OOB;
        }
    }
    for (int e = tid; e < (UPM || MCAV_DIAG >= 5 ? 0 : p.taps * BM); e += 256) {
        const int tp = e / BM, r = e - tp * BM;
        const int ky = tp / p.kw, kx = tp - ky * p.kw;
        const int n = s_rn[r], dy = s_ry[r], dx = s_rx[r];
        int sy, sx;
        bool ok = n >= 0;
        if (REFL) {
            sy = dy + 1 - ky;
            sx = dx + 1 - kx;
        } else if (g.mode == MCAV_G_ADJ_STRIDE2) {
            const int ty = dy + g.offset - ky, tx = dx + g.offset - kx;      // x[d] collects dy[(d + pad - k) / 2] when whole
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
        if constexpr (REFL) {      // the outputs whose reflected tap landed on this pixel: row 0 / H - 1 seen from rows 1 / H - 2, same for columns
            const int ey = (dy == 1 && ky == 0) ? 0 : ((dy == g.Hs - 2 && ky == 2) ? g.Hs - 1 : -1);
            const int ex = (dx == 1 && kx == 0) ? 0 : ((dx == g.Ws - 2 && kx == 2) ? g.Ws - 1 : -1);
            const bool syok = (unsigned)sy < (unsigned)g.Hs, sxok = (unsigned)sx < (unsigned)g.Ws;
            const int rowb = n * g.Hs;
            s_e0[tp * BM + r] = (ok && ey >= 0 && sxok) ? (unsigned)(((rowb + ey) * g.Ws + sx) * g.C1) * 4u : OOB;
            s_e1[tp * BM + r] = (ok && ex >= 0 && syok) ? (unsigned)(((rowb + sy) * g.Ws + ex) * g.C1) * 4u : OOB;
            s_e2[tp * BM + r] = (ok && ey >= 0 && ex >= 0) ? (unsigned)(((rowb + ey) * g.Ws + ex) * g.C1) * 4u : OOB;
        }
        ok = ok && (unsigned)sy < (unsigned)g.Hs && (unsigned)sx < (unsigned)g.Ws;
        const int pix = (n * g.Hs + sy) * g.Ws + sx;
        const int pix1 = g.up1 ? ((n * (g.Hs >> 1) + (sy >> 1)) * (g.Ws >> 1) + (sx >> 1)) : pix;
        s_o1[tp * BM + r] = ok ? (unsigned)(pix1 * g.C1) * 4u : OOB;
        if (g.C2 > 0) s_o2[tp * BM + r] = ok ? (unsigned)(pix * g.C2) * 4u : OOB;
    }
    __syncthreads();

#if MCAV_DIAG == 4 || MCAV_DIAG == 8
    if constexpr (MCAV_DIAG == 8) {      // timing only: the real set-up, then every table entry redirected to the same few hot lines
        for (int e = tid; e < p.taps * BM; e += 256) s_o1[e] = (unsigned)(e & 63) * 64u;
        __syncthreads();
    }
#endif
    const int ntaps = s_nt;
    const int nchunks = p.Kp / CKT;
    const int nch1 = g.C1 / CKT, nch2 = g.C2 / CKT;      // UPM: K-tiles per merged tap of x1 / per tap of x2
    const int T_all = UPM ? 4 * nch1 + 9 * nch2 : ntaps * nchunks;
    const int t_first = p.ksplit > 1 ? (int)((long)T_all * ks / p.ksplit) : 0;          // this workgroup's K-tiles [t_first, t_first + T_total)
    const int T_total = p.ksplit > 1 ? (int)((long)T_all * (ks + 1) / p.ksplit) - t_first : T_all;
    const unsigned bytes1 = (unsigned)((size_t)g.B * (g.up1 ? (g.Hs >> 1) * (g.Ws >> 1) : g.Hs * g.Ws) * g.C1 * 4);
    const unsigned bytes2 = (unsigned)((size_t)g.B * g.Hs * g.Ws * g.C2 * 4);
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(g.x1, bytes1);
    const __amdgpu_buffer_rsrc_t rs2 = make_rsrc(g.C2 > 0 ? g.x2 : g.x1, g.C2 > 0 ? bytes2 : 0u);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(p.w, (unsigned)((size_t)(p.n_begin + p.n_count) * p.Kstride * 4));

    // ---- this thread's share of a K-tile: AROWS rows of A (16-byte column c4), BVECS 16-byte pieces of B
    const int c4 = tid % T::LPR, r0 = tid / T::LPR;
    unsigned boff[T::BVECS];
#pragma unroll
    for (int j = 0; j < T::BVECS; ++j) {
        const int e = tid + 256 * j, nn = e / T::LPR, cb = e % T::LPR;
        const bool ok = nn < BN && n0 + nn < p.n_count;
        boff[j] = ok ? (unsigned)(((p.n_begin + n0 + nn) * p.Kstride + cb * 4) * 4) : OOB;
    }
    bool wave_border = false;
    if constexpr (REFL) {
        bool bd = false;
#pragma unroll
        for (int j = 0; j < T::AROWS; ++j) {
            const int r = r0 + T::RPP * j;
            const int y = s_ry[r], x = s_rx[r];
            bd = bd || (s_rn[r] >= 0 && (y <= 1 || y >= g.Hs - 2 || x <= 1 || x >= g.Ws - 2));
        }
        wave_border = __any(bd);
    }
    f32x4 ex0[REFL ? T::AROWS : 1], ex1[REFL ? T::AROWS : 1], ex2[REFL ? T::AROWS : 1];
    int ti = UPM ? 0 : t_first / nchunks, chunk = UPM ? 0 : t_first - ti * nchunks;      // the ISSUE pointer: next K-tile to load
    int tap = UPM ? 0 : __builtin_amdgcn_readfirstlane(s_tl[ti]);
    int seg = 0;                                 // UPM: 0 = merged taps of x1, 1 = taps of x2
    unsigned oa[T::AROWS], ob[T::AROWS];
    unsigned oe0[REFL ? T::AROWS : 1], oe1[REFL ? T::AROWS : 1], oe2[REFL ? T::AROWS : 1];
    unsigned boffm[UPM ? T::BVECS : 1];          // UPM: byte offsets into the merged filter copy [cls][Np][4][C1]
    unsigned bcur[UPM ? T::BVECS : 1];           // UPM: the B offsets of the current segment
    __amdgpu_buffer_rsrc_t rsm = rsw;
    if constexpr (UPM) {
        const int cls = (m0 / BM) & 3;
        rsm = make_rsrc(p.wm, (unsigned)((size_t)4 * p.Np_all * 4 * g.C1 * 4));
#pragma unroll
        for (int j = 0; j < T::BVECS; ++j) {
            const int e = tid + 256 * j, nn = e / T::LPR, cb = e % T::LPR;
            const bool ok = nn < BN && n0 + nn < p.n_count;
            boffm[j] = ok ? (unsigned)((((cls * p.Np_all + p.n_begin + n0 + nn) * 4) * g.C1 + cb * 4) * 4) : OOB;
        }
    }
    auto refresh = [&]() {
        if constexpr (UPM) {       // one offset array for both segments: rows 0..3 of the table = merged taps of x1, rows 4..12 = taps of x2
            const int row = seg == 0 ? tap : (tap < 9 ? 4 + tap : 4);
#pragma unroll
            for (int j = 0; j < T::AROWS; ++j) oa[j] = s_dyn[row * BM + r0 + T::RPP * j] + (unsigned)c4 * 16u;
#pragma unroll
            for (int j = 0; j < T::BVECS; ++j) bcur[j] = seg == 0 ? boffm[j] : boff[j];      // (kept in ONE register array: see issue())
            return;
        }
#pragma unroll
        for (int j = 0; j < T::AROWS; ++j) {
            oa[j] = s_o1[tap * BM + r0 + T::RPP * j] + (unsigned)c4 * 16u;      // OOB + 16 c4 is still out of range
            if constexpr (!REFL) ob[j] = s_o2[tap * BM + r0 + T::RPP * j] + (unsigned)c4 * 16u;      // (aliases s_o1 when there is no second source)
            else ob[j] = oa[j];
        }
        if constexpr (REFL) {
            if (wave_border) {
#pragma unroll
                for (int j = 0; j < T::AROWS; ++j) {
                    oe0[j] = s_e0[tap * BM + r0 + T::RPP * j] + (unsigned)c4 * 16u;
                    oe1[j] = s_e1[tap * BM + r0 + T::RPP * j] + (unsigned)c4 * 16u;
                    oe2[j] = s_e2[tap * BM + r0 + T::RPP * j] + (unsigned)c4 * 16u;
                }
            }
        }
    };
    refresh();
    auto issue = [&](f32x4 (&ra)[T::AROWS], f32x4 (&rb)[T::BVECS]) {
        if constexpr (UPM) {
            const int cb4 = chunk * CKT * 4;
            // (the asm comments keep the two arms from being folded into per-lane selects of the buffer resources, which would turn
            //  every load into a waterfall loop over "possibly divergent" descriptors)
            if (__builtin_amdgcn_readfirstlane(seg) == 0) {
                asm volatile("; x1 segment: merged taps" ::: "memory");
#pragma unroll
                for (int j = 0; j < T::AROWS; ++j) ra[j] = buf_load4s(rs1, oa[j], cb4);
                const int kb = (tap * g.C1) * 4 + cb4;
#pragma unroll
                for (int j = 0; j < T::BVECS; ++j) rb[j] = buf_load4s(rsm, bcur[j], kb);
            } else {
                asm volatile("; x2 segment: nine taps" ::: "memory");
#pragma unroll
                for (int j = 0; j < T::AROWS; ++j) ra[j] = buf_load4s(rs2, oa[j], cb4);
                const int kb = (tap * p.Kp + g.C1) * 4 + cb4;
#pragma unroll
                for (int j = 0; j < T::BVECS; ++j) rb[j] = buf_load4s(rsw, bcur[j], kb);
            }
            if (++chunk == (seg == 0 ? nch1 : nch2)) {
                chunk = 0;
                if (++tap == 4 && seg == 0) { seg = 1; tap = 0; }
                refresh();
            }
            return;
        }
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
#pragma unroll
                for (int j = 0; j < T::AROWS; ++j) {
                    ex0[j] = buf_load4s(rs1, oe0[j], cbase * 4);
                    ex1[j] = buf_load4s(rs1, oe1[j], cbase * 4);
                    ex2[j] = buf_load4s(rs1, oe2[j], cbase * 4);
                }
            }
        }
        const int kb = (tap * p.Kp + cbase) * 4;
#pragma unroll
        for (int j = 0; j < T::BVECS; ++j) rb[j] = buf_load4s(rsw, boff[j], kb);
        if (++chunk == nchunks) {
            chunk = 0;
            ++ti;
            tap = __builtin_amdgcn_readfirstlane(s_tl[ti]);      // s_tl[ntaps] = 0: harmless after the last tile
            refresh();
        }
    };
    constexpr bool BFULL = (BN * T::LPR) % 256 == 0;             // every thread owns a B piece: no predicated LDS write in the loop
    auto store = [&](f32x4 (&ra)[T::AROWS], const f32x4 (&rb)[T::BVECS], auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        if constexpr (REFL) {
            if (wave_border) {
#pragma unroll
                for (int j = 0; j < T::AROWS; ++j) ra[j] += (ex0[j] + ex1[j]) + ex2[j];
            }
        }
#pragma unroll
        for (int j = 0; j < T::AROWS; ++j) *reinterpret_cast<f32x4*>(&As[buf][r0 + T::RPP * j][c4 * 4]) = ra[j];
#pragma unroll
        for (int j = 0; j < T::BVECS; ++j) {
            const int e = tid + 256 * j, nn = e / T::LPR, cb = e % T::LPR;
            if (BFULL || nn < BN) *reinterpret_cast<f32x4*>(&Bs[buf][nn][cb * 4]) = rb[j];
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

    constexpr int MFR = T::MF;                                     // fragment rows: 32 (32x32x2) or 16 (16x16x4)
    constexpr int KSUB = MFR == 32 ? 8 : 16;                       // K depth covered by one ds_read_b128 per lane (4 MFMAs)
    const int frow = lane & (MFR - 1), fk = (lane / MFR) * 4;
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;

    // ---- main loop ("write early").  Invariant at the top of step t: LDS buffer t & 1 holds tile t, the registers hold the
    // in-flight loads of tile t + 1, the issue pointer is at tile t + 2.  A step writes tile t + 1 to the other buffer (the only
    // loads outstanding are exactly that tile's), issues tile t + 2 into the same registers, runs the MFMAs of tile t -- which
    // cover the LDS-write latency and the flight of the new loads -- and ends with the one barrier.
    f32x4 ra[T::AROWS], rb[T::BVECS];
    // Steady-state step.  A wavefront issues in order and (with one accumulator tile) its MFMAs form one dependent chain, so the rest of
    // the step -- LDS stores of tile t + 1, global loads of tile t + 2, the pointer advance -- is cut into pieces and one or two pieces
    // are placed behind each MFMA group, where they issue while the MFMA pipe works, instead of in a block in front of the MFMAs.
    constexpr int NKS = CKT / KSUB, NSLOT = NKS * 4, NPIECE = T::AROWS + T::BVECS + 4;
    f32x4 fa[2][T::TM], fb[2][T::TN];
    auto rdf = [&](auto bufc, int s, int ks) {
        constexpr int buf = decltype(bufc)::value;
#pragma unroll
        for (int i = 0; i < T::TM; ++i) fa[s][i] = *reinterpret_cast<const f32x4*>(&As[buf][wm0 + i * MFR + frow][ks * KSUB + fk]);
#pragma unroll
        for (int j = 0; j < T::TN; ++j) fb[s][j] = *reinterpret_cast<const f32x4*>(&Bs[buf][wn0 + j * MFR + frow][ks * KSUB + fk]);
    };
    auto piece = [&](auto pc, auto nxt) {
        constexpr int P = decltype(pc)::value, buf = decltype(nxt)::value;
        if constexpr (MCAV_DIAG >= 2 && P < T::AROWS + T::BVECS) {
            // diagnostic build: no LDS stores either
        } else if constexpr (P < T::AROWS) {                            // A store (+ the reflected contributions of a border wavefront)
            if constexpr (REFL) {
                if (wave_border) ra[P] += (ex0[P] + ex1[P]) + ex2[P];
            }
            *reinterpret_cast<f32x4*>(&As[buf][r0 + T::RPP * P][c4 * 4]) = ra[P];
        } else if constexpr (P < T::AROWS + T::BVECS) {                 // B store
            constexpr int j = P - T::AROWS;
            const int e = tid + 256 * j, nn = e / T::LPR, cb = e % T::LPR;
            if (BFULL || nn < BN) *reinterpret_cast<f32x4*>(&Bs[buf][nn][cb * 4]) = rb[j];
        } else if constexpr (MCAV_DIAG >= 1 && (P == T::AROWS + T::BVECS || P == T::AROWS + T::BVECS + 2)) {
            // diagnostic build: the steady-state loop re-uses the registers it has (no global loads)
        } else if constexpr (P == T::AROWS + T::BVECS) {                // A loads of the tile after next
            if constexpr (UPM) {
                const int cb4 = chunk * CKT * 4;
                if (__builtin_amdgcn_readfirstlane(seg) == 0) {
                    asm volatile("; x1 segment: merged taps (A)" ::: "memory");
#pragma unroll
                    for (int j = 0; j < T::AROWS; ++j) ra[j] = buf_load4s(rs1, oa[j], cb4);
                } else {
                    asm volatile("; x2 segment: nine taps (A)" ::: "memory");
#pragma unroll
                    for (int j = 0; j < T::AROWS; ++j) ra[j] = buf_load4s(rs2, oa[j], cb4);
                }
            } else {
                const int cbase = chunk * CKT;
                if (cbase < g.C1) {
#pragma unroll
                    for (int j = 0; j < T::AROWS; ++j) ra[j] = buf_load4s(rs1, oa[j], cbase * 4);
                } else {
#pragma unroll
                    for (int j = 0; j < T::AROWS; ++j) ra[j] = buf_load4s(rs2, ob[j], (cbase - g.C1) * 4);
                }
            }
        } else if constexpr (P == T::AROWS + T::BVECS + 1) {            // REFL: the up to three reflected sources of a border wavefront
            if constexpr (REFL) {
                if (wave_border) {
                    const int cbase = chunk * CKT;
#pragma unroll
                    for (int j = 0; j < T::AROWS; ++j) {
                        ex0[j] = buf_load4s(rs1, oe0[j], cbase * 4);
                        ex1[j] = buf_load4s(rs1, oe1[j], cbase * 4);
                        ex2[j] = buf_load4s(rs1, oe2[j], cbase * 4);
                    }
                }
            }
        } else if constexpr (P == T::AROWS + T::BVECS + 2) {            // B loads
            if constexpr (UPM) {
                const int cb4 = chunk * CKT * 4;
                if (__builtin_amdgcn_readfirstlane(seg) == 0) {
                    asm volatile("; x1 segment: merged taps (B)" ::: "memory");
                    const int kb = (tap * g.C1) * 4 + cb4;
#pragma unroll
                    for (int j = 0; j < T::BVECS; ++j) rb[j] = buf_load4s(rsm, bcur[j], kb);
                } else {
                    asm volatile("; x2 segment: nine taps (B)" ::: "memory");
                    const int kb = (tap * p.Kp + g.C1) * 4 + cb4;
#pragma unroll
                    for (int j = 0; j < T::BVECS; ++j) rb[j] = buf_load4s(rsw, bcur[j], kb);
                }
            } else {
                const int kb = (tap * p.Kp + chunk * CKT) * 4;
#pragma unroll
                for (int j = 0; j < T::BVECS; ++j) rb[j] = buf_load4s(rsw, boff[j], kb);
            }
        } else {                                                        // advance the issue pointer
            if constexpr (UPM) {
                if (++chunk == (seg == 0 ? nch1 : nch2)) {
                    chunk = 0;
                    if (++tap == 4 && seg == 0) { seg = 1; tap = 0; }
                    refresh();
                }
            } else {
                if (++chunk == nchunks) {
                    chunk = 0;
                    ++ti;
                    tap = __builtin_amdgcn_readfirstlane(s_tl[ti]);
                    refresh();
                }
            }
        }
    };
    // ST / IS: the step stores tile t + 1 / issues tile t + 2 (the last two steps of a workgroup have nothing left to store / issue)
    auto step = [&](auto cur, auto nxt, auto stc, auto isc) {
        constexpr bool ST = decltype(stc)::value, IS = decltype(isc)::value;
        // The merged-tap and reflection-adjoint kinds keep their loads in one block (cut up, the two-segment loads of the first made the
        // compiler spill the offset arrays; the second measured slower): for them the pieces are the stores, then issue() as a whole.
        constexpr bool WHOLE_ISSUE = TK != 0;
        constexpr int NP = WHOLE_ISSUE ? T::AROWS + T::BVECS + 1 : NPIECE;
        rdf(cur, 0, 0);
        static_for<NKS>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            if constexpr (ks + 1 < NKS) rdf(cur, (ks + 1) & 1, ks + 1);
            __builtin_amdgcn_sched_barrier(0);
            static_for<4>([&](auto qc) {
                constexpr int q = decltype(qc)::value, S = ks * 4 + q;
#pragma unroll
                for (int i = 0; i < T::TM; ++i)
#pragma unroll
                    for (int j = 0; j < T::TN; ++j) {
                        if constexpr (MFR == 32) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ks & 1][i][q], fb[ks & 1][j][q], acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[ks & 1][i][q], fb[ks & 1][j][q], acc[i][j], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
                // pieces go to the EARLIEST slots: the registers are stored first and re-loaded right after, so that the new loads have the
                // rest of this step (and the first slots of the next) to land
                constexpr int per = (NP + NSLOT - 1) / NSLOT, lo = S * per < NP ? S * per : NP, hi = lo + per < NP ? lo + per : NP;
                static_for<hi - lo>([&](auto kc) {
                    constexpr int P = lo + decltype(kc)::value;
                    constexpr bool is_store = P < T::AROWS + T::BVECS;
                    if constexpr ((is_store && !ST) || (!is_store && !IS)) {}
                    else if constexpr (WHOLE_ISSUE && P == T::AROWS + T::BVECS) issue(ra, rb);
                    else piece(std::integral_constant<int, P>{}, nxt);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        if constexpr (MCAV_DIAG < 3) __syncthreads();
    };
    MCAV_STAMP(1);
    int t = 0;
    using Yes = std::true_type;
    using No = std::false_type;
    if (T_total > 0) {
        if constexpr (REFL) {                    // (the border wavefronts' extra registers serve one tile at a time)
            issue(ra, rb);
            store(ra, rb, B0{});
            if (T_total > 1) issue(ra, rb);
        } else if constexpr (MCAV_DIAG >= 6) {   // timing only: no first loads
#pragma unroll
            for (int j = 0; j < T::AROWS; ++j) ra[j] = f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
            for (int j = 0; j < T::BVECS; ++j) rb[j] = f32x4{1.f, 1.f, 1.f, 1.f};
            store(ra, rb, B0{});
        } else {                                 // tiles 0 and 1 in flight together: one memory round trip before the loop instead of two
            f32x4 ra0[T::AROWS], rb0[T::BVECS];
            issue(ra0, rb0);
            if (T_total > 1) issue(ra, rb);
            store(ra0, rb0, B0{});
        }
    }
    __syncthreads();
    MCAV_STAMP(2);
    for (; t + 3 < T_total; t += 2) {
        step(B0{}, B1{}, Yes{}, Yes{});
        step(B1{}, B0{}, Yes{}, Yes{});
    }
    MCAV_STAMP(3);
    // the last one to three tiles run the same interleaved step, without the stores / loads that have no tile left
    const int rest = T_total - t;
    if (rest == 3) {
        step(B0{}, B1{}, Yes{}, Yes{});
        step(B1{}, B0{}, Yes{}, No{});
        step(B0{}, B1{}, No{}, No{});
    } else if (rest == 2) {
        step(B0{}, B1{}, Yes{}, No{});
        step(B1{}, B0{}, No{}, No{});
    } else if (rest == 1) {
        step(B0{}, B1{}, No{}, No{});
    }
    MCAV_STAMP(4);
    if (p.ksplit > 1) {      // raw partial tile into this split's y-shaped slab; bias / activation / aux factor / addend follow in splitk_finish_kernel
        IgemmParams q = p;
        q.y = p.kslab + (size_t)ks * ((size_t)p.g.B * p.Hd * p.Wd * p.Cd);
        q.bias = nullptr; q.act = MCAV_ACT_NONE; q.dact_aux = nullptr; q.addend = nullptr; q.stats = nullptr; q.stats_x = nullptr;
        igemm_epilogue_lean<T>(q, acc, s_out, s_stat, tid, wm0, wn0, n0, mt);
    } else {
        igemm_epilogue_lean<T>(p, acc, s_out, s_stat, tid, wm0, wn0, n0, mt);
    }
    MCAV_STAMP(5);
}

// ------------------------------------------------------------------------------------------------ weight gradient
// out[kflat, n] = sum_pix A[pix, kflat] * dy[pix, n]: GEMM rows = flattened (tap, c) of the filter, columns = output
// channels, reduction over pixels (split across workgroups; partial tiles go to a slab and are summed in fixed order).
template <class T, int KIND>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradParams p) {
    constexpr int BM = T::BM, BN = T::BN;
    __shared__ __attribute__((aligned(16))) float Xs[2][KP][BM];   // [pixel][kflat]  (A operand, k-major)
    __shared__ __attribute__((aligned(16))) float Ys[2][KP][BN];   // [pixel][n]      (B operand, k-major)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per_split = p.mtiles * p.ntiles;
    // XCD-local splits: every row / column tile of one pixel split reads the same pixels (a shifted view of x per tap, the same dy), so
    // consecutive logical ids share an XCD and the split's pixels are fetched into ONE L2 instead of eight.  On the MFMA-bound fp32 kernels
    // this changes no launch's time by more than noise (measured in round 1 and again in round 2) but it takes the fabric traffic of the
    // weight gradients from 2-10x their algorithmic bytes to about 1x (profiles/r02_pmc_fetch_write_per_kernel.txt).
    const int lid_ = xcd_remap(blockIdx.x, gridDim.x);
    const int split = lid_ / per_split, rem = lid_ - split * per_split;
    const int nt = rem % p.ntiles, mt = rem / p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;      // m0: first kflat row, n0: first output channel
    const GatherSrc& g = p.g;

    // A loads: BM/4 float4 columns x KP pixels per tile; thread owns column group(s) and strides over pixels
    constexpr int ACOLS = BM / 4;                       // float4 columns
    constexpr int APIX = 256 / ACOLS;                   // pixels covered per pass (BM=128: 8, BM=64: 16, BM=256: 4)
    constexpr int APASS = KP / APIX;
    // a wavefront owns ACOLS/4 consecutive 16-byte columns (contiguous bytes per pixel: coalesced loads, conflict-free
    // ds_write_b128) of 64/(ACOLS/4) pixels: its channel range is one 4*ACOLS/4-wide group, i.e. it lies in x1 or in x2
    constexpr int CPW = ACOLS / 4;
    const int acol = wave * CPW + lane % CPW, apix = lane / CPW;
    const int kflat = m0 + acol * 4;
    int aky = -1, akx = 0, ac = 0;
    if (kflat < p.Ktot) {
        const int tap = kflat / p.Kp;
        ac = kflat - tap * p.Kp;
        aky = tap / p.kw; akx = tap - aky * p.kw;
        if (tap >= p.taps) aky = -1;
    }
    // B loads: BN/4 float4 columns x KP pixels
    constexpr int BCOLS = BN / 4;
    constexpr int BPIX = 256 / BCOLS > KP ? KP : 256 / BCOLS;
    constexpr int BPASS = KP / BPIX;
    const int bcol = tid % BCOLS, bpix = tid / BCOLS;

    const int pix_begin = split * p.pix_per_split;
    const int pix_end = min(p.Mpix, pix_begin + p.pix_per_split);
    const int T_total = pix_end > pix_begin ? (pix_end - pix_begin + KP - 1) / KP : 0;

    // destination pixel of this thread's first A row of the current tile, advanced incrementally (no per-tile division)
    int pn, py, px;
    {
        const int m = pix_begin + apix;
        pn = m / (p.Hd * p.Wd);
        const int r = m - pn * (p.Hd * p.Wd);
        py = r / p.Wd; px = r - py * p.Wd;
    }
    auto step_pix = [&](int& n, int& y, int& x, int by) {
        x += by;
        if constexpr (KIND == K_FAST) {       // Wd >= APIX is a dispatch condition of the fast kind: at most one wrap
            const bool wx = x >= p.Wd;
            x = wx ? x - p.Wd : x;
            y += wx ? 1 : 0;
            const bool wy = y >= p.Hd;
            y = wy ? 0 : y;
            n += wy ? 1 : 0;
        } else {
            while (x >= p.Wd) { x -= p.Wd; if (++y == p.Hd) { y = 0; ++n; } }
        }
    };
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    const bool do_bias = p.want_bias && mt == 0;
    // K_FAST: branch-free raw buffer loads (out-of-range offset reads as zero), x1/x2 chosen per wavefront
    const bool use2 = KIND == K_FAST && __builtin_amdgcn_readfirstlane((int)(g.C2 > 0 && ac >= g.C1)) != 0;
    const unsigned bytes1 = (unsigned)((size_t)g.B * (g.up1 ? (g.Hs >> 1) * (g.Ws >> 1) : g.Hs * g.Ws) * g.C1 * 4);
    const unsigned bytes2 = (unsigned)((size_t)g.B * g.Hs * g.Ws * g.C2 * 4);
    const __amdgpu_buffer_rsrc_t rsx = use2 ? make_rsrc(g.x2, bytes2) : make_rsrc(g.x1, bytes1);
    const __amdgpu_buffer_rsrc_t rsy = make_rsrc(p.dy, (unsigned)((size_t)p.Mpix * p.Cdy * 4));
    const int acc_ = use2 ? ac - g.C1 : ac;                                    // channel inside the chosen source
    const int Csrc = use2 ? g.C2 : g.C1;
    const bool a_ok = aky >= 0 && acc_ < Csrc;
    auto load_tile = [&](int t, f32x4 (&ra)[APASS], f32x4 (&rb)[BPASS]) {      // must be called with t = 0, 1, 2, ... in order
        const int pb = pix_begin + t * KP;
        int n = pn, dy = py, dx = px;
#pragma unroll
        for (int j = 0; j < APASS; ++j) {
            const int m = pb + apix + j * APIX;
            if constexpr (KIND == K_FAST) {
                int sy = dy * g.stride + aky + g.offset, sx = dx * g.stride + akx + g.offset;      // sign = +1 (forward gather)
                if (g.pad_mode == MCAV_PAD_REFLECT) {
                    sy = reflect_idx(sy, g.Hs);
                    sx = reflect_idx(sx, g.Ws);
                }
                const bool ok = a_ok && m < pix_end && (unsigned)sy < (unsigned)g.Hs && (unsigned)sx < (unsigned)g.Ws;
                const int pix = (!use2 && g.up1) ? ((n * (g.Hs >> 1) + (sy >> 1)) * (g.Ws >> 1) + (sx >> 1)) : ((n * g.Hs + sy) * g.Ws + sx);
                ra[j] = buf_load4(rsx, ok ? (unsigned)(pix * Csrc + acc_) * 4u : OOB);
            } else {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (m < pix_end && aky >= 0) v = gather4(g, n, dy, dx, aky, akx, ac);
                ra[j] = v;
            }
            step_pix(n, dy, dx, APIX);
        }
        pn = n; py = dy; px = dx;       // APASS * APIX == KP: now at this thread's first row of tile t + 1
#pragma unroll
        for (int j = 0; j < BPASS; ++j) {
            const int pl = bpix + j * BPIX;
            const int m = pb + pl;
            if constexpr (KIND == K_FAST) {
                const int c = n0 + bcol * 4;
                const bool ok = pl < KP && m < pix_end && c + 4 <= p.CoutLoad;
                rb[j] = buf_load4(rsy, ok ? (unsigned)(m * p.Cdy + p.dy_choff + c) * 4u : OOB);
                continue;
            }
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pl < KP && m < pix_end) {
                const int c = n0 + bcol * 4;
                const float* q = p.dy + (size_t)m * p.Cdy + p.dy_choff + c;
                if ((p.Cdy & 3) == 0 && (p.dy_choff & 3) == 0 && c + 4 <= p.Cout) {
                    v = *reinterpret_cast<const f32x4*>(q);
                } else {
                    if (c + 0 < p.Cout) v.x = q[0];
                    if (c + 1 < p.Cout) v.y = q[1];
                    if (c + 2 < p.Cout) v.z = q[2];
                    if (c + 3 < p.Cout) v.w = q[3];
                }
            }
            rb[j] = v;
        }
    };
    auto store_tile = [&](int buf, const f32x4 (&ra)[APASS], const f32x4 (&rb)[BPASS]) {
        if (do_bias) {      // column sums of dy for the bias gradient, taken when the data has landed anyway
#pragma unroll
            for (int j = 0; j < BPASS; ++j) bsum += rb[j];
        }
#pragma unroll
        for (int j = 0; j < APASS; ++j) *reinterpret_cast<f32x4*>(&Xs[buf][apix + j * APIX][acol * 4]) = ra[j];
#pragma unroll
        for (int j = 0; j < BPASS; ++j) {
            const int pl = bpix + j * BPIX;
            if (pl < KP) *reinterpret_cast<f32x4*>(&Ys[buf][pl][bcol * 4]) = rb[j];
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

    auto compute = [&](int buf) {
        if constexpr (T::MF == 32) {
            const int fr = lane & 31, fk = lane >> 5;
            // operands of 4 k-steps form a batch; batch u+1 is read from LDS before the MFMAs of batch u are issued and the
            // scheduler is fenced so that it cannot fold the reads back behind the MFMAs (one exposed LDS latency per tile)
            constexpr int NB = KP / 2 / 4;
            float a[2][4][T::TM], b[2][4][T::TN];
            auto rd = [&](int s, int k0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int i = 0; i < T::TM; ++i) a[s][u][i] = Xs[buf][(k0 + u) * 2 + fk][wm0 + i * 32 + fr];
#pragma unroll
                    for (int j = 0; j < T::TN; ++j) b[s][u][j] = Ys[buf][(k0 + u) * 2 + fk][wn0 + j * 32 + fr];
                }
            };
            rd(0, 0);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) {
                if (bt + 1 < NB) rd((bt + 1) & 1, (bt + 1) * 4);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int i = 0; i < T::TM; ++i)
#pragma unroll
                        for (int j = 0; j < T::TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[bt & 1][u][i], b[bt & 1][u][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            const int fr = lane & 15, fk = lane >> 4;
            constexpr int NB = KP / 4 / 4;
            float a[2][4][T::TM], b[2][4][T::TN];
            auto rd = [&](int s, int k0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int i = 0; i < T::TM; ++i) a[s][u][i] = Xs[buf][(k0 + u) * 4 + fk][wm0 + i * 16 + fr];
#pragma unroll
                    for (int j = 0; j < T::TN; ++j) b[s][u][j] = Ys[buf][(k0 + u) * 4 + fk][wn0 + j * 16 + fr];
                }
            };
            rd(0, 0);
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) {
                if (bt + 1 < NB) rd((bt + 1) & 1, (bt + 1) * 4);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int i = 0; i < T::TM; ++i)
#pragma unroll
                        for (int j = 0; j < T::TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[bt & 1][u][i], b[bt & 1][u][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    f32x4 ra0[APASS], rb0[BPASS], ra1[APASS], rb1[BPASS];
    if (T_total > 0) load_tile(0, ra0, rb0);
    if (T_total > 1) load_tile(1, ra1, rb1);
    if (T_total > 0) store_tile(0, ra0, rb0);
    __syncthreads();
    for (int t = 0; t < T_total; t += 2) {
        if (t + 2 < T_total) load_tile(t + 2, ra0, rb0);
        compute(0);
        if (t + 1 < T_total) store_tile(1, ra1, rb1);
        __syncthreads();
        if (t + 1 >= T_total) break;
        if (t + 3 < T_total) load_tile(t + 3, ra1, rb1);
        compute(1);
        if (t + 2 < T_total) store_tile(0, ra0, rb0);
        __syncthreads();
    }

    constexpr int MF = T::MF;
    const int ccol = lane & (MF - 1);
    float* slab = p.slab + (size_t)split * (p.Ktot + 1) * p.slabN;
    if (do_bias) {      // column sums of dy over this split's pixels: lanes with the same column group, then the pixel lanes
        if (bpix < BPIX) *reinterpret_cast<f32x4*>(&Ys[0][bpix][bcol * 4]) = bsum;
        __syncthreads();
        if (tid < BN) {
            float t = 0.f;
            for (int k = 0; k < BPIX; ++k) t += Ys[0][k][tid];
            if (n0 + tid < p.slabN) slab[(size_t)p.Ktot * p.slabN + n0 + tid] = t;
        }
    }
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j) {
            const int n = n0 + wn0 + j * MF + ccol;
#pragma unroll
            for (int r = 0; r < T::ACC; ++r) {
                const int row = m0 + wm0 + i * MF + (MF == 32 ? (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) : 4 * (lane >> 4) + r);
                if (row < p.Ktot && n < p.slabN) slab[(size_t)row * p.slabN + n] = acc[i][j][r];
            }
        }
}

// Table-driven weight-gradient kernel: same GEMM, tiling, LDS panels and epilogue as wgrad_kernel, with the addressing taken
// out of the K loop (the GEMM's reduction runs over PIXELS here, so every K-tile needs fresh source addresses):
//   * the source byte offset of every (pixel, filter tap touched by this row tile, source tensor) lives in an LDS table --
//     padding, reflection, stride, upsampling are resolved there, invalid taps hold an out-of-range offset.  A split whose
//     pixels fit has the whole table built once; longer splits use it as two half-buffers of CHT tiles each, the next
//     chunk being computed (one pixel per thread) while the current one is consumed, fenced by the per-tile barriers;
//   * per A load the loop does one ds_read_b32 (prefetched a tile ahead) and one v_add (the lane's channel offset);
//   * dy rows advance linearly: their loads use the instruction's SCALAR offset, and the buffer resource ends at this
//     split's last pixel, so the ragged last tile needs no masking.

// UPM: the weight gradient of the upsampled half of conv(cat(up2(a), skip)) in merged-tap form (see mcav_wgrad_desc.upm): the K
// dimension runs over LOW-resolution pixels, a row tile belongs to one output parity class (py, px) and one or more of its 4 merged
// taps; the A operand is the clamped low-resolution source of that tap, the B operand the class's dy pixel (2 y2 + py, 2 x2 + px) --
// both through the offset table (dy is not linear in the low-resolution pixel index).  The 16 x Kp result rows are un-merged into
// the 9 filter taps by the reduce kernel.
template <class T, bool UPM>
// (register budget: four 64x64 workgroups per CU -- 40 KB of LDS each with the 2048-entry table --, three of the 128x32 ones: 96->32 at
//  96x320 0.518 -> 0.410 ms, 64->64 at 48x160 0.126 -> 0.119 ms)
__global__ __launch_bounds__(256, (T::BM == 64 && T::BN == 64) ? 4 : ((T::BM == 128 && T::BN == 32) ? 3 : 1)) void wgrad_tab_kernel(WgradParams p) {
    constexpr int BM = T::BM, BN = T::BN;
    __shared__ __attribute__((aligned(16))) float Xs[2][KP][BM];
    __shared__ __attribute__((aligned(16))) float Ys[2][KP][BN];
    __shared__ unsigned s_tab[WG_TABCAP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per_split = p.mtiles * p.ntiles;
    // XCD-local splits: every row / column tile of one pixel split reads the same pixels (a shifted view of x per tap, the same dy), so
    // consecutive logical ids share an XCD and the split's pixels are fetched into ONE L2 instead of eight.  On the MFMA-bound fp32 kernels
    // this changes no launch's time by more than noise (measured in round 1 and again in round 2) but it takes the fabric traffic of the
    // weight gradients from 2-10x their algorithmic bytes to about 1x (profiles/r02_pmc_fetch_write_per_kernel.txt).
    const int lid_ = xcd_remap(blockIdx.x, gridDim.x);
    const int split = lid_ / per_split, rem = lid_ - split * per_split;
    const int nt = rem % p.ntiles, mt = rem / p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const GatherSrc& g = p.g;

    constexpr int ACOLS = BM / 4, APIX = 256 / ACOLS, APASS = KP / APIX, CPW = ACOLS / 4;
    const int acol = wave * CPW + lane % CPW, apix = lane / CPW;
    const int kflat = m0 + acol * 4;
    // Xs rows are BM floats: with BM = 64 two consecutive pixels start on the same bank, and both an 8-lane ds_write_b128 group (two
    // pixels x four 16-byte columns) and a 32-lane read group of the 16-wide MFMA (two pixels x 16 rows) would be 2-way conflicts.
    // Odd pixels therefore store their row with the two 16-float halves of every 32 swapped (column ^ 16); pixel parity is a per-lane
    // constant on both sides (APIX and KSTEP are even), so the swizzle costs no instruction in the loop.
    constexpr bool XSWZ = BM == 64;
    const int xswz_w = XSWZ ? (apix & 1) << 4 : 0;
    const int xswz_r = XSWZ ? ((lane / T::MF) & 1) << 4 : 0;
    constexpr int BCOLS = BN / 4;
    constexpr int BPIX = 256 / BCOLS > KP ? KP : 256 / BCOLS;
    constexpr int BPASS = KP / BPIX;
    const int bcol = tid % BCOLS, bpix = tid / BCOLS;

    const int pix_begin = split * p.pix_per_split;
    const int pix_end = min(p.Mpix, pix_begin + p.pix_per_split);
    const int T_total = pix_end > pix_begin ? (pix_end - pix_begin + KP - 1) / KP : 0;
    const int tap_lo = m0 / p.Kp;
    const int tap_hi = min(p.taps - 1, (m0 + BM - 1) / p.Kp);
    const int NT = tap_hi - tap_lo + 1;
    const bool two = !UPM && g.C2 > 0;                    // second table: offsets into x2 (x1 may be the upsampled source)
    // chunking: CHT = 2^cht tiles per half-buffer; a split that fits is one chunk (cht large, second half never used)
    const int cht = p.tab_cht_log2;
    const int npc = min(T_total, 1 << min(cht, 20)) * KP;   // pixels per chunk
    const int half = (UPM ? NT + 1 : NT * (two ? 2 : 1)) * npc;      // entries per half-buffer (UPM: + the dy offsets)

    // chunk c -> half-buffer c & 1: [source][tap - tap_lo][pixel within the chunk]
    auto build_chunk = [&](int c) {
        unsigned* tb = s_tab + (c & 1) * half;
        for (int pl = tid; pl < npc; pl += 256) {
            const int m = pix_begin + c * npc + pl;
            const bool live = m < pix_end;
            const int n = m / (p.Hd * p.Wd);
            const int r = m - n * (p.Hd * p.Wd);
            const int dy = r / p.Wd, dx = r - dy * p.Wd;
            if constexpr (UPM) {           // (dy, dx) = low-resolution pixel (y2, x2); this row tile's class from its first merged tap
                const int cls = tap_lo >> 2, py = cls >> 1, px = cls & 1;
                for (int tl = 0; tl < NT; ++tl) {
                    const int mtap = (tap_lo + tl) & 3;
                    const int sy = min(max(dy - 1 + py + (mtap >> 1), 0), p.Hd - 1), sx = min(max(dx - 1 + px + (mtap & 1), 0), p.Wd - 1);
                    tb[tl * npc + pl] = live ? (unsigned)(((n * p.Hd + sy) * p.Wd + sx) * g.C1) * 4u : OOB;
                }
                tb[NT * npc + pl] = live ? (unsigned)(((n * p.Hf + 2 * dy + py) * p.Wf + 2 * dx + px) * p.Cdy) * 4u : OOB;
                continue;
            }
            for (int tl = 0; tl < NT; ++tl) {
                const int tap = tap_lo + tl;
                const int ky = tap / p.kw, kx = tap - ky * p.kw;
                int sy = dy * g.stride + ky + g.offset, sx = dx * g.stride + kx + g.offset;      // sign = +1 (forward gather)
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
    const int nchunks = npc > 0 ? (T_total * KP + npc - 1) / npc : 0;
    if (nchunks > 0) build_chunk(0);
    if (nchunks > 1) build_chunk(1);

    // ---- this thread's A column: filter tap and channel; a wavefront's columns lie in one source
    int tl_own = 0, ac = 0;
    bool a_ok = false;
    if (kflat < p.Ktot) {
        const int tap = kflat / p.Kp;
        ac = kflat - tap * p.Kp;
        tl_own = tap - tap_lo;
        a_ok = tap < p.taps;
    }
    const bool use2 = __builtin_amdgcn_readfirstlane((int)(two && ac >= g.C1)) != 0;
    const unsigned bytes1 = UPM ? (unsigned)((size_t)g.B * p.Hd * p.Wd * g.C1 * 4)
                                : (unsigned)((size_t)g.B * (g.up1 ? (g.Hs >> 1) * (g.Ws >> 1) : g.Hs * g.Ws) * g.C1 * 4);
    const unsigned bytes2 = (unsigned)((size_t)g.B * g.Hs * g.Ws * g.C2 * 4);
    const __amdgpu_buffer_rsrc_t rsx = use2 ? make_rsrc(g.x2, bytes2) : make_rsrc(g.x1, bytes1);
    const __amdgpu_buffer_rsrc_t rsy = UPM ? make_rsrc(p.dy, (unsigned)((size_t)g.B * p.Hf * p.Wf * p.Cdy * 4))
                                           : make_rsrc(p.dy, (unsigned)((size_t)pix_end * p.Cdy * 4));      // ends at this split's last pixel
    const int acc_ = use2 ? ac - g.C1 : ac;
    a_ok = a_ok && acc_ < (use2 ? g.C2 : g.C1);
    // Lanes on K-padding channels or on rows past Ktot add the out-of-range bit: they read zero (or, under an out-of-image tap whose
    // table entry carries the same bit, whatever sits at the wrapped offset) into GEMM rows that nothing consumes.
    const unsigned chan = a_ok ? (unsigned)acc_ * 4u : OOB;
    const int trow = ((use2 ? NT : 0) + (a_ok ? tl_own : 0)) * npc + apix;      // this lane's row inside a half-buffer
    unsigned boff[BPASS];
#pragma unroll
    for (int j = 0; j < BPASS; ++j) {
        const int pl = bpix + j * BPIX, c = n0 + bcol * 4;
        const bool ok = pl < KP && c + 4 <= p.CoutLoad;
        boff[j] = ok ? (UPM ? (unsigned)((p.dy_choff + c) * 4) : (unsigned)(((pix_begin + pl) * p.Cdy + p.dy_choff + c) * 4)) : OOB;
    }
    const int brow = NT * npc + (bpix < KP ? bpix : 0);      // UPM: this lane's position in the dy-offset row of a half-buffer
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    const bool do_bias = p.want_bias && mt == 0;
    __syncthreads();

    unsigned toff[APASS];                                 // table values of the next tile to issue
    unsigned tboff[UPM ? BPASS : 1];                      // UPM: dy pixel offsets of the next tile
    int u = 0;                                            // the issue pointer
    auto fetch = [&]() {
        const int uc = u >> cht, ul = u - (uc << cht);    // chunk of tile u and its position inside it
        const unsigned* tr = s_tab + (uc & 1) * half + trow + ul * KP;
#pragma unroll
        for (int j = 0; j < APASS; ++j) toff[j] = tr[j * APIX];
        if constexpr (UPM) {
            const unsigned* tq = s_tab + (uc & 1) * half + brow + ul * KP;
#pragma unroll
            for (int j = 0; j < BPASS; ++j) tboff[j] = tq[j * BPIX < KP ? j * BPIX : 0];
        }
    };
    auto issue = [&](f32x4 (&ra)[APASS], f32x4 (&rb)[BPASS]) {
#pragma unroll
        for (int j = 0; j < APASS; ++j) ra[j] = buf_load4(rsx, toff[j] + chan);
        if constexpr (UPM) {
#pragma unroll
            for (int j = 0; j < BPASS; ++j) rb[j] = buf_load4(rsy, ((tboff[j] | boff[j]) & OOB) ? OOB : tboff[j] + boff[j]);
        } else {
            const int sb = u * KP * p.Cdy * 4;
#pragma unroll
            for (int j = 0; j < BPASS; ++j) rb[j] = buf_load4s(rsy, boff[j], sb);
        }
        ++u;
        if (u < T_total) fetch();
    };
    auto store = [&](const f32x4 (&ra)[APASS], const f32x4 (&rb)[BPASS], auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        if (do_bias) {
#pragma unroll
            for (int j = 0; j < BPASS; ++j) bsum += rb[j];
        }
#pragma unroll
        for (int j = 0; j < APASS; ++j) *reinterpret_cast<f32x4*>(&Xs[buf][apix + j * APIX][(acol * 4) ^ xswz_w]) = ra[j];
#pragma unroll
        for (int j = 0; j < BPASS; ++j) {
            const int pl = bpix + j * BPIX;
            if (256 / BCOLS <= KP || pl < KP) *reinterpret_cast<f32x4*>(&Ys[buf][pl][bcol * 4]) = rb[j];      // narrow N: idle threads past KP rows
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
    constexpr int KSTEP = MFR == 32 ? 2 : 4;              // pixels consumed per MFMA
    const int fr = lane & (MFR - 1), fk = lane / MFR;
    constexpr int NB = KP / KSTEP / 4;                    // batches of 4 MFMA k-steps; batch b + 1 is read before batch b is issued
    float a[2][4][T::TM], b[2][4][T::TN];
    auto rd = [&](auto bufc, int s, int k0) {
        constexpr int buf = decltype(bufc)::value;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int i = 0; i < T::TM; ++i) a[s][q][i] = Xs[buf][(k0 + q) * KSTEP + fk][(wm0 + i * MFR + fr) ^ xswz_r];
#pragma unroll
            for (int j = 0; j < T::TN; ++j) b[s][q][j] = Ys[buf][(k0 + q) * KSTEP + fk][wn0 + j * MFR + fr];
        }
    };
    auto mma = [&](int s, int q) {
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int j = 0; j < T::TN; ++j) {
                if constexpr (MFR == 32) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][q][i], b[s][q][j], acc[i][j], 0, 0, 0);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][q][i], b[s][q][j], acc[i][j], 0, 0, 0);
            }
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    // Two register stages: while tile t is multiplied, tile t + 1 moves from registers to the other LDS buffer and the loads of tiles
    // t + 2 and t + 3 are in flight -- a load has two MFMA phases (of this wavefront alone) to land, not one.  Stage s holds the tiles of
    // parity s.
    f32x4 ra0[APASS], rb0[BPASS], ra1[APASS], rb1[BPASS];
    // Steady-state step.  A wavefront issues in order and its MFMAs form one dependent chain (64 cycles apart), so everything else of
    // the step -- the LDS stores of tile t + 1, the global loads of tile t + 3, the table fetch -- is cut into pieces and one piece is
    // placed behind each MFMA, where it issues while the MFMA pipe works, instead of in a block of its own in front of the MFMAs.
    constexpr int NSLOT = NB * 4;                         // MFMA k-steps of a tile = slots for pieces
    constexpr int NPIECE = 2 * APASS + 2 * BPASS + 1;     // A stores, B stores, A loads, B loads, advance
    auto piece = [&](auto pc, auto nxt, f32x4 (&ra)[APASS], f32x4 (&rb)[BPASS]) {
        constexpr int P = decltype(pc)::value, buf = decltype(nxt)::value;
        if constexpr (P < APASS) {
            *reinterpret_cast<f32x4*>(&Xs[buf][apix + P * APIX][(acol * 4) ^ xswz_w]) = ra[P];
        } else if constexpr (P < APASS + BPASS) {
            constexpr int j = P - APASS;
            if (do_bias) bsum += rb[j];
            const int pl = bpix + j * BPIX;
            if (256 / BCOLS <= KP || pl < KP) *reinterpret_cast<f32x4*>(&Ys[buf][pl][bcol * 4]) = rb[j];
        } else if constexpr (P < 2 * APASS + BPASS) {
            constexpr int j = P - APASS - BPASS;
            ra[j] = buf_load4(rsx, toff[j] + chan);
        } else if constexpr (P < 2 * APASS + 2 * BPASS) {
            constexpr int j = P - 2 * APASS - BPASS;
            if constexpr (UPM) rb[j] = buf_load4(rsy, ((tboff[j] | boff[j]) & OOB) ? OOB : tboff[j] + boff[j]);
            else rb[j] = buf_load4s(rsy, boff[j], u * KP * p.Cdy * 4);
        } else {
            ++u;
            if (u < T_total) fetch();
        }
    };
    // fl = ST | 2 IS: the step stores tile t + 1 / issues tile t + 3 (the last steps of a split have no tile left to store / issue)
    auto pieces_of_slot = [&](auto sc, auto nxt, auto fl, f32x4 (&ra)[APASS], f32x4 (&rb)[BPASS]) {
        constexpr int S = decltype(sc)::value, FL = decltype(fl)::value;
        constexpr int lo = S * NPIECE / NSLOT, hi = (S + 1) * NPIECE / NSLOT;
        static_for<hi - lo>([&](auto kc) {
            constexpr int P = lo + decltype(kc)::value;
            constexpr bool is_store = P < APASS + BPASS;
            if constexpr (is_store ? (FL & 1) != 0 : (FL & 2) != 0) piece(std::integral_constant<int, P>{}, nxt, ra, rb);
        });
        static_assert(hi - lo <= 3, "at most three pieces per MFMA slot");
    };
    auto slot = [&](auto sc, auto nxt, auto fl, f32x4 (&ra)[APASS], f32x4 (&rb)[BPASS]) {
        constexpr int S = decltype(sc)::value;
        mma((S >> 2) & 1, S & 3);
        __builtin_amdgcn_sched_barrier(0);
        pieces_of_slot(sc, nxt, fl, ra, rb);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto step = [&](auto cur, auto nxt, auto fl, f32x4 (&ra)[APASS], f32x4 (&rb)[BPASS]) {      // (ra, rb): the stage of tiles t + 1 and t + 3
        rd(cur, 0, 0);
        static_assert(NB == 2 || NB == 4, "two or four operand batches per tile");
        if constexpr (NB > 1) rd(cur, 1, 4);
        __builtin_amdgcn_sched_barrier(0);
        slot(std::integral_constant<int, 0>{}, nxt, fl, ra, rb); slot(std::integral_constant<int, 1>{}, nxt, fl, ra, rb);
        slot(std::integral_constant<int, 2>{}, nxt, fl, ra, rb); slot(std::integral_constant<int, 3>{}, nxt, fl, ra, rb);
        if constexpr (NB > 2) rd(cur, 0, 8);
        __builtin_amdgcn_sched_barrier(0);
        slot(std::integral_constant<int, 4>{}, nxt, fl, ra, rb); slot(std::integral_constant<int, 5>{}, nxt, fl, ra, rb);
        slot(std::integral_constant<int, 6>{}, nxt, fl, ra, rb); slot(std::integral_constant<int, 7>{}, nxt, fl, ra, rb);
        if constexpr (NB > 2) {
            rd(cur, 1, 12);
            __builtin_amdgcn_sched_barrier(0);
            slot(std::integral_constant<int, 8>{}, nxt, fl, ra, rb); slot(std::integral_constant<int, 9>{}, nxt, fl, ra, rb);
            slot(std::integral_constant<int, 10>{}, nxt, fl, ra, rb); slot(std::integral_constant<int, 11>{}, nxt, fl, ra, rb);
            slot(std::integral_constant<int, 12>{}, nxt, fl, ra, rb); slot(std::integral_constant<int, 13>{}, nxt, fl, ra, rb);
            slot(std::integral_constant<int, 14>{}, nxt, fl, ra, rb); slot(std::integral_constant<int, 15>{}, nxt, fl, ra, rb);
        }
        __syncthreads();
    };
    int t = 0;
    if (T_total > 0) {
        fetch();
        issue(ra0, rb0);
        store(ra0, rb0, B0{});
        if (T_total > 1) issue(ra1, rb1);
        if (T_total > 2) issue(ra0, rb0);
    }
    __syncthreads();
    // Chunk c + 1 is computed one step before the compute pointer enters chunk c: its half-buffer held chunk c - 1, whose last entry was
    // fetched five steps ago, and its first entry (tile (c + 1) CHT, fetched while tile (c + 1) CHT - 4 is computed) is fetched no earlier
    // than the next step (CHT >= 4) -- barriers on both sides.
    auto maybe_build = [&](int tt) {
        if (((tt + 1) & ((1 << cht) - 1)) == 0 && ((tt + 1) >> cht) + 1 < nchunks) build_chunk(((tt + 1) >> cht) + 1);
    };
    using F3 = std::integral_constant<int, 3>;
    using F1 = std::integral_constant<int, 1>;
    using F0 = std::integral_constant<int, 0>;
    for (; t + 4 < T_total; t += 2) {
        maybe_build(t);
        step(B0{}, B1{}, F3{}, ra1, rb1);
        maybe_build(t + 1);
        step(B1{}, B0{}, F3{}, ra0, rb0);
    }
    // the last one to four tiles: the same interleaved step without the pieces that have no tile left (t is even)
    const int rest = T_total - t;
    if (rest == 4) {
        maybe_build(t);     step(B0{}, B1{}, F3{}, ra1, rb1);
        maybe_build(t + 1); step(B1{}, B0{}, F1{}, ra0, rb0);
        maybe_build(t + 2); step(B0{}, B1{}, F1{}, ra1, rb1);
        maybe_build(t + 3); step(B1{}, B0{}, F0{}, ra0, rb0);
    } else if (rest == 3) {
        maybe_build(t);     step(B0{}, B1{}, F1{}, ra1, rb1);
        maybe_build(t + 1); step(B1{}, B0{}, F1{}, ra0, rb0);
        maybe_build(t + 2); step(B0{}, B1{}, F0{}, ra1, rb1);
    } else if (rest == 2) {
        maybe_build(t);     step(B0{}, B1{}, F1{}, ra1, rb1);
        maybe_build(t + 1); step(B1{}, B0{}, F0{}, ra0, rb0);
    } else if (rest == 1) {
        maybe_build(t);     step(B0{}, B1{}, F0{}, ra1, rb1);
    }

    constexpr int MF = T::MF;
    const int ccol = lane & (MF - 1);
    float* slab = p.slab + (size_t)split * (p.Ktot + 1) * p.slabN;
    if (do_bias) {
        if (bpix < BPIX) *reinterpret_cast<f32x4*>(&Ys[0][bpix][bcol * 4]) = bsum;
        __syncthreads();
        if (tid < BN) {
            float tsum = 0.f;
            for (int k = 0; k < BPIX; ++k) tsum += Ys[0][k][tid];
            if (n0 + tid < p.slabN) slab[(size_t)p.Ktot * p.slabN + n0 + tid] = tsum;
        }
    }
    // partial tile -> slab: raw buffer stores (rows at or past Ktot fall outside the resource and are dropped): one vector add per value
    const __amdgpu_buffer_rsrc_t rslab = make_rsrc(slab, (unsigned)((size_t)p.Ktot * p.slabN * 4));
    const int rowb = p.slabN * 4;
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j) {
            const int n = n0 + wn0 + j * MF + ccol;
            if (n < p.slabN) {
                const unsigned base = (unsigned)(m0 + wm0 + i * MF + (MF == 32 ? 4 * (lane >> 5) : 4 * (lane >> 4))) * (unsigned)rowb + (unsigned)n * 4u;
#pragma unroll
                for (int r = 0; r < T::ACC; ++r) {
                    const int dr = MF == 32 ? (r & 3) + 8 * (r >> 2) : r;
                    buf_store1(rslab, base + (unsigned)(dr * rowb), acc[i][j][r]);      // (the range check covers the vector offset only)
                }
            }
        }
}

// First reduction level when there are many pixel splits: dst[g][e] = sum of the splits of group g (fixed order).
__device__ __forceinline__ void wgrad_presum_body(const float* slab, int splits, size_t elems, int per_group, float* dst, unsigned bx, unsigned by) {
    // elems is a multiple of 16 (slabN is): 16-byte accesses, four independent partial sums keep several loads in flight
    const size_t e4 = (size_t)bx * 256 + threadIdx.x;
    if (e4 * 4 >= elems) return;
    const int k0 = by * per_group, k1 = min(splits, k0 + per_group);
    const f32x4* src = reinterpret_cast<const f32x4*>(slab) + e4;
    const size_t stride4 = elems / 4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int k = k0;
    for (; k + 3 < k1; k += 4) {
        s0 += src[(size_t)k * stride4];
        s1 += src[(size_t)(k + 1) * stride4];
        s2 += src[(size_t)(k + 2) * stride4];
        s3 += src[(size_t)(k + 3) * stride4];
    }
    for (; k < k1; ++k) s0 += src[(size_t)k * stride4];
    reinterpret_cast<f32x4*>(dst)[(size_t)by * stride4 + e4] = (s0 + s1) + (s2 + s3);
}

__global__ __launch_bounds__(256) void wgrad_presum_kernel(const float* slab, int splits, size_t elems, int per_group, float* dst) {
    wgrad_presum_body(slab, splits, elems, per_group, dst, blockIdx.x, blockIdx.y);
}

// slab [splits][Ktot + 1][slabN] -> OIHW gradient (+ bias gradient), fixed summation order, optional accumulate.
// One block owns 32 output channels x CI_T input channels x all taps: slab reads are coalesced along the output channel,
// the tile is transposed through LDS, and each output channel's run of CI_T * taps floats is written contiguously.
// upm: the slab holds 16 x Kp rows (class, merged tap, ci); filter tap (ky, kx) = the sum over the four classes of the merged tap it belongs to.
// Cin_total / ci_off: the OIHW tensor written has Cin_total input channels and this launch owns [ci_off, ci_off + Cin) of them.
__device__ __forceinline__ void wgrad_reduce_body(float* lds, const float* slab, int splits, int Ktot, int slabN, int Kp, int taps, int Cout, int Cin,
                                                  int CI_T, float* dw, float* dbias, int accumulate, int upm, int Cin_total, int ci_off, int bx, int by) {
    const int co0 = bx * 32, ci0 = by * CI_T;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int run = CI_T * taps, stride = run + 1;
    const size_t split_stride = (size_t)(Ktot + 1) * slabN;
    const bool col_ok = co0 + tx < slabN;
    if (upm) {
        for (int idx = ty; idx < run; idx += 8) {
            const int tap = idx / CI_T, r = idx - tap * CI_T, ci = ci0 + r;
            float sum = 0.f;
            if (ci < Cin && col_ok) {
                const int ky = tap / 3, kx = tap - ky * 3;
                for (int cls = 0; cls < 4; ++cls) {
                    const int py = cls >> 1, px = cls & 1;
                    const int ta = py == 0 ? (ky == 0 ? 0 : 1) : (ky == 2 ? 1 : 0), tb = px == 0 ? (kx == 0 ? 0 : 1) : (kx == 2 ? 1 : 0);
                    const float* src = slab + (size_t)((cls * 4 + ta * 2 + tb) * Kp + ci) * slabN + co0 + tx;
                    for (int k = 0; k < splits; ++k) sum += src[(size_t)k * split_stride];
                }
            }
            lds[tx * stride + r * taps + tap] = sum;
        }
    } else {
        // four filter elements per pass, their split loops interleaved: up to 16 independent loads in flight per lane (one element at a
        // time, the kernel was bound by the round trip of each dependent add)
        for (int idx0 = ty; idx0 < run; idx0 += 32) {
            const float* src[4];
            float sum[4];
            int at[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = idx0 + 8 * u, tap = idx / CI_T, r = idx - tap * CI_T;
                const bool ok = idx < run && ci0 + r < Cin && col_ok;
                src[u] = ok ? slab + (size_t)(tap * Kp + ci0 + r) * slabN + co0 + tx : nullptr;
                at[u] = idx < run ? tx * stride + r * taps + tap : -1;
                sum[u] = 0.f;
            }
            int k = 0;
            for (; k + 3 < splits; k += 4) {
                float v[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[u][j] = src[u] ? src[u][(size_t)(k + j) * split_stride] : 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) sum[u] += v[u][j];
            }
            for (; k < splits; ++k) {
#pragma unroll
                for (int u = 0; u < 4; ++u) sum[u] += src[u] ? src[u][(size_t)k * split_stride] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (at[u] >= 0) lds[at[u]] = sum[u];
        }
    }
    if (dbias && by == 0 && ty == 0 && co0 + tx < Cout) {
        const float* src = slab + (size_t)Ktot * slabN + co0 + tx;
        float sum = 0.f;
        for (int k = 0; k < splits; ++k) sum += src[(size_t)k * split_stride];
        dbias[co0 + tx] = accumulate ? dbias[co0 + tx] + sum : sum;
    }
    __syncthreads();
    const int valid_run = (Cin - ci0 < CI_T ? Cin - ci0 : CI_T) * taps;
    for (int e = threadIdx.x; e < 32 * run; e += 256) {
        const int col = e / run, q = e - col * run;
        if (co0 + col < Cout && q < valid_run) {
            const size_t o = ((size_t)(co0 + col) * Cin_total + ci_off + ci0) * taps + q;
            const float v = lds[col * stride + q];
            dw[o] = accumulate ? dw[o] + v : v;
        }
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* slab, int splits, int Ktot, int slabN, int Kp, int taps, int Cout, int Cin,
                                                           int CI_T, float* dw, float* dbias, int accumulate, int upm, int Cin_total, int ci_off) {
    extern __shared__ float lds[];
    wgrad_reduce_body(lds, slab, splits, Ktot, slabN, Kp, taps, Cout, Cin, CI_T, dw, dbias, accumulate, upm, Cin_total, ci_off, blockIdx.x, blockIdx.y);
}

// Batched form (mcav_wgrad_deferred + mcav_wgrad_reduce_multi): the weight-gradient GEMMs of a whole gradient bucket leave their slabs in
// place and ONE presum launch + ONE reduce launch sum them all, a device table of items telling every workgroup which layer it belongs to.
// Same bodies, same fixed summation order: bit-identical to the per-layer launches (68 launches per step -> 2 per bucket).
__device__ __forceinline__ int find_item(const mcav_wgrad_reduce_item* items, int n, int blk, bool presum) {
    int lo = 0;
    for (int i = 1; i < n; ++i)
        if ((presum ? items[i].pre_first : items[i].red_first) <= blk) lo = i;      // first-block columns are non-decreasing
    return lo;
}

__global__ __launch_bounds__(256) void wgrad_presum_multi_kernel(const mcav_wgrad_reduce_item* items, int n) {
    __shared__ int s_i;
    if (threadIdx.x == 0) {
        int best = -1;
        for (int i = 0; i < n; ++i)
            if (items[i].groups > 0 && items[i].pre_first <= (int)blockIdx.x) best = i;
        s_i = best;
    }
    __syncthreads();
    if (s_i < 0) return;
    const mcav_wgrad_reduce_item it = items[s_i];
    const int local = (int)blockIdx.x - it.pre_first;
    if (local >= it.pre_bx * it.groups) return;
    wgrad_presum_body(it.slab, it.splits, (size_t)it.elems, it.per_group, it.pre, (unsigned)(local % it.pre_bx), (unsigned)(local / it.pre_bx));
}

__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const mcav_wgrad_reduce_item* items, int n) {
    extern __shared__ float lds[];
    __shared__ int s_i;
    if (threadIdx.x == 0) s_i = find_item(items, n, (int)blockIdx.x, false);
    __syncthreads();
    const mcav_wgrad_reduce_item it = items[s_i];
    const int local = (int)blockIdx.x - it.red_first;
    if (local >= it.red_gx * it.red_gy) return;
    const float* src = it.groups > 0 ? it.pre : it.slab;
    wgrad_reduce_body(lds, src, it.groups > 0 ? it.groups : it.splits, it.Ktot, it.slabN, it.Kp, it.taps, it.Cout, it.Cin, it.ci_t, it.dw, it.dbias,
                      it.accumulate, it.upm, it.cin_total, it.ci_off, local % it.red_gx, local / it.red_gx);
}

// OIHW -> packed.  transposed = 0: packed[n = co][tap][k = ci];  transposed = 1: packed[n = ci][tap][k = co].
__global__ void pack_weights_kernel(const float* w, int Cout, int Cin, int taps, int transposed, float* packed, int Np, int Kp, int Kstride) {
    const size_t total = (size_t)Np * Kstride;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(e / Kstride), kf = (int)(e - (size_t)n * Kstride);
        const int tap = kf / Kp, k = kf - tap * Kp;
        float v = 0.f;
        if (tap < taps) {
            const int co = transposed ? k : n, ci = transposed ? n : k;
            if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * taps + tap];
        }
        packed[e] = v;
    }
}

// Many filters in one launch: after an optimiser step every packed copy is stale; one launch re-derives them all.
// Both directions go through an LDS tile so that the OIHW reads and the packed writes are each contiguous:
//   forward copy   : one block per packed row n (= output channel): its Cin x taps filter is one contiguous run;
//   transposed copy: one block per (PK_CI input channels x 64 output channels) tile: 64 runs of PK_CI x taps floats in,
//                    PK_CI x taps runs of 64 floats out.
struct PackItem {
    const float* src;
    float* dst;
    int Cout, Cin, taps, transposed, Np, Kp, Kstride, first_block;   // first_block: prefix sum of pack_blocks() over the preceding items
};

constexpr int PK_LDS = 5120;                  // floats of staging per block (20 KB: eight blocks per CU; at 48 KB three blocks left the copies latency-bound)
constexpr int PK_CO = 64;

__host__ __device__ inline int pack_ci_tile(int taps) { const int t = 79 / taps; return t < 1 ? 1 : (t > 8 ? 8 : t); }      // 64 x (tile x taps + 1) <= PK_LDS

// Merged-tap copies of the decoder's upsample convolutions (mcav_pack_weights_upmerge / _upmerge_adj) as items of the same launch (round 4:
// ten small launches fewer per step): PackItem.transposed bit 3 = the forward copy [4 classes][Np][4 merged taps][C1], bit 4 = the adjoint copy
// [Np = c1][16 taps (u, v)][Kp = co]; PackItem.taps carries C1; a block handles PK_MERGE consecutive elements.
constexpr int PK_MERGE = 2048;

__device__ __forceinline__ float upmerge_elem(const float* w, int Cout, int Cin, int C1, int Np, size_t e) {
    const int c = (int)(e % C1);
    const int mt = (int)((e / C1) % 4), n = (int)((e / ((size_t)C1 * 4)) % Np), cls = (int)(e / ((size_t)C1 * 4 * Np));
    const int py = cls >> 1, px = cls & 1, a = mt >> 1, b = mt & 1;
    float sum = 0.f;
    if (n < Cout) {
        const float* wc = w + ((size_t)n * Cin + c) * 9;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const bool iny = py == 0 ? (a == 0 ? ky == 0 : ky >= 1) : (a == 0 ? ky <= 1 : ky == 2);
                const bool inx = px == 0 ? (b == 0 ? kx == 0 : kx >= 1) : (b == 0 ? kx <= 1 : kx == 2);
                if (iny && inx) sum += wc[ky * 3 + kx];
            }
    }
    return sum;
}

// V[u][v][c1][co] = sum of w[co][c1][ky][kx] over ky in Sy(u), kx in Sy(v); Sy(0) = {2}, Sy(1) = {1,2}, Sy(2) = {0,1}, Sy(3) = {0}.
__device__ __forceinline__ float upmerge_adj_elem(const float* w, int Cout, int Cin, int C1, int Kp, size_t e) {
    const int k = (int)(e % Kp), tap = (int)((e / Kp) % 16), n = (int)(e / ((size_t)Kp * 16));
    const int u = tap >> 2, v = tap & 3;
    float sum = 0.f;
    if (n < C1 && k < Cout) {
        const float* wc = w + ((size_t)k * Cin + n) * 9;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const bool iny = u == 0 ? ky == 2 : (u == 1 ? ky >= 1 : (u == 2 ? ky <= 1 : ky == 0));
                const bool inx = v == 0 ? kx == 2 : (v == 1 ? kx >= 1 : (v == 2 ? kx <= 1 : kx == 0));
                if (iny && inx) sum += wc[ky * 3 + kx];
            }
    }
    return sum;
}

__host__ __device__ inline size_t pack_merge_total(int transposed, int c1, int Np, int Kp) {
    return (transposed & 16) ? (size_t)Np * 16 * Kp : (size_t)4 * Np * 4 * c1;
}

__host__ __device__ inline int pack_blocks(int transposed, int taps, int Np, int Kp) {
    if (transposed & 24) return (int)((pack_merge_total(transposed, taps, Np, Kp) + PK_MERGE - 1) / PK_MERGE);
    if (!(transposed & 1)) return Np;
    const int ct = pack_ci_tile(taps);
    return ((Np + ct - 1) / ct) * ((Kp + PK_CO - 1) / PK_CO);
}

__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const PackItem* items, int nitems) {
    __shared__ float buf[PK_LDS];
    int lo = 0, hi = nitems - 1;               // binary search the item that owns this block
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    PackItem it = items[lo];
    if (it.transposed & 24) {                           // a merged-tap copy: taps = C1
        const size_t total = pack_merge_total(it.transposed, it.taps, it.Np, it.Kp);
        const size_t e0 = (size_t)(blockIdx.x - it.first_block) * PK_MERGE;
        for (int k = 0; k < PK_MERGE / 256; ++k) {
            const size_t e = e0 + (size_t)k * 256 + threadIdx.x;
            if (e < total)
                it.dst[e] = (it.transposed & 16) ? upmerge_adj_elem(it.src, it.Cout, it.Cin, it.taps, it.Kp, e)
                                                 : upmerge_elem(it.src, it.Cout, it.Cin, it.taps, it.Np, e);
        }
        return;
    }
    const bool to_bf16 = (it.transposed & 6) != 0;      // bit 1: the destination holds bf16 (the filter copies of the bf16 MFMA kernels)
    const bool planes = (it.transposed & 4) != 0;       // bit 2: ... as three planes h, m, l of Np x Kstride elements each (the fp32 contraction on split operands)
    it.transposed &= 1;
    __bf16* const dst16 = reinterpret_cast<__bf16*>(it.dst);
    const size_t plane = (size_t)it.Np * it.Kstride;
    auto put16 = [&](size_t o, float v) {
        const __bf16 h = (__bf16)v;
        dst16[o] = h;
        if (planes) {
            const float r1 = v - (float)h;
            const __bf16 m = (__bf16)r1;
            dst16[plane + o] = m;
            dst16[2 * plane + o] = (__bf16)(r1 - (float)m);
        }
    };
    const int bl = blockIdx.x - it.first_block, tid = threadIdx.x;
    if (!it.transposed) {
        const int n = bl, run = it.Cin * it.taps;
        float* drow = it.dst + (size_t)n * it.Kstride;
        const bool staged = run <= PK_LDS;
        if (n < it.Cout && staged) {
            const float* srow = it.src + (size_t)n * run;
            for (int q = tid; q < run; q += 256) buf[q] = srow[q];
        }
        __syncthreads();
        for (int kf = tid; kf < it.Kstride; kf += 256) {
            const int tap = kf / it.Kp, k = kf - tap * it.Kp;
            float v = 0.f;
            if (n < it.Cout && tap < it.taps && k < it.Cin) v = staged ? buf[k * it.taps + tap] : it.src[((size_t)n * it.Cin + k) * it.taps + tap];
            if (to_bf16) put16((size_t)n * it.Kstride + kf, v);
            else drow[kf] = v;
        }
        return;
    }
    const int ct = pack_ci_tile(it.taps), run = ct * it.taps, ld = run + 1;
    const int cob = (it.Kp + PK_CO - 1) / PK_CO;
    const int ci0 = (bl / cob) * ct, co0 = (bl % cob) * PK_CO;
    for (int e = tid; e < PK_CO * run; e += 256) {
        const int co = e / run, q = e - co * run;
        float v = 0.f;
        if (co0 + co < it.Cout && ci0 + q / it.taps < it.Cin) v = it.src[((size_t)(co0 + co) * it.Cin + ci0) * it.taps + q];
        buf[co * ld + q] = v;
    }
    __syncthreads();
    for (int e = tid; e < PK_CO * run; e += 256) {
        const int q = e / PK_CO, co = e - q * PK_CO;      // q = r * taps + tap
        const int r = q / it.taps, tap = q - r * it.taps;
        if (ci0 + r < it.Np && co0 + co < it.Kp) {
            const size_t o = (size_t)(ci0 + r) * it.Kstride + tap * it.Kp + co0 + co;
            if (to_bf16) put16(o, buf[co * ld + q]);
            else it.dst[o] = buf[co * ld + q];
        }
    }
}

inline int round_up(int v, int a) { return (v + a - 1) / a * a; }

inline int kstride_of(int taps, int Kp) { return round_up(taps * Kp, CK); }

inline int pick_tile(const mcav_igemm_desc* d, long M) {
    if (d->tile) return d->tile;
    if (d->n_count <= 16) return (M >= 256 * 64 && d->kh * d->kw <= 16) ? 4 : 6;      // (a 25-tap table for 256 rows is 26 KB of LDS: 64x16 tiles, 0.071 -> 0.051 ms)
    // 32 output channels: 128x32 tiles (31 KB of LDS with the table, five workgroups per CU) beat 256x32 (56-60 KB, two) on every such
    // launch of the step: 96->32 @96x320 merged-tap forward 0.333 -> 0.289 ms, its pooled adjoint 0.089 -> 0.071, 64->32 @48x160 0.081 -> 0.070
    if (d->n_count <= 32) return 7;
    // The reflection-adjoint kind carries three extra register stages for its border loads: on the 128-row tiles that is 122 VGPRs and two
    // workgroups per CU.  Measured (MI355X, bench.py --layer-report, 96x320 / 48x160 maps): 128x64 0.370 / 0.129 / 0.181 ms, 64x64x32
    // 0.331 / 0.113 / 0.159, 64x64x16 0.300 / 0.100 / 0.154 -- the small maps keep the choices below.  (Re-measured with the reflected sources
    // in tables: 64x64x16 0.265 / 0.078 / 0.131, 128x64 0.274 / 0.080 / 0.134, 64x64x32 0.292 / 0.085 / 0.138.)
    if (d->mode == MCAV_G_ADJ_REFLECT && ((M + 127) / 128) * ((d->n_count + 63) / 64) >= 1024) return 2;
    // measured on MI355X (tools/conv_bench.py): small output tiles with 32-deep K-tiles win at every layer shape of the step --
    // more co-resident workgroups hide the load round trips, and a 32-deep tile halves the barriers per FLOP
    if (d->mode != MCAV_G_SMALLC && ((M + 127) / 128) * ((d->n_count + 63) / 64) >= 1024) return 1;   // many rows (layer1, 48x160): 128x64 tiles, 104 vs 96 TF/s
    // few 64x64 workgroups (layer4's 6x20 maps, M = 2880): 32x64 tiles double them -- 84 -> 104 TF/s on 512->512
    // ... when halving the tiles shortens the longest CU's queue: up to 128 tiles (each half on a CU of its own) or 257..511 (two halves
    // balance better than one whole); in between the halves only pair up on the same CUs (6x20 512->256, 180 tiles: 74 TF/s halved, 87 whole)
    const long t64 = ((M + 63) / 64) * ((d->n_count + 63) / 64);
    if (d->mode != MCAV_G_SMALLC && d->Kp % 32 == 0 && (t64 <= 128 || (t64 > 256 && t64 < 512))) return 12;
    if (d->mode != MCAV_G_SMALLC && d->Kp % 32 == 0) return 10;
    const long t128x64 = ((M + 127) / 128) * ((d->n_count + 63) / 64);
    return t128x64 >= 1024 ? 1 : 2;
}

void tile_dims(int id, int& BM, int& BN) {
    switch (id) {
        case 1: BM = 128; BN = 64; break;
        case 2: BM = 64; BN = 64; break;
        case 3: BM = 256; BN = 32; break;
        case 4: BM = 256; BN = 16; break;
        case 5: BM = 128; BN = 128; break;
        case 7: BM = 128; BN = 32; break;
        case 8: BM = 128; BN = 64; break;
        case 9: BM = 128; BN = 128; break;
        case 10: BM = 64; BN = 64; break;
        case 11: BM = 64; BN = 64; break;
        case 12: BM = 32; BN = 64; break;
        default: BM = 64; BN = 16; break;
    }
}

bool fill_params(const mcav_igemm_desc* d, IgemmParams& p, int& tile) {
    if (!d || !d->x1 || !d->w || !d->y) return false;
    if (d->B <= 0 || d->Hs <= 0 || d->Ws <= 0 || d->Hd <= 0 || d->Wd <= 0 || d->C1 <= 0 || d->C2 < 0) return false;
    if (d->C2 > 0 && !d->x2) return false;
    if (d->kh <= 0 || d->kw <= 0 || d->Np <= 0 || d->Kp <= 0 || d->kh * d->kw > 64) return false;
    if (d->n_begin < 0 || d->n_count <= 0 || d->n_begin + d->n_count > d->Np) return false;
    if (d->mode == MCAV_G_SMALLC) { if (d->Kp != 4 || d->C1 != 4 || d->C2 != 0) return false; }
    else if (d->Kp % CK != 0 || d->Kp < d->C1 + d->C2) return false;
    if (d->C2 > 0 && (d->C1 % 4 != 0)) return false;
    if (d->mode == MCAV_G_ADJ_REFLECT && (d->kh != 3 || d->kw != 3 || d->Hs != d->Hd || d->Ws != d->Wd || d->Hs < 2 || d->Ws < 2)) return false;
    if (d->pad_mode == MCAV_PAD_REFLECT && (d->Hs < 2 || d->Ws < 2)) return false;
    if (d->up1 && ((d->Hs & 1) || (d->Ws & 1))) return false;
    if (d->pool && ((d->Hd & 1) || (d->Wd & 1) || d->mode == MCAV_G_ADJ_STRIDE2)) return false;
    if (d->pool && d->stats) return false;
    p.g.x1 = d->x1; p.g.x2 = d->x2; p.g.B = d->B; p.g.Hs = d->Hs; p.g.Ws = d->Ws; p.g.C1 = d->C1; p.g.C2 = d->C2; p.g.up1 = d->up1;
    p.g.mode = d->mode; p.g.stride = d->stride; p.g.sign = d->sign; p.g.offset = d->offset; p.g.pad_mode = d->pad_mode;
    p.w = d->w; p.kh = d->kh; p.kw = d->kw; p.Kp = d->Kp; p.taps = d->kh * d->kw; p.Kstride = kstride_of(p.taps, d->Kp);
    p.y = d->y; p.Hd = d->Hd; p.Wd = d->Wd; p.Cd = d->Cd; p.n_begin = d->n_begin; p.n_count = d->n_count; p.y_choff = d->y_choff;
    p.bias = d->bias; p.act = d->act; p.dact_aux = d->dact_aux; p.dact = d->dact; p.addend = d->addend; p.pool = d->pool; p.stats = d->stats;
    if (d->stats_x && (!d->stats || !d->stats_mean || !d->stats_invstd || d->pool || d->y_choff != 0 || d->n_begin != 0 || d->n_count != d->Cd)) return false;
    p.stats_x = d->stats ? d->stats_x : nullptr; p.stats_mean = d->stats_mean; p.stats_invstd = d->stats_invstd;
    const long Mlin = (long)d->B * d->Hd * d->Wd;
    if ((long)d->B * d->Hs * d->Ws * (d->C1 > d->C2 ? d->C1 : d->C2) * 4 >= 0x7fffffffL) return false;   // 32-bit byte offsets
    if ((long)d->Np * kstride_of(d->kh * d->kw, d->Kp) * 4 >= 0x7fffffffL) return false;
    p.no_tab = (d->tile >> 8) & 1;
    p.ksplit = 1; p.kslab = nullptr;
    p.wm = d->w_upmerge; p.Np_all = d->Np; p.upm = 0;
    tile = pick_tile(d, Mlin) & 0xff;
    if (tile == 0) { mcav_igemm_desc dd = *d; dd.tile = 0; tile = pick_tile(&dd, Mlin); }
    if (((tile >= 8 && tile <= 10) || tile == 12) && (d->mode == MCAV_G_SMALLC || d->Kp % 32 != 0)) return false;      // 32-deep K-tiles need Kp % 32 == 0
    if (tile == 11 && (d->mode == MCAV_G_SMALLC || d->Kp % 64 != 0)) return false;
    int BM, BN;
    tile_dims(tile, BM, BN);
    p.groups = d->groups > 1 ? d->groups : 1;
    if (p.groups > 1) {
        if ((d->mode != MCAV_G_DIRECT && d->mode != MCAV_G_SMALLC) || d->pool || d->B % p.groups != 0) return false;
        p.Mc = (int)(Mlin / p.groups);
        p.McP = round_up(p.Mc, BM);
        p.M = p.groups * p.McP;
    } else if (d->mode == MCAV_G_ADJ_STRIDE2) {
        p.Mc = d->B * ((d->Hd + 1) / 2) * ((d->Wd + 1) / 2);
        p.McP = round_up(p.Mc, BM);
        p.M = 4 * p.McP;
    } else {
        p.Mc = 0; p.McP = 1;
        if (Mlin > 0x7fffffffL) return false;
        p.M = (int)Mlin;
        // merged-tap upsample (see mcav_igemm_desc.w_upmerge): everything the table-driven UPM kernel needs must hold, else the plain path
        const int ck = (tile >= 8 && tile <= 10) || tile == 12 ? 32 : (tile == 11 ? 64 : 16);
        if (d->w_upmerge && d->mode == MCAV_G_DIRECT && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->sign == 1 && d->offset == -1 &&
            d->pad_mode == MCAV_PAD_REFLECT && d->up1 == 1 && d->C2 > 0 && !(d->Hd & 1) && !(d->Wd & 1) && d->Hd == d->Hs && d->Wd == d->Ws &&
            !d->pool && !d->stats && d->C1 % ck == 0 && d->C2 % ck == 0 && d->C1 + d->C2 == d->Kp && (d->C1 & 3) == 0 && !p.no_tab &&
            (long)4 * d->Np * 4 * d->C1 * 4 < 0x7fffffffL) {
            p.upm = 1;
            p.bm = BM;
            p.Mc = d->B * (d->Hd / 2) * (d->Wd / 2);
            p.McP = round_up(p.Mc, BM);
            p.M = 4 * p.McP;
        }
    }
    p.mtiles = (p.M + BM - 1) / BM;
    p.ntiles = (d->n_count + BN - 1) / BN;
    return true;
}

// y = epilogue(sum of the K-split partials): one element per thread and pass, fixed summation order
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ slab, int splits, size_t n, int Cd, float* __restrict__ y,
                                                            const float* __restrict__ bias, int n_begin, int act, const float* __restrict__ dact_aux,
                                                            int dact, const float* __restrict__ addend) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = 0.f;
        for (int k = 0; k < splits; ++k) v += slab[(size_t)k * n + i];
        if (bias) v += bias[n_begin + (int)(i % (size_t)Cd)];
        v = act_fwd(v, act);
        if (dact_aux && !(dact & MCAV_DACT_AFTER_ADDEND)) v *= act_bwd(dact_aux[i], dact & 0xff);
        if (addend) v += addend[i];
        if (dact_aux && (dact & MCAV_DACT_AFTER_ADDEND)) v *= act_bwd(dact_aux[i], dact & 0xff);
        y[i] = v;
    }
}

// per-stream slab of the K-split launches (grown on demand, never freed; not grown while the stream is being captured)
inline float* ksplit_slab(hipStream_t s, size_t bytes) {
    static std::mutex mu;
    static std::map<hipStream_t, std::pair<void*, size_t>> slabs;
    std::lock_guard<std::mutex> lock(mu);
    auto& e = slabs[s];
    if (e.second < bytes) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) != hipSuccess) return nullptr;
        if (st != hipStreamCaptureStatusNone) {
            // No allocation inside a capture.  The step is captured on one stream after eager warm-up steps (mcav/graph.py), so a slab of
            // the warm-up's stream is large enough and nothing else uses it while the graph runs: borrow it (same splits, same bits as eager).
            for (auto& kv : slabs)
                if (kv.second.second >= bytes) return reinterpret_cast<float*>(kv.second.first);
            return nullptr;
        }
        void* ptr = nullptr;
        if (hipMalloc(&ptr, bytes) != hipSuccess) return nullptr;
        e = {ptr, bytes};          // (the old, smaller slab is left to the launches still queued on it)
    }
    return reinterpret_cast<float*>(e.first);
}

template <class T>
inline void launch_igemm(const IgemmParams& p_in, hipStream_t s) {
    IgemmParams p = p_in;
    p.ksplit = 1; p.kslab = nullptr;
    const bool c4ok = (p.g.C1 & 3) == 0 && (p.g.C2 & 3) == 0 && (p.g.C2 == 0 || (p.g.C1 & 15) == 0);
    const bool fast_mode = p.g.mode == MCAV_G_DIRECT || (p.g.mode == MCAV_G_ADJ_STRIDE2 && p.g.C2 == 0) || p.g.mode == MCAV_G_SMALLC;
    const bool tab = (p.g.mode == MCAV_G_DIRECT || (p.g.mode == MCAV_G_ADJ_STRIDE2 && p.g.C2 == 0)) && c4ok && p.taps <= TAB_TAPS &&
                     p.g.C1 + p.g.C2 == p.Kp && p.Kp % T::KD == 0 && (p.g.C2 == 0 || p.g.C1 % T::KD == 0) && !p.no_tab;
    // (tall tiles put rows of many image lines into one wavefront: most wavefronts would take the border path, so they keep the general kernel)
    const bool tab_refl = p.g.mode == MCAV_G_ADJ_REFLECT && c4ok && p.g.C2 == 0 && p.g.C1 == p.Kp && p.Kp % T::KD == 0 && !p.no_tab && T::AROWS <= 2;
    if (tab && !p.upm) {
        // A handful of tiles with a long K loop (PoseNet conv5-7 and their data gradients: 2 .. 90 tiles of 36 .. 72 K-tiles, one workgroup
        // per CU working through them alone): K is cut over up to 8 workgroups per tile, splitk_finish_kernel sums the partials (fixed order)
        // and applies the epilogue.
        const bool plain = !p.pool && !p.stats && p.groups <= 1 && p.y_choff == 0 && p.n_count == p.Cd && p.n_begin == 0;
        const int tiles = p.mtiles * p.ntiles, ktiles = p.Kp / T::KD * p.taps;
        static const int enabled = MCAV_KNOB_INT("MCAV_KSPLIT", 1);
        static const int max_tiles = MCAV_KNOB_INT("MCAV_KSPLIT_TILES", 256);
        if (enabled && plain && tiles <= max_tiles && ktiles >= 32) {
            int ksp = 1024 / tiles;
            if (ksp > 8) ksp = 8;
            if (ksp > ktiles / 4) ksp = ktiles / 4;
            if (ksp >= 2) {
                float* slab = ksplit_slab(s, (size_t)p.g.B * p.Hd * p.Wd * p.Cd * 4 * ksp);
                if (slab) { p.ksplit = ksp; p.kslab = slab; }
            }
        }
    }
    const int grid = p.mtiles * p.ntiles * p.ksplit;
    const size_t tab_bytes = sizeof(unsigned) * (size_t)p.taps * T::BM * (p.g.C2 > 0 ? 2 : 1);
    const size_t refl_bytes = sizeof(unsigned) * (size_t)p.taps * T::BM * 4;      // + the three tables of reflected sources
    if (p.upm) timed_launch(igemm_tab_kernel<T, 2>, grid, dim3(256), sizeof(unsigned) * 13 * T::BM, s, p);
    else if (tab) timed_launch(igemm_tab_kernel<T, 0>, grid, dim3(256), tab_bytes, s, p);
    else if (tab_refl) timed_launch(igemm_tab_kernel<T, 1>, grid, dim3(256), refl_bytes, s, p);
    else if (fast_mode && c4ok) timed_launch(igemm_kernel<T, K_FAST>, grid, dim3(256), 0, s, p);
    else if (p.g.mode == MCAV_G_ADJ_REFLECT && c4ok && p.g.C2 == 0) timed_launch(igemm_kernel<T, K_REFLADJ>, grid, dim3(256), 0, s, p);
    else timed_launch(igemm_kernel<T, K_GENERIC>, grid, dim3(256), 0, s, p);
    if (p.ksplit > 1) {
        const size_t n = (size_t)p.g.B * p.Hd * p.Wd * p.Cd;
        const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
        timed_launch(splitk_finish_kernel, blocks, dim3(256), 0, s, (const float*)p.kslab, p.ksplit, n, p.Cd, p.y, p.bias, p.n_begin, p.act, p.dact_aux,
                     p.dact, p.addend);
    }
}

}  // namespace mcav

using namespace mcav;

int mcav_bf16_igemm_mtiles(const mcav_igemm_desc* d);             // conv_bf16.hip: 0 = the launch does not run there
int mcav_patch_f32_mtiles(const mcav_igemm_desc* d);              // conv_bf16.hip: blocks of the fp32 patch-in-LDS kernel, 0 = not one of its launches
bool mcav_try_patch_f32(const mcav_igemm_desc* d, hipStream_t s);
int mcav_stem_mtiles(const mcav_igemm_desc* d);                   // conv_stem.hip: 0 = not the image stem

MCAV_EXPORT int mcav_igemm_mtiles(const mcav_igemm_desc* d) {
    IgemmParams p;
    int tile;
    if (!fill_params(d, p, tile)) return MCAV_E_INVALID;
    if (const int st = mcav_stem_mtiles(d)) return st;             // the stem kernel's 8 x 32 output tiles
    if (const int pt = mcav_patch_f32_mtiles(d)) return pt;         // the fp32 patch-in-LDS kernel's blocks (conv_bf16.hip)
    if (d->mma != 0) {                              // the bf16 kernels choose their own tile shape
        const int mt = mcav_bf16_igemm_mtiles(d);
        if (mt > 0) return mt;
    }
    return p.mtiles;
}

bool mcav_try_halo(const mcav_igemm_desc* d, hipStream_t s);      // conv_halo.hip: narrow high-resolution 3x3 layers (forward and reflect-adjoint)
int mcav_bf16_igemm(const mcav_igemm_desc* d, hipStream_t s);      // conv_bf16.hip: 1 = not eligible
bool mcav_try_stem(const mcav_igemm_desc* d, const mcav::IgemmParams& p, hipStream_t s);      // conv_stem.hip: the 7x7 stride-2 image stem
int mcav_stem_mtiles(const mcav_igemm_desc* d);
bool mcav_bf16_wgrad_plan(const mcav_wgrad_desc* d, mcav::WgradPlan& pl);
void mcav_bf16_wgrad_launch(const mcav::WgradParams& p, hipStream_t s);

MCAV_EXPORT int mcav_igemm(const mcav_igemm_desc* d, void* stream) {
    IgemmParams p;
    int tile;
    if (!fill_params(d, p, tile)) return MCAV_E_INVALID;
    hipStream_t s = as_stream(stream);
    if (mcav_try_halo(d, s)) return launch_status();
    if (mcav_try_stem(d, p, s)) return launch_status();
    if (mcav_try_patch_f32(d, s)) return launch_status();
    if (d->mma != 0) {                    // bf16 MFMA tiles (conv_bf16.hip) where the launch qualifies, else the fp32 kernels below
        const int rc = mcav_bf16_igemm(d, s);
        if (rc != 1) return rc;
    }
    switch (tile) {
        case 1: launch_igemm<Tile128x64>(p, s); break;
        case 2: launch_igemm<Tile64x64>(p, s); break;
        case 3: launch_igemm<Tile256x32>(p, s); break;
        case 4: launch_igemm<Tile256x16>(p, s); break;
        case 5: launch_igemm<Tile128x128>(p, s); break;
        case 6: launch_igemm<Tile64x16>(p, s); break;
        case 7: launch_igemm<Tile128x32>(p, s); break;
        case 8: launch_igemm<Tile128x64k32>(p, s); break;
        case 9: launch_igemm<Tile128x128k32>(p, s); break;
        case 10: launch_igemm<Tile64x64k32>(p, s); break;
        case 11: launch_igemm<Tile64x64k64>(p, s); break;
        case 12: launch_igemm<Tile32x64k32>(p, s); break;
        default: return MCAV_E_INVALID;
    }
    return launch_status();
}

namespace mcav {

}  // namespace mcav
int mcav_stem_wgrad_splits(const mcav_wgrad_desc* d);
void mcav_stem_wgrad_launch(const mcav_wgrad_desc* d, float* slab, int Ktot, int slabN, int splits, hipStream_t s);
int mcav_halo_wgrad_splits(const mcav_wgrad_desc* d);
void mcav_halo_wgrad_launch(const mcav_wgrad_desc* d, float* slab, int slabN, int splits, hipStream_t s);
namespace mcav {

bool plan_wgrad(const mcav_wgrad_desc* d, WgradPlan& pl) {
    if (!d || !d->x1 || !d->dy || !d->dw_oihw) return false;
    if (d->B <= 0 || d->Hs <= 0 || d->Ws <= 0 || d->Hd <= 0 || d->Wd <= 0 || d->C1 <= 0 || d->C2 < 0 || d->Cout <= 0 || d->Cin <= 0) return false;
    if (d->C2 > 0 && (!d->x2 || d->C1 % 4 != 0)) return false;
    if (d->mode != MCAV_G_DIRECT && d->mode != MCAV_G_SMALLC) return false;
    if (d->mode == MCAV_G_SMALLC && (d->Kp != 4 || d->C1 != 4 || d->C2 != 0)) return false;
    if (d->mode == MCAV_G_DIRECT && (d->Kp % CK != 0 || d->Kp < d->C1 + d->C2)) return false;
    if (d->Cin > d->Kp) return false;
    WgradParams& p = pl.p;
    p.g.x1 = d->x1; p.g.x2 = d->x2; p.g.B = d->B; p.g.Hs = d->Hs; p.g.Ws = d->Ws; p.g.C1 = d->C1; p.g.C2 = d->C2; p.g.up1 = d->up1;
    p.g.mode = d->mode; p.g.stride = d->stride; p.g.sign = d->sign; p.g.offset = d->offset; p.g.pad_mode = d->pad_mode;
    p.kh = d->kh; p.kw = d->kw; p.Kp = d->Kp; p.taps = d->kh * d->kw; p.Ktot = p.taps * d->Kp;
    p.dy = d->dy; p.Hd = d->Hd; p.Wd = d->Wd; p.Cdy = d->Cdy; p.dy_choff = d->dy_choff; p.Cout = d->Cout;
    p.CoutLoad = round_up(d->Cout, 4) <= d->Cdy - d->dy_choff ? round_up(d->Cout, 4) : d->Cout;
    const long M = (long)d->B * d->Hd * d->Wd;
    if (M > 0x7fffffffL) return false;
    if ((long)d->B * d->Hs * d->Ws * (d->C1 > d->C2 ? d->C1 : d->C2) * 4 >= 0x7fffffffL || M * d->Cdy * 4 >= 0x7fffffffL) return false;
    p.Mpix = (int)M;
    p.slabN = round_up(d->Cout, 16);
    p.upm = 0; p.Hf = d->Hd; p.Wf = d->Wd; p.patch = 0; p.split_planes = 0;
    if (d->upm) {      // merged-tap upsample half: 16 (class, merged tap) x Kp rows, reduction over the low-resolution pixels
        if (d->mode != MCAV_G_DIRECT || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->sign != 1 || d->offset != -1 || d->pad_mode != MCAV_PAD_REFLECT ||
            !d->up1 || d->C2 != 0 || d->Hd != d->Hs || d->Wd != d->Ws || (d->Hd & 1) || (d->Wd & 1) || d->dbias || d->Kp != d->C1 || (d->C1 & 15) ||
            (d->Cdy & 3) || (d->dy_choff & 3) || (p.CoutLoad & 3))
            return false;
        p.upm = 1;
        p.taps = 16; p.Ktot = 16 * d->Kp;
        p.Hd = d->Hd / 2; p.Wd = d->Wd / 2;
        p.Mpix = d->B * p.Hd * p.Wd;
    }
    int tile = d->tile & 0xff;
    if (!tile) {
        if (d->Cout <= 16) tile = round_up(p.Ktot, 64) < round_up(p.Ktot, 256) ? 6 : 4;
        else if (d->Cout <= 32) tile = 7;      // 128x32 (48 KB of LDS, three workgroups per CU; 256x32 needs 80 KB): 16x25 -> 32 0.052 -> 0.044 ms
        else tile = 2;            // 64x64 beats 128x64 on every layer shape of the step (tools/conv_bench.py wgrad)
    }
    if (p.upm) tile = d->Cout <= 16 ? 6 : 2;                     // 64-row tiles: a row tile never straddles a parity class (4 Kp rows each)
    if (tile != 1 && tile != 2 && tile != 3 && tile != 4 && tile != 6 && tile != 7) return false;
    pl.tile = tile;
    int BM, BN;
    tile_dims(tile, BM, BN);
    p.mtiles = (p.Ktot + BM - 1) / BM;
    p.ntiles = (d->Cout + BN - 1) / BN;
    const int out_tiles = p.mtiles * p.ntiles;
    static const int target_wgs = MCAV_KNOB_INT("MCAV_WGRAD_TARGET_WGS", 0);
    int max_splits = (p.Mpix + 8 * KP - 1) / (8 * KP);           // at least 8 K-tiles each
    if (max_splits > 512) max_splits = 512;
    if (max_splits < 1) max_splits = 1;
    int splits;
    if (target_wgs > 0) {                                        // tuning knob: aim at that many workgroups
        splits = (target_wgs + out_tiles - 1) / out_tiles;
    } else {
        // Whole generations of resident workgroups: 1044 workgroups on 1024 slots (128->128 at 24x80 with "about 1024") run a second
        // generation for 2 % of the work (0.142 ms; 0.124 at twice the splits).  Take the fewest splits, from one generation's worth up to
        // three, whose workgroup count fills its last generation to 95 % (else the best fill).
        const int occ = tile == 2 ? 4 : (tile == 7 ? 3 : (tile == 6 ? 4 : 2)), slots = 256 * occ;
        auto wgs_of = [&](int sp) {
            const int pps = round_up((p.Mpix + sp - 1) / sp, KP);
            return out_tiles * ((p.Mpix + pps - 1) / pps);
        };
        int lo = slots / out_tiles, hi = 3 * slots / out_tiles + 1;
        if (lo < 1) lo = 1;
        if (lo > max_splits) lo = max_splits;
        if (hi > max_splits) hi = max_splits;
        splits = lo;
        double best = -1.0;
        for (int sp = lo; sp <= hi; ++sp) {
            const int w = wgs_of(sp), gens = (w + slots - 1) / slots;
            const double fill = (double)w / ((double)gens * slots);
            if (fill > best + 1e-9) { best = fill; splits = sp; }
            if (fill >= 0.95) { splits = sp; break; }
        }
    }
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    p.pix_per_split = round_up((p.Mpix + splits - 1) / splits, KP);
    // table-driven kernel: the per-workgroup offset table (pixels of a split x taps touched by a row tile x sources) must fit
    pl.use_tab = false;
    const int wave_ch = BM / 4;
    const bool fast = d->mode == MCAV_G_DIRECT && (d->C1 & 3) == 0 && (d->C2 & 3) == 0 && (d->C2 == 0 || d->C1 % wave_ch == 0) &&
                      (p.CoutLoad & 3) == 0 && (d->Cdy & 3) == 0 && (d->dy_choff & 3) == 0 && !((d->tile >> 8) & 1);
    p.tab_cht_log2 = 20;
    if (fast) {
        int ntmax = 1;
        for (int mt = 0; mt < p.mtiles; ++mt) {
            const int lo = mt * BM / d->Kp, hi = (mt * BM + BM - 1) / d->Kp < p.taps - 1 ? (mt * BM + BM - 1) / d->Kp : p.taps - 1;
            if (hi - lo + 1 > ntmax) ntmax = hi - lo + 1;
        }
        const int epp = p.upm ? ntmax + 1 : ntmax * (d->C2 > 0 ? 2 : 1);  // table entries per pixel (upm: + the dy offset)
        if ((long)p.pix_per_split * epp <= WG_TABCAP) {
            pl.use_tab = true;                                            // one chunk
        } else if (epp <= 16) {
            int lg = 2;                                                   // two half-buffers of 2^lg tiles: 2^lg * KP * epp <= WG_TABCAP / 2
            while ((2 << lg) * KP * epp <= WG_TABCAP / 2) ++lg;
            p.tab_cht_log2 = lg;
            pl.use_tab = true;
        }
    }
    p.splits = (p.Mpix + p.pix_per_split - 1) / p.pix_per_split;
    if (p.upm && !pl.use_tab) return false;                      // the merged form exists in the table-driven kernel only
    const int halo_splits = p.upm ? 0 : mcav_halo_wgrad_splits(d);      // narrow high-resolution layers: conv_halo.hip writes the slab partials
    pl.use_halo = halo_splits > 0;
    if (pl.use_halo) { p.splits = halo_splits; pl.use_tab = false; }
    const int stem_splits = p.upm ? 0 : mcav_stem_wgrad_splits(d);      // the image stem: conv_stem.hip writes the slab partials
    pl.use_stem = stem_splits > 0;
    if (pl.use_stem) { p.splits = stem_splits; pl.use_tab = false; }
    p.want_bias = d->dbias != nullptr;
    pl.slab_bytes = align_up(sizeof(float) * (size_t)p.splits * (p.Ktot + 1) * p.slabN, 256);
    pl.groups = p.splits > 8 ? 8 : 0;                            // two-level reduction above 8 splits
    pl.per_group = pl.groups ? (p.splits + pl.groups - 1) / pl.groups : 0;
    if (pl.groups) pl.groups = (p.splits + pl.per_group - 1) / pl.per_group;
    pl.pre_bytes = align_up(sizeof(float) * (size_t)pl.groups * (p.Ktot + 1) * p.slabN, 256);
    const int out_taps = p.upm ? 9 : p.taps;                     // filter taps written to OIHW
    int ci_t = 32;                                               // 32 x (CI_T * taps + 1) floats of LDS <= 64 KB ...
    while (ci_t > 1 && ci_t * out_taps > 480) ci_t >>= 1;
    if (ci_t * out_taps > 480) return false;
    const int co_tiles = (d->Cout + 31) / 32;                    // ... and enough blocks to fill the chip
    while (ci_t > 1 && co_tiles * ((d->Cin + ci_t - 1) / ci_t) < 256) ci_t >>= 1;
    pl.ci_t = ci_t;
    return true;
}

template <class T>
inline void launch_wgrad(const WgradParams& p, bool use_tab, hipStream_t s) {
    const int grid = p.splits * p.mtiles * p.ntiles;
    if (use_tab && p.upm) { timed_launch(wgrad_tab_kernel<T, true>, grid, dim3(256), 0, s, p); return; }
    if (use_tab) { timed_launch(wgrad_tab_kernel<T, false>, grid, dim3(256), 0, s, p); return; }
    const int wave_ch = T::BM / 4;          // channels one wavefront's A columns span
    const bool fast = (p.g.mode == MCAV_G_DIRECT || p.g.mode == MCAV_G_SMALLC) && (p.g.C1 & 3) == 0 && (p.g.C2 & 3) == 0 && (p.g.C2 == 0 || p.g.C1 % wave_ch == 0) &&
                      (p.CoutLoad & 3) == 0 && (p.Cdy & 3) == 0 && (p.dy_choff & 3) == 0 && p.Wd >= 16;
    if (fast) timed_launch(wgrad_kernel<T, K_FAST>, grid, dim3(256), 0, s, p);
    else timed_launch(wgrad_kernel<T, K_GENERIC>, grid, dim3(256), 0, s, p);
}

}  // namespace mcav

MCAV_EXPORT size_t mcav_wgrad_workspace_bytes(const mcav_wgrad_desc* d) {
    WgradPlan pl;
    if (!(d && d->mma != 0 && mcav_bf16_wgrad_plan(d, pl)) && !plan_wgrad(d, pl)) return 0;
    return pl.slab_bytes + pl.pre_bytes;
}

MCAV_EXPORT int mcav_wgrad_uses_bf16(const mcav_wgrad_desc* d) {
    WgradPlan pl;
    return d && d->mma != 0 && mcav_bf16_wgrad_plan(d, pl);
}

// the GEMM part of a weight gradient: partial tiles into the slab at `workspace`
static int wgrad_gemm(const mcav_wgrad_desc* d, void* workspace, size_t workspace_bytes, hipStream_t s, WgradPlan& pl) {
    const bool bf16 = d && d->mma != 0 && mcav_bf16_wgrad_plan(d, pl);      // bf16 MFMA tiles where the launch qualifies (conv_bf16.hip)
    if ((!bf16 && !plan_wgrad(d, pl)) || !workspace) return MCAV_E_INVALID;
    if (workspace_bytes < pl.slab_bytes + pl.pre_bytes) return MCAV_E_WORKSPACE;
    const int cin_total = d->Cin_total > 0 ? d->Cin_total : d->Cin;
    if (d->ci_offset < 0 || d->ci_offset + d->Cin > cin_total) return MCAV_E_INVALID;
    pl.p.slab = reinterpret_cast<float*>(workspace);
    if (bf16) mcav_bf16_wgrad_launch(pl.p, s);
    else if (pl.use_stem) mcav_stem_wgrad_launch(d, pl.p.slab, pl.p.Ktot, pl.p.slabN, pl.p.splits, s);
    else if (pl.use_halo) mcav_halo_wgrad_launch(d, pl.p.slab, pl.p.slabN, pl.p.splits, s);
    else switch (pl.tile) {
        case 1: launch_wgrad<Tile128x64>(pl.p, pl.use_tab, s); break;
        case 2: launch_wgrad<Tile64x64>(pl.p, pl.use_tab, s); break;
        case 3: launch_wgrad<Tile256x32>(pl.p, pl.use_tab, s); break;
        case 4: launch_wgrad<Tile256x16>(pl.p, pl.use_tab, s); break;
        case 6: launch_wgrad<Tile64x16>(pl.p, pl.use_tab, s); break;
        case 7: launch_wgrad<Tile128x32>(pl.p, pl.use_tab, s); break;
        default: return MCAV_E_INVALID;
    }
    return MCAV_OK;
}

static void fill_reduce_item(const mcav_wgrad_desc* d, const WgradPlan& pl, void* workspace, mcav_wgrad_reduce_item& it) {
    it.slab = pl.p.slab;
    it.pre = pl.groups ? reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + pl.slab_bytes) : nullptr;
    it.dw = d->dw_oihw;
    it.dbias = d->dbias;
    it.elems = (unsigned long long)(pl.p.Ktot + 1) * pl.p.slabN;
    it.splits = pl.p.splits; it.groups = pl.groups; it.per_group = pl.per_group;
    it.Ktot = pl.p.Ktot; it.slabN = pl.p.slabN; it.Kp = pl.p.Kp;
    it.taps = pl.p.upm ? 9 : pl.p.taps;
    it.Cout = d->Cout; it.Cin = d->Cin; it.ci_t = pl.ci_t;
    it.accumulate = d->accumulate; it.upm = pl.p.upm;
    it.cin_total = d->Cin_total > 0 ? d->Cin_total : d->Cin;
    it.ci_off = d->ci_offset;
    it.pre_bx = pl.groups ? (int)((it.elems / 4 + 255) / 256) : 0;
    it.red_gx = (d->Cout + 31) / 32;
    it.red_gy = (d->Cin + pl.ci_t - 1) / pl.ci_t;
    it.pre_first = 0; it.red_first = 0;
}

MCAV_EXPORT int mcav_wgrad(const mcav_wgrad_desc* d, void* workspace, size_t workspace_bytes, void* stream) {
    WgradPlan pl;
    hipStream_t s = as_stream(stream);
    const int rc = wgrad_gemm(d, workspace, workspace_bytes, s, pl);
    if (rc != MCAV_OK) return rc;
    mcav_wgrad_reduce_item it;
    fill_reduce_item(d, pl, workspace, it);
    const size_t lds_bytes = sizeof(float) * 32 * (size_t)(it.ci_t * it.taps + 1);
    const float* rsrc = it.slab;
    int rsplits = it.splits;
    if (it.groups) {
        // (timed with the GEMM: the weight gradient is not finished until the slab is reduced -- bench.py's wgrad stage counts both)
        timed_launch(wgrad_presum_kernel, dim3((unsigned)it.pre_bx, it.groups), dim3(256), 0, s, (const float*)it.slab, it.splits, (size_t)it.elems,
                     it.per_group, it.pre);
        rsrc = it.pre;
        rsplits = it.groups;
    }
    timed_launch(wgrad_reduce_kernel, dim3(it.red_gx, it.red_gy), dim3(256), lds_bytes, s, rsrc, rsplits, it.Ktot, it.slabN, it.Kp, it.taps, it.Cout, it.Cin,
                 it.ci_t, it.dw, it.dbias, it.accumulate, it.upm, it.cin_total, it.ci_off);
    return launch_status();
}

MCAV_EXPORT int mcav_wgrad_deferred(const mcav_wgrad_desc* d, void* workspace, size_t workspace_bytes, mcav_wgrad_reduce_item* item, void* stream) {
    if (!item) return MCAV_E_INVALID;
    WgradPlan pl;
    const int rc = wgrad_gemm(d, workspace, workspace_bytes, as_stream(stream), pl);
    if (rc != MCAV_OK) return rc;
    fill_reduce_item(d, pl, workspace, *item);
    return launch_status();
}

MCAV_EXPORT int mcav_wgrad_reduce_plan(mcav_wgrad_reduce_item* items, int n, int* presum_blocks, int* reduce_blocks, size_t* lds_bytes) {
    if (!items || n <= 0 || !presum_blocks || !reduce_blocks || !lds_bytes) return MCAV_E_INVALID;
    int pb = 0, rb = 0;
    size_t lds = 0;
    for (int i = 0; i < n; ++i) {
        items[i].pre_first = pb;
        items[i].red_first = rb;
        pb += items[i].pre_bx * items[i].groups;
        rb += items[i].red_gx * items[i].red_gy;
        const size_t l = sizeof(float) * 32 * (size_t)(items[i].ci_t * items[i].taps + 1);
        if (l > lds) lds = l;
    }
    *presum_blocks = pb; *reduce_blocks = rb; *lds_bytes = lds;
    return MCAV_OK;
}

MCAV_EXPORT int mcav_wgrad_reduce_multi(const mcav_wgrad_reduce_item* items_dev, int n, int presum_blocks, int reduce_blocks, size_t lds_bytes, void* stream) {
    if (!items_dev || n <= 0 || reduce_blocks <= 0 || presum_blocks < 0) return MCAV_E_INVALID;
    hipStream_t s = as_stream(stream);
    if (presum_blocks > 0) timed_launch(wgrad_presum_multi_kernel, dim3(presum_blocks), dim3(256), 0, s, items_dev, n);
    timed_launch(wgrad_reduce_multi_kernel, dim3(reduce_blocks), dim3(256), lds_bytes, s, items_dev, n);
    return launch_status();
}

MCAV_EXPORT int mcav_pack_weights(const float* w_oihw, int Cout, int Cin, int kh, int kw, int transposed, float* packed, int Np, int Kp,
                                  void* stream) {
    if (!w_oihw || !packed || Cout <= 0 || Cin <= 0 || kh <= 0 || kw <= 0 || Np <= 0 || Kp <= 0) return MCAV_E_INVALID;
    if (!transposed && (Np < Cout || Kp < Cin)) return MCAV_E_INVALID;
    if (transposed && (Np < Cin || Kp < Cout)) return MCAV_E_INVALID;
    const int taps = kh * kw, Kstride = kstride_of(taps, Kp);
    const size_t total = (size_t)Np * Kstride;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    pack_weights_kernel<<<blocks, 256, 0, as_stream(stream)>>>(w_oihw, Cout, Cin, taps, transposed, packed, Np, Kp, Kstride);
    return launch_status();
}

// items: device array of PackItem-compatible records {src, dst, Cout, Cin, taps, transposed, Np, Kp, Kstride, first_block}
namespace mcav {
__global__ __launch_bounds__(256) void pack_upmerge_kernel(const float* w, int Cout, int Cin, int C1, int Np, float* out) {
    const size_t total = (size_t)4 * Np * 4 * C1;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) out[e] = upmerge_elem(w, Cout, Cin, C1, Np, e);
}
}  // namespace mcav

MCAV_EXPORT int mcav_pack_weights_upmerge(const float* w_oihw, int Cout, int Cin, int C1, float* packed, int Np, void* stream) {
    if (!w_oihw || !packed || Cout <= 0 || Cin <= 0 || C1 <= 0 || C1 > Cin || Np < Cout) return MCAV_E_INVALID;
    const size_t total = (size_t)4 * Np * 4 * C1;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    pack_upmerge_kernel<<<blocks, 256, 0, as_stream(stream)>>>(w_oihw, Cout, Cin, C1, Np, packed);
    return launch_status();
}

namespace mcav {
// Adjoint counterpart: packed [n = c1][16 taps (u, v)][k = co] (upmerge_adj_elem).
__global__ __launch_bounds__(256) void pack_upmerge_adj_kernel(const float* w, int Cout, int Cin, int C1, int Np, int Kp, float* out) {
    const size_t total = (size_t)Np * 16 * Kp;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) out[e] = upmerge_adj_elem(w, Cout, Cin, C1, Kp, e);
}
}  // namespace mcav

MCAV_EXPORT int mcav_pack_weights_upmerge_adj(const float* w_oihw, int Cout, int Cin, int C1, float* packed, int Np, int Kp, void* stream) {
    if (!w_oihw || !packed || Cout <= 0 || Cin <= 0 || C1 <= 0 || C1 > Cin || Np < C1 || Kp < Cout) return MCAV_E_INVALID;
    const size_t total = (size_t)Np * 16 * Kp;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    pack_upmerge_adj_kernel<<<blocks, 256, 0, as_stream(stream)>>>(w_oihw, Cout, Cin, C1, Np, Kp, packed);
    return launch_status();
}

MCAV_EXPORT int mcav_pack_weights_blocks(int taps, int transposed, int Np, int Kp) {
    if (taps <= 0 || Np <= 0 || Kp <= 0) return 0;
    return pack_blocks(transposed, taps, Np, Kp);
}

MCAV_EXPORT int mcav_pack_weights_multi(const void* items_dev, int nitems, int nblocks, void* stream) {
    if (!items_dev || nitems <= 0 || nblocks <= 0) return MCAV_E_INVALID;
    pack_weights_multi_kernel<<<nblocks, 256, 0, as_stream(stream)>>>(reinterpret_cast<const PackItem*>(items_dev), nitems);
    return launch_status();
}

#if MCAV_DIAG == 4
MCAV_EXPORT int mcav_diag_stamps(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(mcav::g_diag_stamps), sizeof(unsigned long long) * 8) != hipSuccess) return MCAV_E_INVALID;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(mcav::g_diag_stamps), z, sizeof(z)) != hipSuccess) return MCAV_E_INVALID; }
    return MCAV_OK;
}
#endif
