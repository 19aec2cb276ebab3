// Loss stage of the training step on gfx950: fused inverse-warp -> bilinear sample -> L1 photometric ->
// second-order smoothness, forward and backward in one pass (K10 + K11), plus the standalone geometry
// entry points (inverse_warp, reconstruct, project, disp_to_depth, SSIM).
//
// HBM-bound streaming/gather work (52 B/pixel compulsory for the fused kernel): one thread per
// target pixel, 32x8 pixel tiles so that a wavefront covers two 32-pixel row segments (128-B coalesced plane
// reads), the tgt-depth tile (+2 halo) staged in LDS for the smoothness stencil, bilinear taps served from
// L1/L2 (neighbouring pixels sample neighbouring source texels), deterministic two-level reduction
// (wavefront shuffles -> LDS -> per-block slab -> fp64 finalize) for the 2 loss scalars and the 3x12 dP sums.
#include "mcav_common.h"
#include "kernel_timer.h"
#include "warp_math.h"

namespace mcav {

constexpr int TW = 32, TH = 8, HALO = 2, LW = TW + 2 * HALO, LH = TH + 2 * HALO;
constexpr int WL_SUB = 4, WLH = TH * WL_SUB, WL_LH = WLH + 2 * HALO;      // the fused loss kernel: 32 x 32 pixel tiles, 4 pixels per thread --
                                                                          // the 38-value block reduction is paid once per 4 pixels
constexpr int NACC = 38;        // loss_mam, loss_smooth, 3 x dP[12]
constexpr int SLAB = 40;        // floats per block in the slab (padded)

struct PrepConst {
    float Kf[9];
    SampleConst sc;
};

struct WLArgs {
    const float *tgt, *ref0, *ref1, *disp_t, *disp_r0, *poses;
    const float* upstream;
    float *d_disp_t, *d_disp_r0;
    const PrepConst* pc;
    float* slab;
    int B, H, W;
    unsigned flags;
    float tw[3];
    float* dbg;          // test-only instantiation (mcav_warp_loss_debug_taps): [B][3 warps][WL_DBG planes][H][W]
};
constexpr int WL_DBG = 7;      // ix, iy, d loss / d ix, d loss / d iy, res[0..2]

__device__ __forceinline__ void load_K(const void* K, bool f64, int b, double* Kd) {
    if (f64) {
        const double* p = reinterpret_cast<const double*>(K) + (size_t)b * 9;
        for (int i = 0; i < 9; ++i) Kd[i] = p[i];
    } else {
        const float* p = reinterpret_cast<const float*>(K) + (size_t)b * 9;
        for (int i = 0; i < 9; ++i) Kd[i] = (double)p[i];
    }
}

// mode 0: triplet (poses [B,2,6] -> w0 = pose0, w1 = pose1, w2 = inverse(pose0))
// mode 1: single pose [B,6] with `inv`  -> w0
// mode 2: explicit Tcw [B,4,4]          -> w0
// mode 3: intrinsics only (Kinv)
__global__ void pose_prepare_kernel(const float* poses, const void* K, const float* Tcw, int B, int mode, int inv,
                                    int k_f64, PrepConst* out, float* ones) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0 && ones) { ones[0] = 1.0f; ones[1] = 1.0f; reinterpret_cast<unsigned*>(ones)[2] = 0u; }      // [2]: the finalize kernel's ticket
    if (b >= B) return;
    double Kd[9], Ki[9];
    load_K(K, k_f64 != 0, b, Kd);
    invert3x3(Kd, Ki);
    PrepConst pc;
    for (int i = 0; i < 9; ++i) { pc.Kf[i] = (float)Kd[i]; pc.sc.Kinv[i] = (float)Ki[i]; }
    float R[9], t[3];
    if (mode == 0) {
        const float* p = poses + (size_t)b * 12;
        pose_to_Rt(p, false, R, t);      make_P(pc.Kf, R, t, pc.sc.w[0].P);
        pose_to_Rt(p + 6, false, R, t);  make_P(pc.Kf, R, t, pc.sc.w[1].P);
        pose_to_Rt(p, true, R, t);       make_P(pc.Kf, R, t, pc.sc.w[2].P);
    } else if (mode == 1) {
        pose_to_Rt(poses + (size_t)b * 6, inv != 0, R, t);
        make_P(pc.Kf, R, t, pc.sc.w[0].P);
        for (int i = 0; i < 12; ++i) { pc.sc.w[1].P[i] = 0.f; pc.sc.w[2].P[i] = 0.f; }
    } else if (mode == 2) {
        const float* T = Tcw + (size_t)b * 16;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) R[i * 3 + j] = T[i * 4 + j];
            t[i] = T[i * 4 + 3];
        }
        make_P(pc.Kf, R, t, pc.sc.w[0].P);
        for (int i = 0; i < 12; ++i) { pc.sc.w[1].P[i] = 0.f; pc.sc.w[2].P[i] = 0.f; }
    } else {   // mode 3: intrinsics only
        for (int w = 0; w < 3; ++w)
            for (int i = 0; i < 12; ++i) pc.sc.w[w].P[i] = 0.f;
    }
    out[b] = pc;
}

__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// Reduce N per-thread floats over a 256-thread block and store them at dst[0..N).
template <int N>
__device__ __forceinline__ void block_reduce_store(float* acc, float* dst, float (*sred)[SLAB]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < N; ++k) acc[k] = wave_sum(acc[k]);
    if (lane == 0) {                          // one exec-masked region for the N stores (not one per value)
#pragma unroll
        for (int k = 0; k < N; ++k) sred[wave][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < N) dst[threadIdx.x] = (sred[0][threadIdx.x] + sred[1][threadIdx.x]) + (sred[2][threadIdx.x] + sred[3][threadIdx.x]);
}

// Branch-free bilinear gathers: an image (3 planes) is a raw buffer resource, a tap outside the image carries an out-of-range offset and
// reads as zero in hardware.  No exec-mask branch and no value merge sits between a load and its use, so the 36 gathers of a pixel's three
// warps are issued back to back behind one counted wait (the predicated form compiled to ~90 exec-masked regions, each with its own wait).
constexpr unsigned WL_OOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t image_rsrc(const float* img, size_t plane) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img), 0, (unsigned)(3 * plane * sizeof(float)), 0x00020000);
}

struct TapOff { unsigned o[4]; };

__device__ __forceinline__ TapOff tap_offsets(const Tap& t, int W) {
    const int base = (t.y0 * W + t.x0) * 4;
    TapOff f;
    f.o[0] = t.in00 ? (unsigned)base : WL_OOB;
    f.o[1] = t.in01 ? (unsigned)(base + 4) : WL_OOB;
    f.o[2] = t.in10 ? (unsigned)(base + W * 4) : WL_OOB;
    f.o[3] = t.in11 ? (unsigned)(base + W * 4 + 4) : WL_OOB;
    return f;
}

__device__ __forceinline__ void gather_taps(__amdgpu_buffer_rsrc_t rs, const TapOff& f, int plane_bytes, float (*q)[4]) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) q[c][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)f.o[k], c * plane_bytes, 0));
}

template <bool DBG>
__global__ __launch_bounds__(256) void warp_loss_kernel(WLArgs a) {
    const float g0 = a.upstream[0], g1 = a.upstream[1];
    if ((a.flags & MCAV_WL_SKIP_IF_UNIT) && g0 == 1.0f && g1 == 1.0f) return;
    __shared__ float sD[WL_LH][LW + 1];
    __shared__ float sred[4][SLAB];
    const int H = a.H, W = a.W, b = blockIdx.z;
    const int bx0 = blockIdx.x * TW, by0 = blockIdx.y * WLH;
    const size_t plane = (size_t)H * W;
    const bool in_depth = (a.flags & MCAV_WL_INPUT_DEPTH) != 0;
    const float* dt = a.disp_t + (size_t)b * plane;
    for (int i = threadIdx.x; i < WL_LH * LW; i += 256) {
        const int ly = i / LW, lx = i - ly * LW;
        const int gy = by0 - HALO + ly, gx = bx0 - HALO + lx;
        float D = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const float v = dt[(size_t)gy * W + gx];
            D = in_depth ? v : 1.0f / (10.0f * v + 0.01f);
        }
        sD[ly][lx] = D;
    }
    __syncthreads();

    const int tx = threadIdx.x & 31, ty0 = threadIdx.x >> 5;
    const int x = bx0 + tx;
    float acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.f;

    const PrepConst& pc = a.pc[b];
    const float* tgt = a.tgt + (size_t)b * 3 * plane;
    const float* ref0 = a.ref0 + (size_t)b * 3 * plane;
    const float* ref1 = a.ref1 + (size_t)b * 3 * plane;
    const int pb = (int)(plane * sizeof(float));
    const __amdgpu_buffer_rsrc_t rs_t = image_rsrc(tgt, plane), rs_r0 = image_rsrc(ref0, plane), rs_r1 = image_rsrc(ref1, plane);
    const __amdgpu_buffer_rsrc_t rs_dr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.disp_r0 + (size_t)b * plane), 0, (unsigned)pb, 0x00020000);
    const float invN = 1.0f / (float)((size_t)a.B * 3 * plane);
    const float cxx = 1.0f / (float)((size_t)a.B * H * (W - 2));
    const float cyy = 1.0f / (float)((size_t)a.B * (H - 2) * W);
    const float cxy = 2.0f / (float)((size_t)a.B * (H - 1) * (W - 1));   // dxdy and dydx are the same field
    // the target-aligned values of a pixel (3 tgt, 3 ref1, 1 disparity) are fetched ONE PIXEL AHEAD, branch-free (a pixel outside the image
    // reads zeros through the out-of-range offset and is skipped), so their flight overlaps the previous pixel's arithmetic
    auto fetch = [&](int sub, float (&v)[7]) {      // v = tgt[0..2], ref1[0..2], disparity of ref0
        const int y = by0 + sub * TH + ty0;
        const unsigned off = (x < W && y < H) ? (unsigned)((y * W + x) * 4) : WL_OOB;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            v[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_t, (int)off, c * pb, 0));
            v[3 + c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r1, (int)off, c * pb, 0));
        }
        v[6] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_dr, (int)off, 0, 0));
    };
    float pv[WL_SUB][7];
    fetch(0, pv[0]);
#pragma unroll
    for (int sub = 0; sub < WL_SUB; ++sub) {
        const int ty = sub * TH + ty0, y = by0 + ty;
        if (sub + 1 < WL_SUB) fetch(sub + 1, pv[sub + 1]);
        const float tvv[3] = {pv[sub][0], pv[sub][1], pv[sub][2]}, rvv[3] = {pv[sub][3], pv[sub][4], pv[sub][5]};
        const float vr = pv[sub][6];
        if (x < W && y < H) {
            const size_t pix = (size_t)y * W + x;
            const int cy = ty + HALO, cx = tx + HALO;
            const float Dt = sD[cy][cx];
            const float Dr = in_depth ? vr : 1.0f / (10.0f * vr + 0.01f);
            const Ray r = pixel_ray(pc.sc.Kinv, (float)x, (float)y);
            float dDt = 0.f, dDr = 0.f;
            // the three projections first, then all 36 gathers, then the arithmetic (same operations in the same order as warp_pixel)
            // warp 0: ref0 -> tgt view, depth(tgt), pose[0];  warp 1: ref1 -> tgt view, depth(tgt), pose[1];
            // warp 2: tgt -> "ref1 view", depth(ref0), inverse(pose[0])   (reference quirk, losses.py:203-207)
            const Tap t0 = project_pixel(pc.sc.w[0].P, r, Dt, H, W);
            const Tap t1 = project_pixel(pc.sc.w[1].P, r, Dt, H, W);
            const Tap t2 = project_pixel(pc.sc.w[2].P, r, Dr, H, W);
            float q0[3][4], q1[3][4], q2[3][4];
            gather_taps(rs_r0, tap_offsets(t0, W), pb, q0);
            gather_taps(rs_r1, tap_offsets(t1, W), pb, q1);
            gather_taps(rs_t, tap_offsets(t2, W), pb, q2);
            float dbg[DBG ? 3 : 1][WL_DBG];
            warp_pixel_from(q0, tvv, pc.sc.w[0].P, r, t0, H, W, a.tw[0] * invN, g0 * a.tw[0] * invN, acc[0], dDt, acc + 2, DBG ? dbg[0] : nullptr);
            warp_pixel_from(q1, tvv, pc.sc.w[1].P, r, t1, H, W, a.tw[1] * invN, g0 * a.tw[1] * invN, acc[0], dDt, acc + 14, DBG ? dbg[DBG ? 1 : 0] : nullptr);
            warp_pixel_from(q2, rvv, pc.sc.w[2].P, r, t2, H, W, a.tw[2] * invN, g0 * a.tw[2] * invN, acc[0], dDr, acc + 26, DBG ? dbg[DBG ? 2 : 0] : nullptr);
            if constexpr (DBG) {
#pragma unroll
                for (int w = 0; w < 3; ++w)
#pragma unroll
                    for (int k = 0; k < WL_DBG; ++k) a.dbg[(((size_t)b * 3 + w) * WL_DBG + k) * plane + pix] = dbg[w][k];
            }
            if (!(a.flags & MCAV_WL_NO_SMOOTH)) {
                float gs = 0.f, ls = 0.f;
                smooth_terms_sel([&](int dy, int dx) { return sD[cy + dy][cx + dx]; }, x, y, H, W, cxx, cyy, cxy, ls, gs);
                acc[1] += ls;
                dDt += g1 * gs;
            }
            a.d_disp_t[(size_t)b * plane + pix] = in_depth ? dDt : dDt * (-10.0f * Dt * Dt);
            a.d_disp_r0[(size_t)b * plane + pix] = in_depth ? dDr : dDr * (-10.0f * Dr * Dr);
        }
    }
    const int nblk = gridDim.x * gridDim.y;
    const int blk = blockIdx.y * gridDim.x + blockIdx.x;
    block_reduce_store<NACC>(acc, a.slab + ((size_t)b * nblk + blk) * SLAB, sred);
}

// ---------------------------------------------------------------------------------------------- SSIM + L1 photometric (MCAV_WL_SSIM)
// The north_star's photometric mix, 0.85 * SSIM distance + 0.15 * L1 (weights of losses.py:77, SSIM of losses.py:12-54), fused with
// the warp, forward and backward in one pass.  Per 32 x 32 tile and per warp the three warped channel planes and the three target
// planes of the tile + 2 halo are staged in LDS (values at padded positions -1 / H / W are the warp at the REFLECTED pixel, which is
// what ReflectionPad2d of the warped image holds); per channel the 3x3 window statistics are evaluated on tile + 1 halo and turned
// into three coefficient fields -- dS(p)/dx(q) = a(p) + b(p) x(q) + c(p) y(q) for every q in p's window -- so that the gradient at a
// pixel is a second 3x3 gather of (a, b, c) (plus the windows it enters through the reflection), no atomics.
constexpr int SS_P = WLH + 2;      // statistics region: tile + 1 halo

struct SsimPoint { float S, a, b, c; };

// x: 3x3 window of the warped image, y: of the target, row-major.  S = clamp((1 - SSIM) / 2, 0, 1); (a, b, c): see above.
__device__ __forceinline__ SsimPoint ssim_point(const float* x, const float* y) {
    // no fma contraction: SSIM(x, x) must cancel exactly, as in ssim_kernel and in the reference
#pragma clang fp contract(off)
    float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) { sx += x[i]; sy += y[i]; sxx += x[i] * x[i]; syy += y[i] * y[i]; sxy += x[i] * y[i]; }
    const float C1 = 1e-4f, C2 = 9e-4f, k = 1.0f / 9.0f;
    const float mx = sx * k, my = sy * k;
    const float mxy = mx * my, mxx = mx * mx, myy = my * my;
    const float vx = sxx * k - mxx, vy = syy * k - myy, vxy = sxy * k - mxy;
    const float A1 = 2.f * mxy + C1, A2 = 2.f * vxy + C2, B1 = mxx + myy + C1, B2 = vx + vy + C2;
    const float den = B1 * B2;
    const float ssim = (A1 * A2) / den;
    const float v = (1.0f - ssim) / 2.0f;
    SsimPoint o;
    o.S = fminf(fmaxf(v, 0.0f), 1.0f);
    const float m = (v >= 0.0f && v <= 1.0f) ? -0.5f * (2.0f / 9.0f) / den : 0.0f;       // d clamp((1 - ssim) / 2) / d ssim, times 2 / (9 den)
    o.a = m * (my * (A2 - A1) - ssim * mx * (B2 - B1));
    o.b = m * (-ssim * B1);
    o.c = m * A1;
    return o;
}

template <bool DBG>
__global__ __launch_bounds__(256) void warp_loss_ssim_kernel(WLArgs a) {
    const float g0 = a.upstream[0], g1 = a.upstream[1];
    if ((a.flags & MCAV_WL_SKIP_IF_UNIT) && g0 == 1.0f && g1 == 1.0f) return;
    __shared__ float sD[WL_LH][LW + 1];
    __shared__ float sX[3][WL_LH][LW + 1];
    __shared__ float sT[3][WL_LH][LW + 1];
    __shared__ float sC[3][SS_P][SS_P + 1];
    __shared__ float sred[4][SLAB];
    const int H = a.H, W = a.W, b = blockIdx.z;
    const int bx0 = blockIdx.x * TW, by0 = blockIdx.y * WLH;
    const size_t plane = (size_t)H * W;
    const bool in_depth = (a.flags & MCAV_WL_INPUT_DEPTH) != 0;
    const float* dt = a.disp_t + (size_t)b * plane;
    const float* dr = a.disp_r0 + (size_t)b * plane;
    for (int i = threadIdx.x; i < WL_LH * LW; i += 256) {
        const int ly = i / LW, lx = i - ly * LW;
        const int gy = by0 - HALO + ly, gx = bx0 - HALO + lx;
        float D = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const float v = dt[(size_t)gy * W + gx];
            D = in_depth ? v : 1.0f / (10.0f * v + 0.01f);
        }
        sD[ly][lx] = D;
    }
    const PrepConst& pc = a.pc[b];
    const float* img_t = a.tgt + (size_t)b * 3 * plane;
    const float* img_r0 = a.ref0 + (size_t)b * 3 * plane;
    const float* img_r1 = a.ref1 + (size_t)b * 3 * plane;
    const int tx = threadIdx.x & 31, ty0 = threadIdx.x >> 5;
    const int x = bx0 + tx;
    const float invN = 1.0f / (float)((size_t)a.B * 3 * plane);
    const float WS = 0.85f, WL1 = 0.15f;              // losses.py:77
    constexpr int NONE = -(1 << 30);
    float acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.f;
    float dDt[WL_SUB], dDr[WL_SUB], Dr[WL_SUB];
#pragma unroll
    for (int sub = 0; sub < WL_SUB; ++sub) {
        dDt[sub] = 0.f; dDr[sub] = 0.f; Dr[sub] = 0.f;
        const int y = by0 + sub * TH + ty0;
        if (x < W && y < H) {
            const float vr = dr[(size_t)y * W + x];
            Dr[sub] = in_depth ? vr : 1.0f / (10.0f * vr + 0.01f);
        }
    }

#pragma unroll
    for (int w = 0; w < 3; ++w) {
        const float* src = w == 0 ? img_r0 : (w == 1 ? img_r1 : img_t);
        const float* tar = w == 2 ? img_r1 : img_t;
        const float* P = pc.sc.w[w].P;
        const float lw = a.tw[w] * invN, gw = g0 * lw;
        __syncthreads();                       // sD is filled (w == 0) / the previous warp's readers are done
        // ---- phase 1: warped and target planes on tile + 2 halo
        for (int i = threadIdx.x; i < WL_LH * LW; i += 256) {
            const int ly = i / LW, lx = i - ly * LW;
            const int gy = by0 - HALO + ly, gx = bx0 - HALO + lx;
            float xv[3] = {0.f, 0.f, 0.f}, tv[3] = {0.f, 0.f, 0.f};
            if (gy >= -1 && gy <= H && gx >= -1 && gx <= W) {
                const int ry = reflect1(gy, H), rx = reflect1(gx, W);
                float D;
                if (w < 2) D = sD[ry - by0 + HALO][rx - bx0 + HALO];
                else {
                    const float v = dr[(size_t)ry * W + rx];
                    D = in_depth ? v : 1.0f / (10.0f * v + 0.01f);
                }
                const Ray r = pixel_ray(pc.sc.Kinv, (float)rx, (float)ry);
                const Tap t = project_pixel(P, r, D, H, W);
#pragma unroll
                for (int c = 0; c < 3; ++c) xv[c] = bilinear(src + c * plane, W, t).v;
                if (w != 1) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) tv[c] = tar[c * plane + (size_t)ry * W + rx];
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                sX[c][ly][lx] = xv[c];
                if (w != 1) sT[c][ly][lx] = tv[c];         // warps 0 and 1 share the target
            }
        }
        __syncthreads();
        float gp0[WL_SUB], gp1[WL_SUB], gp2[WL_SUB];        // d loss / d warped value at the thread's own pixels, per channel
#pragma unroll 1
        for (int c = 0; c < 3; ++c) {
            // ---- phase 2: window statistics -> S and the gradient coefficient fields on tile + 1 halo
            for (int i = threadIdx.x; i < SS_P * SS_P; i += 256) {
                const int py = i / SS_P, px = i - py * SS_P;
                const int gy = by0 - 1 + py, gx = bx0 - 1 + px;
                SsimPoint o;
                o.S = 0.f; o.a = 0.f; o.b = 0.f; o.c = 0.f;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
                    float xw[9], yw[9];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) { xw[dy * 3 + dx] = sX[c][py + dy][px + dx]; yw[dy * 3 + dx] = sT[c][py + dy][px + dx]; }
                    o = ssim_point(xw, yw);
                    if (py >= 1 && py <= WLH && px >= 1 && px <= TW) acc[0] += lw * (WS * o.S + WL1 * fabsf(xw[4] - yw[4]));
                }
                sC[0][py][px] = o.a; sC[1][py][px] = o.b; sC[2][py][px] = o.c;
            }
            __syncthreads();
            // ---- phase 3: gather the coefficient fields of every window the thread's own pixels take part in
#pragma unroll
            for (int sub = 0; sub < WL_SUB; ++sub) {
                const int ty = sub * TH + ty0, y = by0 + ty;
                if (c == 0) gp0[sub] = 0.f; else if (c == 1) gp1[sub] = 0.f; else gp2[sub] = 0.f;
                if (!(x < W && y < H)) continue;
                float SA = 0.f, SB = 0.f, SC = 0.f;
                // the pixel's own 3x3 neighbourhood: always inside the statistics region, zero outside the image
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) { SA += sC[0][ty + dy][tx + dx]; SB += sC[1][ty + dy][tx + dx]; SC += sC[2][ty + dy][tx + dx]; }
                if (y == 1 || y == H - 2 || x == 1 || x == W - 2) {
                    // one pixel in from the border: the pixel is also the reflection at padded row -1 / H or column -1 / W
                    for (int yi = 0; yi < 3; ++yi) {
                        const int yc = yi == 0 ? y : (yi == 1 ? (y == 1 ? -1 : NONE) : (y == H - 2 ? H : NONE));
                        if (yc == NONE) continue;
                        for (int xi = (yi == 0 ? 1 : 0); xi < 3; ++xi) {
                            const int xc = xi == 0 ? x : (xi == 1 ? (x == 1 ? -1 : NONE) : (x == W - 2 ? W : NONE));
                            if (xc == NONE) continue;
                            for (int dy = -1; dy <= 1; ++dy) {
                                const int qy = yc + dy;
                                if (qy < 0 || qy >= H) continue;
                                for (int dx = -1; dx <= 1; ++dx) {
                                    const int qx = xc + dx;
                                    if (qx < 0 || qx >= W) continue;
                                    SA += sC[0][qy - by0 + 1][qx - bx0 + 1];
                                    SB += sC[1][qy - by0 + 1][qx - bx0 + 1];
                                    SC += sC[2][qy - by0 + 1][qx - bx0 + 1];
                                }
                            }
                        }
                    }
                }
                const float xq = sX[c][ty + HALO][tx + HALO], tq = sT[c][ty + HALO][tx + HALO];
                const float gv = gw * (WL1 * sgn(xq - tq) + WS * (SA + xq * SB + tq * SC));
                if (c == 0) gp0[sub] = gv; else if (c == 1) gp1[sub] = gv; else gp2[sub] = gv;
            }
            __syncthreads();                   // sC is rewritten by the next channel
        }
        // ---- phase 4: chain through the bilinear sample to the sampling position, the depth and P
#pragma unroll
        for (int sub = 0; sub < WL_SUB; ++sub) {
            const int ty = sub * TH + ty0, y = by0 + ty;
            if (!(x < W && y < H)) continue;
            const Ray r = pixel_ray(pc.sc.Kinv, (float)x, (float)y);
            const Tap t = project_pixel(P, r, w < 2 ? sD[ty + HALO][tx + HALO] : Dr[sub], H, W);
            float gix = 0.f, giy = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const Sample sm = bilinear(src + c * plane, W, t);
                const float gv = c == 0 ? gp0[sub] : (c == 1 ? gp1[sub] : gp2[sub]);
                gix += gv * sm.dvdx;
                giy += gv * sm.dvdy;
            }
            const float d = backproject_grad(P, r, t, gix, giy, H, W, acc + 2 + 12 * w);
            if (w < 2) dDt[sub] += d; else dDr[sub] += d;
            if constexpr (DBG) {      // (the residual planes of the dump stay zero: the mix's value-level kinks are judged on the oracle's margins)
                const float v[WL_DBG] = {t.ix, t.iy, gix, giy, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < WL_DBG; ++k) a.dbg[(((size_t)b * 3 + w) * WL_DBG + k) * plane + (size_t)y * W + x] = v[k];
            }
        }
    }

#pragma unroll
    for (int sub = 0; sub < WL_SUB; ++sub) {
        const int ty = sub * TH + ty0, y = by0 + ty;
        if (!(x < W && y < H)) continue;
        const int cy = ty + HALO, cx = tx + HALO;
        const size_t pix = (size_t)y * W + x;
        const float Dt = sD[cy][cx];
        float d = dDt[sub];
        if (!(a.flags & MCAV_WL_NO_SMOOTH)) {
            const float cxx = 1.0f / (float)((size_t)a.B * H * (W - 2));
            const float cyy = 1.0f / (float)((size_t)a.B * (H - 2) * W);
            const float cxy = 2.0f / (float)((size_t)a.B * (H - 1) * (W - 1));
            float gs = 0.f, ls = 0.f;
            smooth_terms([&](int dy, int dx) { return sD[cy + dy][cx + dx]; }, x, y, H, W, cxx, cyy, cxy, ls, gs);
            acc[1] += ls;
            d += g1 * gs;
        }
        a.d_disp_t[(size_t)b * plane + pix] = in_depth ? d : d * (-10.0f * Dt * Dt);
        a.d_disp_r0[(size_t)b * plane + pix] = in_depth ? dDr[sub] : dDr[sub] * (-10.0f * Dr[sub] * Dr[sub]);
    }
    const int nblk = gridDim.x * gridDim.y;
    const int blk = blockIdx.y * gridDim.x + blockIdx.x;
    block_reduce_store<NACC>(acc, a.slab + ((size_t)b * nblk + blk) * SLAB, sred);
}

// One block per sample: sum the slab in fp64 (26 slices of the tile list per accumulator: short load chains), turn dP into pose
// gradients, emit per-sample loss sums; the LAST block to finish (ticket in the workspace, zeroed by pose_prepare_kernel) adds the
// samples up in index order -- one launch less than a separate total kernel, same fixed summation order.
constexpr int FIN_PARTS = 26;       // 26 x 38 = 988 threads

__global__ __launch_bounds__(1024) void warp_loss_finalize_kernel(const float* slab, int nblk, const PrepConst* pcs, const float* poses,
                                                                  const float* upstream, unsigned flags, float* d_poses, double* sample_loss,
                                                                  unsigned* ticket, int B, float* losses) {
    if ((flags & MCAV_WL_SKIP_IF_UNIT) && upstream[0] == 1.0f && upstream[1] == 1.0f) return;
    __shared__ double s[FIN_PARTS][NACC];
    __shared__ bool last;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < FIN_PARTS * NACC) {
        const int part = tid / NACC, k = tid - part * NACC;
        double sum = 0.0;
        for (int blk = part; blk < nblk; blk += FIN_PARTS) sum += (double)slab[((size_t)b * nblk + blk) * SLAB + k];
        s[part][k] = sum;
    }
    __syncthreads();
    if (tid < NACC) {
        double t = 0.0;
        for (int p = 0; p < FIN_PARTS; ++p) t += s[p][tid];
        s[0][tid] = t;
    }
    __syncthreads();
    if (tid == 0) {
        const PrepConst& pc = pcs[b];
        const float* p = poses + (size_t)b * 12;
        double g0[6], g1[6], g2[6];
        pose_grad_from_dP(&s[0][2], pc.Kf, p, false, g0);
        pose_grad_from_dP(&s[0][14], pc.Kf, p + 6, false, g1);
        pose_grad_from_dP(&s[0][26], pc.Kf, p, true, g2);
        for (int i = 0; i < 6; ++i) {
            d_poses[(size_t)b * 12 + i] = (float)(g0[i] + g2[i]);
            d_poses[(size_t)b * 12 + 6 + i] = (float)g1[i];
        }
        sample_loss[b * 2 + 0] = s[0][0];
        sample_loss[b * 2 + 1] = s[0][1];
        __threadfence();                                        // this sample's sums are visible before the ticket is taken
        last = atomicAdd(ticket, 1u) == (unsigned)(B - 1);
        if (last) {
            __threadfence();
            double a = 0.0, sm = 0.0;
            const volatile double* sl = sample_loss;            // written by other workgroups: read from L2
            for (int i = 0; i < B; ++i) { a += sl[i * 2]; sm += sl[i * 2 + 1]; }
            losses[0] = (float)a;
            losses[1] = (float)sm;
            *ticket = 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------- standalone warp
__global__ __launch_bounds__(256) void inverse_warp_fwd_kernel(const float* img, const float* depth, const PrepConst* pcs, int B, int H, int W, float* out) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * TW + (threadIdx.x & 31), y = blockIdx.y * TH + (threadIdx.x >> 5);
    if (x >= W || y >= H) return;
    const size_t plane = (size_t)H * W, pix = (size_t)y * W + x;
    const PrepConst& pc = pcs[b];
    const Ray r = pixel_ray(pc.sc.Kinv, (float)x, (float)y);
    const Tap t = project_pixel(pc.sc.w[0].P, r, depth[(size_t)b * plane + pix], H, W);
#pragma unroll
    for (int c = 0; c < 3; ++c) out[((size_t)b * 3 + c) * plane + pix] = bilinear(img + ((size_t)b * 3 + c) * plane, W, t).v;
}

__global__ __launch_bounds__(256) void inverse_warp_bwd_kernel(const float* img, const float* depth, const PrepConst* pcs, const float* go,
                                                               int B, int H, int W, float* d_depth, float* slab) {
    __shared__ float sred[4][SLAB];
    const int b = blockIdx.z;
    const int x = blockIdx.x * TW + (threadIdx.x & 31), y = blockIdx.y * TH + (threadIdx.x >> 5);
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    if (x < W && y < H) {
        const size_t plane = (size_t)H * W, pix = (size_t)y * W + x;
        const PrepConst& pc = pcs[b];
        const Ray r = pixel_ray(pc.sc.Kinv, (float)x, (float)y);
        const Tap t = project_pixel(pc.sc.w[0].P, r, depth[(size_t)b * plane + pix], H, W);
        float gix = 0.f, giy = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const Sample s = bilinear(img + ((size_t)b * 3 + c) * plane, W, t);
            const float g = go[((size_t)b * 3 + c) * plane + pix];
            gix += g * s.dvdx;
            giy += g * s.dvdy;
        }
        d_depth[(size_t)b * plane + pix] = backproject_grad(pc.sc.w[0].P, r, t, gix, giy, H, W, acc);
    }
    const int nblk = gridDim.x * gridDim.y;
    const int blk = blockIdx.y * gridDim.x + blockIdx.x;
    block_reduce_store<12>(acc, slab + ((size_t)b * nblk + blk) * SLAB, sred);
}

__global__ __launch_bounds__(64) void inverse_warp_bwd_finalize_kernel(const float* slab, int nblk, const PrepConst* pcs, const float* pose, int inv, float* d_pose) {
    __shared__ double s[4][12];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < 48) {
        const int part = tid / 12, k = tid - part * 12;
        double sum = 0.0;
        for (int blk = part; blk < nblk; blk += 4) sum += (double)slab[((size_t)b * nblk + blk) * SLAB + k];
        s[part][k] = sum;
    }
    __syncthreads();
    if (tid == 0) {
        double dP[12], g[6];
        for (int k = 0; k < 12; ++k) dP[k] = (s[0][k] + s[1][k]) + (s[2][k] + s[3][k]);
        pose_grad_from_dP(dP, pcs[b].Kf, pose + (size_t)b * 6, inv != 0, g);
        for (int i = 0; i < 6; ++i) d_pose[(size_t)b * 6 + i] = (float)g[i];
    }
}

__global__ __launch_bounds__(256) void reconstruct_kernel(const float* depth, const PrepConst* pcs, int B, int H, int W, float* pts) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * TW + (threadIdx.x & 31), y = blockIdx.y * TH + (threadIdx.x >> 5);
    if (x >= W || y >= H) return;
    const size_t plane = (size_t)H * W, pix = (size_t)y * W + x;
    const Ray r = pixel_ray(pcs[b].sc.Kinv, (float)x, (float)y);
    const float D = depth[(size_t)b * plane + pix];
    pts[((size_t)b * 3 + 0) * plane + pix] = r.r0 * D;
    pts[((size_t)b * 3 + 1) * plane + pix] = r.r1 * D;
    pts[((size_t)b * 3 + 2) * plane + pix] = r.r2 * D;
}

__global__ __launch_bounds__(256) void project_kernel(const float* pts, const void* K, int k_f64, const float* Tcw, int B, int H, int W, float* grid) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * TW + (threadIdx.x & 31), y = blockIdx.y * TH + (threadIdx.x >> 5);
    if (x >= W || y >= H) return;
    double Kd[9];
    load_K(K, k_f64 != 0, b, Kd);
    float Kf[9], R[9], t[3], P[12];
    for (int i = 0; i < 9; ++i) Kf[i] = (float)Kd[i];
    const float* T = Tcw + (size_t)b * 16;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) R[i * 3 + j] = T[i * 4 + j];
        t[i] = T[i * 4 + 3];
    }
    make_P(Kf, R, t, P);
    const size_t plane = (size_t)H * W, pix = (size_t)y * W + x;
    const float X0 = pts[((size_t)b * 3 + 0) * plane + pix], X1 = pts[((size_t)b * 3 + 1) * plane + pix], X2 = pts[((size_t)b * 3 + 2) * plane + pix];
    const float c0 = P[0] * X0 + P[1] * X1 + P[2] * X2 + P[3];
    const float c1 = P[4] * X0 + P[5] * X1 + P[6] * X2 + P[7];
    const float c2 = P[8] * X0 + P[9] * X1 + P[10] * X2 + P[11];
    const float z = c2 + 1e-5f;
    grid[((size_t)b * plane + pix) * 2 + 0] = ((c0 / z) / (float)(W - 1) - 0.5f) * 2.0f;
    grid[((size_t)b * plane + pix) * 2 + 1] = ((c1 / z) / (float)(H - 1) - 0.5f) * 2.0f;
}

__global__ void pose_to_matrix_kernel(const float* pose, int B, int invert, float* T) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float R[9], t[3];
    pose_to_Rt(pose + (size_t)b * 6, invert != 0, R, t);
    float* o = T + (size_t)b * 16;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) o[i * 4 + j] = R[i * 3 + j];
        o[i * 4 + 3] = t[i];
    }
    o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
}

__global__ void invert_pose_kernel(const float* T, int B, float* Ti) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* s = T + (size_t)b * 16;
    float* o = Ti + (size_t)b * 16;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) o[i * 4 + j] = s[j * 4 + i];
        o[i * 4 + 3] = (-1.0f * s[0 * 4 + i]) * s[3] + (-1.0f * s[1 * 4 + i]) * s[7] + (-1.0f * s[2 * 4 + i]) * s[11];
    }
    o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
}

__global__ void disp_to_depth_kernel(const float* disp, float* depth, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        depth[i] = 1.0f / (10.0f * disp[i] + 0.01f);
}

__global__ void disp_to_depth_bwd_kernel(const float* disp, const float* dD, float* dd, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float D = 1.0f / (10.0f * disp[i] + 0.01f);
        dd[i] = dD[i] * (-10.0f * D * D);
    }
}


__global__ __launch_bounds__(256) void ssim_kernel(const float* xs, const float* ys, int N, int H, int W, float C1, float C2, float* out) {
    // no fma contraction in this kernel: SSIM(x, x) must cancel exactly (num == den), as it does in the reference
#pragma clang fp contract(off)
    const int n = blockIdx.z;
    const int x = blockIdx.x * TW + (threadIdx.x & 31), y = blockIdx.y * TH + (threadIdx.x >> 5);
    if (x >= W || y >= H) return;
    const float* px = xs + (size_t)n * H * W;
    const float* py = ys + (size_t)n * H * W;
    float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = reflect1(y + dy, H);
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = reflect1(x + dx, W);
            const float a = px[(size_t)yy * W + xx], b = py[(size_t)yy * W + xx];
            sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
        }
    }
    const float k = 1.0f / 9.0f;
    const float mx = sx * k, my = sy * k;
    const float mxy = mx * my, mxx = mx * mx, myy = my * my;
    const float vx = sxx * k - mxx, vy = syy * k - myy, vxy = sxy * k - mxy;
    const float num = (2.f * mxy + C1) * (2.f * vxy + C2);
    const float den = (mxx + myy + C1) * (vx + vy + C2);
    float v = (1.0f - num / den) / 2.0f;
    v = fminf(fmaxf(v, 0.0f), 1.0f);
    out[(size_t)n * H * W + (size_t)y * W + x] = v;
}

// ---------------------------------------------------------------------------------------------- smoothness, one scale
__global__ __launch_bounds__(256) void smooth_kernel(const float* depth, int B, int H, int W, float weight, const float* upstream,
                                                     float* d_depth, int accumulate, float* slab) {
    __shared__ float sD[LH][LW + 1];
    __shared__ float sred[4][SLAB];
    const int b = blockIdx.z;
    const int bx0 = blockIdx.x * TW, by0 = blockIdx.y * TH;
    const size_t plane = (size_t)H * W;
    const float* dp = depth + (size_t)b * plane;
    for (int i = threadIdx.x; i < LH * LW; i += 256) {
        const int ly = i / LW, lx = i - ly * LW;
        const int gy = by0 - HALO + ly, gx = bx0 - HALO + lx;
        sD[ly][lx] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? dp[(size_t)gy * W + gx] : 0.f;
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int x = bx0 + tx, y = by0 + ty;
    float acc[1] = {0.f};
    if (x < W && y < H) {
        const int cy = ty + HALO, cx = tx + HALO;
        const float cxx = weight / (float)((size_t)B * H * (W - 2));
        const float cyy = weight / (float)((size_t)B * (H - 2) * W);
        const float cxy = 2.0f * weight / (float)((size_t)B * (H - 1) * (W - 1));
        float gs = 0.f, ls = 0.f;
        smooth_terms([&](int dy, int dx) { return sD[cy + dy][cx + dx]; }, x, y, H, W, cxx, cyy, cxy, ls, gs);
        acc[0] = ls;
        const float g = upstream ? upstream[0] : 1.0f;
        const size_t o = (size_t)b * plane + (size_t)y * W + x;
        d_depth[o] = (accumulate ? d_depth[o] : 0.f) + g * gs;
    }
    const int nblk = gridDim.x * gridDim.y * gridDim.z;
    const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    (void)nblk;
    block_reduce_store<1>(acc, slab + (size_t)blk * SLAB, sred);
}

__global__ __launch_bounds__(256) void smooth_finalize_kernel(const float* slab, int nblk, float* loss_accum) {
    __shared__ double s[256];
    double sum = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) sum += (double)slab[(size_t)i * SLAB];
    s[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_accum[0] += (float)s[0];
}

struct WsLayout {
    size_t pc_off, ones_off, slab_off, sl_off, total;
    int nblk;
};

inline WsLayout ws_layout(int B, int H, int W) {
    WsLayout l;
    l.nblk = ((W + TW - 1) / TW) * ((H + TH - 1) / TH);
    size_t o = 0;
    l.pc_off = o;   o = align_up(o + sizeof(PrepConst) * (size_t)B, 256);
    l.ones_off = o; o = align_up(o + 16, 256);
    l.slab_off = o; o = align_up(o + sizeof(float) * SLAB * (size_t)B * l.nblk, 256);
    l.sl_off = o;   o = align_up(o + sizeof(double) * 2 * (size_t)B, 256);
    l.total = o;
    return l;
}

inline dim3 pix_grid(int B, int H, int W) { return dim3((W + TW - 1) / TW, (H + TH - 1) / TH, B); }

}  // namespace mcav

using namespace mcav;

MCAV_EXPORT int mcav_abi_version(void) { return 1; }

MCAV_EXPORT size_t mcav_warp_loss_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return ws_layout(B, H, W).total;
}

static int warp_loss_launch(const float* tgt, const float* ref0, const float* ref1, const float* disp_t, const float* disp_r0,
                            const float* poses, const void* K, int B, int H, int W, unsigned flags, const float* upstream,
                            const float* term_weights, float* losses, float* d_disp_t, float* d_disp_r0, float* d_poses,
                            void* workspace, size_t workspace_bytes, void* stream, float* dbg) {
    if (!tgt || !ref0 || !ref1 || !disp_t || !disp_r0 || !poses || !K || !losses || !d_disp_t || !d_disp_r0 || !d_poses || !workspace)
        return MCAV_E_INVALID;
    if (B <= 0 || H < 3 || W < 3 || B > 65535) return MCAV_E_INVALID;
    const WsLayout l = ws_layout(B, H, W);
    if (workspace_bytes < l.total) return MCAV_E_WORKSPACE;
    char* ws = reinterpret_cast<char*>(workspace);
    PrepConst* pc = reinterpret_cast<PrepConst*>(ws + l.pc_off);
    float* ones = reinterpret_cast<float*>(ws + l.ones_off);
    float* slab = reinterpret_cast<float*>(ws + l.slab_off);
    double* sl = reinterpret_cast<double*>(ws + l.sl_off);
    hipStream_t s = as_stream(stream);
    const float* up = upstream ? upstream : ones;
    timed_launch(pose_prepare_kernel, dim3((B + 63) / 64), dim3(64), 0, s, poses, K, (const float*)nullptr, B, 0, 0, (flags & MCAV_WL_K_F64) ? 1 : 0, pc, ones);
    WLArgs a;
    a.tgt = tgt; a.ref0 = ref0; a.ref1 = ref1; a.disp_t = disp_t; a.disp_r0 = disp_r0; a.poses = poses;
    a.upstream = up; a.d_disp_t = d_disp_t; a.d_disp_r0 = d_disp_r0; a.pc = pc; a.slab = slab;
    a.B = B; a.H = H; a.W = W; a.flags = flags;
    a.tw[0] = term_weights ? term_weights[0] : 0.25f;
    a.tw[1] = term_weights ? term_weights[1] : 0.25f;
    a.tw[2] = term_weights ? term_weights[2] : 0.5f;
    a.dbg = dbg;
    const dim3 wl_grid((W + TW - 1) / TW, (H + WLH - 1) / WLH, B);
    if (flags & MCAV_WL_SSIM) {
        if (dbg) timed_launch(warp_loss_ssim_kernel<true>, wl_grid, dim3(256), 0, s, a);
        else timed_launch(warp_loss_ssim_kernel<false>, wl_grid, dim3(256), 0, s, a);
    } else if (dbg) timed_launch(warp_loss_kernel<true>, wl_grid, dim3(256), 0, s, a);
    else timed_launch(warp_loss_kernel<false>, wl_grid, dim3(256), 0, s, a);
    timed_launch(warp_loss_finalize_kernel, dim3(B), dim3(1024), 0, s, (const float*)slab, (int)(wl_grid.x * wl_grid.y), (const PrepConst*)pc, poses, up, flags,
                 d_poses, sl, reinterpret_cast<unsigned*>(ones) + 2, B, losses);
    return launch_status();
}

MCAV_EXPORT int mcav_warp_loss_fwd_bwd(const float* tgt, const float* ref0, const float* ref1, const float* disp_t, const float* disp_r0,
                                       const float* poses, const void* K, int B, int H, int W, unsigned flags, const float* upstream,
                                       const float* term_weights, float* losses, float* d_disp_t, float* d_disp_r0, float* d_poses,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    return warp_loss_launch(tgt, ref0, ref1, disp_t, disp_r0, poses, K, B, H, W, flags, upstream, term_weights, losses, d_disp_t, d_disp_r0, d_poses,
                            workspace, workspace_bytes, stream, nullptr);
}

// Diagnostic twin of mcav_warp_loss_fwd_bwd: the same kernel bodies instantiated with their per-pixel dump on.
MCAV_EXPORT int mcav_warp_loss_debug_taps(const float* tgt, const float* ref0, const float* ref1, const float* disp_t, const float* disp_r0,
                                          const float* poses, const void* K, int B, int H, int W, unsigned flags, const float* term_weights,
                                          float* losses, float* d_disp_t, float* d_disp_r0, float* d_poses, void* workspace,
                                          size_t workspace_bytes, float* taps, size_t taps_floats, void* stream) {
    if (!taps || (flags & MCAV_WL_SKIP_IF_UNIT) || B <= 0 || H <= 0 || W <= 0) return MCAV_E_INVALID;
    if (taps_floats < (size_t)B * 3 * WL_DBG * H * W) return MCAV_E_WORKSPACE;
    return warp_loss_launch(tgt, ref0, ref1, disp_t, disp_r0, poses, K, B, H, W, flags, nullptr, term_weights, losses, d_disp_t, d_disp_r0, d_poses,
                            workspace, workspace_bytes, stream, taps);
}

MCAV_EXPORT int mcav_inverse_warp_fwd(const float* img, const float* depth, const float* pose, const void* K, int B, int H, int W, int pose_inv,
                                      unsigned flags, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!img || !depth || !pose || !K || !out || !workspace || B <= 0 || H < 2 || W < 2 || B > 65535) return MCAV_E_INVALID;
    const WsLayout l = ws_layout(B, H, W);
    if (workspace_bytes < l.total) return MCAV_E_WORKSPACE;
    PrepConst* pc = reinterpret_cast<PrepConst*>(reinterpret_cast<char*>(workspace) + l.pc_off);
    hipStream_t s = as_stream(stream);
    pose_prepare_kernel<<<(B + 63) / 64, 64, 0, s>>>(pose, K, nullptr, B, 1, pose_inv, (flags & MCAV_WL_K_F64) ? 1 : 0, pc, nullptr);
    inverse_warp_fwd_kernel<<<pix_grid(B, H, W), 256, 0, s>>>(img, depth, pc, B, H, W, out);
    return launch_status();
}

MCAV_EXPORT int mcav_inverse_warp_bwd(const float* img, const float* depth, const float* pose, const void* K, const float* grad_out, int B, int H,
                                      int W, int pose_inv, unsigned flags, float* d_depth, float* d_pose, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    if (!img || !depth || !pose || !K || !grad_out || !d_depth || !d_pose || !workspace || B <= 0 || H < 2 || W < 2 || B > 65535)
        return MCAV_E_INVALID;
    const WsLayout l = ws_layout(B, H, W);
    if (workspace_bytes < l.total) return MCAV_E_WORKSPACE;
    char* ws = reinterpret_cast<char*>(workspace);
    PrepConst* pc = reinterpret_cast<PrepConst*>(ws + l.pc_off);
    float* slab = reinterpret_cast<float*>(ws + l.slab_off);
    hipStream_t s = as_stream(stream);
    pose_prepare_kernel<<<(B + 63) / 64, 64, 0, s>>>(pose, K, nullptr, B, 1, pose_inv, (flags & MCAV_WL_K_F64) ? 1 : 0, pc, nullptr);
    inverse_warp_bwd_kernel<<<pix_grid(B, H, W), 256, 0, s>>>(img, depth, pc, grad_out, B, H, W, d_depth, slab);
    inverse_warp_bwd_finalize_kernel<<<B, 64, 0, s>>>(slab, l.nblk, pc, pose, pose_inv, d_pose);
    return launch_status();
}

MCAV_EXPORT int mcav_reconstruct(const float* depth, const void* K, int B, int H, int W, unsigned flags, float* points, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    if (!depth || !K || !points || !workspace || B <= 0 || H <= 0 || W <= 0 || B > 65535) return MCAV_E_INVALID;
    const WsLayout l = ws_layout(B, H, W);
    if (workspace_bytes < l.total) return MCAV_E_WORKSPACE;
    PrepConst* pc = reinterpret_cast<PrepConst*>(reinterpret_cast<char*>(workspace) + l.pc_off);
    hipStream_t s = as_stream(stream);
    pose_prepare_kernel<<<(B + 63) / 64, 64, 0, s>>>(nullptr, K, nullptr, B, 3, 0, (flags & MCAV_WL_K_F64) ? 1 : 0, pc, nullptr);
    reconstruct_kernel<<<pix_grid(B, H, W), 256, 0, s>>>(depth, pc, B, H, W, points);
    return launch_status();
}

MCAV_EXPORT int mcav_project(const float* points, const void* K, const float* Tcw, int B, int H, int W, unsigned flags, float* grid, void* stream) {
    if (!points || !K || !Tcw || !grid || B <= 0 || H < 2 || W < 2 || B > 65535) return MCAV_E_INVALID;
    project_kernel<<<pix_grid(B, H, W), 256, 0, as_stream(stream)>>>(points, K, (flags & MCAV_WL_K_F64) ? 1 : 0, Tcw, B, H, W, grid);
    return launch_status();
}

MCAV_EXPORT int mcav_pose_to_matrix(const float* pose, int B, int invert, float* T, void* stream) {
    if (!pose || !T || B <= 0) return MCAV_E_INVALID;
    pose_to_matrix_kernel<<<(B + 63) / 64, 64, 0, as_stream(stream)>>>(pose, B, invert, T);
    return launch_status();
}

MCAV_EXPORT int mcav_invert_pose(const float* T, int B, float* Tinv, void* stream) {
    if (!T || !Tinv || B <= 0) return MCAV_E_INVALID;
    invert_pose_kernel<<<(B + 63) / 64, 64, 0, as_stream(stream)>>>(T, B, Tinv);
    return launch_status();
}

MCAV_EXPORT int mcav_disp_to_depth(const float* disp, float* depth, size_t n, void* stream) {
    if (!disp || !depth) return MCAV_E_INVALID;
    if (n == 0) return MCAV_OK;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    disp_to_depth_kernel<<<blocks, 256, 0, as_stream(stream)>>>(disp, depth, n);
    return launch_status();
}

MCAV_EXPORT int mcav_disp_to_depth_bwd(const float* disp, const float* d_depth, float* d_disp, size_t n, void* stream) {
    if (!disp || !d_depth || !d_disp) return MCAV_E_INVALID;
    if (n == 0) return MCAV_OK;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    disp_to_depth_bwd_kernel<<<blocks, 256, 0, as_stream(stream)>>>(disp, d_depth, d_disp, n);
    return launch_status();
}

MCAV_EXPORT int mcav_ssim_fwd(const float* x, const float* y, int N, int H, int W, float C1, float C2, float* out, void* stream) {
    if (!x || !y || !out || N <= 0 || H < 2 || W < 2 || N > 65535) return MCAV_E_INVALID;
    ssim_kernel<<<pix_grid(N, H, W), 256, 0, as_stream(stream)>>>(x, y, N, H, W, C1, C2, out);
    return launch_status();
}

MCAV_EXPORT size_t mcav_smooth_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return ws_layout(B, H, W).total;
}

MCAV_EXPORT int mcav_smooth_loss_fwd_bwd(const float* depth, int B, int H, int W, float weight, const float* upstream, float* loss_accum,
                                         float* d_depth, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    if (!depth || !loss_accum || !d_depth || !workspace || B <= 0 || H < 3 || W < 3 || B > 65535) return MCAV_E_INVALID;
    const WsLayout l = ws_layout(B, H, W);
    if (workspace_bytes < l.total) return MCAV_E_WORKSPACE;
    float* slab = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + l.slab_off);
    hipStream_t s = as_stream(stream);
    smooth_kernel<<<pix_grid(B, H, W), 256, 0, s>>>(depth, B, H, W, weight, upstream, d_depth, accumulate, slab);
    smooth_finalize_kernel<<<1, 256, 0, s>>>(slab, l.nblk * B, loss_accum);
    return launch_status();
}
