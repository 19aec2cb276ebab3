// TEST INFRASTRUCTURE (CPU): drives the per-pixel math of csrc/warp_math.h -- the very functions the HIP
// kernels call -- with plain host loops, so the arithmetic can be checked against the oracle in the
// GPU-less build container.  Built by tests/test_hostcheck.py with g++; never loaded by the product.
#include <cstddef>
#include <cstring>
#include <vector>

#include "warp_math.h"

using namespace mcav;

// The fused kernels' per-pixel path (csrc/warp_math.h "lean forms": project_fast, bilinear_lerp, warp_unit_fast) driven by host loops.
static void sample_consts(const double* K, const float* poses, int b, float* Kf, float* Kinv, WarpFast* wf) {
    double Ki[9];
    invert3x3(K + b * 9, Ki);
    for (int i = 0; i < 9; ++i) { Kf[i] = (float)K[b * 9 + i]; Kinv[i] = (float)Ki[i]; }
    float R[9], t[3], P[12];
    const float* p = poses + (size_t)b * 12;
    pose_to_Rt(p, false, R, t);     make_P(Kf, R, t, P); make_fast(P, Kinv, wf[0]);
    pose_to_Rt(p + 6, false, R, t); make_P(Kf, R, t, P); make_fast(P, Kinv, wf[1]);
    pose_to_Rt(p, true, R, t);      make_P(Kf, R, t, P); make_fast(P, Kinv, wf[2]);
}

// one warp at one pixel; returns sum |res|; dbg may be null
static float unit(const WarpFast& wf, const float* Kinv, const float* src, const float* tar, size_t plane, size_t pix, int x, int y, float D, int H, int W,
                  float gw, float& dD, float* dP, float* dbg) {
    const FTap t = project_fast(wf, (float)x, (float)y, D, H, W);
    float q[3][4], tv[3];
    for (int c = 0; c < 3; ++c) { texels_of(src + c * plane, W, t, q[c]); tv[c] = tar[c * plane + pix]; }
    const float fx = (float)x, fy = (float)y;
    const float X[3] = {fmaf(Kinv[0], fx, fmaf(Kinv[1], fy, Kinv[2])) * D, fmaf(Kinv[3], fx, fmaf(Kinv[4], fy, Kinv[5])) * D,
                        fmaf(Kinv[6], fx, fmaf(Kinv[7], fy, Kinv[8])) * D};
    float labs = 0.f;
    warp_unit_fast(q, tv, t, X, H, W, gw, labs, dD, dP, dbg);
    return labs;
}

extern "C" int hostcheck_warp_loss(const float* tgt, const float* ref0, const float* ref1, const float* disp_t, const float* disp_r0,
                                   const float* poses, const double* K, int B, int H, int W, const float* upstream, float* losses,
                                   float* d_disp_t, float* d_disp_r0, float* d_poses) {
    const size_t plane = (size_t)H * W;
    const float g0 = upstream[0], g1 = upstream[1];
    const float tw[3] = {0.25f, 0.25f, 0.5f};
    double loss_mam = 0.0, loss_smooth = 0.0;
    std::vector<float> Dt(plane);
    for (int b = 0; b < B; ++b) {
        float Kf[9], Kinv[9];
        WarpFast wf[3];
        sample_consts(K, poses, b, Kf, Kinv, wf);
        for (size_t i = 0; i < plane; ++i) Dt[i] = rcp_nr(fmaf(10.0f, disp_t[b * plane + i], 0.01f));
        double dP[3][12];
        std::memset(dP, 0, sizeof(dP));
        const float invN = 1.0f / (float)((size_t)B * 3 * plane);
        const float cxx = 1.0f / (float)((size_t)B * H * (W - 2));
        const float cyy = 1.0f / (float)((size_t)B * (H - 2) * W);
        const float cxy = 2.0f / (float)((size_t)B * (H - 1) * (W - 1));
        const float* it = tgt + (size_t)b * 3 * plane;
        const float* i0 = ref0 + (size_t)b * 3 * plane;
        const float* i1 = ref1 + (size_t)b * 3 * plane;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const size_t pix = (size_t)y * W + x;
                const float Dr = rcp_nr(fmaf(10.0f, disp_r0[b * plane + pix], 0.01f));
                float dDt = 0.f, dDr = 0.f, q[3][12];
                std::memset(q, 0, sizeof(q));
                float l = 0.f;
                l = fmaf(unit(wf[0], Kinv, i0, it, plane, pix, x, y, Dt[pix], H, W, g0 * tw[0] * invN, dDt, q[0], nullptr), tw[0] * invN, l);
                l = fmaf(unit(wf[1], Kinv, i1, it, plane, pix, x, y, Dt[pix], H, W, g0 * tw[1] * invN, dDt, q[1], nullptr), tw[1] * invN, l);
                l = fmaf(unit(wf[2], Kinv, it, i1, plane, pix, x, y, Dr, H, W, g0 * tw[2] * invN, dDr, q[2], nullptr), tw[2] * invN, l);
                for (int w = 0; w < 3; ++w)
                    for (int k = 0; k < 12; ++k) dP[w][k] += q[w][k];
                float ls = 0.f, gs = 0.f;
                smooth_terms([&](int dy, int dx) { return Dt[(size_t)(y + dy) * W + (x + dx)]; }, x, y, H, W, cxx, cyy, cxy, ls, gs);
                dDt += g1 * gs;
                loss_mam += l;
                loss_smooth += ls;
                d_disp_t[b * plane + pix] = dDt * (-10.0f * Dt[pix] * Dt[pix]);
                d_disp_r0[b * plane + pix] = dDr * (-10.0f * Dr * Dr);
            }
        const float* p = poses + (size_t)b * 12;
        double ga[6], gb[6], gc[6];
        pose_grad_from_dP(dP[0], Kf, p, false, ga);
        pose_grad_from_dP(dP[1], Kf, p + 6, false, gb);
        pose_grad_from_dP(dP[2], Kf, p, true, gc);
        for (int i = 0; i < 6; ++i) {
            d_poses[b * 12 + i] = (float)(ga[i] + gc[i]);
            d_poses[b * 12 + 6 + i] = (float)gb[i];
        }
    }
    losses[0] = (float)loss_mam;
    losses[1] = (float)loss_smooth;
    return 0;
}

// Per-pixel dump of the same arithmetic (the host twin of mcav_warp_loss_debug_taps): taps[B][3 warps][7][H][W] =
// {ix, iy, d loss_mam / d ix, d loss_mam / d iy, res[0..2]}, unit upstream.  tests/flip_finder.py diffs it against the float64 oracle.
extern "C" int hostcheck_warp_taps(const float* tgt, const float* ref0, const float* ref1, const float* disp_t, const float* disp_r0,
                                   const float* poses, const double* K, int B, int H, int W, float* taps) {
    const size_t plane = (size_t)H * W;
    const float tw[3] = {0.25f, 0.25f, 0.5f};
    for (int b = 0; b < B; ++b) {
        float Kf[9], Kinv[9];
        WarpFast wf[3];
        sample_consts(K, poses, b, Kf, Kinv, wf);
        const float invN = 1.0f / (float)((size_t)B * 3 * plane);
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const size_t pix = (size_t)y * W + x;
                const float Dt = rcp_nr(fmaf(10.0f, disp_t[b * plane + pix], 0.01f)), Dr = rcp_nr(fmaf(10.0f, disp_r0[b * plane + pix], 0.01f));
                for (int w = 0; w < 3; ++w) {
                    const float* src = (w == 0 ? ref0 : (w == 1 ? ref1 : tgt)) + (size_t)b * 3 * plane;
                    const float* tar = (w == 2 ? ref1 : tgt) + (size_t)b * 3 * plane;
                    float dbg[7], dD = 0.f, dP[12] = {0};
                    unit(wf[w], Kinv, src, tar, plane, pix, x, y, w == 2 ? Dr : Dt, H, W, tw[w] * invN, dD, dP, dbg);
                    for (int k = 0; k < 7; ++k) taps[(((size_t)b * 3 + w) * 7 + k) * plane + pix] = dbg[k];
                }
            }
    }
    return 0;
}
