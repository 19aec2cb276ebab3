// TEST INFRASTRUCTURE (CPU): drives the per-pixel math of csrc/warp_math.h -- the very functions the HIP
// kernels call -- with plain host loops, so the arithmetic can be checked against the oracle in the
// GPU-less build container.  Built by tests/test_hostcheck.py with g++; never loaded by the product.
#include <cstddef>
#include <cstring>
#include <vector>

#include "warp_math.h"

using namespace mcav;

extern "C" int hostcheck_warp_loss(const float* tgt, const float* ref0, const float* ref1, const float* disp_t, const float* disp_r0,
                                   const float* poses, const double* K, int B, int H, int W, const float* upstream, float* losses,
                                   float* d_disp_t, float* d_disp_r0, float* d_poses) {
    const size_t plane = (size_t)H * W;
    const float g0 = upstream[0], g1 = upstream[1];
    const float tw[3] = {0.25f, 0.25f, 0.5f};
    double loss_mam = 0.0, loss_smooth = 0.0;
    std::vector<float> Dt(plane);
    for (int b = 0; b < B; ++b) {
        double Ki[9];
        invert3x3(K + b * 9, Ki);
        float Kf[9], Kinv[9];
        for (int i = 0; i < 9; ++i) { Kf[i] = (float)K[b * 9 + i]; Kinv[i] = (float)Ki[i]; }
        float R[9], t[3], P[3][12];
        const float* p = poses + (size_t)b * 12;
        pose_to_Rt(p, false, R, t);     make_P(Kf, R, t, P[0]);
        pose_to_Rt(p + 6, false, R, t); make_P(Kf, R, t, P[1]);
        pose_to_Rt(p, true, R, t);      make_P(Kf, R, t, P[2]);
        for (size_t i = 0; i < plane; ++i) Dt[i] = 1.0f / (10.0f * disp_t[b * plane + i] + 0.01f);
        double dP[3][12];
        std::memset(dP, 0, sizeof(dP));
        const float invN = 1.0f / (float)((size_t)B * 3 * plane);
        const float cxx = 1.0f / (float)((size_t)B * H * (W - 2));
        const float cyy = 1.0f / (float)((size_t)B * (H - 2) * W);
        const float cxy = 2.0f / (float)((size_t)B * (H - 1) * (W - 1));
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const size_t pix = (size_t)y * W + x;
                const float Dr = 1.0f / (10.0f * disp_r0[b * plane + pix] + 0.01f);
                float tv[3], rv[3];
                for (int c = 0; c < 3; ++c) { tv[c] = tgt[(b * 3 + c) * plane + pix]; rv[c] = ref1[(b * 3 + c) * plane + pix]; }
                const Ray r = pixel_ray(Kinv, (float)x, (float)y);
                float l = 0.f, dDt = 0.f, dDr = 0.f, q[3][12];
                std::memset(q, 0, sizeof(q));
                warp_pixel(ref0 + (size_t)b * 3 * plane, plane, tv, P[0], r, Dt[pix], H, W, tw[0] * invN, g0 * tw[0] * invN, l, dDt, q[0]);
                warp_pixel(ref1 + (size_t)b * 3 * plane, plane, tv, P[1], r, Dt[pix], H, W, tw[1] * invN, g0 * tw[1] * invN, l, dDt, q[1]);
                warp_pixel(tgt + (size_t)b * 3 * plane, plane, rv, P[2], r, Dr, H, W, tw[2] * invN, g0 * tw[2] * invN, l, dDr, q[2]);
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
        double Ki[9];
        invert3x3(K + b * 9, Ki);
        float Kf[9], Kinv[9];
        for (int i = 0; i < 9; ++i) { Kf[i] = (float)K[b * 9 + i]; Kinv[i] = (float)Ki[i]; }
        float R[9], t[3], P[3][12];
        const float* p = poses + (size_t)b * 12;
        pose_to_Rt(p, false, R, t);     make_P(Kf, R, t, P[0]);
        pose_to_Rt(p + 6, false, R, t); make_P(Kf, R, t, P[1]);
        pose_to_Rt(p, true, R, t);      make_P(Kf, R, t, P[2]);
        const float invN = 1.0f / (float)((size_t)B * 3 * plane);
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const size_t pix = (size_t)y * W + x;
                const float Dt = 1.0f / (10.0f * disp_t[b * plane + pix] + 0.01f), Dr = 1.0f / (10.0f * disp_r0[b * plane + pix] + 0.01f);
                const Ray r = pixel_ray(Kinv, (float)x, (float)y);
                for (int w = 0; w < 3; ++w) {
                    const float* src = (w == 0 ? ref0 : (w == 1 ? ref1 : tgt)) + (size_t)b * 3 * plane;
                    const float* tar = (w == 2 ? ref1 : tgt) + (size_t)b * 3 * plane;
                    const Tap tp = project_pixel(P[w], r, w == 2 ? Dr : Dt, H, W);
                    float q[3][4], tv[3], dbg[7], l = 0.f, dD = 0.f, dP[12] = {0};
                    for (int c = 0; c < 3; ++c) {
                        const float* pl = src + c * plane;
                        q[c][0] = tp.in00 ? pl[tp.y0 * W + tp.x0] : 0.f;
                        q[c][1] = tp.in01 ? pl[tp.y0 * W + tp.x0 + 1] : 0.f;
                        q[c][2] = tp.in10 ? pl[(tp.y0 + 1) * W + tp.x0] : 0.f;
                        q[c][3] = tp.in11 ? pl[(tp.y0 + 1) * W + tp.x0 + 1] : 0.f;
                        tv[c] = tar[c * plane + pix];
                    }
                    warp_pixel_from(q, tv, P[w], r, tp, H, W, tw[w] * invN, tw[w] * invN, l, dD, dP, dbg);
                    for (int k = 0; k < 7; ++k) taps[(((size_t)b * 3 + w) * 7 + k) * plane + pix] = dbg[k];
                }
            }
    }
    return 0;
}
