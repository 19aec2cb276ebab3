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
