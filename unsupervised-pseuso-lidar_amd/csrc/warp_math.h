// Per-pixel / per-sample arithmetic of the inverse-warp + photometric stage, shared by the HIP kernels
// (warp_loss.hip) and by the host-compiled index/math check used in tests/hostcheck (never by the product).
//
// Follows, operation for operation in fp32 where the reference is fp32:
//   geometry/transform.py:74-105   reconstruct   Xc = (K^-1 [x y 1]^T) * D
//   geometry/transform.py:114-150  project       P = (K4 @ Tcw)[:3];  pix = P [Xc;1];  /(z + 1e-5);  x/(W-1), y/(H-1), (.-0.5)*2
//   geometry/pose_geometry.py:110-199  Rodrigues (+1e-7), T = Trans @ Rot, rigid inverse
//   geometry/pose_geometry.py:227  F.grid_sample(bilinear, zeros, align_corners=True)  (+ its backward w.r.t. the grid)
//   geometry/pose_geometry.py:81-82  D = 1 / (10 * disp + 0.01)
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define MCAV_HD __host__ __device__ __forceinline__
#else
#define MCAV_HD inline
#endif

namespace mcav {

// Per (sample, warp) constants, produced once per launch by pose_prepare().
struct WarpConst {
    float P[12];     // rows of (K4 @ Tcw)[:3, :4]
};
struct SampleConst {
    float Kinv[9];   // float(K^-1), K inverted in fp64 as torch does for an fp64 K
    WarpConst w[3];
};

MCAV_HD void invert3x3(const double* K, double* Ki) {
    const double a = K[0], b = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], i = K[8];
    const double A = e * i - f * h, Bc = -(d * i - f * g), C = d * h - e * g;
    const double det = a * A + b * Bc + c * C;
    const double r = 1.0 / det;
    Ki[0] = A * r;  Ki[1] = -(b * i - c * h) * r;  Ki[2] = (b * f - c * e) * r;
    Ki[3] = Bc * r; Ki[4] = (a * i - c * g) * r;   Ki[5] = -(a * f - c * d) * r;
    Ki[6] = C * r;  Ki[7] = -(a * h - b * g) * r;  Ki[8] = (a * e - b * d) * r;
}

// axis-angle v[3] -> R[9] (row-major), fp32, same expression tree as the reference.
MCAV_HD void rodrigues(const float* v, float* R) {
    const float angle = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const float inv = 1.0f / (angle + 1e-7f);
    const float x = v[0] * inv, y = v[1] * inv, z = v[2] * inv;
    const float ca = cosf(angle), sa = sinf(angle), C = 1.0f - ca;
    const float xs = x * sa, ys = y * sa, zs = z * sa;
    const float xC = x * C, yC = y * C, zC = z * C;
    const float xyC = x * yC, yzC = y * zC, zxC = z * xC;
    R[0] = x * xC + ca; R[1] = xyC - zs;    R[2] = zxC + ys;
    R[3] = xyC + zs;    R[4] = y * yC + ca; R[5] = yzC - xs;
    R[6] = zxC - ys;    R[7] = yzC + xs;    R[8] = z * zC + ca;
}

// pose[6] = (axis-angle, translation) -> 3x4 [R|t] of Tcw (optionally the rigid inverse).
MCAV_HD void pose_to_Rt(const float* pose, bool invert, float* R, float* t) {
    float R0[9];
    rodrigues(pose, R0);
    if (!invert) {
        for (int i = 0; i < 9; ++i) R[i] = R0[i];
        t[0] = pose[3]; t[1] = pose[4]; t[2] = pose[5];
    } else {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R[i * 3 + j] = R0[j * 3 + i];
        for (int i = 0; i < 3; ++i)
            t[i] = (-1.0f * R[i * 3 + 0]) * pose[3] + (-1.0f * R[i * 3 + 1]) * pose[4] + (-1.0f * R[i * 3 + 2]) * pose[5];
    }
}

// P = K[3x3] @ [R|t]  (the first three rows of K4 @ Tcw).
MCAV_HD void make_P(const float* K, const float* R, const float* t, float* P) {
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j)
            P[i * 4 + j] = K[i * 3 + 0] * R[0 * 3 + j] + K[i * 3 + 1] * R[1 * 3 + j] + K[i * 3 + 2] * R[2 * 3 + j];
        P[i * 4 + 3] = K[i * 3 + 0] * t[0] + K[i * 3 + 1] * t[1] + K[i * 3 + 2] * t[2];
    }
}

struct Ray { float r0, r1, r2; };

MCAV_HD Ray pixel_ray(const float* Kinv, float x, float y) {
    Ray r;
    r.r0 = Kinv[0] * x + Kinv[1] * y + Kinv[2];
    r.r1 = Kinv[3] * x + Kinv[4] * y + Kinv[5];
    r.r2 = Kinv[6] * x + Kinv[7] * y + Kinv[8];
    return r;
}

// Everything one warp needs at one target pixel.
struct Tap {
    float ix, iy;          // un-normalised sampling position in source pixels
    float pix_x, pix_y;    // projected pixel coordinates (before normalisation)
    float zi;              // 1 / (z + 1e-5)
    float X0, X1, X2;      // camera point
    int x0, y0;            // floor
    float wx1, wx0, wy1, wy0;   // (ix - x0), (x1 - ix), (iy - y0), (y1 - iy)
    bool in00, in01, in10, in11;  // tap (y0,x0) (y0,x1) (y1,x0) (y1,x1) inside the image
};

MCAV_HD Tap project_pixel(const float* P, const Ray& r, float D, int H, int W) {
    Tap t;
    t.X0 = r.r0 * D; t.X1 = r.r1 * D; t.X2 = r.r2 * D;
    const float c0 = P[0] * t.X0 + P[1] * t.X1 + P[2] * t.X2 + P[3];
    const float c1 = P[4] * t.X0 + P[5] * t.X1 + P[6] * t.X2 + P[7];
    const float c2 = P[8] * t.X0 + P[9] * t.X1 + P[10] * t.X2 + P[11];
    const float z = c2 + 1e-5f;
    t.zi = 1.0f / z;
    t.pix_x = c0 / z;
    t.pix_y = c1 / z;
    const float gx = (t.pix_x / (float)(W - 1) - 0.5f) * 2.0f;
    const float gy = (t.pix_y / (float)(H - 1) - 0.5f) * 2.0f;
    t.ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
    t.iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
    // clamp to a range where float->int is defined; anything outside [-1, size] has no in-bounds tap anyway
    // (NaN compares false everywhere and ends up with all taps out of bounds)
    float cx = t.ix, cy = t.iy;
    if (!(cx > -2.0f)) cx = -2.0f;
    if (!(cx < (float)W + 1.0f)) cx = (float)W + 1.0f;
    if (!(cy > -2.0f)) cy = -2.0f;
    if (!(cy < (float)H + 1.0f)) cy = (float)H + 1.0f;
    const float fx = floorf(cx), fy = floorf(cy);
    t.x0 = (int)fx; t.y0 = (int)fy;
    const bool finite = (t.ix == t.ix) && (t.iy == t.iy) && (cx == t.ix) && (cy == t.iy);
    t.wx1 = t.ix - fx;            // weight of the east column
    t.wx0 = (fx + 1.0f) - t.ix;   // weight of the west column
    t.wy1 = t.iy - fy;
    t.wy0 = (fy + 1.0f) - t.iy;
    const bool xin0 = finite && t.x0 >= 0 && t.x0 < W, xin1 = finite && t.x0 + 1 >= 0 && t.x0 + 1 < W;
    const bool yin0 = finite && t.y0 >= 0 && t.y0 < H, yin1 = finite && t.y0 + 1 >= 0 && t.y0 + 1 < H;
    t.in00 = yin0 && xin0; t.in01 = yin0 && xin1; t.in10 = yin1 && xin0; t.in11 = yin1 && xin1;
    return t;
}

// Bilinear value and its derivative w.r.t. (ix, iy) for one channel plane.
struct Sample { float v, dvdx, dvdy; };

// the four texels of a tap (zero where the tap lies outside the image) -> value and derivatives
MCAV_HD Sample bilinear_from(float nw, float ne, float sw, float se, const Tap& t) {
    Sample s;
    s.v = nw * (t.wx0 * t.wy0) + ne * (t.wx1 * t.wy0) + sw * (t.wx0 * t.wy1) + se * (t.wx1 * t.wy1);
    s.dvdx = -nw * t.wy0 + ne * t.wy0 - sw * t.wy1 + se * t.wy1;
    s.dvdy = -nw * t.wx0 - ne * t.wx1 + sw * t.wx0 + se * t.wx1;
    return s;
}

MCAV_HD Sample bilinear(const float* plane, int W, const Tap& t) {
    const float nw = t.in00 ? plane[t.y0 * W + t.x0] : 0.0f;
    const float ne = t.in01 ? plane[t.y0 * W + t.x0 + 1] : 0.0f;
    const float sw = t.in10 ? plane[(t.y0 + 1) * W + t.x0] : 0.0f;
    const float se = t.in11 ? plane[(t.y0 + 1) * W + t.x0 + 1] : 0.0f;
    Sample s;
    s.v = nw * (t.wx0 * t.wy0) + ne * (t.wx1 * t.wy0) + sw * (t.wx0 * t.wy1) + se * (t.wx1 * t.wy1);
    s.dvdx = -nw * t.wy0 + ne * t.wy0 - sw * t.wy1 + se * t.wy1;
    s.dvdy = -nw * t.wx0 - ne * t.wx1 + sw * t.wx0 + se * t.wx1;
    return s;
}

// Chain d(loss)/d(ix, iy) back to depth and to the 12 entries of P.
// gix/giy: d loss / d ix, iy.  Returns d loss / d D and accumulates dP (12 floats).
MCAV_HD float backproject_grad(const float* P, const Ray& r, const Tap& t, float gix, float giy, int H, int W, float* dP) {
    // ix = ((g+1)/2)(W-1), g = (pix/(W-1) - .5)*2  ->  d pix = gix * ((W-1)/2) * (2/(W-1))
    const float dpx = (gix * ((float)(W - 1) / 2.0f)) * (2.0f / (float)(W - 1));
    const float dpy = (giy * ((float)(H - 1) / 2.0f)) * (2.0f / (float)(H - 1));
    const float dc0 = dpx * t.zi;
    const float dc1 = dpy * t.zi;
    const float dc2 = -(dpx * t.pix_x + dpy * t.pix_y) * t.zi;
    dP[0] += dc0 * t.X0; dP[1] += dc0 * t.X1; dP[2] += dc0 * t.X2;  dP[3] += dc0;
    dP[4] += dc1 * t.X0; dP[5] += dc1 * t.X1; dP[6] += dc1 * t.X2;  dP[7] += dc1;
    dP[8] += dc2 * t.X0; dP[9] += dc2 * t.X1; dP[10] += dc2 * t.X2; dP[11] += dc2;
    const float q0 = P[0] * r.r0 + P[1] * r.r1 + P[2] * r.r2;
    const float q1 = P[4] * r.r0 + P[5] * r.r1 + P[6] * r.r2;
    const float q2 = P[8] * r.r0 + P[9] * r.r1 + P[10] * r.r2;
    return dc0 * q0 + dc1 * q1 + dc2 * q2;
}

MCAV_HD float sgn(float v) { return (float)((v > 0.0f) - (v < 0.0f)); }

// Second-order smoothness (losses.py:242-260) at pixel (x, y): DD(dy, dx) reads the depth at (y+dy, x+dx).
// ls += this pixel's share of the loss (each term is counted once, by the pixel at its origin);
// gs += d loss / d D[y][x] (gathered from every term D[y][x] takes part in).
// cxx/cyy/cxy: weight / element count of the dx2, dy2 and (dxdy + dydx) means.
template <class F>
MCAV_HD void smooth_terms(F DD, int x, int y, int H, int W, float cxx, float cyy, float cxy, float& ls, float& gs) {
    // second difference along x: t(x') = D[x'+2] - 2 D[x'+1] + D[x'], x' in [0, W-3]
    if (x <= W - 3) { const float t = DD(0, 2) - 2.f * DD(0, 1) + DD(0, 0); ls += fabsf(t) * cxx; gs += sgn(t) * cxx; }
    if (x >= 1 && x <= W - 2) { const float t = DD(0, 1) - 2.f * DD(0, 0) + DD(0, -1); gs -= 2.f * sgn(t) * cxx; }
    if (x >= 2) { const float t = DD(0, 0) - 2.f * DD(0, -1) + DD(0, -2); gs += sgn(t) * cxx; }
    // second difference along y
    if (y <= H - 3) { const float t = DD(2, 0) - 2.f * DD(1, 0) + DD(0, 0); ls += fabsf(t) * cyy; gs += sgn(t) * cyy; }
    if (y >= 1 && y <= H - 2) { const float t = DD(1, 0) - 2.f * DD(0, 0) + DD(-1, 0); gs -= 2.f * sgn(t) * cyy; }
    if (y >= 2) { const float t = DD(0, 0) - 2.f * DD(-1, 0) + DD(-2, 0); gs += sgn(t) * cyy; }
    // mixed difference u(y',x') = (D[y'+1,x'+1] - D[y'+1,x']) - (D[y',x'+1] - D[y',x']); dxdy and dydx are the same field
    if (y <= H - 2 && x <= W - 2) { const float u = (DD(1, 1) - DD(1, 0)) - (DD(0, 1) - DD(0, 0)); ls += fabsf(u) * cxy; gs += sgn(u) * cxy; }
    if (y <= H - 2 && x >= 1) { const float u = (DD(1, 0) - DD(1, -1)) - (DD(0, 0) - DD(0, -1)); gs -= sgn(u) * cxy; }
    if (y >= 1 && x <= W - 2) { const float u = (DD(0, 1) - DD(0, 0)) - (DD(-1, 1) - DD(-1, 0)); gs -= sgn(u) * cxy; }
    if (y >= 1 && x >= 1) { const float u = (DD(0, 0) - DD(0, -1)) - (DD(-1, 0) - DD(-1, -1)); gs += sgn(u) * cxy; }
}

// The same ten terms with every condition turned into a 0 / weight factor (no exec-mask branches in the fused kernel).  DD must be readable
// at every offset within +-2 (the kernel's tile carries a zero-filled halo of 2), and a masked term adds +-0: bit-identical sums.
template <class F>
MCAV_HD void smooth_terms_sel(F DD, int x, int y, int H, int W, float cxx, float cyy, float cxy, float& ls, float& gs) {
    const float d00 = DD(0, 0);
    { const float t = DD(0, 2) - 2.f * DD(0, 1) + d00; const float m = x <= W - 3 ? cxx : 0.f; ls += fabsf(t) * m; gs += sgn(t) * m; }
    { const float t = DD(0, 1) - 2.f * d00 + DD(0, -1); const float m = (x >= 1 && x <= W - 2) ? cxx : 0.f; gs -= 2.f * sgn(t) * m; }
    { const float t = d00 - 2.f * DD(0, -1) + DD(0, -2); const float m = x >= 2 ? cxx : 0.f; gs += sgn(t) * m; }
    { const float t = DD(2, 0) - 2.f * DD(1, 0) + d00; const float m = y <= H - 3 ? cyy : 0.f; ls += fabsf(t) * m; gs += sgn(t) * m; }
    { const float t = DD(1, 0) - 2.f * d00 + DD(-1, 0); const float m = (y >= 1 && y <= H - 2) ? cyy : 0.f; gs -= 2.f * sgn(t) * m; }
    { const float t = d00 - 2.f * DD(-1, 0) + DD(-2, 0); const float m = y >= 2 ? cyy : 0.f; gs += sgn(t) * m; }
    { const float u = (DD(1, 1) - DD(1, 0)) - (DD(0, 1) - d00); const float m = (y <= H - 2 && x <= W - 2) ? cxy : 0.f; ls += fabsf(u) * m; gs += sgn(u) * m; }
    { const float u = (DD(1, 0) - DD(1, -1)) - (d00 - DD(0, -1)); const float m = (y <= H - 2 && x >= 1) ? cxy : 0.f; gs -= sgn(u) * m; }
    { const float u = (DD(0, 1) - d00) - (DD(-1, 1) - DD(-1, 0)); const float m = (y >= 1 && x <= W - 2) ? cxy : 0.f; gs -= sgn(u) * m; }
    { const float u = (d00 - DD(0, -1)) - (DD(-1, 0) - DD(-1, -1)); const float m = (y >= 1 && x >= 1) ? cxy : 0.f; gs += sgn(u) * m; }
}

// One warp at one pixel: photometric L1 over 3 channel planes + gradient back to depth and P.
// src: 3 planes of the source image (stride `plane`), tv: the 3 target values.
// lw: weight of |res| in the loss; gw: weight of sign(res) in the gradient (= upstream * lw).
// The same with the 3 x 4 texels already fetched (q[c][0..3] = nw, ne, sw, se of channel c): the fused kernel issues the gathers of all
// three warps of a pixel before it consumes any (csrc/warp_loss.hip).
// dbg (test builds of the kernel only, tests/flip_finder.py): receives {ix, iy, d loss / d ix, d loss / d iy, res[0..2]} of this warp at
// this pixel -- the per-pixel quantities the fp64 oracle is diffed against to NAME a pixel whose cell / L1 sign an fp32 evaluation flips.
MCAV_HD void warp_pixel_from(const float (*q)[4], const float* tv, const float* P, const Ray& r, const Tap& t, int H, int W,
                             float lw, float gw, float& loss, float& dD, float* dP, float* dbg = nullptr) {
    float gix = 0.f, giy = 0.f;
    for (int c = 0; c < 3; ++c) {
        const Sample s = bilinear_from(q[c][0], q[c][1], q[c][2], q[c][3], t);
        const float res = s.v - tv[c];
        loss += fabsf(res) * lw;
        const float sg = sgn(res) * gw;
        gix += sg * s.dvdx;
        giy += sg * s.dvdy;
        if (dbg) dbg[4 + c] = res;
    }
    if (dbg) { dbg[0] = t.ix; dbg[1] = t.iy; dbg[2] = gix; dbg[3] = giy; }
    dD += backproject_grad(P, r, t, gix, giy, H, W, dP);
}

MCAV_HD void warp_pixel(const float* src, size_t plane, const float* tv, const float* P, const Ray& r, float D, int H, int W,
                        float lw, float gw, float& loss, float& dD, float* dP) {
    const Tap t = project_pixel(P, r, D, H, W);
    float gix = 0.f, giy = 0.f;
    for (int c = 0; c < 3; ++c) {
        const Sample s = bilinear(src + c * plane, W, t);
        const float res = s.v - tv[c];
        loss += fabsf(res) * lw;
        const float sg = sgn(res) * gw;
        gix += sg * s.dvdx;
        giy += sg * s.dvdy;
    }
    dD += backproject_grad(P, r, t, gix, giy, H, W, dP);
}

// ---------------------------------------------------------------------------------------------- the fused kernels' lean forms (round 3)
// Same quantities as project_pixel / bilinear_from / backproject_grad with a third of the instructions; what changes is rounding, not formulas:
//   * the camera coordinates c = P [r D; 1] with r = K^-1 [x y 1]^T are evaluated as c_i = D q_i + P_i3, q_i = (P[:, :3] K^-1 [x y 1]^T)_i, and
//     q_i is affine in the pixel: q_i = Q_i0 x + Q_i1 y + Q_i2 with Q = P[:, :3] K^-1 formed once per (sample, warp) in float64 -- two fmas
//     per coordinate instead of the reference's r (6), X = r D (3) and P X (12); closer to the exact value than the fp32 chain it replaces;
//   * divisions become v_rcp_f32 + one Newton step (<= 1 ulp from the correctly rounded quotient) times the numerator;
//   * the reference's normalise / un-normalise round trip of the sampling position is kept -- (p / (W-1) - 0.5) * 2 and ((g + 1) / 2) * (W-1) --
//     with its exact scalings by 2 and 1/2 folded into the neighbouring roundings (identical results), the division by W - 1 as a product;
//   * the bilinear sample is the nested lerp (its x / y derivatives fall out of it), not four weighted texels.
// Everything is within a few ulp of the reference's fp32 value; tests/hostcheck runs these very functions against the reference goldens.
struct WarpFast {
    float Q[9];      // rows of P[:, :3] @ K^-1
    float p3[3];     // P[:, 3]
};

MCAV_HD void make_fast(const float* P, const float* Kinv, WarpFast& f) {
    for (int i = 0; i < 3; ++i) {
        for (int k = 0; k < 3; ++k)
            f.Q[i * 3 + k] = (float)((double)P[i * 4 + 0] * (double)Kinv[0 * 3 + k] + (double)P[i * 4 + 1] * (double)Kinv[1 * 3 + k] +
                                     (double)P[i * 4 + 2] * (double)Kinv[2 * 3 + k]);
        f.p3[i] = P[i * 4 + 3];
    }
}

MCAV_HD float rcp_nr(float z) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float r = __builtin_amdgcn_rcpf(z);
    return fmaf(r, fmaf(-z, r, 1.0f), r);
#else
    return 1.0f / z;
#endif
}

struct FTap {
    float wx1, wy1;            // position inside the bilinear cell
    float zi, px, py;          // 1 / (z + 1e-5), projected pixel
    float q0, q1, q2;          // d c / d D
    float ix, iy;              // sampling position (source pixels)
    int x0, y0;                // the cell; clamped to [-2, size + 1]
    bool in00, in01, in10, in11;
};

// live: false forces every tap out of the image (pixels past the image edge in a ragged tile)
MCAV_HD FTap project_fast(const WarpFast& f, float x, float y, float D, int H, int W, bool live = true) {
    FTap t;
    t.q0 = fmaf(f.Q[0], x, fmaf(f.Q[1], y, f.Q[2]));
    t.q1 = fmaf(f.Q[3], x, fmaf(f.Q[4], y, f.Q[5]));
    t.q2 = fmaf(f.Q[6], x, fmaf(f.Q[7], y, f.Q[8]));
    const float c0 = fmaf(D, t.q0, f.p3[0]), c1 = fmaf(D, t.q1, f.p3[1]), c2 = fmaf(D, t.q2, f.p3[2]);
    t.zi = rcp_nr(c2 + 1e-5f);
    t.px = c0 * t.zi;
    t.py = c1 * t.zi;
    const float w1 = (float)(W - 1), h1 = (float)(H - 1);
    const float gx = fmaf(t.px * rcp_nr(w1), 2.0f, -1.0f), gy = fmaf(t.py * rcp_nr(h1), 2.0f, -1.0f);      // (p / (W-1) - 0.5) * 2
    t.ix = (gx + 1.0f) * (0.5f * w1);                                                                       // ((g + 1) / 2) * (W-1)
    t.iy = (gy + 1.0f) * (0.5f * h1);
    // clamp before the conversion (NaN -> -2): beyond [-2, size + 1] no tap is inside the image anyway, and the weights stay finite
    const float cx = fminf(fmaxf(t.ix, -2.0f), w1 + 2.0f), cy = fminf(fmaxf(t.iy, -2.0f), h1 + 2.0f);
    const float fx = floorf(cx), fy = floorf(cy);
    t.wx1 = cx - fx;
    t.wy1 = cy - fy;
    t.x0 = (int)fx;
    t.y0 = (int)fy;
    const bool xa = (unsigned)t.x0 < (unsigned)W, xb = (unsigned)(t.x0 + 1) < (unsigned)W;
    const bool ya = live && (unsigned)t.y0 < (unsigned)H, yb = live && (unsigned)(t.y0 + 1) < (unsigned)H;
    t.in00 = ya && xa; t.in01 = ya && xb; t.in10 = yb && xa; t.in11 = yb && xb;
    return t;
}

// value and derivatives w.r.t. (ix, iy) from the four texels (zero outside the image)
MCAV_HD Sample bilinear_lerp(float nw, float ne, float sw, float se, float wx1, float wy1) {
    const float t0 = ne - nw, t1 = se - sw;
    const float top = fmaf(wx1, t0, nw), bot = fmaf(wx1, t1, sw);
    Sample s;
    s.dvdy = bot - top;
    s.v = fmaf(wy1, s.dvdy, top);
    s.dvdx = fmaf(wy1, t1 - t0, t0);
    return s;
}

MCAV_HD float sgn_exact(float v) {      // copysign(1, v), 0 at 0
    const float s = copysignf(1.0f, v);
    return v == 0.0f ? 0.0f : s;
}

// Chain d loss / d (ix, iy) back to the depth and to the 12 entries of P (X = camera point r * D).  Returns d loss / d D.
MCAV_HD float backproject_fast(const FTap& t, const float* X, float gix, float giy, int H, int W, float* dP) {
    const float w1 = (float)(W - 1), h1 = (float)(H - 1);
    const float dpx = (gix * (w1 * 0.5f)) * (2.0f / w1), dpy = (giy * (h1 * 0.5f)) * (2.0f / h1);
    const float dc0 = dpx * t.zi, dc1 = dpy * t.zi;
    const float dc2 = -(fmaf(dpx, t.px, dpy * t.py)) * t.zi;
    dP[0] = fmaf(dc0, X[0], dP[0]); dP[1] = fmaf(dc0, X[1], dP[1]); dP[2] = fmaf(dc0, X[2], dP[2]);   dP[3] += dc0;
    dP[4] = fmaf(dc1, X[0], dP[4]); dP[5] = fmaf(dc1, X[1], dP[5]); dP[6] = fmaf(dc1, X[2], dP[6]);   dP[7] += dc1;
    dP[8] = fmaf(dc2, X[0], dP[8]); dP[9] = fmaf(dc2, X[1], dP[9]); dP[10] = fmaf(dc2, X[2], dP[10]); dP[11] += dc2;
    return fmaf(dc0, t.q0, fmaf(dc1, t.q1, dc2 * t.q2));
}

// One warp at one pixel from its 3 x 4 texels: sum of |residual| (unweighted), d loss / d depth, the 12 dP sums.
// gw: weight of sign(res) in the gradient (upstream * term weight / N); X: the camera point r * D; dbg as in warp_pixel_from.
MCAV_HD void warp_unit_fast(const float (*q)[4], const float* tv, const FTap& t, const float* X, int H, int W, float gw,
                            float& labs, float& dD, float* dP, float* dbg = nullptr) {
    float gix = 0.f, giy = 0.f;
    for (int c = 0; c < 3; ++c) {
        const Sample s = bilinear_lerp(q[c][0], q[c][1], q[c][2], q[c][3], t.wx1, t.wy1);
        const float res = s.v - tv[c];
        labs += fabsf(res);
        const float sg = sgn_exact(res);
        gix = fmaf(sg, s.dvdx, gix);
        giy = fmaf(sg, s.dvdy, giy);
        if (dbg) dbg[4 + c] = res;
    }
    gix *= gw;
    giy *= gw;
    if (dbg) { dbg[0] = t.ix; dbg[1] = t.iy; dbg[2] = gix; dbg[3] = giy; }
    dD += backproject_fast(t, X, gix, giy, H, W, dP);
}

// the four texels of a tap straight from a plane (standalone / SSIM kernels, host check)
MCAV_HD void texels_of(const float* plane, int W, const FTap& t, float* q4) {
    q4[0] = t.in00 ? plane[t.y0 * W + t.x0] : 0.0f;
    q4[1] = t.in01 ? plane[t.y0 * W + t.x0 + 1] : 0.0f;
    q4[2] = t.in10 ? plane[(t.y0 + 1) * W + t.x0] : 0.0f;
    q4[3] = t.in11 ? plane[(t.y0 + 1) * W + t.x0 + 1] : 0.0f;
}

// d loss / dP (3x4, summed over pixels, fp64) -> d loss / d pose[6], through K, the optional rigid inverse,
// T = Trans @ Rot and Rodrigues with the +1e-7 guard.  All in fp64.
MCAV_HD void pose_grad_from_dP(const double* dP, const float* Kf, const float* pose, bool invert, double* dpose) {
    // dM = K^T dP  (M = [R|t] actually used, 3x4)
    double dM[12];
    for (int k = 0; k < 3; ++k)
        for (int j = 0; j < 4; ++j)
            dM[k * 4 + j] = (double)Kf[0 * 3 + k] * dP[0 * 4 + j] + (double)Kf[1 * 3 + k] * dP[1 * 4 + j] + (double)Kf[2 * 3 + k] * dP[2 * 4 + j];
    const double vx = pose[0], vy = pose[1], vz = pose[2];
    const double tx = pose[3], ty = pose[4], tz = pose[5];
    const double angle = sqrt(vx * vx + vy * vy + vz * vz);
    const double inv = 1.0 / (angle + (double)1e-7f);
    const double x = vx * inv, y = vy * inv, z = vz * inv;
    const double ca = cos(angle), sa = sin(angle), C = 1.0 - ca;
    double R0[9] = {x * x * C + ca, x * y * C - z * sa, z * x * C + y * sa,
                    x * y * C + z * sa, y * y * C + ca, y * z * C - x * sa,
                    z * x * C - y * sa, y * z * C + x * sa, z * z * C + ca};
    double dR[9], dt[3];
    if (!invert) {
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) dR[i * 3 + j] = dM[i * 4 + j];
            dt[i] = dM[i * 4 + 3];
        }
    } else {
        // M[:, :3] = R0^T ; M[:, 3]_i = -sum_a R0[a][i] t_a
        const double t0[3] = {tx, ty, tz};
        for (int a = 0; a < 3; ++a) {
            double acc = 0.0;
            for (int i = 0; i < 3; ++i) {
                dR[a * 3 + i] = dM[i * 4 + a] - dM[i * 4 + 3] * t0[a];
                acc -= R0[a * 3 + i] * dM[i * 4 + 3];
            }
            dt[a] = acc;
        }
    }
    const double dx = dR[0] * 2 * x * C + (dR[1] + dR[3]) * y * C + (dR[2] + dR[6]) * z * C + (dR[7] - dR[5]) * sa;
    const double dy = (dR[1] + dR[3]) * x * C + dR[4] * 2 * y * C + (dR[5] + dR[7]) * z * C + (dR[2] - dR[6]) * sa;
    const double dz = (dR[2] + dR[6]) * x * C + (dR[5] + dR[7]) * y * C + dR[8] * 2 * z * C + (dR[3] - dR[1]) * sa;
    const double dC = dR[0] * x * x + (dR[1] + dR[3]) * x * y + (dR[2] + dR[6]) * z * x + dR[4] * y * y + (dR[5] + dR[7]) * y * z + dR[8] * z * z;
    const double dca = dR[0] + dR[4] + dR[8];
    const double dsa = (dR[3] - dR[1]) * z + (dR[2] - dR[6]) * y + (dR[7] - dR[5]) * x;
    double dangle = -sa * dca + ca * dsa + sa * dC;
    // axis_i = v_i * inv ; inv = 1/(angle+eps)
    const double daxis_dot_v = dx * vx + dy * vy + dz * vz;
    dangle -= daxis_dot_v * inv * inv;
    const double s = angle > 0.0 ? dangle / angle : 0.0;   // d|v|/dv = v/|v| (0 at v = 0, as torch.norm's backward)
    dpose[0] = dx * inv + s * vx;
    dpose[1] = dy * inv + s * vy;
    dpose[2] = dz * inv + s * vz;
    dpose[3] = dt[0]; dpose[4] = dt[1]; dpose[5] = dt[2];
}

}  // namespace mcav
