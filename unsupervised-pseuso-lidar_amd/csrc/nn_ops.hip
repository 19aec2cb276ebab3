// HBM-bound network kernels for gfx950: layout conversion, BatchNorm (training statistics / apply / backward),
// MaxPool 3x3 s2, activation backward, spatial mean, fused Adam.  All NHWC, 16-byte accesses, grid-stride loops;
// reductions are two-stage (per-block partials, fixed-order finalize in fp64) so results are run-to-run identical.
#include "conv_gather.h"

namespace mcav {

inline int grid_for(size_t work_items, int per_block = 256, int cap = 4096) {
    size_t b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (int)(b < (size_t)cap ? b : (size_t)cap);
}

// ---------------------------------------------------------------------------------------------- layout
__global__ void nchw_to_nhwc_kernel(const float* src, int B, int C, int H, int W, float* dst, int Cp, int choff) {
    const size_t plane = (size_t)H * W, total = (size_t)B * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / plane, pix = i - b * plane;
        for (int c = 0; c < C; ++c) dst[i * Cp + choff + c] = src[(b * C + c) * plane + pix];
    }
}

// The image case: C <= 4 planes into a 4-channel pixel written as one 16-byte store (channels past C keep what the buffer held).
__global__ __launch_bounds__(256) void nchw_to_nhwc4_kernel(const float* src, int B, int C, int H, int W, float* dst) {
    const size_t plane = (size_t)H * W, total = (size_t)B * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / plane, pix = i - b * plane;
        f32x4 v = *reinterpret_cast<const f32x4*>(dst + i * 4);
        const float* s0 = src + (b * C) * plane + pix;
        v.x = s0[0];
        if (C > 1) v.y = s0[plane];
        if (C > 2) v.z = s0[2 * plane];
        if (C > 3) v.w = s0[3 * plane];
        *reinterpret_cast<f32x4*>(dst + i * 4) = v;
    }
}

// The pose network's input: cat(s0, s1, s2) along channels (C planes each) into Cp-channel NHWC pixels, zeros past 3 C.  One lane per
// 16-byte chunk of a pixel: a wavefront stores 1 KB contiguous, and reads 64-byte runs of each plane.
__global__ __launch_bounds__(256) void nchw3_to_nhwc_kernel(const float* s0, const float* s1, const float* s2, int B, int C, int H, int W,
                                                            float* dst, int Cp) {
    const size_t plane = (size_t)H * W;
    const int chunks = Cp >> 2;
    const size_t total = (size_t)B * plane * chunks;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / chunks;
        const int q = (int)(i - p * chunks);
        const size_t b = p / plane, pix = p - b * plane;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = q * 4 + j;
            v[j] = 0.f;
            if (ch < 3 * C) {
                const int which = ch / C, c = ch - which * C;
                const float* src = which == 0 ? s0 : (which == 1 ? s1 : s2);
                v[j] = src[(b * C + c) * plane + pix];
            }
        }
        f32x4 o;
        o.x = v[0]; o.y = v[1]; o.z = v[2]; o.w = v[3];
        *reinterpret_cast<f32x4*>(dst + i * 4) = o;
    }
}

__global__ void nhwc_to_nchw_kernel(const float* src, int B, int C, int H, int W, int Cp, int choff, float* dst) {
    const size_t plane = (size_t)H * W, total = (size_t)B * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / plane, pix = i - b * plane;
        for (int c = 0; c < C; ++c) dst[(b * C + c) * plane + pix] = src[i * Cp + choff + c];
    }
}

// ---------------------------------------------------------------------------------------------- BatchNorm
// stats: [groups][mtiles][2][C] per-tile sums of x and x^2 (from the conv epilogue).
// Stage 1 (only when there are many tiles): grid (C/64, slices, groups); a block sums one slice of the tiles for 64 channels
// with 4 tile lanes, in fp64, into part[group][slice][2][C].  Stage 2: 32 channels x 8 slice lanes per block, fixed order.
constexpr int BNF_SLICES = 128;        // upper bound of stage-1 slices
constexpr int BNF_TICKETS = 256;       // completion tickets at the head of the finalize workspace: one per 64 channels (C <= 16384), 1 KB
constexpr int BNF_DIRECT = 64;         // up to this many tiles the finalize kernel reads the fp32 slabs itself

__global__ __launch_bounds__(256) void bn_partial_kernel(const float* stats, int mtiles, int C, int per_slice, double* part) {
    __shared__ double s1[4][64], s2[4][64];
    const int cl = threadIdx.x & 63, tl = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    const int slice = blockIdx.y, g = blockIdx.z, nsl = gridDim.y;
    const float* st = stats + (size_t)g * mtiles * 2 * C;
    const int t0 = slice * per_slice, t1 = min(mtiles, t0 + per_slice);
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int t = t0 + tl; t < t1; t += 4) {
            a += (double)st[((size_t)t * 2 + 0) * C + c];
            b += (double)st[((size_t)t * 2 + 1) * C + c];
        }
    s1[tl][cl] = a; s2[tl][cl] = b;
    __syncthreads();
    if (tl == 0 && c < C) {
        for (int k = 1; k < 4; ++k) { a += s1[k][cl]; b += s2[k][cl]; }
        double* o = part + ((size_t)g * nsl + slice) * 2 * C;
        o[c] = a;
        o[C + c] = b;
    }
}

// Stage 1 and the finalize in ONE launch (round 4: 15 launches fewer per step).  Every block leaves its slice's partial sums in `part`
// (doubles, agent-scope stores: the hand-off protocol of mcav_common.h); a ticket per 64-channel column names the LAST block of that column,
// which sums the slices in index order (bit-reproducible whichever block finishes) for every group in turn -- running statistics see pass
// 0, then pass 1 -- and leaves the ticket at zero for the next launch on this workspace.
__global__ __launch_bounds__(256) void bn_partial_finalize_kernel(const float* stats, int mtiles, int C, int per_slice, double* part, unsigned* tickets,
                                                                  double count, const float* gamma, const float* beta, float eps, float momentum,
                                                                  float* running_mean, float* running_var, float* scale, float* shift,
                                                                  float* save_mean, float* save_invstd) {
    __shared__ double s1[4][64], s2[4][64];
    __shared__ int s_last;
    const int cl = threadIdx.x & 63, tl = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    const int slice = blockIdx.y, g = blockIdx.z, nsl = gridDim.y, groups = gridDim.z;
    const float* st = stats + (size_t)g * mtiles * 2 * C;
    const int t0 = slice * per_slice, t1 = min(mtiles, t0 + per_slice);
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int t = t0 + tl; t < t1; t += 4) {
            a += (double)st[((size_t)t * 2 + 0) * C + c];
            b += (double)st[((size_t)t * 2 + 1) * C + c];
        }
    s1[tl][cl] = a; s2[tl][cl] = b;
    __syncthreads();
    if (tl == 0 && c < C) {
        for (int k = 1; k < 4; ++k) { a += s1[k][cl]; b += s2[k][cl]; }
        double* o = part + ((size_t)g * nsl + slice) * 2 * C;
        handoff_store(o + c, a);
        handoff_store(o + C + c, b);
    }
    handoff_release();
    __syncthreads();
    if (threadIdx.x == 0) s_last = handoff_ticket(&tickets[blockIdx.x]) == (unsigned)(nsl * groups - 1);
    __syncthreads();
    if (!s_last) return;
    for (int gg = 0; gg < groups; ++gg) {
        a = 0.0; b = 0.0;
        if (c < C) {
            // eight slices' partials in flight, added in slice order (one dependent agent-scope load per addition took the finisher 10 us)
            const double* base = part + (size_t)gg * nsl * 2 * C;
            int sl = tl;
            for (; sl + 28 < nsl; sl += 32) {
                double va[8], vb[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    va[u] = handoff_load(base + (size_t)(sl + 4 * u) * 2 * C + c);
                    vb[u] = handoff_load(base + (size_t)(sl + 4 * u) * 2 * C + C + c);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { a += va[u]; b += vb[u]; }
            }
            for (; sl < nsl; sl += 4) {
                a += handoff_load(base + (size_t)sl * 2 * C + c);
                b += handoff_load(base + (size_t)sl * 2 * C + C + c);
            }
        }
        __syncthreads();
        s1[tl][cl] = a; s2[tl][cl] = b;
        __syncthreads();
        if (tl == 0 && c < C) {
            a = ((s1[0][cl] + s1[1][cl]) + s1[2][cl]) + s1[3][cl];
            b = ((s2[0][cl] + s2[1][cl]) + s2[2][cl]) + s2[3][cl];
            const double mean = a / count;
            double var = b / count - mean * mean;
            if (var < 0.0) var = 0.0;
            const float invstd = (float)(1.0 / sqrt(var + (double)eps));
            const float sc = gamma[c] * invstd;
            scale[gg * C + c] = sc;
            shift[gg * C + c] = beta[c] - (float)mean * sc;
            save_mean[gg * C + c] = (float)mean;
            save_invstd[gg * C + c] = invstd;
            if (running_mean) {
                const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
                running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
                running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
            }
        }
    }
    if (threadIdx.x == 0) handoff_store(&tickets[blockIdx.x], 0u);
}

constexpr int FIN_SL = 32;             // slice lanes of the finalize kernels: 32 channels x 32 slices = 1024 threads (short dependent-load chains)

template <class TIn>
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const TIn* stats, int mtiles, int C, double count, const float* gamma, const float* beta,
                                                          float eps, float momentum, float* running_mean, float* running_var, float* scale,
                                                          float* shift, float* save_mean, float* save_invstd, int groups) {
    __shared__ double s1[FIN_SL][32], s2[FIN_SL][32];
    const int cl = threadIdx.x & 31, part = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
    for (int g = 0; g < groups; ++g) {          // groups are finalised in order: running statistics see pass 0, then pass 1, ...
        const TIn* st = stats + (size_t)g * mtiles * 2 * C;
        double a = 0.0, b = 0.0;
        if (c < C)
            for (int t = part; t < mtiles; t += FIN_SL) {
                a += (double)st[((size_t)t * 2 + 0) * C + c];
                b += (double)st[((size_t)t * 2 + 1) * C + c];
            }
        s1[part][cl] = a; s2[part][cl] = b;
        __syncthreads();
        if (part == 0 && c < C) {
            for (int k = 1; k < FIN_SL; ++k) { a += s1[k][cl]; b += s2[k][cl]; }
            const double mean = a / count;
            double var = b / count - mean * mean;
            if (var < 0.0) var = 0.0;
            const float invstd = (float)(1.0 / sqrt(var + (double)eps));
            const float sc = gamma[c] * invstd;
            scale[g * C + c] = sc;
            shift[g * C + c] = beta[c] - (float)mean * sc;
            save_mean[g * C + c] = (float)mean;
            save_invstd[g * C + c] = invstd;
            if (running_mean) {
                const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
                running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
                running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
            }
        }
        __syncthreads();
    }
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const f32x4* x, const f32x4* scale, const f32x4* shift, const f32x4* res, int relu, size_t n4, int C4, f32x4* y,
                                                       size_t group4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) + (int)(i / group4) * C4;
        f32x4 v = x[i] * scale[c] + shift[c];
        if (res) v += res[i];
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        y[i] = v;
    }
}

__device__ __forceinline__ f32x4 relu_mask(f32x4 g, f32x4 y) {
    g.x = y.x > 0.f ? g.x : 0.f; g.y = y.y > 0.f ? g.y : 0.f; g.z = y.z > 0.f ? g.z : 0.f; g.w = y.w > 0.f ? g.w : 0.f;
    return g;
}

// Per-block partial sums of dz and dz * xhat.  G = C/4 channel groups; a block covers PL = 256 / min(G,256) pixel lanes.
constexpr int BNR_BLOCKS = 1024;
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const f32x4* dy, const f32x4* yact, const f32x4* x, const f32x4* mean, const f32x4* invstd,
                                                            int relu, size_t n_pix /* per group */, int C4, float* part /* [groups][blocks][2][C] */) {
    __shared__ f32x4 sh[2][256];
    const int grp = blockIdx.y;
    dy += (size_t)grp * n_pix * C4; x += (size_t)grp * n_pix * C4;
    if (yact) yact += (size_t)grp * n_pix * C4;
    mean += (size_t)grp * C4; invstd += (size_t)grp * C4;
    part += (size_t)grp * gridDim.x * 2 * C4 * 4;
    const size_t per = (n_pix + gridDim.x - 1) / gridDim.x;
    const size_t pb = (size_t)blockIdx.x * per, pe = pb + per < n_pix ? pb + per : n_pix;
    for (int g0 = 0; g0 < C4; g0 += 256) {
        const int G = C4 - g0 < 256 ? C4 - g0 : 256;
        const int PL = 256 / G;
        const int cg = threadIdx.x % G, pl = threadIdx.x / G;
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
        if (pl < PL) {
            const f32x4 mu = mean[g0 + cg], is = invstd[g0 + cg];
            // four pixels per pass, all their loads issued before the first use (a pass of one pixel left three loads in flight per lane)
            size_t m = pb + pl;
            for (; m + 3 * (size_t)PL < pe; m += 4 * (size_t)PL) {
                f32x4 g[4], xv[4], ya[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const size_t i = (m + (size_t)u * PL) * C4 + g0 + cg;
                    g[u] = dy[i];
                    xv[u] = x[i];
                    if (relu) ya[u] = yact[i];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (relu) g[u] = relu_mask(g[u], ya[u]);
                    s1 += g[u];
                    s2 += g[u] * ((xv[u] - mu) * is);
                }
            }
            for (; m < pe; m += PL) {
                const size_t i = m * C4 + g0 + cg;
                f32x4 g = dy[i];
                if (relu) g = relu_mask(g, yact[i]);
                s1 += g;
                s2 += g * ((x[i] - mu) * is);
            }
        }
        sh[0][threadIdx.x] = s1; sh[1][threadIdx.x] = s2;
        __syncthreads();
        if (threadIdx.x < G) {
            f32x4 a = sh[0][threadIdx.x], b = sh[1][threadIdx.x];
            for (int k = 1; k < PL; ++k) { a += sh[0][threadIdx.x + k * G]; b += sh[1][threadIdx.x + k * G]; }
            float* o = part + (size_t)blockIdx.x * 2 * C4 * 4;
            *reinterpret_cast<f32x4*>(o + (size_t)(g0 + threadIdx.x) * 4) = a;
            *reinterpret_cast<f32x4*>(o + (size_t)C4 * 4 + (size_t)(g0 + threadIdx.x) * 4) = b;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* part, int nblk, int C, float* dgamma, float* dbeta, int accumulate, float* sums,
                                                              int groups) {
    __shared__ double s1[FIN_SL][32], s2[FIN_SL][32];
    const int cl = threadIdx.x & 31, slice = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
    double ta = 0.0, tb = 0.0;
    for (int g = 0; g < groups; ++g) {
        const float* pg = part + (size_t)g * nblk * 2 * C;
        double a = 0.0, b = 0.0;
        if (c < C) {
            int k = slice;
            for (; k + 7 * FIN_SL < nblk; k += 8 * FIN_SL) {          // eight blocks' partials in flight, added in block order
                float va[8], vb[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    va[u] = pg[(size_t)(k + u * FIN_SL) * 2 * C + c];
                    vb[u] = pg[(size_t)(k + u * FIN_SL) * 2 * C + C + c];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { a += (double)va[u]; b += (double)vb[u]; }
            }
            for (; k < nblk; k += FIN_SL) {
                a += (double)pg[(size_t)k * 2 * C + c];
                b += (double)pg[(size_t)k * 2 * C + C + c];
            }
        }
        s1[slice][cl] = a; s2[slice][cl] = b;
        __syncthreads();
        if (slice == 0 && c < C) {
            for (int k = 1; k < FIN_SL; ++k) { a += s1[k][cl]; b += s2[k][cl]; }
            sums[(size_t)g * 2 * C + c] = (float)a;
            sums[(size_t)g * 2 * C + C + c] = (float)b;
            ta += a; tb += b;
        }
        __syncthreads();
    }
    if (slice == 0 && c < C) {
        if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)ta : (float)ta;
        if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)tb : (float)tb;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const f32x4* dy, const f32x4* yact, const f32x4* x, const f32x4* gamma, const f32x4* mean,
                                                           const f32x4* invstd, const f32x4* sums, int relu, size_t n4, int C4, float inv_count,
                                                           f32x4* dx, f32x4* dres, int dres_acc, size_t group4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4), grp = (int)(i / group4);
        f32x4 g = dy[i];
        if (relu) g = relu_mask(g, yact[i]);
        if (dres) dres[i] = dres_acc ? dres[i] + g : g;
        const f32x4 is = invstd[grp * C4 + c];
        const f32x4 xh = (x[i] - mean[grp * C4 + c]) * is;
        const f32x4* sg = sums + (size_t)grp * 2 * C4;
        dx[i] = (gamma[c] * is) * (g - sg[c] * inv_count - xh * (sg[C4 + c] * inv_count));
    }
}

// ---------------------------------------------------------------------------------------------- MaxPool 3x3 s2 p1
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* x, int B, int H, int W, int C, int Ho, int Wo, float* y, uint8_t* idx) {
    const int C4 = C >> 2;
    const size_t total = (size_t)B * Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        size_t r = i / C4;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const float ninf = -__builtin_huge_valf();
        f32x4 best = {ninf, ninf, ninf, ninf};
        int bi[4] = {0, 0, 0, 0};
        bool first[4] = {true, true, true, true};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((size_t)(b * H + iy) * W + ix) * C + c * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // torch's rule: take a new value when it is greater or NaN; the first in-bounds element initialises
                    if (first[e] || v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = ky * 3 + kx; first[e] = false; }
                }
            }
        }
        *reinterpret_cast<f32x4*>(y + i * 4) = best;
        uchar4 o;
        o.x = (uint8_t)bi[0]; o.y = (uint8_t)bi[1]; o.z = (uint8_t)bi[2]; o.w = (uint8_t)bi[3];
        *reinterpret_cast<uchar4*>(idx + i * 4) = o;
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* dy, const uint8_t* idx, int B, int H, int W, int C, int Ho, int Wo, float* dx, int accumulate) {
    const int C4 = C >> 2;
    const size_t total = (size_t)B * H * W * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        size_t r = i / C4;
        const int ix = (int)(r % W); r /= W;
        const int iy = (int)(r % H);
        const int b = (int)(r / H);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        // windows (oy, ox) with oy*2-1 <= iy <= oy*2+1
        const int oy_lo = iy >> 1, oy_hi = (iy + 1) >> 1;     // ceil((iy-1)/2) = iy>>1 for iy >= 0 ; floor((iy+1)/2)
        const int ox_lo = ix >> 1, ox_hi = (ix + 1) >> 1;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            if (oy >= Ho) continue;
            const int ky = iy - (oy * 2 - 1);
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                if (ox >= Wo) continue;
                const int kx = ix - (ox * 2 - 1);
                const int tap = ky * 3 + kx;
                const size_t o = ((size_t)(b * Ho + oy) * Wo + ox) * C + c * 4;
                const uchar4 k = *reinterpret_cast<const uchar4*>(idx + o);
                const f32x4 d = *reinterpret_cast<const f32x4*>(dy + o);
                if (k.x == tap) g.x += d.x;
                if (k.y == tap) g.y += d.y;
                if (k.z == tap) g.z += d.z;
                if (k.w == tap) g.w += d.w;
            }
        }
        f32x4* out = reinterpret_cast<f32x4*>(dx + i * 4);
        *out = accumulate ? *out + g : g;
    }
}

// ---------------------------------------------------------------------------------------------- elementwise
__device__ __forceinline__ float dact(float y, int act) {
    if (act == MCAV_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == MCAV_ACT_ELU) return y > 0.f ? 1.f : y + 1.f;
    if (act == MCAV_ACT_SIGMOID) return y * (1.f - y);
    return 1.f;
}

__global__ void act_bwd_kernel(const float* dy, const float* y, int act, size_t n, float* dx, int accumulate) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = dy[i] * dact(y[i], act);
        dx[i] = accumulate ? dx[i] + v : v;
    }
}

__global__ void act_bwd_strided_kernel(const float* dy, const float* y, int act, size_t n, float* dx, int stride) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dx[i * stride] = dy[i] * dact(y[i], act);
}

__global__ void add_kernel(const float* a, const float* b, size_t n, float* out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = a[i] + b[i];
}

__global__ void spatial_mean_kernel(const float* x, int B, int n_pix, int C, float scale, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    float s = 0.f;
    for (int p = 0; p < n_pix; ++p) s += x[((size_t)b * n_pix + p) * C + c];
    out[i] = scale * (s / (float)n_pix);
}

__global__ void spatial_mean_bwd_kernel(const float* dout, int B, int n_pix, int C, float scale, float* dx) {
    const size_t total = (size_t)B * n_pix * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int b = (int)(i / ((size_t)n_pix * C));
        dx[i] = dout[b * C + c] * (scale / (float)n_pix);
    }
}

// ---------------------------------------------------------------------------------------------- Adam
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                                                   float bc1, float bc2_sqrt, float gscale) {
    const float step = lr / bc1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gr = g[i] * gscale;
        const float mi = m[i] + (gr - m[i]) * (1.f - b1);     // exp_avg.lerp_(grad, 1 - beta1), as torch.optim.Adam
        const float vi = b2 * v[i] + (1.f - b2) * gr * gr;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step * (mi / denom);
    }
}

// Capturable form (hipGraph): the step count, learning rate and gradient scale live in a DEVICE record, so the same two launches are valid
// for every replay.  state = { step (as float, exact to 2^24), lr, grad_scale, bias_correction1, sqrt(bias_correction2), 0, 0, 0 }.
__global__ void adam_prepare_kernel(float* st, float b1, float b2) {
    const float step = st[0] + 1.f;
    st[0] = step;
    st[3] = (float)(1.0 - pow((double)b1, (double)step));
    st[4] = (float)sqrt(1.0 - pow((double)b2, (double)step));
}

__global__ __launch_bounds__(256) void adam_dev_kernel(float* p, const float* g, float* m, float* v, size_t n, float b1, float b2, float eps,
                                                       const float* __restrict__ st) {
    const float lr = st[1], gscale = st[2], bc1 = st[3], bc2_sqrt = st[4];
    const float step = lr / bc1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gr = g[i] * gscale;
        const float mi = m[i] + (gr - m[i]) * (1.f - b1);
        const float vi = b2 * v[i] + (1.f - b2) * gr * gr;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step * (mi / denom);
    }
}

}  // namespace mcav

using namespace mcav;

MCAV_EXPORT int mcav_nchw_to_nhwc(const float* src, int B, int C, int H, int W, float* dst, int Cp, int choff, void* stream) {
    if (!src || !dst || B <= 0 || C <= 0 || H <= 0 || W <= 0 || choff < 0 || choff + C > Cp) return MCAV_E_INVALID;
    if (Cp == 4 && choff == 0) nchw_to_nhwc4_kernel<<<grid_for((size_t)B * H * W), 256, 0, as_stream(stream)>>>(src, B, C, H, W, dst);
    else nchw_to_nhwc_kernel<<<grid_for((size_t)B * H * W), 256, 0, as_stream(stream)>>>(src, B, C, H, W, dst, Cp, choff);
    return launch_status();
}

MCAV_EXPORT int mcav_nchw3_to_nhwc(const float* s0, const float* s1, const float* s2, int B, int C, int H, int W, float* dst, int Cp, void* stream) {
    if (!s0 || !s1 || !s2 || !dst || B <= 0 || C <= 0 || H <= 0 || W <= 0 || (Cp & 3) || 3 * C > Cp) return MCAV_E_INVALID;
    nchw3_to_nhwc_kernel<<<grid_for((size_t)B * H * W * (Cp >> 2)), 256, 0, as_stream(stream)>>>(s0, s1, s2, B, C, H, W, dst, Cp);
    return launch_status();
}

MCAV_EXPORT int mcav_nhwc_to_nchw(const float* src, int B, int C, int H, int W, int Cp, int choff, float* dst, void* stream) {
    if (!src || !dst || B <= 0 || C <= 0 || H <= 0 || W <= 0 || choff < 0 || choff + C > Cp) return MCAV_E_INVALID;
    nhwc_to_nchw_kernel<<<grid_for((size_t)B * H * W), 256, 0, as_stream(stream)>>>(src, B, C, H, W, Cp, choff, dst);
    return launch_status();
}

MCAV_EXPORT size_t mcav_bn_finalize_workspace_bytes(int mtiles, int C, int groups) {
    if (groups < 1) groups = 1;
    if (mtiles <= BNF_DIRECT || C <= 0) return 0;
    // one completion ticket per 64-channel column (the caller zero-fills a workspace when it allocates it; the kernel leaves the tickets at
    // zero) + the slices' partial sums.  The tickets sit at a FIXED place and size, whatever C: a cached workspace serves layers of different
    // widths, and a ticket word that another layer's partial sums had covered would not be zero (the first version put them behind the
    // partials: every large-shape test failed).
    if (C > 64 * BNF_TICKETS) return 0;
    return sizeof(unsigned) * BNF_TICKETS + align_up(sizeof(double) * 2 * (size_t)C * BNF_SLICES * groups, 256);
}

MCAV_EXPORT int mcav_bn_finalize(const float* stats, int mtiles, int C, double count, const float* gamma, const float* beta, float eps, float momentum,
                                 float* running_mean, float* running_var, float* scale, float* shift, float* save_mean, float* save_invstd,
                                 int groups, void* workspace, size_t workspace_bytes, void* stream) {
    if (groups < 1) groups = 1;
    if (!stats || !gamma || !beta || !scale || !shift || !save_mean || !save_invstd || mtiles <= 0 || C <= 0 || count <= 0) return MCAV_E_INVALID;
    if ((running_mean == nullptr) != (running_var == nullptr)) return MCAV_E_INVALID;
    hipStream_t s = as_stream(stream);
    if (mtiles <= BNF_DIRECT) {
        bn_finalize_kernel<float><<<(C + 31) / 32, 32 * FIN_SL, 0, s>>>(stats, mtiles, C, count, gamma, beta, eps, momentum, running_mean, running_var, scale,
                                                                shift, save_mean, save_invstd, groups);
        return launch_status();
    }
    if (!workspace || workspace_bytes < mcav_bn_finalize_workspace_bytes(mtiles, C, groups)) return MCAV_E_WORKSPACE;
    int per_slice = (mtiles + BNF_SLICES - 1) / BNF_SLICES;
    if (per_slice < 8) per_slice = 8;
    const int slices = (mtiles + per_slice - 1) / per_slice;
    unsigned* tickets = reinterpret_cast<unsigned*>(workspace);
    double* part = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + sizeof(unsigned) * BNF_TICKETS);
    bn_partial_finalize_kernel<<<dim3((C + 63) / 64, slices, groups), 256, 0, s>>>(stats, mtiles, C, per_slice, part, tickets, count, gamma, beta, eps, momentum,
                                                                                   running_mean, running_var, scale, shift, save_mean, save_invstd);
    return launch_status();
}

MCAV_EXPORT int mcav_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps, int C,
                                    float* scale, float* shift, void* stream) {
    if (!gamma || !beta || !running_mean || !running_var || !scale || !shift || C <= 0) return MCAV_E_INVALID;
    bn_eval_coeffs_kernel<<<(C + 63) / 64, 64, 0, as_stream(stream)>>>(gamma, beta, running_mean, running_var, eps, C, scale, shift);
    return launch_status();
}

MCAV_EXPORT int mcav_bn_apply(const float* x, const float* scale, const float* shift, const float* residual, int act, size_t n_pix, int C, float* y,
                              size_t pix_per_group, void* stream) {
    if (!x || !scale || !shift || !y || C <= 0 || (C & 3) || (act != MCAV_ACT_NONE && act != MCAV_ACT_RELU)) return MCAV_E_INVALID;
    if (n_pix == 0) return MCAV_OK;
    const size_t n4 = n_pix * (size_t)(C / 4);
    bn_apply_kernel<<<grid_for(n4), 256, 0, as_stream(stream)>>>((const f32x4*)x, (const f32x4*)scale, (const f32x4*)shift, (const f32x4*)residual,
                                                                 act == MCAV_ACT_RELU, n4, C / 4, (f32x4*)y,
                                                                 (pix_per_group ? pix_per_group : n_pix) * (size_t)(C / 4));
    return launch_status();
}

MCAV_EXPORT size_t mcav_bn_bwd_workspace_bytes(size_t n_pix, int C, int groups) {
    (void)n_pix;
    if (groups < 1) groups = 1;
    return C > 0 ? align_up(sizeof(float) * 2 * (size_t)C * BNR_BLOCKS * groups, 256) : 0;
}

MCAV_EXPORT int mcav_bn_bwd_reduce(const float* dy, const float* y_act, const float* x, const float* save_mean, const float* save_invstd, int relu,
                                   size_t n_pix, int C, float* dgamma, float* dbeta, int accumulate, float* sums, int groups, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    if (groups < 1) groups = 1;
    if (!dy || !x || !save_mean || !save_invstd || !sums || !workspace || C <= 0 || (C & 3) || n_pix == 0 || n_pix % groups) return MCAV_E_INVALID;
    if (relu && !y_act) return MCAV_E_INVALID;
    const int C4 = C / 4;
    if (C4 < 256 && (256 % C4) != 0) return MCAV_E_INVALID;
    if (C4 > 256 && (C4 % 256) != 0) return MCAV_E_INVALID;
    if (workspace_bytes < mcav_bn_bwd_workspace_bytes(n_pix, C, groups)) return MCAV_E_WORKSPACE;
    const size_t pg = n_pix / groups;
    int blocks = (int)((pg + 31) / 32 < (size_t)BNR_BLOCKS ? (pg + 31) / 32 : (size_t)BNR_BLOCKS);
    float* part = reinterpret_cast<float*>(workspace);
    hipStream_t s = as_stream(stream);
    bn_bwd_reduce_kernel<<<dim3(blocks, groups), 256, 0, s>>>((const f32x4*)dy, (const f32x4*)y_act, (const f32x4*)x, (const f32x4*)save_mean,
                                                              (const f32x4*)save_invstd, relu, pg, C4, part);
    bn_bwd_finalize_kernel<<<(C + 31) / 32, 32 * FIN_SL, 0, s>>>(part, blocks, C, dgamma, dbeta, accumulate, sums, groups);
    return launch_status();
}

MCAV_EXPORT int mcav_bn_bwd_finalize(const float* partial, int nblk, int C, float* dgamma, float* dbeta, int accumulate, float* sums, int groups, void* stream) {
    if (groups < 1) groups = 1;
    if (!partial || !sums || nblk <= 0 || C <= 0) return MCAV_E_INVALID;
    bn_bwd_finalize_kernel<<<(C + 31) / 32, 32 * FIN_SL, 0, as_stream(stream)>>>(partial, nblk, C, dgamma, dbeta, accumulate, sums, groups);
    return launch_status();
}

MCAV_EXPORT int mcav_bn_bwd_apply(const float* dy, const float* y_act, const float* x, const float* gamma, const float* save_mean,
                                  const float* save_invstd, const float* sums, int relu, size_t n_pix, int C, float* dx, float* dres,
                                  int dres_accumulate, int groups, void* stream) {
    if (groups < 1) groups = 1;
    if (!dy || !x || !gamma || !save_mean || !save_invstd || !sums || !dx || C <= 0 || (C & 3) || n_pix == 0 || n_pix % groups) return MCAV_E_INVALID;
    if (relu && !y_act) return MCAV_E_INVALID;
    const size_t n4 = n_pix * (size_t)(C / 4);
    bn_bwd_apply_kernel<<<grid_for(n4), 256, 0, as_stream(stream)>>>((const f32x4*)dy, (const f32x4*)y_act, (const f32x4*)x, (const f32x4*)gamma,
                                                                     (const f32x4*)save_mean, (const f32x4*)save_invstd, (const f32x4*)sums, relu, n4,
                                                                     C / 4, (float)(1.0 / (double)(n_pix / groups)), (f32x4*)dx, (f32x4*)dres, dres_accumulate,
                                                                     (n_pix / groups) * (size_t)(C / 4));
    return launch_status();
}

MCAV_EXPORT int mcav_maxpool3s2_fwd(const float* x, int B, int H, int W, int C, float* y, uint8_t* idx, void* stream) {
    if (!x || !y || !idx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return MCAV_E_INVALID;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    maxpool_fwd_kernel<<<grid_for((size_t)B * Ho * Wo * (C / 4)), 256, 0, as_stream(stream)>>>(x, B, H, W, C, Ho, Wo, y, idx);
    return launch_status();
}

MCAV_EXPORT int mcav_maxpool3s2_bwd(const float* dy, const uint8_t* idx, int B, int H, int W, int C, float* dx, int accumulate, void* stream) {
    if (!dy || !idx || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return MCAV_E_INVALID;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    maxpool_bwd_kernel<<<grid_for((size_t)B * H * W * (C / 4)), 256, 0, as_stream(stream)>>>(dy, idx, B, H, W, C, Ho, Wo, dx, accumulate);
    return launch_status();
}

MCAV_EXPORT int mcav_act_bwd(const float* dy, const float* y, int act, size_t n, float* dx, int accumulate, void* stream) {
    if (!dy || !y || !dx) return MCAV_E_INVALID;
    if (n == 0) return MCAV_OK;
    act_bwd_kernel<<<grid_for(n), 256, 0, as_stream(stream)>>>(dy, y, act, n, dx, accumulate);
    return launch_status();
}

MCAV_EXPORT int mcav_act_bwd_strided(const float* dy, const float* y, int act, size_t n, float* dx, int stride, void* stream) {
    if (!dy || !y || !dx || stride < 1) return MCAV_E_INVALID;
    if (n == 0) return MCAV_OK;
    act_bwd_strided_kernel<<<grid_for(n), 256, 0, as_stream(stream)>>>(dy, y, act, n, dx, stride);
    return launch_status();
}

MCAV_EXPORT int mcav_add(const float* a, const float* b, size_t n, float* out, void* stream) {
    if (!a || !b || !out) return MCAV_E_INVALID;
    if (n == 0) return MCAV_OK;
    add_kernel<<<grid_for(n), 256, 0, as_stream(stream)>>>(a, b, n, out);
    return launch_status();
}

MCAV_EXPORT int mcav_spatial_mean(const float* x, int B, int n_pix, int C, float scale, float* out, void* stream) {
    if (!x || !out || B <= 0 || n_pix <= 0 || C <= 0) return MCAV_E_INVALID;
    spatial_mean_kernel<<<(B * C + 63) / 64, 64, 0, as_stream(stream)>>>(x, B, n_pix, C, scale, out);
    return launch_status();
}

MCAV_EXPORT int mcav_spatial_mean_bwd(const float* dout, int B, int n_pix, int C, float scale, float* dx, void* stream) {
    if (!dout || !dx || B <= 0 || n_pix <= 0 || C <= 0) return MCAV_E_INVALID;
    spatial_mean_bwd_kernel<<<grid_for((size_t)B * n_pix * C), 256, 0, as_stream(stream)>>>(dout, B, n_pix, C, scale, dx);
    return launch_status();
}

MCAV_EXPORT int mcav_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1, float beta2,
                               float eps, int step, float grad_scale, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || step < 1) return MCAV_E_INVALID;
    if (n == 0) return MCAV_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    adam_kernel<<<grid_for(n, 256, 8192), 256, 0, as_stream(stream)>>>(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, (float)bc1,
                                                                       (float)sqrt(bc2), grad_scale);
    return launch_status();
}

MCAV_EXPORT int mcav_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float beta1, float beta2, float eps,
                                   float* state8, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !state8) return MCAV_E_INVALID;
    if (n == 0) return MCAV_OK;
    adam_prepare_kernel<<<1, 1, 0, as_stream(stream)>>>(state8, beta1, beta2);
    adam_dev_kernel<<<grid_for(n, 256, 8192), 256, 0, as_stream(stream)>>>(param, grad, exp_avg, exp_avg_sq, n, beta1, beta2, eps, state8);
    return launch_status();
}

// ---------------------------------------------------------------------------------------------- conv-kernel timer (kernel_timer.h)
#include <mutex>
#include <vector>

#include "kernel_timer.h"

namespace mcav {
static std::mutex g_kt_mutex;                       // forward launches come from the caller's thread, backward ones from autograd's
static bool g_kt_on = false;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_kt_events;

bool kernel_timer_on() { return g_kt_on; }
void kernel_timer_add(hipEvent_t e0, hipEvent_t e1) {
    std::lock_guard<std::mutex> lock(g_kt_mutex);
    g_kt_events.emplace_back(e0, e1);
}
}  // namespace mcav

MCAV_EXPORT int mcav_kernel_timer_begin(void) {
    std::lock_guard<std::mutex> lock(mcav::g_kt_mutex);
    for (auto& e : mcav::g_kt_events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    mcav::g_kt_events.clear();
    mcav::g_kt_on = true;
    return MCAV_OK;
}

MCAV_EXPORT int mcav_kernel_timer_count(void) {
    std::lock_guard<std::mutex> lock(mcav::g_kt_mutex);
    return (int)mcav::g_kt_events.size();
}

MCAV_EXPORT int mcav_kernel_timer_end(float* ms, int capacity) {
    std::lock_guard<std::mutex> lock(mcav::g_kt_mutex);
    mcav::g_kt_on = false;
    const int n = (int)mcav::g_kt_events.size();
    int rc = MCAV_OK;
    for (int i = 0; i < n; ++i) {
        auto& e = mcav::g_kt_events[i];
        float t = 0.f;
        if (hipEventSynchronize(e.second) != hipSuccess || hipEventElapsedTime(&t, e.first, e.second) != hipSuccess) rc = MCAV_E_LAUNCH;
        if (ms && i < capacity) ms[i] = t;
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    mcav::g_kt_events.clear();
    return rc == MCAV_OK ? n : rc;
}
