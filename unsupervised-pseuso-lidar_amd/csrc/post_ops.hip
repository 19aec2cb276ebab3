// The steps AFTER the training-step hot path, both HBM-bound single passes:
//   * depth metrics      : reference evaluate.py:6-39 (compute_errors) -- ten masked sums in one pass over gt and the disparity
//   * depth -> pseudo-LiDAR: reference pseudo-lidar/utils/PseudoLiDAR.py:69-110 (project_PL) -- un-project, rigid transform to
//                            the velodyne frame, x >= 0 & z < 1 m filter, every sparsity-th survivor: an ORDER-PRESERVING
//                            stream compaction (count -> scan -> scatter), float64 like the reference's numpy arithmetic.
// Reductions are two-stage with a fixed order (bit-reproducible); no float atomics.
#include "mcav_common.h"

namespace mcav {

// ---------------------------------------------------------------------------------------------- depth metrics
// sums: 0 d1, 1 d2, 2 d3 (counts), 3 (gt-pred)^2, 4 (ln gt - ln pred)^2, 5 |gt-pred|/gt, 6 (gt-pred)^2/gt, 7 err, 8 err^2, 9 |log10|,
//       10 number of elements taken
constexpr int NMET = 11;
constexpr int MET_BLOCKS = 2048;

typedef float mf32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void met_accum(float g, float dsp, float min_gt, double (&s)[NMET]) {
    if (!(g > min_gt)) return;                              // min_gt < 0: every element, as the reference
    const float p = 1.0f / (10.0f * dsp + 0.01f);           // pose_geometry.py:82-83
    const float t = fmaxf(g / p, p / g);
    s[0] += t < 1.25f ? 1.0 : 0.0;
    s[1] += t < 1.25f * 1.25f ? 1.0 : 0.0;
    s[2] += t < 1.25f * 1.25f * 1.25f ? 1.0 : 0.0;
    const float d = g - p;
    const float lg = logf(g), lp = logf(p);
    s[3] += (double)(d * d);
    s[4] += (double)((lg - lp) * (lg - lp));
    s[5] += (double)(fabsf(d) / g);
    s[6] += (double)(d * d / g);
    const float e = lp - lg;
    s[7] += (double)e;
    s[8] += (double)(e * e);
    s[9] += (double)fabsf(log10f(p) - log10f(g));
    s[10] += 1.0;
}

// 16-byte loads, two of each array in flight per lane and iteration (the pass is latency-bound otherwise: 8 B per element)
__global__ __launch_bounds__(256) void depth_metrics_partial_kernel(const float* __restrict__ gt, const float* __restrict__ disp, size_t n,
                                                                    float min_gt, double* __restrict__ part) {
    __shared__ double red[4][NMET];
    double s[NMET];
#pragma unroll
    for (int k = 0; k < NMET; ++k) s[k] = 0.0;
    const size_t n4 = n / 4, stride = (size_t)gridDim.x * 256;
    const mf32x4* g4 = reinterpret_cast<const mf32x4*>(gt);
    const mf32x4* d4 = reinterpret_cast<const mf32x4*>(disp);
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {
        const mf32x4 ga = g4[i], da = d4[i], gb = g4[i + stride], db = d4[i + stride];
#pragma unroll
        for (int c = 0; c < 4; ++c) { met_accum(ga[c], da[c], min_gt, s); met_accum(gb[c], db[c], min_gt, s); }
    }
    if (i < n4) {
        const mf32x4 ga = g4[i], da = d4[i];
#pragma unroll
        for (int c = 0; c < 4; ++c) met_accum(ga[c], da[c], min_gt, s);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) met_accum(gt[n4 * 4 + threadIdx.x], disp[n4 * 4 + threadIdx.x], min_gt, s);      // ragged tail
#pragma unroll
    for (int k = 0; k < NMET; ++k) {
        double v = s[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < NMET) part[(size_t)blockIdx.x * NMET + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// out[9] in the reference's key order: silog, abs_rel, log10, rms, sq_rel, log_rms, d1, d2, d3; out[9] = element count
__global__ __launch_bounds__(256) void depth_metrics_finalize_kernel(const double* part, int blocks, float* out) {
    __shared__ double red[4][NMET];
    __shared__ double tot[NMET];
    double s[NMET];
#pragma unroll
    for (int k = 0; k < NMET; ++k) s[k] = 0.0;
    for (int b = threadIdx.x; b < blocks; b += 256)
#pragma unroll
        for (int k = 0; k < NMET; ++k) s[k] += part[(size_t)b * NMET + k];
#pragma unroll
    for (int k = 0; k < NMET; ++k) {
        double v = s[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < NMET) tot[threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    __syncthreads();
    if (threadIdx.x == 0) {
        const double n = tot[10] > 0.0 ? tot[10] : 1.0;
        const double me = tot[7] / n, me2 = tot[8] / n;
        double var = me2 - me * me;
        if (var < 0.0) var = 0.0;
        out[0] = (float)(sqrt(var) * 100.0);
        out[1] = (float)(tot[5] / n);
        out[2] = (float)(tot[9] / n);
        out[3] = (float)sqrt(tot[3] / n);
        out[4] = (float)(tot[6] / n);
        out[5] = (float)sqrt(tot[4] / n);
        out[6] = (float)(tot[0] / n);
        out[7] = (float)(tot[1] / n);
        out[8] = (float)(tot[2] / n);
        out[9] = (float)tot[10];
    }
}

// ---------------------------------------------------------------------------------------------- pseudo-LiDAR
struct PLCalib {
    double cu, cv, fu, fv, bx, by;       // from P_rect_02 (PseudoLiDAR.py:78-83)
    double ti[3][4];                     // rows 0..2 of inverse_rigid_trans(T) (PseudoLiDAR.py:39-46); its 4th row is zero
};

constexpr int PL_THREADS = 256;

#pragma clang fp contract(off)           // the reference is numpy float64: keep mul / add separate as it does
__device__ __forceinline__ bool pl_point(const float* depth, int cols, const PLCalib& c, size_t i, double (&q)[3]) {
    const int r = (int)(i / cols), cc = (int)(i - (size_t)r * cols);
    const double d = (double)depth[i];
    const double x = (((double)cc - c.cu) * d) / c.fu + c.bx;
    const double y = (((double)r - c.cv) * d) / c.fv + c.by;
#pragma unroll
    for (int j = 0; j < 3; ++j) q[j] = ((x * c.ti[j][0] + y * c.ti[j][1]) + d * c.ti[j][2]) + c.ti[j][3];
    return q[0] >= 0.0 && q[2] < 1.0;
}

__global__ __launch_bounds__(PL_THREADS) void pl_count_kernel(const float* depth, int cols, size_t n, PLCalib c, unsigned* counts) {
    __shared__ unsigned wsum[PL_THREADS / 64];
    const size_t i = (size_t)blockIdx.x * PL_THREADS + threadIdx.x;
    double q[3];
    const bool v = i < n && pl_point(depth, cols, c, i, q);
    const unsigned long long m = __ballot(v);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (unsigned)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// exclusive scan of the per-block counts, one block (the image has at most a few thousand blocks); total -> counts[nblocks]
__global__ __launch_bounds__(1024) void pl_scan_kernel(unsigned* counts, int nblocks) {
    __shared__ unsigned sh[1024];
    __shared__ unsigned carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += 1024) {
        const int i = base + threadIdx.x;
        const unsigned v = i < nblocks ? counts[i] : 0u;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const unsigned t = (int)threadIdx.x >= off ? sh[threadIdx.x - off] : 0u;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nblocks) counts[i] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += sh[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) counts[nblocks] = carry;
}

__global__ __launch_bounds__(PL_THREADS) void pl_scatter_kernel(const float* depth, int cols, size_t n, PLCalib c, const unsigned* offsets,
                                                                int sparsity, double* cloud, size_t capacity) {
    __shared__ unsigned wsum[PL_THREADS / 64];
    const size_t i = (size_t)blockIdx.x * PL_THREADS + threadIdx.x;
    double q[3];
    const bool v = i < n && pl_point(depth, cols, c, i, q);
    const unsigned long long m = __ballot(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wsum[wave] = (unsigned)__popcll(m);
    __syncthreads();
    unsigned before = offsets[blockIdx.x];
    for (int w = 0; w < wave; ++w) before += wsum[w];
    const unsigned k = before + (unsigned)__popcll(m & ((1ull << lane) - 1ull));      // rank of this point among the valid ones, in pixel order
    if (v) {
        const unsigned step = sparsity > 0 ? (unsigned)sparsity : 1u;
        if (k % step == 0) {
            const size_t o = k / step;
            if (o < capacity) {
                cloud[o * 4 + 0] = q[0]; cloud[o * 4 + 1] = q[1]; cloud[o * 4 + 2] = q[2]; cloud[o * 4 + 3] = 0.0;
            }
        }
    }
}

}  // namespace mcav

using namespace mcav;

MCAV_EXPORT size_t mcav_depth_metrics_workspace_bytes(void) { return align_up(sizeof(double) * MET_BLOCKS * NMET, 256); }

MCAV_EXPORT int mcav_depth_metrics(const float* gt, const float* disp, size_t n, float min_gt, float* out10, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    if (!gt || !disp || !out10 || !workspace || n == 0) return MCAV_E_INVALID;
    if (workspace_bytes < mcav_depth_metrics_workspace_bytes()) return MCAV_E_WORKSPACE;
    hipStream_t s = as_stream(stream);
    if ((reinterpret_cast<uintptr_t>(gt) | reinterpret_cast<uintptr_t>(disp)) & 15) return MCAV_E_INVALID;      // 16-byte loads
    size_t blocks = (n / 4 + 511) / 512;              // two 16-byte pieces per lane and iteration
    if (blocks < 1) blocks = 1;
    if (blocks > MET_BLOCKS) blocks = MET_BLOCKS;
    double* part = reinterpret_cast<double*>(workspace);
    depth_metrics_partial_kernel<<<(int)blocks, 256, 0, s>>>(gt, disp, n, min_gt, part);
    depth_metrics_finalize_kernel<<<1, 256, 0, s>>>(part, (int)blocks, out10);
    return launch_status();
}

MCAV_EXPORT size_t mcav_pseudo_lidar_workspace_bytes(int rows, int cols) {
    if (rows <= 0 || cols <= 0) return 0;
    const size_t nblocks = ((size_t)rows * cols + PL_THREADS - 1) / PL_THREADS;
    return align_up(sizeof(unsigned) * (nblocks + 1), 256);
}

// T_velo_to_cam: 4x4 row-major (calib_velo_to_cam R|T); P_rect: 3x4 row-major (calib_cam_to_cam P_rect_02).  Host pointers (16 + 12 doubles).
MCAV_EXPORT int mcav_pseudo_lidar_project(const float* depth, int rows, int cols, const double* T_velo_to_cam, const double* P_rect, int sparsity,
                                          double* cloud, size_t capacity_points, unsigned* count_out_dev, void* workspace, size_t workspace_bytes,
                                          void* stream) {
    if (!depth || !T_velo_to_cam || !P_rect || !cloud || !count_out_dev || !workspace || rows <= 0 || cols <= 0 || sparsity < 0) return MCAV_E_INVALID;
    if (workspace_bytes < mcav_pseudo_lidar_workspace_bytes(rows, cols)) return MCAV_E_WORKSPACE;
    const double* T = T_velo_to_cam;
    const double* P = P_rect;
    PLCalib c;
    c.cu = P[2]; c.cv = P[4 + 2]; c.fu = P[0]; c.fv = P[4 + 1];
    c.bx = P[3] / (-c.fu); c.by = P[4 + 3] / (-c.fv);
    for (int i = 0; i < 3; ++i) {                         // inverse_rigid_trans: [R' | -R' t]
        for (int j = 0; j < 3; ++j) c.ti[i][j] = T[j * 4 + i];
        double acc = 0.0;
        for (int j = 0; j < 3; ++j) acc += -T[j * 4 + i] * T[j * 4 + 3];
        c.ti[i][3] = acc;
    }
    const size_t n = (size_t)rows * cols;
    const int nblocks = (int)((n + PL_THREADS - 1) / PL_THREADS);
    unsigned* counts = reinterpret_cast<unsigned*>(workspace);
    hipStream_t s = as_stream(stream);
    pl_count_kernel<<<nblocks, PL_THREADS, 0, s>>>(depth, cols, n, c, counts);
    pl_scan_kernel<<<1, 1024, 0, s>>>(counts, nblocks);
    pl_scatter_kernel<<<nblocks, PL_THREADS, 0, s>>>(depth, cols, n, c, counts, sparsity, cloud, capacity_points);
    // number of valid points BEFORE sparsification (the caller derives ceil(valid / sparsity) rows)
    hipError_t e = hipMemcpyAsync(count_out_dev, counts + nblocks, sizeof(unsigned), hipMemcpyDeviceToDevice, s);
    return e == hipSuccess ? launch_status() : MCAV_E_LAUNCH;
}
