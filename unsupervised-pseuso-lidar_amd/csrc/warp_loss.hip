// Loss stage of the training step on gfx950: fused inverse-warp -> bilinear sample -> L1 photometric ->
// second-order smoothness, forward and backward in one pass (K10 + K11), plus the standalone geometry
// entry points (inverse_warp, reconstruct, project, disp_to_depth, SSIM).
//
// HBM-bound streaming/gather work (52 B/pixel compulsory for the fused kernel): one thread per
// target pixel, 32x8 pixel tiles so that a wavefront covers two 32-pixel row segments (128-B coalesced plane
// reads), the tgt-depth tile (+2 halo) staged in LDS for the smoothness stencil, bilinear taps served from
// L1/L2 (neighbouring pixels sample neighbouring source texels), deterministic two-level reduction
// (wavefront shuffles -> LDS -> per-block slab -> fp64 finalize) for the 2 loss scalars and the 3x12 dP sums.
#include <stdlib.h>
#include <mutex>
#include <vector>

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
    const void* K;                 // [B,3,3] fp64 (MCAV_WL_K_F64) or fp32
    const float* upstream;         // 2 floats on the device, or nullptr = (1, 1)
    float *d_disp_t, *d_disp_r0, *d_poses, *losses;
    float* slab;                   // [B][nblk][SLAB] per-workgroup partial sums
    double* sample_loss;           // [B][2]
    unsigned* tickets;             // [B + 1], zero between launches: workgroups done per sample; samples done
    int B, H, W;
    int G0, G1;                    // L1 kernel: workgroups per sample of pass 0 (warps 0, 1 + smoothness) / pass 1 (warp 2)
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
template <int N, bool AGENT = false>
__device__ __forceinline__ void block_reduce_store(float* acc, float* dst, float (*sred)[SLAB]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < N; ++k) acc[k] = wave_sum(acc[k]);
    if (lane == 0) {                          // one exec-masked region for the N stores (not one per value)
#pragma unroll
        for (int k = 0; k < N; ++k) sred[wave][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < N) {
        const float v = (sred[0][threadIdx.x] + sred[1][threadIdx.x]) + (sred[2][threadIdx.x] + sred[3][threadIdx.x]);
        if constexpr (AGENT) __hip_atomic_store(dst + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else dst[threadIdx.x] = v;
    }
}

// Branch-free bilinear gathers: an image (3 planes) is a raw buffer resource, a tap outside the image carries an out-of-range offset and
// reads as zero in hardware.  No exec-mask branch and no value merge sits between a load and its use, so the 36 gathers of a pixel's three
// warps are issued back to back behind one counted wait (the predicated form compiled to ~90 exec-masked regions, each with its own wait).
constexpr unsigned WL_OOB = 0x80000000u;

// (the base is wave-uniform by construction; said explicitly, because a 64-bit sample offset formed on the vector ALU made the compiler
//  treat the descriptor as divergent and wrap every gather in a waterfall loop)
__device__ __forceinline__ const float* uniform_ptr(const float* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t image_rsrc(const float* img, size_t plane) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(img)), 0, (unsigned)(3 * plane * sizeof(float)), 0x00020000);
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

// ---------------------------------------------------------------------------------------------- one launch: prologue and epilogue of the fused kernels
// Round 3 folds the two helper launches into the fused kernels (they were 13 of the loss stage's 83 us):
//   * prologue: every workgroup derives its sample's constants itself (lanes 0..2: Rodrigues, P = K [R|t], Q = P[:, :3] K^-1 for one warp each;
//     float64 inverse of K) into LDS, and every wavefront lifts them into scalar registers;
//   * epilogue: per-workgroup partial sums go to the slab; a per-sample ticket names the LAST workgroup of a sample, which sums that sample's
//     slab entries in float64 (fixed order: bit-reproducible), turns dP into the pose gradients and emits the sample's loss sums; a second
//     ticket names the last SAMPLE, which adds the samples up in index order.  Tickets are left at zero for the next launch.
struct alignas(16) SampleFast {
    WarpFast w[3];       // 12 floats each
    float Kinv[12];      // 9 used
};

// Twelve wave-uniform floats from LDS as three 16-byte broadcast reads, NOT hoisted out of the pixel loop (volatile): held in registers for
// the whole loop the three warps' constants cost 45 registers -- as scalars they overflowed the scalar file and the buffer resources with them.
struct F12 { float v[12]; };
__device__ __forceinline__ F12 lds12(const float* p) {
    using f4 = __attribute__((ext_vector_type(4))) float;
    // LDS byte address of p (the low 32 bits of a generic pointer into LDS are its LDS offset); ds_read_b128 through inline asm keeps the
    // loads where they are written -- three per use, nothing held across the loop
    const unsigned addr = (unsigned)reinterpret_cast<unsigned long long>(p);
    F12 r;
    f4 t0, t1, t2;
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %3 offset:32\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(t0), "=&v"(t1), "=&v"(t2) : "v"(addr) : "memory");
    r.v[0] = t0.x; r.v[1] = t0.y; r.v[2] = t0.z; r.v[3] = t0.w;
    r.v[4] = t1.x; r.v[5] = t1.y; r.v[6] = t1.z; r.v[7] = t1.w;
    r.v[8] = t2.x; r.v[9] = t2.y; r.v[10] = t2.z; r.v[11] = t2.w;
    return r;
}
__device__ __forceinline__ WarpFast lds_warp(const WarpFast& w) {
    const F12 r = lds12(w.Q);
    WarpFast u;
#pragma unroll
    for (int i = 0; i < 9; ++i) u.Q[i] = r.v[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) u.p3[i] = r.v[9 + i];
    return u;
}

__device__ __forceinline__ void block_prepare(const WLArgs& a, int b, SampleFast* sf) {
    if (threadIdx.x < 3) {
        const int w = threadIdx.x;
        double Kd[9], Ki[9];
        load_K(a.K, (a.flags & MCAV_WL_K_F64) != 0, b, Kd);
        invert3x3(Kd, Ki);
        float Kf[9], Kinv[9], R[9], t[3], P[12];
        for (int i = 0; i < 9; ++i) { Kf[i] = (float)Kd[i]; Kinv[i] = (float)Ki[i]; }
        pose_to_Rt(a.poses + (size_t)b * 12 + (w == 1 ? 6 : 0), w == 2, R, t);
        make_P(Kf, R, t, P);
        make_fast(P, Kinv, sf->w[w]);
        if (w == 0)
            for (int i = 0; i < 9; ++i) sf->Kinv[i] = Kinv[i];
    }
    __syncthreads();
}

__device__ __forceinline__ float uniform_of(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ WarpFast uniform_warp(const WarpFast& w) {
    WarpFast u;
#pragma unroll
    for (int i = 0; i < 9; ++i) u.Q[i] = uniform_of(w.Q[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) u.p3[i] = uniform_of(w.p3[i]);
    return u;
}

// What one workgroup writes for ANOTHER to read inside the same launch (slab entries, per-sample sums) follows the hand-off protocol of
// mcav_common.h: agent-scope (sc1) stores, an explicit s_waitcnt vmcnt(0) in every thread before the ticket, sc1 loads in the finisher.  The
// first version used __threadfence(), i.e. a write-back and invalidate of the XCD's whole L2 per workgroup: 660 (L1) / 3840 (SSIM) of those per
// launch took the gathers' cached lines with them and doubled the kernels' time.
__device__ __forceinline__ void slab_store(float* p, float v) { handoff_store(p, v); }
__device__ __forceinline__ void order_before_ticket() { handoff_release(); }

constexpr int RED_LD = 256 + 8;      // one pad float per 32 threads: the transposed reads are conflict-free

// Sum N per-thread values over the 256 threads (fixed order) into out[slot(k)]: each value's 256 addends are written to LDS, 8 lanes
// take 32 of them each, three xor-shuffles finish.  ~N + 50 instructions per thread instead of 12 N for N wavefront butterflies.
template <int N, class Slot>
__device__ __forceinline__ void block_sum_to_slab(const float* acc, float (*sred)[RED_LD], float* out, Slot slot) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < N; ++k) sred[k][tid + (tid >> 5)] = acc[k];
    __syncthreads();
    const int k = tid >> 3, part = tid & 7;
    float s = 0.f;
    if (k < N) {
#pragma unroll
        for (int j = 0; j < 32; ++j) s += sred[k][part * 33 + j];
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (k < N && part == 0) slab_store(out + slot(k), s);
}

// after the workgroup's slab entry is written: ticket, and the last workgroup of the sample finishes the sample
__device__ __forceinline__ void block_finish(const WLArgs& a, int b, int nblk, double (*s64)[SLAB], int* s_flag) {
    const int tid = threadIdx.x;
    order_before_ticket();                                       // this workgroup's (agent-scope) slab stores have completed ...
    __syncthreads();
    if (tid == 0) *s_flag = handoff_ticket(&a.tickets[b]) == (unsigned)(nblk - 1);      // ... before its ticket is taken
    __syncthreads();
    if (!*s_flag) return;
    const float* slab = a.slab + (size_t)b * nblk * SLAB;
    constexpr int PARTS = 256 / SLAB;                            // 6 x 40 = 240 threads
    if (tid < PARTS * SLAB) {
        const int part = tid / SLAB, k = tid - part * SLAB;
        double sum = 0.0;
        for (int blk = part; blk < nblk; blk += PARTS)
            sum += (double)handoff_load(slab + (size_t)blk * SLAB + k);      // written by other workgroups
        s64[part][k] = sum;
    }
    __syncthreads();
    if (tid < SLAB) {
        double t = 0.0;
#pragma unroll
        for (int p = 0; p < PARTS; ++p) t += s64[p][tid];
        s64[0][tid] = t;
    }
    __syncthreads();
    if (tid < 3) {                                               // one warp's pose gradient per lane
        double Kd[9];
        load_K(a.K, (a.flags & MCAV_WL_K_F64) != 0, b, Kd);
        float Kf[9];
        for (int i = 0; i < 9; ++i) Kf[i] = (float)Kd[i];
        double g[6];
        pose_grad_from_dP(&s64[0][2 + 12 * tid], Kf, a.poses + (size_t)b * 12 + (tid == 1 ? 6 : 0), tid == 2, g);
        for (int i = 0; i < 6; ++i) s64[1 + tid][i] = g[i];
    }
    __syncthreads();
    if (tid == 0) {
        for (int i = 0; i < 6; ++i) {
            a.d_poses[(size_t)b * 12 + i] = (float)(s64[1][i] + s64[3][i]);      // pose[0]: warp 0 and (through its inverse) warp 2
            a.d_poses[(size_t)b * 12 + 6 + i] = (float)s64[2][i];
        }
        handoff_store(a.sample_loss + b * 2 + 0, s64[0][0]);
        handoff_store(a.sample_loss + b * 2 + 1, s64[0][1]);
        handoff_store(&a.tickets[b], 0u);
        order_before_ticket();
        if (handoff_ticket(&a.tickets[a.B]) == (unsigned)(a.B - 1)) {
            double l0 = 0.0, l1 = 0.0;
            for (int i = 0; i < a.B; ++i) {
                l0 += handoff_load(a.sample_loss + i * 2);
                l1 += handoff_load(a.sample_loss + i * 2 + 1);
            }
            a.losses[0] = (float)l0;
            a.losses[1] = (float)l1;
            handoff_store(&a.tickets[a.B], 0u);
        }
    }
}

// ---------------------------------------------------------------------------------------------- fused warp + L1 + smoothness (round 3)
// The kernel is bound by instruction issue, not by HBM (52 B/pixel against ~1100 vector instructions per pixel in round 2).  What changed:
//   * the per-pixel arithmetic (csrc/warp_math.h, "lean forms"): affine q, reciprocals, nested lerp -- about 500 instructions per pixel;
//   * ONE warp at a time: a workgroup of pass 0 runs warps 0 and 1 (both use depth(tgt) and the target's pixels) and the smoothness term,
//     a workgroup of pass 1 runs warp 2 (depth(ref0)): 26 / 13 running sums instead of 38, <= 128 registers, four wavefronts per SIMD;
//   * the 12 gathers of a warp are issued one unit AHEAD of their use (two register sets: pass 0 alternates the two warps of a pixel,
//     pass 1 the two pixels a thread owns in a tile), the target-aligned values one pixel ahead of that;
//   * a workgroup walks tiles g, g + G, ... of one sample (32 x 16 pixels, two per thread) and reduces its sums ONCE, through LDS;
//   * no helper launches (see above).
constexpr int T2H = 16, T2LH = T2H + 2 * HALO;
constexpr int RED_N0 = 26, RED_N1 = 13;
constexpr int WL_STAGE = 16;      // pixels per thread between two flushes of the staged disparity gradients (see the kernel)

template <bool DBG>
__global__ __launch_bounds__(256, 3) void warp_loss_l1_kernel(WLArgs a) {
    float g0 = 1.0f, g1 = 1.0f;
    if (a.upstream) {
        g0 = a.upstream[0];
        g1 = a.upstream[1];
        if ((a.flags & MCAV_WL_SKIP_IF_UNIT) && g0 == 1.0f && g1 == 1.0f) return;
    }
    __shared__ SampleFast s_sf;
    __shared__ float sD[T2LH][LW + 1];
    __shared__ __attribute__((aligned(16))) float sred[RED_N0][RED_LD];
    __shared__ int s_flag;
    const int H = a.H, W = a.W, b = blockIdx.y, tid = threadIdx.x;
    const bool pass1 = (int)blockIdx.x >= a.G0;
    const int g = pass1 ? (int)blockIdx.x - a.G0 : (int)blockIdx.x, G = pass1 ? a.G1 : a.G0;
    const int ntx = (W + TW - 1) / TW, nty = (H + T2H - 1) / T2H, ntiles = ntx * nty;
    const int nmine = (ntiles - g + G - 1) / G;                  // tiles g, g + G, ... of this sample (the host keeps G <= ntiles)
    const float inv_ntx = 1.0f / (float)ntx;
    block_prepare(a, b, &s_sf);

    // Scalar-register budget: only the GATHERED images are buffer resources (a tap outside the image = an out-of-range offset that reads
    // zero); the pixel-aligned reads and the gradient stores are plain global accesses under the pixel's in-image predicate.  With all seven
    // tensors as resources plus three warps' constants the descriptors spilled into vector registers and every load became a waterfall loop.
    const size_t plane = (size_t)H * W;
    const int pb = (int)(plane * sizeof(float));
    const bool in_depth = (a.flags & MCAV_WL_INPUT_DEPTH) != 0;
    const float invN = 1.0f / (float)((size_t)a.B * 3 * plane);
    const int tx = tid & 31, ty0 = tid >> 5;

    // j-th pixel of this thread: tile j >> 1 of the workgroup's list, upper / lower half of its 16 rows
    auto pixel = [&](int j, int& x, int& y, unsigned& off) {
        const int t = g + (j >> 1) * G;
        const int tyi = (int)(((float)t + 0.5f) * inv_ntx), txi = t - tyi * ntx;
        x = txi * TW + tx;
        y = tyi * T2H + (j & 1) * TH + ty0;
        off = ((j >> 1) < nmine && x < W && y < H) ? (unsigned)((y * W + x) * 4) : WL_OOB;
    };
    auto depth_of = [&](float v) { return in_depth ? v : rcp_nr(fmaf(10.0f, v, 0.01f)); };
    struct Set { FTap t; float q[3][4]; };
    auto issue = [&](const WarpFast& wlds, __amdgpu_buffer_rsrc_t src, int x, int y, float D, bool live, Set& s) {
        s.t = project_fast(lds_warp(wlds), (float)x, (float)y, D, H, W, live);
        TapOff f;
        const int base = (s.t.y0 * W + s.t.x0) * 4;
        f.o[0] = s.t.in00 ? (unsigned)base : WL_OOB;
        f.o[1] = s.t.in01 ? (unsigned)(base + 4) : WL_OOB;
        f.o[2] = s.t.in10 ? (unsigned)(base + W * 4) : WL_OOB;
        f.o[3] = s.t.in11 ? (unsigned)(base + W * 4 + 4) : WL_OOB;
        gather_taps(src, f, pb, s.q);
    };
    auto camera_point = [&](int x, int y, float D, float* X) {
        const F12 k = lds12(s_sf.Kinv);
        const float fx = (float)x, fy = (float)y;
        X[0] = fmaf(k.v[0], fx, fmaf(k.v[1], fy, k.v[2])) * D;
        X[1] = fmaf(k.v[3], fx, fmaf(k.v[4], fy, k.v[5])) * D;
        X[2] = fmaf(k.v[6], fx, fmaf(k.v[7], fy, k.v[8])) * D;
    };
    auto dump = [&](int w, unsigned off, const float* v) {
        if (off == WL_OOB) return;
#pragma unroll
        for (int k = 0; k < WL_DBG; ++k) a.dbg[(((size_t)b * 3 + w) * WL_DBG + k) * plane + (off >> 2)] = v[k];
    };
    float acc[RED_N0];
#pragma unroll
    for (int k = 0; k < RED_N0; ++k) acc[k] = 0.f;
    // d loss / d disparity of a pixel is STAGED in LDS and written out every WL_STAGE pixels.  On gfx9-family parts stores share vmcnt with
    // loads and complete out of order with them, so with a store pending every wait for an older load becomes vmcnt(0): one global store per
    // pixel drained the gather pipeline once per pixel (the next unit's gathers, just issued, had to land before the current unit's could be
    // used).  Staged, that full drain happens once per 16 pixels.  The stage borrows the block reduction's scratch (used after the loop).
    static_assert(sizeof(float) * WL_STAGE * 256 <= sizeof(float) * RED_N0 * RED_LD, "gradient stage fits the reduction scratch");
    float* const stage = &sred[0][0];
    auto flush = [&](int j_first, int count, float* dst) {
        for (int k = 0; k < count; ++k) {
            int fx, fy;
            unsigned foff;
            pixel(j_first + k, fx, fy, foff);
            if (foff != WL_OOB) dst[foff >> 2] = stage[k * 256 + tid];
        }
    };
    const int npix = 2 * nmine;
    float* const slab = a.slab + ((size_t)b * (a.G0 + a.G1) + blockIdx.x) * SLAB;

    if (!pass1) {
        // ---- pass 0: warps 0 (ref0 -> tgt) and 1 (ref1 -> tgt) with depth(tgt); the smoothness term; d loss / d disp(tgt)
        const float gw0 = g0 * a.tw[0] * invN, gw1 = g0 * a.tw[1] * invN, lw0 = a.tw[0] * invN, lw1 = a.tw[1] * invN;
        const float cxx = 1.0f / (float)((size_t)a.B * H * (W - 2));
        const float cyy = 1.0f / (float)((size_t)a.B * (H - 2) * W);
        const float cxy = 2.0f / (float)((size_t)a.B * (H - 1) * (W - 1));   // dxdy and dydx are the same field
        const bool smooth = !(a.flags & MCAV_WL_NO_SMOOTH);
        const WarpFast &w0 = s_sf.w[0], &w1 = s_sf.w[1];
        const __amdgpu_buffer_rsrc_t rs_r0 = image_rsrc(a.ref0 + (size_t)b * 3 * plane, plane), rs_r1 = image_rsrc(a.ref1 + (size_t)b * 3 * plane, plane);
        const float* const dtp = a.disp_t + (size_t)b * plane;
        const float* const tgp = a.tgt + (size_t)b * 3 * plane;
        float* const gtp = a.d_disp_t + (size_t)b * plane;
        auto fetch = [&](unsigned off, float (&v)[4]) {          // disparity and the target's three channels at a pixel (zeros past the image)
            v[0] = v[1] = v[2] = v[3] = 0.f;
            if (off != WL_OOB) {
                const unsigned i = off >> 2;
                v[0] = dtp[i];
#pragma unroll
                for (int c = 0; c < 3; ++c) v[1 + c] = tgp[c * plane + i];
            }
        };
        int x, y, xn, yn;
        unsigned off, offn;
        float cur[4], nxt[4];
        Set s0, s1;
        pixel(0, x, y, off);
        fetch(off, cur);
        float D = depth_of(cur[0]);
        issue(w0, rs_r0, x, y, D, off != WL_OOB, s0);
        for (int j = 0; j < npix; ++j) {
            pixel(j + 1, xn, yn, offn);
            fetch(offn, nxt);                                    // the next pixel's aligned values fly during this pixel's work
            issue(w1, rs_r1, x, y, D, off != WL_OOB, s1);        // warp 1's gathers fly while warp 0 is consumed
            float X[3], dDt = 0.f, labs = 0.f, dbg[DBG ? WL_DBG : 1];
            camera_point(x, y, D, X);
            warp_unit_fast(s0.q, cur + 1, s0.t, X, H, W, gw0, labs, dDt, acc + 2, DBG ? dbg : nullptr);
            acc[0] = fmaf(labs, lw0, acc[0]);
            if constexpr (DBG) dump(0, off, dbg);
            const float Dn = depth_of(nxt[0]);
            issue(w0, rs_r0, xn, yn, Dn, offn != WL_OOB, s0);    // the next pixel's warp 0 flies while warp 1 and the smoothness term are worked
            labs = 0.f;
            warp_unit_fast(s1.q, cur + 1, s1.t, X, H, W, gw1, labs, dDt, acc + 14, DBG ? dbg : nullptr);
            acc[0] = fmaf(labs, lw1, acc[0]);
            if constexpr (DBG) dump(1, off, dbg);
            if (smooth) {
                if (!(j & 1)) {                                  // first pixel of a tile: publish its depth tile (+ 2 halo)
                    const int by0 = y - ty0, bx0 = x - tx;
                    __syncthreads();                             // the previous tile's readers are done
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
                        const int i = tid + 256 * m, ly = i / LW, lx = i - ly * LW;
                        const int gy = by0 - HALO + ly, gx = bx0 - HALO + lx;
                        if (i < T2LH * LW) sD[ly][lx] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? depth_of(dtp[gy * W + gx]) : 0.f;
                    }
                    __syncthreads();
                }
                if (off != WL_OOB) {
                    const int cy = (j & 1) * TH + ty0 + HALO, cx = tx + HALO;
                    float gs = 0.f, ls = 0.f;
                    smooth_terms_sel([&](int dy, int dx) { return sD[cy + dy][cx + dx]; }, x, y, H, W, cxx, cyy, cxy, ls, gs);
                    acc[1] += ls;
                    dDt = fmaf(g1, gs, dDt);
                }
            }
            stage[(j & (WL_STAGE - 1)) * 256 + tid] = in_depth ? dDt : dDt * (-10.0f * D * D);
            if ((j & (WL_STAGE - 1)) == WL_STAGE - 1 || j + 1 == npix) flush(j & ~(WL_STAGE - 1), (j & (WL_STAGE - 1)) + 1, gtp);
            x = xn; y = yn; off = offn; D = Dn;
#pragma unroll
            for (int k = 0; k < 4; ++k) cur[k] = nxt[k];
        }
        __syncthreads();                                         // (sD and sred do not alias, but every wavefront must be out of the loop's barriers)
        block_sum_to_slab<RED_N0>(acc, sred, slab, [](int k) { return k; });
        if (tid >= RED_N0 && tid < SLAB) slab_store(slab + tid, 0.f);      // warp 2's slots
    } else {
        // ---- pass 1: warp 2 (tgt sampled with depth(ref0) and the inverted pose[0], compared with ref1: losses.py:203-207); d loss / d disp(ref0)
        const float gw2 = g0 * a.tw[2] * invN, lw2 = a.tw[2] * invN;
        const WarpFast& w2 = s_sf.w[2];
        const __amdgpu_buffer_rsrc_t rs_t = image_rsrc(a.tgt + (size_t)b * 3 * plane, plane);
        const float* const drp = a.disp_r0 + (size_t)b * plane;
        const float* const r1p = a.ref1 + (size_t)b * 3 * plane;
        float* const grp = a.d_disp_r0 + (size_t)b * plane;
        auto fetch = [&](unsigned off, float (&v)[4]) {          // disparity of ref0 and ref1's three channels at a pixel (zeros past the image)
            v[0] = v[1] = v[2] = v[3] = 0.f;
            if (off != WL_OOB) {
                const unsigned i = off >> 2;
                v[0] = drp[i];
#pragma unroll
                for (int c = 0; c < 3; ++c) v[1 + c] = r1p[c * plane + i];
            }
        };
        int xa, ya, xb, yb, xna, yna, xnb, ynb;
        unsigned offa, offb, offna, offnb;
        float ca[4], cb[4], na[4], nb[4];
        Set s0, s1;
        pixel(0, xa, ya, offa);
        pixel(1, xb, yb, offb);
        fetch(offa, ca);
        fetch(offb, cb);
        float Da = depth_of(ca[0]), Db = depth_of(cb[0]);
        issue(w2, rs_t, xa, ya, Da, offa != WL_OOB, s0);
        for (int j = 0; j < npix; j += 2) {
            pixel(j + 2, xna, yna, offna);
            pixel(j + 3, xnb, ynb, offnb);
            fetch(offna, na);
            fetch(offnb, nb);
            issue(w2, rs_t, xb, yb, Db, offb != WL_OOB, s1);
            float X[3], dDr = 0.f, labs = 0.f, dbg[DBG ? WL_DBG : 1];
            camera_point(xa, ya, Da, X);
            warp_unit_fast(s0.q, ca + 1, s0.t, X, H, W, gw2, labs, dDr, acc + 1, DBG ? dbg : nullptr);
            if constexpr (DBG) dump(2, offa, dbg);
            stage[(j & (WL_STAGE - 1)) * 256 + tid] = in_depth ? dDr : dDr * (-10.0f * Da * Da);
            const float Dna = depth_of(na[0]), Dnb = depth_of(nb[0]);
            issue(w2, rs_t, xna, yna, Dna, offna != WL_OOB, s0);
            dDr = 0.f;
            camera_point(xb, yb, Db, X);
            warp_unit_fast(s1.q, cb + 1, s1.t, X, H, W, gw2, labs, dDr, acc + 1, DBG ? dbg : nullptr);
            if constexpr (DBG) dump(2, offb, dbg);
            stage[((j + 1) & (WL_STAGE - 1)) * 256 + tid] = in_depth ? dDr : dDr * (-10.0f * Db * Db);
            if (((j + 1) & (WL_STAGE - 1)) == WL_STAGE - 1 || j + 2 >= npix) flush(j & ~(WL_STAGE - 1), ((j + 1) & (WL_STAGE - 1)) + 1, grp);
            acc[0] = fmaf(labs, lw2, acc[0]);
            xa = xna; ya = yna; offa = offna; Da = Dna;
            xb = xnb; yb = ynb; offb = offnb; Db = Dnb;
#pragma unroll
            for (int k = 0; k < 4; ++k) { ca[k] = na[k]; cb[k] = nb[k]; }
        }
        __syncthreads();                                         // (the stage and sred alias)
        block_sum_to_slab<RED_N1>(acc, sred, slab, [](int k) { return k == 0 ? 0 : 25 + k; });      // loss share; dP of warp 2 -> slots 26..37
        if (tid >= 1 && tid < 26) slab_store(slab + tid, 0.f);
        if (tid >= 38 && tid < SLAB) slab_store(slab + tid, 0.f);
    }
    block_finish(a, b, a.G0 + a.G1, reinterpret_cast<double (*)[SLAB]>(&sred[0][0]), &s_flag);
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
    float g0 = 1.0f, g1 = 1.0f;
    if (a.upstream) {
        g0 = a.upstream[0];
        g1 = a.upstream[1];
        if ((a.flags & MCAV_WL_SKIP_IF_UNIT) && g0 == 1.0f && g1 == 1.0f) return;
    }
    __shared__ float sD[WL_LH][LW + 1];
    __shared__ float sX[3][WL_LH][LW + 1];
    __shared__ float sT[3][WL_LH][LW + 1];
    __shared__ __attribute__((aligned(16))) float sC[3][SS_P][SS_P + 1];
    // LDS budget: 52.2 KB = THREE workgroups per CU.  The per-sample constants live where the block reduction's scratch will be (the
    // reduction runs after the last use of the constants), the float64 finalize scratch on top of the coefficient fields (free by then):
    // as separate arrays they added 2.1 KB, 163 KB for three workgroups, and the kernel ran at two per CU (0.62 -> 1.0 ms at 320x1024).
    __shared__ __attribute__((aligned(16))) float s_red_sf[4 * SLAB];
    float (*const sred)[SLAB] = reinterpret_cast<float (*)[SLAB]>(s_red_sf);
    SampleFast& s_sf = *reinterpret_cast<SampleFast*>(s_red_sf);
    static_assert(sizeof(SampleFast) <= sizeof(float) * 4 * SLAB, "constants fit the reduction scratch");
    double (*const s64)[SLAB] = reinterpret_cast<double (*)[SLAB]>(&sC[0][0][0]);
    static_assert(sizeof(double) * (256 / SLAB) * SLAB <= sizeof(float) * 3 * SS_P * (SS_P + 1), "finalize scratch fits the coefficient fields");
    __shared__ int s_flag;
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
            D = in_depth ? v : rcp_nr(fmaf(10.0f, v, 0.01f));
        }
        sD[ly][lx] = D;
    }
    block_prepare(a, b, &s_sf);
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
            Dr[sub] = in_depth ? vr : rcp_nr(fmaf(10.0f, vr, 0.01f));
        }
    }

#pragma unroll
    for (int w = 0; w < 3; ++w) {
        const float* src = w == 0 ? img_r0 : (w == 1 ? img_r1 : img_t);
        const float* tar = w == 2 ? img_r1 : img_t;
        const WarpFast wf = lds_warp(s_sf.w[w]);
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
                    D = in_depth ? v : rcp_nr(fmaf(10.0f, v, 0.01f));
                }
                const FTap t = project_fast(wf, (float)rx, (float)ry, D, H, W);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float q4[4];
                    texels_of(src + c * plane, W, t, q4);
                    xv[c] = bilinear_lerp(q4[0], q4[1], q4[2], q4[3], t.wx1, t.wy1).v;
                }
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
            const float Dp = w < 2 ? sD[ty + HALO][tx + HALO] : Dr[sub];
            const FTap t = project_fast(wf, (float)x, (float)y, Dp, H, W);
            float gix = 0.f, giy = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float q4[4];
                texels_of(src + c * plane, W, t, q4);
                const Sample sm = bilinear_lerp(q4[0], q4[1], q4[2], q4[3], t.wx1, t.wy1);
                const float gv = c == 0 ? gp0[sub] : (c == 1 ? gp1[sub] : gp2[sub]);
                gix += gv * sm.dvdx;
                giy += gv * sm.dvdy;
            }
            const F12 k = lds12(s_sf.Kinv);
            const float fx = (float)x, fy = (float)y;
            const float X[3] = {fmaf(k.v[0], fx, fmaf(k.v[1], fy, k.v[2])) * Dp, fmaf(k.v[3], fx, fmaf(k.v[4], fy, k.v[5])) * Dp,
                                fmaf(k.v[6], fx, fmaf(k.v[7], fy, k.v[8])) * Dp};
            const float d = backproject_fast(t, X, gix, giy, H, W, acc + 2 + 12 * w);
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
    float* const slab = a.slab + ((size_t)b * nblk + blk) * SLAB;
    __syncthreads();                                   // every reader of the constants (and of sC) is done: their LDS is re-used below
    block_reduce_store<NACC, true>(acc, slab, sred);
    if (threadIdx.x >= NACC && threadIdx.x < SLAB) slab_store(slab + threadIdx.x, 0.f);
    block_finish(a, b, nblk, s64, &s_flag);
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
    size_t pc_off, tick_off, slab_off, sl_off, total;
    int nblk;
};

constexpr int WL_MAX_B = 4095;      // samples per launch of the fused kernels (ticket capacity)

inline WsLayout ws_layout(int B, int H, int W) {
    WsLayout l;
    l.nblk = ((W + TW - 1) / TW) * ((H + TH - 1) / TH);            // workgroups per sample of the standalone kernels (32 x 8 pixel tiles)
    const int cap = ((W + TW - 1) / TW) * 2 * ((H + T2H - 1) / T2H);      // >= the workgroups per sample of the fused kernels
    const int rows = l.nblk > cap ? l.nblk : cap;
    // The tickets sit at a FIXED place and size, whatever (B, H, W): a caller's cached workspace serves launches of different shapes, and a
    // ticket word that another shape's slab had used would not be zero.  Nothing else is ever written to this region.
    size_t o = 0;
    l.tick_off = o; o = align_up(o + sizeof(unsigned) * ((size_t)WL_MAX_B + 1), 256);
    l.pc_off = o;   o = align_up(o + sizeof(PrepConst) * (size_t)B, 256);
    l.slab_off = o; o = align_up(o + sizeof(float) * SLAB * (size_t)B * rows, 256);
    l.sl_off = o;   o = align_up(o + sizeof(double) * 2 * (size_t)B, 256);
    l.total = o;
    return l;
}

// workgroups per sample of the fused L1 kernel: pass 0 (warps 0, 1 + smoothness: ~2.7 units of work per tile) and pass 1 (warp 2: 1 unit), sized
// so that every workgroup of the launch is resident at once (CUs x workgroups per CU, asked of the runtime) and the two kinds take about the
// same time -- a second, partly filled generation would double the launch's duration
inline int l1_slots() {
    static int slots = 0;
    if (slots == 0) {
        int dev = 0, cus = 256, per_cu = 3;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, warp_loss_l1_kernel<false>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 3;
        (void)hipGetLastError();
        slots = cus * per_cu;
    }
    return slots;
}

inline void l1_grid(int B, int H, int W, int& G0, int& G1) {
    const int ntiles = ((W + TW - 1) / TW) * ((H + T2H - 1) / T2H);
    const int slots = l1_slots();
    // tiles of a pass-1 workgroup per tile of a pass-0 workgroup.  Counted in instructions a pass-0 tile is 2.7 pass-1 tiles; measured on
    // MI355X (tools/loss_bench.py, network-like disparities, 12 x 192x640 / 12 x 320x1024): 1.0 -> 73.8 / 201 us, 2.0 -> 66.2 / 171, 2.7 -> 69.2 /
    // 172.  MCAV_WL_RATIO overrides (experiments).  Also measured and not kept: the smoothness tile's depths fetched a pixel ahead (68.4 us),
    // a three-stage pipeline with 24 .. 48 gathers in flight at two wavefronts per SIMD (68.6 .. 73 us).
    static const float ratio = [] { const float r = MCAV_KNOB_FLOAT("MCAV_WL_RATIO", 2.0f); return r > 0.05f ? r : 2.0f; }();
    const int per_sample = slots / B > 0 ? slots / B : 1;
    int n0 = (int)((ntiles * (1.0f + 1.0f / ratio) + per_sample - 1) / per_sample);
    if (n0 < 1) n0 = 1;
    for (;; ++n0) {
        int n1 = (int)(ratio * n0 + 0.5f);
        if (n1 < 1) n1 = 1;
        G0 = (ntiles + n0 - 1) / n0;
        G1 = (ntiles + n1 - 1) / n1;
        if ((G0 + G1) * B <= slots || G0 + G1 <= 2) break;       // (rounding up twice can overshoot by a few workgroups)
    }
    // XCD-local sharing between the passes (round 4).  Tile t of a sample is walked by pass-0 workgroup t mod G0 and by pass-1 workgroup
    // t mod G1 at about the same time (the ratio above), and both read the target's and ref1's planes there.  Workgroups are dealt to the 8
    // XCDs round-robin by their linear index b (G0 + G1) + x: with G0 and G1 multiples of 8 both workgroups of a tile sit on XCD t mod 8 and
    // the second reader hits the first one's lines in that XCD's L2 -- otherwise the planes the passes share come from the fabric twice
    // (round 3 measured 93.5 MB per launch against 76.7 MB algorithmic).  Rounded up where the launch still fits one generation, else down.
    static const int xcd_align = MCAV_KNOB_INT("MCAV_WL_XCD", 1);
    if (xcd_align && G0 >= 8 && G1 >= 8) {
        int g0 = (G0 + 7) / 8 * 8, g1 = (G1 + 7) / 8 * 8;
        if ((g0 + g1) * B > slots) g1 = G1 / 8 * 8;
        if ((g0 + g1) * B > slots) g0 = G0 / 8 * 8;
        if (g0 >= 8 && g1 >= 8 && g0 <= ntiles && g1 <= ntiles && (g0 + g1) * B <= slots) { G0 = g0; G1 = g1; }
    }
}

inline dim3 pix_grid(int B, int H, int W) { return dim3((W + TW - 1) / TW, (H + TH - 1) / TH, B); }

}  // namespace mcav

using namespace mcav;

MCAV_EXPORT int mcav_abi_version(void) { return 1; }

MCAV_EXPORT size_t mcav_warp_loss_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return ws_layout(B, H, W).total;
}

// The fused kernels keep their completion tickets in the caller's workspace and leave them at zero (block_finish).  A launch that returned an
// error may not have run to that point: its workspace is remembered here and the NEXT launch on it zeroes the ticket region first (one
// 16 KB memset on the stream, only ever after a failure) -- otherwise no later launch on that cached workspace would elect a finisher and
// `losses` / `d_poses` would silently keep stale values.
static std::mutex g_suspect_mu;
static std::vector<void*> g_suspect_ws;
static bool take_suspect(void* ws) {
    std::lock_guard<std::mutex> lk(g_suspect_mu);
    for (size_t i = 0; i < g_suspect_ws.size(); ++i)
        if (g_suspect_ws[i] == ws) { g_suspect_ws.erase(g_suspect_ws.begin() + i); return true; }
    return false;
}
static void mark_suspect(void* ws) {
    std::lock_guard<std::mutex> lk(g_suspect_mu);
    for (void* p : g_suspect_ws) if (p == ws) return;
    g_suspect_ws.push_back(ws);
}

static int warp_loss_launch(const float* tgt, const float* ref0, const float* ref1, const float* disp_t, const float* disp_r0,
                            const float* poses, const void* K, int B, int H, int W, unsigned flags, const float* upstream,
                            const float* term_weights, float* losses, float* d_disp_t, float* d_disp_r0, float* d_poses,
                            void* workspace, size_t workspace_bytes, void* stream, float* dbg) {
    if (!tgt || !ref0 || !ref1 || !disp_t || !disp_r0 || !poses || !K || !losses || !d_disp_t || !d_disp_r0 || !d_poses || !workspace)
        return MCAV_E_INVALID;
    if (B <= 0 || H < 3 || W < 3 || B > WL_MAX_B) return MCAV_E_INVALID;
    const WsLayout l = ws_layout(B, H, W);
    if (workspace_bytes < l.total) return MCAV_E_WORKSPACE;
    char* ws = reinterpret_cast<char*>(workspace);
    hipStream_t s = as_stream(stream);
    WLArgs a;
    a.tgt = tgt; a.ref0 = ref0; a.ref1 = ref1; a.disp_t = disp_t; a.disp_r0 = disp_r0; a.poses = poses;
    a.K = K; a.upstream = upstream; a.d_disp_t = d_disp_t; a.d_disp_r0 = d_disp_r0; a.d_poses = d_poses; a.losses = losses;
    a.slab = reinterpret_cast<float*>(ws + l.slab_off);
    a.sample_loss = reinterpret_cast<double*>(ws + l.sl_off);
    a.tickets = reinterpret_cast<unsigned*>(ws + l.tick_off);
    a.B = B; a.H = H; a.W = W; a.flags = flags; a.G0 = 0; a.G1 = 0;
    a.tw[0] = term_weights ? term_weights[0] : 0.25f;
    a.tw[1] = term_weights ? term_weights[1] : 0.25f;
    a.tw[2] = term_weights ? term_weights[2] : 0.5f;
    a.dbg = dbg;
    if (take_suspect(workspace) && hipMemsetAsync(a.tickets, 0, sizeof(unsigned) * ((size_t)WL_MAX_B + 1), s) != hipSuccess) {
        (void)hipGetLastError();
        mark_suspect(workspace);
        return MCAV_E_LAUNCH;
    }
    if (flags & MCAV_WL_SSIM) {
        const dim3 wl_grid((W + TW - 1) / TW, (H + WLH - 1) / WLH, B);
        if (dbg) timed_launch(warp_loss_ssim_kernel<true>, wl_grid, dim3(256), 0, s, a);
        else timed_launch(warp_loss_ssim_kernel<false>, wl_grid, dim3(256), 0, s, a);
    } else {
        l1_grid(B, H, W, a.G0, a.G1);
        const dim3 grid(a.G0 + a.G1, B);
        if (dbg) timed_launch(warp_loss_l1_kernel<true>, grid, dim3(256), 0, s, a);
        else timed_launch(warp_loss_l1_kernel<false>, grid, dim3(256), 0, s, a);
    }
    const int rc = launch_status();
    if (rc != MCAV_OK) mark_suspect(workspace);
    return rc;
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
