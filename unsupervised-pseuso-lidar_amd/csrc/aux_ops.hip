// Small HBM-bound helpers used by the secondary networks (DispNetS, PoseFc) and the multi-scale loss:
// channel copy (concat / slice), bilinear resize of 1-channel maps (align_corners = False) and its adjoint,
// affine map, elementwise product, per-channel column sums.
#include "conv_gather.h"

namespace mcav {

inline int aux_grid(size_t n, int cap = 4096) {
    size_t b = (n + 255) / 256;
    if (b < 1) b = 1;
    return (int)(b < (size_t)cap ? b : (size_t)cap);
}

__global__ void copy_channels_kernel(const float* src, size_t n_pix, int Cs, int soff, float* dst, int Cd, int doff, int C, int accumulate) {
    const size_t total = n_pix * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / C;
        const int c = (int)(i - p * C);
        const float v = src[p * Cs + soff + c];
        float* d = dst + p * Cd + doff + c;
        *d = accumulate ? *d + v : v;
    }
}

// PyTorch's bilinear source index with align_corners = False: max(0, scale * (dst + 0.5) - 0.5)
__device__ __forceinline__ void bil_src(int o, float scale, int n_in, int& i0, int& i1, float& lam) {
    float s = scale * ((float)o + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    if (i0 > n_in - 1) i0 = n_in - 1;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    lam = s - (float)i0;
}

__global__ void resize_bilinear_fwd_kernel(const float* src, int B, int h, int w, float* dst, int H, int W, float sy, float sx) {
    const size_t total = (size_t)B * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % W);
        const int oy = (int)((i / W) % H);
        const int b = (int)(i / ((size_t)W * H));
        int y0, y1, x0, x1;
        float ly, lx;
        bil_src(oy, sy, h, y0, y1, ly);
        bil_src(ox, sx, w, x0, x1, lx);
        const float* p = src + (size_t)b * h * w;
        const float top = p[y0 * w + x0] * (1.f - lx) + p[y0 * w + x1] * lx;
        const float bot = p[y1 * w + x0] * (1.f - lx) + p[y1 * w + x1] * lx;
        dst[i] = top * (1.f - ly) + bot * ly;
    }
}

// adjoint as a gather: every source pixel collects from the destination pixels whose taps touch it (fixed order)
__global__ void resize_bilinear_bwd_kernel(const float* ddst, int B, int h, int w, float* dsrc, int H, int W, float sy, float sx, int accumulate) {
    const size_t total = (size_t)B * h * w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ix = (int)(i % w);
        const int iy = (int)((i / w) % h);
        const int b = (int)(i / ((size_t)w * h));
        int oy_lo = (int)floorf(((float)iy - 0.5f) / sy - 0.5f) - 1, oy_hi = (int)ceilf(((float)iy + 1.5f) / sy - 0.5f) + 1;
        int ox_lo = (int)floorf(((float)ix - 0.5f) / sx - 0.5f) - 1, ox_hi = (int)ceilf(((float)ix + 1.5f) / sx - 0.5f) + 1;
        if (iy == 0) oy_lo = 0;          // the clamp max(0, .) maps a run of leading outputs onto source 0
        if (ix == 0) ox_lo = 0;
        oy_lo = oy_lo < 0 ? 0 : oy_lo; ox_lo = ox_lo < 0 ? 0 : ox_lo;
        oy_hi = oy_hi > H - 1 ? H - 1 : oy_hi; ox_hi = ox_hi > W - 1 ? W - 1 : ox_hi;
        if (iy == h - 1) oy_hi = H - 1;
        if (ix == w - 1) ox_hi = W - 1;
        const float* g = ddst + (size_t)b * H * W;
        float acc = 0.f;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            int y0, y1;
            float ly;
            bil_src(oy, sy, h, y0, y1, ly);
            float wy = 0.f;
            if (y0 == iy) wy += 1.f - ly;
            if (y1 == iy) wy += ly;
            if (wy == 0.f) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                int x0, x1;
                float lx;
                bil_src(ox, sx, w, x0, x1, lx);
                float wx = 0.f;
                if (x0 == ix) wx += 1.f - lx;
                if (x1 == ix) wx += lx;
                if (wx != 0.f) acc += wy * wx * g[(size_t)oy * W + ox];
            }
        }
        dsrc[i] = accumulate ? dsrc[i] + acc : acc;
    }
}

__global__ void affine_kernel(const float* x, float a, float b, size_t n, float* y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = a * x[i] + b;
}

__global__ void mul_kernel(const float* a, const float* b, size_t n, float* y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = a[i] * b[i];
}

constexpr int CS_BLOCKS = 128;
__global__ __launch_bounds__(256) void colsum_part_kernel(const float* x, size_t n_pix, int C, float* part) {
    const size_t per = (n_pix + gridDim.x - 1) / gridDim.x;
    const size_t pb = (size_t)blockIdx.x * per, pe = pb + per < n_pix ? pb + per : n_pix;
    // thread (c, lane): lanes stride over pixels; consecutive threads read consecutive channels (coalesced)
    const int Cc = C < 256 ? C : 256, PL = 256 / Cc;
    __shared__ float sh[256];
    for (int c0 = 0; c0 < C; c0 += Cc) {
        const int c = c0 + (int)(threadIdx.x % Cc), pl = threadIdx.x / Cc;
        float s = 0.f;
        if (pl < PL && c < C)
            for (size_t m = pb + pl; m < pe; m += PL) s += x[m * C + c];
        sh[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x < Cc && c0 + (int)threadIdx.x < C) {
            float t = 0.f;
            for (int k = 0; k < PL; ++k) t += sh[threadIdx.x + k * Cc];
            part[(size_t)blockIdx.x * C + c0 + threadIdx.x] = t;
        }
        __syncthreads();
    }
}

__global__ void colsum_final_kernel(const float* part, int nblk, int C, float* out, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += (double)part[(size_t)b * C + c];
    out[c] = accumulate ? out[c] + (float)s : (float)s;
}

// Fold of the pooled upsample adjoint (see mcav_upsample_adj_fold): tmp is the gradient on the edge-replicated low-resolution domain
// [B][Hl+2][Wl+2][C]; the ring is added back onto the border pixels it was replicated from, then * act'(aux) + addend.
__global__ __launch_bounds__(256) void upsample_adj_fold_kernel(const float* __restrict__ tmp, int B, int Hl, int Wl, int C4, const float* __restrict__ aux,
                                                                int dact, const float* __restrict__ addend, float* __restrict__ out) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    const size_t total = (size_t)B * Hl * Wl * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        size_t r = i / C4;
        const int t = (int)(r % Wl); r /= Wl;
        const int s = (int)(r % Hl);
        const int b = (int)(r / Hl);
        const int ys[3] = {s + 1, s == 0 ? 0 : -1, s == Hl - 1 ? Hl + 1 : -1};
        const int xs[3] = {t + 1, t == 0 ? 0 : -1, t == Wl - 1 ? Wl + 1 : -1};
        v4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int e = 0; e < 3; ++e)
                if (ys[a] >= 0 && xs[e] >= 0)
                    g += reinterpret_cast<const v4*>(tmp)[(((size_t)b * (Hl + 2) + ys[a]) * (Wl + 2) + xs[e]) * C4 + c];
        if (aux) {
            const v4 y = reinterpret_cast<const v4*>(aux)[i];
            if (dact == MCAV_ACT_RELU) { g.x = y.x > 0.f ? g.x : 0.f; g.y = y.y > 0.f ? g.y : 0.f; g.z = y.z > 0.f ? g.z : 0.f; g.w = y.w > 0.f ? g.w : 0.f; }
            else if (dact == MCAV_ACT_ELU) { g.x *= y.x > 0.f ? 1.f : y.x + 1.f; g.y *= y.y > 0.f ? 1.f : y.y + 1.f; g.z *= y.z > 0.f ? 1.f : y.z + 1.f; g.w *= y.w > 0.f ? 1.f : y.w + 1.f; }
            else if (dact == MCAV_ACT_SIGMOID) { g.x *= y.x * (1.f - y.x); g.y *= y.y * (1.f - y.y); g.z *= y.z * (1.f - y.z); g.w *= y.w * (1.f - y.w); }
        }
        if (addend) g += reinterpret_cast<const v4*>(addend)[i];
        reinterpret_cast<v4*>(out)[i] = g;
    }
}


// nearest-neighbour x2 upsampling of NCHW planes (reference models/depth/layers.py:55-58 `upsample`) and its adjoint (2x2 sum)
__global__ __launch_bounds__(256) void upsample_nearest2x_kernel(const float* __restrict__ src, size_t planes, int h, int w, float* __restrict__ dst) {
    const int W = 2 * w, H = 2 * h;
    const size_t total = planes * (size_t)H * (W / 2);          // one thread writes the two output pixels of a source pixel's row
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % w);
        const size_t r = i / w;
        const int oy = (int)(r % H);
        const size_t pl = r / H;
        const float v = src[(pl * h + (oy >> 1)) * w + x];
        typedef float v2 __attribute__((ext_vector_type(2)));
        const v2 o = {v, v};
        reinterpret_cast<v2*>(dst)[(pl * H + oy) * w + x] = o;
    }
}

__global__ __launch_bounds__(256) void upsample_nearest2x_bwd_kernel(const float* __restrict__ g, size_t planes, int h, int w, float* __restrict__ out) {
    const int W = 2 * w;
    const size_t total = planes * (size_t)h * w;
    typedef float v2 __attribute__((ext_vector_type(2)));
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % w);
        const size_t r = i / w;
        const int y = (int)(r % h);
        const size_t pl = r / h;
        const v2 a = reinterpret_cast<const v2*>(g)[((pl * 2 * h + 2 * y) * W + 2 * x) / 2];
        const v2 b = reinterpret_cast<const v2*>(g)[((pl * 2 * h + 2 * y + 1) * W + 2 * x) / 2];
        out[i] = (a.x + a.y) + (b.x + b.y);
    }
}

}  // namespace mcav

using namespace mcav;

MCAV_EXPORT int mcav_copy_channels(const float* src, size_t n_pix, int Cs, int soff, float* dst, int Cd, int doff, int C, int accumulate, void* stream) {
    if (!src || !dst || C <= 0 || soff < 0 || doff < 0 || soff + C > Cs || doff + C > Cd) return MCAV_E_INVALID;
    if (n_pix == 0) return MCAV_OK;
    copy_channels_kernel<<<aux_grid(n_pix * C), 256, 0, as_stream(stream)>>>(src, n_pix, Cs, soff, dst, Cd, doff, C, accumulate);
    return launch_status();
}

MCAV_EXPORT int mcav_resize_bilinear_fwd(const float* src, int B, int h, int w, float* dst, int H, int W, float scale_y, float scale_x, void* stream) {
    if (!src || !dst || B <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return MCAV_E_INVALID;
    const float sy = scale_y > 0.f ? scale_y : (float)h / (float)H, sx = scale_x > 0.f ? scale_x : (float)w / (float)W;
    resize_bilinear_fwd_kernel<<<aux_grid((size_t)B * H * W), 256, 0, as_stream(stream)>>>(src, B, h, w, dst, H, W, sy, sx);
    return launch_status();
}

MCAV_EXPORT int mcav_resize_bilinear_bwd(const float* ddst, int B, int h, int w, float* dsrc, int H, int W, float scale_y, float scale_x, int accumulate,
                                         void* stream) {
    if (!ddst || !dsrc || B <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return MCAV_E_INVALID;
    const float sy = scale_y > 0.f ? scale_y : (float)h / (float)H, sx = scale_x > 0.f ? scale_x : (float)w / (float)W;
    if (sy > 1.f || sx > 1.f) return MCAV_E_INVALID;       // the gather-form adjoint assumes upsampling
    resize_bilinear_bwd_kernel<<<aux_grid((size_t)B * h * w), 256, 0, as_stream(stream)>>>(ddst, B, h, w, dsrc, H, W, sy, sx, accumulate);
    return launch_status();
}

MCAV_EXPORT int mcav_affine(const float* x, float a, float b, size_t n, float* y, void* stream) {
    if (!x || !y) return MCAV_E_INVALID;
    if (n == 0) return MCAV_OK;
    affine_kernel<<<aux_grid(n), 256, 0, as_stream(stream)>>>(x, a, b, n, y);
    return launch_status();
}

MCAV_EXPORT int mcav_mul(const float* a, const float* b, size_t n, float* y, void* stream) {
    if (!a || !b || !y) return MCAV_E_INVALID;
    if (n == 0) return MCAV_OK;
    mul_kernel<<<aux_grid(n), 256, 0, as_stream(stream)>>>(a, b, n, y);
    return launch_status();
}

MCAV_EXPORT size_t mcav_colsum_workspace_bytes(int C) { return C > 0 ? align_up(sizeof(float) * (size_t)CS_BLOCKS * C, 256) : 0; }

MCAV_EXPORT int mcav_colsum(const float* x, size_t n_pix, int C, float* out, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !out || !workspace || C <= 0 || n_pix == 0) return MCAV_E_INVALID;
    if (C > 256 && C % 256 != 0) return MCAV_E_INVALID;
    if (workspace_bytes < mcav_colsum_workspace_bytes(C)) return MCAV_E_WORKSPACE;
    const int blocks = (int)(n_pix < (size_t)CS_BLOCKS ? n_pix : (size_t)CS_BLOCKS);
    float* part = reinterpret_cast<float*>(workspace);
    hipStream_t s = as_stream(stream);
    colsum_part_kernel<<<blocks, 256, 0, s>>>(x, n_pix, C, part);
    colsum_final_kernel<<<(C + 63) / 64, 64, 0, s>>>(part, blocks, C, out, accumulate);
    return launch_status();
}

MCAV_EXPORT int mcav_upsample_adj_fold(const float* tmp, int B, int Hl, int Wl, int C, const float* dact_aux, int dact, const float* addend, float* out,
                                       void* stream) {
    if (!tmp || !out || B <= 0 || Hl <= 0 || Wl <= 0 || C <= 0 || (C & 3)) return MCAV_E_INVALID;
    const size_t n4 = (size_t)B * Hl * Wl * (C / 4);
    const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    upsample_adj_fold_kernel<<<blocks, 256, 0, as_stream(stream)>>>(tmp, B, Hl, Wl, C / 4, dact_aux, dact, addend, out);
    return launch_status();
}

MCAV_EXPORT int mcav_upsample_nearest2x(const float* src, size_t planes, int h, int w, float* dst, void* stream) {
    if (!src || !dst || planes == 0 || h <= 0 || w <= 0) return MCAV_E_INVALID;
    upsample_nearest2x_kernel<<<aux_grid(planes * 2 * h * w, 8192), 256, 0, as_stream(stream)>>>(src, planes, h, w, dst);
    return launch_status();
}

MCAV_EXPORT int mcav_upsample_nearest2x_bwd(const float* grad_out, size_t planes, int h, int w, float* grad_in, void* stream) {
    if (!grad_out || !grad_in || planes == 0 || h <= 0 || w <= 0) return MCAV_E_INVALID;
    upsample_nearest2x_bwd_kernel<<<aux_grid(planes * h * w, 8192), 256, 0, as_stream(stream)>>>(grad_out, planes, h, w, grad_in);
    return launch_status();
}

// ---------------------------------------------------------------------------------------------- external events of a captured step
// A hipGraph-captured training step that must hand data to work OUTSIDE the graph while it is still running (the gradient buckets of
// mcav/dist.py: all-reduced on a communication stream while the rest of the backward pass replays) marks the hand-off points with external
// event-record nodes: hipEventRecordWithFlags(..., hipEventRecordExternal) on the capturing stream.  On every replay the node records the
// event when the graph gets there; a stream outside the graph waits for it with mcav_stream_wait_event before its collective is issued.
MCAV_EXPORT int mcav_event_create(void** event) {
    if (!event) return MCAV_E_INVALID;
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return MCAV_E_LAUNCH; }
    *event = e;
    return MCAV_OK;
}

MCAV_EXPORT int mcav_event_destroy(void* event) {
    if (!event) return MCAV_E_INVALID;
    return hipEventDestroy(reinterpret_cast<hipEvent_t>(event)) == hipSuccess ? MCAV_OK : MCAV_E_LAUNCH;
}

// Under stream capture: an external event-record node (the event is signalled by each replay, for waiters outside the graph).  On a stream
// that is not capturing: a plain hipEventRecord.
MCAV_EXPORT int mcav_event_record_external(void* event, void* stream) {
    if (!event) return MCAV_E_INVALID;
    hipStream_t s = as_stream(stream);
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    hipError_t e = hipStreamIsCapturing(s, &st);
    if (e != hipSuccess) {
        fprintf(stderr, "mcav_event_record_external: hipStreamIsCapturing: %s\n", hipGetErrorName(e));
        (void)hipGetLastError();
        return -(2000 + (int)e);
    }
    if (st != hipStreamCaptureStatusActive) {
        e = hipEventRecord(reinterpret_cast<hipEvent_t>(event), s);
        if (e != hipSuccess) { (void)hipGetLastError(); return -(1000 + (int)e); }
        return MCAV_OK;
    }
    // Under capture: the node is added to the graph being captured explicitly -- an event-record node behind the stream's current dependency
    // set, which then becomes the set.  (hipEventRecordWithFlags(..., hipEventRecordExternal) on the capturing stream is the short spelling;
    // this ROCm's runtime answers it with hipErrorInvalidValue under torch's capture, measured round 4, so it is only the fallback.)
    unsigned long long id = 0;
    hipGraph_t graph = nullptr;
    const hipGraphNode_t* deps = nullptr;
    size_t ndeps = 0;
    e = hipStreamGetCaptureInfo_v2(s, &st, &id, &graph, &deps, &ndeps);
    if (e == hipSuccess && graph) {
        hipGraphNode_t node = nullptr;
        e = hipGraphAddEventRecordNode(&node, graph, deps, ndeps, reinterpret_cast<hipEvent_t>(event));
        if (e == hipSuccess) e = hipStreamUpdateCaptureDependencies(s, &node, 1, hipStreamSetCaptureDependencies);
        if (e == hipSuccess) return MCAV_OK;
        fprintf(stderr, "mcav_event_record_external: explicit event-record node: %s\n", hipGetErrorName(e));
    }
    (void)hipGetLastError();
    e = hipEventRecordWithFlags(reinterpret_cast<hipEvent_t>(event), s, hipEventRecordExternal);
    if (e != hipSuccess) {
        fprintf(stderr, "mcav_event_record_external: hipEventRecordWithFlags(external): %s\n", hipGetErrorName(e));
        (void)hipGetLastError();
        return -(1000 + (int)e);      // (-1000 - hipError_t: the binding prints the number)
    }
    return MCAV_OK;
}

// The wait side of the same mechanism.  On a CAPTURING stream: an external event-WAIT node (the graph's later nodes wait, on every replay, for
// the event as another graph or stream last recorded it: two captured graphs replayed on two streams order their work through such pairs,
// mcav/graph.py's weight-gradient graph).  On a stream that is not capturing: hipStreamWaitEvent.
MCAV_EXPORT int mcav_event_wait_external(void* event, void* stream) {
    if (!event) return MCAV_E_INVALID;
    hipStream_t s = as_stream(stream);
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    hipError_t e = hipStreamIsCapturing(s, &st);
    if (e != hipSuccess) { (void)hipGetLastError(); return -(2000 + (int)e); }
    if (st != hipStreamCaptureStatusActive) {
        e = hipStreamWaitEvent(s, reinterpret_cast<hipEvent_t>(event), 0);
        if (e != hipSuccess) { (void)hipGetLastError(); return -(1000 + (int)e); }
        return MCAV_OK;
    }
    unsigned long long id = 0;
    hipGraph_t graph = nullptr;
    const hipGraphNode_t* deps = nullptr;
    size_t ndeps = 0;
    e = hipStreamGetCaptureInfo_v2(s, &st, &id, &graph, &deps, &ndeps);
    if (e == hipSuccess && graph) {
        hipGraphNode_t node = nullptr;
        e = hipGraphAddEventWaitNode(&node, graph, deps, ndeps, reinterpret_cast<hipEvent_t>(event));
        if (e == hipSuccess) e = hipStreamUpdateCaptureDependencies(s, &node, 1, hipStreamSetCaptureDependencies);
        if (e == hipSuccess) return MCAV_OK;
    }
    fprintf(stderr, "mcav_event_wait_external: explicit event-wait node: %s\n", hipGetErrorName(e));
    (void)hipGetLastError();
    return -(1000 + (int)e);
}

MCAV_EXPORT int mcav_stream_wait_event(void* stream, void* event) {
    if (!event) return MCAV_E_INVALID;
    if (hipStreamWaitEvent(as_stream(stream), reinterpret_cast<hipEvent_t>(event), 0) != hipSuccess) { (void)hipGetLastError(); return MCAV_E_LAUNCH; }
    return MCAV_OK;
}

