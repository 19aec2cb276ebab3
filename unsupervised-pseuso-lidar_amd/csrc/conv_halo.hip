// Halo-tile 3x3 stride-1 convolution for the NARROW, high-resolution decoder layers (16 / 32 channels at 96x320 .. 192x640:
// reference layers.py:22-58 ConvBlock / Conv3x3 at the last two decoder levels, resnet_dispnet.py:81-96).
//
// In the general implicit-GEMM kernel these layers have only 9..18 K-tiles per workgroup: table building, the cold first loads
// and the epilogue are paid for a handful of MFMAs, and every input pixel is fetched nine times through L1/L2.  Here a
// workgroup owns an 8 x 32 pixel output tile and
//   * loads its 10 x 34 input halo ONCE into LDS (padding / reflection / the fused nearest-upsample resolved in that one pass),
//   * keeps the whole filter (9 taps x C x N) in registers as MFMA B operands,
//   * reads every A operand of the 9 taps straight from the halo with ds_read_b128 (pixel rows padded to C + 4 floats:
//     conflict-free) -- one read feeds 4 MFMAs (lane-group g takes channels 4g..4g+3, A and B permuted identically),
//   * has no K loop, no barrier after the one that publishes the halo, and stores through the usual fused epilogue.
// v_mfma_f32_16x16x4_f32; a wavefront owns 2 rows x 32 pixels = 4 fragments of 16 consecutive pixels.
#include "conv_gather.h"
#include "kernel_timer.h"

namespace mcav {

constexpr int HT_H = 8, HT_W = 32;                  // output tile
constexpr int HH = HT_H + 2, HW = HT_W + 2;         // halo

struct HaloParams {
    const float* x;       // source NHWC [B, Hs(/2), Ws(/2), C]
    int B, H, W;          // logical (output = input) image size
    int up;               // the source is stored at half resolution (fused nearest upsample)
    int pad_mode;
    const float* w;       // packed [Np][9][C]
    int n_begin;          // first packed filter row of this launch (channel-range split of a concatenated input's gradient)
    const float* bias;
    int act;
    float* y;
    int Cd, n_count;      // output row stride and real channels
    int co0;              // first output channel of this launch (wide outputs are produced in blocks of 32 channels)
    const float* dact_aux;
    int dact;
    const float* addend;
    int pool;
    int tiles_x, tiles_y;
};

// ADJ = false: forward gather (zero / reflection padding, optional fused nearest upsample).
// ADJ = true : the data gradient of the reflection-padded conv.  The source is the output gradient with ZERO padding and the
//   taps are mirrored (dx[p] collects dy[p + 1 - k] w[k]); on top of that the image's second / second-to-last lines also collect
//   the border line's outputs whose padded tap reflected onto them (line 1 <- line 0 under tap 0, line H-2 <- line H-1 under
//   tap 2; same for columns; corners get the product).  Rows are whole fragments (extra MFMAs on a fragment-uniform test),
//   columns are single lanes of a fragment (the A operand is masked to that lane).
//   Epilogue of the gradient: 2x2 sum-pool (adjoint of the nearest upsample), * act'(aux), + addend.
template <int C, int NF, bool ADJ>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(HaloParams p) {
    constexpr int LDP = C + 4;                      // floats per halo pixel
    constexpr int C4 = C / 4, KC = C / 16;          // 16-byte columns per pixel; 16-channel chunks
    __shared__ __attribute__((aligned(16))) float halo[HH * HW * LDP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = lid % p.tiles_x, ty = (lid / p.tiles_x) % p.tiles_y, n = lid / (p.tiles_x * p.tiles_y);
    const int y0 = ty * HT_H, x0 = tx * HT_W;

    // ---- the filter: lane (column nn = lane & 15, group g = lane >> 4) holds w[nn][tap][16 kc + 4 g .. + 3] for every tap
    const int nn = lane & 15, g = lane >> 4;
    f32x4 wv[NF][9][KC];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
                wv[f][t][kc] = *reinterpret_cast<const f32x4*>(p.w + ((size_t)(p.n_begin + p.co0 + f * 16 + nn) * 9 + t) * C + kc * 16 + g * 4);

    // ---- halo: one pass, all loads in flight together
    const int Hs = p.up ? p.H >> 1 : p.H, Ws = p.up ? p.W >> 1 : p.W;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.x, (unsigned)((size_t)p.B * Hs * Ws * C * 4));
    constexpr int NLD = (HH * HW * C4 + 255) / 256;
    f32x4 hv[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int i = tid + 256 * j;
        const int pix = i / C4, c4 = i - pix * C4;
        const int hy = pix / HW, hx = pix - hy * HW;
        int sy = y0 - 1 + hy, sx = x0 - 1 + hx;
        bool ok = i < HH * HW * C4;
        if (!ADJ && p.pad_mode == MCAV_PAD_REFLECT) {
            sy = reflect_idx(sy, p.H); sx = reflect_idx(sx, p.W);
            sy = min(max(sy, 0), p.H - 1); sx = min(max(sx, 0), p.W - 1);       // rows of a ragged last tile: anything valid
        } else {
            ok = ok && (unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W;
        }
        if (p.up) { sy >>= 1; sx >>= 1; }
        hv[j] = buf_load4(rs, ok ? (unsigned)((((n * Hs + sy) * Ws + sx) * C + c4 * 4) * 4) : OOB);
    }
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int i = tid + 256 * j;
        const int pix = i / C4, c4 = i - pix * C4;
        if (i < HH * HW * C4) *reinterpret_cast<f32x4*>(&halo[pix * LDP + c4 * 4]) = hv[j];
    }
    __syncthreads();

    // ---- 9 taps x 4 fragments: fragment f = (row 2 wave + (f >> 1), columns 16 (f & 1) ..); lane row r = pixel column
    f32x4 acc[4][NF];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int h = 0; h < NF; ++h) acc[f][h] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r = lane & 15;
    auto mac = [&](int f, int hy, int hx, int tap, bool keep) {      // acc[f] += halo(hy, hx)[.] x w[tap]; keep = false zeroes this lane's pixel
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            f32x4 a = *reinterpret_cast<const f32x4*>(&halo[(hy * HW + hx) * LDP + kc * 16 + g * 4]);
            if (!keep) a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int h = 0; h < NF; ++h) {
                const f32x4 b = wv[h][tap][kc];
                acc[f][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[f][h], 0, 0, 0);
                acc[f][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[f][h], 0, 0, 0);
                acc[f][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[f][h], 0, 0, 0);
                acc[f][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[f][h], 0, 0, 0);
            }
        }
    };
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                // forward: source = p + k - 1 -> halo (row + ky, col + kx); adjoint: source = p + 1 - k -> halo (row + 2 - ky, col + 2 - kx)
                const int hy = 2 * wave + (f >> 1) + (ADJ ? 2 - ky : ky), hx = 16 * (f & 1) + r + (ADJ ? 2 - kx : kx);
                mac(f, hy, hx, ky * 3 + kx, true);
            }
    if constexpr (ADJ) {
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int fy = 2 * wave + (f >> 1), py = y0 + fy;                  // fragment row (tile / image)
            const int fx = 16 * (f & 1), px = x0 + fx + r;                    // this lane's pixel column
            // extra source lines of this fragment row: image line 0 under ky = 0 (rows on line 1), line H-1 under ky = 2 (line H-2)
            const bool y0e = py == 1, y2e = py == p.H - 2;
            const int hy0 = 1 - y0, hy2 = p.H - y0;                            // their halo rows (the halo starts at image line y0 - 1)
            // extra source columns of single lanes: column 0 under kx = 0 (pixels of column 1), column W-1 under kx = 2 (column W-2)
            const bool x0e = px == 1, x2e = px == p.W - 2;
            const int hx0 = 1 - x0, hx2 = p.W - x0;
            const bool f0 = __any(x0e), f2 = __any(x2e);                        // fragment-uniform: run the masked MFMAs at all?
            if (y0e) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) mac(f, hy0, fx + r + 2 - kx, 0 * 3 + kx, true);
            }
            if (y2e) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) mac(f, hy2, fx + r + 2 - kx, 2 * 3 + kx, true);
            }
            if (f0) {
                const int cx = x0e ? hx0 : 0;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) mac(f, fy + 2 - ky, cx, ky * 3 + 0, x0e);
                if (y0e) mac(f, hy0, cx, 0 * 3 + 0, x0e);
                if (y2e) mac(f, hy2, cx, 2 * 3 + 0, x0e);
            }
            if (f2) {
                const int cx = x2e ? hx2 : 0;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) mac(f, fy + 2 - ky, cx, ky * 3 + 2, x2e);
                if (y0e) mac(f, hy0, cx, 0 * 3 + 2, x2e);
                if (y2e) mac(f, hy2, cx, 2 * 3 + 2, x2e);
            }
        }
    }

    // ---- epilogue.  C/D layout 16x16: column = lane & 15 (output channel), rows 4 (lane >> 4) + e (pixel column within the fragment).
    // Raw buffer accesses (round 4: an out-of-range element gets the OOB offset and is dropped in hardware -- no branch and no 64-bit address per
    // element; the tensors are < 2 GiB, checked on the host).
    const unsigned ybytes = (unsigned)((size_t)p.B * ((ADJ && p.pool) ? (p.H >> 1) * (p.W >> 1) : p.H * p.W) * p.Cd * 4);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.y, ybytes);
    const __amdgpu_buffer_rsrc_t rax = make_rsrc(p.dact_aux ? p.dact_aux : p.y, ybytes);
    const __amdgpu_buffer_rsrc_t rad = make_rsrc(p.addend ? p.addend : p.y, ybytes);
#pragma unroll
    for (int h = 0; h < NF; ++h) {
        const int co = p.co0 + h * 16 + nn;
        const bool cok = co < p.n_count;
        const float bv = (p.bias && cok) ? p.bias[co] : 0.f;
        if (ADJ && p.pool) {
            // 2x2 sums: rows 2 wave, 2 wave + 1 are fragments f and f + 2; columns 4 g + {0,1} and {2,3} sit in one lane
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const int oy = (y0 >> 1) + wave;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int ox = (x0 >> 1) + 8 * f + 2 * g + e;
                    const bool ok = cok && oy < (p.H >> 1) && ox < (p.W >> 1);
                    float v = (acc[f][h][2 * e] + acc[f][h][2 * e + 1]) + (acc[f + 2][h][2 * e] + acc[f + 2][h][2 * e + 1]);
                    const unsigned off = ok ? (unsigned)((((n * (p.H >> 1) + oy) * (p.W >> 1) + ox) * p.Cd + co) * 4) : OOB;
                    if (p.dact_aux) v *= act_bwd(buf_load1(rax, off), p.dact);
                    if (p.addend) v += buf_load1(rad, off);
                    buf_store1(ry, off, v);
                }
            }
        } else {
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const int oy = y0 + 2 * wave + (f >> 1);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ox = x0 + 16 * (f & 1) + 4 * g + e;
                    const bool ok = cok && oy < p.H && ox < p.W;
                    const unsigned off = ok ? (unsigned)((((n * p.H + oy) * p.W + ox) * p.Cd + co) * 4) : OOB;
                    float v = act_fwd(acc[f][h][e] + bv, p.act);
                    if (ADJ) {
                        if (p.dact_aux) v *= act_bwd(buf_load1(rax, off), p.dact);
                        if (p.addend) v += buf_load1(rad, off);
                    }
                    buf_store1(ry, off, v);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ fused upsample, merged taps
// Forward 3x3 reflection-padded conv of a nearest-2x-UPSAMPLED map (decoder conv (0,1): reference layers.py:49-58 upsample +
// resnet_dispnet.py:88-92).  On the upsampled grid several of the nine taps of an output pixel read the SAME source pixel: for output
// parity py the rows ky = 0 | 1,2 (py = 0) or ky = 0,1 | 2 (py = 1) coincide, likewise in x.  Per parity class the conv is therefore a
// 2x2 conv on the low-resolution map with pre-summed filters -- 4 taps instead of 9 (2.25x fewer MFMAs), a 6 x 18 low-resolution halo
// instead of a 10 x 34 upsampled one, and reflection padding of the upsampled grid becomes clamping of the source index.
// A fragment is 16 same-parity pixels of one row (x = x0 + 2 r + px), so its filter set is uniform; the four fragments of a wavefront
// (2 rows x 2 column parities) are exactly the four classes.  The sums w[ky] + w[ky'] are formed in registers from the packed filter
// (they differ from the 9-tap result only by fp32 rounding of the weight sums).
template <int C, int NF>
__global__ __launch_bounds__(256) void conv3x3_halo_up_kernel(HaloParams p) {
    constexpr int LDP = C + 4, C4 = C / 4, KC = C / 16;
    constexpr int LH = HT_H / 2 + 2, LW = HT_W / 2 + 2;       // low-resolution halo: 6 x 18
    __shared__ __attribute__((aligned(16))) float halo[LH * LW * LDP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = lid % p.tiles_x, ty = (lid / p.tiles_x) % p.tiles_y, n = lid / (p.tiles_x * p.tiles_y);
    const int y0 = ty * HT_H, x0 = tx * HT_W;
    const int Hs = p.H >> 1, Ws = p.W >> 1;
    const int nn = lane & 15, g = lane >> 4;

    // ---- merged filters: class (py, px) x merged tap (a, b); set S(parity, a): (0,0) = {0}, (0,1) = {1,2}, (1,0) = {0,1}, (1,1) = {2}
    f32x4 wm[4][NF][4][KC];
#pragma unroll
    for (int h = 0; h < NF; ++h)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            f32x4 w9[9];
#pragma unroll
            for (int t = 0; t < 9; ++t)
                w9[t] = *reinterpret_cast<const f32x4*>(p.w + ((size_t)(p.co0 + h * 16 + nn) * 9 + t) * C + kc * 16 + g * 4);
#pragma unroll
            for (int cls = 0; cls < 4; ++cls) {
                const int py = cls >> 1, px = cls & 1;
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) {
                                const bool iny = py == 0 ? (a == 0 ? ky == 0 : ky >= 1) : (a == 0 ? ky <= 1 : ky == 2);
                                const bool inx = px == 0 ? (b == 0 ? kx == 0 : kx >= 1) : (b == 0 ? kx <= 1 : kx == 2);
                                if (iny && inx) sum += w9[ky * 3 + kx];
                            }
                        wm[cls][h][a * 2 + b][kc] = sum;
                    }
            }
        }

    // ---- low-resolution halo (source rows y0/2 - 1 .., clamped)
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.x, (unsigned)((size_t)p.B * Hs * Ws * C * 4));
    constexpr int NLD = (LH * LW * C4 + 255) / 256;
    f32x4 hv[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int i = tid + 256 * j;
        const int pix = i / C4, c4 = i - pix * C4;
        const int hy = pix / LW, hx = pix - hy * LW;
        const int sy = min(max((y0 >> 1) - 1 + hy, 0), Hs - 1), sx = min(max((x0 >> 1) - 1 + hx, 0), Ws - 1);
        hv[j] = buf_load4(rs, i < LH * LW * C4 ? (unsigned)((((n * Hs + sy) * Ws + sx) * C + c4 * 4) * 4) : OOB);
    }
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int i = tid + 256 * j;
        const int pix = i / C4, c4 = i - pix * C4;
        if (i < LH * LW * C4) *reinterpret_cast<f32x4*>(&halo[pix * LDP + c4 * 4]) = hv[j];
    }
    __syncthreads();

    // ---- fragment f: output row y0 + 2 wave + (f >> 1) (py = f >> 1), columns x0 + 2 r + (f & 1)
    f32x4 acc[4][NF];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int h = 0; h < NF; ++h) acc[f][h] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r = lane & 15;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int py = f >> 1, px = f & 1;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(&halo[((wave + py + a) * LW + r + px + b) * LDP + kc * 16 + g * 4]);
#pragma unroll
                    for (int h = 0; h < NF; ++h) {
                        const f32x4 bv = wm[f][h][a * 2 + b][kc];
                        acc[f][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[f][h], 0, 0, 0);
                        acc[f][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[f][h], 0, 0, 0);
                        acc[f][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc[f][h], 0, 0, 0);
                        acc[f][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc[f][h], 0, 0, 0);
                    }
                }
    }
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.y, (unsigned)((size_t)p.B * p.H * p.W * p.Cd * 4));
#pragma unroll
    for (int h = 0; h < NF; ++h) {
        const int co = p.co0 + h * 16 + nn;
        const bool cok = co < p.n_count;
        const float bv = (p.bias && cok) ? p.bias[co] : 0.f;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int oy = y0 + 2 * wave + (f >> 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ox = x0 + 2 * (4 * g + e) + (f & 1);
                const bool ok = cok && oy < p.H && ox < p.W;
                buf_store1(ry, ok ? (unsigned)((((n * p.H + oy) * p.W + ox) * p.Cd + co) * 4) : OOB, act_fwd(acc[f][h][e] + bv, p.act));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dw[co][tap][ci] = sum_p dy[p][co] x[src(p, tap)][ci] for the same narrow layers.  Per filter tap a 16x16 (ci x co) MFMA tile
// whose reduction runs over PIXELS: a workgroup walks over 8 x 32 pixel tiles (grid-stride), stages the input halo and the
// dy tile in LDS once per tile, and every wavefront accumulates all 9 taps over its 64 pixels -- the dy fragment of a
// 4-pixel k-step is read once and reused by the 9 taps.  LDS pixel strides (16 / 48 floats) put the 4 pixel groups of a
// ds_read_b32 on disjoint banks.  At the end the 4 wavefronts are summed through LDS and the workgroup writes ONE partial
// in the slab layout of wgrad_kernel ([split][tap * Kp + ci][co], bias row last), so the same fixed-order reduction follows.
struct HaloWgradParams {
    const float* x;
    int B, H, W, up, pad_mode;
    const float* dy;
    int Cdy, dy_choff, Cout;
    float* slab;
    int slabN, Ktot, want_bias;
    int tiles_x, tiles_y, tiles;
};

template <int C, int NF>
__global__ __launch_bounds__(256) void conv3x3_halo_wgrad_kernel(HaloWgradParams p) {
    constexpr int LDX = C == 16 ? 16 : 48;          // floats per halo pixel (bank-disjoint pixel groups, see above)
    constexpr int LDY = NF == 1 ? 16 : 48;          // floats per dy pixel
    constexpr int C4 = C / 4, MFR = C / 16, N4 = NF * 4;
    constexpr int XS = HH * HW * LDX, YS = HT_H * HT_W * LDY;
    __shared__ __attribute__((aligned(16))) float lds[XS + YS];
    float* const xs = lds;
    float* const ys = lds + XS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kk = lane >> 4;

    f32x4 acc[9][MFR][NF];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int a = 0; a < MFR; ++a)
#pragma unroll
            for (int b = 0; b < NF; ++b) acc[t][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

    const int Hs = p.up ? p.H >> 1 : p.H, Ws = p.up ? p.W >> 1 : p.W;
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(p.x, (unsigned)((size_t)p.B * Hs * Ws * C * 4));
    const __amdgpu_buffer_rsrc_t rsy = make_rsrc(p.dy, (unsigned)((size_t)p.B * p.H * p.W * p.Cdy * 4));
    constexpr int NLX = (HH * HW * C4 + 255) / 256;
    constexpr int NLY = (HT_H * HT_W * N4 + 255) / 256;

    for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, n = tile / (p.tiles_x * p.tiles_y);
        const int y0 = ty * HT_H, x0 = tx * HT_W;
        f32x4 hv[NLX], dv[NLY];
#pragma unroll
        for (int j = 0; j < NLX; ++j) {
            const int i = tid + 256 * j;
            const int pix = i / C4, c4 = i - pix * C4;
            const int hy = pix / HW, hx = pix - hy * HW;
            int sy = y0 - 1 + hy, sx = x0 - 1 + hx;
            bool ok = i < HH * HW * C4;
            if (p.pad_mode == MCAV_PAD_REFLECT) {
                sy = reflect_idx(sy, p.H); sx = reflect_idx(sx, p.W);
                sy = min(max(sy, 0), p.H - 1); sx = min(max(sx, 0), p.W - 1);
            } else {
                ok = ok && (unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W;
            }
            if (p.up) { sy >>= 1; sx >>= 1; }
            hv[j] = buf_load4(rsx, ok ? (unsigned)((((n * Hs + sy) * Ws + sx) * C + c4 * 4) * 4) : OOB);
        }
#pragma unroll
        for (int j = 0; j < NLY; ++j) {      // dy tile; pixels past the image (ragged tiles) and channels past Cout read as zero
            const int i = tid + 256 * j;
            const int pix = i / N4, c4 = i - pix * N4;
            const int ly = pix / HT_W, lx = pix - ly * HT_W;
            const bool ok = i < HT_H * HT_W * N4 && y0 + ly < p.H && x0 + lx < p.W && c4 * 4 < p.Cout;
            dv[j] = buf_load4(rsy, ok ? (unsigned)((((n * p.H + y0 + ly) * p.W + x0 + lx) * p.Cdy + p.dy_choff + c4 * 4) * 4) : OOB);
        }
        __syncthreads();                     // the previous tile's MFMAs have read the panels
#pragma unroll
        for (int j = 0; j < NLX; ++j) {
            const int i = tid + 256 * j;
            const int pix = i / C4, c4 = i - pix * C4;
            if (i < HH * HW * C4) *reinterpret_cast<f32x4*>(&xs[pix * LDX + c4 * 4]) = hv[j];
        }
#pragma unroll
        for (int j = 0; j < NLY; ++j) {
            const int i = tid + 256 * j;
            const int pix = i / N4, c4 = i - pix * N4;
            if (i < HT_H * HT_W * N4) {
                *reinterpret_cast<f32x4*>(&ys[pix * LDY + c4 * 4]) = dv[j];
                if (p.want_bias) bsum += dv[j];          // 256 % N4 == 0: a thread always holds the same 4 output channels
            }
        }
        __syncthreads();
        // this wavefront's 64 pixels: rows 2 wave, 2 wave + 1; k-step s = 4 consecutive pixels (row 2 wave + (s >> 3), columns 4 (s & 7) ..)
#pragma unroll 4
        for (int s4 = 0; s4 < 16; ++s4) {
            const int ly = 2 * wave + (s4 >> 3), lx = 4 * (s4 & 7) + kk;
            float bval[NF];
#pragma unroll
            for (int b = 0; b < NF; ++b) bval[b] = ys[(ly * HT_W + lx) * LDY + b * 16 + i16];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int a = 0; a < MFR; ++a) {
                        const float av = xs[((ly + ky) * HW + lx + kx) * LDX + a * 16 + i16];
#pragma unroll
                        for (int b = 0; b < NF; ++b)
                            acc[ky * 3 + kx][a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bval[b], acc[ky * 3 + kx][a][b], 0, 0, 0);
                    }
        }
    }

    // ---- sum the 4 wavefronts through LDS (two rounds), then one slab partial per workgroup
    __syncthreads();
    constexpr int NACC = 9 * MFR * NF;                     // f32x4 accumulators per lane
    f32x4* const red = reinterpret_cast<f32x4*>(lds);       // [2][NACC][64]
    static_assert((size_t)2 * NACC * 64 * 16 <= sizeof(float) * (XS + YS), "reduction buffer fits the panels");
    auto dump = [&](int slot) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int a = 0; a < MFR; ++a)
#pragma unroll
                for (int b = 0; b < NF; ++b) red[(slot * NACC + (t * MFR + a) * NF + b) * 64 + lane] = acc[t][a][b];
    };
    auto absorb = [&](int slot) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int a = 0; a < MFR; ++a)
#pragma unroll
                for (int b = 0; b < NF; ++b) acc[t][a][b] += red[(slot * NACC + (t * MFR + a) * NF + b) * 64 + lane];
    };
    if (wave >= 2) dump(wave - 2);
    __syncthreads();
    if (wave < 2) absorb(wave);
    __syncthreads();
    if (wave == 1) dump(0);
    __syncthreads();
    float* const slab = p.slab + (size_t)blockIdx.x * (p.Ktot + 1) * p.slabN;
    if (wave == 0) {
        absorb(0);
        // D layout: row (input channel) = 4 (lane >> 4) + e, column (output channel) = lane & 15
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int a = 0; a < MFR; ++a)
#pragma unroll
                for (int b = 0; b < NF; ++b)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int ci = a * 16 + 4 * kk + e, co = b * 16 + i16;
                        if (co < p.slabN) slab[(size_t)(t * C + ci) * p.slabN + co] = acc[t][a][b][e];
                    }
    }
    if (p.want_bias) {                                      // column sums of dy: threads with the same channel group, fixed order
        __syncthreads();
        red[tid] = bsum;
        __syncthreads();
        if (tid < N4) {
            f32x4 tsum = {0.f, 0.f, 0.f, 0.f};
            for (int k = tid; k < 256; k += N4) tsum += red[k];
            const int co = tid * 4;
            float* o = slab + (size_t)p.Ktot * p.slabN + co;
            if (co + 0 < p.slabN) o[0] = tsum.x;
            if (co + 1 < p.slabN) o[1] = tsum.y;
            if (co + 2 < p.slabN) o[2] = tsum.z;
            if (co + 3 < p.slabN) o[3] = tsum.w;
        }
    }
}

// Weight gradient of the conv on the nearest-2x-UPSAMPLED map (decoder conv (0,1)) with the same merged taps as conv3x3_halo_up_kernel:
// per parity class 4 taps on the low-resolution source (2.25x fewer MFMAs); k-steps are class-pure so every MFMA belongs to one
// (class, merged tap) accumulator; at the end the 16 accumulators are un-merged into the 9 filter taps and reduced as usual.
template <int C, int NF>
__global__ __launch_bounds__(256) void conv3x3_halo_wgrad_up_kernel(HaloWgradParams p) {
    constexpr int LDX = C == 16 ? 16 : 48;          // floats per halo pixel (bank-disjoint pixel groups, see above)
    constexpr int LDY = NF == 1 ? 16 : 48;          // floats per dy pixel
    constexpr int C4 = C / 4, MFR = C / 16, N4 = NF * 4;
    constexpr int LH = HT_H / 2 + 2, LW = HT_W / 2 + 2;       // low-resolution halo of the upsampled source: 6 x 18
    constexpr int XS = LH * LW * LDX, YS = HT_H * HT_W * LDY;
    __shared__ __attribute__((aligned(16))) float lds[XS + YS];
    float* const xs = lds;
    float* const ys = lds + XS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kk = lane >> 4;

    f32x4 accm[4][4][MFR][NF];                               // [parity class][merged tap]
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int a = 0; a < MFR; ++a)
#pragma unroll
                for (int b = 0; b < NF; ++b) accm[c][t][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

    const int Hs = p.up ? p.H >> 1 : p.H, Ws = p.up ? p.W >> 1 : p.W;
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(p.x, (unsigned)((size_t)p.B * Hs * Ws * C * 4));
    const __amdgpu_buffer_rsrc_t rsy = make_rsrc(p.dy, (unsigned)((size_t)p.B * p.H * p.W * p.Cdy * 4));
    constexpr int NLX = (LH * LW * C4 + 255) / 256;
    constexpr int NLY = (HT_H * HT_W * N4 + 255) / 256;

    for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, n = tile / (p.tiles_x * p.tiles_y);
        const int y0 = ty * HT_H, x0 = tx * HT_W;
        f32x4 hv[NLX], dv[NLY];
#pragma unroll
        for (int j = 0; j < NLX; ++j) {      // low-resolution source, clamped (= reflection padding of the upsampled grid)
            const int i = tid + 256 * j;
            const int pix = i / C4, c4 = i - pix * C4;
            const int hy = pix / LW, hx = pix - hy * LW;
            const int sy = min(max((y0 >> 1) - 1 + hy, 0), Hs - 1), sx = min(max((x0 >> 1) - 1 + hx, 0), Ws - 1);
            hv[j] = buf_load4(rsx, i < LH * LW * C4 ? (unsigned)((((n * Hs + sy) * Ws + sx) * C + c4 * 4) * 4) : OOB);
        }
#pragma unroll
        for (int j = 0; j < NLY; ++j) {      // dy tile; pixels past the image (ragged tiles) and channels past Cout read as zero
            const int i = tid + 256 * j;
            const int pix = i / N4, c4 = i - pix * N4;
            const int ly = pix / HT_W, lx = pix - ly * HT_W;
            const bool ok = i < HT_H * HT_W * N4 && y0 + ly < p.H && x0 + lx < p.W && c4 * 4 < p.Cout;
            dv[j] = buf_load4(rsy, ok ? (unsigned)((((n * p.H + y0 + ly) * p.W + x0 + lx) * p.Cdy + p.dy_choff + c4 * 4) * 4) : OOB);
        }
        __syncthreads();                     // the previous tile's MFMAs have read the panels
#pragma unroll
        for (int j = 0; j < NLX; ++j) {
            const int i = tid + 256 * j;
            const int pix = i / C4, c4 = i - pix * C4;
            if (i < LH * LW * C4) *reinterpret_cast<f32x4*>(&xs[pix * LDX + c4 * 4]) = hv[j];
        }
#pragma unroll
        for (int j = 0; j < NLY; ++j) {
            const int i = tid + 256 * j;
            const int pix = i / N4, c4 = i - pix * N4;
            if (i < HT_H * HT_W * N4) {
                *reinterpret_cast<f32x4*>(&ys[pix * LDY + c4 * 4]) = dv[j];
                if (p.want_bias) bsum += dv[j];          // 256 % N4 == 0: a thread always holds the same 4 output channels
            }
        }
        __syncthreads();
        // this wavefront's 64 pixels: rows 2 wave + py; a k-step = 4 SAME-PARITY pixels x = 2 (4 q + kk) + px, so that its MFMA
        // belongs to one (parity class, merged tap) accumulator; source of merged tap (a, b): low-resolution (wave + py + a, 4 q + kk + px + b)
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) {
            const int py = s4 >> 3, px = (s4 >> 2) & 1, q = s4 & 3;
            const int ly = 2 * wave + py, lx = 2 * (4 * q + kk) + px;
            float bval[NF];
#pragma unroll
            for (int b = 0; b < NF; ++b) bval[b] = ys[(ly * HT_W + lx) * LDY + b * 16 + i16];
#pragma unroll
            for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                    for (int a = 0; a < MFR; ++a) {
                        const float av = xs[((wave + py + ta) * LW + 4 * q + kk + px + tb) * LDX + a * 16 + i16];
#pragma unroll
                        for (int b = 0; b < NF; ++b)
                            accm[py * 2 + px][ta * 2 + tb][a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bval[b], accm[py * 2 + px][ta * 2 + tb][a][b], 0, 0, 0);
                    }
        }
    }

    // ---- un-merge: the gradient of filter tap (ky, kx) is the sum over the four classes of the merged tap it belongs to there
    f32x4 acc[9][MFR][NF];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int a = 0; a < MFR; ++a)
#pragma unroll
                for (int b = 0; b < NF; ++b) {
                    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int py = 0; py < 2; ++py)
#pragma unroll
                        for (int px = 0; px < 2; ++px) {
                            const int ta = py == 0 ? (ky == 0 ? 0 : 1) : (ky == 2 ? 1 : 0);
                            const int tb = px == 0 ? (kx == 0 ? 0 : 1) : (kx == 2 ? 1 : 0);
                            sum += accm[py * 2 + px][ta * 2 + tb][a][b];
                        }
                    acc[ky * 3 + kx][a][b] = sum;
                }

    // ---- sum the 4 wavefronts through LDS (two rounds), then one slab partial per workgroup
    __syncthreads();
    constexpr int NACC = 9 * MFR * NF;                     // f32x4 accumulators per lane
    f32x4* const red = reinterpret_cast<f32x4*>(lds);       // [2][NACC][64]
    static_assert((size_t)2 * NACC * 64 * 16 <= sizeof(float) * (XS + YS), "reduction buffer fits the panels");
    auto dump = [&](int slot) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int a = 0; a < MFR; ++a)
#pragma unroll
                for (int b = 0; b < NF; ++b) red[(slot * NACC + (t * MFR + a) * NF + b) * 64 + lane] = acc[t][a][b];
    };
    auto absorb = [&](int slot) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int a = 0; a < MFR; ++a)
#pragma unroll
                for (int b = 0; b < NF; ++b) acc[t][a][b] += red[(slot * NACC + (t * MFR + a) * NF + b) * 64 + lane];
    };
    if (wave >= 2) dump(wave - 2);
    __syncthreads();
    if (wave < 2) absorb(wave);
    __syncthreads();
    if (wave == 1) dump(0);
    __syncthreads();
    float* const slab = p.slab + (size_t)blockIdx.x * (p.Ktot + 1) * p.slabN;
    if (wave == 0) {
        absorb(0);
        // D layout: row (input channel) = 4 (lane >> 4) + e, column (output channel) = lane & 15
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int a = 0; a < MFR; ++a)
#pragma unroll
                for (int b = 0; b < NF; ++b)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int ci = a * 16 + 4 * kk + e, co = b * 16 + i16;
                        if (co < p.slabN) slab[(size_t)(t * C + ci) * p.slabN + co] = acc[t][a][b][e];
                    }
    }
    if (p.want_bias) {                                      // column sums of dy: threads with the same channel group, fixed order
        __syncthreads();
        red[tid] = bsum;
        __syncthreads();
        if (tid < N4) {
            f32x4 tsum = {0.f, 0.f, 0.f, 0.f};
            for (int k = tid; k < 256; k += N4) tsum += red[k];
            const int co = tid * 4;
            float* o = slab + (size_t)p.Ktot * p.slabN + co;
            if (co + 0 < p.slabN) o[0] = tsum.x;
            if (co + 1 < p.slabN) o[1] = tsum.y;
            if (co + 2 < p.slabN) o[2] = tsum.z;
            if (co + 3 < p.slabN) o[3] = tsum.w;
        }
    }
}

}  // namespace mcav

using namespace mcav;

// Called by mcav_igemm (conv_igemm.hip) for the shapes these kernels cover; returns false when they do not apply.
bool mcav_try_halo(const mcav_igemm_desc* d, hipStream_t s) {
    if (d->kh != 3 || d->kw != 3 || d->stride != 1) return false;
    if (d->C2 != 0 || d->x2 || (d->C1 != 16 && d->C1 != 32) || d->Kp != d->C1) return false;
    if (d->n_count > 32 || d->stats || d->groups > 1 || d->y_choff != 0) return false;      // wider outputs: the table-driven kernel is faster
    if (d->Hd != d->Hs || d->Wd != d->Ws || d->Hs < 2 || d->Ws < 2) return false;
    if ((d->tile >> 9) & 1) return false;                      // desc.tile bit 9: force the general kernels (A/B timing, parity of both)
    const bool adj = d->mode == MCAV_G_ADJ_REFLECT;
    if (adj) {
        if (d->dact & ~0xff) return false;                     // MCAV_DACT_AFTER_ADDEND and any later flag: the table-driven kernel's epilogue honours them, this one applies act' first
        if (d->up1 || d->bias || d->act != MCAV_ACT_NONE) return false;
        if (d->pool && ((d->Hs & 1) || (d->Ws & 1))) return false;
    } else {
        if (d->mode != MCAV_G_DIRECT || d->sign != 1 || d->offset != -1) return false;
        if (d->n_begin != 0 || d->pool || d->dact_aux || d->addend) return false;
    }
    // 32 input channels x 32 output channels would hold 144 filter registers per lane (one wavefront per SIMD): two launches of 16
    const int nf = (d->n_count > 16 && d->C1 == 16) ? 2 : 1;
    if (d->n_begin + (d->n_count + 15) / 16 * 16 > d->Np) return false;            // every filter row a launch touches exists in the packed copy
    HaloParams p;
    p.x = d->x1; p.B = d->B; p.H = d->Hs; p.W = d->Ws; p.up = d->up1; p.pad_mode = d->pad_mode;
    p.w = d->w; p.n_begin = d->n_begin; p.bias = d->bias; p.act = d->act; p.y = d->y; p.Cd = d->Cd; p.n_count = d->n_count;
    p.dact_aux = d->dact_aux; p.dact = d->dact; p.addend = d->addend; p.pool = d->pool;
    p.tiles_x = (p.W + HT_W - 1) / HT_W; p.tiles_y = (p.H + HT_H - 1) / HT_H;
    const int grid = p.B * p.tiles_x * p.tiles_y;
#define HALO_LAUNCH(CC, NN) \
    do { if (adj) timed_launch(conv3x3_halo_kernel<CC, NN, true>, grid, dim3(256), 0, s, p); else timed_launch(conv3x3_halo_kernel<CC, NN, false>, grid, dim3(256), 0, s, p); } while (0)
    // upsampled source + reflection padding: 4 merged taps on the low-resolution map (desc.tile bit 10 keeps the 9-tap kernel)
    if (!adj && d->up1 && d->pad_mode == MCAV_PAD_REFLECT && !(d->Hs & 1) && !(d->Ws & 1) && !((d->tile >> 10) & 1)) {
        for (p.co0 = 0; p.co0 < d->n_count; p.co0 += 16 * nf) {
            if (d->C1 == 16 && nf == 1) timed_launch(conv3x3_halo_up_kernel<16, 1>, grid, dim3(256), 0, s, p);
            else if (d->C1 == 16) timed_launch(conv3x3_halo_up_kernel<16, 2>, grid, dim3(256), 0, s, p);
            else timed_launch(conv3x3_halo_up_kernel<32, 1>, grid, dim3(256), 0, s, p);
        }
        return true;
    }
    for (p.co0 = 0; p.co0 < d->n_count; p.co0 += 16 * nf) {
        if (d->C1 == 16 && nf == 1) HALO_LAUNCH(16, 1);
        else if (d->C1 == 16) HALO_LAUNCH(16, 2);
        else HALO_LAUNCH(32, 1);
    }
#undef HALO_LAUNCH
    return true;
}

// Weight-gradient counterpart (called by mcav_wgrad's planner / launcher in conv_igemm.hip).
int mcav_halo_wgrad_splits(const mcav_wgrad_desc* d) {      // 0 = not applicable, else the number of slab partials (= workgroups)
    if (d->mode != MCAV_G_DIRECT || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->sign != 1 || d->offset != -1) return 0;
    if (d->C2 != 0 || d->x2 || (d->C1 != 16 && d->C1 != 32) || d->Kp != d->C1 || d->Cin != d->C1) return 0;
    if (d->Cout > 32 || (d->C1 == 32 && d->Cout > 16)) return 0;
    if (d->Hd != d->Hs || d->Wd != d->Ws || d->Hs < 2 || d->Ws < 2) return 0;
    if ((d->Cdy & 3) || (d->dy_choff & 3) || ((d->tile >> 9) & 1)) return 0;
    const int cl = (d->Cout + 3) / 4 * 4;
    if (cl > d->Cdy - d->dy_choff) return 0;                 // the 16-byte dy loads need the (zero) padding channels to exist
    const long tiles = (long)d->B * ((d->Hs + HT_H - 1) / HT_H) * ((d->Ws + HT_W - 1) / HT_W);
    return (int)(tiles < 512 ? tiles : 512);
}

void mcav_halo_wgrad_launch(const mcav_wgrad_desc* d, float* slab, int slabN, int splits, hipStream_t s) {
    HaloWgradParams p;
    p.x = d->x1; p.B = d->B; p.H = d->Hs; p.W = d->Ws; p.up = d->up1; p.pad_mode = d->pad_mode;
    p.dy = d->dy; p.Cdy = d->Cdy; p.dy_choff = d->dy_choff; p.Cout = d->Cout;
    p.slab = slab; p.slabN = slabN; p.Ktot = 9 * d->Kp; p.want_bias = d->dbias != nullptr;
    p.tiles_x = (p.W + HT_W - 1) / HT_W; p.tiles_y = (p.H + HT_H - 1) / HT_H; p.tiles = p.B * p.tiles_x * p.tiles_y;
    if (p.up && p.pad_mode == MCAV_PAD_REFLECT && !(p.H & 1) && !(p.W & 1) && d->C1 == 16 && d->Cout <= 16 && !((d->tile >> 10) & 1))
        timed_launch(conv3x3_halo_wgrad_up_kernel<16, 1>, dim3(splits), dim3(256), 0, s, p);
    else if (d->C1 == 16 && d->Cout <= 16) timed_launch(conv3x3_halo_wgrad_kernel<16, 1>, dim3(splits), dim3(256), 0, s, p);
    else if (d->C1 == 16) timed_launch(conv3x3_halo_wgrad_kernel<16, 2>, dim3(splits), dim3(256), 0, s, p);
    else timed_launch(conv3x3_halo_wgrad_kernel<32, 1>, dim3(splits), dim3(256), 0, s, p);
}
