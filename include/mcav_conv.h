/* mcav_conv.h -- C ABI of the network kernels of libmcav_depth.so (gfx950): implicit-GEMM convolution on the
 * fp32 MFMA (v_mfma_f32_32x32x2_f32 / 16x16x4_f32), its data- and weight-gradients, BatchNorm, pooling, layout
 * and the optimiser.  These replace the torch.nn / torchvision ops the reference's networks are made of:
 *   models/depth/resnet_dispnet.py:12-107 (ResnetEncoder via torchvision resnet, DepthDecoder, DispResNet)
 *   models/depth/layers.py:22-58 (Conv3x3 reflection pad + conv, ConvBlock ELU, nearest upsample)
 *   models/pose/pose_net.py:31-77, models/pose/pose_fc.py:21-84, models/depth/disp_net.py:51-141
 *   trainer.py:71-76,261-266 (Adam, zero_grad, step)
 *
 * Activations are NHWC fp32 ("pixel-major"): element (b, y, x, c) at ((b*H + y)*W + x)*C + c.
 * Conv weights are consumed in the packed form [Np][kh*kw][Kp] (output channel, tap, input channel; Np and Kp
 * are the channel counts rounded up to 16, zero filled) produced by mcav_pack_weights from the reference's
 * OIHW parameters; weight gradients are written back in OIHW.
 * All pointers are device memory; `stream` is a hipStream_t; return codes as in mcav_depth.h.
 */
#ifndef MCAV_CONV_H
#define MCAV_CONV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* gather modes: how a destination pixel d and a filter tap k address the source */
#define MCAV_G_DIRECT 0      /* s = d*stride + sign*k + offset per axis; zero or reflection padding */
#define MCAV_G_SMALLC 1      /* as DIRECT, zero padding, source has exactly 4 channels (image stem) */
#define MCAV_G_ADJ_REFLECT 2 /* adjoint of a 3x3 stride-1 reflection-padded conv (border pixels sum 2 sources per axis) */
#define MCAV_G_ADJ_STRIDE2 3 /* adjoint of a stride-2 zero-padded conv: s = (d + pad - k)/2 when even; parity-class tiles */

#define MCAV_PAD_ZERO 0
#define MCAV_PAD_REFLECT 1

#define MCAV_ACT_NONE 0
#define MCAV_ACT_RELU 1
#define MCAV_ACT_ELU 2
#define MCAV_ACT_SIGMOID 3

/* One gather-GEMM launch: y[pix, n] = epilogue( sum_{tap, c} src(pix, tap)[c] * w[n][tap][c] ).
 * Used for the forward convolution and (with the adjoint gather modes / transposed packed weights) for dgrad. */
#define MCAV_DACT_AFTER_ADDEND 0x100

typedef struct mcav_igemm_desc {
    /* source: logical [B, Hs, Ws, C1 + C2]; channels [0, C1) come from x1, [C1, C1+C2) from x2 (fused concat) */
    const float* x1;
    const float* x2;
    int B, Hs, Ws, C1, C2;
    int up1;          /* 1: x1 is stored at (Hs/2, Ws/2) and nearest-upsampled x2 on the fly (decoder) */
    /* packed filter [Np][kh*kw][Kp], Kp >= C1 + C2 */
    const float* w;
    int kh, kw, Np, Kp;
    int mode, stride, sign, offset, pad_mode;
    /* destination: logical [B, Hd, Wd, *]; this launch computes filter rows [n_begin, n_begin + n_count) and writes
     * them to channels [y_choff, y_choff + n_count) of y, whose pixel stride is Cd floats */
    float* y;
    int Hd, Wd, Cd, n_begin, n_count, y_choff;
    /* epilogue, in this order: + bias[n]; activation; * act'(dact_aux) ; + addend; 2x2 sum pooling */
    const float* bias;
    int act;
    const float* dact_aux;  /* same layout as y; dact = MCAV_ACT_* whose derivative (as a function of the OUTPUT) multiplies */
    int dact;               /* | MCAV_DACT_AFTER_ADDEND: the factor multiplies AFTER the addend has been added ((conv + addend) * act'(aux): the gradient
                             * arriving at a residual block's output ReLU is the sum of two branches) */
    const float* addend;    /* same layout as y */
    int pool;               /* 1: destination pixels are visited in 2x2 blocks and summed: y is [B, Hd/2, Wd/2, Cd] */
    float* stats;           /* NULL or [mtiles][2][n_count]: per-tile column sums of y and y^2 (BatchNorm batch statistics) */
    int tile;               /* 0 = choose automatically; else a tile-config id in the low byte; bits 8-15 = test / tuning switches that force or forbid one
                             * of the kernel forms (bit 13: the fp32 patch-in-LDS kernel for 3x3 stride-1 zero-padded launches; bit 14: the second
                             * patch kernel of the split form -- 128- / 256-row tiles, LDS-DMA filter ring, one workgroup per CU --, bit 15: its
                             * 128 x 32 configuration at two workgroups per CU) */
    int groups;             /* 0/1 = one group.  G > 1: the batch is G equal groups (e.g. the tgt and ref0 passes of the depth net run as
                             * one launch); output tiles never straddle a group, so `stats` rows [g * mtiles/G, (g+1) * mtiles/G) belong to
                             * group g (per-pass BatchNorm statistics).  Only with the DIRECT / SMALLC gathers and pool == 0. */
    const float* w_upmerge; /* NULL, or mcav_pack_weights_upmerge's copy of the same filter: with up1 = 1, a 3x3 stride-1 reflection-padded
                             * DIRECT gather and a second source x2, the upsampled part of the contraction then runs as FOUR merged taps on
                             * the low-resolution x1 instead of nine on the upsampled one (on the upsampled grid several taps of an output
                             * pixel read the same source pixel; per output parity class their filters are pre-summed).  Same result up to
                             * fp32 rounding of the filter sums; ignored where the launch does not qualify. */
    int mma;                /* 0 = fp32 MFMA (exact fp32 products).  1 = bf16 MFMA tiles (BASELINE.json configs[2] / [4]): the source pixels are
                             * rounded to bf16 on their way into LDS, the filter comes from w16, accumulation / epilogue / outputs stay fp32.
                             * Launches the bf16 kernels do not cover (image stem, narrow high-resolution layers, 1-channel heads, pooled and
                             * merged-tap forms) run the fp32 kernels with w; mcav_igemm_uses_bf16() tells which.
                             * 2 = the fp32 contraction carried by the bf16 MFMA (round 3): every operand element is split into three bf16 planes
                             * a = h + m + l (h = bf16(a), m = bf16(a - h), l = bf16(a - h - m); the remainder is below 2^-26 |a|) and the six plane
                             * products hh, hm, mh, hl, lh, mm are accumulated in fp32 (dropped: ml, lm, ll <= 2^-26 of a product).  Same launches
                             * as mma = 1, result at least as exact as the fp32 MFMA's (profiles/r03_mfma_split_exactness.txt) at 6 / 16 of its
                             * MFMA time.  Runs where it pays: the 3x3 stride-1 zero-padded convolutions and their data gradients (patch-in-LDS
                             * kernel: operands converted once per chunk); every other launch takes the fp32 kernels with w.
                             * 3 = as 2, on every launch mma = 1 covers (the table-driven kernels convert per tap: measured no faster than the
                             * fp32 MFMA kernels; kept for the parity tests). */
    const void* w16;        /* mma = 1: bf16 copy of the packed filter, same [Np][kh*kw][Kp] layout and row stride in ELEMENTS
                             * (mcav_pack_weights_multi with transposed | 2, or mcav_f32_to_bf16 of a packed fp32 copy).
                             * mma = 2 / 3: three such copies back to back, the planes h, m, l of the packed filter, Np * Kstride elements each
                             * (mcav_pack_weights_multi with transposed | 4, or mcav_f32_to_bf16_planes of a packed fp32 copy) */
    /* BatchNorm BACKWARD statistics in the epilogue of the data gradient that produces dy (round 3; with `stats`): when stats_x is set the
     * second statistic of a column is the sum of y * xhat, xhat = (stats_x - stats_mean[g][n]) * stats_invstd[g][n], instead of the sum of
     * y^2 -- y being what the epilogue stores (after the act'(dact_aux) factor and the addend), stats_x the raw output of the convolution
     * whose BatchNorm is being differentiated (same layout as y), g the tile's group.  The slab then holds exactly the partial sums
     * mcav_bn_bwd_reduce would have produced in a pass of its own over dy, the activations and x; mcav_bn_bwd_finalize sums it. */
    const float* stats_x;
    const float* stats_mean;
    const float* stats_invstd;
    const float* w_stem;    /* NULL, or mcav_pack_stem_weights' copy of the filter: the 7x7 stride-2 stems (the image stem: SMALLC gather, 3 -> 64;
                             * PoseNet conv1: 9 of 16 stored channels -> 16) then run on the patch-in-LDS kernels of conv_stem.hip; ignored by
                             * every other launch */
} mcav_igemm_desc;

/* number of M-tiles (rows of `stats`) the launch will use with its chosen tile config */
int mcav_igemm_mtiles(const mcav_igemm_desc* d);
int mcav_igemm(const mcav_igemm_desc* d, void* stream);
int mcav_igemm_uses_bf16(const mcav_igemm_desc* d);      /* 1: this descriptor (mma = 1 or 2, w16 set) runs on the bf16 MFMA kernels */

/* Weight gradient: dw[n][tap][c] = sum_pix dy[pix, n] * src(pix, tap)[c], reduced over pixel splits and written
 * (accumulated if accumulate != 0) in OIHW [Cout][Cin][kh][kw] to dw_oihw.  The source is gathered exactly as in
 * the forward launch (modes DIRECT / SMALLC). */
typedef struct mcav_wgrad_desc {
    const float* x1;
    const float* x2;
    int B, Hs, Ws, C1, C2, up1;
    int kh, kw, Kp;
    int mode, stride, sign, offset, pad_mode;
    const float* dy;        /* [B, Hd, Wd, Cdy], channels [dy_choff, dy_choff + Cout) are used */
    int Hd, Wd, Cdy, dy_choff;
    int Cout, Cin;          /* true channel counts of the OIHW gradient */
    float* dw_oihw;
    int accumulate;
    float* dbias;           /* NULL or [Cout]: sum over pixels of dy (accumulated if accumulate != 0) */
    int tile;
    /* The filter of conv(cat(up2(a), skip)) in two launches (all zero = one ordinary launch):
     *  - the skip half: an ordinary launch with x1 = skip, Cin = its channels, Cin_total = the filter's input channels, ci_offset = where
     *    the skip channels start in it (dw_oihw then addresses [Cout][Cin_total][kh][kw]);
     *  - the upsampled half, upm = 1: x1 = a stored at [B, Hs/2, Ws/2, C1] (up1 = 1, C2 = 0, 3x3 stride 1 reflection pad, Hd = Hs, Wd = Ws even):
     *    the merged-tap form of mcav_igemm_desc.w_upmerge -- per output parity class 4 taps on the low-resolution a, the reduction running over
     *    low-resolution pixels (2.25x fewer MACs); the 16 partial filters are un-merged into the 9 taps when the slab is reduced.  dbias must be
     *    NULL (the skip launch carries it).  Same result as one ordinary launch up to fp32 summation order. */
    int upm, Cin_total, ci_offset;
    int mma;                /* 1: both operands rounded to bf16, reduction over pixels on the bf16 MFMA, fp32 slab / gradient (launches with
                             * >= 32 output channels and 16-channel-aligned sources; others run the fp32 kernels).
                             * 2: an fp32 contraction on three bf16 planes per operand (mcav_igemm_desc.mma = 2) WHERE THAT FORM IS AHEAD: the
                             *    single-source 3x3 stride-1 launches (zero or reflection padding) with input channels a multiple of 64 and 64
                             *    or more (or exactly 32) outputs run wgrad3x3_patch_kernel (a staged patch read through ds_read_b64_tr_b16);
                             *    every other launch runs the fp32 MFMA kernels.  mcav_wgrad_uses_bf16() tells which.
                             * 3: the split form on every launch the bf16 kernels cover (parity tests).
                             * (1 takes the same patch kernel in its one-plane form where it applies.) */
} mcav_wgrad_desc;

size_t mcav_wgrad_workspace_bytes(const mcav_wgrad_desc* d);
int mcav_wgrad_uses_bf16(const mcav_wgrad_desc* d);
int mcav_wgrad(const mcav_wgrad_desc* d, void* workspace, size_t workspace_bytes, void* stream);

/* Batched slab reduction.  mcav_wgrad = GEMM (partial tiles of the pixel splits into a slab) + one or two reduction launches per layer: 68
 * small launches per step for ResNet-18 + PoseNet.  mcav_wgrad_deferred runs the GEMM only and describes the pending reduction in *item (HOST
 * memory; the workspace must stay untouched until it has been reduced: one workspace per pending layer).  The caller collects the items of a
 * gradient bucket (autograd's backward of reference trainer.py:264 fills them in launch order), calls mcav_wgrad_reduce_plan once on the host
 * array (fills the first-block columns; returns the two grids and the dynamic LDS size), copies the array to the device -- it does not change from
 * step to step, so once -- and reduces all of them with mcav_wgrad_reduce_multi: one presum + one reduce launch, the per-layer kernels' bodies and
 * summation order (bit-identical gradients). */
typedef struct mcav_wgrad_reduce_item {
    const float* slab;
    float* pre;
    float* dw;
    float* dbias;
    unsigned long long elems;
    int splits, groups, per_group, Ktot, slabN, Kp, taps, Cout, Cin, ci_t, accumulate, upm, cin_total, ci_off;
    int pre_bx, red_gx, red_gy, pre_first, red_first, reserved;
} mcav_wgrad_reduce_item;
int mcav_wgrad_deferred(const mcav_wgrad_desc* d, void* workspace, size_t workspace_bytes, mcav_wgrad_reduce_item* item, void* stream);
int mcav_wgrad_reduce_plan(mcav_wgrad_reduce_item* items, int n, int* presum_blocks, int* reduce_blocks, size_t* lds_bytes);
int mcav_wgrad_reduce_multi(const mcav_wgrad_reduce_item* items_dev, int n, int presum_blocks, int reduce_blocks, size_t lds_bytes, void* stream);

/* OIHW [Cout][Cin][kh][kw] -> packed forward filter [Np][taps][Kp] (transposed = 0)
 *                          or packed data-gradient filter [Kp'][taps][Np'] with in/out swapped (transposed = 1). */
int mcav_pack_weights(const float* w_oihw, int Cout, int Cin, int kh, int kw, int transposed, float* packed, int Np, int Kp,
                      void* stream);

/* The same for many filters in ONE launch.  items_dev: device array of nitems records
 *   { const float* src; float* dst; int Cout, Cin, taps, transposed, Np, Kp, Kstride, first_block; }   (48 bytes each)
 * where Kstride = taps * Kp rounded up to 16 and first_block is the running sum of mcav_pack_weights_blocks(...) over the
 * preceding records; nblocks = that sum over all records.  transposed bits: 0 = the data-gradient layout, 1 = bf16 destination, 2 = three
 * bf16 planes h, m, l; bit 3 (8) = the merged-tap forward copy of mcav_pack_weights_upmerge and bit 4 (16) = the adjoint copy of
 * mcav_pack_weights_upmerge_adj, both with C1 in `taps` (Np, Kp as in those entry points; Kstride unused). */
int mcav_pack_weights_multi(const void* items_dev, int nitems, int nblocks, void* stream);
/* A record whose `transposed` has bit 1 set (2 or 3) writes its packed copy as bf16 (same layout, 2-byte elements): the filter copies
 * of the bf16 MFMA kernels, re-derived from the fp32 master weights after every optimiser step in the same launch.  Bit 2 (4 or 5): as
 * three bf16 planes h, m, l of Np * Kstride elements each (mcav_igemm_desc.mma = 2). */
/* round-to-nearest-even fp32 -> bf16 of a flat buffer (the special packed copies: merged-tap adjoint filters) */
int mcav_f32_to_bf16(const float* src, void* dst_bf16, size_t n, void* stream);
/* dst[0 .. n) = h, dst[n .. 2n) = m, dst[2n .. 3n) = l with src = h + m + l up to 2^-26 |src| (the filter planes of mma = 2) */
int mcav_f32_to_bf16_planes(const float* src, void* dst_bf16, size_t n, void* stream);
int mcav_pack_weights_blocks(int taps, int transposed, int Np, int Kp);   /* workgroups one record needs */
/* A 7x7 stride-2 stem filter -- the depth net's conv1 (reference resnet_dispnet.py:38: torchvision resnet, OIHW [64][3][7][7]) or PoseNet's
 * conv1 (pose_net.py:40: [16][9][7][7]) -- in the K order of the patch-in-LDS kernels: [Cin * 56][Cout], row ((c * 7 + ky) * 4 + j) * 2 + h =
 * w[:, c, ky, 2 j + h], zero rows for the padding column kx = 7. */
int mcav_pack_stem_weights(const float* w_oihw, int Cout, int Cin, float* packed, void* stream);
/* Merged-tap copy of a 3x3 filter for mcav_igemm_desc.w_upmerge: packed [4 parity classes (py, px)][Np][4 merged taps (a, b)][C1] with
 * packed[cls][n][a*2+b][c] = sum of w[n][c][ky][kx] over ky in S(py, a), kx in S(px, b);  S(0,0) = {0}, S(0,1) = {1,2}, S(1,0) = {0,1},
 * S(1,1) = {2};  c < C1 (the upsampled source's channels = the filter's first C1 input channels); rows n >= Cout are zero. */
int mcav_pack_weights_upmerge(const float* w_oihw, int Cout, int Cin, int C1, float* packed, int Np, void* stream);
/* The ADJOINT of the same trick: the gradient of conv(reflect_pad(up2(a))) w.r.t. the low-resolution a is a 4x4 stride-2 conv of the
 * output gradient dy (16 taps instead of 4 pixels x 9) on the EDGE-REPLICATED low-resolution domain, followed by folding the one-pixel
 * ring back onto the border it replicates:
 *   1. mcav_pack_weights_upmerge_adj: packed [Np >= C1][16 taps u*4+v][Kp >= Cout], V[u][v][c1][co] = sum of w[co][c1][ky][kx] over
 *      ky in Sy(u), kx in Sy(v);  Sy(0) = {2}, Sy(1) = {1,2}, Sy(2) = {0,1}, Sy(3) = {0};
 *   2. mcav_igemm: x1 = dy [B,H,W,Cout], mode DIRECT, kh = kw = 4, stride 2, sign 1, offset -3, zero padding, Hd = H/2 + 2, Wd = W/2 + 2,
 *      w = that copy  ->  tmp [B, H/2 + 2, W/2 + 2, C1];
 *   3. mcav_upsample_adj_fold: out[s,t] = sum of tmp over {s+1} u ({0} if s = 0) u ({Hl+1} if s = Hl-1) x the same in t, then
 *      * act'(dact_aux) + addend (both optional, same layout as out [B,Hl,Wl,C]). */
int mcav_pack_weights_upmerge_adj(const float* w_oihw, int Cout, int Cin, int C1, float* packed, int Np, int Kp, void* stream);
int mcav_upsample_adj_fold(const float* tmp, int B, int Hl, int Wl, int C, const float* dact_aux, int dact, const float* addend, float* out,
                           void* stream);

/* 3x3 reflection-padded convolution with ONE output channel -- the decoder's disparity heads (reference
 * models/depth/resnet_dispnet.py:66-68 `dispconv`, layers.py:42-58 Conv3x3, applied with a sigmoid at :93-94).  HBM-bound
 * stencil kernels instead of an N = 1 implicit GEMM.  x NHWC [B,H,W,C], C in {16, 32, 64, 128}; w_oihw [1][C][3][3].
 *   fwd: y[B,H,W] = act(conv(x) + bias[0]).
 *   bwd: with dpre = dy * act'(y):  dx = adjoint(dpre) * x_act'(x) + addend   (x_act = the activation that PRODUCED x, its
 *        derivative taken through x; addend may be NULL),  dw_oihw (+)= wgrad,  dbias[0] (+)= sum(dpre)  -- one pass over x.
 *        y = NULL with act = MCAV_ACT_NONE: dy already is dpre (every dpre value is gathered by 9 neighbours: pre-multiplying
 *        it once with mcav_act_bwd halves the gathers). */
int mcav_conv3x3r_c1_fwd(const float* x, int B, int H, int W, int C, const float* w_oihw, const float* bias, int act, float* y,
                         void* stream);
size_t mcav_conv3x3r_c1_bwd_workspace_bytes(int C);
int mcav_conv3x3r_c1_bwd(const float* x, int B, int H, int W, int C, const float* w_oihw, const float* dy, const float* y, int act,
                         int x_act, const float* addend, float* dx, float* dw_oihw, float* dbias, int accumulate, void* workspace,
                         size_t workspace_bytes, void* stream);

/* Measurement hook for bench.py's roofline figure (no reference counterpart).  Between begin() and end() every conv-stage kernel
 * (implicit-GEMM forward / adjoint, weight gradient, halo and 1-channel stencil kernels; not the packing / reduction helpers) is dispatched
 * with a start/stop event pair bound to that dispatch.  count(): dispatches recorded so far.  end(): waits for them, writes their durations
 * in milliseconds, in launch order, to ms[0 .. min(count, capacity)), stops the timer and returns the count (< 0: MCAV_E_*). */
int mcav_kernel_timer_begin(void);
int mcav_kernel_timer_count(void);
int mcav_kernel_timer_end(float* ms, int capacity);

/* NCHW image [B, C, H, W] -> NHWC [B, H, W, Cp] at channel offset choff (other channels untouched; zero the buffer first). */
int mcav_nchw_to_nhwc(const float* src, int B, int C, int H, int W, float* dst, int Cp, int choff, void* stream);
int mcav_nhwc_to_nchw(const float* src, int B, int C, int H, int W, int Cp, int choff, float* dst, void* stream);
/* torch.cat([s0, s1, s2], 1) of three NCHW images (reference pose_net.py:59-61) -> NHWC [B, H, W, Cp], channels past 3 C zeroed: the pose
 * network's input in one pass.  Cp % 4 == 0, 3 C <= Cp. */
int mcav_nchw3_to_nhwc(const float* s0, const float* s1, const float* s2, int B, int C, int H, int W, float* dst, int Cp, void* stream);

/* BatchNorm2d, training mode (torchvision BasicBlock/Bottleneck; batch statistics, eps, momentum as nn.BatchNorm2d).
 * finalize: reduce the per-tile column sums written by mcav_igemm into mean / biased variance, produce
 *   scale = gamma * rsqrt(var + eps), shift = beta - mean * scale, save mean and invstd for backward, and update
 *   running_mean / running_var (unbiased) with `momentum`.  count = pixels per channel. */
int mcav_bn_finalize(const float* stats, int mtiles, int C, double count, const float* gamma, const float* beta, float eps,
                     float momentum, float* running_mean, float* running_var, float* scale, float* shift, float* save_mean,
                     float* save_invstd, int groups, void* workspace, size_t workspace_bytes, void* stream);
/* Layers with many tiles are reduced in two stages inside ONE launch (fp64 partials and one completion ticket per 64 channels in
 * `workspace`): the workspace must be ZERO-FILLED when it is allocated; every launch leaves the tickets at zero, so a cached workspace serves
 * any number of launches issued one after the other on a stream (not two launches at once).  0 bytes = the workspace may be NULL. */
size_t mcav_bn_finalize_workspace_bytes(int mtiles, int C, int groups);
/* groups > 1: `stats` holds groups * mtiles rows, `count` is per group, scale/shift/save_* are [groups][C], and the running
 * statistics are updated once per group, in group order (exactly as `groups` consecutive forward passes would). */
/* eval mode: scale/shift from the running statistics */
int mcav_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps, int C,
                        float* scale, float* shift, void* stream);
/* y = act(x * scale[g][c] + shift[g][c] (+ residual)); n_pix pixels of C channels; act = NONE or RELU;
 * pix_per_group = pixels per group (0 = one group) */
int mcav_bn_apply(const float* x, const float* scale, const float* shift, const float* residual, int act, size_t n_pix, int C, float* y,
                  size_t pix_per_group, void* stream);
/* backward, pass 1: dz = dy * (y > 0 if relu); per-channel sums of dz and dz * xhat -> dgamma, dbeta (accumulated if accumulate) */
size_t mcav_bn_bwd_workspace_bytes(size_t n_pix, int C, int groups);
int mcav_bn_bwd_reduce(const float* dy, const float* y_act, const float* x, const float* save_mean, const float* save_invstd, int relu,
                       size_t n_pix, int C, float* dgamma, float* dbeta, int accumulate, float* sums /* [groups][2][C] out */,
                       int groups, void* workspace, size_t workspace_bytes, void* stream);
/* backward, pass 2: dx = gamma * invstd * (dz - sum_dz / N - xhat * sum_dz_xhat / N); optionally also stores dz (the
 * gradient flowing to the residual branch) to dres (added into it if dres_accumulate) */
/* Second half of mcav_bn_bwd_reduce on its own: partial sums [groups][nblk][2][C] (the reduce kernel's, or the statistics slab of a data
 * gradient launched with stats_x) -> sums [groups][2][C] and the (accumulated) gamma / beta gradients, fixed summation order in float64. */
int mcav_bn_bwd_finalize(const float* partial, int nblk, int C, float* dgamma, float* dbeta, int accumulate, float* sums, int groups, void* stream);
int mcav_bn_bwd_apply(const float* dy, const float* y_act, const float* x, const float* gamma, const float* save_mean,
                      const float* save_invstd, const float* sums, int relu, size_t n_pix, int C, float* dx, float* dres,
                      int dres_accumulate, int groups, void* stream);

/* MaxPool2d(3, stride 2, pad 1) on NHWC; idx stores the winning tap (0..8, first maximum in row-major window order). */
int mcav_maxpool3s2_fwd(const float* x, int B, int H, int W, int C, float* y, uint8_t* idx, void* stream);
/* dx (+)= scatter of dy through idx, gathered per input pixel; then multiplied by (xact > 0) when relu_mask != NULL... */
int mcav_maxpool3s2_bwd(const float* dy, const uint8_t* idx, int B, int H, int W, int C, float* dx, int accumulate, void* stream);

/* elementwise helpers on flat buffers */
int mcav_act_bwd(const float* dy, const float* y, int act, size_t n, float* dx, int accumulate, void* stream);
/* dx[i * stride] = dy[i] * act'(y[i]): a 1-channel gradient map written into channel 0 of a wider (pre-zeroed) NHWC tensor */
int mcav_act_bwd_strided(const float* dy, const float* y, int act, size_t n, float* dx, int stride, void* stream);
int mcav_add(const float* a, const float* b, size_t n, float* out, void* stream);
/* out[b, c] = scale * mean over pixels of x[b, pix, c]  (PoseNet head: mean(3).mean(2) * 0.06), and its backward */
int mcav_spatial_mean(const float* x, int B, int n_pix, int C, float scale, float* out, void* stream);
int mcav_spatial_mean_bwd(const float* dout, int B, int n_pix, int C, float scale, float* dx, void* stream);

/* --- helpers of the secondary networks (DispNetS: models/depth/disp_net.py:97-141; PoseFc: models/pose/pose_fc.py:63-84) and of the
 * multi-scale loss (losses.py:212-216) --- */
/* dst[p, doff + c] (+)= src[p, soff + c] for c < C: channel concat / slice between NHWC tensors of pixel strides Cs and Cd */
int mcav_copy_channels(const float* src, size_t n_pix, int Cs, int soff, float* dst, int Cd, int doff, int C, int accumulate, void* stream);
/* F.interpolate(mode='bilinear', align_corners=False) on 1-channel maps [B, h, w] -> [B, H, W], and its adjoint (gather form).
 * scale_y/x = source step per output pixel; 0 means h/H, w/W (size given); 0.5 = scale_factor 2 followed by a crop to H x W */
int mcav_resize_bilinear_fwd(const float* src, int B, int h, int w, float* dst, int H, int W, float scale_y, float scale_x, void* stream);
int mcav_resize_bilinear_bwd(const float* ddst, int B, int h, int w, float* dsrc, int H, int W, float scale_y, float scale_x, int accumulate,
                             void* stream);
/* `upsample` (reference models/depth/layers.py:55-58: F.interpolate(x, scale_factor=2, mode="nearest")) on `planes` = B*C contiguous
 * [h, w] planes (NCHW) -> [2h, 2w], and its adjoint (each source pixel collects its 2x2 block).  Inside the depth decoder the upsample is
 * fused into the following conv's gather (mcav_igemm_desc.up1); these are the standalone op of the reference's call surface. */
int mcav_upsample_nearest2x(const float* src, size_t planes, int h, int w, float* dst, void* stream);
int mcav_upsample_nearest2x_bwd(const float* grad_out, size_t planes, int h, int w, float* grad_in, void* stream);
int mcav_affine(const float* x, float a, float b, size_t n, float* y, void* stream);      /* y = a x + b */
int mcav_mul(const float* a, const float* b, size_t n, float* y, void* stream);          /* y = a * b */
/* out[c] (+)= sum over pixels of x[p, c]  (bias gradient of ConvTranspose2d) */
size_t mcav_colsum_workspace_bytes(int C);
int mcav_colsum(const float* x, size_t n_pix, int C, float* out, int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* Adam (torch.optim.Adam defaults: betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad) over one flat arena.
 * step is the 1-based step count AFTER this update.  grad_scale multiplies the gradient first (1/world_size for DP). */
int mcav_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1, float beta2,
                   float eps, int step, float grad_scale, void* stream);

/* The same update with its per-step scalars in DEVICE memory, so that the launch can be captured in a hipGraph and replayed (reference site:
 * optimizer.step() at trainer.py:266 inside the captured step).  state8: 8 floats { step count BEFORE this update (incremented by the call),
 * lr, grad_scale, and five words the call owns }.  The host sets step / lr / grad_scale; every call advances the count by one. */
int mcav_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float beta1, float beta2, float eps,
                       float* state8, void* stream);

/* External events of a hipGraph-captured step (BASELINE.json configs[4]: "hipGraph-captured step + overlapped all-reduce"; the reference has no
 * counterpart -- the slot is between loss.backward() and optimizer.step(), trainer.py:264-266).  mcav_event_record_external on a CAPTURING
 * stream adds an external event-record node (hipEventRecordExternal): every replay signals the event when the graph reaches the node, and
 * work outside the graph (a gradient bucket's all-reduce on a communication stream) orders itself behind it with mcav_stream_wait_event.
 * On a stream that is not capturing the record is an ordinary one.  Events are created without timing. */
int mcav_event_create(void** event);
int mcav_event_destroy(void* event);
int mcav_event_record_external(void* event, void* stream);
/* the wait side: on a capturing stream an external event-WAIT node (two captured graphs replayed on two streams order their work through
 * record / wait pairs); otherwise hipStreamWaitEvent */
int mcav_event_wait_external(void* event, void* stream);
int mcav_stream_wait_event(void* stream, void* event);

#ifdef __cplusplus
}
#endif
#endif /* MCAV_CONV_H */
