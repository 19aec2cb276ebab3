/* mcav_depth.h -- C ABI of libmcav_depth.so: the MI355X (gfx950) hot path of the self-supervised
 * depth+pose training step.  Plain pointers and sizes only; no torch types.
 *
 * The reference (Monash-Connected-Autonomous-Vehicle/unsupervised-pseuso-LiDAR) is pure Python/PyTorch and
 * has no FFI of its own: its "operator interface" for this path is the Python call surface
 * trainer.py:290-313 / losses.py:262-271 / geometry/pose_geometry.py:201-229 / models/.  Each entry point
 * below names the reference call it replaces.  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - every pointer is DEVICE memory owned by the caller (torch tensors' data_ptr()), fp32 unless noted;
 *   - image-like tensors at the API boundary are NCHW contiguous (the reference's layout);
 *     network-internal activations are NHWC (see mcav_conv.h section below);
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - return value: 0 on success, negative MCAV_E_* on error; nothing throws across the ABI;
 *   - stateless and thread-safe apart from the caller-provided workspace.
 */
#ifndef MCAV_DEPTH_H
#define MCAV_DEPTH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCAV_OK 0
#define MCAV_E_INVALID (-1)   /* bad argument (null pointer, non-positive size, unsupported shape) */
#define MCAV_E_WORKSPACE (-2) /* workspace too small */
#define MCAV_E_LAUNCH (-3)    /* HIP launch/runtime error (hipGetLastError) */

/* ABI version of this header (bumped on any signature change). */
int mcav_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Loss stage (K10 + K11): fused inverse-warp -> bilinear sample -> L1 photometric -> second-order
 * smoothness, forward AND backward in one pass over the triplet.
 * Replaces Losses.forward (losses.py:262-271) = disp_to_depth (pose_geometry.py:70-95)
 *   + reprojection_loss (losses.py:183-240: 3 x inverse_warp (pose_geometry.py:201-229,
 *     transform.py:74-150, F.grid_sample) + nn.L1Loss) + smooth_loss(depths[0]) (losses.py:242-260)
 * and its autograd backward (trainer.py:264).
 *
 * flags */
#define MCAV_WL_K_F64 1u          /* intrinsics are fp64 [B,3,3] (as the reference's loader gives); else fp32 */
#define MCAV_WL_SKIP_IF_UNIT 2u   /* return without touching outputs when upstream[0]==upstream[1]==1 */
#define MCAV_WL_NO_SMOOTH 4u      /* leave the smoothness term out (used for scales > 0 of multi-scale nets) */
#define MCAV_WL_INPUT_DEPTH 8u    /* disp_t / disp_r0 already hold depths; gradients are w.r.t. depth */
#define MCAV_WL_SSIM 16u          /* photometric term = 0.85 * SSIM distance + 0.15 * L1 (reference losses.py:12-54, weights of :77) instead of L1 */

/* The workspace holds per-workgroup partial sums and the completion tickets of the fused kernel (since round 3 the per-sample constants and
 * the float64 finalize run INSIDE it: one launch).  It must be ZERO-FILLED before its first use; every launch leaves the tickets at zero, so
 * one buffer serves all later calls on the same stream.  One launch at a time per workspace. */
size_t mcav_warp_loss_workspace_bytes(int B, int H, int W);

/* tgt, ref0, ref1: [B,3,H,W].  disp_t, disp_r0: [B,1,H,W] sigmoid disparities of tgt and ref0 (scale 0).
 * poses: [B,2,6] (axis-angle, translation).  K: [B,3,3] fp64 or fp32 (flag).
 * upstream: 2 floats on the DEVICE = d(total)/d(loss_mam), d(total)/d(loss_smooth); NULL means (1,1).
 * term_weights: 3 floats on the HOST, weight of each warp's L1 mean inside loss_mam; NULL means
 *   (0.25, 0.25, 0.5) = the reference's single-scale combination (losses.py:227-240).
 * Outputs: losses[2] = (loss_mam, loss_smooth); d_disp_t, d_disp_r0: [B,1,H,W]; d_poses: [B,2,6]. */
int mcav_warp_loss_fwd_bwd(const float* tgt, const float* ref0, const float* ref1,
                           const float* disp_t, const float* disp_r0, const float* poses, const void* K,
                           int B, int H, int W, unsigned flags, const float* upstream, const float* term_weights,
                           float* losses, float* d_disp_t, float* d_disp_r0, float* d_poses,
                           void* workspace, size_t workspace_bytes, void* stream);

/* DIAGNOSTIC twin of mcav_warp_loss_fwd_bwd (upstream (1,1)): the same kernel bodies with a per-pixel dump, used by
 * tests/flip_finder.py to NAME the pixels at which an fp32 evaluation takes another bilinear cell / L1 sign than float64 does
 * (losses.py:183-240 is only piecewise smooth).  taps: [B][3 warps][7][H][W] = { ix, iy (un-normalised sampling position, source pixels),
 * d loss_mam / d ix, d loss_mam / d iy, residual of channel 0..2 } per warp (0: ref0 -> tgt, 1: ref1 -> tgt, 2: tgt with depth(ref0) and the
 * inverted pose[0]; with MCAV_WL_SSIM the residual planes are left zero).  Not used by the training path. */
int mcav_warp_loss_debug_taps(const float* tgt, const float* ref0, const float* ref1,
                              const float* disp_t, const float* disp_r0, const float* poses, const void* K,
                              int B, int H, int W, unsigned flags, const float* term_weights,
                              float* losses, float* d_disp_t, float* d_disp_r0, float* d_poses,
                              void* workspace, size_t workspace_bytes, float* taps, size_t taps_floats, void* stream);

/* Standalone inverse_warp (pose_geometry.py:201-229): img [B,3,H,W], depth [B,H,W], pose [B,6], K [B,3,3]
 * -> out [B,3,H,W].  flags: MCAV_WL_K_F64.  pose_inv as in the reference. */
int mcav_inverse_warp_fwd(const float* img, const float* depth, const float* pose, const void* K,
                          int B, int H, int W, int pose_inv, unsigned flags, float* out,
                          void* workspace, size_t workspace_bytes, void* stream);
/* Its backward w.r.t. depth [B,H,W] and pose [B,6] given grad_out [B,3,H,W] (images are leaves on this path). */
int mcav_inverse_warp_bwd(const float* img, const float* depth, const float* pose, const void* K,
                          const float* grad_out, int B, int H, int W, int pose_inv, unsigned flags,
                          float* d_depth, float* d_pose, void* workspace, size_t workspace_bytes, void* stream);

/* Transform.reconstruct (transform.py:74-105): depth [B,H,W], K -> points [B,3,H,W]. */
int mcav_reconstruct(const float* depth, const void* K, int B, int H, int W, unsigned flags, float* points,
                     void* workspace, size_t workspace_bytes, void* stream);
/* Transform.project (transform.py:114-150): points [B,3,H,W], K, Tcw [B,4,4] fp32 -> grid [B,H,W,2]. */
int mcav_project(const float* points, const void* K, const float* Tcw, int B, int H, int W, unsigned flags,
                 float* grid, void* stream);

/* transformation_from_parameters / rot_from_axisangle / get_translation_matrix (pose_geometry.py:124-199):
 * pose [B,6] = (axis-angle, translation) -> T [B,4,4] = Trans @ Rot, or its rigid inverse. */
int mcav_pose_to_matrix(const float* pose, int B, int invert, float* T, void* stream);
/* invert_pose (pose_geometry.py:110-115): T [B,4,4] rigid -> [R^T | -R^T t]. */
int mcav_invert_pose(const float* T, int B, float* Tinv, void* stream);

/* disp_to_depth (pose_geometry.py:81-82) and its backward: n elements. */
int mcav_disp_to_depth(const float* disp, float* depth, size_t n, void* stream);
int mcav_disp_to_depth_bwd(const float* disp, const float* d_depth, float* d_disp, size_t n, void* stream);

/* SSIM.standard_loss (losses.py:12-54): x, y [N,H,W] planes (N = B*C) -> clamp((1-SSIM)/2, 0, 1). */
int mcav_ssim_fwd(const float* x, const float* y, int N, int H, int W, float C1, float C2, float* out, void* stream);

/* Losses.smooth_loss for ONE scale (losses.py:242-260): depth [B,1,H,W]; loss_accum[0] += weight * term,
 * d_depth (+)= upstream[0] * weight * d term.  accumulate != 0 adds into d_depth. */
size_t mcav_smooth_workspace_bytes(int B, int H, int W);
int mcav_smooth_loss_fwd_bwd(const float* depth, int B, int H, int W, float weight, const float* upstream,
                             float* loss_accum, float* d_depth, int accumulate,
                             void* workspace, size_t workspace_bytes, void* stream);

/* ---- after the training step ------------------------------------------------------------------ */

/* Depth metrics, reference evaluate.py:6-39 (compute_errors): one pass over the ground-truth depth and the network's sigmoid
 * disparity (depth = 1 / (10 disp + 0.01), pose_geometry.py:82-83), float32 per element as the reference, float64 sums.
 * out10 (device) = { silog, abs_rel, log10, rms, sq_rel, log_rms, d1, d2, d3, number of elements taken }.  sq_rel is the real
 * mean((gt-pred)^2 / gt); the reference returns rms under that key (evaluate.py:36).  Elements with gt <= min_gt are skipped
 * (KITTI ground truth is sparse); min_gt < 0 takes every element, as the reference does. */
size_t mcav_depth_metrics_workspace_bytes(void);
int mcav_depth_metrics(const float* gt, const float* disp, size_t n, float min_gt, float* out10, void* workspace, size_t workspace_bytes,
                       void* stream);

/* Depth image -> pseudo-LiDAR cloud, reference pseudo-lidar/utils/PseudoLiDAR.py:69-110 (project_PL) with :39-46
 * (inverse_rigid_trans): un-project with P_rect_02, transform into the velodyne frame, keep x >= 0 and z < 1 m, keep every
 * sparsity-th survivor (0 = all), in pixel order; float64 like the reference's numpy arithmetic; 4th column 0 as in the reference.
 * depth [rows, cols] float32 on the device; T_velo_to_cam (4x4) and P_rect (3x4) are HOST pointers, row-major doubles.
 * cloud: device [capacity_points][4] doubles (rows * cols is always enough).  count_out_dev (device unsigned) receives the number of
 * valid points BEFORE sparsification: the cloud has ceil(count / max(sparsity, 1)) rows. */
size_t mcav_pseudo_lidar_workspace_bytes(int rows, int cols);
int mcav_pseudo_lidar_project(const float* depth, int rows, int cols, const double* T_velo_to_cam, const double* P_rect, int sparsity,
                              double* cloud, size_t capacity_points, unsigned* count_out_dev, void* workspace, size_t workspace_bytes,
                              void* stream);

/* ---- before the training step ------------------------------------------------------------------------ */

/* The reference's image transform chain (dataloaders.py:32-49 load_img, trainer.py:97-103): decoded uint8 RGB -> /255 -> ToTensor ->
 * ToPILImage -> Resize((h, w)) [Pillow bilinear, antialiased, 8-bit fixed point, horizontal then vertical] -> ToTensor -> Normalize,
 * for a batch of equally sized images, bit-exact against Pillow's resize.
 * mcav_resample_coeffs (HOST): Pillow's coefficient tables for one axis; call it with kk = NULL to get the capacity (out_size * ksize)
 * and *ksize_out, then again with host arrays bounds[out_size * 2], kk[capacity]; copy both to the device for mcav_image_preprocess.
 * An axis whose size does not change still takes its (identity) table. */
int mcav_resample_coeffs(int in_size, int out_size, int* ksize_out, int* bounds, int* kk, int kk_capacity);
size_t mcav_image_preprocess_workspace_bytes(int B, int H0, int w);
int mcav_image_preprocess(const unsigned char* src_bhwc, int B, int H0, int W0, int h, int w, const int* hbounds, const int* hkk, int hksize,
                          const int* vbounds, const int* vkk, int vksize, const float* mean3, const float* std3, float* dst_bchw,
                          void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MCAV_DEPTH_H */
