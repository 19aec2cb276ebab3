"""geometry/pose_geometry.py -- drop-in for the reference module of the same path, on MI355X.

Call surface kept (reference geometry/pose_geometry.py): disp_to_depth (:70-95), inverse_warp (:201-229),
transformation_from_parameters (:124-141), rot_from_axisangle (:160-199), get_translation_matrix (:144-157),
invert_pose (:110-115).  All arithmetic is in csrc/warp_loss.hip / csrc/warp_math.h.
"""
import torch

from mcav import lib as L
from .transform import Transform  # noqa: F401  (the reference re-exports it the same way)

L.register({
    "mcav_pose_to_matrix": (L.c_i, [L.c_p, L.c_i, L.c_i, L.c_p, L.c_p]),
    "mcav_invert_pose": (L.c_i, [L.c_p, L.c_i, L.c_p, L.c_p]),
})


class _DispToDepthFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp):
        disp = L.dev(disp.contiguous(), "disparity")
        out = torch.empty_like(disp)
        L.check(L.lib().mcav_disp_to_depth(L.ptr(disp), L.ptr(out), disp.numel(), L.stream()), "mcav_disp_to_depth")
        ctx.save_for_backward(disp)
        return out

    @staticmethod
    def backward(ctx, g):
        (disp,) = ctx.saved_tensors
        g = L.dev(g.contiguous(), "grad")
        out = torch.empty_like(disp)
        L.check(L.lib().mcav_disp_to_depth_bwd(L.ptr(disp), L.ptr(g), L.ptr(out), disp.numel(), L.stream()), "mcav_disp_to_depth_bwd")
        return out


def disp_to_depth(disps):
    """Nested list [time][scale] of sigmoid disparities -> depths 1 / (10 d + 0.01)."""
    return [[_DispToDepthFn.apply(d) for d in per_time] for per_time in disps]


class _InverseWarpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, depth, pose, K, pose_inv):
        img = L.dev(img.contiguous(), "img")
        depth = L.dev(depth.contiguous(), "depth")
        pose = L.dev(pose.contiguous(), "pose")
        K = L.dev(K.contiguous(), "intrinsics", K.dtype)
        B, C, H, W = img.shape
        if C != 3 or depth.shape != (B, H, W) or pose.shape != (B, 6):
            raise L.MCAVError("inverse_warp: img [B,3,H,W], depth [B,H,W], pose [B,6] expected")
        flags = L.WL_K_F64 if K.dtype == torch.float64 else 0
        h = L.lib()
        ws = L.workspace(h.mcav_warp_loss_workspace_bytes(B, H, W), img.device, "inverse_warp")
        out = torch.empty_like(img)
        L.check(h.mcav_inverse_warp_fwd(L.ptr(img), L.ptr(depth), L.ptr(pose), L.ptr(K), B, H, W, int(bool(pose_inv)), flags, L.ptr(out),
                                        L.ptr(ws), ws.numel(), L.stream()), "mcav_inverse_warp_fwd")
        ctx.save_for_backward(img, depth, pose, K)
        ctx.meta = (int(bool(pose_inv)), flags)
        return out

    @staticmethod
    def backward(ctx, go):
        img, depth, pose, K = ctx.saved_tensors
        inv, flags = ctx.meta
        B, _, H, W = img.shape
        go = L.dev(go.contiguous(), "grad_out")
        h = L.lib()
        ws = L.workspace(h.mcav_warp_loss_workspace_bytes(B, H, W), img.device, "inverse_warp")
        gd = torch.empty_like(depth)
        gp = torch.empty_like(pose)
        L.check(h.mcav_inverse_warp_bwd(L.ptr(img), L.ptr(depth), L.ptr(pose), L.ptr(K), L.ptr(go), B, H, W, inv, flags, L.ptr(gd), L.ptr(gp),
                                        L.ptr(ws), ws.numel(), L.stream()), "mcav_inverse_warp_bwd")
        return None, gd, gp, None, None


def inverse_warp(img, depth, pose, K, pose_inv, rotation_mode='euler', padding_mode='zeros'):
    """Warp source image `img` [B,3,H,W] to the target view: depth [B,H,W] of the target, pose [B,6], K [B,3,3].

    Differentiable w.r.t. depth and pose (images are leaves on the training path, losses.py:183-240)."""
    if padding_mode != 'zeros':
        raise L.MCAVError("inverse_warp: only padding_mode='zeros' (the reference's only use) is implemented")
    if depth.dim() == 4:
        depth = depth[:, 0]
    return _InverseWarpFn.apply(img, depth, pose, K, pose_inv)


def _pose_matrix(pose6, invert):
    pose6 = L.dev(pose6.contiguous(), "pose")
    B = pose6.shape[0]
    out = torch.empty(B, 4, 4, dtype=torch.float32, device=pose6.device)
    L.check(L.lib().mcav_pose_to_matrix(L.ptr(pose6), B, int(bool(invert)), L.ptr(out), L.stream()), "mcav_pose_to_matrix")
    return out


def transformation_from_parameters(axisangle, translation, invert=False):
    """(axisangle [B,1,3], translation [B,1,3]) -> [B,4,4] = Trans @ Rot (or its rigid inverse).  Forward only."""
    B = axisangle.shape[0]
    return _pose_matrix(torch.cat([axisangle.reshape(B, 3), translation.reshape(B, 3)], 1), invert)


def rot_from_axisangle(vec):
    B = vec.shape[0]
    return _pose_matrix(torch.cat([vec.reshape(B, 3), torch.zeros(B, 3, dtype=vec.dtype, device=vec.device)], 1), False)


def get_translation_matrix(translation_vector):
    B = translation_vector.shape[0]
    t = translation_vector.reshape(B, 3)
    return _pose_matrix(torch.cat([torch.zeros_like(t), t], 1), False)


def invert_pose(T):
    T = L.dev(T.contiguous(), "T")
    out = torch.empty_like(T)
    L.check(L.lib().mcav_invert_pose(L.ptr(T), T.shape[0], L.ptr(out), L.stream()), "mcav_invert_pose")
    return out
