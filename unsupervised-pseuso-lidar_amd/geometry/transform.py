"""geometry/transform.py -- drop-in for the reference's Transform (geometry/transform.py:12-150) on MI355X.

reconstruct / project run as HIP kernels (csrc/warp_loss.hip); unlike the reference they accept any batch size
(its k_hom hard-codes repeat(4), transform.py:110).  Forward only: the differentiable path is
geometry.pose_geometry.inverse_warp / losses.Losses, which fuse these steps.
"""
import torch

from mcav import lib as L


class Transform:
    def reconstruct(self, depth, K):
        """depth [B,H,W] (or [B,1,H,W]), K [B,3,3] -> camera points [B,3,H,W]."""
        if depth.dim() == 4:
            depth = depth[:, 0]
        depth = L.dev(depth.contiguous(), "depth")
        K = L.dev(K.contiguous(), "intrinsics", K.dtype)
        B, H, W = depth.shape
        h = L.lib()
        ws = L.workspace(h.mcav_warp_loss_workspace_bytes(B, H, W), depth.device, "transform")
        out = torch.empty(B, 3, H, W, dtype=torch.float32, device=depth.device)
        flags = L.WL_K_F64 if K.dtype == torch.float64 else 0
        L.check(h.mcav_reconstruct(L.ptr(depth), L.ptr(K), B, H, W, flags, L.ptr(out), L.ptr(ws), ws.numel(), L.stream()), "mcav_reconstruct")
        return out

    def project(self, X, K, Tcw):
        """X [B,3,H,W], K [B,3,3], Tcw [B,4,4] -> sampling grid [B,H,W,2] in [-1, 1]."""
        X = L.dev(X.contiguous(), "X")
        K = L.dev(K.contiguous(), "intrinsics", K.dtype)
        Tcw = L.dev(Tcw.contiguous(), "Tcw")
        B, _, H, W = X.shape
        out = torch.empty(B, H, W, 2, dtype=torch.float32, device=X.device)
        flags = L.WL_K_F64 if K.dtype == torch.float64 else 0
        L.check(L.lib().mcav_project(L.ptr(X), L.ptr(K), L.ptr(Tcw), B, H, W, flags, L.ptr(out), L.stream()), "mcav_project")
        return out
