"""Oracle (test infrastructure): photometric / smoothness losses on CPU tensors.

Restates reference losses.py:
  * SSIM.standard_loss                 losses.py:12-54
  * Losses.reprojection_loss (L1 only, 3 warps per triplet incl. the tgt->refs[1] warp that
    uses depth(ref0) with the inverted pose[0])                          losses.py:183-240
  * Losses.smooth_loss (second-order differences, weight /2.3 per scale)  losses.py:242-260
  * Losses.forward                                                       losses.py:262-271
plus the north_star's SSIM+L1 photometric mix (0.85/0.15, losses.py:77) as an opt-in extension
(`ssim_weight`), which the reference cannot execute (self.SSIM is commented out, losses.py:59).
"""
import torch
import torch.nn.functional as F

from .geometry import disp_to_depth, inverse_warp


def ssim_distance(x, y, C1=1e-4, C2=9e-4):
    """clamp((1 - SSIM) / 2, 0, 1) with a 3x3 box filter over reflection-padded inputs."""
    xp = F.pad(x, (1, 1, 1, 1), mode="reflect")
    yp = F.pad(y, (1, 1, 1, 1), mode="reflect")
    mu_x = F.avg_pool2d(xp, 3, 1)
    mu_y = F.avg_pool2d(yp, 3, 1)
    mu_xy = mu_x * mu_y
    mu_xx = mu_x.pow(2)
    mu_yy = mu_y.pow(2)
    sig_x = F.avg_pool2d(xp.pow(2), 3, 1) - mu_xx
    sig_y = F.avg_pool2d(yp.pow(2), 3, 1) - mu_yy
    sig_xy = F.avg_pool2d(xp * yp, 3, 1) - mu_xy
    num = (2 * mu_xy + C1) * (2 * sig_xy + C2)
    den = (mu_xx + mu_yy + C1) * (sig_x + sig_y + C2)
    return torch.clamp((1.0 - num / den) / 2.0, 0.0, 1.0)


def photometric(pred, target, ssim_weight=0.0):
    """Scalar photometric distance.  ssim_weight == 0 -> nn.L1Loss() mean (the live reference path)."""
    l1 = (pred - target).abs()
    if ssim_weight == 0.0:
        return l1.mean()
    return (ssim_weight * ssim_distance(pred, target) + (1.0 - ssim_weight) * l1).mean()


def warp_plan(tgt, refs, depths, poses):
    """The three (source, target, depth-list, pose, invert) tuples the reference evaluates per triplet."""
    p0, p1 = poses[:, 0, :], poses[:, 1, :]
    return [
        # indx 0: both references warped into the target view with depth(tgt)
        dict(group=0, src=refs[0], target=tgt, depth=depths[0], pose=p0, inv=False),
        dict(group=0, src=refs[1], target=tgt, depth=depths[0], pose=p1, inv=False),
        # indx 1: tgt warped "into" refs[1] with depth(ref0) and the inverse of pose[0]
        # (frame mismatch is the reference's own behaviour, losses.py:203-207)
        dict(group=1, src=tgt, target=refs[1], depth=depths[1], pose=p0, inv=True),
    ]


def reprojection_loss(tgt, refs, depths, poses, K, ssim_weight=0.0):
    plan = warp_plan(tgt, refs, depths, poses)
    terms = []
    for group in (0, 1):
        members = [w for w in plan if w["group"] == group]
        H, W = members[0]["depth"][0].shape[-2:]
        for s in range(len(members[0]["depth"])):
            per_warp = []
            for w in members:
                D = w["depth"][s]
                if D.shape[-1] != W:
                    D = F.interpolate(D, [H, W], mode="bilinear", align_corners=False)
                D = D[:, 0]
                proj = inverse_warp(w["src"], D, w["pose"], K, w["inv"])
                per_warp.append(photometric(proj, w["target"], ssim_weight))
            terms.append(torch.mean(torch.stack(per_warp)))
    return sum(terms) / len(terms)


def smooth_loss(depth_scales):
    if not isinstance(depth_scales, (list, tuple)):
        depth_scales = [depth_scales]
    total = 0
    weight = 1.0
    for D in depth_scales:
        dy = D[:, :, 1:] - D[:, :, :-1]
        dx = D[:, :, :, 1:] - D[:, :, :, :-1]
        dx2 = dx[:, :, :, 1:] - dx[:, :, :, :-1]
        dxdy = dx[:, :, 1:] - dx[:, :, :-1]
        dydx = dy[:, :, :, 1:] - dy[:, :, :, :-1]
        dy2 = dy[:, :, 1:] - dy[:, :, :-1]
        total = total + (dx2.abs().mean() + dxdy.abs().mean() + dydx.abs().mean() + dy2.abs().mean()) * weight
        weight /= 2.3
    return total


def losses_forward(tgt, refs, disparity, poses, K, ssim_weight=0.0):
    """-> [loss_mam, loss_smooth]; disparity = [disps(tgt), disps(ref0)], each a list over scales."""
    depths = disp_to_depth(disparity)
    return [reprojection_loss(tgt, refs, depths, poses, K, ssim_weight), smooth_loss(depths[0])]
