"""evaluate.py -- depth metrics as the reference defines them (evaluate.py:6-39), computed on the GPU.

`compute_errors(gt, pred)` keeps the reference's call shape: `gt` a depth tensor, `pred` the depth network's output (a list whose
first entry is the full-resolution sigmoid disparity, or that tensor itself).  One fused pass (mcav_depth_metrics) instead of ~25
numpy temporaries; only the ten result floats cross PCIe.  Differences from the reference, on purpose: it works (the reference's
`disp_to_depth(pred[0]).cpu()` raises on the nested list it gets, evaluate.py:11-12) and 'sq_rel' is the squared-relative error
(the reference stores rms under that key, evaluate.py:36).  `min_gt`: ground-truth values <= min_gt are skipped (sparse KITTI
ground truth); the default -1 takes every element, as the reference does.
"""
import torch

from mcav import lib as L

L.register({
    "mcav_depth_metrics_workspace_bytes": (L.c_sz, []),
    "mcav_depth_metrics": (L.c_i, [L.c_p, L.c_p, L.c_sz, L.c_f, L.c_p, L.c_p, L.c_sz, L.c_p]),
})

KEYS = ("silog", "abs_rel", "log10", "rms", "sq_rel", "log_rms", "d1", "d2", "d3")


def compute_errors(gt, pred, min_gt=-1.0):
    disp = pred[0] if isinstance(pred, (list, tuple)) else pred
    gt = L.dev(gt.detach().to(torch.float32).contiguous(), "gt")
    disp = L.dev(disp.detach().to(torch.float32).contiguous(), "pred")
    if gt.numel() != disp.numel():
        raise L.MCAVError("compute_errors: gt has %d elements, the prediction %d" % (gt.numel(), disp.numel()))
    h = L.lib()
    ws = L.workspace(h.mcav_depth_metrics_workspace_bytes(), gt.device, "metrics")
    out = torch.empty(10, dtype=torch.float32, device=gt.device)
    L.check(h.mcav_depth_metrics(L.ptr(gt), L.ptr(disp), gt.numel(), float(min_gt), L.ptr(out), L.ptr(ws), ws.numel(), L.stream()),
            "mcav_depth_metrics")
    vals = out.cpu().tolist()
    acc = dict(zip(KEYS, vals[:9]))
    acc["count"] = int(vals[9])
    return acc
