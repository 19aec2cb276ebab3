"""evaluate.py -- depth metrics as the reference defines them (evaluate.py:6-39), on device tensors.

Host-side metric code (not a hot-path kernel).  Fixes of the reference's two bugs are explicit:
`pred` is a disparity tensor (not a nested list) and 'sq_rel' reports sq_rel (the reference returns rms there).
"""
import numpy as np


def compute_errors(gt, pred_disp):
    gt = gt.detach().cpu().numpy().astype(np.float64)
    pred = (1.0 / (10.0 * pred_disp.detach().cpu().numpy().astype(np.float64) + 0.01))
    thresh = np.maximum(gt / pred, pred / gt)
    out = {"d1": (thresh < 1.25).mean(), "d2": (thresh < 1.25 ** 2).mean(), "d3": (thresh < 1.25 ** 3).mean()}
    out["rms"] = np.sqrt(((gt - pred) ** 2).mean())
    out["log_rms"] = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    out["abs_rel"] = np.mean(np.abs(gt - pred) / gt)
    out["sq_rel"] = np.mean(((gt - pred) ** 2) / gt)
    err = np.log(pred) - np.log(gt)
    out["silog"] = np.sqrt(np.mean(err ** 2) - np.mean(err) ** 2) * 100
    out["log10"] = np.mean(np.abs(np.log10(pred) - np.log10(gt)))
    return out
