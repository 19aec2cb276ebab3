"""CPU restatement of the reference depth metrics -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows evaluate.py:6-39 line by line in numpy, in the dtype numpy would use on float32 tensors.  The reference function is
broken as written (`disp_to_depth(pred[0])` returns a nested list, `.cpu()` on it raises: evaluate.py:11-12) and reports rms
under the key 'sq_rel' (evaluate.py:36); the golden vectors (tests/golden/metrics.npz) come from the reference function with a
harness-side shim that hands it a tensor through its own disp_to_depth, so the arithmetic below is pinned, quirk included.
"""
import numpy as np


def compute_errors(gt, pred_disp):
    """gt: depth array; pred_disp: sigmoid disparity array.  Returns the reference's dict (same keys, same quirk) + 'sq_rel_fixed'."""
    gt = np.asarray(gt)
    pred = 1 / (10 * np.asarray(pred_disp) + 0.01)          # pose_geometry.py:82-83
    thresh = np.maximum((gt / pred), (pred / gt))           # evaluate.py:15
    d1 = (thresh < 1.25).mean()
    d2 = (thresh < 1.25 ** 2).mean()
    d3 = (thresh < 1.25 ** 3).mean()
    rms = np.sqrt(((gt - pred) ** 2).mean())                # :20-21
    log_rms = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())      # :23-24
    abs_rel = np.mean(np.abs(gt - pred) / gt)               # :26
    sq_rel = np.mean(((gt - pred) ** 2) / gt)               # :27
    err = np.log(pred) - np.log(gt)                         # :29-30
    silog = np.sqrt(np.mean(err ** 2) - np.mean(err) ** 2) * 100
    log10 = np.mean(np.abs(np.log10(pred) - np.log10(gt)))  # :32-33
    return {"silog": silog, "abs_rel": abs_rel, "log10": log10, "rms": rms, "sq_rel": rms, "log_rms": log_rms,
            "d1": d1, "d2": d2, "d3": d3, "sq_rel_fixed": sq_rel}
