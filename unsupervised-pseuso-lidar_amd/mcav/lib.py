"""ctypes binding of libmcav_depth.so (the drop-in boundary, include/mcav_depth.h)."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MCAV_LIB_PATH") or os.path.join(_HERE, "libmcav_depth.so")      # (override: kernel-diagnostic builds, csrc/Makefile `diag`)
_LIB = None

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_u = ctypes.c_uint
c_f = ctypes.c_float
c_sz = ctypes.c_size_t

WL_K_F64, WL_SKIP_IF_UNIT, WL_NO_SMOOTH, WL_INPUT_DEPTH, WL_SSIM = 1, 2, 4, 8, 16


class MCAVError(RuntimeError):
    pass


_ERR = {-1: "MCAV_E_INVALID (bad argument)", -2: "MCAV_E_WORKSPACE (workspace too small)", -3: "MCAV_E_LAUNCH (HIP launch error)"}

_SIGNATURES = {
    "mcav_abi_version": (c_i, []),
    "mcav_warp_loss_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "mcav_warp_loss_fwd_bwd": (c_i, [c_p] * 7 + [c_i, c_i, c_i, c_u, c_p, c_p] + [c_p] * 4 + [c_p, c_sz, c_p]),
    "mcav_warp_loss_debug_taps": (c_i, [c_p] * 7 + [c_i, c_i, c_i, c_u, c_p] + [c_p] * 4 + [c_p, c_sz, c_p, c_sz, c_p]),
    "mcav_inverse_warp_fwd": (c_i, [c_p] * 4 + [c_i, c_i, c_i, c_i, c_u, c_p, c_p, c_sz, c_p]),
    "mcav_inverse_warp_bwd": (c_i, [c_p] * 5 + [c_i, c_i, c_i, c_i, c_u, c_p, c_p, c_p, c_sz, c_p]),
    "mcav_reconstruct": (c_i, [c_p, c_p, c_i, c_i, c_i, c_u, c_p, c_p, c_sz, c_p]),
    "mcav_project": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_u, c_p, c_p]),
    "mcav_disp_to_depth": (c_i, [c_p, c_p, c_sz, c_p]),
    "mcav_disp_to_depth_bwd": (c_i, [c_p, c_p, c_p, c_sz, c_p]),
    "mcav_ssim_fwd": (c_i, [c_p, c_p, c_i, c_i, c_i, c_f, c_f, c_p, c_p]),
    "mcav_smooth_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "mcav_smooth_loss_fwd_bwd": (c_i, [c_p, c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_i, c_p, c_sz, c_p]),
}


def register(signatures):
    """Let other host modules (conv, norm, optimiser ...) declare the entry points they bind."""
    _SIGNATURES.update(signatures)
    if _LIB is not None:
        _declare(_LIB, signatures)


def _declare(handle, signatures):
    for name, (res, args) in signatures.items():
        fn = getattr(handle, name)      # AttributeError here = header and library out of sync
        fn.restype = res
        fn.argtypes = args


def lib():
    """The loaded library.  Fails loudly when it has not been built: the product has no other path."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise MCAVError("libmcav_depth.so is missing (%s). Build it: python -c 'import __graft_entry__ as g; g.build()' "
                            "or `make -C unsupervised-pseuso-lidar_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        _declare(handle, _SIGNATURES)
        _LIB = handle
    return _LIB


def check(rc, what):
    if rc != 0:
        raise MCAVError("%s failed: %s" % (what, _ERR.get(rc, rc)))


def dev(t, name="tensor", dtype=torch.float32):
    """Validate a tensor handed to the C ABI: on the GPU, contiguous, expected dtype."""
    if not t.is_cuda:
        raise MCAVError("%s must live on the GPU: the MI355X path has no CPU fallback" % name)
    if t.dtype != dtype:
        raise MCAVError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise MCAVError("%s must be contiguous" % name)
    return t


def ptr(t):
    return c_p(t.data_ptr()) if t is not None else c_p(0)


def stream():
    return c_p(torch.cuda.current_stream().cuda_stream)


_WS = {}
_WS_RETIRED = []


def workspace(nbytes, device, key="default", zero=False):
    """A cached byte workspace per (device, key, stream); grown on demand, reused across calls on that stream.
    zero: the buffer is zero-filled when it is (re)allocated -- the fused loss kernels keep their completion tickets in it and leave them at
    zero after every launch (include/mcav_depth.h: mcav_warp_loss_fwd_bwd)."""
    k = (str(device), key, torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0)   # one per stream: streams overlap
    buf = _WS.get(k)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _WS_RETIRED.append(buf)       # never freed: a captured hipGraph may replay launches that have this address baked in
        alloc = torch.zeros if zero else torch.empty
        buf = alloc(max(int(nbytes), 1), dtype=torch.uint8, device=device)
        _WS[k] = buf
    return buf
