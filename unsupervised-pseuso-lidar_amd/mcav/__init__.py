"""mcav: host-side runtime of the MI355X-native depth+pose training path.

Thin Python over the C ABI of libmcav_depth.so (include/mcav_depth.h).  PyTorch is used for device memory,
streams, autograd linkage and torch.distributed only; every arithmetic op of the hot path is a HIP kernel in
csrc/.  There is no CPU fallback: importing is cheap, but any op raises if the library is not built or a
tensor is not on the GPU.
"""
from . import lib  # noqa: F401
from .lib import MCAVError  # noqa: F401
