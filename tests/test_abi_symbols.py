"""CPU: the C-ABI library loads and exports every symbol include/*.h declares (no compute calls)."""
import ctypes
import glob
import os
import re

import pytest

from conftest import PKG, REPO

LIB = os.path.join(PKG, "mcav", "libmcav_depth.so")


def declared_symbols():
    names = []
    for h in glob.glob(os.path.join(REPO, "include", "*.h")):
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(mcav_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(LIB):
        import __graft_entry__ as g
        g.build()
    handle = ctypes.CDLL(LIB)
    names = declared_symbols()
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(handle, n)]
    assert not missing, missing
    handle.mcav_abi_version.restype = ctypes.c_int
    assert handle.mcav_abi_version() >= 1


def test_python_binding_covers_header():
    """Every declared entry point has a ctypes signature registered by the host package."""
    import mcav.lib as L
    import losses  # noqa: F401
    import geometry.pose_geometry  # noqa: F401
    import geometry.transform  # noqa: F401
    import mcav.nn  # noqa: F401
    import mcav.tape  # noqa: F401
    import evaluate  # noqa: F401
    import pseudo_lidar  # noqa: F401
    import dataloaders  # noqa: F401
    missing = [n for n in declared_symbols() if n not in L._SIGNATURES]
    assert not missing, missing


def test_product_refuses_cpu_tensors():
    import torch
    import mcav.lib as L
    from losses import Losses
    B, H, W = 2, 8, 16
    t = torch.zeros(B, 3, H, W)
    with pytest.raises(L.MCAVError):
        Losses().forward(t, [t, t], [[torch.zeros(B, 1, H, W)], [torch.zeros(B, 1, H, W)]], torch.zeros(B, 2, 6),
                         torch.eye(3).repeat(B, 1, 1), None)
