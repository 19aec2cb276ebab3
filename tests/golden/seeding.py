"""Deterministic, construction-order-independent weight initialisation shared by make_golden.py and the tests.

Every floating-point entry of a state_dict is refilled from a generator seeded by (seed, crc32(key)), so two
implementations with equal state_dict keys/shapes get identical weights whatever order they build layers in.
Biases, BatchNorm affine parameters and running statistics get non-trivial values on purpose.
"""
import math
import zlib

import torch


def reinit_by_name(model, seed):
    sd = model.state_dict()
    with torch.no_grad():
        for key in sorted(sd):
            t = sd[key]
            if not t.dtype.is_floating_point:
                continue
            g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(key.encode())) % (2 ** 31))
            leaf = key.rsplit(".", 1)[-1]
            if leaf == "running_var":
                v = 0.5 + torch.rand(t.shape, generator=g)
            elif leaf == "running_mean":
                v = 0.1 * torch.randn(t.shape, generator=g)
            elif t.dim() == 1 and leaf == "weight":          # BatchNorm gamma
                v = 1.0 + 0.1 * torch.randn(t.shape, generator=g)
            elif t.dim() == 1:                                # biases / BatchNorm beta
                v = 0.05 * torch.randn(t.shape, generator=g)
            else:
                fan_in = t[0].numel()
                v = torch.randn(t.shape, generator=g) * math.sqrt(2.0 / fan_in)
            t.copy_(v)
    return model
