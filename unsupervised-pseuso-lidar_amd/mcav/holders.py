"""Parameter holders: nn.Modules that only own parameters/buffers under the reference's names and shapes.

They have no forward(): the arithmetic is done by the HIP engines (mcav/depthnet.py, mcav/posenet.py).
Initialisation follows what the reference gets from torch / torchvision defaults.
"""
import math

import torch
import torch.nn as nn


class ConvParams(nn.Module):
    """weight [Cout, Cin, kh, kw] (+ bias), initialised like nn.Conv2d (kaiming_uniform(a=sqrt(5)), bias U(+-1/sqrt(fan_in)))."""

    def __init__(self, cin, cout, k, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1.0 / math.sqrt(cin * k * k)
            nn.init.uniform_(self.bias, -bound, bound)


class DeconvParams(nn.Module):
    """ConvTranspose2d parameters: weight [Cin, Cout, kh, kw] + bias (torch default init)."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cin, cout, k, k))
        self.bias = nn.Parameter(torch.empty(cout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(cout * k * k)       # torch computes fan_in from weight.size(1) * k * k
        nn.init.uniform_(self.bias, -bound, bound)


class LinearParams(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.empty(cout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(cin)
        nn.init.uniform_(self.bias, -bound, bound)


class BNParams(nn.Module):
    """nn.BatchNorm2d's parameters and buffers (eps 1e-5, momentum 0.1)."""

    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.eps = 1e-5
        self.momentum = 0.1


class Holder(nn.Module):
    """Generic named container."""

    def __init__(self, **children):
        super().__init__()
        for k, v in children.items():
            if v is not None:
                setattr(self, k, v)
            else:
                self.register_module(k, None)
