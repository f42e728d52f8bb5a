"""VGG frame front-ends with the reference's interface (scripts/CNNs.py).

    getVGG3LOutputDimension / getVGG4LOutputDimension      CNNs.py:7-20
    VGG3L(kernel_size), VGG4L(kernel_size)                 CNNs.py:22-91

Each block is [conv3x3 pad1 + ReLU] x2 followed by a 2x2 stride-2 ceil-mode max-pool on a
[B,1,T,F] map; the output is flattened to [B,T',C*F'] with feature index c*F'+f
(CNNs.py:49-50 / 88-89).  Parameter names (conv11 ... conv42) match the reference so its
checkpoints load with strict=True.

Round-1 status: the convolutions run through PyTorch-ROCm (MIOpen) in channels_last
memory format; hand-written MFMA implicit-GEMM kernels are the "next" row of SURVEY 8(f).
"""
import math

import torch
from torch import nn
from torch.nn import functional as F

__all__ = ["getVGG3LOutputDimension", "getVGG4LOutputDimension", "VGG3L", "VGG4L"]


def _pooled(n, times):
    for _ in range(times):
        n = int(math.ceil(n / 2))
    return n


def getVGG3LOutputDimension(inputDimension, outputChannel=128):
    return _pooled(int(inputDimension), 3) * outputChannel


def getVGG4LOutputDimension(inputDimension, outputChannel=128):
    return _pooled(int(inputDimension), 4) * outputChannel


class _VGG(nn.Module):
    """n_blocks VGG blocks; block i (1-based) has kernel_size / 2**(n_blocks-i) channels."""

    n_blocks = 0

    def __init__(self, kernel_size):
        super().__init__()
        cin = 1
        for blk in range(1, self.n_blocks + 1):
            cout = int(kernel_size / (2 ** (self.n_blocks - blk)))
            setattr(self, "conv%d1" % blk, nn.Conv2d(cin, cout, 3, stride=1, padding=1))
            setattr(self, "conv%d2" % blk, nn.Conv2d(cout, cout, 3, stride=1, padding=1))
            cin = cout
        self.channels_last = True

    def forward(self, paddedInputTensor):
        B, T, Fdim = paddedInputTensor.shape
        h = paddedInputTensor.reshape(B, 1, T, Fdim)
        if self.channels_last and h.is_cuda:
            h = h.contiguous(memory_format=torch.channels_last)
        for blk in range(1, self.n_blocks + 1):
            h = F.relu(getattr(self, "conv%d1" % blk)(h))
            h = F.relu(getattr(self, "conv%d2" % blk)(h))
            h = F.max_pool2d(h, 2, stride=2, ceil_mode=True)
        # [B,C,T',F'] -> [B,T',C*F'] with feature = c*F' + f
        return h.permute(0, 2, 1, 3).reshape(B, h.size(2), h.size(1) * h.size(3))


class VGG3L(_VGG):
    n_blocks = 3


class VGG4L(_VGG):
    n_blocks = 4
