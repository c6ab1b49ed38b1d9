"""Drop-in for the reference's ``vunet.layers`` (parameter holders + the two rearrangement ops).

The classes keep the reference's names, constructor arguments and ``state_dict`` key nesting
(vunet/layers.py:21-170) but carry no arithmetic: ``Vunet_fix_res`` (models.py) issues the fused
libfusg launches.  ``DepthToSpace`` / ``SpaceToDepth`` are callable and run the DCR-order kernels.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..nn_base import ConvP, WNConvP


class Activation(nn.Module):
    """ELU marker (reference layers.py:6-15); fused into the consumer conv's tile staging."""

    def __init__(self, activation):
        super().__init__()
        self.activation = activation


class MyConv2d(nn.Module):
    def __init__(self, c_in, c_out, kernel_size, stride, padding, w_norm: bool):
        super().__init__()
        self.c_in, self.c_out, self.kernel_size, self.stride, self.padding, self.w_norm = \
            c_in, c_out, kernel_size, stride, padding, w_norm
        self.conv = WNConvP(c_in, c_out, kernel_size) if w_norm else ConvP(c_in, c_out, kernel_size)


class NiN(nn.Module):
    def __init__(self, c_in, c_out, w_norm):
        super().__init__()
        self.layers = nn.Sequential(Activation("elu"), MyConv2d(c_in, c_out, 1, 1, 0, w_norm))


class Residual(nn.Module):
    def __init__(self, c_in, c_out, activation, drop_prob, w_norm, use_sampling=False):
        super().__init__()
        self.c_in, self.c_out = c_in, c_out
        self.layers = nn.Sequential(Activation("elu"), nn.Identity(), MyConv2d(c_in, c_out, 3, 1, 1, w_norm))


class DownSample(nn.Module):
    def __init__(self, c_in, c_out, w_norm):
        super().__init__()
        self.down = MyConv2d(c_in, c_out, 3, 2, 1, w_norm)


class UpSample(nn.Module):
    def __init__(self, c_in, c_out, w_norm, mode):
        super().__init__()
        if mode != "subpixel":
            raise NotImplementedError(f"UpSample(mode={mode!r}): the reference runs 'subpixel' only (run_test.py:82)")
        self.mode = mode
        self.depth4x = MyConv2d(c_in, 4 * c_out, 3, 1, 1, w_norm)


class Sampler(nn.Module):
    def __init__(self, c_in, c_out, w_norm):
        super().__init__()
        self.conv = MyConv2d(c_in, c_out, 3, 1, 1, w_norm)


class DepthToSpace(nn.Module):
    """DCR order: out[b,c,2h+i,2w+j] = in[b,(2i+j)C+c,h,w] (reference layers.py:173-196)."""

    def __init__(self, block_size):
        super().__init__()
        if block_size != 2:
            raise NotImplementedError("block_size 2 only")

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.depth_to_space2(ops.as_nhwc(x))


class SpaceToDepth(nn.Module):
    def __init__(self, block_size):
        super().__init__()
        if block_size != 2:
            raise NotImplementedError("block_size 2 only")

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.space_to_depth2(ops.as_nhwc(x))
