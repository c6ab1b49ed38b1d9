"""Oracle: Warp&Learn image-completion network G_Resnet (reference warp_learn/models.py:15-208)."""
from __future__ import annotations

from typing import Mapping

import torch
import torch.nn.functional as F

SD = Mapping[str, torch.Tensor]


def _layer_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """Custom LayerNorm, warp_learn/models.py:26-35: per-sample mean and UNBIASED std over C*H*W,
    eps added to the std (not the variance), then per-channel gamma/beta."""
    shape = [-1] + [1] * (x.dim() - 1)
    mean = x.view(x.size(0), -1).mean(1).view(*shape)
    std = x.view(x.size(0), -1).std(1).view(*shape)
    x = (x - mean) / (std + eps)
    shape = [1, -1] + [1] * (x.dim() - 2)
    return x * gamma.view(*shape) + beta.view(*shape)


def _block(sd: SD, p: str, x: torch.Tensor, pad: int, stride: int, norm: str, act: str) -> torch.Tensor:
    """Conv2dBlock.forward, warp_learn/models.py:84-90 (pad_type is always 'reflect' for G_Resnet,
    models.py:195; the conv itself has padding 0, models.py:81-82)."""
    x = F.conv2d(F.pad(x, (pad, pad, pad, pad), mode="reflect"),
                 sd[p + ".conv.weight"], sd[p + ".conv.bias"], stride=stride)
    if norm == "inst":
        x = F.instance_norm(x, use_input_stats=True, eps=1e-5)           # models.py:56
    elif norm == "ln":
        x = _layer_norm(x, sd[p + ".norm.gamma"], sd[p + ".norm.beta"])  # models.py:58
    if act == "relu":
        x = F.relu(x)
    elif act == "tanh":
        x = torch.tanh(x)
    return x


def _resblocks(sd: SD, p: str, x: torch.Tensor, n_res: int) -> torch.Tensor:
    """ResBlocks / ResBlock, warp_learn/models.py:93-124."""
    for j in range(n_res):
        q = f"{p}.model.{j}.model"
        out = _block(sd, q + ".0", x, 1, 1, "inst", "relu")
        out = _block(sd, q + ".1", out, 1, 1, "inst", "none")
        out += x
        x = out
    return x


def icn_forward(sd: SD, x: torch.Tensor, num_downs: int = 2, n_res: int = 3) -> torch.Tensor:
    """G_Resnet.forward, warp_learn/models.py:205-208 = Decoder(ContentEncoder(x))."""
    # ContentEncoder, models.py:127-148
    p = "enc_content.model"
    x = _block(sd, f"{p}.0", x, 3, 1, "inst", "relu")
    for i in range(num_downs):
        x = _block(sd, f"{p}.{1 + i}", x, 1, 2, "inst", "relu")
    x = _resblocks(sd, f"{p}.{1 + num_downs}", x, n_res)
    # Decoder, models.py:162-187 (model.0 ResBlocks; then [Upsample, Conv2dBlock] pairs; then 7x7 tanh)
    p = "dec.model"
    x = _resblocks(sd, f"{p}.0", x, n_res)
    for i in range(num_downs):
        x = F.interpolate(x, scale_factor=2, mode="nearest")             # models.py:157-159
        x = _block(sd, f"{p}.{2 + 2 * i}", x, 2, 1, "ln", "relu")
    x = _block(sd, f"{p}.{1 + 2 * num_downs}", x, 3, 1, "none", "tanh")
    return x
