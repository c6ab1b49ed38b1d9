"""Oracle: stacked-hourglass keypoint network (reference stacked_hourglass/models.py)."""
from __future__ import annotations

from typing import Dict, List, Mapping

import numpy as np
import torch
import torch.nn.functional as F

SD = Mapping[str, torch.Tensor]


def _bn(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    # nn.BatchNorm2d in eval mode (stacked_hourglass/models.py:11,13,16,99,137)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.1, 1e-5)


def _conv(sd: SD, p: str, x: torch.Tensor, stride: int = 1, padding: int = 0) -> torch.Tensor:
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def _bottleneck(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """Pre-activation bottleneck, stacked_hourglass/models.py:22-42."""
    residual = x
    out = _conv(sd, p + ".conv1", F.relu(_bn(sd, p + ".bn1", x)))
    out = _conv(sd, p + ".conv2", F.relu(_bn(sd, p + ".bn2", out)), padding=1)
    out = _conv(sd, p + ".conv3", F.relu(_bn(sd, p + ".bn3", out)))
    if (p + ".downsample.0.weight") in sd:                       # models.py:37-38, 126-127
        residual = _conv(sd, p + ".downsample.0", x)
    out += residual                                               # models.py:40
    return out


def _residual_seq(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """nn.Sequential of Bottlenecks (models.py:53-57,123-134)."""
    i = 0
    while (p + f".{i}.conv1.weight") in sd:
        x = _bottleneck(sd, p + f".{i}", x)
        i += 1
    return x


def _hourglass(sd: SD, p: str, n: int, x: torch.Tensor) -> torch.Tensor:
    """Recursive hourglass, stacked_hourglass/models.py:70-83 (depth index n-1)."""
    up1 = _residual_seq(sd, f"{p}.hg.{n - 1}.0", x)
    low1 = F.max_pool2d(x, 2, stride=2)
    low1 = _residual_seq(sd, f"{p}.hg.{n - 1}.1", low1)
    if n > 1:
        low2 = _hourglass(sd, p, n - 1, low1)
    else:
        low2 = _residual_seq(sd, f"{p}.hg.{n - 1}.3", low1)
    low3 = _residual_seq(sd, f"{p}.hg.{n - 1}.2", low2)
    up2 = F.interpolate(low3, scale_factor=2)                     # nn.Upsample(scale_factor=2), nearest
    return up1 + up2


def hourglass_forward(sd: SD, x: torch.Tensor, num_stacks: int = 2, depth: int = 4
                      ) -> Dict[str, List[torch.Tensor]]:
    """HourglassNet.forward, stacked_hourglass/models.py:141-167."""
    heatmaps = []
    x = F.relu(_bn(sd, "bn1", _conv(sd, "conv1", x, stride=2, padding=3)))
    x = _residual_seq(sd, "layer1", x)
    x = F.max_pool2d(x, 2, stride=2)
    x = _residual_seq(sd, "layer2", x)
    x = _residual_seq(sd, "layer3", x)
    for i in range(num_stacks):
        y = _hourglass(sd, f"hg.{i}", depth, x)
        y = _residual_seq(sd, f"res.{i}", y)
        y = F.relu(_bn(sd, f"fc.{i}.1", _conv(sd, f"fc.{i}.0", y)))       # models.py:136-139
        score = _conv(sd, f"score.{i}", y)
        heatmaps.append(score)
        if i < num_stacks - 1:
            fc_ = _conv(sd, f"fc_.{i}", y)
            score_ = _conv(sd, f"score_.{i}", score)
            x = x + fc_ + score_                                           # models.py:163
    return {"heatmaps": heatmaps}


def heatmap_argmax(heat: torch.Tensor) -> np.ndarray:
    """Row-major first-occurrence argmax per (b, c): int64 [B, C] flat indices y*W + x.

    Integer contract behind get_maxima (utils/keypoint_utils.py:85-88)."""
    h = heat.detach().to("cpu").numpy()
    b, c = h.shape[:2]
    return h.reshape(b, c, -1).argmax(axis=2).astype(np.int64)


def get_maxima(heat: torch.Tensor) -> np.ndarray:
    """utils/keypoint_utils.py:66-92: (x / w, y / h) float64 of the per-channel argmax.

    The caller first nearest-upsamples 64->256 (trajectory_inference.py:76-79); nearest
    upsampling by an integer factor k maps the first-occurrence argmax (y0, x0) to (k*y0, k*x0),
    so the result equals (x0 / w0, y0 / h0) of the un-upsampled map."""
    b, c, hh, ww = heat.shape
    idx = heatmap_argmax(heat)
    out = np.zeros((b, c, 2))
    out[..., 0] = (idx % ww) / ww
    out[..., 1] = (idx // ww) / hh
    return out
