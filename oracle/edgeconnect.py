"""Oracle: EdgeConnect generators and the inference half of their model wrappers
(reference edgeconnect/networks.py, edgeconnect/models.py:130-135,236-240)."""
from __future__ import annotations

from typing import Mapping

import torch
import torch.nn.functional as F

SD = Mapping[str, torch.Tensor]


def _weight(sd: SD, p: str, transposed: bool = False) -> torch.Tensor:
    """Plain weight, or eval-mode spectral norm (networks.py:206-210 -> nn.utils.spectral_norm):
    w = weight_orig / sigma, sigma = u . (W_mat v) with the STORED u, v (no power iteration in
    eval).  W_mat flattens around dim 0 for Conv2d and dim 1 for ConvTranspose2d."""
    if (p + ".weight") in sd:
        return sd[p + ".weight"]
    w = sd[p + ".weight_orig"]
    wm = w.permute(1, 0, 2, 3).reshape(w.shape[1], -1) if transposed else w.reshape(w.shape[0], -1)
    sigma = torch.dot(sd[p + ".weight_u"], torch.mv(wm, sd[p + ".weight_v"]))
    return w / sigma


def _bias(sd: SD, p: str):
    return sd.get(p + ".bias", None)


def _in(x):
    return F.instance_norm(x, use_input_stats=True, eps=1e-5)


def _generator_trunk(sd: SD, x: torch.Tensor, residual_blocks: int = 8) -> torch.Tensor:
    """Shared skeleton of InpaintGenerator (networks.py:41-74) and EdgeGenerator (networks.py:92-125)."""
    x = F.conv2d(F.pad(x, (3, 3, 3, 3), mode="reflect"), _weight(sd, "encoder.1"), _bias(sd, "encoder.1"))
    x = F.relu(_in(x))
    x = F.relu(_in(F.conv2d(x, _weight(sd, "encoder.4"), _bias(sd, "encoder.4"), stride=2, padding=1)))
    x = F.relu(_in(F.conv2d(x, _weight(sd, "encoder.7"), _bias(sd, "encoder.7"), stride=2, padding=1)))
    for i in range(residual_blocks):                              # ResnetBlock(256, dilation=2), networks.py:184-203
        p = f"middle.{i}.conv_block"
        y = F.conv2d(F.pad(x, (2, 2, 2, 2), mode="reflect"), _weight(sd, p + ".1"), _bias(sd, p + ".1"), dilation=2)
        y = F.relu(_in(y))
        y = F.conv2d(F.pad(y, (1, 1, 1, 1), mode="reflect"), _weight(sd, p + ".5"), _bias(sd, p + ".5"))
        x = x + _in(y)
    x = F.relu(_in(F.conv_transpose2d(x, _weight(sd, "decoder.0", True), _bias(sd, "decoder.0"), stride=2, padding=1)))
    x = F.relu(_in(F.conv_transpose2d(x, _weight(sd, "decoder.3", True), _bias(sd, "decoder.3"), stride=2, padding=1)))
    x = F.conv2d(F.pad(x, (3, 3, 3, 3), mode="reflect"), _weight(sd, "decoder.7"), _bias(sd, "decoder.7"))
    return x


def edge_generator_forward(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """EdgeGenerator.forward, networks.py:130-135."""
    return torch.sigmoid(_generator_trunk(sd, x))


def inpaint_generator_forward(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """InpaintGenerator.forward, networks.py:79-85."""
    return (torch.tanh(_generator_trunk(sd, x)) + 1) / 2


def edge_model_forward(sd: SD, images, edges, masks) -> torch.Tensor:
    """EdgeModel.forward, edgeconnect/models.py:130-135 (``sd`` = the generator's state_dict)."""
    edges_masked = edges * (1 - masks)
    images_masked = (images * (1 - masks)) + masks
    return edge_generator_forward(sd, torch.cat((images_masked, edges_masked, masks), dim=1))


def inpaint_model_forward(sd: SD, images, edges, masks) -> torch.Tensor:
    """InpaintingModel.forward, edgeconnect/models.py:236-240."""
    images_masked = (images * (1 - masks).float()) + masks
    return inpaint_generator_forward(sd, torch.cat((images_masked, edges), dim=1))
