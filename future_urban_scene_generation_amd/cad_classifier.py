"""CAD-model classifier of the reference on MI355X: VGG-19 with a 10-way last layer (run_test.py:47-58,
``models.vgg19(pretrained=True); classifier[6] = Linear(4096, 10)``; called once per vehicle on the same 256 x 256
ImageNet-normalised crop as the hourglass, trajectory_inference.py:59-69, the CAD index is ``argmax`` of the logits).

``VGG19Classifier(num_classes=10)`` exposes the ``state_dict`` schema of that torchvision module - ``features.{0, 2,
5, 7, 10, 12, 14, 16, 19, 21, 23, 25, 28, 30, 32, 34}.{weight, bias}`` and ``classifier.{0, 3, 6}.{weight, bias}`` -
so the reference's ``cads/model.pth`` loads unchanged, and the usual ``.to / .eval / load_state_dict / __call__``.
torchvision is not a dependency (and is absent from the build container): the architecture is restated from the
published VGG-19 configuration "E"; PARITY UNPINNED against torchvision itself.

Execution: 16 fused 3x3 conv + ReLU launches (tap-unit kernel for the 3-channel stem, halo kernel for the rest), 5
max-pool launches, and the classifier as three more convolution launches: ``AdaptiveAvgPool2d((7, 7))`` followed by
``Linear(25088, 4096)`` is one linear map of the s x s x 512 feature map (s = H / 32), folded at pack time into an
s x s "valid" convolution; the other two Linear layers are 1 x 1 convolutions on a 1 x 1 image.  The logits decide an
integer (the CAD index), so like the hourglass this network keeps the fp32-class path under ``precision="bf16"``.
"""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib as L
from . import ops, pack
from .nn_base import ConvP, FusedNet, entry_point

CFG_E = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M")


class LinearP(nn.Module):
    """Parameter holder with nn.Linear's schema."""

    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin).normal_(0.0, 0.01))
        self.bias = nn.Parameter(torch.zeros(cout))


def vgg19_schema(num_classes: int = 10) -> "OrderedDict[str, tuple]":
    """state_dict schema (key -> (shape, dtype)) of torchvision's vgg19 with its last Linear replaced."""
    out, idx, cin = OrderedDict(), 0, 3
    for v in CFG_E:
        if v == "M":
            idx += 1
            continue
        out[f"features.{idx}.weight"] = ((v, cin, 3, 3), "float32")
        out[f"features.{idx}.bias"] = ((v,), "float32")
        idx, cin = idx + 2, v                                  # conv, ReLU
    for i, (a, b) in zip((0, 3, 6), ((512 * 49, 4096), (4096, 4096), (4096, num_classes))):
        out[f"classifier.{i}.weight"] = ((b, a), "float32")
        out[f"classifier.{i}.bias"] = ((b,), "float32")
    return out


class VGG19Classifier(FusedNet):
    def __init__(self, num_classes: int = 10):
        super().__init__()
        feats, idx, cin = OrderedDict(), 0, 3
        self._order = []                                       # ("conv", key) | ("pool",)
        for v in CFG_E:
            if v == "M":
                self._order.append(("pool",))
                idx += 1
                continue
            feats[str(idx)] = ConvP(cin, v, 3)
            self._order.append(("conv", str(idx)))
            idx, cin = idx + 2, v
        self.features = nn.ModuleDict(feats)
        self.classifier = nn.ModuleDict({"0": LinearP(512 * 49, 4096), "3": LinearP(4096, 4096),
                                         "6": LinearP(4096, num_classes)})
        self.num_classes = num_classes

    def _build_plans(self, device) -> dict:
        P = {"convs": {k: pack.pack_conv(m.weight, m.bias, pad=1).to(device) for k, m in self.features.items()},
             "fc1": {},                                        # per feature-map size s: pool + Linear folded
             "fc2": pack.pack_conv(self.classifier["3"].weight[:, :, None, None], self.classifier["3"].bias).to(device),
             "fc3": pack.pack_conv(self.classifier["6"].weight[:, :, None, None], self.classifier["6"].bias).to(device)}
        return P

    def _fc1(self, P: dict, s: int, device):
        """AdaptiveAvgPool2d((7, 7)) then Linear(25088, 4096) as one s x s valid convolution: W'[o, c, y, x] =
        sum_{p} W[o, c, p] * pool[p, (y, x)] with pool = the averaging matrix torch applies for an s x s input."""
        if s not in P["fc1"]:
            with torch.no_grad():
                w = self.classifier["0"].weight.detach().to("cpu", torch.float32).view(4096, 512, 49)
                eye = torch.eye(s * s, dtype=torch.float32).view(s * s, 1, s, s)
                pool = torch.nn.functional.adaptive_avg_pool2d(eye, (7, 7)).view(s * s, 49).t().contiguous()   # [49, s*s]
                wf = (w.double() @ pool.double()).to(torch.float32).view(4096, 512, s, s) if s != 7 else w.view(4096, 512, 7, 7)
                P["fc1"][s] = pack.pack_conv(wf, self.classifier["0"].bias).to(device)
        return P["fc1"][s]

    @entry_point
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if ops.PRECISION == "bf16":                            # an argmax decides the CAD model: fp32-class path only
            with ops.precision("f16x3"):
                return self._forward(x)
        return self._forward(x)

    def _forward(self, x: torch.Tensor) -> torch.Tensor:
        P = self._ensure(x)
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != x.shape[3] or x.shape[2] % 32 or x.shape[2] < 32:
            raise ValueError(f"VGG19Classifier expects [B, 3, S, S] with S a multiple of 32, got {tuple(x.shape)}")
        t = ops.as_nhwc(x)
        for op in self._order:
            t = ops.maxpool2(t) if op[0] == "pool" else ops.conv(P["convs"][op[1]], t, act=L.ACT_RELU)
        t = ops.conv(self._fc1(P, t.shape[2], x.device), t, act=L.ACT_RELU)      # Dropout: identity in eval
        t = ops.conv(P["fc2"], t, act=L.ACT_RELU)
        t = ops.conv(P["fc3"], t, nchw_out=True)
        return t.reshape(t.shape[0], self.num_classes)
