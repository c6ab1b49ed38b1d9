"""Drop-in for the reference's ``stacked_hourglass.models`` on MI355X.

Same public surface as the reference (stacked_hourglass/models.py:89-167): ``HourglassNet(num_stacks,
num_blocks, num_classes)``, identical ``state_dict`` keys/shapes (684 entries for (2, 1, 12)),
``forward(x[B,3,H,W]) -> {'heatmaps': [Tensor[B,num_classes,H/4,W/4]] * num_stacks}``.

Execution is a flat sequence of libfusg launches over NHWC activations:
  * every eval BatchNorm that directly follows a convolution (bn2, bn3, the stem's bn1, the fc BN)
    is folded into that convolution's filter at pack time; its ReLU is the conv epilogue;
  * a Bottleneck's leading ``bn1 -> ReLU`` cannot be folded (its input is also the residual), so it
    is applied while the consumer 1x1 conv stages its input tile (PRE_AFFINE_RELU);
  * residual adds (``out += residual``, ``x + fc_ + score_``) are conv epilogues;
  * max-pool and ``up1 + upsample(low3)`` are single HBM-bound kernels.
A Bottleneck is therefore 3 launches (4 with a downsample conv) instead of 10 framework ops - and, on the
split-fp16 path, ONE launch per block (fusg_hg_bottleneck, 64 or 128 planes: both intermediates stay in LDS).
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops, pack
from ..nn_base import BNP, ConvP, FusedNet, dev_vec, entry_point


class Bottleneck(nn.Module):
    """Parameter holder for the pre-activation bottleneck (reference models.py:5-42)."""
    expansion = 2

    def __init__(self, inplanes: int, planes: int, stride: int = 1, downsample=None):
        super().__init__()
        if stride != 1:
            raise NotImplementedError("the hourglass only instantiates stride-1 bottlenecks")
        self.bn1 = BNP(inplanes)
        self.conv1 = ConvP(inplanes, planes, 1)
        self.bn2 = BNP(planes)
        self.conv2 = ConvP(planes, planes, 3)
        self.bn3 = BNP(planes)
        self.conv3 = ConvP(planes, planes * 2, 1)
        self.downsample = downsample
        self.stride = stride


class Hourglass(nn.Module):
    """Parameter holder: ``hg[d][j]`` = Sequential of Bottlenecks (reference models.py:59-68)."""

    def __init__(self, block, num_blocks: int, planes: int, depth: int):
        super().__init__()
        self.depth = depth

        def residual():
            return nn.Sequential(*[block(planes * block.expansion, planes) for _ in range(num_blocks)])

        self.hg = nn.ModuleList([nn.ModuleList([residual() for _ in range(4 if d == 0 else 3)])
                                 for d in range(depth)])


class HourglassNet(FusedNet):
    def __init__(self, num_stacks: int, num_blocks: int, num_classes: int):
        super().__init__()
        block = Bottleneck
        self.inplanes = 64
        self.num_feats = 128
        self.num_stacks = num_stacks
        self.num_classes = num_classes
        self.conv1 = ConvP(3, self.inplanes, 7)
        self.bn1 = BNP(self.inplanes)
        self.layer1 = self._make_residual(block, self.inplanes, 1)
        self.layer2 = self._make_residual(block, self.inplanes, 1)
        self.layer3 = self._make_residual(block, self.num_feats, 1)
        ch = self.num_feats * block.expansion
        hg, res, fc, score, fc_, score_ = [], [], [], [], [], []
        for i in range(num_stacks):
            hg.append(Hourglass(block, num_blocks, self.num_feats, 4))
            res.append(self._make_residual(block, self.num_feats, num_blocks))
            fc.append(nn.Sequential(ConvP(ch, ch, 1), BNP(ch)))
            score.append(ConvP(ch, num_classes, 1))
            if i < num_stacks - 1:
                fc_.append(ConvP(ch, ch, 1))
                score_.append(ConvP(num_classes, ch, 1))
        self.hg = nn.ModuleList(hg)
        self.res = nn.ModuleList(res)
        self.fc = nn.ModuleList(fc)
        self.score = nn.ModuleList(score)
        self.fc_ = nn.ModuleList(fc_)
        self.score_ = nn.ModuleList(score_)

    def _make_residual(self, block, planes: int, blocks: int):
        downsample = None
        if self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(ConvP(self.inplanes, planes * block.expansion, 1))
        layers = [block(self.inplanes, planes, 1, downsample)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    # ------------------------------------------------------------------ packing
    @staticmethod
    def _bn_ss(bn: BNP):
        return pack.bn_scale_shift(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)

    def _pack_bottleneck(self, blk: Bottleneck, device) -> dict:
        s1, h1 = self._bn_ss(blk.bn1)
        w1, b1 = pack.fold_bn_after_conv(blk.conv1.weight, blk.conv1.bias, *self._bn_ss(blk.bn2))
        w2, b2 = pack.fold_bn_after_conv(blk.conv2.weight, blk.conv2.bias, *self._bn_ss(blk.bn3))
        p = {"pre": (dev_vec(s1, device), dev_vec(h1, device)),
             "c1": pack.pack_conv(w1, b1).to(device),
             "c2": pack.pack_conv(w2, b2, pad=1).to(device),
             "c3": pack.pack_conv(blk.conv3.weight, blk.conv3.bias).to(device),
             "ds": None}
        if blk.downsample is not None:
            p["ds"] = pack.pack_conv(blk.downsample[0].weight, blk.downsample[0].bias).to(device)
        return p

    def _pack_seq(self, seq, device) -> List[dict]:
        return [self._pack_bottleneck(b, device) for b in seq]

    def _build_plans(self, device) -> dict:
        w, b = pack.fold_bn_after_conv(self.conv1.weight, self.conv1.bias, *self._bn_ss(self.bn1))
        P = {"stem": pack.pack_conv(w, b, stride=2, pad=3).to(device),
             "layer1": self._pack_seq(self.layer1, device),
             "layer2": self._pack_seq(self.layer2, device),
             "layer3": self._pack_seq(self.layer3, device),
             "hg": [], "res": [], "fc": [], "score": [], "fc_": [], "score_": []}
        for i in range(self.num_stacks):
            P["hg"].append([[self._pack_seq(seq, device) for seq in level] for level in self.hg[i].hg])
            P["res"].append(self._pack_seq(self.res[i], device))
            wf, bf = pack.fold_bn_after_conv(self.fc[i][0].weight, self.fc[i][0].bias, *self._bn_ss(self.fc[i][1]))
            P["fc"].append(pack.pack_conv(wf, bf).to(device))
            P["score"].append(pack.pack_conv(self.score[i].weight, self.score[i].bias).to(device))
            if i < self.num_stacks - 1:
                P["fc_"].append(pack.pack_conv(self.fc_[i].weight, self.fc_[i].bias).to(device))
                P["score_"].append(pack.pack_conv(self.score_[i].weight, self.score_[i].bias).to(device))
                # x + fc_(y) + score_(score) (models.py:164-166) as ONE two-source 1x1 convolution of cat[y, score] with
                # the residual x: the score map's 12 channels are zero-padded to a 32-channel source (halo kernel)
                wj = torch.cat([self.fc_[i].weight.detach().float().cpu(), self.score_[i].weight.detach().float().cpu()], dim=1)
                bj = self.fc_[i].bias.detach().float().cpu() + self.score_[i].bias.detach().float().cpu()
                P.setdefault("join", []).append(pack.pack_conv(wj, bj, c_split=(wj.shape[1] - self.num_classes, self.num_classes),
                                                               cin_pad=32).to(device))
                # ... and the score conv with its output channels zero-padded to 32, so that the launch itself writes the
                # padding every pass (a recorded pass replays addresses: a memset at record time would not be replayed)
                ws = torch.zeros(32, self.score[i].weight.shape[1], 1, 1)
                bs = torch.zeros(32)
                ws[:self.num_classes] = self.score[i].weight.detach().float().cpu()
                bs[:self.num_classes] = self.score[i].bias.detach().float().cpu()
                P.setdefault("score32", []).append(pack.pack_conv(ws, bs).to(device))
        return P

    # ------------------------------------------------------------------ execution
    @staticmethod
    def _bottleneck(p: dict, x: torch.Tensor) -> torch.Tensor:
        if ops.bottleneck_ok(p, x):                  # f16x3: the whole block in one launch
            return ops.bottleneck(p, x, None if p["ds"] is None else ops.conv(p["ds"], x))
        t = ops.conv(p["c1"], x, pre_op=L.PRE_AFFINE_RELU, pre=p["pre"], act=L.ACT_RELU)
        t = ops.conv(p["c2"], t, act=L.ACT_RELU)
        r = x if p["ds"] is None else ops.conv(p["ds"], x)
        return ops.conv(p["c3"], t, res0=r)

    def _seq(self, plans: List[dict], x: torch.Tensor) -> torch.Tensor:
        for p in plans:
            x = self._bottleneck(p, x)
        return x

    def _hourglass(self, hp, n: int, x: torch.Tensor) -> torch.Tensor:
        up1 = self._seq(hp[n - 1][0], x)
        low1 = self._seq(hp[n - 1][1], ops.maxpool2(x))
        low2 = self._hourglass(hp, n - 1, low1) if n > 1 else self._seq(hp[n - 1][3], low1)
        low3 = self._seq(hp[n - 1][2], low2)
        return ops.upsample2_add(low3, up1)

    @entry_point
    def forward(self, x: torch.Tensor) -> Dict[str, List[torch.Tensor]]:
        if ops.PRECISION == "bf16":
            # the heat-map argmax is an integer contract (utils/keypoint_utils.py:85-88): single-pass bf16 moves 1 of 12
            # keypoints on the fixtures (tests::test_reduced_precision_evidence), so this network keeps the fp32-class path
            with ops.precision("f16x3"):
                return self._forward(x)
        return self._forward(x)

    def _forward(self, x: torch.Tensor) -> Dict[str, List[torch.Tensor]]:
        P = self._ensure(x)
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] % 64 or x.shape[3] % 64:
            raise ValueError(f"HourglassNet expects [B,3,H,W] with H,W multiples of 64, got {tuple(x.shape)}")
        heatmaps = []
        x = ops.conv(P["stem"], ops.as_nhwc(x), act=L.ACT_RELU)
        x = self._seq(P["layer1"], x)
        x = ops.maxpool2(x)
        x = self._seq(P["layer2"], x)
        x = self._seq(P["layer3"], x)
        for i in range(self.num_stacks):
            y = self._hourglass(P["hg"][i], 4, x)
            y = self._seq(P["res"][i], y)
            y = ops.conv(P["fc"][i], y, act=L.ACT_RELU)
            last = i == self.num_stacks - 1
            joined = not last and ops.halo_precision() and y.shape[1] % 32 == 0 and self.num_classes <= 32
            # (for the joined form the score map is the first num_classes channels of a 32-channel launch whose other
            # channels are exact zeros: a 32-channel source of the next launch)
            score = ops.conv(P["score32"][i], y)[:, :self.num_classes] if joined else ops.conv(P["score"][i], y)
            heatmaps.append(ops.to_nchw(score))
            if joined:
                x = ops.conv(P["join"][i], y, score, res0=x)
            elif not last:
                t = ops.conv(P["fc_"][i], y, res0=x)
                x = ops.conv(P["score_"][i], score, res0=t)
        return {"heatmaps": heatmaps}


def get_maxima_device(heatmaps: torch.Tensor):
    """Device-side integer half of get_maxima (utils/keypoint_utils.py:66-92): returns the float64
    [B, C, 2] array of (x / w, y / h) like the reference, computed from a fused argmax kernel on the
    un-upsampled map (nearest upsampling preserves the first-occurrence argmax up to the scale)."""
    import numpy as np
    b, c, h, w = heatmaps.shape
    idx = ops.argmax_hw(heatmaps).to("cpu").numpy().astype(np.int64)
    out = np.zeros((b, c, 2))
    out[..., 0] = (idx % w) / w
    out[..., 1] = (idx // w) / h
    return out
