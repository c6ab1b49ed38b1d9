"""Drop-in for the generator half of the reference's ``edgeconnect.networks`` on MI355X.

Public surface kept (edgeconnect/networks.py:37-135): ``InpaintGenerator(residual_blocks=8,
init_weights=True)``, ``EdgeGenerator(residual_blocks=8, use_spectral_norm=True, init_weights=True)``
with the reference's ``state_dict`` schema (``encoder.{1,4,7}``, ``middle.{i}.conv_block.{1,5}``,
``decoder.{0,3,7}``; spectral-normed convs expose ``weight_orig / weight_u / weight_v``).
``Discriminator`` is training-only in the reference and is not provided.

Execution: spectral norm is folded once at pack time (eval mode does no power iteration); the
dilated / reflect-padded 3x3 convs, the zero-padded 4x4 stride-2 convs and the two
ConvTranspose2d(k4,s2,p1) layers (as four 2x2 phase convolutions in one launch) all run on the
same implicit-GEMM kernel; InstanceNorm+ReLU is applied by the consumer conv while staging.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops, pack
from ..nn_base import ConvP, FusedNet, SNConvP, entry_point


def _holder(cin, cout, k, spectral: bool, bias: bool = True, transposed: bool = False):
    if spectral:
        return SNConvP(cin, cout, k, bias=bias, transposed=transposed)
    return ConvP(cin, cout, k, bias=bias, transposed=transposed, init="normal02")


class ResnetBlock(nn.Module):
    """Parameter holder (reference networks.py:184-203): conv_block.1 (dilated) and conv_block.5."""

    def __init__(self, dim, dilation=1, use_spectral_norm=False):
        super().__init__()
        self.dilation = dilation
        I = nn.Identity
        self.conv_block = nn.Sequential(
            I(), _holder(dim, dim, 3, use_spectral_norm, bias=not use_spectral_norm), I(), I(),
            I(), _holder(dim, dim, 3, use_spectral_norm, bias=not use_spectral_norm), I())


class _Generator(FusedNet):
    def __init__(self, cin: int, cout: int, residual_blocks: int, spectral: bool, final_act: int):
        super().__init__()
        self.cin, self.cout, self.final_act = cin, cout, final_act
        I = nn.Identity
        self.encoder = nn.Sequential(
            I(), _holder(cin, 64, 7, spectral), I(), I(),
            _holder(64, 128, 4, spectral), I(), I(),
            _holder(128, 256, 4, spectral), I(), I())
        self.middle = nn.Sequential(*[ResnetBlock(256, 2, use_spectral_norm=spectral) for _ in range(residual_blocks)])
        self.decoder = nn.Sequential(
            _holder(256, 128, 4, spectral, transposed=True), I(), I(),
            _holder(128, 64, 4, spectral, transposed=True), I(), I(),
            I(), ConvP(64, cout, 7, init="normal02"))

    def init_weights(self, init_type="normal", gain=0.02):
        """Kept for API compatibility (reference networks.py:9-34); holders are already N(0, gain)."""
        if init_type != "normal":
            raise NotImplementedError(init_type)
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, ConvP):
                    m.weight.normal_(0.0, gain)
                    if m.bias is not None:
                        m.bias.zero_()
        self.refresh()

    # ------------------------------------------------------------------ packing
    @staticmethod
    def _wb(h):
        if isinstance(h, SNConvP):
            return pack.fold_spectral_norm(h.weight_orig, h.weight_u, h.weight_v, h.transposed), h.bias
        return h.weight, h.bias

    def _build_plans(self, device) -> dict:
        e, d = self.encoder, self.decoder
        P = {"stem": pack.pack_conv(*self._wb(e[1]), pad=3, pad_mode=L.PAD_REFLECT).to(device),
             "down1": pack.pack_conv(*self._wb(e[4]), stride=2, pad=1).to(device),
             "down2": pack.pack_conv(*self._wb(e[7]), stride=2, pad=1).to(device),
             "mid": [],
             "up1": pack.pack_conv_transpose_k4s2p1(*self._wb(d[0])).to(device),
             "up2": pack.pack_conv_transpose_k4s2p1(*self._wb(d[3])).to(device),
             # the same filters as four dense 2x2 plans each: halo-kernel launches with fused IN statistics
             "up1_ph": [q.to(device) for q in pack.pack_conv_transpose_k4s2p1_phases(*self._wb(d[0]))],
             "up2_ph": [q.to(device) for q in pack.pack_conv_transpose_k4s2p1_phases(*self._wb(d[3]))],
             "head": pack.pack_conv_rowsplit(*self._wb(d[7]), pad=3, pad_mode=L.PAD_REFLECT).to(device)}
        for blk in self.middle:
            dl = blk.dilation
            P["mid"].append((pack.pack_conv(*self._wb(blk.conv_block[1]), pad=dl, dil=dl, pad_mode=L.PAD_REFLECT).to(device),
                             pack.pack_conv(*self._wb(blk.conv_block[5]), pad=1, pad_mode=L.PAD_REFLECT).to(device)))
        return P

    # ------------------------------------------------------------------ execution
    def _run(self, x_nhwc: torch.Tensor) -> torch.Tensor:
        P = self._plans
        AR = L.PRE_AFFINE_RELU
        c, st = ops.conv_in(P["stem"], x_nhwc)
        c, st = ops.conv_in(P["down1"], c, pre_op=AR, pre=st, pre_bstride=c.shape[1])
        c, st = ops.conv_in(P["down2"], c, pre_op=AR, pre=st, pre_bstride=c.shape[1])
        y = ops.affine_act(c, st[0], st[1], L.ACT_RELU)
        for pa, pb in P["mid"]:
            a, sa = ops.conv_in(pa, y)
            b, sb = ops.conv_in(pb, a, pre_op=AR, pre=sa, pre_bstride=a.shape[1])
            y = ops.affine_act(b, sb[0], sb[1], L.ACT_NONE, res=y)
        def halo_ok(t):                                      # every phase launch qualifies for the halo kernel
            return ops.halo_precision() and t.shape[1] % 32 == 0 and t.shape[2] % 8 == 0 and t.shape[3] % 16 == 0
        c, st = ops.conv_in(P["up1_ph"] if halo_ok(y) else P["up1"], y)
        c, st = ops.conv_in(P["up2_ph"] if halo_ok(c) else P["up2"], c, pre_op=AR, pre=st, pre_bstride=c.shape[1])
        return ops.conv_rowsplit(P["head"], c, pre_op=AR, pre=st, pre_bstride=c.shape[1], act=self.final_act)

    @entry_point
    def run_model(self, images: torch.Tensor, edges: torch.Tensor, masks: torch.Tensor, mode: int) -> torch.Tensor:
        """EdgeModel.forward (mode 0) / InpaintingModel.forward (mode 1): input assembly + generator
        (edgeconnect/models.py:130-135, 236-240)."""
        self._ensure(images)
        return self._run(ops.ec_inputs(images, edges, masks, mode))

    @entry_point
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._ensure(x)
        if x.dim() != 4 or x.shape[1] != self.cin or x.shape[2] % 4 or x.shape[3] % 4:
            raise ValueError(f"{type(self).__name__} expects [B,{self.cin},H,W] with H,W multiples of 4, got {tuple(x.shape)}")
        return self._run(ops.as_nhwc(x))


class InpaintGenerator(_Generator):
    def __init__(self, residual_blocks=8, init_weights=True):
        super().__init__(4, 3, residual_blocks, spectral=False, final_act=L.ACT_TANH01)


class EdgeGenerator(_Generator):
    def __init__(self, residual_blocks=8, use_spectral_norm=True, init_weights=True):
        super().__init__(3, 1, residual_blocks, spectral=use_spectral_norm, final_act=L.ACT_SIGMOID)


class Discriminator(nn.Module):
    """Name kept importable for `from edgeconnect.networks import InpaintGenerator, EdgeGenerator, Discriminator`
    (reference edgeconnect/models.py:5).  The PatchGAN discriminator (reference networks.py:138-181) is used only by
    the training half of EdgeModel / InpaintingModel (models.py:62, 154: adversarial loss), which is outside the
    inference hot path: constructing it here is an error rather than a silent CPU module."""

    def __init__(self, in_channels, use_sigmoid=True, use_spectral_norm=True, init_weights=True):
        super().__init__()
        raise NotImplementedError("edgeconnect.networks.Discriminator belongs to EdgeConnect's training half, which the "
                                  "MI355X inference path does not provide (SURVEY.md §2 row 8)")
