"""Drop-in for the network half of the reference's ``warp_learn.models`` on MI355X.

Public surface kept (warp_learn/models.py:190-208): ``G_Resnet(input_nc, output_nc=3, num_downs=2,
n_res=3, ngf=64, norm='inst', nl_layer='relu')`` with ``forward(image)`` and ``decode(content)``,
and the reference's ``state_dict`` schema (``enc_content.model.{i}.conv.*``,
``...model.{j}.model.{0,1}.conv.*``, ``dec.model.{k}.norm.{gamma,beta}``).

Execution (NHWC, libfusg): every convolution reads its input through the fused reflect-pad
address mapping; ``InstanceNorm -> ReLU -> conv`` chains never materialise the normalised tensor
(a streaming statistics pass produces per-(b,c) scale/shift which the consumer conv applies while
staging its tile); the decoder's ``nearest-upsample -> 5x5 conv`` reads the low-resolution tensor
directly (upsample folded into the gather), its custom LayerNorm (unbiased std, eps on std)
becomes the next conv's per-(b,c) affine prologue, and ``x + IN(conv(.))`` is one elementwise pass.
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops, pack
from ..nn_base import ConvP, FusedNet, dev_vec, entry_point


class LayerNorm(nn.Module):
    """Parameter holder of the ICN's custom LayerNorm (gamma, beta); reference models.py:15-35."""

    def __init__(self, num_features: int, eps: float = 1e-5, affine: bool = True):
        super().__init__()
        self.num_features, self.eps, self.affine = num_features, eps, affine
        if affine:
            self.gamma = nn.Parameter(torch.Tensor(num_features).uniform_())
            self.beta = nn.Parameter(torch.zeros(num_features))


class Conv2dBlock(nn.Module):
    """Parameter holder: [norm.{gamma,beta}], conv.{weight,bias} (reference models.py:38-90)."""

    def __init__(self, input_dim, output_dim, kernel_size, stride, padding=0, norm="none", activation="relu",
                 pad_type="zero"):
        super().__init__()
        if norm not in ("inst", "ln", "none") or activation not in ("relu", "tanh", "none") \
                or pad_type not in ("reflect", "zero"):
            raise NotImplementedError(f"Conv2dBlock(norm={norm}, activation={activation}, pad_type={pad_type})")
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.norm_type, self.act_type, self.pad_type = norm, activation, pad_type
        if norm == "ln":
            self.norm = LayerNorm(output_dim)
        self.conv = ConvP(input_dim, output_dim, kernel_size)


class ResBlock(nn.Module):
    def __init__(self, dim, norm="inst", activation="relu", pad_type="zero"):
        super().__init__()
        self.model = nn.Sequential(Conv2dBlock(dim, dim, 3, 1, 1, norm=norm, activation=activation, pad_type=pad_type),
                                   Conv2dBlock(dim, dim, 3, 1, 1, norm=norm, activation="none", pad_type=pad_type))


class ResBlocks(nn.Module):
    def __init__(self, num_blocks, dim, norm="inst", activation="relu", pad_type="zero"):
        super().__init__()
        self.model = nn.Sequential(*[ResBlock(dim, norm=norm, activation=activation, pad_type=pad_type)
                                     for _ in range(num_blocks)])


class ContentEncoder(nn.Module):
    def __init__(self, n_downsample, n_res, input_dim, dim, norm, activ, pad_type="zero"):
        super().__init__()
        layers = [Conv2dBlock(input_dim, dim, 7, 1, 3, norm=norm, activation=activ, pad_type="reflect")]
        for _ in range(n_downsample):
            layers.append(Conv2dBlock(dim, 2 * dim, 4, 2, 1, norm=norm, activation=activ, pad_type="reflect"))
            dim *= 2
        layers.append(ResBlocks(n_res, dim, norm=norm, activation=activ, pad_type=pad_type))
        self.model = nn.Sequential(*layers)
        self.output_dim = dim


class Decoder(nn.Module):
    def __init__(self, n_upsample, n_res, dim, output_dim, norm="batch", activ="relu", pad_type="zero"):
        super().__init__()
        layers = [ResBlocks(n_res, dim, norm, activ, pad_type=pad_type)]
        for _ in range(n_upsample):
            layers += [nn.Identity(),                                   # slot of the parameter-free Upsample
                       Conv2dBlock(dim, dim // 2, 5, 1, 2, norm="ln", activation=activ, pad_type="reflect")]
            dim //= 2
        layers.append(Conv2dBlock(dim, output_dim, 7, 1, 3, norm="none", activation="tanh", pad_type="reflect"))
        self.model = nn.Sequential(*layers)


class G_Resnet(FusedNet):
    def __init__(self, input_nc, output_nc=3, num_downs=2, n_res=3, ngf=64, norm="inst", nl_layer="relu"):
        super().__init__()
        if norm != "inst" or nl_layer != "relu":
            raise NotImplementedError("G_Resnet: only norm='inst', nl_layer='relu' (the reference's configuration)")
        self.input_nc, self.output_nc, self.num_downs, self.n_res = input_nc, output_nc, num_downs, n_res
        self.enc_content = ContentEncoder(num_downs, n_res, input_nc, ngf, norm, nl_layer, pad_type="reflect")
        self.dec = Decoder(num_downs, n_res, self.enc_content.output_dim, output_nc, norm=norm, activ=nl_layer,
                           pad_type="reflect")

    # ------------------------------------------------------------------ packing
    @staticmethod
    def _pack_block(blk: Conv2dBlock, device, upsample: int = 0):
        return pack.pack_conv(blk.conv.weight, blk.conv.bias, stride=blk.stride, pad=blk.padding,
                              pad_mode=L.PAD_REFLECT if blk.pad_type == "reflect" else L.PAD_ZERO,
                              upsample=upsample).to(device)

    def _pack_head(self, blk: Conv2dBlock, device):
        """7x7 -> output_nc head: row-split form when output_nc*7 fits one 32-wide N tile."""
        if blk.conv.cout * blk.kernel_size <= 32 and blk.pad_type == "reflect":
            return pack.pack_conv_rowsplit(blk.conv.weight, blk.conv.bias, pad=blk.padding, pad_mode=L.PAD_REFLECT).to(device)
        return self._pack_block(blk, device)

    def _pack_res(self, rbs: ResBlocks, device) -> List[tuple]:
        return [(self._pack_block(rb.model[0], device), self._pack_block(rb.model[1], device)) for rb in rbs.model]

    def _build_plans(self, device) -> dict:
        enc, dec = self.enc_content.model, self.dec.model
        nd = self.num_downs
        # 7x7 stem on 21 channels: K-channels padded to 24 - the tap-unit kernel stages the whole 14x22x24 halo once
        # and walks K in 8-channel units (74 MFMA k-steps; padding the channels to 32 for the halo kernel took 98)
        stem = enc[0]
        P = {"stem": pack.pack_conv(stem.conv.weight, stem.conv.bias, stride=1, pad=stem.padding, pad_mode=L.PAD_REFLECT,
                                    cin_pad=4).to(device),
             "down": [self._pack_block(enc[1 + i], device) for i in range(nd)],
             "enc_res": self._pack_res(enc[1 + nd], device),
             "dec_res": self._pack_res(dec[0], device),
             "up": [], "up_phases": [], "up_ring": [], "ln": [],
             "head": self._pack_head(dec[1 + 2 * nd], device)}
        for i in range(nd):
            blk = dec[2 + 2 * i]
            P["up"].append(self._pack_block(blk, device, upsample=1))
            # 5x5 reflect-padded conv after a 2x nearest upsample: also packed as the four 3x3 phase convolutions of
            # the low-res input (pack.up2_phase_weights: 2.8x fewer MACs), stacked into one launch whose DepthToSpace
            # store interleaves the phases; ops.conv_up2 picks the route per input
            ok = blk.kernel_size == 5 and blk.padding == 2 and blk.stride == 1 and blk.pad_type == "reflect"
            ok = ok and blk.conv.weight.shape[0] % 4 == 0
            P["up_phases"].append(pack.pack_conv_up2_d2s(blk.conv.weight, blk.conv.bias).to(device) if ok else None)
            # ... and, on request only (FUSG_UP2_RING9: measured no faster than the four 25-tap windows, ops.conv_up2), the
            # outermost ring of output pixels as twelve border-specific 3x3 launches (pack.pack_conv_up2_ring)
            P["up_ring"].append({k: v.to(device) for k, v in pack.pack_conv_up2_ring(blk.conv.weight, blk.conv.bias).items()}
                                if ok and ops._env_set("FUSG_UP2_RING9") else None)
            P["ln"].append((dev_vec(blk.norm.gamma, device), dev_vec(blk.norm.beta, device), blk.norm.eps))
        return P

    # ------------------------------------------------------------------ execution
    @staticmethod
    def _resblocks(plans, y: torch.Tensor) -> torch.Tensor:
        for pa, pb in plans:
            a, sa = ops.conv_in(pa, y)                      # InstanceNorm statistics come out of the conv epilogue
            b, sb = ops.conv_in(pb, a, pre_op=L.PRE_AFFINE_RELU, pre=sa, pre_bstride=a.shape[1])
            y = ops.affine_act(b, sb[0], sb[1], L.ACT_NONE, res=y)
        return y

    def _encode(self, P, x: torch.Tensor) -> torch.Tensor:
        c, st = ops.conv_in(P["stem"], ops.as_nhwc(x, cpad=P["stem"].c0k))
        for p in P["down"]:
            c, st = ops.conv_in(p, c, pre_op=L.PRE_AFFINE_RELU, pre=st, pre_bstride=c.shape[1])
        y = ops.affine_act(c, st[0], st[1], L.ACT_RELU)
        return self._resblocks(P["enc_res"], y)

    def _decode(self, P, y: torch.Tensor) -> torch.Tensor:
        y = self._resblocks(P["dec_res"], y)
        pre_op, pre, bs = L.PRE_NONE, None, 0
        for p, ph, rg, (gamma, beta, eps) in zip(P["up"], P["up_phases"], P["up_ring"], P["ln"]):
            if ph is not None and ops.up2_phases_ok(y):
                y = ops.conv_up2(p, ph, y, pre_op=pre_op, pre=pre, pre_bstride=bs, ring=rg)
                pre = ops.layernorm_stats(y, gamma, beta, eps)     # the border pass rewrites pixels: no fused statistics
            else:
                y, pre = ops.conv_ln(p, y, gamma, beta, eps, pre_op=pre_op, pre=pre, pre_bstride=bs)
            pre_op, bs = L.PRE_AFFINE_RELU, y.shape[1]
        if P["head"].rowsplit is not None:
            return ops.conv_rowsplit(P["head"], y, pre_op=pre_op, pre=pre, pre_bstride=bs, act=L.ACT_TANH)
        return ops.conv(P["head"], y, pre_op=pre_op, pre=pre, pre_bstride=bs, act=L.ACT_TANH, nchw_out=True)

    @entry_point
    def decode(self, content: torch.Tensor) -> torch.Tensor:
        P = self._ensure(content)
        return self._decode(P, ops.as_nhwc(content))

    @entry_point
    def forward(self, image: torch.Tensor) -> torch.Tensor:
        P = self._ensure(image)
        if image.dim() != 4 or image.shape[1] != self.input_nc or image.shape[2] % 4 or image.shape[3] % 4:
            raise ValueError(f"G_Resnet expects [B,{self.input_nc},H,W] with H,W multiples of 4, got {tuple(image.shape)}")
        return self._decode(P, self._encode(P, image))


def get_icn_inputs(planes, sketch_normal, sketch_mask, central_crop, icn_w: int, icn_h: int):
    """ICN input assembly (reference warp_learn/models.py:323-366): crop all inputs to the 1.1x square box around the
    sketch mask, resize to (icn_w, icn_h), convert to Lab, scale to [-1, 1] and stack [sketch(3), central crop(3),
    planes(5x3)] -> float32 [1, 21, icn_h, icn_w] + crop_info.  Runs on the device (one libfusg launch,
    planes_utils.get_icn_inputs); numpy or CUDA uint8 images are accepted.  OpenCV arithmetic, parity unpinned."""
    from .planes_utils import get_icn_inputs as _impl
    return _impl(planes, sketch_normal, sketch_mask, central_crop, icn_w, icn_h)
