"""Drop-in for the reference's ``vunet.models`` on MI355X.

Public surface kept (vunet/models.py:191-481): ``Vunet_fix_res(args)`` with ``args`` an
``argparse.Namespace(up_mode, w_norm, drop_prob, vunet_256)``; the reference's ``state_dict`` schema
(336 entries ``...conv.{bias,weight_g,weight_v}`` for the shipped configuration); and the five
entry points ``forward_enc_up / forward_enc_down / forward_dec_up / forward_dec_down / forward``
with the same argument order, return structure and side effect (``forward_dec_down`` pops the
caller's ``skips`` list empty).

Execution notes
  * weight_norm is a pure function of the parameters: folded once at pack time.
  * ``Residual`` = ``x + conv3x3(ELU(cat[x, skip]))`` is ONE launch: the concat is two gather
    sources, ELU is applied while staging the tile, the residual add is the epilogue.
  * ``UpSample('subpixel')`` = 3x3 conv with a DepthToSpace(DCR) store permutation: one launch.
  * ``Sampler`` noise is drawn exactly like the reference (vunet/layers.py:163-167): on the CPU
    default generator, shape ``mu.size()``, in the reference's call order - so the same
    ``torch.manual_seed`` reproduces the reference's stochastic output.
  * The four sampler means of an autoregressive block are written straight into the channel slices
    of the buffer that DepthToSpace consumes (no torch.cat).
"""
from __future__ import annotations

import argparse
from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops, pack
from ..nn_base import FusedNet, WNConvP, entry_point
from .layers import (Activation, DepthToSpace, DownSample, MyConv2d, NiN, Residual, Sampler, SpaceToDepth,
                     UpSample)


# ------------------------------------------------------------------------------------------------
# parameter-holder blocks (same attribute names / registration order as the reference)
# ------------------------------------------------------------------------------------------------
def _res(c_in, c_out, dp, wn):
    return Residual(c_in=c_in, c_out=c_out, activation=Activation("elu"), drop_prob=dp, w_norm=wn)


class AutoRegressiveBlock(nn.Module):
    def __init__(self, drop_prob, w_norm):
        super().__init__()
        self.s2d = SpaceToDepth(2)
        self.d2s = DepthToSpace(2)
        self.residual_init = _res(256, 128, drop_prob, w_norm)
        self.sampler_0 = Sampler(512, 128, w_norm)
        self.residual_0 = _res(1024, 512, drop_prob, w_norm)
        self.sampler_1 = Sampler(512, 128, w_norm)
        self.residual_1 = _res(1024, 512, drop_prob, w_norm)
        self.sampler_2 = Sampler(512, 128, w_norm)
        self.residual_2 = _res(1024, 512, drop_prob, w_norm)
        self.sampler_3 = Sampler(512, 128, w_norm)
        self.nin_0 = NiN(128, 512, w_norm)
        self.nin_1 = NiN(128, 512, w_norm)
        self.nin_2 = NiN(128, 512, w_norm)
        self.residual_s2d = _res(128, 128, drop_prob, w_norm)


class DownBlock(nn.Module):
    def __init__(self, c_in, c_out, drop_prob, w_norm):
        super().__init__()
        self.down = DownSample(c_in, c_out, w_norm)
        self.residual_0 = _res(c_out, c_out, drop_prob, w_norm)
        self.residual_1 = _res(c_out, c_out, drop_prob, w_norm)


class UpBlock(nn.Module):
    def __init__(self, c_in, c_middle, c_out, up_mode, drop_prob, w_norm):
        super().__init__()
        self.residual_0 = _res(c_in, c_middle, drop_prob, w_norm)
        self.residual_1 = _res(c_in, c_middle, drop_prob, w_norm)
        self.up = UpSample(c_middle, c_out, w_norm, up_mode)


class InitBlock(nn.Module):
    def __init__(self, c_in, c_out, drop_prob, w_norm):
        super().__init__()
        self.nin = NiN(c_in, c_out, w_norm)
        self.residual_0 = _res(c_out, c_out, drop_prob, w_norm)
        self.residual_1 = _res(c_out, c_out, drop_prob, w_norm)


class EndBlock(nn.Module):
    def __init__(self, c_in, c_middle, c_out, drop_prob, w_norm):
        super().__init__()
        self.residual_0 = _res(c_in, c_middle, drop_prob, w_norm)
        self.residual_1 = _res(c_in, c_middle, drop_prob, w_norm)
        self.conv = MyConv2d(c_middle, c_out, 3, 1, 1, w_norm)


class Vunet_fix_res(FusedNet):
    def __init__(self, args: argparse.Namespace):
        super().__init__()
        self.args = args
        wn = self.w_norm = args.w_norm
        dp = self.drop_prob = args.drop_prob
        um = self.up_mode = args.up_mode
        self.vunet_256 = args.vunet_256
        # appearance encoder
        self.app_encoder_1 = InitBlock(6, 128, dp, wn)
        self.app_encoder_1_a = DownBlock(128, 128, dp, wn)
        self.app_encoder_1_b = DownBlock(128, 128, dp, wn)
        if self.vunet_256:
            self.app_encoder_1_c = DownBlock(128, 128, dp, wn)
        self.app_encoder_2 = DownBlock(128, 128, dp, wn)
        self.app_encoder_3 = DownBlock(128, 128, dp, wn)
        self.app_encoder_4 = DownBlock(128, 128, dp, wn)
        self.app_skip_3_c = NiN(128, 128, wn)
        self.app_skip_4_c = NiN(128, 128, wn)
        # appearance decoder
        self.app_bottleneck = MyConv2d(128, 128, 1, 1, 0, wn)
        self.app_decoder_1_a = _res(256, 128, dp, wn)
        self.app_decoder_1_b = Sampler(128, 128, wn)
        self.app_decoder_1_c = MyConv2d(256, 128, 1, 1, 0, wn)
        self.app_decoder_1_d = _res(256, 128, dp, wn)
        self.app_decoder_1_e = UpSample(128, 128, wn, um)
        self.app_decoder_2_a = _res(128, 128, dp, wn)
        self.app_decoder_2_b = Sampler(128, 128, wn)
        # shape encoder
        self.shape_encoder_1 = InitBlock(3, 32, dp, wn)
        if self.vunet_256:
            self.shape_encoder_1_a = DownBlock(32, 32, dp, wn)
        self.shape_encoder_2 = DownBlock(32, 64, dp, wn)
        self.shape_encoder_3 = DownBlock(64, 128, dp, wn)
        self.shape_encoder_4 = DownBlock(128, 128, dp, wn)
        self.shape_encoder_5 = DownBlock(128, 128, dp, wn)
        self.shape_encoder_6 = DownBlock(128, 128, dp, wn)
        self.shape_skip_1_b = NiN(32, 32, wn)
        self.shape_skip_1_c = NiN(32, 32, wn)
        if self.vunet_256:
            self.shape_skip_1_a_b = NiN(32, 32, wn)
            self.shape_skip_1_a_c = NiN(32, 32, wn)
        for i, c in ((2, 64), (3, 128), (4, 128), (5, 128), (6, 128)):
            setattr(self, f"shape_skip_{i}_b", NiN(c, c, wn))
            setattr(self, f"shape_skip_{i}_c", NiN(c, c, wn))
        # shape decoder
        self.shape_bottleneck = MyConv2d(128, 128, 1, 1, 0, wn)
        self.shape_decoder_1 = AutoRegressiveBlock(dp, wn)
        self.shape_decoder_1_n = NiN(256, 128, wn)
        self.shape_decoder_1_o = _res(256, 128, dp, wn)
        self.shape_decoder_1_p = UpSample(128, 128, wn, um)
        self.shape_decoder_2 = AutoRegressiveBlock(dp, wn)
        self.shape_decoder_2_n = NiN(256, 128, wn)
        self.shape_decoder_2_o = _res(256, 128, dp, wn)
        self.shape_decoder_2_p = UpSample(128, 128, wn, um)
        self.shape_decoder_3 = UpBlock(256, 128, 128, um, dp, wn)
        self.shape_decoder_4 = UpBlock(256, 128, 64, um, dp, wn)
        self.shape_decoder_5 = UpBlock(128, 64, 32, um, dp, wn)
        if self.vunet_256:
            self.shape_decoder_5_a = UpBlock(64, 32, 32, um, dp, wn)
        self.shape_decoder_6 = EndBlock(64, 32, 3, dp, wn)

    # ------------------------------------------------------------------ packing
    _TWO_SOURCE = {"shape_decoder_1_n.layers.1": (128, 128), "shape_decoder_2_n.layers.1": (128, 128),
                   "app_decoder_1_c": (128, 128)}

    def _build_plans(self, device) -> dict:
        parents = {}
        for pname, pm in self.named_modules():
            if isinstance(pm, Residual):
                parents[pname + ".layers.2"] = pm
        P = {}
        for name, m in self.named_modules():
            if not isinstance(m, MyConv2d):
                continue
            h = m.conv
            w = pack.fold_weight_norm(h.weight_v, h.weight_g) if isinstance(h, WNConvP) else h.weight
            split = self._TWO_SOURCE.get(name)
            r = parents.get(name)
            if r is not None and r.c_in != r.c_out:
                split = (r.c_out, r.c_in - r.c_out)
            if name == "shape_decoder_6.conv" and w.shape[0] * w.shape[3] <= 32 and split is None and m.stride == 1:
                # the 3x3 -> 3 channel head at full resolution: 3x1 implicit GEMM with 9 columns + horizontal gather-sum
                # (a third of the matrix work of the 32-column-padded 3x3, and the NCHW result is written by a
                # streaming kernel instead of 4-byte scattered epilogue stores)
                P[name] = pack.pack_conv_rowsplit(w, h.bias, pad=m.padding).to(device)
                continue
            P[name] = pack.pack_conv(w, h.bias, c_split=split, stride=m.stride, pad=m.padding).to(device)
        return P

    # ------------------------------------------------------------------ fused building blocks
    def _residual(self, name: str, x, skip=None, out=None):
        return ops.conv(self._plans[name + ".layers.2"], x, skip, pre_op=L.PRE_ELU, res0=x, out=out)

    def _nin(self, name: str, x, x1=None, out=None):
        return ops.conv(self._plans[name + ".layers.1"], x, x1, pre_op=L.PRE_ELU, out=out)

    def _upsample(self, name: str, x):
        return ops.conv(self._plans[name + ".depth4x"], x, store=L.STORE_D2S)

    # ---- "fusion by cache blocking" of the high-resolution 32-channel blocks (round 4 experiment, FUSG_VU_SUBBATCH=n) ----
    # The verdict's fused Residual -> Residual block would keep the intermediates of the 256 x 256 / 128 x 128 levels out of
    # HBM.  The same traffic saving without a new kernel: run those levels depth-first over sub-batches of n images, so that
    # every intermediate (67 MB per tensor at n = 8, 32 channels, 256 x 256) is consumed out of the 256 MiB Infinity Cache
    # right after it was written, and the sub-batch temporaries are the same allocator blocks every time.  Same launches on
    # slices: the results are bit-identical.  Measured: profiles/r04_ab_experiments.txt.
    @staticmethod
    def _subbatch(b: int) -> int:
        import os
        n = int(os.environ.get("FUSG_VU_SUBBATCH", "0"))
        return n if 0 < n < b else 0

    def _shape_encoder_top(self, x, n):
        """forward_dec_up's first two levels (InitBlock at full resolution, DownBlock 1_a at half) over sub-batches of n."""
        b, _, h, w = x.shape
        dev = x.device
        sb, sc = ops.nhwc_empty(b, 32, h, w, dev), ops.nhwc_empty(b, 32, h, w, dev)
        ab, ac = ops.nhwc_empty(b, 32, h // 2, w // 2, dev), ops.nhwc_empty(b, 32, h // 2, w // 2, dev)
        a1 = ops.nhwc_empty(b, 32, h // 2, w // 2, dev)
        for lo in range(0, b, n):
            hi = min(b, lo + n)
            t = self._nin("shape_encoder_1.nin", x[lo:hi])
            s0 = self._residual("shape_encoder_1.residual_0", t)
            s1 = self._residual("shape_encoder_1.residual_1", s0)
            self._nin("shape_skip_1_b", s0, out=sb[lo:hi])
            self._nin("shape_skip_1_c", s1, out=sc[lo:hi])
            d = ops.conv(self._plans["shape_encoder_1_a.down.down"], s1)
            a0 = self._residual("shape_encoder_1_a.residual_0", d)
            self._residual("shape_encoder_1_a.residual_1", a0, out=a1[lo:hi])
            self._nin("shape_skip_1_a_b", a0, out=ab[lo:hi])
            self._nin("shape_skip_1_a_c", a1[lo:hi], out=ac[lo:hi])
        return a1, [sb, sc, ab, ac]

    def _shape_decoder_tail(self, x, skips, n):
        """forward_dec_down's last two blocks (UpBlock 5_a at half resolution, EndBlock at full) over sub-batches of n;
        `skips` = [skip_a, skip_b] of 5_a, then of the EndBlock (full-batch tensors)."""
        b, _, h, w = x.shape
        head = self._plans["shape_decoder_6.conv"]
        out = torch.empty((b, 3, 2 * h, 2 * w), device=x.device, dtype=torch.float32)
        for lo in range(0, b, n):
            hi = min(b, lo + n)
            r = self._residual("shape_decoder_5_a.residual_0", x[lo:hi], skips[0][lo:hi])
            r = self._residual("shape_decoder_5_a.residual_1", r, skips[1][lo:hi])
            u = self._upsample("shape_decoder_5_a.up", r)
            e = self._residual("shape_decoder_6.residual_0", u, skips[2][lo:hi])
            e = self._residual("shape_decoder_6.residual_1", e, skips[3][lo:hi])
            o = ops.conv_rowsplit(head, e, nchw_out=True) if head.rowsplit is not None else ops.conv(head, e, nchw_out=True)
            ops.copy4d(o, out[lo:hi], 0)                      # (3 channels: a 6 MB copy per sub-batch)
        return out

    def set_vehicle_seeds(self, seeds=None):
        """Optional, beyond the reference: give every sample of the next passes its own noise stream
        (`torch.Generator` seeded per vehicle, consumed in the reference's sampler order) instead of the
        shared global CPU generator.  A vehicle's noise then does not depend on which other vehicles share
        its batch, so a clip sharded over N GPUs renders the same images as on one (noise
        caveat).  `None` restores the reference behaviour (global generator, batch-shaped draws)."""
        if seeds is None:
            self.__dict__.pop("_vehicle_gens", None)
            return
        gens = []
        for sd in seeds:
            g = torch.Generator(device="cpu")
            g.manual_seed(int(sd))
            gens.append(g)
        self.__dict__["_vehicle_gens"] = gens

    def _rng_snapshot(self):
        """State of the CPU generators the samplers draw from (nn_base.entry_point rewinds them before a repeat)."""
        gens = self.__dict__.get("_vehicle_gens")
        return (torch.get_rng_state(), None if gens is None else [g.get_state() for g in gens])

    def _rng_restore(self, snap) -> None:
        torch.set_rng_state(snap[0])
        gens = self.__dict__.get("_vehicle_gens")
        if gens is not None and snap[1] is not None:
            for g, st in zip(gens, snap[1]):
                g.set_state(st)

    @staticmethod
    def _fill_noise(buf, shapes, gens=None):
        """Writes the N(0,1) draws for `shapes` (in order) into the flat host buffer `buf`; returns
        [(offset, numel, shape)].  gens=None: `torch.randn(*shape)` on the global CPU generator, exactly the
        reference's draw (layers.py:166).  Else one generator per sample: sample b of every shape comes from
        gens[b], in shape order."""
        off, views = 0, []
        for s in shapes:
            n = int(torch.Size(s).numel())
            v = buf[off:off + n].view(*s)
            if gens is None:
                torch.randn(*s, out=v)                              # == torch.randn(*s): same stream, same values
            else:
                if len(gens) != s[0]:
                    raise ValueError(f"set_vehicle_seeds: {len(gens)} seeds for a batch of {s[0]}")
                for b, g in enumerate(gens):
                    torch.randn(*s[1:], out=v[b], generator=g)
            views.append((off, n, s))
            off += n
        return views

    def _draw_noise(self, shapes, device):
        """All Sampler draws of one entry point, up front: same CPU default generator, same shapes,
        same order as the reference's `torch.randn(*mu.size())` calls (layers.py:166), but written
        into one pinned staging buffer and shipped with ONE asynchronous H2D copy instead of ten
        synchronous ones (each of which would stall the launch queue).  A small ring of staging
        buffers, guarded by events, keeps a buffer alive until its copy has executed."""
        total = sum(int(torch.Size(s).numel()) for s in shapes)
        if ops.RECORDER is not None:
            # recorded pass: this call site gets its own pinned source and device destination for the life of the
            # plan; the copy becomes a plan operation and the host refills the source before every replay
            ring = torch.empty((ops.RECORDER.NSLOTS, total), dtype=torch.float32, pin_memory=True)
            views = self._fill_noise(ring[0], shapes, self.__dict__.get("_vehicle_gens"))
            dev_buf = torch.empty(total, dtype=torch.float32, device=device)
            ops.RECORDER.h2d(dev_buf, ring)
            ops.RECORDER.noise_slots.append((ring, [tuple(s) for s in shapes]))
            ops.RECORDER.keep.append(dev_buf)
            return [dev_buf[o:o + n].view(*s) for o, n, s in views]
        ring = self.__dict__.setdefault("_noise_ring", [])
        slot = self.__dict__.get("_noise_slot", 0)
        self.__dict__["_noise_slot"] = (slot + 1) % 4
        while len(ring) < 4:
            ring.append([None, None])
        buf, ev = ring[slot]
        if ev is not None:
            ev.synchronize()                                        # copy issued 4 calls ago: long done
        if buf is None or buf.numel() < total:
            buf = torch.empty(max(total, 1 << 16), dtype=torch.float32, pin_memory=True)
        views = self._fill_noise(buf, shapes, self.__dict__.get("_vehicle_gens"))
        dev_buf = buf[:total].to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        ring[slot] = [buf, ev]
        return [dev_buf[o:o + n].view(*s) for o, n, s in views]

    def _sampler(self, name: str, x, noise, mu_out=None, c_off=0, z_out=None):
        """Sampler (layers.py:163-167).  Returns (mu, z); with mu_out/z_out the results go to the
        channel slice [c_off, c_off+128) of those buffers.  `noise` = this sampler's pre-drawn N(0,1)."""
        p = self._plans[name + ".conv"]
        if mu_out is None:
            mu = ops.conv(p, x)
        else:
            ops.conv(p, x, out=mu_out, out_c_off=c_off)
            mu = mu_out[:, c_off:c_off + p.cout]
        assert tuple(noise.shape) == tuple(mu.shape), (noise.shape, mu.shape)
        z = ops.add4d(mu, noise, None if z_out is None else z_out[:, c_off:c_off + p.cout])
        return mu, z

    def _init_block(self, name, x):
        x = self._nin(name + ".nin", x)
        s0 = x = self._residual(name + ".residual_0", x)
        s1 = x = self._residual(name + ".residual_1", x)
        return x, [s0, s1]

    def _down_block(self, name, x):
        x = ops.conv(self._plans[name + ".down.down"], x)
        s0 = x = self._residual(name + ".residual_0", x)
        s1 = x = self._residual(name + ".residual_1", x)
        return x, [s0, s1]

    def _up_block(self, name, x, skip_a, skip_b):
        x = self._residual(name + ".residual_0", x, skip_a)
        x = self._residual(name + ".residual_1", x, skip_b)
        return self._upsample(name + ".up", x)

    def _ar_block(self, name, x, skip_a, enc_down_mu, noise):
        """AutoRegressiveBlock.forward (reference models.py:56-86); `noise` = its four sampler draws."""
        x = self._residual(name + ".residual_init", x, skip_a)
        x_ = ops.space_to_depth2(self._residual(name + ".residual_s2d", x))
        g = None
        if enc_down_mu is not None:
            gs = ops.space_to_depth2(ops.as_nhwc(enc_down_mu))
            g = [self._nin(f"{name}.nin_{k}", gs[:, 128 * k:128 * (k + 1)]) for k in range(3)]
        b, _, h, w = x_.shape
        mus = ops.nhwc_empty(b, 512, h, w, x_.device)
        zs = ops.nhwc_empty(b, 512, h, w, x_.device)
        for k in range(4):
            _, z_k = self._sampler(f"{name}.sampler_{k}", x_, noise[k], mus, 128 * k, zs)
            if k < 3:
                cond = g[k] if g is not None else self._nin(f"{name}.nin_{k}", z_k)
                x_ = self._residual(f"{name}.residual_{k}", x_, cond)
        return x, ops.depth_to_space2(mus), ops.depth_to_space2(zs)

    # ------------------------------------------------------------------ reference entry points
    @entry_point
    def forward_enc_up(self, x):
        self._ensure(x)
        x = ops.as_nhwc(x)
        x, _ = self._init_block("app_encoder_1", x)
        names = ["app_encoder_1_a", "app_encoder_1_b"] + (["app_encoder_1_c"] if self.vunet_256 else []) + \
                ["app_encoder_2", "app_encoder_3"]
        for n in names:
            x, _ = self._down_block(n, x)
        skips = [self._nin("app_skip_3_c", x)]
        x, sl = self._down_block("app_encoder_4", x)
        outputs = [sl[-2], x]
        skips.append(self._nin("app_skip_4_c", x))
        return outputs, skips

    @entry_point
    def forward_dec_up(self, x):
        self._ensure(x)
        x = ops.as_nhwc(x)
        skips: List[torch.Tensor] = []
        nsub = self._subbatch(x.shape[0]) if self.vunet_256 else 0
        if nsub:
            x, skips = self._shape_encoder_top(x, nsub)
        else:
            x, sl = self._init_block("shape_encoder_1", x)
            skips += [self._nin("shape_skip_1_b", sl[-2]), self._nin("shape_skip_1_c", sl[-1])]
            if self.vunet_256:
                x, sl = self._down_block("shape_encoder_1_a", x)
                skips += [self._nin("shape_skip_1_a_b", sl[-2]), self._nin("shape_skip_1_a_c", sl[-1])]
        for i in range(2, 7):
            x, sl = self._down_block(f"shape_encoder_{i}", x)
            skips += [self._nin(f"shape_skip_{i}_b", sl[-2]), self._nin(f"shape_skip_{i}_c", sl[-1])]
        return [x], skips

    @entry_point
    def forward_enc_down(self, enc_up_outputs: Sequence[torch.Tensor], skips: Sequence[torch.Tensor]):
        self._ensure(enc_up_outputs[-1])
        P = self._plans
        o1, o2 = ops.as_nhwc(enc_up_outputs[-1]), ops.as_nhwc(enc_up_outputs[-2])
        b, _, h, w = o1.shape
        noise = self._draw_noise([(b, 128, h, w), (b, 128, 2 * h, 2 * w)], o1.device)    # 1_b, then 2_b
        x = ops.conv(P["app_bottleneck"], o1)
        x = self._residual("app_decoder_1_a", x, ops.as_nhwc(skips[-1]))
        mu_0, z_0 = self._sampler("app_decoder_1_b", x, noise[0])
        x_ = ops.conv(P["app_decoder_1_c"], o2, z_0)
        x = self._residual("app_decoder_1_d", x, x_)
        x = self._upsample("app_decoder_1_e", x)
        x = self._residual("app_decoder_2_a", x, None)
        mu_1, z_1 = self._sampler("app_decoder_2_b", x, noise[1])
        return [mu_0, mu_1], [z_0, z_1]

    @entry_point
    def forward_dec_down(self, dec_up_outputs, skips: List[torch.Tensor], enc_down_mu=()):
        self._ensure(dec_up_outputs[-1])
        P = self._plans
        mu, z = [], []
        x0 = ops.as_nhwc(dec_up_outputs[-1])
        b, _, h, w = x0.shape
        # reference draw order: block 1 samplers 0..3 at half the bottleneck resolution, then block 2's
        noise = self._draw_noise([(b, 128, h // 2, w // 2)] * 4 + [(b, 128, h, w)] * 4, x0.device)
        x = ops.conv(P["shape_bottleneck"], x0)
        for blk in (1, 2):
            skip_a = ops.as_nhwc(skips.pop())
            skip_b = ops.as_nhwc(skips.pop())
            m = None if len(enc_down_mu) == 0 else enc_down_mu[blk - 1]
            x, mu_k, z_k = self._ar_block(f"shape_decoder_{blk}", x, skip_a, m, noise[4 * (blk - 1):4 * blk])
            mu.append(mu_k)
            z.append(z_k)
            x = self._nin(f"shape_decoder_{blk}_n", x, z_k)
            x = self._residual(f"shape_decoder_{blk}_o", x, skip_b)
            x = self._upsample(f"shape_decoder_{blk}_p", x)
        nsub = self._subbatch(x0.shape[0]) if self.vunet_256 else 0
        names = ["shape_decoder_3", "shape_decoder_4", "shape_decoder_5"] + \
                (["shape_decoder_5_a"] if (self.vunet_256 and not nsub) else [])
        for n in names:
            skip_a = ops.as_nhwc(skips.pop())
            skip_b = ops.as_nhwc(skips.pop())
            x = self._up_block(n, x, skip_a, skip_b)
        if nsub:
            tail = [ops.as_nhwc(skips.pop()) for _ in range(4)]
            assert not skips
            return self._shape_decoder_tail(x, tail, nsub), mu, z
        skip_a = ops.as_nhwc(skips.pop())
        skip_b = ops.as_nhwc(skips.pop())
        x = self._residual("shape_decoder_6.residual_0", x, skip_a)
        x = self._residual("shape_decoder_6.residual_1", x, skip_b)
        head = P["shape_decoder_6.conv"]
        x = ops.conv_rowsplit(head, x, nchw_out=True) if head.rowsplit is not None else ops.conv(head, x, nchw_out=True)
        assert not skips
        return x, mu, z

    @entry_point
    def forward(self, y_tilde, x=None, mean_mode="mean_appearance"):
        want = 256 if self.vunet_256 else 128
        assert y_tilde.shape[-1] == want
        if x is not None:
            assert x.shape[-1] == want
        assert mean_mode in ["mean_appearance", "mean_shape"]
        if mean_mode == "mean_appearance":
            output_enc_up, skips_enc_up = self.forward_enc_up(x)
            mu_app, z_app = self.forward_enc_down(output_enc_up, skips_enc_up)
            output_dec_up, skips_dec_up = self.forward_dec_up(y_tilde)
            x_tilde, mu_shape, z_shape = self.forward_dec_down(output_dec_up, skips_dec_up, z_app)
            return x_tilde, mu_app, mu_shape
        output_dec_up, skips_dec_up = self.forward_dec_up(y_tilde)
        x_tilde, mu_shape, z_shape = self.forward_dec_down(output_dec_up, skips_dec_up)
        return x_tilde
