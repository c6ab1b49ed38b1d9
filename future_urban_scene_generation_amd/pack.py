"""Load-time weight pre-packing for the fused implicit-GEMM convolution (csrc/conv_igemm.hip).

Everything here is host-side, one-off work done when a module's parameters change
(``load_state_dict`` / ``.to(device)``): folding of the re-parametrisations that are pure functions
of the parameters (weight_norm, spectral_norm, conv-following eval BatchNorm), re-layout of the
filter into the kernel's ``[cout_pad][k_pad]`` K-contiguous panel, and construction of the k-table
that drives the on-the-fly im2col gather (see include/fusg.h, ``fusg_conv_desc``).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List,  Optional, Sequence, Tuple

import numpy as np
import torch

BK = 32


def _ru(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass
class ConvPlan:
    """Packed parameters + static geometry of one convolution call site (host copies; `.dev`
    holds the uploaded device tensors)."""
    wpack: torch.Tensor            # [nphase, cout_pad, k_pad] f32
    bias: torch.Tensor             # [cout_pad] f32
    ktab: torch.Tensor             # [nphase, k_pad // 4, 2] int32
    cout: int
    cout_pad: int
    k_pad: int
    c_split: Tuple[int, ...]       # logical concat channel counts (C0,) or (C0, C1)
    c0k: int                       # K-channels taken from src0 (multiple of 4)
    c1k: int
    kh: int
    kw: int
    stride: int = 1
    pad: int = 0
    dil: int = 1
    pad_mode: int = 0              # 0 zero, 1 reflect
    upsample: int = 0
    nphase: int = 1                # 4 = ConvTranspose2d(k4, s2, p1)
    flops_per_pixel: float = 0.0   # algorithmic 2*MAC per q-space output pixel (all phases: per input pixel)
    pad_w: int = -1                # horizontal padding when it differs from `pad` (row-split heads: 0)
    rowsplit: Optional[dict] = None
    dev: dict = field(default_factory=dict)

    def to(self, device) -> "ConvPlan":
        key = str(device)
        if self.dev.get("key") != key:
            self.dev = {"key": key,
                        "wpack": self.wpack.to(device).contiguous(),
                        "bias": self.bias.to(device).contiguous(),
                        "ktab": self.ktab.to(device).contiguous()}
            if self.rowsplit is not None:
                self.dev["rs_bias"] = self.rowsplit["bias"].to(device).contiguous()
            wsplit, wscale = split_f16x3(self.wpack)
            self.dev["wpack_h"] = wsplit.to(device).contiguous()
            self.dev["wscale"] = wscale.to(device).contiguous()
            frag = frag_f16x3(wsplit, self) if self.nphase == 1 else None
            if frag is not None and self.s2d_ok():
                frag = frag[s2d_tap_order(self.kh)].contiguous()       # slabs in (parity quadrant, local tap) order
            if frag is None and self.tapunit_ok():
                frag = frag_tapunit(wsplit, self)                      # few-channel k x k layer: per-k-step order
            self.dev["wfrag"] = None if frag is None else frag.to(device).contiguous()
            fb = frag_bf16(self.wpack, self) if self.nphase == 1 else None
            if fb is not None and self.s2d_ok():
                fb = fb[s2d_tap_order(self.kh)].contiguous()
            if fb is None and frag is not None and self.tapunit_ok():
                fb = frag_tapunit_bf16(self.wpack, self)               # the stem in single-pass bf16 (tap-unit kernel, MODE 1)
            self.dev["wfrag_bf16"] = None if fb is None else fb.to(device).contiguous()
            self.dev["wfrag_order"] = 1 if self.s2d_ok() else (2 if self.tapunit_ok() else 0)   # fusg_conv_desc.wfrag_order
        return self

    def frag_f32_dev(self):
        """fusg_conv_desc.wfrag_f32 (exact-fp32 halo kernel), built and uploaded on first use: only `precision="f32"` passes
        and the range guard's fallback need it.  None when the layer cannot use the halo kernel."""
        if "wfrag_f32" not in self.dev:
            ff = None
            if self.nphase == 1 and self.dev.get("wfrag") is not None:
                if self.dev["wfrag_order"] in (0, 1):
                    ff = frag_f32(self.wpack, self)
                elif self.dev["wfrag_order"] == 2:
                    ff = frag_tapunit_f32(self.wpack, self)                 # few-channel stems: units of 4 channels
            if ff is not None and self.s2d_ok():
                ff = ff[s2d_tap_order(self.kh)].contiguous()
            self.dev["wfrag_f32"] = None if ff is None else ff.to(self.dev["wpack"].device).contiguous()
        return self.dev["wfrag_f32"]

    def s2d_ok(self) -> bool:
        """Stride-2 k3/k4 pad-1 layer whose halo-kernel weights are stored in parity-quadrant order
        (fusg_conv_desc.wfrag_order = 1, see s2d_tap_order)."""
        import os
        if os.environ.get("FUSG_NO_S2D"):
            return False
        return (self.nphase == 1 and self.stride == 2 and self.kh == self.kw and self.kh in (3, 4) and self.pad == 1
                and self.dil == 1 and self.upsample == 0 and self.pad_w < 0 and self.c1k == 0 and self.c0k % 32 == 0
                and self.c0k > 0)

    def tapunit_ok(self) -> bool:
        """Few-channel k x k layer (the 7x7 stems) whose weights are stored for the tap-unit kernel
        (fusg_conv_desc.wfrag_order = 2, csrc/conv_kernel_tapunit.h)."""
        import os
        taps = self.kh * self.kw
        unit = 8 if self.c0k % 8 == 0 else 4
        return (not os.environ.get("FUSG_NO_TAPUNIT") and self.nphase == 1 and (taps >= 9 or self.c0k <= 8) and self.c1k == 0
                and 4 <= self.c0k <= 24 and self.dil == 1 and self.upsample == 0 and self.stride in (1, 2)
                and self.k_pad >= taps * self.c0k and taps * (self.c0k // unit) <= 160 and self.rowsplit is None)

    def out_hw(self, h: int, w: int) -> Tuple[int, int]:
        """q-space output grid for an input of h x w."""
        if self.nphase == 4:
            return h, w
        hv, wv = h << self.upsample, w << self.upsample
        pw = self.pad if self.pad_w < 0 else self.pad_w
        return ((hv + 2 * self.pad - self.dil * (self.kh - 1) - 1) // self.stride + 1,
                (wv + 2 * pw - self.dil * (self.kw - 1) - 1) // self.stride + 1)


def split_f16x3(wpack: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """[nphase, cout_pad, k_pad] f32 -> ([nphase, 2, cout_pad, k_pad] f16 = (hi, lo) of w * s[n], 1 / s [cout_pad] f32):
    the weight half of the split-precision contraction (csrc/conv_kernel_h3.h).  s[n] is the power of two that puts
    the largest |w| of output channel n (over all phases) in [2^13, 2^14): hi = RTN(w s) then has 11 significant bits
    for every weight down to 2^-27 of the channel's largest, lo = RTN(w s - hi) is stored unscaled (its absolute floor,
    fp16's 2^-24, is 2^-37 of the largest weight) and hi * 2^-11 - formed in registers for the third product - stays
    exact down to 2^-17 of it.  The scale is exact to undo: the conv epilogue multiplies the accumulator by 1 / s[n]."""
    w = wpack.to(torch.float64)
    rowmax = w.abs().amax(dim=(0, 2))
    ok = torch.isfinite(rowmax) & (rowmax > 0)
    _, e = torch.frexp(torch.where(ok, rowmax, torch.ones_like(rowmax)))       # rowmax = m * 2^e, m in [0.5, 1)
    sh = torch.where(ok, 14 - e, torch.zeros_like(e)).clamp(-100, 100)
    s = torch.pow(torch.tensor(2.0, dtype=torch.float64), sh.to(torch.float64))
    ws = w * s[None, :, None]
    hi = ws.to(torch.float32).to(torch.float16)
    lo = (ws - hi.to(torch.float64)).to(torch.float32).to(torch.float16)
    return torch.stack([hi, lo], dim=1), (1.0 / s).to(torch.float32)


def frag_f16x3(wsplit: torch.Tensor, plan: "ConvPlan") -> Optional[torch.Tensor]:
    """MFMA-fragment order of the split weights for the halo kernel (csrc/conv_kernel_halo.h):
    [tap][chunk32][cout_pad/32][16-column half][hi|lo][lane 64][8 halves], lane = (k >> 3) * 16 + column, i.e. exactly
    what lane l of a wave feeds v_mfma_f32_16x16x32_f16 as B[k = 8*(l>>4) + j][col = l&15]: one fragment per
    (32 k, 16 columns).  Returns None when the layer cannot use that kernel (channels not a multiple of 32 per source)."""
    taps = plan.kh * plan.kw
    ctot = plan.c0k + plan.c1k
    if plan.nphase != 1 or taps < 1 or plan.c0k % 32 or plan.c1k % 32 or ctot == 0 or plan.k_pad != taps * ctot:
        return None
    w = wsplit[0]                                               # [2 (hi, lo), cout_pad, k_pad]
    nt32, nch = plan.cout_pad // 32, ctot // 32
    w = w.view(2, nt32, 2, 16, taps, nch, 4, 8)                 # hl, nt, ct, r, tap, chunk, g, j
    w = w.permute(4, 5, 1, 2, 0, 6, 3, 7)                       # tap, chunk, nt, ct, hl, g, r, j
    return w.reshape(taps, nch, nt32, 2, 2, 64, 8).contiguous()


def frag_bf16(wpack: torch.Tensor, plan: "ConvPlan") -> Optional[torch.Tensor]:
    """The weights rounded to bf16 (ties to even) in the halo kernel's fragment order for the single-pass bf16 mode
    (fusg_conv_desc.wfrag_bf16): [tap][chunk32][cout_pad/32][16-column half][lane 64][8], lane = (k >> 3) * 16 + column.
    None when the layer cannot use the halo kernel."""
    taps = plan.kh * plan.kw
    ctot = plan.c0k + plan.c1k
    if plan.nphase != 1 or taps < 1 or plan.c0k % 32 or plan.c1k % 32 or ctot == 0 or plan.k_pad != taps * ctot:
        return None
    w = wpack[0].to(torch.bfloat16)                             # [cout_pad, k_pad]
    nt32, nch = plan.cout_pad // 32, ctot // 32
    w = w.view(nt32, 2, 16, taps, nch, 4, 8)                    # nt, ct, r, tap, chunk, g, j
    w = w.permute(3, 4, 0, 1, 5, 2, 6)                          # tap, chunk, nt, ct, g, r, j
    return w.reshape(taps, nch, nt32, 2, 64, 8).contiguous()


def frag_f32(wpack: torch.Tensor, plan: "ConvPlan") -> Optional[torch.Tensor]:
    """The fp32 weights in the halo kernel's fragment order for v_mfma_f32_16x16x4_f32 (fusg_conv_desc.wfrag_f32):
    [tap][chunk32][cout_pad/32][16-column half][h][lane 64][4 floats], lane = g * 16 + column holding
    w[column][32 chunk + 16 h + 4 g + e], e = 0..3 - what lane (column, g) feeds the instruction (h, e) of a chunk as B.
    None when the layer cannot use the halo kernel."""
    taps = plan.kh * plan.kw
    ctot = plan.c0k + plan.c1k
    if plan.nphase != 1 or taps < 1 or plan.c0k % 32 or plan.c1k % 32 or ctot == 0 or plan.k_pad != taps * ctot:
        return None
    w = wpack[0]                                                # [cout_pad, k_pad]
    nt32, nch = plan.cout_pad // 32, ctot // 32
    w = w.view(nt32, 2, 16, taps, nch, 2, 4, 4)                 # nt, ct, r, tap, chunk, h, g, e
    w = w.permute(3, 4, 0, 1, 5, 6, 2, 7)                       # tap, chunk, nt, ct, h, g, r, e
    return w.reshape(taps, nch, nt32, 2, 2, 64, 4).contiguous()


def s2d_quadrant_taps(k: int):
    """A stride-2, pad-1 convolution reads input row 2Y - 1 + ky: parity i = (ky - 1) mod 2 of the rows, sub-row
    Y + (ky - 1) // 2.  So it is, per parity quadrant (i, j) of the input (x[2Y+i, 2X+j]), a small STRIDE-1
    convolution of that quarter-size sub-image.  Returns, for q = 2*i + j, the list of (ky, kx, dY, dX)."""
    per_axis = {0: [], 1: []}
    for kk in range(k):
        per_axis[(kk - 1) % 2].append((kk, (kk - 1) // 2))
    return [[(ky, kx, dy, dx) for ky, dy in per_axis[q >> 1] for kx, dx in per_axis[q & 1]] for q in range(4)]


def s2d_tap_order(k: int) -> List[int]:
    """Order of the k*k taps (row-major index ky*k + kx) in the parity-quadrant weight layout."""
    return [ky * k + kx for quad in s2d_quadrant_taps(k) for ky, kx, _, _ in quad]


def frag_tapunit(wsplit: torch.Tensor, plan: "ConvPlan") -> torch.Tensor:
    """Weights of a few-channel layer in the k-step order of the tap-unit kernel: K index = (tap, unit of `unit`
    channels), 16 / unit units per MFMA k-step -> [step][cout_pad/32][hi|lo][64 lanes][8 halves] with
    lane = (k >> 3 & 1) * 32 + column; k-values past the last unit are zero."""
    w = wsplit[0]                                               # [2, cout_pad, k_pad], K order (tap, channel) with c0k per tap
    taps, c = plan.kh * plan.kw, plan.c0k
    k = taps * c
    nsteps = (k + 15) // 16
    wk = torch.zeros(2, plan.cout_pad, nsteps * 16, dtype=w.dtype)
    wk[:, :, :k] = w[:, :, :k]
    nt32 = plan.cout_pad // 32
    wk = wk.view(2, nt32, 32, nsteps, 2, 8)                     # hl, nt, r, step, h, j
    wk = wk.permute(3, 1, 0, 4, 2, 5)                           # step, nt, hl, h, r, j
    return wk.reshape(nsteps, nt32, 2, 64, 8).contiguous()


def frag_tapunit_bf16(wpack: torch.Tensor, plan: "ConvPlan") -> torch.Tensor:
    """bf16 weights (ties to even) of a few-channel layer in the k-step order of the tap-unit kernel's bf16 mode: `frag_tapunit`
    without the (hi | lo) axis -> [step][cout_pad/32][64 lanes][8], lane = (k >> 3 & 1) * 32 + column."""
    w = wpack[0]                                                # [cout_pad, k_pad], K order (tap, channel) with c0k per tap
    taps, c = plan.kh * plan.kw, plan.c0k
    k = taps * c
    nsteps = (k + 15) // 16
    wk = torch.zeros(plan.cout_pad, nsteps * 16, dtype=torch.float32)
    wk[:, :k] = w[:, :k]
    nt32 = plan.cout_pad // 32
    wk = wk.to(torch.bfloat16).view(nt32, 32, nsteps, 2, 8)     # nt, r, step, h, j
    wk = wk.permute(2, 0, 3, 1, 4)                              # step, nt, h, r, j
    return wk.reshape(nsteps, nt32, 64, 8).contiguous()


def frag_tapunit_f32(wpack: torch.Tensor, plan: "ConvPlan") -> torch.Tensor:
    """fp32 weights of a few-channel layer for the exact-fp32 tap-unit kernel (csrc/conv_kernel_tapunit_f32.h): K walked in units of
    4 channels of one tap -> [unit][cout_pad/32][64 lanes][2] with lane = g * 32 + column holding w[column][4 unit + g] and
    w[column][4 unit + 2 + g]: what lane half g feeds the unit's two v_mfma_f32_32x32x2_f32."""
    w = wpack[0]                                                # [cout_pad, k_pad], K order (tap, channel) with c0k per tap
    taps, c = plan.kh * plan.kw, plan.c0k
    assert c % 4 == 0
    nunits = taps * c // 4
    nt32 = plan.cout_pad // 32
    wk = w[:, :nunits * 4].reshape(nt32, 32, nunits, 2, 2)      # nt, col, unit, e, g
    wk = wk.permute(2, 0, 4, 1, 3)                              # unit, nt, g, col, e
    return wk.reshape(nunits, nt32, 64, 2).contiguous()


def _entry(dy: int, dx: int, coff: int, src: int, invalid: bool = False):
    x = (dy & 0xFFFF) | ((dx & 0xFFFF) << 16)
    y = (coff & 0x3FFFFFFF) | (src << 30) | ((1 << 31) if invalid else 0)
    # to signed int32
    x = x - (1 << 32) if x >= (1 << 31) else x
    y = y - (1 << 32) if y >= (1 << 31) else y
    return x, y


def _pack_panel(w: torch.Tensor, taps: Sequence[Tuple[int, int, int, int]], c_split: Sequence[int], cin_pad: int = 4):
    """w: [cout, cin, kh, kw] (correlation form).  taps: list of (ky, kx, dy, dx).
    Returns (panel [cout_pad, k_pad], ktab [k_pad/4, 2]) for K order (tap, concat channel)."""
    cout, cin = w.shape[0], w.shape[1]
    assert sum(c_split) == cin, (c_split, cin)
    c0 = c_split[0]
    c1 = c_split[1] if len(c_split) > 1 else 0
    c0k, c1k = _ru(c0, cin_pad), _ru(c1, cin_pad)
    ctot = c0k + c1k
    k = len(taps) * ctot
    k_pad = _ru(k, BK)
    cout_pad = _ru(cout, 32)
    panel = torch.zeros(cout_pad, k_pad, dtype=torch.float32)
    tab = np.zeros((k_pad // 4, 2), dtype=np.int64)
    for t, (ky, kx, dy, dx) in enumerate(taps):
        base = t * ctot
        panel[:cout, base:base + c0] = w[:, :c0, ky, kx]
        if c1:
            panel[:cout, base + c0k:base + c0k + c1] = w[:, c0:, ky, kx]
        for q in range(ctot // 4):
            cc = q * 4
            src, coff = (0, cc) if cc < c0k else (1, cc - c0k)
            tab[(base + cc) // 4] = _entry(dy, dx, coff, src)
    for q in range(k // 4, k_pad // 4):
        tab[q] = _entry(0, 0, 0, 0, invalid=True)
    return panel, torch.from_numpy(tab.astype(np.int32)), c0k, c1k, k_pad, cout_pad


def pack_conv(weight: torch.Tensor, bias: Optional[torch.Tensor], *, c_split: Optional[Sequence[int]] = None,
              stride: int = 1, pad: int = 0, dil: int = 1, pad_mode: int = 0, upsample: int = 0,
              cin_pad: int = 4) -> ConvPlan:
    """nn.Conv2d-style filter [cout, cin, kh, kw] -> ConvPlan.  `cin_pad` = granularity the K-channels of
    each source are padded to (zero weights): 4 by default, 32 to make a small-Cin k x k layer eligible
    for the halo kernel (its source buffer must then have a channel pitch >= the padded count, with
    zeros in the padding: ops.as_nhwc(x, cpad=32))."""
    w = weight.detach().to("cpu", torch.float32)
    cout, cin, kh, kw = w.shape
    c_split = tuple(c_split) if c_split is not None else (cin,)
    taps = [(ky, kx, ky * dil - pad, kx * dil - pad) for ky in range(kh) for kx in range(kw)]
    panel, tab, c0k, c1k, k_pad, cout_pad = _pack_panel(w, taps, c_split, cin_pad)
    b = torch.zeros(cout_pad, dtype=torch.float32)
    if bias is not None:
        b[:cout] = bias.detach().to("cpu", torch.float32)
    return ConvPlan(wpack=panel[None].contiguous(), bias=b, ktab=tab[None].contiguous(), cout=cout, cout_pad=cout_pad,
                    k_pad=k_pad, c_split=c_split, c0k=c0k, c1k=c1k, kh=kh, kw=kw, stride=stride, pad=pad, dil=dil,
                    pad_mode=pad_mode, upsample=upsample, nphase=1, flops_per_pixel=2.0 * cout * cin * kh * kw)


UP2_GROUPS = (((0, 1), (2, 3), (4,)), ((0,), (1, 2), (3, 4)))     # per output parity: 5 taps -> low-res rows -1, 0, +1


def up2_phase_weights(weight: torch.Tensor) -> List[torch.Tensor]:
    """nn.Upsample(2, nearest) -> 5x5 conv (pad 2), seen from the low-resolution side.  Output pixel
    (2y+py, 2x+px) reads upsampled rows 2y+py-2 .. 2y+py+2, i.e. low-res rows y-1, y, y+1 with the 5 taps
    grouped as UP2_GROUPS[py]; the same along x.  So each of the 4 output parities is a 3x3 convolution of the
    LOW-res image whose weights are sums of the original taps: 9 MACs per output instead of 25.  Exact in real
    arithmetic (fp32 rounding differs like any re-association) wherever the 5-tap window stays inside the
    upsampled image; with the reference's ReflectionPad2d(2) (warp_learn/models.py:176-178) and edge-replicate
    padding on the low-res side it also holds on the second ring, and fails only on the outermost ring of
    output pixels (row/col 0 and 2H-1 / 2W-1), which the caller recomputes with the 25-tap form.
    Returns [w_00, w_01, w_10, w_11] (index 2*py+px), each [cout, cin, 3, 3]."""
    w = weight.detach().to("cpu", torch.float32)
    assert w.shape[2] == 5 and w.shape[3] == 5, w.shape
    out = []
    for py in range(2):
        for px in range(2):
            wp = torch.zeros(w.shape[0], w.shape[1], 3, 3, dtype=torch.float32)
            for a, gy in enumerate(UP2_GROUPS[py]):
                for b, gx in enumerate(UP2_GROUPS[px]):
                    # fixed summation order (ky, then kx ascending): the packed weights are reproducible
                    acc = torch.zeros(w.shape[0], w.shape[1], dtype=torch.float32)
                    for ky in gy:
                        for kx in gx:
                            acc = acc + w[:, :, ky, kx]
                    wp[:, :, a, b] = acc
            out.append(wp)
    return out


# The outermost ring of nn.Upsample(2) -> ReflectionPad2d(2) -> 5x5, also from the low-resolution side.  For an output row y the
# five taps ky read upsampled rows y - 2 .. y + 2 (reflected at the border), i.e. low-res rows Y - 1, Y, Y + 1 of Y = y >> 1:
#   "e" (y = 2Y, interior): Y-1, Y-1, Y, Y, Y+1      "o" (y = 2Y + 1, interior): Y-1, Y, Y, Y+1, Y+1        (= UP2_GROUPS)
#   "f" (y = 0):  u[2], u[1], u[0], u[1], u[2]  = low-res rows 1, 0, 0, 0, 1              -> offsets +1, 0, 0, 0, +1
#   "l" (y = 2H-1): u[2H-3], u[2H-2], u[2H-1], u[2H-2], u[2H-3] = rows H-2, H-1, H-1, H-1, H-2 -> offsets -1, 0, 0, 0, -1
# and the same along x: every ring pixel is a 3x3 convolution of the low-res image with regrouped weights (9 MACs instead of
# the 25 of the exact form the ring was recomputed with in rounds 1-2).
UP2_ROWMAP = {"e": (0, 0, 1, 1, 2), "o": (0, 1, 1, 2, 2), "f": (2, 1, 1, 1, 2), "l": (0, 1, 1, 1, 0)}


def up2_border_weights(weight: torch.Tensor, ry: str, rx: str) -> torch.Tensor:
    """[cout, cin, 3, 3] weights of the low-res 3x3 convolution that gives the output pixels of row kind `ry` and column
    kind `rx` (UP2_ROWMAP); sums in fixed (ky, kx ascending) order."""
    w = weight.detach().to("cpu", torch.float32)
    assert w.shape[2] == 5 and w.shape[3] == 5, w.shape
    wp = torch.zeros(w.shape[0], w.shape[1], 3, 3, dtype=torch.float32)
    for ky in range(5):
        for kx in range(5):
            a, b = UP2_ROWMAP[ry][ky], UP2_ROWMAP[rx][kx]
            wp[:, :, a, b] = wp[:, :, a, b] + w[:, :, ky, kx]
    return wp


def up2_ring_launches(h: int, w: int):
    """The twelve windows that tile the outermost ring of the 2h x 2w output: (row kind, column kind, low-res window
    (Y0, X0, nY, nX), output parity (py, px)); output pixel = (2 Y + py, 2 X + px)."""
    out = []
    for ry, Y0, py in (("f", 0, 0), ("l", h - 1, 1)):                  # top and bottom row
        out += [(ry, "e", (Y0, 1, 1, w - 1), (py, 0)), (ry, "o", (Y0, 0, 1, w - 1), (py, 1)),
                (ry, "f", (Y0, 0, 1, 1), (py, 0)), (ry, "l", (Y0, w - 1, 1, 1), (py, 1))]
    for rx, X0, px in (("f", 0, 0), ("l", w - 1, 1)):                  # left and right column without the corners
        out += [("e", rx, (1, X0, h - 1, 1), (0, px)), ("o", rx, (0, X0, h - 1, 1), (1, px))]
    return out


def pack_conv_up2_ring(weight: torch.Tensor, bias: Optional[torch.Tensor]) -> dict:
    """{(row kind, column kind): ConvPlan} for the twelve kind pairs of up2_ring_launches (edge-replicate padding 1: the
    offsets a border kind does not use carry zero weights)."""
    from . import _lib as L
    kinds = sorted({(ry, rx) for ry, rx, _, _ in up2_ring_launches(4, 4)})
    return {k: pack_conv(up2_border_weights(weight, *k), bias, stride=1, pad=1, pad_mode=L.PAD_REPLICATE) for k in kinds}


def pack_conv_up2_d2s(weight: torch.Tensor, bias: Optional[torch.Tensor]) -> ConvPlan:
    """The four phase convolutions of up2_phase_weights stacked along the output channels - phase 2*py+px in channel
    block [(2*py+px)*cout, +cout) - as ONE 3x3 conv (edge-replicate padding 1) whose DepthToSpace store
    (FUSG_STORE_D2S, the VUnet's DCR order) interleaves the phases: the low-res halo is staged once for all four
    phases and the column tiles are 128 wide even for cout = 64."""
    from . import _lib as L
    ws = torch.cat(up2_phase_weights(weight), dim=0)                       # [4*cout, cin, 3, 3]
    bs = None if bias is None else bias.detach().to("cpu", torch.float32).repeat(4)
    return pack_conv(ws, bs, stride=1, pad=1, pad_mode=L.PAD_REPLICATE)


def pack_conv_up2_phases(weight: torch.Tensor, bias: Optional[torch.Tensor]) -> List[ConvPlan]:
    """The four 3x3 phase convolutions of up2_phase_weights as ConvPlans (edge-replicate padding 1)."""
    from . import _lib as L
    return [pack_conv(wp, bias, stride=1, pad=1, pad_mode=L.PAD_REPLICATE) for wp in up2_phase_weights(weight)]


def pack_conv_transpose_k4s2p1(weight: torch.Tensor, bias: Optional[torch.Tensor]) -> ConvPlan:
    """nn.ConvTranspose2d(k=4, s=2, p=1) filter [cin, cout, 4, 4] -> four 2x2 phase convolutions.

    out[2q + py] gathers, per axis, ky in {1, 3} with input offsets {0, -1} (py = 0) or ky in {0, 2}
    with offsets {+1, 0} (py = 1): oy = 2*iy - 1 + ky."""
    w = weight.detach().to("cpu", torch.float32)
    cin, cout, kh, kw = w.shape
    assert kh == 4 and kw == 4
    wc = w.permute(1, 0, 2, 3).contiguous()          # [cout, cin, ky, kx], no flip needed with the mapping above
    axis = {0: [(1, 0), (3, -1)], 1: [(0, 1), (2, 0)]}
    panels, tabs = [], []
    for py in (0, 1):
        for px in (0, 1):
            taps = [(ky, kx, dy, dx) for (ky, dy) in axis[py] for (kx, dx) in axis[px]]
            panel, tab, c0k, c1k, k_pad, cout_pad = _pack_panel(wc, taps, (cin,))
            panels.append(panel)
            tabs.append(tab)
    b = torch.zeros(cout_pad, dtype=torch.float32)
    if bias is not None:
        b[:cout] = bias.detach().to("cpu", torch.float32)
    return ConvPlan(wpack=torch.stack(panels).contiguous(), bias=b, ktab=torch.stack(tabs).contiguous(), cout=cout,
                    cout_pad=cout_pad, k_pad=k_pad, c_split=(cin,), c0k=c0k, c1k=c1k, kh=4, kw=4, stride=1, pad=0, dil=1,
                    pad_mode=0, upsample=0, nphase=4, flops_per_pixel=2.0 * cout * cin * 16)


def pack_conv_transpose_k4s2p1_phases(weight: torch.Tensor, bias: Optional[torch.Tensor]) -> List[ConvPlan]:
    """The same four phase convolutions as pack_conv_transpose_k4s2p1, as separate dense 2x2 stride-1 ConvPlans
    (index 2*py + px) so that each can run on the halo kernel with strided stores (ops.conv_transpose_phases):
    along an axis, output parity 0 reads input offsets {-1, 0} with ky {3, 1} (a 2-tap filter with padding 1),
    parity 1 reads {0, +1} with ky {2, 0} (padding 0); taps that fall outside the input contribute nothing."""
    w = weight.detach().to("cpu", torch.float32)
    cin, cout, kh, kw = w.shape
    assert kh == 4 and kw == 4
    wc = w.permute(1, 0, 2, 3).contiguous()          # [cout, cin, ky, kx]
    kmap = {0: (3, 1), 1: (2, 0)}
    plans = []
    for py in (0, 1):
        for px in (0, 1):
            ph, pw = 1 - py, 1 - px
            taps = [(kmap[py][a], kmap[px][b], a - ph, b - pw) for a in range(2) for b in range(2)]
            panel, tab, c0k, c1k, k_pad, cout_pad = _pack_panel(wc, taps, (cin,))
            bb = torch.zeros(cout_pad, dtype=torch.float32)
            if bias is not None:
                bb[:cout] = bias.detach().to("cpu", torch.float32)
            plans.append(ConvPlan(wpack=panel[None].contiguous(), bias=bb, ktab=tab[None].contiguous(), cout=cout,
                                  cout_pad=cout_pad, k_pad=k_pad, c_split=(cin,), c0k=c0k, c1k=c1k, kh=2, kw=2, stride=1,
                                  pad=ph, dil=1, pad_mode=0, upsample=0, nphase=1, flops_per_pixel=2.0 * cout * cin * 4,
                                  pad_w=pw))
    return plans


def pack_conv_rowsplit(weight: torch.Tensor, bias: Optional[torch.Tensor], *, pad: int, pad_mode: int = 0) -> ConvPlan:
    """Small-cout kh x kw convolution as a kh x 1 implicit GEMM with cout*kw (<= 32) output columns:
    t[., co*kw + kx] = sum_{ky, c} in[y + ky - pad, x, c] * w[co, c, ky, kx]; the horizontal taps are
    summed afterwards by fusg_hshift_sum (which also adds the bias and applies the activation).
    Cuts the MFMA work of the 7x7 -> 3 / 1 channel heads by kw (the N tile is 32 wide either way)."""
    w = weight.detach().to("cpu", torch.float32)
    cout, cin, kh, kw = w.shape
    assert cout * kw <= 32, (cout, kw)
    w2 = w.permute(0, 3, 1, 2).reshape(cout * kw, cin, kh, 1)        # [co*kw + kx, c, ky, 0]
    taps = [(ky, 0, ky - pad, 0) for ky in range(kh)]
    panel, tab, c0k, c1k, k_pad, cout_pad = _pack_panel(w2, taps, (cin,))
    plan = ConvPlan(wpack=panel[None].contiguous(), bias=torch.zeros(cout_pad), ktab=tab[None].contiguous(),
                    cout=cout * kw, cout_pad=cout_pad, k_pad=k_pad, c_split=(cin,), c0k=c0k, c1k=c1k, kh=kh, kw=1,
                    stride=1, pad=pad, dil=1, pad_mode=pad_mode, upsample=0, nphase=1,
                    flops_per_pixel=2.0 * cout * cin * kh * kw, pad_w=0)
    plan.rowsplit = {"kw": kw, "pad": pad, "cout": cout,
                     "bias": torch.zeros(cout) if bias is None else bias.detach().to("cpu", torch.float32).clone()}
    return plan


# ---- parameter folds (pure functions of the parameters; exact formulas of the reference's wrappers) ----

def fold_weight_norm(v: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """torch.nn.utils.weight_norm(dim=0) (vunet/layers.py:29-31): w = g * v / ||v||, norm over dims 1..3."""
    v = v.detach().to("cpu", torch.float32)
    g = g.detach().to("cpu", torch.float32)
    return torch._weight_norm(v, g, 0)


def fold_spectral_norm(w_orig: torch.Tensor, u: torch.Tensor, v: torch.Tensor, transposed: bool = False) -> torch.Tensor:
    """nn.utils.spectral_norm in eval mode (edgeconnect/networks.py:206-210): w = w_orig / (u . W_mat v)
    with the stored u, v; W_mat flattens around dim 1 for ConvTranspose2d."""
    w = w_orig.detach().to("cpu", torch.float32)
    wm = w.permute(1, 0, 2, 3).reshape(w.shape[1], -1) if transposed else w.reshape(w.shape[0], -1)
    sigma = torch.dot(u.detach().to("cpu", torch.float32), torch.mv(wm, v.detach().to("cpu", torch.float32)))
    return w / sigma


def bn_scale_shift(weight, bias, running_mean, running_var, eps: float = 1e-5):
    """Eval BatchNorm2d as y = x*scale + shift (computed in fp64, rounded once)."""
    w = weight.detach().to("cpu", torch.float64)
    b = bias.detach().to("cpu", torch.float64)
    rm = running_mean.detach().to("cpu", torch.float64)
    rv = running_var.detach().to("cpu", torch.float64)
    scale = w / torch.sqrt(rv + eps)
    shift = b - rm * scale
    return scale.to(torch.float32), shift.to(torch.float32)


def fold_bn_after_conv(w: torch.Tensor, b: Optional[torch.Tensor], scale: torch.Tensor, shift: torch.Tensor):
    """conv followed directly by eval BatchNorm: w' = w*scale[cout], b' = b*scale + shift."""
    w = w.detach().to("cpu", torch.float32)
    b0 = torch.zeros(w.shape[0]) if b is None else b.detach().to("cpu", torch.float32)
    return w * scale.view(-1, 1, 1, 1), b0 * scale + shift
