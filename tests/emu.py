"""CPU emulation of the gather semantics of csrc/conv_igemm.hip, driven by the SAME packed panel and
k-table the kernel consumes (test infrastructure: validates pack.py without a GPU)."""
import numpy as np
import torch
import torch.nn.functional as F


def _nhwc_pad(x, ck):
    """NCHW -> NHWC with channels zero-padded to ck."""
    b, c, h, w = x.shape
    out = torch.zeros(b, h, w, ck, dtype=x.dtype)
    out[..., :c] = x.permute(0, 2, 3, 1)
    return out


def emulate_conv(plan, x0, x1=None, pre=None, pre_scale=None, pre_shift=None):
    """Returns the raw q-space GEMM result [nphase, B, qh, qw, cout] (bias added)."""
    s0 = _nhwc_pad(x0, plan.c0k)
    s1 = _nhwc_pad(x1, plan.c1k) if x1 is not None else None
    B, H, W, _ = s0.shape
    Hv, Wv = H << plan.upsample, W << plan.upsample
    qh, qw = plan.out_hw(H, W)
    oy = torch.arange(qh).view(-1, 1) * plan.stride
    ox = torch.arange(qw).view(1, -1) * plan.stride
    outs = []
    for ph in range(plan.nphase):
        A = torch.zeros(B, qh, qw, plan.k_pad)
        tab = plan.ktab[ph].numpy().astype(np.int64)
        for q in range(plan.k_pad // 4):
            ex, ey = int(tab[q, 0]) & 0xFFFFFFFF, int(tab[q, 1]) & 0xFFFFFFFF
            if ey >> 31:
                continue
            dy = ex & 0xFFFF
            dy = dy - 65536 if dy >= 32768 else dy
            dx = (ex >> 16) & 0xFFFF
            dx = dx - 65536 if dx >= 32768 else dx
            src = (ey >> 30) & 1
            coff = ey & 0x3FFFFFFF
            s = s1 if src else s0
            iy = (oy + dy).expand(qh, qw).clone()
            ix = (ox + dx).expand(qh, qw).clone()
            if plan.pad_mode == 1:
                iy = torch.where(iy < 0, -iy, torch.where(iy >= Hv, 2 * Hv - 2 - iy, iy))
                ix = torch.where(ix < 0, -ix, torch.where(ix >= Wv, 2 * Wv - 2 - ix, ix))
                ok = torch.ones(qh, qw, dtype=torch.bool)
            else:
                ok = (iy >= 0) & (iy < Hv) & (ix >= 0) & (ix < Wv)
            iyc = (iy.clamp(0, Hv - 1) >> plan.upsample)
            ixc = (ix.clamp(0, Wv - 1) >> plan.upsample)
            v = s[:, iyc, ixc, coff:coff + 4]                         # [B, qh, qw, 4]
            cidx = coff + (plan.c0k if src else 0)
            if pre == "relu":
                v = F.relu(v)
            elif pre == "elu":
                v = F.elu(v)
            elif pre in ("affine_relu", "affine"):
                sc = pre_scale[..., cidx:cidx + 4].view(-1, 1, 1, 4)
                sh = pre_shift[..., cidx:cidx + 4].view(-1, 1, 1, 4)
                v = v * sc + sh
                if pre == "affine_relu":
                    v = F.relu(v)
            A[..., q * 4:q * 4 + 4] = torch.where(ok.view(1, qh, qw, 1), v, torch.zeros(()))
        o = A.reshape(-1, plan.k_pad).double() @ plan.wpack[ph].double().t()
        o = (o + plan.bias.double()).float().view(B, qh, qw, plan.cout_pad)[..., :plan.cout]
        outs.append(o)
    return torch.stack(outs)


def assemble_normal(plan, raw):
    """[nphase, B, qh, qw, cout] -> NCHW output for NORMAL store (phases interleaved for nphase 4)."""
    nph, B, qh, qw, c = raw.shape
    if nph == 1:
        return raw[0].permute(0, 3, 1, 2).contiguous()
    out = torch.zeros(B, c, 2 * qh, 2 * qw)
    for ph in range(4):
        out[:, :, (ph >> 1)::2, (ph & 1)::2] = raw[ph].permute(0, 3, 1, 2)
    return out
