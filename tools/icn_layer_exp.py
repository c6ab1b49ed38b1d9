#!/usr/bin/env python3
"""Why does the ICN's 256 -> 256 3x3 residual conv run slower inside the pass (379 TFLOP/s, profiles/r04_layer_profile.txt) than
standing alone (432, r04_halo_layers_sustained.txt)?  Sustained timing of the launch with the pass's features switched on one by
one: reflect padding, fused InstanceNorm statistics, per-image affine pre-op (GPU box, analysis tool).
    FUSG_LIB=.../libfusg_X.so python tools/icn_layer_exp.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from future_urban_scene_generation_amd import _lib as L  # noqa: E402
from future_urban_scene_generation_amd import ops, pack  # noqa: E402
from tools.halo_exp import sustained  # noqa: E402

dev = torch.device("cuda:0")


def main():
    g = torch.Generator().manual_seed(0)
    B, C, H = 32, 256, 64
    w = torch.randn(C, C, 3, 3, generator=g) * (C * 9) ** -0.5
    x = ops.as_nhwc(torch.randn(B, C, H, H, generator=g).to(dev))
    fl = 2.0 * B * H * H * C * C * 9
    glob = (torch.rand(C, generator=g).to(dev) + 0.5, torch.randn(C, generator=g).to(dev) * 0.1)
    perb = (torch.rand(B, C, generator=g).to(dev) + 0.5, torch.randn(B, C, generator=g).to(dev) * 0.1)
    tag = os.path.basename(os.environ.get("FUSG_LIB", "libfusg.so"))
    for name, pm, stats, pre_op, pre, bs in [
        ("zero pad, global affine (halo_exp's case)", L.PAD_ZERO, False, L.PRE_AFFINE_RELU, glob, 0),
        ("reflect pad", L.PAD_REFLECT, False, L.PRE_AFFINE_RELU, glob, 0),
        ("reflect + statistics", L.PAD_REFLECT, True, L.PRE_AFFINE_RELU, glob, 0),
        ("reflect + statistics + per-image affine (conv b of a block)", L.PAD_REFLECT, True, L.PRE_AFFINE_RELU, perb, C),
        ("reflect + statistics, no pre-op (conv a of a block)", L.PAD_REFLECT, True, L.PRE_NONE, None, 0),
    ]:
        plan = pack.pack_conv(w, None, stride=1, pad=1, pad_mode=pm)
        ms, mhz, wat = sustained(lambda: ops.conv(plan, x, pre_op=pre_op, pre=pre, pre_bstride=bs, want_stats=stats, precision="f16x3"))
        print(f"{tag:18s} k{ops.last_conv_kernel()} {name:62s} {ms:8.4f} ms {fl / ms / 1e9:7.1f} TF {mhz:5.0f} MHz {wat:5.0f} W", flush=True)


if __name__ == "__main__":
    main()
