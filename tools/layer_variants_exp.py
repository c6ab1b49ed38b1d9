#!/usr/bin/env python3
"""Where does a big halo layer lose time inside the pass?  One layer shape, variants of what the pass adds to it
(reflect padding, fused statistics epilogue, residual add), each timed (a) back to back and (b) "cold": a 600 MB fill
between launches, one HIP-event pair per launch (GPU box, analysis tool).
    python tools/layer_variants_exp.py [icn|vu]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from future_urban_scene_generation_amd import _lib as L  # noqa: E402
from future_urban_scene_generation_amd import ops, pack  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, cold, n=30):
    junk = torch.empty(150_000_000, dtype=torch.float32, device=dev) if cold else None
    for _ in range(3):
        fn()
    ms = []
    for _ in range(n):
        if cold:
            junk.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    ms.sort()
    return ms[len(ms) // 2]


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "icn"
    g = torch.Generator().manual_seed(0)
    if which == "icn":
        B, cin, cout, H, pre_op = 32, 256, 256, 64, L.PRE_AFFINE_RELU
    else:
        B, cin, cout, H, pre_op = 32, 128, 128, 256, L.PRE_ELU
    w = torch.randn(cout, cin, 3, 3, generator=g) * (cin * 9) ** -0.5
    bias = torch.randn(cout, generator=g) * 0.1
    x = ops.as_nhwc(torch.randn(B, cin, H, H, generator=g).to(dev))
    res = ops.as_nhwc(torch.randn(B, cout, H, H, generator=g).to(dev))
    pre = None
    if pre_op == L.PRE_AFFINE_RELU:
        pre = (torch.rand(B, cin, generator=g).to(dev) + 0.5, torch.randn(B, cin, generator=g).to(dev) * 0.1)
    fl = 2.0 * B * H * H * cout * cin * 9
    plans = {"zero": pack.pack_conv(w, bias, pad=1), "reflect": pack.pack_conv(w, bias, pad=1, pad_mode=1)}
    out = ops.nhwc_empty(B, cout, H, H, dev)
    cases = [("zero pad", "zero", {}), ("reflect pad", "reflect", {}), ("reflect + stats", "reflect", {"want_stats": True}),
             ("reflect + res", "reflect", {"res0": res}), ("zero + res", "zero", {"res0": res})]
    for name, pk, kw in cases:
        def fn():
            ops.conv(plans[pk], x, out=out, pre_op=pre_op, pre=pre, pre_bstride=cin if pre is not None else 0, precision="f16x3", **kw)
        for cold in (False, True):
            ms = timed(fn, cold)
            print(f"{which} {name:18s} {'cold' if cold else 'warm'}  {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF  kernel {ops.last_conv_kernel()}", flush=True)


if __name__ == "__main__":
    main()
