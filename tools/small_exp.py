#!/usr/bin/env python3
"""GPU box: the small-image kernel (csrc/conv_kernel_small.h) against the launches it replaces, layer by layer.

Every small-spatial conv shape of the crop pass (VUnet bottleneck levels + AutoRegressiveBlocks, hourglass low levels) at
batch B (default 32): N launches back to back on one stream, time per launch with the small kernel and with FUSG_NO_SMALL=1
(generic gather + split-K reduce, or the halo kernel at 16 x 16) - same process, same card, alternating.  `chain` times a
dependent chain of 20 such launches (what the pass contains: each launch waits for the previous one).
    python tools/small_exp.py [B]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from future_urban_scene_generation_amd import _lib as L, ops, pack  # noqa: E402

SHAPES = [  # c0, c1, cout, k, stride, H, W, count per pass, where
    (128, 0, 128, 3, 1, 8, 8, 7, "vunet residual 8x8"),
    (128, 0, 128, 3, 1, 4, 4, 6, "vunet residual 4x4"),
    (128, 128, 128, 3, 1, 4, 4, 4, "vunet residual(x, skip) 4x4"),
    (128, 128, 128, 3, 1, 8, 8, 2, "vunet residual(x, skip) 8x8"),
    (512, 512, 512, 3, 1, 4, 4, 3, "AR residual_k 4x4"),
    (512, 512, 512, 3, 1, 2, 2, 3, "AR residual_k 2x2"),
    (512, 0, 128, 3, 1, 4, 4, 4, "AR sampler 4x4"),
    (512, 0, 128, 3, 1, 2, 2, 4, "AR sampler 2x2"),
    (128, 0, 128, 1, 1, 4, 4, 5, "NiN 4x4"),
    (128, 0, 128, 1, 1, 8, 8, 3, "NiN 8x8"),
    (128, 0, 512, 1, 1, 2, 2, 3, "AR nin_k 2x2"),
    (128, 0, 512, 1, 1, 4, 4, 3, "AR nin_k 4x4"),
    (128, 128, 128, 1, 1, 4, 4, 2, "NiN(x, z) 4x4"),
    (128, 0, 128, 3, 2, 16, 16, 2, "down 16->8"),
    (128, 0, 128, 3, 2, 8, 8, 2, "down 8->4"),
    (128, 0, 512, 3, 1, 4, 4, 2, "up 4x4"),
    (128, 0, 512, 3, 1, 8, 8, 1, "up 8x8"),
    (128, 0, 128, 3, 1, 16, 16, 4, "vunet residual 16x16 (halo kernel today)"),
    (256, 0, 128, 1, 1, 4, 4, 12, "hg conv1 4x4"),
    (128, 0, 128, 3, 1, 4, 4, 12, "hg conv2 4x4"),
    (128, 0, 256, 1, 1, 4, 4, 12, "hg conv3 4x4"),
    (256, 0, 128, 1, 1, 8, 8, 12, "hg conv1 8x8"),
    (128, 0, 128, 3, 1, 8, 8, 12, "hg conv2 8x8"),
    (128, 0, 256, 1, 1, 8, 8, 12, "hg conv3 8x8"),
    (256, 0, 128, 1, 1, 16, 16, 12, "hg conv1 16x16"),
    (128, 0, 128, 3, 1, 16, 16, 12, "hg conv2 16x16"),
    (128, 0, 256, 1, 1, 16, 16, 12, "hg conv3 16x16"),
]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda:0")
    ops.set_precision("f16x3")
    g = torch.Generator().manual_seed(0)
    tot = {"small": 0.0, "old": 0.0}
    print(f"B={B}   us per launch (a recorded plan of 100 launches on one stream, replayed 5 times: kernel + launch boundary)   [small kernel / replaced launches]")
    for c0, c1, cout, k, s, H, W, cnt, name in SHAPES:
        cin = c0 + c1
        w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
        plan = pack.pack_conv(w, torch.zeros(cout), c_split=(c0, c1) if c1 else None, stride=s, pad=k // 2).to(dev)
        x0 = ops.as_nhwc(torch.randn(B, c0, H, W, generator=g).to(dev))
        x1 = ops.as_nhwc(torch.randn(B, c1, H, W, generator=g).to(dev)) if c1 else None
        out = ops.conv(plan, x0, x1, pre_op=L.PRE_ELU)
        res = {}
        for arm in ("small", "old", "small", "old"):
            if arm == "old":
                os.environ["FUSG_NO_SMALL"] = "1"
            else:
                os.environ.pop("FUSG_NO_SMALL", None)
            for _ in range(5):
                ops.conv(plan, x0, x1, pre_op=L.PRE_ELU, out=out)
            kern = ops.last_conv_kernel()
            torch.cuda.synchronize()
            # issued from C (a recorded plan of N launches, ~3 us of host time each): the interpreter's 12-16 us per ctypes call
            # would hide every kernel shorter than that
            N = 100
            rec = ops.PlanRecorder()
            L.check(L.lib().fusg_plan_begin(rec.handle), "plan_begin")
            ops.RECORDER = rec
            try:
                for _ in range(N):
                    ops.conv(plan, x0, x1, pre_op=L.PRE_ELU, out=out)
            finally:
                ops.RECORDER = None
                L.check(L.lib().fusg_plan_end(rec.handle), "plan_end")
            torch.cuda.synchronize()
            L.check(L.lib().fusg_plan_run(rec.handle), "plan_run")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                L.check(L.lib().fusg_plan_run(rec.handle), "plan_run")
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / (5 * N) * 1e6
            L.lib().fusg_plan_destroy(rec.handle)
            res[arm] = min(res.get(arm, 1e9), us)
            res[arm + "_k"] = kern
        os.environ.pop("FUSG_NO_SMALL", None)
        tot["small"] += res["small"] * cnt
        tot["old"] += res["old"] * cnt
        fl = 2.0 * B * (H // s) * (W // s) * cout * cin * k * k
        print(f"{name:44s} {c0 + c1:5d}->{cout:4d} k{k} s{s} {H:2d}x{W:<2d}  {res['small']:7.1f} / {res['old']:7.1f} us   kernel {res['small_k']}/{res['old_k']}"
              f"   {fl / res['small'] / 1e6:7.1f} TF   x{cnt}", flush=True)
    print(f"sum over the pass's counts: small {tot['small'] / 1e3:.3f} ms, replaced {tot['old'] / 1e3:.3f} ms per pass ")


def ring():
    """The twelve ring launches of one ICN up-convolution (pack.pack_conv_up2_ring), each timed alone: small-image kernel vs the
    generic gather (FUSG_NO_SMALL=1), and the four 25-tap windows they replace."""
    from future_urban_scene_generation_amd.pack import up2_ring_launches
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    dev = torch.device("cuda:0")
    ops.set_precision("f16x3")
    g = torch.Generator().manual_seed(0)
    for cin, cout, h in ((256, 128, 64), (128, 64, 128)):
        w = torch.randn(cout, cin, 5, 5, generator=g) / (cin * 25) ** 0.5
        bvec = torch.zeros(cout)
        rg = {k: v.to(dev) for k, v in pack.pack_conv_up2_ring(w, bvec).items()}
        exact = pack.pack_conv(w, bvec, pad=2, pad_mode=1, upsample=1).to(dev)
        x = ops.as_nhwc(torch.randn(B, cin, h, h, generator=g).to(dev))
        sc = torch.rand(B, cin, generator=g).to(dev) + 0.5
        sh = torch.randn(B, cin, generator=g).to(dev) * 0.2
        out = ops.nhwc_empty(B, cout, 2 * h, 2 * h, dev)
        kw = dict(pre_op=L.PRE_AFFINE_RELU, pre=(sc, sh), pre_bstride=cin)

        def timed(fn, n=50):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            rec = ops.PlanRecorder()
            L.check(L.lib().fusg_plan_begin(rec.handle), "plan_begin")
            ops.RECORDER = rec
            try:
                for _ in range(n):
                    fn()
            finally:
                ops.RECORDER = None
                L.check(L.lib().fusg_plan_end(rec.handle), "plan_end")
            torch.cuda.synchronize()
            L.check(L.lib().fusg_plan_run(rec.handle), "plan_run")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                L.check(L.lib().fusg_plan_run(rec.handle), "plan_run")
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / (3 * n) * 1e6
            L.lib().fusg_plan_destroy(rec.handle)
            return us

        print(f"ring of {cin}->{cout} k5 up2 at {h}x{h}, B={B}:  us per launch  small / generic")
        tot = [0.0, 0.0]
        for ry, rx, win, off in up2_ring_launches(h, h):
            res = []
            for arm in ("small", "old"):
                if arm == "old":
                    os.environ["FUSG_NO_SMALL"] = "1"
                else:
                    os.environ.pop("FUSG_NO_SMALL", None)
                res.append(timed(lambda: ops.conv(rg[(ry, rx)], x, out=out, q_window=win, out_stride=2, out_off=off, **kw)))
                res.append(ops.last_conv_kernel())
            os.environ.pop("FUSG_NO_SMALL", None)
            tot[0] += res[0]
            tot[1] += res[2]
            print(f"   kind {ry}{rx} window {win}  {res[0]:7.1f} / {res[2]:7.1f}   kernel {res[1]}/{res[3]}", flush=True)
        t25 = 0.0
        for win in ((0, 0, 1, 2 * h), (2 * h - 1, 0, 1, 2 * h), (0, 0, 2 * h, 1), (0, 2 * h - 1, 2 * h, 1)):
            t25 += timed(lambda: ops.conv(exact, x, out=out, q_window=win, **kw))
        print(f"   twelve launches: small {tot[0]:.1f} us, generic {tot[1]:.1f} us;  the four 25-tap windows: {t25:.1f} us")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "ring":
        ring()
    else:
        main()
