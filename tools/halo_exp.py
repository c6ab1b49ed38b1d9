#!/usr/bin/env python3
"""Sustained timing of the halo-kernel layer shapes that carry most of the crop pass (GPU box, analysis tool).
Each case loops for ~1.5 s while a thread samples the card's shader clock and socket power from sysfs:
short bursts run before DVFS settles and read 20-25 % slow, and at the 1400 W cap the number that
explains a kernel's time is its energy per launch.
    FUSG_LIB=.../libfusg_X.so python tools/halo_exp.py [name-filter] [--burst]"""
import glob
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from future_urban_scene_generation_amd import _lib as L  # noqa: E402
from future_urban_scene_generation_amd import ops, pack  # noqa: E402

dev = torch.device("cuda:0")
FREQ = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))
POWR = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input") +
              glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average"))


def read(path):
    try:
        return int(open(path).read().strip())
    except (OSError, ValueError):
        return 0


def sustained(fn, seconds=1.5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    idle = [read(p) for p in POWR]
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            samples.append((time.time(), [read(p) for p in FREQ], [read(p) for p in POWR]))
            time.sleep(0.05)

    th = threading.Thread(target=sampler)
    th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time()
    n = 0
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e1.record()
    e1.synchronize()
    t1 = time.time()
    stop[0] = True
    th.join()
    ms = e0.elapsed_time(e1) / n
    run = [(f, pw) for ts, f, pw in samples if t0 + 0.5 <= ts <= t1]
    if not run or not POWR:
        return ms, 0.0, 0.0
    # the card under test = the one whose power rose most over its idle reading
    k = max(range(len(POWR)), key=lambda i: sum(r[1][i] for r in run) / len(run) - idle[i])
    mhz = sum(r[0][k] for r in run) / len(run) / 1e6 if len(FREQ) == len(POWR) else 0.0
    return ms, mhz, sum(r[1][k] for r in run) / len(run) / 1e6


def burst(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n, 0.0, 0.0


def main():
    g = torch.Generator().manual_seed(0)
    s2 = [  # name, B, cin, cout, k, pad_mode, H, pre_op
        ("s2 vu 128->128 3x3 @256 elu", 32, 128, 128, 3, 0, 256, L.PRE_ELU),
        ("s2 icn 64->128 4x4 @256 affine", 32, 64, 128, 4, 1, 256, L.PRE_AFFINE_RELU),
        ("s2 icn 128->256 4x4 @128 affine", 32, 128, 256, 4, 1, 128, L.PRE_AFFINE_RELU),
    ]
    args0 = [a for a in sys.argv[1:] if not a.startswith("--")]
    timer0 = burst if "--burst" in sys.argv else sustained
    for name, B, cin, cout, k, pm, H, pre_op in s2:
        if args0 and args0[0] not in name:
            continue
        w = torch.randn(cout, cin, k, k, generator=g) * (cin * k * k) ** -0.5
        plan = pack.pack_conv(w, None, stride=2, pad=1, pad_mode=pm)
        x = ops.as_nhwc(torch.randn(B, cin, H, H, generator=g).to(dev))
        pre = None
        if pre_op == L.PRE_AFFINE_RELU:
            pre = (torch.rand(cin, generator=g).to(dev) + 0.5, torch.randn(cin, generator=g).to(dev) * 0.1)
        fl = 2.0 * B * (H // 2) ** 2 * cout * cin * k * k
        ms, mhz, wat = timer0(lambda: ops.conv(plan, x, pre_op=pre_op, pre=pre, precision="f16x3"))
        print(f"{'kernel ' + str(ops.last_conv_kernel()):22s} {name:30s} {ms:8.4f} ms {fl / ms / 1e9:7.1f} TF {mhz:5.0f} MHz {wat:5.0f} W {ms * wat / 1e3:.3f} J", flush=True)
    cases = [  # name, B, cin, cout, k, pad, H, pre_op, upsample, dil
        ("icn 256->256 3x3 @64 affine", 32, 256, 256, 3, 1, 64, L.PRE_AFFINE_RELU, 0, 1),
        ("vu 128->128 3x3 @256 elu", 32, 128, 128, 3, 1, 256, L.PRE_ELU, 0, 1),
        ("icn 128->64 5x5 up @128", 32, 128, 64, 5, 2, 128, L.PRE_NONE, 1, 1),
        ("icn 256->128 5x5 up @64", 32, 256, 128, 5, 2, 64, L.PRE_NONE, 1, 1),
        ("hg 128->128 3x3 @64 affine", 32, 128, 128, 3, 1, 64, L.PRE_AFFINE_RELU, 0, 1),
        ("hg 256->128 1x1 @64 affine", 32, 256, 128, 1, 0, 64, L.PRE_AFFINE_RELU, 0, 1),
        ("ec 256->256 3x3 d2 @64 affine", 32, 256, 256, 3, 2, 64, L.PRE_AFFINE_RELU, 0, 2),
        ("vu 64->32 3x3 @256 elu", 32, 64, 32, 3, 1, 256, L.PRE_ELU, 0, 1),
        ("vu 32->32 3x3 @256 elu", 32, 32, 32, 3, 1, 256, L.PRE_ELU, 0, 1),
        ("vu 32->32 1x1 @256 elu", 32, 32, 32, 1, 0, 256, L.PRE_ELU, 0, 1),
        ("vu 128->64 3x3 @64 elu", 32, 128, 64, 3, 1, 64, L.PRE_ELU, 0, 1),
        ("hg 128->256 1x1 @64 affine", 32, 128, 256, 1, 0, 64, L.PRE_AFFINE_RELU, 0, 1),
        ("hg 128->128 3x3 @32 affine", 32, 128, 128, 3, 1, 32, L.PRE_AFFINE_RELU, 0, 1),
        ("hg 128->128 3x3 @16 affine", 32, 128, 128, 3, 1, 16, L.PRE_AFFINE_RELU, 0, 1),
        ("vu 6->128 1x1 @256 elu", 32, 6, 128, 1, 0, 256, L.PRE_ELU, 0, 1),
        ("vu 3->32 1x1 @256 elu", 32, 3, 32, 1, 0, 256, L.PRE_ELU, 0, 1),
        ("hg 12->256 1x1 @64", 32, 12, 256, 1, 0, 64, L.PRE_NONE, 0, 1),
    ]
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    only = args[0] if args else None
    timer = burst if "--burst" in sys.argv else sustained
    tag = os.path.basename(os.environ.get("FUSG_LIB", "libfusg.so"))
    for name, B, cin, cout, k, pad, H, pre_op, up, dil in cases:
        if only and only not in name:
            continue
        w = torch.randn(cout, cin, k, k, generator=g) * (cin * k * k) ** -0.5
        plan = pack.pack_conv(w, None, stride=1, pad=pad, upsample=up, dil=dil)
        x = ops.as_nhwc(torch.randn(B, cin, H, H, generator=g).to(dev))
        pre = None
        if pre_op == L.PRE_AFFINE_RELU:
            pre = (torch.rand(cin, generator=g).to(dev) + 0.5, torch.randn(cin, generator=g).to(dev) * 0.1)
        Ho = H * (2 if up else 1)
        fl = 2.0 * B * Ho * Ho * cout * cin * k * k
        prec = os.environ.get("FUSG_EXP_PRECISION", "f16x3")
        ms, mhz, wat = timer(lambda: ops.conv(plan, x, pre_op=pre_op, pre=pre, precision=prec))
        tagk = tag + ":" + prec + ":k" + str(ops.last_conv_kernel())
        print(f"{tagk:26s} {name:30s} {ms:8.4f} ms {fl / ms / 1e9:7.1f} TF {mhz:5.0f} MHz {wat:5.0f} W {ms * wat / 1e3:.3f} J", flush=True)


if __name__ == "__main__":
    main()
