#!/usr/bin/env python3
"""Where a fused hourglass Bottleneck launch spends its time on the small levels (GPU box, analysis tool; needs the diagnostic build
libfusg_stamps.so = conv_bneck.hip compiled with -DFUSG_BNECK_STAMPS, see tools/README.md): s_memtime of wave 0 at the phase
boundaries of the first 64 workgroups, median over workgroups, for warm (back-to-back) and cold-weight launches.
    FUSG_LIB=$PWD/future_urban_scene_generation_amd/libfusg_stamps.so python tools/bneck_stamps.py [B]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from future_urban_scene_generation_amd import _lib as L  # noqa: E402
from future_urban_scene_generation_amd import ops, pack  # noqa: E402

dev = torch.device("cuda:0")
PH = ["prologue", "conv1 (8 chunks)", "T written", "conv2 (36 steps)", "barrier + U written", "conv3 (4 chunks)", "stores issued"]


def stamps():
    buf = (C.c_ulonglong * (64 * 12))()
    fn = L.lib().fusg_debug_bneck_stamps
    fn.argtypes = [C.c_void_p]
    fn.restype = C.c_int
    assert fn(buf) == 0
    return np.frombuffer(buf, dtype=np.uint64).reshape(64, 12).astype(np.int64)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    g = torch.Generator().manual_seed(0)
    rn = lambda *s: torch.randn(*s, generator=g)                                                       # noqa: E731
    cin = 256
    p = {"pre": ((torch.rand(cin, generator=g) + 0.5).to(dev), (rn(cin) * 0.2).to(dev)),
         "c1": pack.pack_conv(rn(128, cin, 1, 1) / cin ** 0.5, rn(128) * 0.1).to(dev),
         "c2": pack.pack_conv(rn(128, 128, 3, 3) / 1152 ** 0.5, rn(128) * 0.1, pad=1).to(dev),
         "c3": pack.pack_conv(rn(256, 128, 1, 1) / 128 ** 0.5, rn(256) * 0.1).to(dev), "ds": None}
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    for hw in (4, 8, 16, 32, 64):
        x = ops.as_nhwc(rn(B, cin, hw, hw).to(dev))
        nwg = min(64, B * ((hw + 7) // 8) ** 2)
        for mode in ("warm", "cold"):
            for _ in range(4):
                if mode == "cold":
                    flush.fill_(1)                       # 512 MiB written: weights and x leave L2 and the Infinity Cache
                ops.bottleneck(p, x, precision="f16x3")
                torch.cuda.synchronize()
            s = stamps()[:nwg]
            d = np.diff(s[:, :8], axis=1)
            med = np.median(d, axis=0)
            tot = np.median(s[:, 7] - s[:, 0])
            rt = np.median(s[:, 9] - s[:, 8]) / 100.0      # s_memrealtime ticks at 100 MHz -> us
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 50 if mode == "warm" else 1
            e0.record()
            for _ in range(n):
                ops.bottleneck(p, x, precision="f16x3")
            e1.record()
            e1.synchronize()
            print(f"B={B} {hw:2d}x{hw:<2d} {mode}: workgroup {tot:8.0f} cycles = {rt:6.1f} us (clock {tot / max(rt, 1e-9) / 1e3:.2f} GHz); launch-to-launch {e0.elapsed_time(e1) / n * 1e3:6.1f} us"
                  if mode == "warm" else
                  f"B={B} {hw:2d}x{hw:<2d} {mode}: workgroup {tot:8.0f} cycles = {rt:6.1f} us", flush=True)
            print("      " + "   ".join(f"{n_} {int(c)}" for n_, c in zip(PH, med)), flush=True)


if __name__ == "__main__":
    main()
