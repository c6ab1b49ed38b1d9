#!/usr/bin/env python3
"""Where a workgroup of the 128-column halo kernel spends its cycles (GPU box, analysis tool; needs the diagnostic build libfusg_hstamps.so =
conv_halo_128.hip compiled with -DFUSG_HALO_STAMPS): s_memtime of every wave of the first 64 workgroups at the chunk boundaries - the taps of a
chunk (9 x (LDS fragments + MFMAs)) against the boundary (barrier, commit of the next chunk's halo, barrier) - for the split-fp16 and the
bf16 mode, on a full grid (two workgroups per CU) and on a grid of one workgroup per CU.
    FUSG_LIB=$PWD/future_urban_scene_generation_amd/libfusg_hstamps.so python tools/halo_stamps.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from future_urban_scene_generation_amd import _lib as L  # noqa: E402
from future_urban_scene_generation_amd import ops, pack  # noqa: E402

dev = torch.device("cuda:0")


def stamps():
    buf = (C.c_ulonglong * (64 * 4 * 40))()
    fn = L.lib().fusg_debug_halo_stamps
    fn.argtypes = [C.c_void_p]
    fn.restype = C.c_int
    assert fn(buf) == 0
    return np.frombuffer(buf, dtype=np.uint64).reshape(64, 4, 40).astype(np.int64)


def main():
    g = torch.Generator().manual_seed(0)
    for name, B, cin, cout, H, pre_op in (("vu 128->128 3x3 @256 elu", 32, 128, 128, 256, L.PRE_ELU), ("vu 128->128 3x3 @256 elu, one workgroup per CU", 1, 128, 128, 128, L.PRE_ELU),
                                          ("icn 256->256 3x3 @64 affine", 32, 256, 256, 64, L.PRE_AFFINE_RELU)):
        w = torch.randn(cout, cin, 3, 3, generator=g) * (cin * 9) ** -0.5
        plan = pack.pack_conv(w, None, stride=1, pad=1)
        x = ops.as_nhwc(torch.randn(B, cin, H, H, generator=g).to(dev))
        pre = (torch.rand(cin, generator=g).to(dev) + 0.5, torch.randn(cin, generator=g).to(dev) * 0.1) if pre_op == L.PRE_AFFINE_RELU else None
        nch = cin // 32
        for prec, mf in (("f16x3", 3), ("bf16", 1)):
            for _ in range(5):
                ops.conv(plan, x, pre_op=pre_op, pre=pre, precision=prec, tile=L.TILE_AUTO)
            torch.cuda.synchronize()
            L.lib().fusg_debug_halo_stamps_clear()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.conv(plan, x, pre_op=pre_op, pre=pre, precision=prec)
            e1.record()
            e1.synchronize()
            s = stamps()
            ok = s[:, :, 0] > 0
            if not ok.any():
                print(name, prec, "no stamps (kernel family", ops.last_conv_kernel(), ")")
                continue
            s = s[ok]                                     # [waves, 40]
            taps = np.stack([s[:, 2 + 3 * c] - s[:, 1 + 3 * c] for c in range(nch)], 1)
            bnd = np.stack([s[:, 3 + 3 * c] - s[:, 2 + 3 * c] for c in range(nch - 1)], 1)
            pro = s[:, 1] - s[:, 0]
            tot = s[:, 2 + 3 * (nch - 1)] - s[:, 0]
            mfma = 9 * 8 * 2 * mf * 16
            tp = np.median(np.diff(s[:, 30:39], axis=1), axis=0)
            for tp_ in ((0, 1) if nch <= 6 else ()):
                b0 = s[:, 30 + tp_]
                print(f"      tap {tp_} of chunk 1: halo issue {int(np.median(s[:, 20 + 4 * tp_] - b0))}, weight loads {int(np.median(s[:, 21 + 4 * tp_] - s[:, 20 + 4 * tp_]))}, "
                      f"fragments + MFMAs issued {int(np.median(s[:, 22 + 4 * tp_] - s[:, 21 + 4 * tp_]))}, to the next tap {int(np.median(s[:, 31 + tp_] - s[:, 22 + 4 * tp_]))}")
            print("      taps 0..7 of chunk 1 (cycles from one tap's start to the next):", " ".join(str(int(v)) for v in tp))
            print(f"{name:48s} {prec:5s} k{ops.last_conv_kernel()} launch {e0.elapsed_time(e1) * 1e3:7.1f} us | per wave, median cycles: prologue {np.median(pro):6.0f}, "
                  f"taps of a chunk {np.median(taps):6.0f} (matrix pipe alone: {mfma}), boundary {np.median(bnd):6.0f}, start -> last tap {np.median(tot):7.0f}", flush=True)


if __name__ == "__main__":
    main()
