import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from future_urban_scene_generation_amd import ops, pack
g = torch.Generator().manual_seed(21)
x = torch.randn(2, 64, 16, 16, generator=g); w = torch.randn(32, 64, 3, 3, generator=g) * 0.05
plan = pack.pack_conv(w, None, pad=1)
xd = ops.as_nhwc(x.cuda())
a = ops.conv(plan, xd, precision="f32").cpu()
for name, bits in (("emu_bf16x2", 16), ("emu_bf16", 8)):
    b = ops.conv(plan, xd, precision=name).cpu()
    xr, wr = ops._round_sig_bits(x, bits), ops._round_sig_bits(w, bits)
    plan2 = pack.pack_conv(wr, None, pad=1)
    c = ops.conv(plan2, ops.as_nhwc(xr.cuda()), precision="f32").cpu()
    wdev = plan.dev["wpack_r%d" % bits].cpu()
    print(name, "emu vs f32", float((a - b).abs().max()), "emu vs f32-on-rounded", float((b - c).abs().max()), "f32-on-rounded vs f32", float((a - c).abs().max()),
          "weights dev-rounded == cpu-rounded", bool(torch.equal(wdev, ops._round_sig_bits(plan.wpack, bits))), "kernel", ops.last_conv_kernel())
xr = ops._round_sig_bits(x, 8)
b2 = ops.conv(plan, ops.as_nhwc(xr.cuda()), precision="emu_bf16").cpu()
plan2 = pack.pack_conv(ops._round_sig_bits(w, 8), None, pad=1)
c2 = ops.conv(plan2, ops.as_nhwc(xr.cuda()), precision="f32").cpu()
print("emu on pre-rounded x vs f32-on-rounded:", float((b2 - c2).abs().max()))
b3 = ops.conv(plan, ops.as_nhwc(xr.cuda()), precision="emu_bf16", ksplit=1).cpu()
print("same, ksplit=1:", float((b3 - c2).abs().max()))
for t in (1, 2, 3, 4, 5):
    try:
        b4 = ops.conv(plan, ops.as_nhwc(xr.cuda()), precision="emu_bf16", ksplit=1, tile=t).cpu()
        print("tile", t, float((b4 - c2).abs().max()))
    except Exception as e:
        print("tile", t, "err", e)
