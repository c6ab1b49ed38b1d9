import sys, os, torch
sys.path.insert(0, os.getcwd())
from future_urban_scene_generation_amd import ops, pack, _lib as L
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
w = torch.randn(256, 256, 3, 3, generator=g) * 0.02
plan = pack.pack_conv(w, None, pad=1)
x = ops.as_nhwc(torch.randn(32, 256, 64, 64, generator=g).to(dev))
prec = sys.argv[1] if len(sys.argv) > 1 else 'f16x3'
for _ in range(3):
    ops.conv(plan, x, precision=prec)
torch.cuda.synchronize()
