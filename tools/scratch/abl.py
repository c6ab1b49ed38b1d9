import sys, os, torch
sys.path.insert(0, os.getcwd())
from future_urban_scene_generation_amd import ops, pack, _lib as L
dev = torch.device('cuda:0')
def t(plan, x, n=5, **kw):
    ops.conv(plan, x, **kw); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.conv(plan, x, **kw)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
g = torch.Generator().manual_seed(0)
w = torch.randn(256, 256, 3, 3, generator=g) * 0.02
plan = pack.pack_conv(w, None, pad=1)
x = ops.as_nhwc(torch.randn(32, 256, 64, 64, generator=g).to(dev))
fl = 2 * 32 * 64 * 64 * 256 * 256 * 9
for prec in ('f16x3', 'f32'):
    ms = t(plan, x, precision=prec)
    print(f'256->256 3x3 @64^2 B32 {prec}: {ms:.3f} ms  {fl/ms/1e9:.1f} TF')
w2 = torch.randn(128, 128, 3, 3, generator=g) * 0.02
plan2 = pack.pack_conv(w2, None, pad=1)
x2 = ops.as_nhwc(torch.randn(32, 128, 256, 256, generator=g).to(dev))
fl2 = 2 * 32 * 256 * 256 * 128 * 128 * 9
for prec in ('f16x3',):
    ms = t(plan2, x2, precision=prec, pre_op=L.PRE_ELU, res0=x2)
    print(f'128->128 3x3 @256^2 B32 elu+res {prec}: {ms:.3f} ms  {fl2/ms/1e9:.1f} TF')
