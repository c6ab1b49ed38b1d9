"""GPU: every libfusg op through the C ABI against torch CPU ops on the same seeded inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["f32", "f16x3"], autouse=True)
def precision(request):
    """Every GPU parity test runs on both contraction paths: exact fp32 MFMA and split-fp16."""
    from future_urban_scene_generation_amd import ops as _ops
    old = _ops.PRECISION
    _ops.set_precision(request.param)
    yield request.param
    _ops.set_precision(old)

from future_urban_scene_generation_amd import _lib as L          # noqa: E402
from future_urban_scene_generation_amd import ops, pack            # noqa: E402


def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _rand(*s, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*s, generator=g) * scale


def _nhwc(x):
    return ops.as_nhwc(x.to(dev()))


def _close(got, ref, rtol=2e-4, atol=2e-4):
    torch.testing.assert_close(got.detach().to("cpu").contiguous(), ref.contiguous(), rtol=rtol, atol=atol)


CONVS = [  # B, cin, cout, k, stride, pad, dil, pad_mode, H, W
    (2, 8, 16, 1, 1, 0, 1, 0, 6, 10),
    (3, 12, 40, 3, 1, 1, 1, 0, 17, 9),
    (2, 12, 40, 3, 2, 1, 1, 0, 8, 12),
    (2, 3, 64, 7, 2, 3, 1, 0, 32, 32),
    (1, 21, 64, 7, 1, 3, 1, 1, 24, 24),
    (2, 64, 128, 4, 2, 1, 1, 1, 16, 16),
    (2, 64, 128, 4, 2, 1, 1, 0, 16, 16),
    (1, 256, 256, 3, 1, 2, 2, 1, 16, 16),
    (1, 128, 64, 5, 1, 2, 1, 1, 12, 12),
    (4, 1024, 512, 3, 1, 1, 1, 0, 2, 2),     # AR-block shape: tiny spatial, split-K
    (2, 64, 3, 7, 1, 3, 1, 1, 16, 16),       # cout 3 head
    (1, 256, 12, 1, 1, 0, 1, 0, 64, 64),     # score conv
    (2, 128, 128, 3, 1, 1, 1, 0, 64, 64),    # big enough for the 128x128 tile
    (2, 64, 128, 4, 2, 1, 1, 1, 32, 64),     # stride 2 in parity-quadrant form (halo kernel): ICN down, reflect
    (2, 64, 128, 4, 2, 1, 1, 0, 32, 64),     # EdgeConnect down, zero
    (2, 32, 64, 3, 2, 1, 1, 0, 32, 64),      # VUnet DownSample
    (1, 128, 128, 3, 2, 1, 1, 0, 16, 32),
    (1, 64, 64, 3, 2, 1, 1, 0, 15, 31),      # odd sizes: same weights, generic gather
]


@pytest.mark.parametrize("B,cin,cout,k,stride,pad,dil,pm,H,W", CONVS)
def test_conv_vs_torch(B, cin, cout, k, stride, pad, dil, pm, H, W):
    x = _rand(B, cin, H, W, seed=1)
    w = _rand(cout, cin, k, k, seed=2, scale=1.0 / (cin * k * k) ** 0.5)
    b = _rand(cout, seed=3)
    plan = pack.pack_conv(w, b, stride=stride, pad=pad, dil=dil, pad_mode=pm)
    xin = F.pad(x, (pad,) * 4, mode="reflect") if pm else x
    ref = F.conv2d(xin, w, b, stride=stride, padding=0 if pm else pad, dilation=dil)
    got = ops.conv(plan, _nhwc(x))
    assert tuple(got.shape) == tuple(ref.shape)
    _close(got, ref)
    got2 = ops.conv(plan, _nhwc(x), nchw_out=True)
    assert got2.is_contiguous()
    _close(got2, ref)


@pytest.mark.parametrize("tile", [L.TILE_128x128, L.TILE_128x64, L.TILE_128x32, L.TILE_64x64, L.TILE_64x128])
@pytest.mark.parametrize("ksplit", [1, 3])
def test_conv_all_tiles_and_splitk(tile, ksplit):
    B, cin, cout, H, W = 2, 96, 128, 20, 13           # M = 520: ragged last tile
    x = _rand(B, cin, H, W, seed=1)
    w = _rand(cout, cin, 3, 3, seed=2, scale=0.03)
    b = _rand(cout, seed=3)
    plan = pack.pack_conv(w, b, pad=1)
    ref = F.conv2d(x, w, b, padding=1)
    _close(ops.conv(plan, _nhwc(x), tile=tile, ksplit=ksplit), ref)


def test_conv_two_sources_elu_residual():
    xa, xb = _rand(2, 32, 9, 9, seed=1), _rand(2, 64, 9, 9, seed=2)
    w = _rand(32, 96, 3, 3, seed=3, scale=0.03)
    b = _rand(32, seed=4)
    plan = pack.pack_conv(w, b, c_split=(32, 64), pad=1)
    ref = F.conv2d(F.elu(torch.cat([xa, xb], 1)), w, b, padding=1) + xa
    a = _nhwc(xa)
    _close(ops.conv(plan, a, _nhwc(xb), pre_op=L.PRE_ELU, res0=a), ref)


def test_conv_bn_affine_relu_and_two_residuals():
    x = _rand(2, 64, 10, 10, seed=1)
    w = _rand(128, 64, 1, 1, seed=2, scale=0.1)
    sc, sh = _rand(64, seed=3), _rand(64, seed=4)
    r0, r1 = _rand(2, 128, 10, 10, seed=5), _rand(2, 128, 10, 10, seed=6)
    plan = pack.pack_conv(w, None)
    ref = F.relu(F.conv2d(F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)), w)) + r0 + r1
    got = ops.conv(plan, _nhwc(x), pre_op=L.PRE_AFFINE_RELU, pre=(sc.to(dev()), sh.to(dev())), act=L.ACT_RELU,
                   res0=_nhwc(r0), res1=r1.to(dev()))            # res1 deliberately NCHW-strided
    _close(got, ref)


def test_conv_per_sample_affine_prologue():
    x = _rand(3, 16, 8, 8, seed=1)
    w = _rand(8, 16, 3, 3, seed=2, scale=0.1)
    sc, sh = _rand(3, 16, seed=3), _rand(3, 16, seed=4)
    plan = pack.pack_conv(w, None, pad=1, pad_mode=1)
    y = F.relu(x * sc.view(3, 16, 1, 1) + sh.view(3, 16, 1, 1))
    ref = F.conv2d(F.pad(y, (1,) * 4, mode="reflect"), w)
    got = ops.conv(plan, _nhwc(x), pre_op=L.PRE_AFFINE_RELU, pre=(sc.to(dev()), sh.to(dev())), pre_bstride=16)
    _close(got, ref)


@pytest.mark.parametrize("act,fn", [(L.ACT_TANH, torch.tanh), (L.ACT_SIGMOID, torch.sigmoid),
                                    (L.ACT_TANH01, lambda t: (torch.tanh(t) + 1) / 2)])
def test_conv_activations(act, fn):
    x = _rand(1, 8, 8, 8, seed=1)
    w = _rand(3, 8, 3, 3, seed=2, scale=0.3)
    plan = pack.pack_conv(w, None, pad=1)
    _close(ops.conv(plan, _nhwc(x), act=act), fn(F.conv2d(x, w, padding=1)), rtol=1e-4, atol=1e-5)


def test_conv_upsample_fused():
    x = _rand(2, 32, 6, 7, seed=1)
    w = _rand(16, 32, 5, 5, seed=2, scale=0.05)
    plan = pack.pack_conv(w, None, pad=2, pad_mode=1, upsample=1)
    ref = F.conv2d(F.pad(F.interpolate(x, scale_factor=2, mode="nearest"), (2,) * 4, mode="reflect"), w)
    _close(ops.conv(plan, _nhwc(x)), ref)


@pytest.mark.parametrize("B,cin,cout,H,W", [(2, 32, 64, 16, 32), (1, 64, 32, 8, 16), (2, 32, 48, 12, 16)])
def test_conv_up2_phases(B, cin, cout, H, W, precision):
    """Upsample(2)+reflect 5x5 through the 4-phase route (when the shape qualifies) == torch; with an affine+ReLU
    pre-op like the ICN decoder; and the route really is taken for the qualifying shapes."""
    x = _rand(B, cin, H, W, seed=1)
    w = _rand(cout, cin, 5, 5, seed=2, scale=1.0 / (cin * 25) ** 0.5)
    b = _rand(cout, seed=3)
    sc, sh = torch.rand(B, cin, generator=torch.Generator().manual_seed(4)) + 0.5, _rand(B, cin, seed=5) * 0.2
    exact = pack.pack_conv(w, b, pad=2, pad_mode=1, upsample=1)
    phases = pack.pack_conv_up2_phases(w, b)
    xin = torch.relu(x * sc[:, :, None, None] + sh[:, :, None, None])
    ref = F.conv2d(F.pad(F.interpolate(xin, scale_factor=2, mode="nearest"), (2,) * 4, mode="reflect"), w, b)
    pre = (sc.to(dev()).contiguous(), sh.to(dev()).contiguous())
    xd = _nhwc(x)
    assert ops.up2_phases_ok(xd)
    got = ops.conv_up2(exact, phases, xd, pre_op=L.PRE_AFFINE_RELU, pre=pre, pre_bstride=cin)
    assert tuple(got.shape) == tuple(ref.shape)
    _close(got, ref)
    got = ops.conv_up2(exact, pack.pack_conv_up2_d2s(w, b), xd, pre_op=L.PRE_AFFINE_RELU, pre=pre, pre_bstride=cin)
    _close(got, ref)                                              # the four phases as one launch with a D2S store


@pytest.mark.parametrize("cin,cout,k,pm,H,W", [(64, 128, 4, 1, 32, 64), (64, 128, 4, 0, 32, 64), (32, 64, 3, 0, 32, 64),
                                                (128, 128, 3, 1, 16, 32)])
def test_conv_stride2_quadrant_form(cin, cout, k, pm, H, W, precision):
    """Stride-2 k3/k4 layers on the halo kernel in parity-quadrant form (and that this IS the path that runs)."""
    x = _rand(2, cin, H, W, seed=1)
    w = _rand(cout, cin, k, k, seed=2, scale=1.0 / (cin * k * k) ** 0.5)
    b = _rand(cout, seed=3)
    plan = pack.pack_conv(w, b, stride=2, pad=1, pad_mode=pm)
    assert plan.s2d_ok()
    xin = F.pad(x, (1,) * 4, mode="reflect") if pm else F.pad(x, (1,) * 4)
    ref = F.conv2d(xin, w, b, stride=2)
    got = ops.conv(plan, _nhwc(x), ksplit=1, pre_op=L.PRE_NONE)
    assert ops.last_conv_kernel() == (3 if precision == "f16x3" else 9)           # (f32: the halo kernel's exact-fp32 mode, round 4)
    _close(got, ref)
    # ELU pre-op + residual through the same path (VUnet DownSample sees ELU'd inputs elsewhere; covers the pre-op kinds)
    got = ops.conv(plan, _nhwc(x), ksplit=1, pre_op=L.PRE_ELU)
    _close(got, F.conv2d(F.pad(F.elu(x), (1,) * 4, mode="reflect") if pm else F.pad(F.elu(x), (1,) * 4), w, b, stride=2))


def test_conv_q_window():
    """q_window computes one window of the output grid (edge rows / columns of a fused-upsample reflect conv), leaves
    the rest of `out` alone, and the library refuses a window that leaves the convolution's output range."""
    x = _rand(2, 32, 9, 11, seed=1)
    w = _rand(24, 32, 5, 5, seed=2, scale=0.05)
    b = _rand(24, seed=3)
    plan = pack.pack_conv(w, b, pad=2, pad_mode=1, upsample=1)
    ref = F.conv2d(F.pad(F.interpolate(x, scale_factor=2, mode="nearest"), (2,) * 4, mode="reflect"), w, b)
    H, W = ref.shape[2:]
    out = ops.nhwc_empty(2, 24, H, W, dev())
    out.fill_(-3.0)
    want = torch.full_like(ref, -3.0)
    for oy, ox, h, ww in ((0, 0, 1, W), (H - 1, 0, 1, W), (0, 0, H, 1), (0, W - 1, H, 1), (3, 4, 5, 6)):
        ops.conv(plan, _nhwc(x), out=out, q_window=(oy, ox, h, ww))
        want[:, :, oy:oy + h, ox:ox + ww] = ref[:, :, oy:oy + h, ox:ox + ww]
    _close(out, want)
    d_ok = False
    try:
        ops.conv(plan, _nhwc(x), out=out, q_window=(0, 0, H, W))
        d_ok = True
    finally:
        assert d_ok
    with pytest.raises(AssertionError):
        ops.conv(plan, _nhwc(x), out=out, q_window=(0, 0, H + 1, W))


@pytest.mark.parametrize("cin,cout,k,stride,pad,pm,H,W", [(21, 64, 7, 1, 3, 1, 16, 32), (3, 64, 7, 2, 3, 0, 32, 64),
                                                           (4, 64, 7, 1, 3, 1, 8, 16), (12, 32, 3, 1, 1, 0, 16, 16),
                                                           (21, 128, 5, 1, 2, 0, 24, 32)])
def test_conv_tapunit_stems(cin, cout, k, stride, pad, pm, H, W, precision):
    """Few-channel k x k layers (the 7x7 stems) on the tap-unit kernel: whole halo staged once, K walked in 8- or
    4-channel units; with the pre-op kinds; and that this IS the path that runs on the split-fp16 side."""
    x = _rand(2, cin, H, W, seed=1)
    w = _rand(cout, cin, k, k, seed=2, scale=1.0 / (cin * k * k) ** 0.5)
    b = _rand(cout, seed=3)
    plan = pack.pack_conv(w, b, stride=stride, pad=pad, pad_mode=pm)
    assert plan.tapunit_ok()
    xd = ops.as_nhwc(x.to(dev()), cpad=plan.c0k)

    def ref_of(xx):
        xin = F.pad(xx, (pad,) * 4, mode="reflect") if pm else F.pad(xx, (pad,) * 4)
        return F.conv2d(xin, w, b, stride=stride)

    got = ops.conv(plan, xd, ksplit=1)
    assert ops.last_conv_kernel() == (4 if precision == "f16x3" else 10)          # (f32: the exact-fp32 tap-unit kernel, round 4)
    _close(got, ref_of(x))
    if precision == "f32":                                                         # ... against the generic fp32 gather it replaces
        import os
        os.environ["FUSG_NO_F32_HALO"] = "1"
        try:
            old = ops.conv(plan, xd, ksplit=1)
            assert ops.last_conv_kernel() == 0
        finally:
            del os.environ["FUSG_NO_F32_HALO"]
        _close(got, old.cpu(), rtol=1e-5, atol=2e-6 * float(old.abs().max()))
    _close(ops.conv(plan, xd, ksplit=1, pre_op=L.PRE_ELU), ref_of(F.elu(x)))
    sc, sh = torch.rand(2, plan.c0k, generator=torch.Generator().manual_seed(4)) + 0.5, _rand(2, plan.c0k, seed=5) * 0.2
    xa = torch.relu(x * sc[:, :cin, None, None] + sh[:, :cin, None, None])
    got = ops.conv(plan, xd, ksplit=1, pre_op=L.PRE_AFFINE_RELU, pre=(sc.to(dev()).contiguous(), sh.to(dev()).contiguous()),
                   pre_bstride=plan.c0k)
    _close(got, ref_of(xa))
    out, (isc, ish) = ops.conv_in(plan, xd, ksplit=1)                   # fused InstanceNorm statistics (ICN / EC stems)
    r = ref_of(x)
    rstd = (r.var(dim=(2, 3), unbiased=False) + 1e-5).rsqrt()
    _close(isc, rstd, rtol=2e-4, atol=2e-4)
    _close(ish, -r.mean(dim=(2, 3)) * rstd, rtol=2e-4, atol=5e-4)


def test_conv_tile_list_and_replicate_pad(precision):
    """PAD_REPLICATE = edge clamp on every conv path; tile_list computes only the listed 8x16 patches (halo kernel)."""
    x = _rand(2, 32, 24, 48, seed=1)
    w = _rand(32, 32, 3, 3, seed=2, scale=0.05)
    ref = F.conv2d(F.pad(x, (1,) * 4, mode="replicate"), w)
    plan = pack.pack_conv(w, None, pad=1, pad_mode=L.PAD_REPLICATE)
    _close(ops.conv(plan, _nhwc(x), ksplit=1), ref)                # halo kernel on the split-fp16 path
    _close(ops.conv(plan, _nhwc(x)), ref)                          # split-K -> generic gather
    if precision != "f16x3":
        with pytest.raises(RuntimeError):
            ops.conv(plan, _nhwc(x), tiles=ops.border_tiles(24, 48, dev()))       # tile lists are a halo-kernel feature
        return
    tiles = ops.border_tiles(24, 48, dev())                        # 3 x 3 patch grid: all but the centre patch
    assert sorted(tiles.cpu().tolist()) == [0, 1, 2, 3, 5, 6, 7, 8]
    out = ops.nhwc_empty(2, 32, 24, 48, dev())
    out.fill_(7.0)
    ops.conv(plan, _nhwc(x), out=out, tiles=tiles, ksplit=1)
    got = out.cpu()
    assert torch.all(got[:, :, 8:16, 16:32] == 7.0)
    mask = torch.ones(24, 48, dtype=torch.bool)
    mask[8:16, 16:32] = False
    torch.testing.assert_close(got[:, :, mask], ref[:, :, mask], rtol=2e-4, atol=2e-4)


def test_conv_transpose():
    x = _rand(2, 32, 7, 9, seed=1)
    w = _rand(32, 24, 4, 4, seed=2, scale=0.1)
    b = _rand(24, seed=3)
    plan = pack.pack_conv_transpose_k4s2p1(w, b)
    _close(ops.conv(plan, _nhwc(x)), F.conv_transpose2d(x, w, b, stride=2, padding=1))


@pytest.mark.parametrize("cin,cout,H,W", [(64, 32, 16, 32), (32, 64, 8, 16), (32, 16, 10, 12)])
def test_conv_transpose_phases(cin, cout, H, W, precision):
    """ConvTranspose2d(k4,s2,p1) as four dense 2x2 launches (halo kernel where the shape qualifies) + the InstanceNorm
    statistics fused across the four launches."""
    x = _rand(2, cin, H, W, seed=1)
    w = _rand(cin, cout, 4, 4, seed=2, scale=0.05)
    b = _rand(cout, seed=3)
    ref = F.conv_transpose2d(x, w, b, stride=2, padding=1)
    phases = pack.pack_conv_transpose_k4s2p1_phases(w, b)
    out, stats = ops.conv_transpose_phases(phases, _nhwc(x), want_stats=True)
    _close(out, ref)
    halo = H % 8 == 0 and W % 16 == 0
    assert ops.last_conv_kernel() == ((2 if precision == "f16x3" else 9) if halo else (1 if precision == "f16x3" else 0))
    out2, (sc, sh) = ops.conv_in(phases, _nhwc(x))
    _close(out2, ref)
    mean, var = ref.mean(dim=(2, 3)), ref.var(dim=(2, 3), unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    _close(sc, rstd, rtol=2e-4, atol=2e-4)
    _close(sh, -mean * rstd, rtol=2e-4, atol=5e-4)


def test_conv_depth_to_space_store_and_channel_slice():
    import oracle
    x = _rand(2, 16, 5, 6, seed=1)
    w = _rand(32, 16, 3, 3, seed=2, scale=0.1)
    plan = pack.pack_conv(w, None, pad=1)
    ref = F.conv2d(x, w, padding=1)
    _close(ops.conv(plan, _nhwc(x), store=L.STORE_D2S), oracle.depth_to_space(ref))
    _close(ops.conv(plan, _nhwc(F.pad(x, (0, 0, 0, 1))), store=L.STORE_S2D),
           oracle.space_to_depth(F.conv2d(F.pad(x, (0, 0, 0, 1)), w, padding=1)))
    buf = ops.nhwc_empty(2, 96, 5, 6, dev(), zero=True)
    ops.conv(plan, _nhwc(x), out=buf, out_c_off=32)
    full = torch.cat([torch.zeros_like(ref), ref, torch.zeros_like(ref)], 1)
    _close(buf, full)


def test_conv_rejects_bad_layout():
    x = _rand(1, 8, 4, 4).to(dev())                       # NCHW-contiguous: not a valid conv source
    plan = pack.pack_conv(_rand(8, 8, 1, 1), None)
    with pytest.raises(L.FusgError):
        ops.conv(plan, x)


def test_instnorm_and_layernorm_stats():
    x = _rand(3, 64, 33, 17, seed=1) * 2 + 5            # large mean: exercises the pivot shift
    xn = _nhwc(x)
    sc, sh = ops.instnorm_stats(xn)
    ref = F.instance_norm(x, use_input_stats=True, eps=1e-5)
    got = x * sc.cpu().view(3, 64, 1, 1) + sh.cpu().view(3, 64, 1, 1)
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4)
    _close(ops.affine_act(xn, sc, sh, L.ACT_RELU, res=xn), F.relu(ref) + x, rtol=1e-4, atol=1e-4)
    gamma, beta = _rand(64, seed=2), _rand(64, seed=3)
    sc, sh = ops.layernorm_stats(xn, gamma.to(dev()), beta.to(dev()))
    mean = x.view(3, -1).mean(1).view(3, 1, 1, 1)
    std = x.view(3, -1).std(1).view(3, 1, 1, 1)
    ref = (x - mean) / (std + 1e-5) * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)
    got = x * sc.cpu().view(3, 64, 1, 1) + sh.cpu().view(3, 64, 1, 1)
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4)


def test_instnorm_stats_wide_and_narrow_channels():
    for c, hw in ((256, (16, 16)), (12, (8, 8)), (320, (4, 4))):
        x = _rand(2, c, *hw, seed=c)
        sc, sh = ops.instnorm_stats(_nhwc(x))
        ref = F.instance_norm(x, use_input_stats=True, eps=1e-5)
        got = x * sc.cpu().view(2, c, 1, 1) + sh.cpu().view(2, c, 1, 1)
        torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4)


def test_pool_upsample_copy_s2d_d2s():
    import oracle
    x = _rand(2, 32, 8, 12, seed=1)
    _close(ops.maxpool2(_nhwc(x)), F.max_pool2d(x, 2, stride=2), 0, 0)
    low = _rand(2, 32, 4, 6, seed=2)
    _close(ops.upsample2_add(_nhwc(low), _nhwc(x)), x + F.interpolate(low, scale_factor=2), 0, 0)
    _close(ops.space_to_depth2(_nhwc(x)), oracle.space_to_depth(x), 0, 0)
    _close(ops.depth_to_space2(_nhwc(x)), oracle.depth_to_space(x), 0, 0)
    y = _rand(2, 3, 5, 7, seed=3)
    yn = _nhwc(y)
    assert yn.stride(1) == 1 and yn.stride(3) == 4
    _close(yn, y, 0, 0)
    _close(ops.to_nchw(yn), y, 0, 0)
    _close(ops.add4d(yn, y.to(dev())), y + y, 0, 0)


def test_argmax_first_occurrence():
    h = _rand(2, 12, 64, 64, seed=5)
    h[0, 0, 10, 20] = h[0, 0, 40, 3] = 50.0            # tie: row-major first wins
    h[1, 3] = -2.0                                     # constant map -> 0
    ref = h.reshape(2, 12, -1).argmax(dim=2)
    ref[1, 3] = 0
    assert ref[0, 0] == 10 * 64 + 20
    for t in (_nhwc(h), h.to(dev())):                  # both layouts
        got = ops.argmax_hw(t).cpu().long()
        assert torch.equal(got, ref)


def test_to_image_and_merge_bit_exact():
    import oracle
    x = torch.linspace(-1.3, 1.3, 2 * 3 * 32 * 32).reshape(2, 3, 32, 32)
    got = ops.to_image_u8(x.to(dev())).cpu().numpy()
    assert np.array_equal(got, oracle.to_image_u8(x))
    assert ops.to_image_u8(torch.full((1, 3, 1, 1), 0.00392).to(dev()))[0, 0, 0, 0].item() == 127
    o, img = torch.rand(2, 3, 16, 16, generator=torch.Generator().manual_seed(1)), torch.rand(2, 3, 16, 16, generator=torch.Generator().manual_seed(2))
    m = (torch.rand(2, 1, 16, 16, generator=torch.Generator().manual_seed(3)) > 0.5).float()
    ref = ((o * m + img * (1 - m)) * 255.0).permute(0, 2, 3, 1).numpy().astype(np.uint8)
    assert np.array_equal(ops.merge_u8(o.to(dev()), img.to(dev()), m.to(dev())).cpu().numpy(), ref)


def test_ec_inputs():
    g = torch.Generator().manual_seed(1)
    gray, img = torch.rand(2, 1, 8, 8, generator=g), torch.rand(2, 3, 8, 8, generator=g)
    edge = (torch.rand(2, 1, 8, 8, generator=g) > 0.7).float()
    m = (torch.rand(2, 1, 8, 8, generator=g) > 0.5).float()
    e = ops.ec_inputs(gray.to(dev()), edge.to(dev()), m.to(dev()), 0)
    _close(e, torch.cat((gray * (1 - m) + m, edge * (1 - m), m), 1), 0, 0)
    p = ops.ec_inputs(img.to(dev()), edge.to(dev()), m.to(dev()), 1)
    _close(p, torch.cat((img * (1 - m) + m, edge), 1), 0, 0)


@pytest.mark.parametrize("cout,act,fn", [(3, L.ACT_TANH, torch.tanh), (1, L.ACT_SIGMOID, torch.sigmoid)])
def test_rowsplit_head(cout, act, fn):
    x = _rand(2, 64, 24, 20, seed=1)
    w = _rand(cout, 64, 7, 7, seed=2, scale=0.02)
    b = _rand(cout, seed=3)
    sc, sh = _rand(2, 64, seed=4), _rand(2, 64, seed=5)
    plan = pack.pack_conv_rowsplit(w, b, pad=3, pad_mode=1)
    y = F.relu(x * sc.view(2, 64, 1, 1) + sh.view(2, 64, 1, 1))
    ref = fn(F.conv2d(F.pad(y, (3,) * 4, mode="reflect"), w, b))
    got = ops.conv_rowsplit(plan, _nhwc(x), pre_op=L.PRE_AFFINE_RELU, pre=(sc.to(dev()), sh.to(dev())), pre_bstride=64, act=act)
    assert got.is_contiguous()
    _close(got, ref)


@pytest.mark.parametrize("cin,cout,pre", [(6, 128, L.PRE_ELU), (3, 32, L.PRE_ELU), (8, 64, L.PRE_NONE), (5, 12, L.PRE_RELU)])
def test_pointwise_from_few_channels(cin, cout, pre, precision):
    """1x1 from <= 8 channels (the VUnet's NiN stems) runs on the streaming fp32-FMA kernel under the split-fp16
    precision: same bits as the exact-fp32 MFMA kernel (an fmaf chain in channel order), residual and activation fused."""
    x = _rand(2, cin, 24, 40, seed=71)
    w = _rand(cout, cin, 1, 1, seed=72, scale=0.4)
    b = _rand(cout, seed=73)
    res = _rand(2, cout, 24, 40, seed=74)
    plan = pack.pack_conv(w, b)
    xin = _nhwc(x)
    rin = _nhwc(res)
    got = ops.conv(plan, xin, pre_op=pre, res0=rin, act=L.ACT_TANH, precision="f16x3")
    fam = ops.last_conv_kernel()
    import os
    os.environ["FUSG_NO_POINTWISE"] = "1"                       # the exact-fp32 MFMA kernel itself (f32 takes the streaming kernel too)
    try:
        want32 = ops.conv(plan, xin, pre_op=pre, res0=rin, act=L.ACT_TANH, precision="f32")
        assert ops.last_conv_kernel() == 0
    finally:
        del os.environ["FUSG_NO_POINTWISE"]
    xp = {L.PRE_NONE: x, L.PRE_RELU: F.relu(x), L.PRE_ELU: F.elu(x)}[pre]
    ref = torch.tanh(F.conv2d(xp, w, b)) + res
    _close(got, ref)
    if cout % 4 == 0:
        assert fam == 7, fam
        assert torch.equal(got, want32)
    assert not ops.range_exceeded(dev())


# ---- split-fp16 contraction: accuracy over operand scale, and the range guard -----------------------------------
SWEEP_LAYERS = [   # name, cin, cout, k, pad, H, W, pre-op, expected kernel family (fusg_last_conv_kernel)
    ("halo_k2304", 256, 64, 3, 1, 16, 16, L.PRE_NONE, 2),
    ("halo_k2304_relu", 256, 64, 3, 1, 16, 16, L.PRE_RELU, 2),
    ("generic_k2304", 256, 64, 3, 1, 15, 17, L.PRE_NONE, 1),
    ("generic_k2304_elu", 256, 64, 3, 1, 15, 17, L.PRE_ELU, 1),
    ("tapunit_k1176", 24, 32, 7, 3, 16, 16, L.PRE_NONE, 4),
]


@pytest.mark.parametrize("name,cin,cout,k,pad,H,W,pre,family", SWEEP_LAYERS)
def test_f16x3_scale_sweep(name, cin, cout, k, pad, H, W, pre, family, precision):
    """the split-fp16 contraction must stay fp32-class when the operands are not O(1).  For operand
    scales {1e-3, 1e-2, 1, 1e2} x {2e-3, 2e-2, 1} the error against an fp64 reference (normalised by |a| * |w|) of the
    f16x3 kernel is compared with that of the exact-fp32 MFMA kernel on the same launch: bar = 2x (observed 0.96-1.27x,
    profiles/r02_parity.json: the claim is "<= 1.3x the exact-fp32 kernel's error", not "<=")."""
    if precision != "f16x3":
        pytest.skip("compares both kernels itself")
    from conftest import record
    worst = 0.0
    for sa in (1e-3, 1e-2, 1.0, 1e2):
        for sw in (2e-3, 2e-2, 1.0):
            x = _rand(2, cin, H, W, seed=11, scale=sa)
            w = _rand(cout, cin, k, k, seed=12, scale=sw)
            b = _rand(cout, seed=13, scale=sa * sw)
            plan = pack.pack_conv(w, b, pad=pad)
            xp = {L.PRE_NONE: x, L.PRE_RELU: F.relu(x), L.PRE_ELU: F.elu(x)}[pre].double()
            ref = F.conv2d(xp, w.double(), b.double(), padding=pad)
            den = F.conv2d(xp.abs(), w.double().abs(), None, padding=pad) + 1e-300
            xin = _nhwc(x)
            ks = 0 if family == 1 else 1                      # (the halo / tap-unit kernels do not split K)
            got3 = ops.conv(plan, xin, pre_op=pre, precision="f16x3", ksplit=ks)
            assert ops.last_conv_kernel() == family, (name, ops.last_conv_kernel())
            assert not ops.range_exceeded(dev())
            got32 = ops.conv(plan, xin, pre_op=pre, precision="f32", ksplit=ks)
            e3 = float(((got3.cpu().double() - ref).abs() / den).max())
            e32 = float(((got32.cpu().double() - ref).abs() / den).max())
            record("f16x3_err_max", e3)
            record("f32_err_max", e32)
            record("ratio_max", e3 / e32)
            worst = max(worst, e3 / e32)
            assert e3 <= 2 * e32 + 1e-9, (name, sa, sw, e3, e32)
    assert worst < 2


def test_f16x3_scale_sweep_fused_bottleneck(precision):
    """The same sweep for the one-launch hourglass Bottleneck (fusg_hg_bottleneck): three chained contractions with both
    intermediates kept as split fp16 in LDS.  Input scales {1e-2, 1, 1e2} x weight scales {1e-2, 1, 30} against fp64,
    compared with the three exact-fp32 launches on the same block: bar = 2x the fp32 path's error (+ 1e-7 of the
    output's range for the residual's own rounding)."""
    if precision != "f16x3":
        pytest.skip("compares both paths itself")
    from conftest import record
    worst = 0.0
    for sa in (1e-2, 1.0, 1e2):
        for sw in (1e-2, 1.0, 30.0):
            p, prm = _bneck_params(256, seed=41, scale=sw)
            x = _rand(2, 256, 16, 16, seed=42, scale=sa)
            ref = _bneck_ref(x, x, prm)
            xin = _nhwc(x)
            got = ops.bottleneck(p, xin)
            assert ops.last_conv_kernel() == 6
            if ops.range_exceeded(dev()):                      # an intermediate left the split's range: a legal outcome
                continue                                       # (the caller redoes the block in fp32), not an accuracy case
            t = ops.conv(p["c1"], xin, pre_op=L.PRE_AFFINE_RELU, pre=p["pre"], act=L.ACT_RELU, precision="f32")
            t = ops.conv(p["c2"], t, act=L.ACT_RELU, precision="f32")
            f32 = ops.conv(p["c3"], t, res0=xin, precision="f32")
            den = float(ref.abs().max())
            e3 = float((got.cpu().double() - ref).abs().max()) / den
            e32 = float((f32.cpu().double() - ref).abs().max()) / den
            record("bneck_f16x3_rel_err", e3)
            record("bneck_f32_rel_err", e32)
            record("bneck_ratio_max", e3 / max(e32, 1e-12))
            worst = max(worst, e3 / max(e32, 1e-12))
            assert e3 <= 2 * e32 + 1e-7, (sa, sw, e3, e32)
    assert worst > 0                                           # at least one in-range combination was compared


def test_bf16_halo_scale_sweep(precision):
    """The single-pass bf16 mode of the halo kernel (precision="bf16", BASELINE configs[4]) over the same operand scales:
    both operands are rounded to 8 significant bits, so the error normalised by sum|a||w| must stay below 2^-8 (two
    roundings of 2^-9 each) at EVERY scale - bf16 has fp32's exponent range, there is no range guard to trip - and it
    is recorded next to the f16x3 error on the same launch (observed ~300x apart: this mode is not fp32-class and is
    never the headline of an fp32 configuration)."""
    if precision != "f16x3":
        pytest.skip("one run is enough")
    from conftest import record
    for sa in (1e-3, 1.0, 1e2, 1e6):
        for sw in (2e-3, 1.0):
            x = _rand(2, 256, 16, 16, seed=51, scale=sa)
            w = _rand(64, 256, 3, 3, seed=52, scale=sw)
            plan = pack.pack_conv(w, None, pad=1)
            ref = F.conv2d(x.double(), w.double(), None, padding=1)
            den = F.conv2d(x.double().abs(), w.double().abs(), None, padding=1) + 1e-300
            got = ops.conv(plan, _nhwc(x), precision="bf16", ksplit=1)
            assert ops.last_conv_kernel() == 5, ops.last_conv_kernel()
            assert not ops.range_exceeded(dev())
            e = float(((got.cpu().double() - ref).abs() / den).max())
            record("bf16_halo_err_max", e)
            assert e <= 2.0 ** -8, (sa, sw, e)


def test_nan_inputs_propagate_except_through_a_fused_relu(precision):
    """ADVICE r2 (documented behaviour, csrc/conv_kernel_h3.h header): a NaN activation reaches every output it feeds
    on both contractions - except through a fused ReLU, where fmaxf(NaN, 0) = 0 on BOTH precisions (torch's relu
    would keep it).  The two precisions agree with each other in every case."""
    x = _rand(1, 64, 16, 16, seed=61)
    x[0, 5, 8, 8] = float("nan")
    w = _rand(32, 64, 3, 3, seed=62, scale=0.05)
    plan = pack.pack_conv(w, None, pad=1)
    for pre in (L.PRE_NONE, L.PRE_ELU):
        got = ops.conv(plan, _nhwc(x), pre_op=pre, ksplit=1).cpu()
        assert bool(torch.isnan(got[0, :, 7:10, 7:10]).all()), pre      # the 3x3 neighbourhood, all channels
        assert int(torch.isnan(got).sum()) == 32 * 9
    ops.range_exceeded(dev())
    got = ops.conv(plan, _nhwc(x), pre_op=L.PRE_RELU, ksplit=1).cpu()
    xz = x.clone()
    xz[0, 5, 8, 8] = 0.0
    ref = F.conv2d(F.relu(xz).double(), w.double(), None, padding=1)
    assert not bool(torch.isnan(got).any())
    assert float((got.double() - ref).abs().max() / ref.abs().max()) < 1e-5
    ops.range_exceeded(dev())


@pytest.mark.parametrize("name,cin,cout,k,pad,H,W,pre,family", SWEEP_LAYERS[::2])
def test_f16x3_out_of_range_raises_status(name, cin, cout, k, pad, H, W, pre, family, precision):
    """An operand the split cannot represent (|x| >= 2^15, inf) is never clamped silently: the launch raises the
    caller-owned status word (fusg_conv_desc.status)."""
    if precision != "f16x3":
        pytest.skip("f16x3 only")
    w = _rand(cout, cin, k, k, seed=2, scale=0.02)
    plan = pack.pack_conv(w, None, pad=pad)
    for bad in (1e5, -4e4, float("inf")):
        x = _rand(1, cin, H, W, seed=1)
        assert not ops.range_exceeded(dev())
        ks = 0 if family == 1 else 1
        ops.conv(plan, _nhwc(x), pre_op=pre, precision="f16x3", ksplit=ks)
        assert not ops.range_exceeded(dev())                     # in range: status stays clear
        x[0, cin // 2, H // 2, W // 3] = bad
        ops.conv(plan, _nhwc(x), pre_op=pre, precision="f16x3", ksplit=ks)
        assert ops.last_conv_kernel() == family
        assert ops.range_exceeded(dev())                         # raised (and cleared by the read)
        assert not ops.range_exceeded(dev())
    # just inside the range: exact handling, no flag
    x = _rand(1, cin, H, W, seed=1)
    x[0, 0, 0, 0] = 32767.0
    got = ops.conv(plan, _nhwc(x), pre_op=pre, precision="f16x3", ksplit=0 if family == 1 else 1)
    assert not ops.range_exceeded(dev())
    ref = F.conv2d({L.PRE_NONE: x, L.PRE_RELU: F.relu(x), L.PRE_ELU: F.elu(x)}[pre].double(), w.double(), None, padding=pad)
    assert float((got.cpu().double() - ref).abs().max() / ref.abs().max()) < 1e-6


def test_reduced_precision_emulation_is_what_it_says(precision):
    """FUSG_PREC_EMU_BF16 / EMU_BF16X2 (the evidence paths of test_reduced_precision_evidence): the launch must equal a
    convolution of operands rounded to 8 / 16 significant bits with exact products - checked against fp64 on operands
    rounded on the CPU - and differ from the unrounded convolution by about 2^-9 / 2^-17 per operand."""
    if precision != "f32":
        pytest.skip("one run")
    x = _rand(2, 64, 16, 16, seed=21)
    w = _rand(32, 64, 3, 3, seed=22, scale=0.05)
    plan = pack.pack_conv(w, None, pad=1)
    exact = F.conv2d(x.double(), w.double(), padding=1)
    den = F.conv2d(x.double().abs(), w.double().abs(), padding=1)
    for name, bits in (("emu_bf16", 8), ("emu_bf16x2", 16)):
        xr, wr = ops._round_sig_bits(x, bits).double(), ops._round_sig_bits(w, bits).double()
        if bits == 8:
            assert torch.equal(xr.float(), x.bfloat16().float())          # 8 significant bits, ties to even = bf16
        ref = F.conv2d(xr, wr, padding=1)
        got = ops.conv(plan, _nhwc(x), precision=name).cpu().double()
        assert float(((got - ref).abs() / den).max()) < 2e-6             # only the fp32 accumulation differs
        err = float(((got - exact).abs() / den).max())
        assert 2.0 ** -(bits + 5) < err < 2.0 ** -(bits - 1), (name, err)


@pytest.mark.parametrize("cin,cout,k,stride,pad,dil,pm,H,W,pre", [
    (128, 128, 3, 1, 1, 1, 0, 32, 32, L.PRE_ELU),          # VUnet residual shape
    (256, 256, 3, 1, 2, 2, 1, 16, 16, L.PRE_AFFINE_RELU),  # EdgeConnect dilated, reflect
    (64, 32, 3, 1, 1, 1, 0, 32, 64, L.PRE_NONE),           # BN = 32 tile
    (128, 64, 5, 1, 2, 1, 1, 16, 32, L.PRE_NONE),          # 5x5 (NI = 8)
    (64, 128, 4, 2, 1, 1, 1, 32, 64, L.PRE_RELU),          # stride 2 in parity-quadrant form
    (128, 256, 1, 1, 0, 1, 0, 16, 16, L.PRE_NONE),         # 1x1
])
def test_bf16_halo_conv(cin, cout, k, stride, pad, dil, pm, H, W, pre, precision):
    """FUSG_PREC_BF16: the halo kernel in single-pass bf16 equals a convolution of bf16-rounded operands with exact
    products (only the fp32 accumulation order differs), on every pre-op / padding / stride form it supports."""
    if precision != "f16x3":
        pytest.skip("one run")
    x = _rand(2, cin, H, W, seed=31)
    w = _rand(cout, cin, k, k, seed=32, scale=1.0 / (cin * k * k) ** 0.5)
    b = _rand(cout, seed=33)
    sc, sh = _rand(cin, seed=34).abs() + 0.5, _rand(cin, seed=35) * 0.1
    plan = pack.pack_conv(w, b, stride=stride, pad=pad, dil=dil, pad_mode=pm)
    xp = {L.PRE_NONE: x, L.PRE_RELU: F.relu(x), L.PRE_ELU: F.elu(x),
          L.PRE_AFFINE_RELU: F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))}[pre]
    xr, wr = xp.bfloat16().double(), w.bfloat16().double()
    xin = F.pad(xr, (pad,) * 4, mode="reflect") if pm else xr
    ref = F.conv2d(xin, wr, b.double(), stride=stride, padding=0 if pm else pad, dilation=dil)
    den = F.conv2d(xin.abs(), wr.abs(), None, stride=stride, padding=0 if pm else pad, dilation=dil) + 1e-30
    kw = dict(pre=(sc.to(dev()), sh.to(dev()))) if pre == L.PRE_AFFINE_RELU else {}
    got = ops.conv(plan, _nhwc(x), pre_op=pre, precision="bf16", ksplit=1, **kw)
    assert ops.last_conv_kernel() == 5, ops.last_conv_kernel()
    assert float(((got.cpu().double() - ref).abs() / den).max()) < (3e-5 if pre == L.PRE_ELU else 3e-6)   # (GPU exp in the ELU)
    # a layer that does not qualify for the halo kernel runs as f16x3 under the same setting
    w2 = _rand(16, 12, 3, 3, seed=36, scale=0.1)
    p2 = pack.pack_conv(w2, None, pad=1)
    x2 = _rand(1, 12, 9, 9, seed=37)
    g2 = ops.conv(p2, _nhwc(x2), precision="bf16")
    assert ops.last_conv_kernel() in (1, 4)
    _close(g2, F.conv2d(x2, w2, padding=1), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("cin,cout,k,stride,pad,pm,H,W,pre", [
    (21, 64, 7, 1, 3, 1, 32, 32, L.PRE_NONE),               # ICN stem: 24 staged channels (units of 8), reflect padding
    (3, 64, 7, 2, 3, 0, 64, 64, L.PRE_NONE),                # hourglass stem shape: 4 staged channels (units of 4), stride 2
    (4, 64, 7, 1, 3, 1, 16, 32, L.PRE_ELU),                 # EdgeConnect stem shape with a pre-op
])
def test_bf16_tapunit_stems(cin, cout, k, stride, pad, pm, H, W, pre, precision):
    """FUSG_PREC_BF16 on the few-channel stems (round 4: the tap-unit kernel's bf16 mode, kernel family 11) equals a convolution of
    bf16-rounded operands with exact products; FUSG_NO_BF16_TAPUNIT keeps the split-fp16 kernel (family 4) under the same setting."""
    if precision != "f16x3":
        pytest.skip("one run")
    import os
    x = _rand(2, cin, H, W, seed=71)
    w = _rand(cout, cin, k, k, seed=72, scale=1.0 / (cin * k * k) ** 0.5)
    b = _rand(cout, seed=73)
    plan = pack.pack_conv(w, b, stride=stride, pad=pad, pad_mode=pm)
    assert plan.tapunit_ok()
    xd = ops.as_nhwc(x.to(dev()), cpad=plan.c0k)
    xp = F.elu(x) if pre == L.PRE_ELU else x
    xr, wr = xp.bfloat16().double(), w.bfloat16().double()
    xin = F.pad(xr, (pad,) * 4, mode="reflect") if pm else xr
    ref = F.conv2d(xin, wr, b.double(), stride=stride, padding=0 if pm else pad)
    den = F.conv2d(xin.abs(), wr.abs(), None, stride=stride, padding=0 if pm else pad) + 1e-30
    got = ops.conv(plan, xd, pre_op=pre, precision="bf16", ksplit=1)
    assert ops.last_conv_kernel() == 11, ops.last_conv_kernel()
    assert float(((got.cpu().double() - ref).abs() / den).max()) < (3e-5 if pre == L.PRE_ELU else 3e-6)
    os.environ["FUSG_NO_BF16_TAPUNIT"] = "1"
    try:
        old = ops.conv(plan, xd, pre_op=pre, precision="bf16", ksplit=1)
        assert ops.last_conv_kernel() == 4
    finally:
        del os.environ["FUSG_NO_BF16_TAPUNIT"]
    exact = F.conv2d(F.pad(xp.double(), (pad,) * 4, mode="reflect") if pm else xp.double(), w.double(), b.double(), stride=stride,
                     padding=0 if pm else pad)
    assert float((old.cpu().double() - exact).abs().max() / exact.abs().max()) < 1e-5        # (the split-fp16 result: fp32-class)


def test_splitk_in_launch_combine_equals_reduce_kernel(precision, monkeypatch):
    """The opt-in last-arriver combine of split-K launches (csrc/conv_kernel.h, splitk_arrive / splitk_combine; off by
    default because it measured slower than the extra launch, ops.py) against the separate reduce kernel: bit-identical (same slab order), on every launch of a long, unevenly loaded sequence - a big
    convolution keeps a second stream busy meanwhile, slab memory and counters are reused from launch to launch (stale
    L1 / L2 lines and a wrong counter would show as a mismatch), every output word is compared."""
    shapes = [(4, 1024, 512, 3, 1, 2, 2), (32, 128, 128, 3, 1, 4, 4), (8, 256, 128, 1, 0, 8, 8), (2, 512, 128, 3, 1, 2, 2)]
    plans, xs = [], []
    for i, (B, cin, cout, k, pad, H, W) in enumerate(shapes):
        w = _rand(cout, cin, k, k, seed=40 + i, scale=1.0 / (cin * k * k) ** 0.5)
        plans.append(pack.pack_conv(w, _rand(cout, seed=50 + i), pad=pad))
        xs.append([_nhwc(_rand(B, cin, H, W, seed=60 + 10 * i + j)) for j in range(3)])
    wb = _rand(128, 128, 3, 3, seed=70, scale=0.03)
    big_plan, big_x = pack.pack_conv(wb, None, pad=1), _nhwc(_rand(8, 128, 128, 128, seed=71))
    side = torch.cuda.Stream()
    want = {}
    for i, plan in enumerate(plans):                                 # default: the separate reduce kernel
        for j, x in enumerate(xs[i]):
            want[(i, j)] = ops.conv(plan, x).clone()
    monkeypatch.setenv("FUSG_SPLITK_IN_LAUNCH", "1")                 # opt-in: combine inside the launch
    torch.cuda.synchronize()
    bad = 0
    for it in range(60):
        if it % 3 == 0:
            with torch.cuda.stream(side):
                ops.conv(big_plan, big_x)
        i, j = it % len(plans), (it // 2) % 3
        got = ops.conv(plans[i], xs[i][j])
        bad += int(not torch.equal(got, want[(i, j)]))
    torch.cuda.synchronize()
    assert bad == 0
    assert int(ops._splitk_counters(dev()).abs().sum()) == 0          # every launch left its counters zero


# ---- fused hourglass Bottleneck (fusg_hg_bottleneck) -----------------------------------------------------------------
def _bneck_params(cin, seed=0, scale=1.0, planes=128):
    """A pre-activation Bottleneck as the hourglass packs it: bn1 affine, conv1 + bn2, conv2 + bn3, conv3."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)                                                       # noqa: E731
    P = planes
    s1, h1 = torch.rand(cin, generator=g) + 0.5, rn(cin) * 0.2
    w1, b1 = rn(P, cin, 1, 1) * scale / cin ** 0.5, rn(P) * 0.1
    w2, b2 = rn(P, P, 3, 3) * scale / (9 * P) ** 0.5, rn(P) * 0.1
    w3, b3 = rn(2 * P, P, 1, 1) * scale / P ** 0.5, rn(2 * P) * 0.1
    p = {"pre": (s1.to(dev()), h1.to(dev())), "c1": pack.pack_conv(w1, b1).to(dev()),
         "c2": pack.pack_conv(w2, b2, pad=1).to(dev()), "c3": pack.pack_conv(w3, b3).to(dev()), "ds": None}
    return p, (s1, h1, w1, b1, w2, b2, w3, b3)


def _bneck_ref(x, res, prm):
    s1, h1, w1, b1, w2, b2, w3, b3 = (t.double() for t in prm)
    t = F.relu(x.double() * s1.view(1, -1, 1, 1) + h1.view(1, -1, 1, 1))
    t = F.relu(F.conv2d(t, w1, b1))
    t = F.relu(F.conv2d(t, w2, b2, padding=1))
    return F.conv2d(t, w3, b3) + res.double()


def _bneck_unfused(p, x, res, precision="f16x3"):
    t = ops.conv(p["c1"], x, pre_op=L.PRE_AFFINE_RELU, pre=p["pre"], act=L.ACT_RELU, precision=precision)
    t = ops.conv(p["c2"], t, act=L.ACT_RELU, precision=precision)
    return ops.conv(p["c3"], t, res0=res, precision=precision)


@pytest.mark.parametrize("B,cin,H,W,own_res,planes", [(2, 256, 8, 8, False, 128), (3, 256, 4, 4, False, 128),
                                                        (2, 256, 16, 16, False, 128), (1, 256, 64, 64, False, 128),
                                                        (2, 128, 32, 32, True, 128), (1, 256, 12, 20, False, 128),
                                                        (2, 64, 5, 9, True, 128), (2, 64, 32, 32, True, 64),
                                                        (1, 128, 16, 24, False, 64), (1, 64, 128, 128, True, 64)])
def test_hg_bottleneck_fused(B, cin, H, W, own_res, planes, precision, monkeypatch):
    """One-launch Bottleneck against an fp64 reference and against the three launches it replaces (same arithmetic,
    different summation order): error relative to the output's largest magnitude <= 2e-6 (observed ~3e-7), and no
    worse than 2x the three-launch path's."""
    from conftest import record
    p, prm = _bneck_params(cin, seed=21, planes=planes)
    x = _rand(B, cin, H, W, seed=22)
    res = _rand(B, 2 * planes, H, W, seed=23) if own_res else x
    assert own_res or cin == 2 * planes
    ref = _bneck_ref(x, res, prm)
    xin = _nhwc(x)
    rin = _nhwc(res) if own_res else xin
    monkeypatch.setattr(ops, "BNECK_MINHW", 0)          # (by default small batches run the low levels as three launches: ops.bottleneck_ok)
    assert ops.bottleneck_ok(p, xin)
    monkeypatch.setattr(ops, "BNECK_MINHW", None)
    assert ops.bottleneck_ok(p, xin) == (precision == "f32" or B > 8 or max(H, W) >= 32)
    got = ops.bottleneck(p, xin, rin)                   # (f32: the exact-fp32 form of the block, round 4)
    assert ops.last_conv_kernel() == 6
    assert not ops.range_exceeded(dev())
    un = _bneck_unfused(p, xin, rin, precision)
    den = float(ref.abs().max())
    e_f = float((got.cpu().double() - ref).abs().max()) / den
    e_u = float((un.cpu().double() - ref).abs().max()) / den
    record("bneck_fused_rel_err", e_f)
    record("bneck_unfused_rel_err", e_u)
    assert e_f <= 2e-6 and e_f <= 2 * e_u + 1e-7, (e_f, e_u)


def test_hg_bottleneck_fused_range_and_scale(precision):
    """Range contract of the fused block: an input holding 1e5 raises the status word (on bn1's output), a conv1
    whose weights push the FIRST INTERMEDIATE past 2^15 raises it too, and operands 100x smaller than usual keep the
    fp32-class error."""
    if precision != "f16x3":
        pytest.skip("split-fp16 path only")
    p, prm = _bneck_params(256, seed=31)
    x = _rand(2, 256, 8, 8, seed=32)
    xin = _nhwc(x)
    ops.bottleneck(p, xin)
    assert not ops.range_exceeded(dev())
    xb = x.clone()
    xb[1, 7, 3, 3] = 1e5
    ops.bottleneck(p, _nhwc(xb))
    assert ops.range_exceeded(dev())
    p2, _ = _bneck_params(256, seed=31, scale=3e4)              # conv1's output ~ 2e4 * N(0, 1): many |t1| >= 2^15
    ops.bottleneck(p2, xin)
    assert ops.range_exceeded(dev())
    p3, prm3 = _bneck_params(256, seed=33, scale=1e-2)
    ref = _bneck_ref(x * 1e-2, x * 1e-2, prm3)
    got = ops.bottleneck(p3, _nhwc(x * 1e-2))
    assert not ops.range_exceeded(dev())
    t = F.relu(x.double() * 1e-2 * prm3[0].double().view(1, -1, 1, 1) + prm3[1].double().view(1, -1, 1, 1))
    assert float((got.cpu().double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max()), float(t.abs().max())


def test_hg_bottleneck_fused_random_shapes(precision):
    """Seeded sweep over image sizes that do not divide into 8 x 8 patches, batch sizes and input widths: the fused
    block equals the three launches to 1e-6 of the output's range everywhere (same arithmetic, different order)."""
    rng = np.random.default_rng(5)
    for it in range(16):
        cin = int(rng.choice([32, 64, 128, 256, 288]))
        B, H, W = int(rng.integers(1, 5)), int(rng.integers(1, 40)), int(rng.integers(1, 40))
        planes = int(rng.choice([64, 128]))
        p, _ = _bneck_params(cin, seed=100 + it, planes=planes)
        x = _rand(B, cin, H, W, seed=200 + it)
        res = _rand(B, 2 * planes, H, W, seed=300 + it)
        xin, rin = _nhwc(x), _nhwc(res)
        got = ops.bottleneck(p, xin, rin)
        un = _bneck_unfused(p, xin, rin, precision)
        err = float((got - un).abs().max() / un.abs().max())
        assert err < 1e-6, (cin, B, H, W, err)
    assert not ops.range_exceeded(dev())


def test_hg_bottleneck_fused_is_deterministic(precision):
    """Race check of the fused block's LDS hand-offs (conv1 -> T -> conv2 -> U -> conv3, T and U overlaid): 100 launches
    on a grid that fills the chip several times over, interleaved with a kernel that dirties LDS-sized state, must give
    bit-identical outputs."""
    for planes, cin, hw in ((128, 256, 64), (64, 64, 96)):
        p, _ = _bneck_params(cin, seed=41, planes=planes)
        x = _nhwc(_rand(8, cin, hw, hw, seed=42))
        r = _nhwc(_rand(8, 2 * planes, hw, hw, seed=43))
        first = ops.bottleneck(p, x, r).clone()
        for it in range(100):
            if it % 10 == 0:
                _bneck_unfused(p, x, r)                      # other kernels in between (different LDS contents)
            got = ops.bottleneck(p, x, r)
            assert torch.equal(got, first), (planes, it)


# ---- the small-image kernel (csrc/conv_kernel_small.h): output images of <= 64 pixels -------------------------------
SMALL = [  # B, c0, c1, cout, k, stride, H, W, pre_op, residual, store
    (32, 128, 0, 128, 3, 1, 8, 8, "elu", True, "normal"),        # VUnet Residual at 8 x 8: 32 rows = half an image
    (32, 128, 0, 128, 3, 1, 4, 4, "elu", True, "normal"),        # two whole images per workgroup
    (32, 512, 512, 512, 3, 1, 2, 2, "elu", True, "normal"),      # AR block residual_k: eight images per workgroup, K ranges + reduce
    (32, 512, 0, 128, 3, 1, 2, 2, "none", False, "normal"),      # AR block sampler
    (5, 128, 128, 128, 3, 1, 4, 4, "elu", True, "normal"),       # ragged: M = 80, the last workgroup half empty; two sources
    (1, 128, 0, 128, 3, 1, 4, 4, "elu", True, "normal"),         # the reference's batch 1: M = 16
    (3, 128, 0, 512, 1, 1, 2, 2, "elu", False, "normal"),        # NiN 128 -> 512, M = 12
    (4, 256, 0, 128, 1, 1, 8, 8, "none", False, "normal"),
    (4, 128, 0, 128, 3, 2, 16, 16, "none", False, "normal"),     # DownSample 16 -> 8 (parity-quadrant weight order)
    (4, 128, 0, 128, 3, 2, 8, 8, "none", False, "normal"),       # 8 -> 4
    (4, 128, 0, 512, 3, 1, 4, 4, "none", False, "d2s"),          # UpSample('subpixel'): DepthToSpace store
    (4, 128, 0, 128, 3, 1, 8, 8, "elu", False, "s2d"),           # SpaceToDepth store
    (4, 256, 0, 128, 1, 1, 4, 4, "affine_relu", False, "normal"),   # hourglass Bottleneck conv1 (bn1 + ReLU on load)
    (4, 128, 0, 256, 1, 1, 4, 4, "relu", True, "normal"),        # hourglass Bottleneck conv3 + residual
    (2, 128, 0, 128, 3, 1, 4, 16, "elu", True, "normal"),        # non-square: Wo = 16, two image rows per workgroup
]


def _d2s(x):       # DCR: out[b, c, 2h+i, 2w+j] = in[b, (2i+j) C + c, h, w]
    b, c4, h, w = x.shape
    c = c4 // 4
    return x.view(b, 2, 2, c, h, w).permute(0, 3, 4, 1, 5, 2).reshape(b, c, 2 * h, 2 * w)


def _s2d(x):
    b, c, h, w = x.shape
    return x.view(b, c, h // 2, 2, w // 2, 2).permute(0, 3, 5, 1, 2, 4).reshape(b, 4 * c, h // 2, w // 2)


@pytest.mark.parametrize("B,c0,c1,cout,k,stride,H,W,pre,res,store", SMALL)
def test_small_image_kernel(B, c0, c1, cout, k, stride, H, W, pre, res, store, precision):
    """Every geometry the small-image kernel takes (rows of one image / whole images per workgroup, ragged M, two sources, K
    ranges over workgroups with the slab reduce, both weight slab orders, every pre-op, DepthToSpace / SpaceToDepth stores,
    residual) against torch (float64) on the CPU; on f16x3 the launch must really be that kernel, and it must agree with the
    launch it replaces (FUSG_NO_SMALL, read per call)."""
    cin = c0 + c1
    x0 = _rand(B, c0, H, W, seed=1)
    x1 = _rand(B, c1, H, W, seed=2) if c1 else None
    w = _rand(cout, cin, k, k, seed=3, scale=1.0 / (cin * k * k) ** 0.5)
    b = _rand(cout, seed=4)
    pad = k // 2
    plan = pack.pack_conv(w, b, c_split=(c0, c1) if c1 else None, stride=stride, pad=pad)
    xc = torch.cat([x0, x1], 1) if c1 else x0
    sc = sh = None
    if pre == "elu":
        xin, pre_op = F.elu(xc), L.PRE_ELU
    elif pre == "relu":
        xin, pre_op = F.relu(xc), L.PRE_RELU
    elif pre == "affine_relu":
        sc, sh = _rand(cin, seed=5), _rand(cin, seed=6)
        xin, pre_op = F.relu(xc * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)), L.PRE_AFFINE_RELU
    else:
        xin, pre_op = xc, L.PRE_NONE
    ref = F.conv2d(xin.double(), w.double(), b.double(), stride=stride, padding=pad).float()
    a0 = _nhwc(x0)
    kw = dict(pre_op=pre_op)
    if sc is not None:
        kw["pre"] = (sc.to(dev()), sh.to(dev()))
    if res:
        rr = x0 if cout == c0 and stride == 1 else _rand(*ref.shape, seed=7)
        ref = ref + rr
        kw["res0"] = a0 if rr is x0 else _nhwc(rr)
    if store == "d2s":
        ref, kw["store"] = _d2s(ref), L.STORE_D2S
    elif store == "s2d":
        ref, kw["store"] = _s2d(ref), L.STORE_S2D
    import os
    os.environ["FUSG_SMALL_KSPLIT"] = "1"          # K ranges over workgroups (+ slab reduce): off by default (measured slower), tested here
    try:
        got = ops.conv(plan, a0, _nhwc(x1) if c1 else None, **kw)
    finally:
        del os.environ["FUSG_SMALL_KSPLIT"]
    if precision == "f16x3":
        assert ops.last_conv_kernel() == 8, ops.last_conv_kernel()
        os.environ["FUSG_NO_SMALL"] = "1"                          # the launches it replaces: generic gather (+ split-K reduce) / halo kernel
        try:
            old = ops.conv(plan, a0, _nhwc(x1) if c1 else None, **kw)
            assert ops.last_conv_kernel() != 8
        finally:
            del os.environ["FUSG_NO_SMALL"]
        _close(got, old.cpu(), rtol=1e-5, atol=1e-5 * float(ref.abs().max()))
    assert tuple(got.shape) == tuple(ref.shape)
    _close(got, ref, rtol=2e-5, atol=2e-5 * float(ref.abs().max()))
    assert not ops.range_exceeded(dev())
    # into a channel slice of a wider buffer (the AR block's sampler means) and as a standard NCHW result
    if store == "normal" and not res:
        wide = ops.nhwc_empty(B, cout + 64, ref.shape[2], ref.shape[3], dev(), zero=True)
        ops.conv(plan, a0, _nhwc(x1) if c1 else None, out=wide, out_c_off=32, **kw)
        _close(wide[:, 32:32 + cout], ref, rtol=2e-5, atol=2e-5 * float(ref.abs().max()))
        assert float(wide[:, :32].abs().max()) == 0.0 and float(wide[:, 32 + cout:].abs().max()) == 0.0
        _close(ops.conv(plan, a0, _nhwc(x1) if c1 else None, nchw_out=True, **kw), ref, rtol=2e-5, atol=2e-5 * float(ref.abs().max()))


def test_small_image_kernel_range_guard_and_determinism(precision):
    if precision != "f16x3":
        pytest.skip("f16x3 only")
    x = _rand(8, 128, 4, 4, seed=1)
    w = _rand(128, 128, 3, 3, seed=2, scale=0.03)
    plan = pack.pack_conv(w, None, pad=1)
    a = _nhwc(x)
    y1 = ops.conv(plan, a, pre_op=L.PRE_ELU)
    assert ops.last_conv_kernel() == 8
    assert torch.equal(y1, ops.conv(plan, a, pre_op=L.PRE_ELU))
    assert not ops.range_exceeded(dev())
    x[3, 17, 2, 1] = 7.0e4                                        # outside the split's range: the status word, not a clamp
    ops.conv(plan, _nhwc(x))
    assert ops.range_exceeded(dev())


# ---- exact-fp32 mode of the halo kernel (csrc/conv_kernel_halo.h MODE 2, fusg_conv_desc.wfrag_f32) -------------------------
F32_HALO = [  # B, c0, c1, cout, k, stride, pad, dil, pad_mode, H, W, pre
    (2, 128, 0, 128, 3, 1, 1, 1, 0, 64, 64, "elu"),       # VUnet residual
    (1, 256, 0, 256, 3, 1, 2, 2, 1, 32, 32, "affine_relu"),   # EdgeConnect dilated block, reflect, IN-on-load
    (2, 64, 0, 128, 4, 2, 1, 1, 1, 32, 64, "none"),       # stride 2 in parity-quadrant form
    (1, 128, 0, 64, 5, 1, 2, 1, 1, 16, 32, "none"),       # 5 x 5
    (2, 32, 64, 32, 3, 1, 1, 1, 0, 16, 32, "elu"),        # two sources, 32-column tile (K split over the waves on this small grid)
    (2, 256, 0, 128, 1, 1, 0, 1, 0, 32, 32, "relu"),      # 1 x 1
]


@pytest.mark.parametrize("B,c0,c1,cout,k,stride,pad,dil,pm,H,W,pre", F32_HALO)
def test_f32_halo_kernel(B, c0, c1, cout, k, stride, pad, dil, pm, H, W, pre, precision):
    """precision="f32" on a halo-eligible layer runs v_mfma_f32_16x16x4_f32 from a staged halo (kernel family 9) and agrees
    with the generic fp32 gather it replaces (FUSG_NO_F32_HALO=1) to fp32 rounding - K runs chunk-major here, tap-major there,
    so not bit for bit - and is as close to an fp64 reference as that kernel is."""
    if precision != "f32":
        pytest.skip("f32 only")
    import os
    cin = c0 + c1
    x0 = _rand(B, c0, H, W, seed=1)
    x1 = _rand(B, c1, H, W, seed=2) if c1 else None
    w = _rand(cout, cin, k, k, seed=3, scale=1.0 / (cin * k * k) ** 0.5)
    b = _rand(cout, seed=4)
    plan = pack.pack_conv(w, b, c_split=(c0, c1) if c1 else None, stride=stride, pad=pad, dil=dil, pad_mode=pm)
    xc = torch.cat([x0, x1], 1) if c1 else x0
    kw = {}
    if pre == "elu":
        xin, kw["pre_op"] = F.elu(xc), L.PRE_ELU
    elif pre == "relu":
        xin, kw["pre_op"] = F.relu(xc), L.PRE_RELU
    elif pre == "affine_relu":
        sc, sh = _rand(B, cin, seed=5), _rand(B, cin, seed=6)
        xin = F.relu(xc * sc.view(B, cin, 1, 1) + sh.view(B, cin, 1, 1))
        kw.update(pre_op=L.PRE_AFFINE_RELU, pre=(sc.to(dev()), sh.to(dev())), pre_bstride=cin)
    else:
        xin = xc
    xp = F.pad(xin.double(), (pad,) * 4, mode="reflect") if pm else xin.double()
    ref = F.conv2d(xp, w.double(), b.double(), stride=stride, padding=0 if pm else pad, dilation=dil)
    a0, a1 = _nhwc(x0), (_nhwc(x1) if c1 else None)
    kw["ksplit"] = 1                              # (K whole: a grid this small would otherwise take the generic gather with split-K)
    got = ops.conv(plan, a0, a1, **kw)
    assert ops.last_conv_kernel() == 9, ops.last_conv_kernel()
    os.environ["FUSG_NO_F32_HALO"] = "1"
    try:
        old = ops.conv(plan, a0, a1, **kw)
        assert ops.last_conv_kernel() == 0
    finally:
        del os.environ["FUSG_NO_F32_HALO"]
    scale = float(ref.abs().max())
    e_new = float((got.cpu().double() - ref).abs().max()) / scale
    e_old = float((old.cpu().double() - ref).abs().max()) / scale
    assert e_new <= max(2 * e_old, 5e-7), (e_new, e_old)
    _close(got, old.cpu(), rtol=1e-5, atol=2e-6 * scale)
    assert torch.equal(got, ops.conv(plan, a0, a1, **kw))          # deterministic


@pytest.mark.parametrize("B,cin,cout,H,W", [(2, 256, 128, 16, 32), (3, 128, 64, 32, 32), (1, 64, 32, 8, 16)])
def test_conv_up2_ring_on_the_small_image_kernel(B, cin, cout, H, W, precision):
    """ICN decoder, Upsample(2) -> reflect 5x5: the outermost ring of the output as twelve 3x3 launches with border-regrouped
    weights (pack.pack_conv_up2_ring, 9 MACs per output instead of 25) - windows of one edge row / column / corner, replicate
    padding - which the small-image kernel takes as runs of 32 pixels of a window (round 4).  Equals torch, and (f16x3) the last
    ring launch really ran on that kernel."""
    x = _rand(B, cin, H, W, seed=1)
    w = _rand(cout, cin, 5, 5, seed=2, scale=1.0 / (cin * 25) ** 0.5)
    b = _rand(cout, seed=3)
    sc, sh = torch.rand(B, cin, generator=torch.Generator().manual_seed(4)) + 0.5, _rand(B, cin, seed=5) * 0.2
    exact = pack.pack_conv(w, b, pad=2, pad_mode=1, upsample=1)
    ring = {k: v.to(dev()) for k, v in pack.pack_conv_up2_ring(w, b).items()}
    xin = torch.relu(x * sc[:, :, None, None] + sh[:, :, None, None])
    ref = F.conv2d(F.pad(F.interpolate(xin, scale_factor=2, mode="nearest"), (2,) * 4, mode="reflect"), w, b)
    pre = (sc.to(dev()).contiguous(), sh.to(dev()).contiguous())
    xd = _nhwc(x)
    got = ops.conv_up2(exact, pack.pack_conv_up2_d2s(w, b), xd, pre_op=L.PRE_AFFINE_RELU, pre=pre, pre_bstride=cin, ring=ring)
    if precision == "f16x3":
        assert ops.last_conv_kernel() == 8, ops.last_conv_kernel()
    _close(got, ref)
    want = ops.conv_up2(exact, pack.pack_conv_up2_d2s(w, b), xd, pre_op=L.PRE_AFFINE_RELU, pre=pre, pre_bstride=cin)   # 25-tap windows
    _close(got, want.cpu(), rtol=1e-4, atol=1e-4)
