"""CPU: packed panels + k-tables reproduce torch's convolutions when run through an emulation of the
kernel's gather (every conv configuration that occurs on the hot path)."""
import pytest
import torch
import torch.nn.functional as F

from emu import emulate_conv, assemble_normal
from future_urban_scene_generation_amd import pack


def _rand(*s, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*s, generator=g)


CASES = [  # cin, cout, k, stride, pad, dil, pad_mode, H
    (8, 16, 1, 1, 0, 1, 0, 6),
    (12, 40, 3, 1, 1, 1, 0, 7),
    (12, 40, 3, 2, 1, 1, 0, 8),
    (3, 64, 7, 2, 3, 1, 0, 16),      # hourglass stem (cin not a multiple of 4)
    (21, 8, 7, 1, 3, 1, 1, 12),      # ICN stem, reflect
    (16, 8, 4, 2, 1, 1, 1, 8),       # ICN down, reflect
    (16, 8, 4, 2, 1, 1, 0, 8),       # EdgeConnect down, zero
    (8, 8, 3, 1, 2, 2, 1, 9),        # EdgeConnect dilated resblock conv
    (8, 4, 5, 1, 2, 1, 1, 6),        # ICN decoder 5x5
    (32, 12, 3, 1, 1, 1, 0, 2),      # tiny spatial
]


@pytest.mark.parametrize("cin,cout,k,stride,pad,dil,pm,H", CASES)
def test_conv_matches_torch(cin, cout, k, stride, pad, dil, pm, H):
    x = _rand(2, cin, H, H + 1, seed=1)
    w = _rand(cout, cin, k, k, seed=2) * 0.2
    b = _rand(cout, seed=3)
    plan = pack.pack_conv(w, b, stride=stride, pad=pad, dil=dil, pad_mode=pm)
    xin = F.pad(x, (pad,) * 4, mode="reflect") if pm else x
    ref = F.conv2d(xin, w, b, stride=stride, padding=0 if pm else pad, dilation=dil)
    out = assemble_normal(plan, emulate_conv(plan, x))
    assert out.shape == ref.shape
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)
    assert plan.k_pad % 32 == 0 and plan.cout_pad % 32 == 0


def test_concat_two_sources_with_elu():
    xa, xb = _rand(1, 12, 5, 5, seed=1), _rand(1, 6, 5, 5, seed=2)     # 6 -> padded to 8 K-channels
    w = _rand(10, 18, 3, 3, seed=3) * 0.2
    plan = pack.pack_conv(w, None, c_split=(12, 6), pad=1)
    ref = F.conv2d(F.elu(torch.cat([xa, xb], 1)), w, None, padding=1)
    out = assemble_normal(plan, emulate_conv(plan, xa, xb, pre="elu"))
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)
    assert plan.c0k == 12 and plan.c1k == 8


def test_upsample_fused_reflect():
    x = _rand(1, 8, 4, 5, seed=1)
    w = _rand(4, 8, 5, 5, seed=2) * 0.1
    plan = pack.pack_conv(w, None, pad=2, pad_mode=1, upsample=1)
    ref = F.conv2d(F.pad(F.interpolate(x, scale_factor=2, mode="nearest"), (2,) * 4, mode="reflect"), w)
    out = assemble_normal(plan, emulate_conv(plan, x))
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)


def test_up2_phase_weights_identity():
    """Upsample(2) -> ReflectionPad2d(2) -> 5x5 conv == four 3x3 convs of the low-res image with summed taps and
    edge-replicate padding, everywhere but on the outermost ring of output pixels (which ops.conv_up2 recomputes)."""
    x = _rand(2, 6, 7, 9, seed=1)
    w = _rand(5, 6, 5, 5, seed=2) * 0.1
    b = _rand(5, seed=3)
    ref = F.conv2d(F.pad(F.interpolate(x, scale_factor=2, mode="nearest"), (2,) * 4, mode="reflect"), w, b)
    got = torch.empty_like(ref)
    xp = F.pad(x, (1,) * 4, mode="replicate")
    for ph, wp in enumerate(pack.up2_phase_weights(w)):
        got[:, :, (ph >> 1)::2, (ph & 1)::2] = F.conv2d(xp, wp, b)
    torch.testing.assert_close(got[:, :, 1:-1, 1:-1], ref[:, :, 1:-1, 1:-1], rtol=1e-5, atol=1e-5)
    ring = torch.ones_like(ref, dtype=torch.bool)
    ring[:, :, 1:-1, 1:-1] = False
    assert (got - ref)[ring].abs().max() > 1e-3          # the ring really differs: it is not optional to redo it
    # the ConvPlans carry exactly these filters
    plans = pack.pack_conv_up2_phases(w, b)
    assert len(plans) == 4 and all(q.kh == 3 and q.pad == 1 and q.pad_mode == 2 and q.upsample == 0 for q in plans)


@pytest.mark.parametrize("k,pad_mode", [(4, "reflect"), (4, "zero"), (3, "zero"), (3, "reflect")])
def test_s2d_quadrant_form_of_stride2_conv(k, pad_mode):
    """A stride-2 pad-1 conv == the sum over the four parity sub-images x[2Y+i, 2X+j] of small stride-1 convs with
    the taps pack.s2d_quadrant_taps assigns to them; reflection of the full image becomes a clamp of the sub-image."""
    x = _rand(2, 5, 12, 16, seed=1)
    w = _rand(7, 5, k, k, seed=2) * 0.2
    xin = F.pad(x, (1,) * 4, mode="reflect") if pad_mode == "reflect" else F.pad(x, (1,) * 4)
    ref = F.conv2d(xin, w, stride=2)
    Ho, Wo = ref.shape[2:]
    got = torch.zeros_like(ref)
    quads = pack.s2d_quadrant_taps(k)
    assert sorted(pack.s2d_tap_order(k)) == list(range(k * k)) and sum(len(q) for q in quads) == k * k
    for q, taps in enumerate(quads):
        sub = x[:, :, (q >> 1)::2, (q & 1)::2]                        # [B, C, H/2, W/2]
        subp = F.pad(sub, (1,) * 4, mode="replicate") if pad_mode == "reflect" else F.pad(sub, (1,) * 4)
        for ky, kx, dy, dx in taps:
            win = subp[:, :, 1 + dy:1 + dy + Ho, 1 + dx:1 + dx + Wo]
            got += torch.einsum("bchw,oc->bohw", win, w[:, :, ky, kx])
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5)


def test_affine_relu_prologue_keeps_padding_zero():
    x = _rand(2, 8, 6, 6, seed=1)
    w = _rand(4, 8, 3, 3, seed=2)
    sc, sh = _rand(2, 8, seed=3), _rand(2, 8, seed=4)
    plan = pack.pack_conv(w, None, pad=1)
    ref = F.conv2d(F.relu(x * sc.view(2, 8, 1, 1) + sh.view(2, 8, 1, 1)), w, padding=1)
    out = assemble_normal(plan, emulate_conv(plan, x, pre="affine_relu", pre_scale=sc, pre_shift=sh))
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)


def test_conv_transpose_phases():
    x = _rand(2, 8, 5, 6, seed=1)
    w = _rand(8, 12, 4, 4, seed=2) * 0.2
    b = _rand(12, seed=3)
    plan = pack.pack_conv_transpose_k4s2p1(w, b)
    ref = F.conv_transpose2d(x, w, b, stride=2, padding=1)
    out = assemble_normal(plan, emulate_conv(plan, x))
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)


def test_folds():
    v, g = _rand(6, 4, 3, 3, seed=1), _rand(6, 1, 1, 1, seed=2).abs() + 0.5
    w = pack.fold_weight_norm(v, g)
    torch.testing.assert_close(w, g * v / v.reshape(6, -1).norm(dim=1).view(6, 1, 1, 1))
    # BN after conv fold
    x = _rand(1, 4, 5, 5, seed=3)
    cw, cb = _rand(6, 4, 1, 1, seed=4), _rand(6, seed=5)
    bw, bb, rm, rv = _rand(6, seed=6), _rand(6, seed=7), _rand(6, seed=8), _rand(6, seed=9).abs() + 0.5
    sc, sh = pack.bn_scale_shift(bw, bb, rm, rv)
    w2, b2 = pack.fold_bn_after_conv(cw, cb, sc, sh)
    ref = F.batch_norm(F.conv2d(x, cw, cb), rm, rv, bw, bb, False, 0.1, 1e-5)
    torch.testing.assert_close(F.conv2d(x, w2, b2), ref, rtol=1e-5, atol=1e-5)
    # spectral norm, both flattening conventions
    for tr in (False, True):
        wo = _rand(5, 7, 4, 4, seed=10)
        n_u = 7 if tr else 5
        u = F.normalize(_rand(n_u, seed=11), dim=0)
        vv = F.normalize(_rand(wo.numel() // n_u, seed=12), dim=0)
        wm = wo.permute(1, 0, 2, 3).reshape(7, -1) if tr else wo.reshape(5, -1)
        torch.testing.assert_close(pack.fold_spectral_norm(wo, u, vv, tr), wo / torch.dot(u, wm.mv(vv)))


def test_rowsplit_head_matches_conv():
    """kh x 1 GEMM + horizontal gather-sum == the full 7x7 reflect conv (small-cout heads)."""
    x = _rand(2, 8, 9, 10, seed=1)
    w = _rand(3, 8, 7, 7, seed=2) * 0.1
    b = _rand(3, seed=3)
    plan = pack.pack_conv_rowsplit(w, b, pad=3, pad_mode=1)
    t = assemble_normal(plan, emulate_conv(plan, x))                  # [B, 21, H, W]
    assert t.shape == (2, 21, 9, 10)
    W = 10
    out = torch.zeros(2, 3, 9, 10) + b.view(1, 3, 1, 1)
    for co in range(3):
        for kx in range(7):
            for xo in range(W):
                xx = xo + kx - 3
                xx = -xx if xx < 0 else (2 * W - 2 - xx if xx >= W else xx)
                out[:, co, :, xo] += t[:, co * 7 + kx, :, xx]
    ref = F.conv2d(F.pad(x, (3,) * 4, mode="reflect"), w, b)
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)


def test_f16x3_split_is_fp32_class():
    """The split-fp16 contraction a*w ~= ah*wh + ah*wl + al'*(wh*2^-11) (csrc/conv_kernel_h3.h): emulate it with exact
    fp64 products of the fp16 parts, over the operand scales of the GPU sweep (test_gpu_ops.py), and compare with plain
    fp16 and with fp32 rounding.  tools/emu_split.py is the long form of this test (fp32 fmaf chain, bf16 splits)."""
    g = torch.Generator().manual_seed(0)
    for sa in (1e-3, 1.0, 1e2):
        for sw in (2e-3, 1.0):
            a = torch.randn(64, 512, generator=g) * sa
            w = torch.randn(512, 64, generator=g) * sw
            exact = a.double() @ w.double()
            wsplit, inv = pack.split_f16x3(w.t().contiguous()[None])        # [1, 2, 64, 512] (hi, lo), [64]
            wh, wl = wsplit[0, 0].t().double(), wsplit[0, 1].t().double()
            # per-channel power-of-two scale, exact to undo; the largest weight of a channel sits in [2^13, 2^14)
            assert torch.equal(torch.frexp(inv)[0], torch.full_like(inv, 0.5))
            top = (wh.abs().amax(dim=0))
            assert bool(((top >= 2.0 ** 13) & (top <= 2.0 ** 14)).all())
            # (hi, lo) carry >= 22 significant bits of the channel's largest weight
            assert float((((wh + wl) * inv.double()) - w.double()).abs().max() / w.abs().max()) < 2.0 ** -21
            whs = (wh * 2.0 ** -11).half().double()                          # third term's B operand (v_pk_mul_f16)
            # kernel: hi = round toward zero to fp16, lo' = (a - hi) * 2^11 rounded to fp16
            hi_rn = a.half()
            over = hi_rn.float().abs() > a.abs()
            hi = torch.where(over, (hi_rn.view(torch.int16) - 1).view(torch.float16), hi_rn)
            assert bool((hi.float().abs() <= a.abs()).all())
            ah = hi.double()
            al = ((a.double() - ah) * 2048.0).float().half().double()
            x3 = (ah @ wh + ah @ wl + al @ whs) * inv.double()
            scale = (a.abs().double() @ w.abs().double())
            e3 = ((x3 - exact).abs() / scale).max().item()
            e1 = ((a.half().double() @ w.half().double() - exact).abs() / scale).max().item()
            e32 = (((a @ w).double() - exact).abs() / scale).max().item()
            assert e3 < 1e-7 and e3 < e1 / 200, (sa, sw, e3, e1, e32)       # ~2^-24 at every scale vs fp16's ~2^-11


# ---- the C-ABI packer (csrc/pack.hip) against pack.py, bit for bit (host code: runs without a GPU) ---------------
C_PACK_CASES = [  # cout, cin, k, c0, stride, pad, dil, upsample, cin_pad  -> expected wfrag_order
    (128, 128, 3, 128, 1, 1, 1, 0, 4, 0),        # halo kernel, plain order
    (256, 256, 3, 256, 1, 2, 2, 0, 4, 0),        # dilated
    (32, 96, 3, 32, 1, 1, 1, 0, 4, 0),           # two sources (32 + 64): VUnet Residual
    (128, 64, 4, 64, 2, 1, 1, 0, 4, 1),          # stride-2 k4: parity-quadrant order
    (128, 128, 3, 128, 2, 1, 1, 0, 4, 1),        # stride-2 k3
    (64, 21, 7, 21, 1, 3, 1, 0, 4, 2),           # ICN stem: tap-unit kernel (24 K-channels, unit 8)
    (64, 3, 7, 3, 2, 3, 1, 0, 4, 2),             # hourglass stem (4 K-channels, unit 4)
    (12, 256, 1, 256, 1, 0, 1, 0, 4, 0),         # score conv: cout padded to 32
    (40, 12, 3, 12, 1, 1, 1, 0, 4, 2),           # 12 channels, 3x3: tap-unit kernel
    (40, 44, 3, 44, 1, 1, 1, 0, 4, -1),          # generic gather only
    (64, 128, 5, 128, 1, 2, 1, 1, 4, 0),         # 5x5 behind the fused upsample
    (64, 21, 7, 21, 1, 3, 1, 0, 32, 0),          # few channels padded to 32 for the halo kernel
]


@pytest.mark.parametrize("cout,cin,k,c0,stride,pad,dil,ups,cin_pad,order", C_PACK_CASES)
def test_c_abi_packer_matches_pack_py(cout, cin, k, c0, stride, pad, dil, ups, cin_pad, order):
    import ctypes as C
    import numpy as np
    from future_urban_scene_generation_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(cout * 131 + cin)
    w = torch.randn(cout, cin, k, k, generator=g) * torch.logspace(-3, 1, cout).view(-1, 1, 1, 1)     # per-channel scales
    b = torch.randn(cout, generator=g)
    plan = pack.pack_conv(w, b, c_split=(c0, cin - c0) if c0 < cin else None, stride=stride, pad=pad, dil=dil,
                          upsample=ups, cin_pad=cin_pad)
    wsplit, wscale = pack.split_f16x3(plan.wpack)
    spec = L.PackSpec(cout=cout, cin=cin, kh=k, kw=k, c0=c0, stride=stride, pad=pad, dil=dil, upsample=ups, cin_pad=cin_pad)
    sz = L.PackSizes()
    L.check(lib.fusg_pack_conv_sizes(C.byref(spec), C.byref(sz)), "pack_conv_sizes")
    assert (sz.cout_pad, sz.k_pad, sz.c0k, sz.c1k) == (plan.cout_pad, plan.k_pad, plan.c0k, plan.c1k)
    assert sz.wfrag_order == order
    wpack = np.empty(sz.wpack_floats, np.float32)
    ktab = np.empty(sz.ktab_ints, np.int32)
    bias = np.empty(sz.cout_pad, np.float32)
    wh = np.empty(sz.wpack_h_halves, np.uint16)
    wsc = np.empty(sz.cout_pad, np.float32)
    wfrag = np.empty(max(1, sz.wfrag_halves), np.uint16)
    wc, bc = w.contiguous().numpy(), b.numpy()
    L.check(lib.fusg_pack_conv_weights(C.byref(spec), wc.ctypes.data, bc.ctypes.data, wpack.ctypes.data, ktab.ctypes.data,
                                       bias.ctypes.data, wh.ctypes.data, wsc.ctypes.data,
                                       wfrag.ctypes.data if sz.wfrag_halves else None), "pack_conv_weights")
    assert np.array_equal(wpack.reshape(plan.cout_pad, plan.k_pad), plan.wpack[0].numpy())
    assert np.array_equal(ktab.reshape(-1, 2), plan.ktab[0].numpy())
    assert np.array_equal(bias, plan.bias.numpy()) and np.array_equal(wsc, wscale.numpy())
    assert np.array_equal(wh, wsplit[0].view(torch.int16).numpy().astype(np.uint16).reshape(-1))
    # the fragment-order copy, as ConvPlan.to() builds it
    frag = pack.frag_f16x3(wsplit, plan)
    if frag is not None and plan.s2d_ok():
        frag = frag[pack.s2d_tap_order(plan.kh)].contiguous()
    if frag is None and plan.tapunit_ok():
        frag = pack.frag_tapunit(wsplit, plan)
    assert (frag is None) == (order < 0)
    if frag is not None:
        assert (1 if plan.s2d_ok() else (2 if plan.tapunit_ok() else 0)) == order
        assert np.array_equal(wfrag, frag.contiguous().view(torch.int16).numpy().astype(np.uint16).reshape(-1))


def test_up2_ring_weights_reproduce_the_reflect_padded_5x5_on_the_outermost_ring():
    """nn.Upsample(2) -> ReflectionPad2d(2) -> 5x5 on the outermost ring of output pixels == twelve low-res 3x3 convolutions
    with border-regrouped weights (pack.up2_border_weights / up2_ring_launches): the windows tile the ring exactly once and
    the values agree in float64; the fp32 weights are the float64 regrouping rounded by fixed-order fp32 sums."""
    import torch.nn.functional as F
    from future_urban_scene_generation_amd import pack
    torch.manual_seed(0)
    B, C, H, W, Co = 2, 3, 5, 6, 4
    x = torch.randn(B, C, H, W, dtype=torch.float64)
    w = torch.randn(Co, C, 5, 5, dtype=torch.float64)
    ref = F.conv2d(F.pad(F.interpolate(x, scale_factor=2, mode="nearest"), (2, 2, 2, 2), mode="reflect"), w)
    out = torch.zeros_like(ref)
    cnt = torch.zeros(2 * H, 2 * W)
    xp = F.pad(x, (1, 1, 1, 1), mode="replicate")
    for ry, rx, (Y0, X0, nY, nX), (py, px) in pack.up2_ring_launches(H, W):
        wb = torch.zeros(Co, C, 3, 3, dtype=torch.float64)
        for ky in range(5):
            for kx in range(5):
                wb[:, :, pack.UP2_ROWMAP[ry][ky], pack.UP2_ROWMAP[rx][kx]] += w[:, :, ky, kx]
        assert torch.allclose(pack.up2_border_weights(w.float(), ry, rx).double(), wb, atol=1e-5)
        full = F.conv2d(xp, wb)
        for Y in range(Y0, Y0 + nY):
            for X in range(X0, X0 + nX):
                out[:, :, 2 * Y + py, 2 * X + px] = full[:, :, Y, X]
                cnt[2 * Y + py, 2 * X + px] += 1
    ring = torch.zeros(2 * H, 2 * W, dtype=torch.bool)
    ring[0] = ring[-1] = True
    ring[:, 0] = ring[:, -1] = True
    assert torch.equal(cnt, ring.float())
    assert float((out[:, :, ring] - ref[:, :, ring]).abs().max()) < 1e-12
    assert sorted(pack.pack_conv_up2_ring(w.float(), None)) == sorted({(a, b) for a, b, _, _ in pack.up2_ring_launches(H, W)})


def test_frag_f32_is_the_lane_order_of_the_fp32_matrix_instruction():
    """fusg_conv_desc.wfrag_f32 (exact-fp32 halo kernel / fused Bottleneck, round 4): [tap][chunk][cout_pad/32][16-column half][h]
    [lane][4] with lane = g * 16 + column holding w[column][32 chunk + 16 h + 4 g + e] - what lane (column, g) feeds
    v_mfma_f32_16x16x4_f32 number (h, e) of a 32-channel chunk; stride-2 layers reorder the tap slabs like the fp16 copy."""
    import torch
    from future_urban_scene_generation_amd import pack
    g = torch.Generator().manual_seed(3)
    w = torch.randn(40, 96, 3, 3, generator=g)
    plan = pack.pack_conv(w, None, c_split=(32, 64), pad=1)
    f = pack.frag_f32(plan.wpack, plan)
    assert tuple(f.shape) == (9, 3, 2, 2, 2, 64, 4) and f.dtype == torch.float32
    for tap, chunk, nt, ct, h, lane, e in ((0, 0, 0, 0, 0, 0, 0), (4, 2, 1, 1, 1, 37, 3), (8, 1, 0, 1, 0, 63, 2), (5, 0, 1, 0, 1, 16, 1)):
        gq, col = lane >> 4, lane & 15
        n = nt * 32 + ct * 16 + col
        k = tap * 96 + chunk * 32 + 16 * h + 4 * gq + e
        assert float(f[tap, chunk, nt, ct, h, lane, e]) == float(plan.wpack[0, n, k])
    assert float(f[4, 2, 1, 1, 1, 37, 3]) == 0.0                                  # column 61 >= cout 40: zero padding
    s2 = pack.pack_conv(torch.randn(32, 32, 3, 3, generator=g), None, stride=2, pad=1)
    assert s2.s2d_ok() and pack.frag_f32(s2.wpack, s2) is not None
    assert pack.frag_f32(pack.pack_conv(torch.randn(8, 3, 7, 7, generator=g), None, pad=3).wpack, pack.pack_conv(torch.randn(8, 3, 7, 7, generator=g), None, pad=3)) is None
