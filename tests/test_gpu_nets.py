"""GPU parity: the HIP-backed drop-in modules against the CPU oracle (same synthetic weights, same
seeded inputs) and against the committed golden vectors produced by the reference itself.

Bars (both contraction paths; only the summation order differs from the reference's CPU kernels).  Every
comparison records the error it observed (conftest.record -> profiles/r02_parity.json); the bars are ~10x the worst
observation of the round, not a generic tolerance:
  * heat-maps / generator outputs: max |diff| relative to the tensor's largest magnitude < TOL;
  * 64x64 heat-map argmax indices and get_maxima floats: bit-exact;
  * to_image-quantised uint8 images: |diff| <= 1 LSB and SSIM >= 0.999 (north_star bar).
"""
from argparse import Namespace
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["f32", "f16x3"], autouse=True)
def precision(request):
    """Every GPU parity test runs on both contraction paths: exact fp32 MFMA and split-fp16."""
    from future_urban_scene_generation_amd import ops as _ops
    old = _ops.PRECISION
    _ops.set_precision(request.param)
    yield request.param
    _ops.set_precision(old)

import oracle                                                              # noqa: E402
from conftest import load_golden, load_schema, record, synth_sd            # noqa: E402
from future_urban_scene_generation_amd import ops                          # noqa: E402
from future_urban_scene_generation_amd.edgeconnect.models import EdgeModel, InpaintingModel   # noqa: E402
from future_urban_scene_generation_amd.edgeconnect.networks import EdgeGenerator, InpaintGenerator   # noqa: E402
from future_urban_scene_generation_amd.stacked_hourglass.models import HourglassNet, get_maxima_device   # noqa: E402
from future_urban_scene_generation_amd.synth import schema_of, synth_inputs   # noqa: E402
from future_urban_scene_generation_amd.vunet.models import Vunet_fix_res   # noqa: E402
from future_urban_scene_generation_amd.warp_learn.models import G_Resnet   # noqa: E402

DEV = "cuda:0"
# Bars on raw network outputs, relative to the tensor's largest magnitude = ~10x the worst value observed on the MI355X
# in round 2, for either precision (profiles/r02_parity.json): hourglass 1.0e-6, VUnet 1.7e-6, EdgeConnect 1.3e-5, ICN
# 2.3e-5 (its fixtures are ill-conditioned on purpose: InstanceNorm of near-constant planes amplifies every rounding).
TOL_HG, TOL_VU, TOL_EC, TOL_ICN = 1e-5, 2e-5, 1.5e-4, 2.5e-4
TOL = TOL_ICN


def _rel(got, ref, what="rel_err"):
    got = got.detach().to("cpu").double()
    ref = torch.as_tensor(ref).double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    r = float((got - ref).abs().max() / (ref.abs().max() + 1e-12))
    record(what, r)
    return r


def _u8(got, ref):
    """uint8 images: records and returns (max |diff| in LSB, SSIM)."""
    d = int(np.abs(got.astype(int) - ref.astype(int)).max())
    s = float(oracle.ssim(got, ref))
    record("u8_max_diff", d)
    record("ssim_min", s, worst=min)
    return d, s


def _build(net):
    if net == "hg":
        m = HourglassNet(num_stacks=2, num_blocks=1, num_classes=12)
    elif net == "icn":
        m = G_Resnet(21)
    elif net == "vunet":
        m = Vunet_fix_res(Namespace(up_mode="subpixel", w_norm=True, drop_prob=0.2, vunet_256=True))
    elif net == "edge":
        m = EdgeGenerator()
    else:
        m = InpaintGenerator()
    assert list(schema_of(m.state_dict()).items()) == list(load_schema(net).items())
    m.load_state_dict(synth_sd(net))
    return m.to(DEV).eval()


_CACHE = {}


def model(net):
    if net not in _CACHE:
        _CACHE[net] = _build(net)
    return _CACHE[net]


@pytest.mark.parametrize("tag,B,R", [("hg_b1_r256", 1, 256), ("hg_b2_r128", 2, 128)])
def test_hourglass(tag, B, R):
    g = load_golden(tag)
    x = synth_inputs("hg", B, R)["x"]
    out = model("hg")(x.to(DEV))
    assert set(out.keys()) == {"heatmaps"} and len(out["heatmaps"]) == 2
    hm = out["heatmaps"]
    assert hm[1].is_contiguous() and tuple(hm[1].shape) == (B, 12, R // 4, R // 4)
    ref = oracle.hourglass_forward(synth_sd("hg"), x)["heatmaps"]
    assert _rel(hm[0], ref[0]) < TOL_HG and _rel(hm[1], ref[1]) < TOL_HG
    assert _rel(hm[1], g["hm1"]) < TOL_HG
    # integer contract: bit-exact argmax / get_maxima vs the reference's golden values
    idx = ops.argmax_hw(hm[1]).cpu().numpy().astype(np.int64)
    assert np.array_equal(idx, g["argmax"])
    assert np.array_equal(get_maxima_device(hm[1]), g["maxima"])
    # the reference's own post-processing applied to our output gives the same thing
    up = torch.nn.functional.interpolate(hm[-1].cpu(), (R, R))
    assert np.array_equal(oracle.get_maxima(up), g["maxima"])


@pytest.mark.parametrize("tag,B,R", [("icn_b1_r256", 1, 256), ("icn_b2_r64", 2, 64)])
def test_icn(tag, B, R):
    g = load_golden(tag)
    x = synth_inputs("icn", B, R)["x"]
    out = model("icn")(x.to(DEV))
    assert out.is_contiguous() and tuple(out.shape) == (B, 3, R, R)
    ref = oracle.icn_forward(synth_sd("icn"), x)
    assert _rel(out, ref) < TOL_ICN
    assert _rel(out, g["out"]) < TOL_ICN
    d, ss = _u8(ops.to_image_u8(out).cpu().numpy(), g["img_u8"])
    assert d <= 1 and ss >= 0.999


@pytest.mark.parametrize("tag,B,R", [("vunet_b1_r256", 1, 256), ("vunet_b2_r128", 2, 128)])
def test_vunet_traj_sequence(tag, B, R, manifest):
    g = load_golden(tag)
    vu = model("vunet")
    i = synth_inputs("vunet", B, R)
    torch.manual_seed(manifest["cases"][tag]["noise_seed"])
    eo, es = vu.forward_enc_up(i["x"].to(DEV))
    mu_app, z_app = vu.forward_enc_down(eo, es)
    do, ds = vu.forward_dec_up(i["y_tilde"].to(DEV))
    assert len(eo) == 2 and len(es) == 2 and len(do) == 1 and len(ds) == 14
    sums = np.array([float(t.cpu().double().sum()) for t in ds])
    np.testing.assert_allclose(sums, g["skip_sums"], rtol=2e-3, atol=float(np.max(g["skip_abs"])) * 1e-5)
    for k in (0, 5, 13):
        assert _rel(ds[k][:, :, :8, :8], g[f"skip{k}_corner"]) < TOL_VU
    xt, mu_s, z_s = vu.forward_dec_down(do, ds, mu_app)
    assert ds == []
    assert xt.is_contiguous()
    for name, t in [("enc_out0", eo[0]), ("enc_out1", eo[1]), ("enc_skip0", es[0]), ("enc_skip1", es[1]),
                    ("mu_app0", mu_app[0]), ("mu_app1", mu_app[1]), ("z_app0", z_app[0]), ("z_app1", z_app[1]),
                    ("dec_out", do[0]), ("x_tilde", xt), ("mu_s0", mu_s[0]), ("mu_s1", mu_s[1]),
                    ("z_s0", z_s[0]), ("z_s1", z_s[1])]:
        assert _rel(t, g[name]) < TOL_VU, name
    d, ss = _u8(ops.to_image_u8(xt).cpu().numpy(), g["img_u8"])
    assert d <= 1 and ss >= 0.999
    # later frame (appearance code reused)
    y2 = synth_inputs("vunet", B, R, 1)["y_tilde"]
    torch.manual_seed(manifest["cases"][tag]["later_seed"])
    do2, ds2 = vu.forward_dec_up(y2.to(DEV))
    assert _rel(vu.forward_dec_down(do2, ds2, mu_app)[0], g["x_tilde_later"]) < TOL_VU


def test_vunet_forward_entry_and_nchw_inputs(manifest):
    g = load_golden("vunet_b1_r256")
    vu = model("vunet")
    i = synth_inputs("vunet", 1, 256)
    torch.manual_seed(manifest["cases"]["vunet_b1_r256"]["fwd_seed"])
    xt, mu_app, mu_shape = vu(i["y_tilde"].to(DEV), i["x"].to(DEV))
    assert _rel(xt, g["fwd_x_tilde"]) < TOL_VU and _rel(mu_shape[0], g["fwd_mu_shape0"]) < TOL_VU
    # callers may hand back plain NCHW copies of the intermediate tensors
    torch.manual_seed(5)
    do, ds = vu.forward_dec_up(i["y_tilde"].to(DEV))
    ds_nchw = [t.contiguous() for t in ds]
    a = vu.forward_dec_down(do, ds, [m.contiguous() for m in mu_app])[0]
    torch.manual_seed(5)
    b = vu.forward_dec_down([do[0].contiguous()], ds_nchw, mu_app)[0]
    assert torch.equal(a, b)


@pytest.mark.parametrize("tag,B,R", [("ec_b1_r256", 1, 256), ("ec_b2_r64", 2, 64)])
def test_edgeconnect(tag, B, R):
    g = load_golden(tag)
    i = synth_inputs("edge", B, R)
    em, im = EdgeModel(None), InpaintingModel(None)
    em.generator.load_state_dict(synth_sd("edge"))
    im.generator.load_state_dict(synth_sd("inpaint"))
    em, im = em.to(DEV).eval(), im.to(DEV).eval()
    gray, edge, mask, img = (i[k].to(DEV) for k in ("gray", "edge", "mask", "img"))
    e = em(gray, edge, mask).detach()
    assert _rel(e, g["edge_out"]) < TOL_EC
    p = im(img, e, mask)
    assert _rel(p, g["inpaint_out"]) < TOL_EC
    d, ss = _u8(ops.merge_u8(p, img, mask).cpu().numpy(), g["merged_u8"])
    assert d <= 1 and ss >= 0.999
    # bare generators behave like the wrappers' generator
    m = i["mask"]
    e2 = model("edge")(torch.cat((i["gray"] * (1 - m) + m, i["edge"] * (1 - m), m), 1).to(DEV))
    assert _rel(e2, g["edge_out"]) < TOL_EC


def test_modules_refuse_cpu_and_training():
    hg = HourglassNet(2, 1, 12).eval()
    with pytest.raises(RuntimeError):
        hg(torch.zeros(1, 3, 64, 64))
    icn = model("icn")
    icn.train()
    try:
        with pytest.raises(RuntimeError):
            icn(torch.zeros(1, 21, 64, 64, device=DEV))
    finally:
        icn.eval()


def test_batch_consistency_icn():
    """A batch is the same as its samples run one by one (no cross-sample leakage in the stats)."""
    x = synth_inputs("icn", 3, 64)["x"].to(DEV)
    full = model("icn")(x)
    for b in range(3):
        one = model("icn")(x[b:b + 1])
        assert _rel(one, full[b:b + 1].cpu()) < 1e-4      # tile / chunk choices depend on B


def test_high_res_512(precision):
    """BASELINE configs[4] resolution: the same modules at 512x512 (all kernels are resolution-agnostic;
    Vunet_fix_res.forward asserts 256 like the reference, the four sub-calls do not)."""
    if precision != "f16x3":
        pytest.skip("one precision is enough at this size")
    x = synth_inputs("icn", 1, 512)["x"]
    out = model("icn")(x.to(DEV))
    ref = oracle.icn_forward(synth_sd("icn"), x)
    assert tuple(out.shape) == (1, 3, 512, 512) and _rel(out, ref) < TOL_ICN
    assert oracle.ssim(ops.to_image_u8(out).cpu().numpy(), oracle.to_image_u8(ref)) >= 0.999
    hx = synth_inputs("hg", 1, 512)["x"]
    hm = model("hg")(hx.to(DEV))["heatmaps"][-1]
    href = oracle.hourglass_forward(synth_sd("hg"), hx)["heatmaps"][-1]
    assert tuple(hm.shape) == (1, 12, 128, 128)
    assert np.array_equal(ops.argmax_hw(hm).cpu().numpy(), oracle.heatmap_argmax(href))


def test_clip_mode_matches_frame_by_frame(precision):
    """Batched clip mode (vehicles x frames in one pass) == the reference's per-vehicle, per-frame call
    sequence, given the same noise: frame f of vehicle v uses vehicle v's frame-0 appearance code."""
    if precision != "f16x3":
        pytest.skip("one precision is enough")
    from future_urban_scene_generation_amd.pipeline import VehiclePipeline, synth_clip
    V, F, R = 2, 3, 128
    pipe = VehiclePipeline(DEV)
    clip = synth_clip(V, F, R, DEV)
    torch.manual_seed(21)
    out = pipe.run_clip(clip)
    assert tuple(out["vunet_u8"].shape) == (V, F, R, R, 3) and tuple(out["icn_u8"].shape) == (V, F, R, R, 3)
    # oracle, same batched noise stream: enc (V draws), then one dec_down over V*F samples
    sd = synth_sd("vunet")
    cpu = {k: v.cpu() for k, v in clip.items()}
    torch.manual_seed(21)
    eo, es = oracle.vunet_enc_up(sd, cpu["vu_x"])
    mu_app, _ = oracle.vunet_enc_down(sd, eo, es)
    do, ds = oracle.vunet_dec_up(sd, cpu["vu_y"].reshape(V * F, 3, R, R))
    xt = oracle.vunet_dec_down(sd, do, ds, [m.repeat_interleave(F, dim=0) for m in mu_app])[0]
    ref = oracle.to_image_u8(xt).reshape(V, F, R, R, 3)
    got = out["vunet_u8"].cpu().numpy()
    assert np.abs(got.astype(int) - ref.astype(int)).max() <= 1 and oracle.ssim(got.reshape(-1, R, R, 3), ref.reshape(-1, R, R, 3)) >= 0.999
    icn_ref = oracle.to_image_u8(oracle.icn_forward(synth_sd("icn"), cpu["icn_x"].reshape(V * F, 21, R, R))).reshape(V, F, R, R, 3)
    assert np.abs(out["icn_u8"].cpu().numpy().astype(int) - icn_ref.astype(int)).max() <= 1


def test_sharded_run_equals_unsharded(precision):
    """with one noise stream per vehicle (seed = base + global vehicle index) the images of a
    vehicle do not depend on which rank's batch it lands in - 4 vehicles in one pass == two shards of 2."""
    if precision != "f16x3":
        pytest.skip("one precision is enough")
    from future_urban_scene_generation_amd.pipeline import VehiclePipeline, shard_range, synth_batch
    n, R = 4, 128
    pipe = VehiclePipeline(DEV)
    batch = synth_batch(n, R, DEV)
    seeds = [1000 + i for i in range(n)]
    full = {k: v.cpu().numpy() for k, v in pipe.run(batch, vehicle_seeds=seeds).items()}
    again = {k: v.cpu().numpy() for k, v in pipe.run(batch, vehicle_seeds=seeds).items()}
    for k in full:
        assert np.array_equal(full[k], again[k]), k                       # same seeds -> same bits
    for r in range(2):
        lo, hi = shard_range(n, r, 2)
        part = pipe.run({k: v[lo:hi] for k, v in batch.items()}, vehicle_seeds=seeds[lo:hi])
        assert np.array_equal(part["kp_idx"].cpu().numpy(), full["kp_idx"][lo:hi])
        for k in ("icn_u8", "vunet_u8"):                                  # tile choices depend on the batch size
            assert np.abs(part[k].cpu().numpy().astype(int) - full[k][lo:hi].astype(int)).max() <= 1, k
    empty = pipe.run({k: v[:0] for k, v in batch.items()})               # a rank with no vehicles
    assert tuple(empty["kp_idx"].shape) == (0, 12) and tuple(empty["vunet_u8"].shape) == (0, R, R, 3)
    # and the default (reference) mode is untouched by a previous seeded run
    torch.manual_seed(3)
    a = pipe.run(batch)["vunet_u8"].cpu().numpy()
    torch.manual_seed(3)
    b = pipe.run(batch)["vunet_u8"].cpu().numpy()
    assert np.array_equal(a, b) and not np.array_equal(a, full["vunet_u8"])


@pytest.mark.parametrize("inpaint", [False, True], ids=["cfg1", "cfg2_inpaint"])
def test_full_size_batch_permutation_equivariance(precision, inpaint):
    """BASELINE configs[1] and configs[2] (--inpaint) at full size (B=32, 256x256), where the CPU oracle is too slow to be the checker: a
    crop's result must not depend on its position in the batch - permuting the vehicles permutes the outputs,
    bit for bit (per-vehicle noise streams follow their vehicle).  Catches cross-sample leakage in tiles, fused
    statistics and stream hand-offs at the size the benchmark runs."""
    if precision != "f16x3":
        pytest.skip("one precision is enough at this size")
    from future_urban_scene_generation_amd.pipeline import VehiclePipeline, synth_batch
    B, R = 32, 256
    pipe = VehiclePipeline(DEV, inpaint=inpaint)
    batch = synth_batch(B, R, DEV, inpaint=inpaint)
    seeds = [500 + i for i in range(B)]
    out = {k: v.cpu().numpy() for k, v in pipe.run(batch, vehicle_seeds=seeds).items()}
    assert ("inpaint_u8" in out) == inpaint
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(9))
    pb = {k: v[perm.to(v.device)].contiguous() for k, v in batch.items()}
    pout = pipe.run(pb, vehicle_seeds=[seeds[i] for i in perm.tolist()])
    for k, v in pout.items():
        assert np.array_equal(v.cpu().numpy(), out[k][perm.numpy()]), k
    assert out["icn_u8"].std() > 10 and out["vunet_u8"].std() > 10          # not a degenerate image
    assert len({tuple(r) for r in out["kp_idx"].tolist()}) > 1


@pytest.mark.parametrize("inpaint", [False, True], ids=["cfg1", "cfg2_inpaint"])
def test_full_size_batch_against_the_oracle(precision, inpaint):
    """VERDICT r2 #5 / r3 #3b: BASELINE configs[1] AND configs[2] at FULL size (B=32, 256x256) against the CPU oracle itself
    (32 s / ~55 s of host time at 16 threads), with the reference's own noise
    contract: ONE `torch.randn(B, 128, h, w)` per sampler on the global CPU generator (vunet/layers.py:163-167), no
    per-vehicle seeds.  Keypoint indices bit-exact, uint8 images within 1 LSB, SSIM >= 0.999 (north_star's bar)."""
    if precision != "f16x3":
        pytest.skip("one precision is enough at this size")
    from future_urban_scene_generation_amd.pipeline import VehiclePipeline, synth_batch
    B, R = 32, 256
    nets = ("hg", "icn", "vunet") + (("edge", "inpaint") if inpaint else ())
    sds = {n: synth_sd(n) for n in nets}
    pipe = VehiclePipeline(DEV, inpaint=inpaint, state_dicts=sds)
    batch = synth_batch(B, R, "cpu", inpaint=inpaint, seed=5)
    torch.manual_seed(123)
    got = {k: v.cpu().numpy() for k, v in pipe.run({k: v.to(DEV) for k, v in batch.items()}).items()}
    nt = torch.get_num_threads()
    torch.set_num_threads(min(16, max(1, len(os.sched_getaffinity(0)))))
    try:
        torch.manual_seed(123)
        ref = oracle.crop_pass(sds, batch, inpaint)
    finally:
        torch.set_num_threads(nt)
    assert np.array_equal(got["kp_idx"], ref["kp_idx"])
    for k in ("icn_u8", "vunet_u8") + (("inpaint_u8",) if inpaint else ()):
        d = int(np.abs(got[k].astype(int) - ref[k].astype(int)).max())
        sv = oracle.ssim(got[k], ref[k])
        record(f"full_size_{k}_max_diff", d)
        record(f"full_size_{k}_ssim", sv)
        assert d <= 1 and sv >= 0.999, (k, d, sv)


@pytest.mark.parametrize("H,W", [(64, 96), (72, 88)])
def test_non_square_and_odd_tiles(H, W, precision):
    """Fully-convolutional behaviour on non-square inputs: 64x96 exercises the halo-tiled kernel with
    rectangular tile grids, 72x88 (not multiples of 8/16 after down-sampling) the generic gather."""
    g = torch.Generator().manual_seed(H * 1000 + W)
    x = torch.rand(2, 21, H, W, generator=g) * 2 - 1
    out = model("icn")(x.to(DEV))
    ref = oracle.icn_forward(synth_sd("icn"), x)
    assert tuple(out.shape) == (2, 3, H, W) and _rel(out, ref) < TOL
    img = torch.rand(2, 3, H, W, generator=g)
    gray = img.mean(1, keepdim=True)
    edge = (torch.rand(2, 1, H, W, generator=g) < 0.05).float()
    mask = torch.zeros(2, 1, H, W)
    mask[:, :, H // 4:H // 2, W // 3:2 * W // 3] = 1
    em = EdgeModel(None)
    em.generator.load_state_dict(synth_sd("edge"))
    em = em.to(DEV).eval()
    e = em(gray.to(DEV), edge.to(DEV), mask.to(DEV))
    assert _rel(e, oracle.edge_model_forward(synth_sd("edge"), gray, edge, mask)) < TOL


def test_hourglass_non_square(precision):
    x = synth_inputs("hg", 1, 256)["x"][:, :, :128, :192].contiguous()
    hm = model("hg")(x.to(DEV))["heatmaps"][-1]
    ref = oracle.hourglass_forward(synth_sd("hg"), x)["heatmaps"][-1]
    assert tuple(hm.shape) == (1, 12, 32, 48) and _rel(hm, ref) < TOL_HG
    assert np.array_equal(ops.argmax_hw(hm).cpu().numpy(), oracle.heatmap_argmax(ref))


def test_out_of_range_input_falls_back_to_exact_fp32(precision):
    """no silent saturation.  A network input holding 1e5 (outside the fp16 split's range) makes the
    first f16x3 launch raise the status word; the entry point then redoes the call on the exact-fp32 path, so the
    caller gets bit for bit what precision="f32" gives - for a plain forward and for the VUnet (noise rewound, the
    consumed `skips` list restored)."""
    if precision != "f16x3":
        pytest.skip("f16x3 only")
    x = synth_inputs("icn", 1, 64)["x"].clone()
    x[0, 3, 10, 10] = 1e5
    got = model("icn")(x.to(DEV))
    with ops.precision("f32"):
        want = model("icn")(x.to(DEV))
    assert torch.equal(got, want)
    assert not ops.range_exceeded(DEV)
    vu = model("vunet")
    i = synth_inputs("vunet", 1, 128)
    y = i["y_tilde"].clone()
    y[0, 1, 5, 5] = 7e4                                           # (positive: the first conv stages ELU(y))
    do, ds = vu.forward_dec_up(y.to(DEV))                         # flagged in its first conv -> redone in fp32
    with ops.precision("f32"):
        do32, ds32 = vu.forward_dec_up(y.to(DEV))
    assert torch.equal(do[0], do32[0]) and len(ds) == 14 and all(torch.equal(a, b) for a, b in zip(ds, ds32))
    # an entry point that consumes a list (`skips`) and CPU noise: poison its input so that the first pass is flagged
    # half-way; the repeat must see the full list again and the same noise
    do_bad = [do[0].clone()]
    do_bad[0][0, 5, 1, 1] = 9e4
    ds_a, ds_b = list(ds), list(ds)
    torch.manual_seed(3)
    xt = vu.forward_dec_down(do_bad, ds_a)[0]
    assert ds_a == []
    with ops.precision("f32"):
        torch.manual_seed(3)
        want = vu.forward_dec_down(do_bad, ds_b)[0]
    assert torch.equal(xt, want)
    assert not ops.range_exceeded(DEV)
    # the pipeline checks a whole pass (per-network checks deferred) and redoes it in fp32
    from future_urban_scene_generation_amd.pipeline import VehiclePipeline, synth_batch
    pipe = VehiclePipeline(DEV)
    batch = synth_batch(2, 128, DEV)
    batch["icn_x"][1, 0, 0, 0] = 1e6
    seeds = [5, 6]
    out = pipe.run(batch, vehicle_seeds=seeds)
    with ops.precision("f32"):
        want = pipe.run(batch, vehicle_seeds=seeds)
    for k in out:
        assert torch.equal(out[k], want[k]), k
    pipe.run(batch, vehicle_seeds=seeds, check="async")
    assert pipe.finish() is True and pipe.finish() is False


def test_grad_enabled_calls_and_detach():
    """traj_test calls the ICN / VUnet / EdgeConnect modules WITH grad enabled and detaches the result
    (trajectory_inference.py:124, planes_utils.py:105): the drop-ins must neither need no_grad nor return tensors
    that break .detach()/.cpu()/.numpy()."""
    with torch.enable_grad():
        x = synth_inputs("icn", 1, 64)["x"].to(DEV)
        out = model("icn")(x)
        assert not out.requires_grad
        a = out.detach().to("cpu").numpy()
        i = synth_inputs("vunet", 1, 128)
        vu = model("vunet")
        torch.manual_seed(1)
        eo, es = vu.forward_enc_up(i["x"].to(DEV))
        mu, _ = vu.forward_enc_down(eo, es)
        do, ds = vu.forward_dec_up(i["y_tilde"].to(DEV))
        xt = vu.forward_dec_down(do, ds, mu)[0]
        b = xt.detach().to("cpu").numpy()
        e = synth_inputs("edge", 1, 64)
        em = EdgeModel(None)
        em.generator.load_state_dict(synth_sd("edge"))
        em = em.to(DEV).eval()
        c = em(e["gray"].to(DEV), e["edge"].to(DEV), e["mask"].to(DEV)).detach().cpu().numpy()
    with torch.no_grad():
        a2 = model("icn")(x).cpu().numpy()
    assert np.array_equal(a, a2) and np.isfinite(b).all() and np.isfinite(c).all()
    for p in model("icn").parameters():
        assert p.requires_grad and p.grad is None


def test_edgeconnect_checkpoint_roundtrip(tmp_path):
    """EdgeConnect's checkpoint contract (edgeconnect/models.py:17-38): {'iteration', 'generator'} at
    <PATH>/<name>_gen.pth, read back by BaseModel.load() into a fresh model."""
    cfg = Namespace(PATH=str(tmp_path))
    src = InpaintingModel(cfg)
    src.generator.load_state_dict(synth_sd("inpaint"))
    src.iteration = 1234
    src.save()
    blob = torch.load(str(tmp_path / "InpaintingModel_gen.pth"), map_location="cpu", weights_only=True)
    assert set(blob.keys()) == {"iteration", "generator"} and list(blob["generator"].keys()) == list(load_schema("inpaint").keys())
    dst = InpaintingModel(cfg)
    dst.load()
    assert dst.iteration == 1234
    i = synth_inputs("edge", 1, 64)
    args = [i[k].to(DEV) for k in ("img", "edge", "mask")]
    assert torch.equal(src.to(DEV).eval()(*args), dst.to(DEV).eval()(*args))
    em = EdgeModel(Namespace(PATH=str(tmp_path / "absent")))
    before = {k: v.clone() for k, v in em.generator.state_dict().items()}
    em.load()                                                     # no file: the initial weights stay
    assert all(torch.equal(v, em.generator.state_dict()[k]) for k, v in before.items())


def test_compiled_pass_replays_the_eager_pass(precision):
    """A recorded pass (fusg_plan, pipeline.CompiledPass) gives bit for bit what the eager pass gives - on the inputs it
    was recorded with, on new inputs, with per-vehicle seeds and with the reference's global generator - and keeps the
    range guard."""
    from future_urban_scene_generation_amd.pipeline import VehiclePipeline, synth_batch
    B, R = 2, 128
    pipe = VehiclePipeline(DEV, inpaint=(precision == "f32"))                 # (one precision also covers EdgeConnect)
    b0 = synth_batch(B, R, DEV, inpaint=pipe.inpaint, seed=0)
    b1 = synth_batch(B, R, DEV, inpaint=pipe.inpaint, seed=1)
    cp = pipe.compile(b0, vehicle_seeds=[7, 8])
    assert cp.size > 200 and len(cp.rec.noise_slots) == 2
    for batch, seeds in ((b0, [7, 8]), (b1, [9, 10]), (b0, [7, 8])):
        want = {k: v.clone() for k, v in pipe.run(batch, vehicle_seeds=seeds).items()}
        got = cp.run(batch, vehicle_seeds=seeds)
        assert set(got) == set(want)
        for k in want:
            assert torch.equal(got[k], want[k]), k
    torch.manual_seed(4)
    want = {k: v.clone() for k, v in pipe.run(b1).items()}
    torch.manual_seed(4)
    got = cp.run(b1)
    assert all(torch.equal(got[k], want[k]) for k in want)
    # the same recording issued by one host thread per recorded stream (fusg_plan_run_mt; FUSG_PLAN_MT=1 makes it the default)
    from future_urban_scene_generation_amd import _lib as L, pipeline as pl
    assert L.lib().fusg_plan_streams(cp.rec.handle) >= 3                     # hourglass / ICN / VUnet branches (+ the VUnet's shape encoder)
    old = pl.PLAN_THREADS
    pl.PLAN_THREADS = True
    try:
        for rep in range(3):                                                  # (back to back: the worker threads are reused)
            for batch, seeds in ((b1, [9, 10]), (b0, [7, 8])):
                want = {k: v.clone() for k, v in pipe.run(batch, vehicle_seeds=seeds).items()}
                got = cp.run(batch, vehicle_seeds=seeds)
                for k in want:
                    assert torch.equal(got[k], want[k]), (k, rep)
    finally:
        pl.PLAN_THREADS = old
    with pytest.raises(ValueError):
        cp.run({k: v[:1] for k, v in b0.items()})
    if precision == "f16x3":
        bad = {k: v.clone() for k, v in b1.items()}
        bad["vu_x"][0, 0, 3, 3] = 5e4
        got = cp.run(bad, vehicle_seeds=[1, 2])
        with ops.precision("f32"):
            want = pipe.run(bad, vehicle_seeds=[1, 2])
        assert all(torch.equal(got[k], want[k]) for k in want)
        assert not ops.range_exceeded(DEV)


def test_compiled_pass_survives_a_larger_eager_pass_and_refuses_stale_weights(precision):
    """a recorded pass bakes raw device pointers.  (1) The split-K workspace it was recorded with must outlive
    a later, larger eager pass on the same stream that replaces ops._WS[...] (the old block used to go back to the
    caching allocator while the plan kept writing partial sums into it): compile(B=1), eager run(B=16), then churn the
    allocator - the replay must still equal the eager pass bit for bit.  (2) After a network's parameters changed
    (load_state_dict -> refresh) the plan points into freed packed weights: run() must refuse."""
    if precision != "f16x3":
        pytest.skip("one precision is enough")
    from future_urban_scene_generation_amd.pipeline import VehiclePipeline, synth_batch
    pipe = VehiclePipeline(DEV)
    b1 = synth_batch(1, 128, DEV, seed=3)
    cp = pipe.compile(b1, vehicle_seeds=[5])
    want = {k: v.clone() for k, v in pipe.run(b1, vehicle_seeds=[5]).items()}
    big = synth_batch(16, 128, DEV, seed=4)
    pipe.run(big, vehicle_seeds=list(range(16)))                  # grows the per-stream split-K workspaces
    torch.cuda.synchronize()
    junk = [torch.full((1 << 20,), float(i), device=DEV) for i in range(64)]   # reuse whatever the allocator got back
    got = cp.run(b1, vehicle_seeds=[5])
    for k in want:
        assert torch.equal(got[k], want[k]), k
    del junk
    pipe.icn.load_state_dict(pipe.icn.state_dict())               # same values, new packed-weight cache
    with pytest.raises(RuntimeError, match="parameters changed"):
        cp.run(b1, vehicle_seeds=[5])
    cp2 = pipe.compile(b1, vehicle_seeds=[5])
    got = cp2.run(b1, vehicle_seeds=[5])
    for k in want:
        assert torch.equal(got[k], want[k]), k


def test_async_range_flag_is_not_consumed_by_an_entry_point_call(precision):
    """the pipeline's passes report to the pipeline's own status word.  An out-of-range pass issued with
    check="async", then a clean module entry-point call (which reads and clears the DEVICE word), then finish(): the
    flag must still be there - and the entry point must not have repeated its own call in fp32 because of it."""
    if precision != "f16x3":
        pytest.skip("f16x3 only")
    from future_urban_scene_generation_amd.pipeline import VehiclePipeline, synth_batch
    pipe = VehiclePipeline(DEV)
    bad = synth_batch(1, 128, DEV, seed=1)
    bad["icn_x"][0, 0, 0, 0] = 1e6
    assert not pipe.finish()
    pipe.run(bad, vehicle_seeds=[1], check="async")
    x = synth_inputs("icn", 1, 64)["x"].to(DEV)
    y = model("icn")(x)                                          # entry point: reads the device word, which is clean
    with ops.precision("f32"):
        y32 = model("icn")(x)
    assert not torch.equal(y, y32)                                # it ran in f16x3, not a needless fp32 repeat
    assert pipe.finish() is True                                  # the async pass's flag survived
    assert pipe.finish() is False


def test_vunet_and_edgeconnect_at_512(precision):
    """BASELINE configs[4] resolution for the two networks test_high_res_512 does not cover: the VUnet's four entry
    points (Vunet_fix_res.forward itself asserts 256 like the reference) and EdgeConnect at 512x512 against the oracle."""
    if precision != "f16x3":
        pytest.skip("one precision is enough at this size")
    R = 512
    vu = model("vunet")
    i = synth_inputs("vunet", 1, R)
    torch.manual_seed(17)
    eo, es = vu.forward_enc_up(i["x"].to(DEV))
    mu_app, _ = vu.forward_enc_down(eo, es)
    do, ds = vu.forward_dec_up(i["y_tilde"].to(DEV))
    xt = vu.forward_dec_down(do, ds, mu_app)[0]
    sd = synth_sd("vunet")
    torch.manual_seed(17)
    reo, res_ = oracle.vunet_enc_up(sd, i["x"])
    rmu, _ = oracle.vunet_enc_down(sd, reo, res_)
    rdo, rds = oracle.vunet_dec_up(sd, i["y_tilde"])
    ref = oracle.vunet_dec_down(sd, rdo, rds, rmu)[0]
    assert tuple(xt.shape) == (1, 3, R, R) and _rel(xt, ref) < TOL_VU
    d, ss = _u8(ops.to_image_u8(xt).cpu().numpy(), oracle.to_image_u8(ref))
    assert d <= 1 and ss >= 0.999
    e = synth_inputs("edge", 1, R)
    em, im = EdgeModel(None), InpaintingModel(None)
    em.generator.load_state_dict(synth_sd("edge"))
    im.generator.load_state_dict(synth_sd("inpaint"))
    em, im = em.to(DEV).eval(), im.to(DEV).eval()
    ge = em(e["gray"].to(DEV), e["edge"].to(DEV), e["mask"].to(DEV))
    re_ = oracle.edge_model_forward(synth_sd("edge"), e["gray"], e["edge"], e["mask"])
    assert _rel(ge, re_, "rel_err_edge") < TOL_EC
    gp = im(e["img"].to(DEV), ge, e["mask"].to(DEV))
    rp = oracle.inpaint_model_forward(synth_sd("inpaint"), e["img"], re_, e["mask"])
    assert _rel(gp, rp, "rel_err_inpaint") < TOL_EC


def test_reduced_precision_evidence(precision):
    """Evidence for the precision decision of BASELINE configs[4] ("bf16 MFMA conv path"): the ICN / hourglass fixtures
    through the arithmetic of a single-pass bf16 contraction (operands rounded to 8 significant bits, exact products,
    fp32 accumulation: FUSG_PREC_EMU_BF16), of a two-piece bf16 split (16 bits) and of the shipped f16x3 and f32 paths.
    Records raw-output error, uint8 difference and SSIM against the reference's golden vectors (profiles/r02_parity.json);
    asserts what the decision rests on: the shipped paths meet the north_star bar (SSIM >= 0.999, keypoint indices
    exact) and the single-pass bf16 arithmetic is orders of magnitude less accurate than f16x3 on the same fixture."""
    if precision != "f16x3":
        pytest.skip("sweeps the precisions itself")
    g = load_golden("icn_b1_r256")
    x = synth_inputs("icn", 1, 256)["x"].to(DEV)
    gh = load_golden("hg_b1_r256")
    hx = synth_inputs("hg", 1, 256)["x"].to(DEV)
    res = {}
    for prec in ("emu_bf16", "emu_bf16x2", "f16x3", "f32"):
        with ops.precision(prec):
            out = model("icn")(x)
            hm = model("hg")(hx)["heatmaps"][-1]
        img = ops.to_image_u8(out).cpu().numpy()
        rel = float((out.cpu().double() - torch.as_tensor(g["out"]).double()).abs().max() / np.abs(g["out"]).max())
        kp_ok = int((ops.argmax_hw(hm).cpu().numpy().astype(np.int64) == gh["argmax"]).sum())
        res[prec] = (rel, int(np.abs(img.astype(int) - g["img_u8"].astype(int)).max()), float(oracle.ssim(img, g["img_u8"])), kp_ok)
        record(f"{prec}_icn_rel_err", rel)
        record(f"{prec}_icn_u8_max_diff", res[prec][1])
        record(f"{prec}_icn_ssim", res[prec][2], worst=min)
        record(f"{prec}_hg_keypoints_exact_of_12", kp_ok, worst=min)
    for prec in ("f16x3", "f32"):
        assert res[prec][2] >= 0.999 and res[prec][1] <= 1 and res[prec][3] == 12, (prec, res[prec])
    assert res["emu_bf16"][0] > 100 * res["f16x3"][0], res
    assert res["emu_bf16x2"][0] > res["f16x3"][0], res
    # the image networks of the rest of the path under single-pass bf16 arithmetic: VUnet first-frame and EdgeConnect
    gv, ge = load_golden("vunet_b1_r256"), load_golden("ec_b1_r256")
    vu, i = model("vunet"), synth_inputs("vunet", 1, 256)
    e = synth_inputs("edge", 1, 256)
    em, im = EdgeModel(None), InpaintingModel(None)
    em.generator.load_state_dict(synth_sd("edge"))
    im.generator.load_state_dict(synth_sd("inpaint"))
    em, im = em.to(DEV).eval(), im.to(DEV).eval()
    for prec in ("emu_bf16", "f16x3"):
        with ops.precision(prec):
            torch.manual_seed(manifest_seed("vunet_b1_r256"))
            eo, es = vu.forward_enc_up(i["x"].to(DEV))
            mu_app, _ = vu.forward_enc_down(eo, es)
            do, ds = vu.forward_dec_up(i["y_tilde"].to(DEV))
            xt = vu.forward_dec_down(do, ds, mu_app)[0]
            ed = em(e["gray"].to(DEV), e["edge"].to(DEV), e["mask"].to(DEV))
            pp = im(e["img"].to(DEV), ed, e["mask"].to(DEV))
            mu8 = ops.merge_u8(pp, e["img"].to(DEV), e["mask"].to(DEV)).cpu().numpy()
        vimg = ops.to_image_u8(xt).cpu().numpy()
        record(f"{prec}_vunet_rel_err", float((xt.cpu().double() - torch.as_tensor(gv["x_tilde"]).double()).abs().max() / np.abs(gv["x_tilde"]).max()))
        record(f"{prec}_vunet_u8_max_diff", int(np.abs(vimg.astype(int) - gv["img_u8"].astype(int)).max()))
        record(f"{prec}_vunet_ssim", float(oracle.ssim(vimg, gv["img_u8"])), worst=min)
        record(f"{prec}_inpaint_rel_err", float((pp.cpu().double() - torch.as_tensor(ge["inpaint_out"]).double()).abs().max() / np.abs(ge["inpaint_out"]).max()))
        record(f"{prec}_inpaint_u8_max_diff", int(np.abs(mu8.astype(int) - ge["merged_u8"].astype(int)).max()))
        record(f"{prec}_inpaint_ssim", float(oracle.ssim(mu8, ge["merged_u8"])), worst=min)


def manifest_seed(tag):
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "manifest.json")) as f:
        return json.load(f)["cases"][tag]["noise_seed"]


def test_bf16_path_meets_the_image_bar_and_keeps_keypoints_exact(precision):
    """BASELINE configs[4] ("bf16 MFMA conv path"): with precision "bf16" the image networks run their halo-kernel layers
    in single-pass bf16 - every uint8 output keeps SSIM >= 0.999 against the reference's golden vectors - while the
    hourglass stays on the fp32-class path, so keypoint indices remain bit-exact."""
    if precision != "f16x3":
        pytest.skip("one run")
    with ops.precision("bf16"):
        g = load_golden("icn_b1_r256")
        out = model("icn")(synth_inputs("icn", 1, 256)["x"].to(DEV))
        assert ops.last_conv_kernel() in (1, 2, 3, 4, 5)
        d, ss = _u8(ops.to_image_u8(out).cpu().numpy(), g["img_u8"])
        record("icn_rel_err", float((out.cpu().double() - torch.as_tensor(g["out"]).double()).abs().max() / np.abs(g["out"]).max()))
        assert ss >= 0.999 and d <= 16
        gh = load_golden("hg_b1_r256")
        hm = model("hg")(synth_inputs("hg", 1, 256)["x"].to(DEV))["heatmaps"][-1]
        assert np.array_equal(ops.argmax_hw(hm).cpu().numpy().astype(np.int64), gh["argmax"])
        assert _rel(hm, gh["hm1"], "hg_rel_err") < TOL_HG                                    # (f16x3 inside)
        gv = load_golden("vunet_b1_r256")
        vu, i = model("vunet"), synth_inputs("vunet", 1, 256)
        torch.manual_seed(manifest_seed("vunet_b1_r256"))
        eo, es = vu.forward_enc_up(i["x"].to(DEV))
        mu_app, _ = vu.forward_enc_down(eo, es)
        do, ds = vu.forward_dec_up(i["y_tilde"].to(DEV))
        xt = vu.forward_dec_down(do, ds, mu_app)[0]
        d, ss = _u8(ops.to_image_u8(xt).cpu().numpy(), gv["img_u8"])
        assert ss >= 0.999 and d <= 16
        ge = load_golden("ec_b1_r256")
        e = synth_inputs("edge", 1, 256)
        em, im = EdgeModel(None), InpaintingModel(None)
        em.generator.load_state_dict(synth_sd("edge"))
        im.generator.load_state_dict(synth_sd("inpaint"))
        em, im = em.to(DEV).eval(), im.to(DEV).eval()
        ed = em(e["gray"].to(DEV), e["edge"].to(DEV), e["mask"].to(DEV))
        pp = im(e["img"].to(DEV), ed, e["mask"].to(DEV))
        d, ss = _u8(ops.merge_u8(pp, e["img"].to(DEV), e["mask"].to(DEV)).cpu().numpy(), ge["merged_u8"])
        assert ss >= 0.999 and d <= 16
        # the whole pass (pipeline) at 512x512, the resolution of configs[4]
        from future_urban_scene_generation_amd.pipeline import VehiclePipeline, synth_batch
        pipe = VehiclePipeline(DEV)
        batch = synth_batch(2, 512, DEV)
        got = pipe.run(batch, vehicle_seeds=[3, 4])
    want = pipe.run(batch, vehicle_seeds=[3, 4])                                         # f16x3
    assert torch.equal(got["kp_idx"], want["kp_idx"])
    for k in ("icn_u8", "vunet_u8"):
        a, b = got[k].cpu().numpy(), want[k].cpu().numpy()
        record(f"pipe512_{k}_ssim_vs_f16x3", float(oracle.ssim(a, b)), worst=min)
        assert oracle.ssim(a, b) >= 0.999
    # ... and against the CPU ORACLE at configs[4]'s own shape (512 x 512; VERDICT r3 #3c): each vehicle rendered alone with the
    # global generator seeded like its stream (same draws, see test_config3_*): keypoints exact, SSIM >= 0.999
    sds = {k: synth_sd(k) for k in ("hg", "icn", "vunet")}
    cpu = {k: v.cpu() for k, v in batch.items()}
    for v, seed in enumerate((3, 4)):
        torch.manual_seed(seed)
        ref = oracle.crop_pass(sds, {k: t[v:v + 1] for k, t in cpu.items()})
        assert np.array_equal(got["kp_idx"][v:v + 1].cpu().numpy(), ref["kp_idx"]), v
        for k in ("icn_u8", "vunet_u8"):
            a = got[k][v:v + 1].cpu().numpy()
            sv = float(oracle.ssim(a, ref[k]))
            record(f"bf16_512_{k}_ssim_vs_oracle", sv, worst=min)
            record(f"bf16_512_{k}_u8_max_diff_vs_oracle", int(np.abs(a.astype(int) - ref[k].astype(int)).max()))
            assert sv >= 0.999, (v, k, sv)


def test_config3_frame_of_64_vehicles_in_8_shards(precision):
    """BASELINE configs[3] at full size on one card: a frame's 64 vehicles cut into the 8 contiguous shards that 8 ranks
    would take (pipeline.shard_range), each shard run as its own pass with its vehicles' global noise seeds, concatenated
    in vehicle order as gather_in_order does on rank 0 - equals the single 64-vehicle pass: keypoint indices bit for bit,
    images within 1 LSB (tile choices depend on the batch size)."""
    if precision != "f16x3":
        pytest.skip("one precision is enough at this size")
    from future_urban_scene_generation_amd.pipeline import VehiclePipeline, shard_range, synth_batch
    n, world, R = 64, 8, 256
    pipe = VehiclePipeline(DEV)
    frame = synth_batch(n, R, DEV)
    seeds = [1000 + i for i in range(n)]
    full = {k: v.cpu().numpy() for k, v in pipe.run(frame, vehicle_seeds=seeds).items()}
    parts = {k: [] for k in full}
    for r in range(world):
        lo, hi = shard_range(n, r, world)
        assert hi - lo == 8
        out = pipe.run({k: v[lo:hi] for k, v in frame.items()}, vehicle_seeds=seeds[lo:hi])
        for k in parts:
            parts[k].append(out[k].cpu().numpy())
    got = {k: np.concatenate(v, 0) for k, v in parts.items()}
    assert np.array_equal(got["kp_idx"], full["kp_idx"])
    for k in ("icn_u8", "vunet_u8"):
        assert got[k].shape == full[k].shape == (n, R, R, 3)
        d = int(np.abs(got[k].astype(int) - full[k].astype(int)).max())
        record(f"{k}_max_diff_sharded_vs_whole", d)
        assert d <= 1, k
    # ... and against the CPU oracle on 8 of the 64 vehicles (one per shard): the oracle renders vehicle v alone with the
    # global generator seeded like v's own stream - the same draws, because a vehicle's stream is consumed in the
    # reference's sampler order with the reference's shapes minus the batch axis
    sds = {k: synth_sd(k) for k in ("hg", "icn", "vunet")}
    cpu_frame = {k: v.cpu() for k, v in frame.items()}
    for v in range(3, n, 8):
        torch.manual_seed(seeds[v])
        ref = oracle.crop_pass(sds, {k: t[v:v + 1] for k, t in cpu_frame.items()})
        assert np.array_equal(full["kp_idx"][v:v + 1], ref["kp_idx"]), v
        for k in ("icn_u8", "vunet_u8"):
            d = int(np.abs(full[k][v:v + 1].astype(int) - ref[k].astype(int)).max())
            sv = oracle.ssim(full[k][v:v + 1], ref[k])
            record(f"cfg3_{k}_max_diff_vs_oracle", d)
            assert d <= 1 and sv >= 0.999, (v, k, d, sv)
