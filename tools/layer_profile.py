#!/usr/bin/env python3
"""Per-launch timing of every conv call site of the crop pass (analysis tool, GPU box).

Wraps ops.conv with HIP-event timing (one sync per launch: kernels run alone, so numbers are
upper bounds of what they cost back-to-back; shortest of three passes per launch) and prints one row per launch, slowest first,
with algorithmic TFLOP/s against the fp32 MFMA peak."""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from future_urban_scene_generation_amd import ops  # noqa: E402
from future_urban_scene_generation_amd.pipeline import VehiclePipeline, synth_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--inpaint", action="store_true")
    ap.add_argument("--top", type=int, default=60)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    pipe = VehiclePipeline(dev, inpaint=args.inpaint)
    batch = synth_batch(args.batch, args.res, dev, inpaint=args.inpaint)
    pipe.run(batch)
    torch.cuda.synchronize()
    rows = []
    orig = ops.conv
    net = ["?"]

    def timed(plan, x0, x1=None, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(plan, x0, x1, **kw)
        e1.record()
        e1.synchronize()
        b, c, h, w = x0.shape
        qh, qw = plan.out_hw(h, w)
        if kw.get("q_window") is not None:                         # one window of the output grid (conv_up2's ring)
            qh, qw = kw["q_window"][2], kw["q_window"][3]
        m = b * qh * qw
        fl = plan.flops_per_pixel * m
        kind = {0: "gen32", 1: "gen16", 2: "halo", 3: "halo-s2", 4: "tapunit", 5: "halo-bf16", 6: "bneck", 7: "pointwise"}.get(ops.last_conv_kernel(), "?")
        rows.append((net[0], f"{sum(plan.c_split)}->{plan.cout} k{plan.kh} s{plan.stride} d{plan.dil} up{plan.upsample} ph{plan.nphase} {kind}",
                     f"{h}x{w}", m, plan.k_pad, e0.elapsed_time(e1), fl))
        return out

    orig_b = ops.bottleneck

    def timed_b(p, x, res=None):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig_b(p, x, res)
        e1.record()
        e1.synchronize()
        b, c, h, w = x.shape
        m = b * h * w
        pl = p["c1"].cout
        rows.append((net[0], f"bottleneck {c}->{pl}->{pl}->{2 * pl} fused", f"{h}x{w}", m, c + 10 * pl, e0.elapsed_time(e1),
                     2.0 * m * (c * pl + 9 * pl * pl + 2 * pl * pl)))
        return out

    ops.bottleneck = timed_b
    ops.conv = timed
    import future_urban_scene_generation_amd.stacked_hourglass.models as m1
    import future_urban_scene_generation_amd.warp_learn.models as m2
    import future_urban_scene_generation_amd.vunet.models as m3
    import future_urban_scene_generation_amd.edgeconnect.networks as m4
    for mod in (m1, m2, m3, m4):
        mod.ops.conv = timed
    def one_pass():
        net[0] = "hg"
        pipe.hg(batch["hg_x"])
        net[0] = "icn"
        pipe.icn(batch["icn_x"])
        net[0] = "vunet"
        vu = pipe.vunet
        eo, es = vu.forward_enc_up(batch["vu_x"])
        mu, _ = vu.forward_enc_down(eo, es)
        do, ds = vu.forward_dec_up(batch["vu_y"])
        vu.forward_dec_down(do, ds, mu)
        if args.inpaint:
            net[0] = "edge"
            e = pipe.edge(batch["ec_gray"], batch["ec_edge"], batch["ec_mask"])
            net[0] = "inpaint"
            pipe.inp(batch["ec_img"], e, batch["ec_mask"])

    # three timed passes, per launch the shortest of the three (a single pass occasionally reads one launch tens of
    # milliseconds long - the first launch after the warm-up pass's synchronisation)
    passes = []
    for _ in range(3):
        del rows[:]
        one_pass()
        passes.append(list(rows))
    assert len({len(p) for p in passes}) == 1
    rows = [min(rs, key=lambda r: r[5]) for rs in zip(*passes)]
    tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for r in rows:
        tot[r[0]][0] += r[5]
        tot[r[0]][1] += r[6]
        tot[r[0]][2] += 1
    print("per net:  net  launches  ms  executed_GFLOP  TFLOP/s   (executed = as launched: the ICN decoder's phase form does 2.8x fewer than the reference)")
    for k, (ms, fl, n) in tot.items():
        print(f"  {k:8s} {n:4d} {ms:9.3f} {fl / 1e9:10.1f} {fl / ms / 1e9:8.1f}")
    allms = sum(v[0] for v in tot.values())
    allfl = sum(v[1] for v in tot.values())
    print(f"  total    {len(rows):4d} {allms:9.3f} {allfl / 1e9:10.1f} {allfl / allms / 1e9:8.1f}")
    # group identical call sites
    grp = collections.OrderedDict()
    for r in rows:
        key = r[:5]
        g = grp.setdefault(key, [0, 0.0, 0.0])
        g[0] += 1
        g[1] += r[5]
        g[2] += r[6]
    print("\nnet      layer                              in      M        K_pad   n   ms_total  TFLOP/s  %time")
    for key, (n, ms, fl) in sorted(grp.items(), key=lambda kv: -kv[1][1])[: args.top]:
        print(f"{key[0]:8s} {key[1]:34s} {key[2]:8s} {key[3]:8d} {key[4]:6d} {n:3d} {ms:9.3f} {fl / ms / 1e9:8.1f} {100 * ms / allms:6.1f}")


if __name__ == "__main__":
    main()
