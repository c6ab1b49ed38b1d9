#!/usr/bin/env python3
"""Headline benchmark: synthesised 256x256 vehicle crops / second on N MI355X (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One *step* = one pass of the hot path over one batch of synthetic crops per rank:
hourglass (+argmax) -> ICN -> VUnet first-frame (enc_up, enc_down, dec_up, dec_down) -> uint8
quantisation [+ EdgeGenerator + InpaintGenerator with --inpaint], then (N > 1) the gather of the
rendered uint8 crops to rank 0 over RCCL.  The default workload is BASELINE.json configs[1]:
batch 32 of 256x256 crops per GPU.  fp32 tensors (the reference's dtype) everywhere; the conv
contraction runs either as exact fp32 MFMA (--precision f32) or, by default, as split-fp16 MFMA with
fp32-class accuracy (f16x3) - both pass the same parity suite.
Inputs and weights are synthetic (no datasets / checkpoints in this environment) and resident in
HBM before the timed region; VUnet's sampler noise is drawn on the CPU generator inside the step,
as the reference does.  Weak scaling: each rank processes its own batch (vehicles are independent).

Extra fields: "roofline" (conv kernels: algorithmic FLOPs of SURVEY.md §8d divided by the kernels'
HIP-event time on the launch stream, measured live in a second pass of the same K steps),
"cpu_baseline" (the CPU oracle = a port of the reference's torch graph, timed on this host's cores
on a bounded sample of the same workload; N=1, rank 0 only), "ssim_vs_cpu_ref" / "kp_idx_exact"
(quality of the GPU path on that very sample) and "clip_mode" (secondary figure: 8 vehicles x 6 frames).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# algorithmic conv FLOPs per crop at 256x256 (2*MAC, measured on the reference)
GFLOP_PER_CROP = {"hg": 17.95, "icn": 130.12, "vunet_first": 76.27, "edge": 96.13, "inpaint": 97.37}
PEAK_F32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0       # MI355X_MICROARCH.md, dense fp16/bf16 MFMA peak (no sparsity)


class PowerSampler:
    """Socket power / shader clock of the card under test, sampled from sysfs hwmon while the timed steps run.
    The big layers run at the 1400 W board limit, so this is part of reading the number."""

    def __init__(self, device_index=0):
        import glob
        self.p = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input") +
                        glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average"))
        try:                                        # keep only the card torch is running on (PCI address match)
            pr = torch.cuda.get_device_properties(device_index)
            bdf = "%04x:%02x:%02x." % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            mine = [x for x in self.p if bdf in os.path.realpath(x.split("/hwmon/")[0])]
            if mine:
                self.p = mine[:1]
        except Exception:
            pass
        self.f = [os.path.join(os.path.dirname(x), "freq1_input") for x in self.p]
        self.samples, self._stop, self._th = [], False, None
        self.idle = [self._read(x) for x in self.p]

    @staticmethod
    def _read(path):
        try:
            return int(open(path).read().strip())
        except (OSError, ValueError):
            return 0

    def start(self):
        import threading

        def loop():
            while not self._stop:
                self.samples.append(([self._read(x) for x in self.p], [self._read(x) for x in self.f]))
                time.sleep(0.01)

        if self.p:
            self._th = threading.Thread(target=loop, daemon=True)
            self._th.start()

    def stop(self):
        self._stop = True
        if self._th is None:
            return None
        self._th.join()
        if not self.samples:
            return None
        self.samples = self.samples[len(self.samples) // 2:]                   # the sensor lags by up to ~1 s: steady half
        n = len(self.samples)
        avg = [sum(s[0][i] for s in self.samples) / n for i in range(len(self.p))]
        k = max(range(len(self.p)), key=lambda i: avg[i] - self.idle[i])       # the card whose power rose
        return {"avg_w": round(avg[k] / 1e6, 1), "max_w": round(max(s[0][k] for s in self.samples) / 1e6, 1),
                "sclk_mhz": round(sum(s[1][k] for s in self.samples) / n / 1e6), "samples": n,
                "source": "sysfs hwmon power1/freq1 of the card under test, 10 ms period, second half of warm-up + timed + roofline legs"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=15)
    ap.add_argument("--vehicles", type=int, default=0,
                    help="strong scaling: this many vehicles in total, sharded over the ranks (default: --batch per rank, weak)")
    ap.add_argument("--settle-s", type=float, default=0.6, help="seconds of untimed steps before the warm-up steps (DVFS settle)")
    ap.add_argument("--batch", type=int, default=32, help="crops per GPU per step")
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--inpaint", action="store_true", help="BASELINE configs[2]: add EdgeConnect")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=8, help="crops in the CPU baseline sample")
    ap.add_argument("--no-clip", action="store_true", help="skip the secondary clip-mode figure")
    ap.add_argument("--no-prof", action="store_true", help="skip the roofline leg (second pass with per-launch HIP events)")
    ap.add_argument("--precision", choices=["f16x3", "f32"], default=None,
                    help="conv contraction: f16x3 = split-fp16 MFMA, fp32-class accuracy (default); f32 = exact fp32 MFMA")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
    if os.environ.get("FUSG_DIST_BACKEND", "nccl") != "nccl":          # rehearsal: ranks share the cards that exist
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        # RCCL ("nccl" on ROCm).  FUSG_DIST_BACKEND=gloo rehearses the multi-rank code path where the ranks have
        # to share one card (development boxes): the gathers are then staged through host memory.
        backend = os.environ.get("FUSG_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from future_urban_scene_generation_amd import ops
    from future_urban_scene_generation_amd.pipeline import VehiclePipeline, gather_in_order, synth_batch
    if args.precision:
        ops.set_precision(args.precision)
    prec = ops.PRECISION

    torch.set_grad_enabled(False)
    pipe = VehiclePipeline(dev, inpaint=args.inpaint)
    if args.vehicles:
        # strong scaling (BASELINE configs[3]: one frame's vehicles sharded over the ranks): this rank's contiguous shard
        from future_urban_scene_generation_amd.pipeline import shard_range
        lo, hi = shard_range(args.vehicles, rank, world)
        args.batch, n_total, first = hi - lo, args.vehicles, lo
        full = synth_batch(args.vehicles, args.res, "cpu", inpaint=args.inpaint, seed=0)
        batch = {k: v[lo:hi].to(dev) for k, v in full.items()}
        del full
    else:
        batch = synth_batch(args.batch, args.res, dev, inpaint=args.inpaint, seed=rank)
        n_total, first = args.batch * world, rank * args.batch
    # VUnet noise: one stream per vehicle, seeded by the vehicle's global index, so that the images do not depend
    # on how the vehicles are spread over ranks
    seeds = [1000 + first + i for i in range(args.batch)]

    def step():
        out = pipe.run(batch, vehicle_seeds=seeds)
        if world > 1:                                   # the path's only exchange: crops -> rank 0
            crops = torch.cat([out["icn_u8"], out["vunet_u8"]], dim=-1)
            gather_in_order(crops, n_total)
            gather_in_order(out["kp_idx"], n_total)
        return out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sampler = PowerSampler() if (rank == 0 and world == 1) else None
    if sampler:
        sampler.start()
    # Clock settle, before (and on top of) the W warm-up steps: the card's DVFS needs ~0.5 s of load to reach its
    # steady operating point (a burst right after start-up reads 5-20 % slow or fast), and lazy
    # initialisation (plan upload, stream creation, RCCL communicators) happens in the first step.
    step()                                             # lazy initialisation
    torch.cuda.synchronize()
    settle = 1
    t_s = time.perf_counter()

    def keep_settling():
        # every rank must run the same number of steps (a step contains collectives): rank 0's clock decides
        go = torch.tensor([1 if time.perf_counter() - t_s < args.settle_s else 0], dtype=torch.int32,
                          device=dev if (world == 1 or dist.get_backend() == "nccl") else "cpu")
        if world > 1:
            dist.broadcast(go, src=0)
        return bool(go.item())

    while keep_settling():
        step()
        torch.cuda.synchronize()
        settle += 1
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    # Roofline leg: the SAME K steps again with one HIP-event pair around every conv launch, recorded on
    # the launch stream.  Kept out of the timed region above because ~460 event records per step cost
    # ~10 % of wall time; kernel durations themselves are unaffected (rocprofv3 agrees, profiles/).
    # The branches of the pass are serialised for this leg (FUSG_STREAMS=0): kernels that share the GPU with
    # another stream's kernels would each read longer than they are.
    prof = not args.no_prof
    if prof:
        streams_env = os.environ.get("FUSG_STREAMS")
        os.environ["FUSG_STREAMS"] = "0"
        step()
        barrier()
        ops.prof_reset()
        ops.prof_enable(True)
        for _ in range(args.steps):
            step()
        barrier()
        ops.prof_enable(False)
        if streams_env is None:
            del os.environ["FUSG_STREAMS"]
        else:
            os.environ["FUSG_STREAMS"] = streams_env
    power = sampler.stop() if sampler else None
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev if (world == 1 or dist.get_backend() == "nccl") else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    crops_per_s = n_total * args.steps / dt
    scale = (args.res / 256.0) ** 2
    gflop_crop = (GFLOP_PER_CROP["hg"] + GFLOP_PER_CROP["icn"] + GFLOP_PER_CROP["vunet_first"] +
                  ((GFLOP_PER_CROP["edge"] + GFLOP_PER_CROP["inpaint"]) if args.inpaint else 0.0)) * scale

    roofline = None
    if prof:
        conv_ms, conv_launches, _ = ops.prof_read(0)
        alg_flops = gflop_crop * 1e9 * args.batch * args.steps          # this rank's conv work in the timed region
        achieved = alg_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(REPO, "profiles", "hbm_traffic_latest.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("conv_bytes_per_launch")
            except Exception:
                traffic = None
        if prec == "f32":
            kern, peak, note = "fusg::conv_igemm_f32 (all tile instantiations)", PEAK_F32_MFMA_TFLOPS, \
                "fp32 MFMA: 1 matrix FLOP per algorithmic FLOP"
        else:
            kern, peak, note = "fusg::conv_halo_h3 + fusg::conv_igemm_h3 (all instantiations)", round(PEAK_F16_MFMA_TFLOPS / 3, 1), \
                ("split-fp16: every algorithmic fp32 FLOP costs 3 fp16 matrix FLOPs (ah*wh + ah*wl + al*wh); peak = dense "
                 "fp16 MFMA peak 2500 TFLOP/s / 3, so frac is the matrix-pipe utilisation")
        roofline = {"bound": "mfma", "kernel": kern,
                    "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": traffic, "note": note,
                    "measured": "second pass of the same K steps, branches serialised on one stream, one HIP-event pair per launch",
                    "launches_per_step": conv_launches // max(1, args.steps),
                    "avg_launch_us": round(conv_ms * 1e3 / max(1, conv_launches), 2),
                    "conv_ms_per_step": round(conv_ms / args.steps, 3),
                    "alg_gflop_per_launch": round(gflop_crop * args.batch * args.steps / max(1, conv_launches), 3)}

    extra = {}
    if rank == 0 and world == 1 and not args.no_clip and not args.inpaint:
        # secondary figure: clip mode = 1 HG + 6 ICN + 1 VUnet-full + 5 VUnet-later per vehicle,
        # 8 vehicles x 6 frames per pass (ICN and the VUnet shape half run at batch 48)
        from future_urban_scene_generation_amd.pipeline import synth_clip
        V, F = 8, 6
        clip = synth_clip(V, F, args.res, dev)
        pipe.run_clip(clip)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            pipe.run_clip(clip)
        torch.cuda.synchronize()
        cdt = (time.perf_counter() - t1) / 3
        extra["clip_mode"] = {"vehicles": V, "frames": F, "ms_per_pass": round(cdt * 1e3, 3),
                              "vehicle_clips_per_s": round(V / cdt, 2), "frames_per_s": round(V * F / cdt, 1)}
        del clip
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        from future_urban_scene_generation_amd.pipeline import load_schema
        from future_urban_scene_generation_amd.synth import synth_state_dict
        ns = max(1, min(args.cpu_sample, args.batch))
        sds = {n: synth_state_dict(n, load_schema(n), 0)
               for n in (("hg", "icn", "vunet") + (("edge", "inpaint") if args.inpaint else ()))}
        cpu_batch = {k: v[:ns].cpu() for k, v in batch.items()}
        oracle.crop_pass(sds, {k: v[:1] for k, v in cpu_batch.items()}, args.inpaint)        # warm-up
        torch.manual_seed(77)
        t1 = time.perf_counter()
        ref = oracle.crop_pass(sds, cpu_batch, args.inpaint)
        cpu_dt = time.perf_counter() - t1
        cpu_baseline = {"value": round(ns / cpu_dt, 4), "unit": "crops/s", "cores": torch.get_num_threads(),
                        "kind": "port", "sample": f"{ns} crops of the same workload (batch {ns}), 1 timed pass after warm-up"}
        # quality of the GPU path on the very same sample and noise seed
        torch.manual_seed(77)
        got = pipe.run({k: v[:ns] for k, v in batch.items()})
        import numpy as np
        extra["ssim_vs_cpu_ref"] = round(min(oracle.ssim(got["icn_u8"].cpu().numpy(), ref["icn_u8"]),
                                             oracle.ssim(got["vunet_u8"].cpu().numpy(), ref["vunet_u8"])), 6)
        extra["kp_idx_exact"] = bool(np.array_equal(got["kp_idx"].cpu().numpy(), ref["kp_idx"]))

    if rank == 0:
        line = {"metric": "synthesised vehicle crops/sec @%dx%d" % (args.res, args.res), "value": round(crops_per_s, 3), "unit": "crops/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if args.vehicles else "weak",
                "vs_baseline": None,
                "dtype": "f32" if prec == "f32" else "f32 operands as 3x f16 split products, f32 accumulate (fp32-class accuracy)",
                "data": "synthetic",
                "config": {"workload": ("configs[2]: batch=%d %dx%d crops/GPU, hourglass->warp_learn(ICN)->vunet + edgeconnect"
                                        if args.inpaint else
                                        "configs[1]: batch=%d %dx%d crops/GPU, hourglass->warp_learn(ICN)->vunet first-frame") % (args.batch, args.res, args.res)
                                       + (" | strong scaling: %d vehicles of one frame sharded over the ranks (configs[3])" % args.vehicles if args.vehicles else ""),
                           "batch_per_gpu": args.batch, "res": args.res, "inpaint": bool(args.inpaint), "precision": prec,
                           "gflop_per_crop": round(gflop_crop, 2), "sharding": "vehicles over ranks, gather of uint8 crops to rank 0"},
                "roofline": roofline, "cpu_baseline": cpu_baseline}
        line.update(extra)
        if power is not None:
            line["power"] = power
        line["settle_steps"] = settle
        line["streams"] = "serial" if os.environ.get("FUSG_STREAMS", "1") == "0" else "one HIP stream per network branch"
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
