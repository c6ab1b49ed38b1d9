#!/usr/bin/env python3
"""GPU box: WHO issues the `__amd_rocclr_copyBuffer` launches of a crop pass (72 per step in profiles/r03_bench_cfg1_kernel_stats.csv,
0.65 ms, unattributed)?  One pass under torch.profiler (CPU activity, with stacks): every aten op that moves bytes or fills
(copy_, _to_copy, clone, contiguous, fill_, zero_, cat, empty_strided is free) grouped by the innermost frame inside this package.
    python tools/copy_sites.py [B]"""
import collections
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from future_urban_scene_generation_amd import ops  # noqa: E402
from future_urban_scene_generation_amd.pipeline import VehiclePipeline, synth_batch  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    ops.set_precision("f16x3")
    pipe = VehiclePipeline(dev)
    batch = synth_batch(B, 256, dev)
    seeds = list(range(B))
    for _ in range(3):
        pipe.run(batch, vehicle_seeds=seeds, check="async")
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        pipe.run(batch, vehicle_seeds=seeds, check="async")
        torch.cuda.synchronize()
    want = ("aten::copy_", "aten::_to_copy", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zero_", "aten::cat", "aten::zeros",
            "aten::to", "aten::index", "aten::slice_copy", "aten::_local_scalar_dense", "aten::item")
    sites = collections.Counter()
    for ev in prof.events():
        if ev.name not in want:
            continue
        frame = "?"
        for fr in ev.stack or []:
            if "future_urban_scene_generation_amd" in fr or "bench.py" in fr:
                frame = fr.split("future_urban_scene_generation_amd/")[-1]
                break
        sites[(ev.name, frame)] += 1
    print("aten ops that move / fill bytes in ONE pass, by call site:")
    for (name, frame), n in sorted(sites.items(), key=lambda kv: -kv[1]):
        print(f"  {n:4d}  {name:28s} {frame}")
    kern = collections.Counter()
    for ev in prof.events():
        if ev.device_type is not None and "DeviceType.CUDA" in str(ev.device_type):
            kern[ev.name[:60]] += 1
    print("device activities in that pass (top):")
    for name, n in kern.most_common(12):
        print(f"  {n:4d}  {name}")


if __name__ == "__main__":
    main()
