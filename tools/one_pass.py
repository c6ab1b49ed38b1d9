#!/usr/bin/env python3
"""GPU box: N crop passes (B = 32 by default) and nothing else - the program to put behind `rocprofv3 --hip-trace
--memory-copy-trace --kernel-trace --stats` when the question is what ONE pass issues (which HIP copies, how many blit kernels).
    python tools/one_pass.py [B] [N]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from future_urban_scene_generation_amd import ops  # noqa: E402
from future_urban_scene_generation_amd.pipeline import VehiclePipeline, synth_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
ops.set_precision("f16x3")
pipe = VehiclePipeline(dev)
batch = synth_batch(B, 256, dev)
seeds = list(range(B))
for _ in range(N):
    pipe.run(batch, vehicle_seeds=seeds, check="async")
torch.cuda.synchronize()
print("passes", N, "range status", pipe.finish())
