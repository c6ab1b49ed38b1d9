#!/usr/bin/env python3
"""Where a run_frame call spends its wall time (GPU box, analysis tool): cProfile of 5 frames of 8 vehicles at 720 x 1280."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from future_urban_scene_generation_amd.pipeline import VehiclePipeline, synth_frame  # noqa: E402

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
pipe = VehiclePipeline(dev)
scene = synth_frame(8, (720, 1280), dev, seed=3)
scene["vehicle_seeds"] = list(range(8))
for _ in range(2):
    pipe.run_frame(scene)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    pipe.run_frame(scene)
torch.cuda.synchronize()
print("ms per frame", (time.perf_counter() - t0) / 5 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    pipe.run_frame(scene)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
