#!/usr/bin/env python3
"""Where the host spends its time issuing one frame (GPU box, analysis tool): cProfile of 10 x VehiclePipeline._issue_frame
(8 vehicles at 720 x 1280, recorded-pass replay) - what bounds `run_frames` once the GPU side is pipelined."""
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
INP = len(sys.argv) > 1 and sys.argv[1] == "inpaint"      # the --inpaint branch (EdgeConnect as a fourth network)
pipe = VehiclePipeline(dev, inpaint=INP)
scene = synth_frame(8, (720, 1280), dev, seed=3, inpaint=INP)
scene["vehicle_seeds"] = list(range(8))
for _ in pipe.run_frames([scene] * 3):
    pass
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in pipe.run_frames([scene] * 10):
    pass
torch.cuda.synchronize()
print("ms per frame (run_frames)", (time.perf_counter() - t0) / 10 * 1e3)
t0 = time.perf_counter()
for _ in range(10):
    pipe._issue_frame(scene, True)
th = (time.perf_counter() - t0) / 10 * 1e3
torch.cuda.synchronize()
print("host ms to issue a frame %.3f; with the GPU drained after it %.3f" % (th, (time.perf_counter() - t0) / 10 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    pipe._issue_frame(scene, True)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
