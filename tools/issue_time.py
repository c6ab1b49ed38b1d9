#!/usr/bin/env python3
"""GPU box: is the recorded-pass replay at small batches bound by the host's issue rate or by the GPU?  Per pass, with
an EMPTY queue at the start: host time to issue the plan's operations vs time until the GPU has finished them."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from future_urban_scene_generation_amd.pipeline import VehiclePipeline, synth_batch  # noqa: E402

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
pipe = VehiclePipeline(dev)
for B in [int(a) for a in (sys.argv[1:] or ["1", "2", "8"])]:
    batch = synth_batch(B, 256, dev)
    seeds = list(range(B))
    cp = pipe.compile(batch, seeds)
    for _ in range(10):
        cp.run(batch, vehicle_seeds=seeds, check="async")
    torch.cuda.synchronize()
    hs, ts = [], []
    for _ in range(30):
        t0 = time.perf_counter()
        cp.run(batch, vehicle_seeds=seeds, check="async")
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        hs.append(t1 - t0)
        ts.append(t2 - t0)
    hs.sort(); ts.sort()
    print(json.dumps({"batch": B, "plan_ops": cp.size, "host_issue_ms_median": round(hs[15] * 1e3, 3),
                      "issue_to_done_ms_median": round(ts[15] * 1e3, 3)}), flush=True)
    assert not pipe.finish()
    del cp
