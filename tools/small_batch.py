#!/usr/bin/env python3
"""GPU box: crops/s of the crop pass at small batches, eager issue vs recorded-plan replay (the 8-vehicles-per-GPU case of
BASELINE configs[3] and the reference's own batch-1 call pattern).  Prints one JSON line per batch size."""
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
for B in [int(a) for a in (sys.argv[1:] or ["1", "2", "4", "8", "16"])]:
    batch = synth_batch(B, 256, dev)
    seeds = list(range(B))
    cp = pipe.compile(batch, seeds)
    res = {"batch": B, "plan_ops": cp.size}
    for name, fn in (("eager", lambda: pipe.run(batch, vehicle_seeds=seeds, check="async")),
                     ("replay", lambda: cp.run(batch, vehicle_seeds=seeds, check="async"))):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        n = max(20, 200 // B)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        t1 = time.perf_counter()
        for _ in range(n):
            fn()
        host = (time.perf_counter() - t1) / n          # host issue time per pass (the queue is not drained in between)
        torch.cuda.synchronize()
        res[name] = {"ms_per_pass": round(dt * 1e3, 3), "crops_per_s": round(B / dt, 1), "host_issue_ms": round(host * 1e3, 3)}
    assert not pipe.finish()
    print(json.dumps(res), flush=True)
    del cp
