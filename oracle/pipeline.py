"""Oracle: one crop pass (hourglass -> ICN -> VUnet first-frame [-> EdgeConnect]) on the CPU, the
counterpart of future_urban_scene_generation_amd.pipeline.VehiclePipeline.run (call order of
trajectory_inference.py:75-79, 182, 230-233, 124-129).  Tests / smoke / bench cpu_baseline only."""
import time
from typing import Dict, Optional

import torch

from .edgeconnect import edge_model_forward, inpaint_model_forward
from .host import to_image_u8
from .hourglass import heatmap_argmax, hourglass_forward
from .icn import icn_forward
from .vunet import vunet_forward


def crop_pass(state_dicts: Dict[str, dict], batch_cpu: Dict[str, torch.Tensor], inpaint: bool = False,
              seconds: Optional[Dict[str, float]] = None, deadline: Optional[float] = None):
    """`seconds`: optional dict that receives the wall time of each network of this pass (bench.py's cpu_baseline).
    `deadline`: optional time.perf_counter() value; the pass raises TimeoutError after the first network that ends
    past it (bench.py bounds its thread sweep with this)."""
    out = {}
    t = [time.perf_counter()]

    def lap(name):
        t.append(time.perf_counter())
        if seconds is not None:
            seconds[name] = t[-1] - t[-2]
        if deadline is not None and t[-1] > deadline:
            raise TimeoutError(f"oracle.crop_pass: past the deadline after '{name}'")

    hm = hourglass_forward(state_dicts["hg"], batch_cpu["hg_x"])["heatmaps"][-1]
    out["kp_idx"] = heatmap_argmax(hm)
    lap("hg")
    out["icn_u8"] = to_image_u8(icn_forward(state_dicts["icn"], batch_cpu["icn_x"]))
    lap("icn")
    xt = vunet_forward(state_dicts["vunet"], batch_cpu["vu_y"], batch_cpu["vu_x"], first_frame_like_traj_test=True)[0]
    out["vunet_u8"] = to_image_u8(xt)
    lap("vunet")
    if inpaint:
        e = edge_model_forward(state_dicts["edge"], batch_cpu["ec_gray"], batch_cpu["ec_edge"], batch_cpu["ec_mask"])
        p = inpaint_model_forward(state_dicts["inpaint"], batch_cpu["ec_img"], e, batch_cpu["ec_mask"])
        m = batch_cpu["ec_mask"]
        out["inpaint_u8"] = ((p * m + batch_cpu["ec_img"] * (1 - m)) * 255.0).permute(0, 2, 3, 1).numpy().astype("uint8")
        lap("edgeconnect")
    return out
