"""Oracle: one frame's vehicles through the reference's per-vehicle chain (trajectory_inference.py:55-250, first frame,
--inpaint off, or on when the scene carries EdgeConnect's inputs), vehicle by vehicle like the reference, on the CPU: the counterpart of
future_urban_scene_generation_amd.pipeline.VehiclePipeline.run_frame.  Tests only.

Pinned parts: the three networks, get_maxima, to_image's quantiser (oracle/*.py, bit-equal to the imported reference)
and the pose fit (oracle/pnp.py, pinned to the reference's CPC_R runs).  UNPINNED parts (OpenCV-defined, marked [cv]):
square crop + resize, findHomography / warpPerspective, Lab conversions, the resize-back of the paste - restated in
oracle/cv_host.py from OpenCV's published 8-bit algorithms.  What the reference renders with Open3D between the pose
fit and the plane warp (sketches, masks, plane corner points) is an input of the scene, as in run_frame."""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from . import cv_host as cv
from . import pnp
from .host import to_image_u8, to_tensor_pm1
from .edgeconnect import edge_model_forward, inpaint_model_forward
from .hourglass import get_maxima, heatmap_argmax, hourglass_forward
from .icn import icn_forward
from .vgg import vgg19_forward
from .vunet import vunet_dec_down, vunet_dec_up, vunet_forward

MEAN = np.array([0.485, 0.456, 0.406], np.float32)          # trajectory_inference.py:62-64
STD = np.array([0.229, 0.224, 0.225], np.float32)


def vunet_inputs(frame: np.ndarray, vehicle_mask: np.ndarray, src_sketch: np.ndarray, dst_sketch: np.ndarray, res: int = 256):
    """trajectory_inference.py:203-228 [cv]: -> (x [1, 6, res, res], y_tilde [1, 3, res, res]) float32."""
    masked = vehicle_mask.astype(bool)[..., None] * frame                                     # :203
    ys, xs = np.nonzero(vehicle_mask)
    bbox = [int(xs.min()), int(ys.min()), int(xs.max()), int(ys.max())]
    a = cv.resize_linear_u8(cv.square_crop(masked, bbox), (res, res))
    b = cv.resize_linear_u8(cv.square_crop(src_sketch, bbox), (res, res))
    c = cv.resize_linear_u8(cv.square_crop(dst_sketch, bbox), (res, res))
    a = a.copy()
    a[np.all(b == 0, axis=-1)] = 255                                                          # :219-220
    x = torch.cat([to_tensor_pm1(a)[None], to_tensor_pm1(np.ascontiguousarray(b[..., ::-1]))[None]], 1)
    return x, to_tensor_pm1(np.ascontiguousarray(c[..., ::-1]))[None]


def frame_keypoints(state_dicts: Dict[str, dict], scene: Dict, res: int = 256) -> np.ndarray:
    """Only the keypoint half of `frame_pass` (trajectory_inference.py:58-65, 75-79, 95-97): box crop -> hourglass ->
    get_maxima -> frame pixels, float32 [V, 12, 2].  Tests use it to build a well-posed pose problem for a frame whose
    heat-maps come from random weights: 3-D points that a known pose projects onto these very keypoints."""
    frame = scene["frame"]
    H, W = frame.shape[:2]
    out = []
    for v in range(len(scene["bboxes"])):
        bbox = [int(t) for t in scene["bboxes"][v]]
        (x0, y0, x1, y1), pb, _ = cv.square_crop_geometry((H, W), bbox)
        img_bbox = cv.resize_linear_u8(cv.square_crop(frame, bbox), (res, res))
        x = torch.from_numpy(img_bbox).permute(2, 0, 1).float().div(255)
        x = ((x - torch.from_numpy(MEAN).view(3, 1, 1)) / torch.from_numpy(STD).view(3, 1, 1))[None]
        kp = get_maxima(hourglass_forward(state_dicts["hg"], x)["heatmaps"][-1])[0]
        kp[:, 0] = kp[:, 0] * (x1 - x0) + x0 - pb[0]
        kp[:, 1] = kp[:, 1] * (y1 - y0) + y0 - pb[1]
        out.append(kp.astype(np.float32))
    return np.stack(out)


def well_posed_kp3d(kp_xy: np.ndarray, focals, centers, seed: int = 0, noise: float = 0.01) -> np.ndarray:
    """3-D model points [V, 12, 3] (float32) that a seeded pose near one of the pose fit's start rotations projects onto
    `kp_xy` [V, 12, 2] (pin-hole model of utils/cpc.py: p2 = f * (R X + t).xy / (R X + t).z + c), each keypoint at its own
    seeded depth, plus `noise` metres of perturbation so that the fit's minimum has a non-zero residual (as real
    detections have).  A problem with a sharp, unique minimum - unlike random 3-D points under random-weight keypoints."""
    g = np.random.default_rng(seed)
    f, c = np.asarray(focals, np.float64).reshape(2), np.asarray(centers, np.float64).reshape(2)
    out = []
    for v in range(kp_xy.shape[0]):
        r = pnp.START_RVECS[(seed + v) % 4].astype(np.float64) + g.normal(0, 0.2, 3)
        t = np.array([g.uniform(-2, 2), g.uniform(-1, 1), g.uniform(9, 20)])
        z = t[2] + g.uniform(-2.0, 2.0, 12)
        pc = np.concatenate([(kp_xy[v].astype(np.float64) - c) / f * z[:, None], z[:, None]], 1)
        X = (pnp.rodrigues(r).astype(np.float64).T @ (pc - t).T).T
        out.append((X + g.normal(0, noise, X.shape)).astype(np.float32))
    return np.stack(out)


def frame_pass(state_dicts: Dict[str, dict], scene: Dict, res: int = 256) -> Dict:
    """scene: the dict of pipeline.synth_frame with every array on the host (numpy).  Returns what run_frame returns
    (numpy), plus the intermediates the chain test compares: 'hg_x', 'icn_x', 'vu_x', 'vu_y', 'warped'."""
    frame = scene["frame"]
    H, W = frame.shape[:2]
    inp = scene.get("inpaint") if "edge" in state_dicts else None
    back = frame if inp is not None else scene.get("background", frame)          # :135-136: the composite starts from the frame
    out_icn, out_vu = back.copy(), back.copy()
    V = len(scene["bboxes"])
    res_ = {k: [] for k in ("kp_idx", "kp_xy", "pose", "icn_u8", "vunet_u8", "hg_x", "icn_x", "vu_x", "vu_y", "warped", "geom")}
    if inp is not None:
        res_["inpaint_u8"] = []
    seeds = scene.get("vehicle_seeds")
    for v in range(V):
        bbox = [int(t) for t in scene["bboxes"][v]]
        (x0, y0, x1, y1), pb, pa = cv.square_crop_geometry((H, W), bbox)
        img_bbox = cv.resize_linear_u8(cv.square_crop(frame, bbox), (res, res))              # [cv] :58-60
        x = torch.from_numpy(img_bbox).permute(2, 0, 1).float().div(255)                      # ToTensor
        x = ((x - torch.from_numpy(MEAN).view(3, 1, 1)) / torch.from_numpy(STD).view(3, 1, 1))[None]   # normalize, :61-65
        hm = hourglass_forward(state_dicts["hg"], x)["heatmaps"][-1]
        kp = get_maxima(hm)[0]                                                                # :76-79 (float64)
        kp[:, 0] = kp[:, 0] * (x1 - x0) + x0 - pb[0]                                          # :95-97
        kp[:, 1] = kp[:, 1] * (y1 - y0) + y0 - pb[1]
        kp32 = kp.astype(np.float32)
        kp3d = scene["kp3d"][v]
        if "vgg" in state_dicts:                                                              # :66-69 (VGG-19: unpinned, oracle/vgg.py)
            cad = int(vgg19_forward(state_dicts["vgg"], x)[0].numpy().argmax())
            res_.setdefault("cad_idx", []).append(np.int64(cad))
            if scene.get("kp3d_bank") is not None:
                kp3d = np.asarray(scene["kp3d_bank"], np.float32)[cad]                        # :82-88
        pose = pnp.cpc_rodr_4_angles(scene["focals"], scene["centers"], kp32, kp3d)[:3]       # :104-105
        if inp is not None:                                                                   # :121-143, inputs given (see run_frame)
            t = lambda k: torch.from_numpy(np.ascontiguousarray(inp[k][v:v + 1]))              # noqa: E731
            e = edge_model_forward(state_dicts["edge"], t("gray"), t("edge"), t("mask"))       # :124
            p = inpaint_model_forward(state_dicts["inpaint"], t("img"), e, t("mask"))          # :125
            merged = (p * t("mask") + t("img") * (1 - t("mask"))) * 255.0                      # :126-127
            u8 = merged.permute(0, 2, 3, 1)[0].numpy().astype(np.uint8)                        # :128-129 (truncation)
            bx0, by0, bx1, by1 = (int(q) for q in inp["boxes"][v])
            img_output = cv.resize_linear_u8(u8, (bx1 - bx0, by1 - by0))                       # [cv] :130-131, dsize = (w, h)
            out_icn[by0:by1, bx0:bx1] = img_output                                             # :140-143
            out_vu[by0:by1, bx0:bx1] = img_output
            res_["inpaint_u8"].append(u8)
        off = int(res * 0.1)                                                                  # vehicle_utils.py:49-52
        central = cv.resize_linear_u8(img_bbox[res // 2 - off:res // 2 + off, res // 2 - off:res // 2 + off].copy(), (res, res))
        warped, _ = cv.warp_unwarp_planes(scene["src_planes"][v], scene["src_kp"][v], scene["dst_kp"][v],
                                          scene["src_vis"][v], scene["dst_vis"][v])           # [cv] :171-175
        mask = scene["masks"][v].astype(bool)
        icn_x, info = cv.get_icn_inputs(warped, scene["dst_sketch"][v], mask, central, res, res)      # [cv] :179-180
        icn_x = torch.from_numpy(np.ascontiguousarray(icn_x))
        net = cv.lab2bgr_u8(to_image_u8(icn_forward(state_dicts["icn"], icn_x)[0]))           # :182 (quantiser pinned, Lab [cv])
        cv.paste_back(out_icn, net, info, mask)                                               # [cv] :184-198
        vx, vy = vunet_inputs(frame, scene["masks"][v], scene["src_sketch"][v], scene["dst_sketch"][v], res)
        if seeds is not None:
            torch.manual_seed(int(seeds[v]))
        xt, mu_app, _ = vunet_forward(state_dicts["vunet"], vy, vx, first_frame_like_traj_test=True)   # :230-233
        res_.setdefault("state", []).append({"appearance": mu_app, "central": central})
        vimg = to_image_u8(xt[0])                                                             # :234
        cv.paste_back(out_vu, vimg, info, mask)                                               # [cv] :236-250
        for k, val in (("kp_idx", heatmap_argmax(hm)[0].astype(np.int32)), ("kp_xy", kp32), ("pose", pose), ("icn_u8", net),
                       ("vunet_u8", vimg), ("hg_x", x[0].numpy()), ("icn_x", icn_x[0].numpy()), ("vu_x", vx[0].numpy()),
                       ("vu_y", vy[0].numpy()), ("warped", warped),
                       ("geom", [info["crop_xy_min"][0], info["crop_xy_min"][1], info["crop_xy_min"][0] + info["crop_size_orig"][1],
                                 info["crop_xy_min"][1] + info["crop_size_orig"][0], *info["pad_xy_before"], *info["pad_xy_after"]])):
            res_[k].append(val)
    out = {k: (np.stack(v) if k not in ("pose", "state") else v) for k, v in res_.items()}
    out["frame_icn"], out["frame_vunet"] = out_icn, out_vu
    return out


def later_frame_pass(state_dicts: Dict[str, dict], scene: Dict, state, res: int = 256) -> Dict:
    """A future frame of the same vehicles (trajectory_inference.py:283-450, one trajectory step), vehicle by vehicle: the
    counterpart of VehiclePipeline.run_later_frame.  state = frame_pass(...)["state"] (per vehicle: the VUnet appearance
    code of the first frame and its central crop)."""
    frame = scene["frame"]
    back = scene.get("background", frame)
    out_icn, out_vu = back.copy(), back.copy()
    V = len(scene["masks"])
    res_ = {k: [] for k in ("icn_u8", "vunet_u8", "geom")}
    seeds = scene.get("vehicle_seeds")
    for v in range(V):
        warped, _ = cv.warp_unwarp_planes(scene["src_planes"][v], scene["src_kp"][v], scene["dst_kp"][v],
                                          scene["src_vis"][v], scene["dst_vis"][v])           # [cv] :376-381
        mask = scene["masks"][v].astype(bool)
        icn_x, info = cv.get_icn_inputs(warped, scene["dst_sketch"][v], mask, state[v]["central"], res, res)   # [cv] :385-387
        net = cv.lab2bgr_u8(to_image_u8(icn_forward(state_dicts["icn"], torch.from_numpy(np.ascontiguousarray(icn_x)))[0]))   # :389
        cv.paste_back(out_icn, net, info, mask)                                               # [cv] :393-410
        _, vy = vunet_inputs(frame, scene["masks"][v], scene["dst_sketch"][v], scene["dst_sketch"][v], res)   # :415-420
        if seeds is not None:
            torch.manual_seed(int(seeds[v]))
        outs, skips = vunet_dec_up(state_dicts["vunet"], vy)                                  # :424
        xt = vunet_dec_down(state_dicts["vunet"], outs, skips, state[v]["appearance"])[0]     # :425
        vimg = to_image_u8(xt[0])                                                             # :426
        cv.paste_back(out_vu, vimg, info, mask)                                               # [cv] :428-445
        res_["icn_u8"].append(net)
        res_["vunet_u8"].append(vimg)
        res_["geom"].append([info["crop_xy_min"][0], info["crop_xy_min"][1], info["crop_xy_min"][0] + info["crop_size_orig"][1],
                             info["crop_xy_min"][1] + info["crop_size_orig"][0], *info["pad_xy_before"], *info["pad_xy_after"]])
    out = {k: np.stack(v) for k, v in res_.items()}
    out["frame_icn"], out["frame_vunet"] = out_icn, out_vu
    return out
