"""Device glue of the per-frame chain (``VehiclePipeline.run_frame``): the uint8 -> float steps that sit between the
frame and the networks in ``trajectory_inference.py:55-79, 200-228`` - box crops, the hourglass / CAD input, the central
crop, the VUnet's inputs, mask bounding boxes and the keypoint coordinates the pose fit reads.  Thin wrappers over
``csrc/cvops.hip`` (``fusg_crop_resize_u8``, ``fusg_vunet_inputs``, ``fusg_mask_bbox_geom``,
``fusg_keypoints_to_frame``); the resize is OpenCV's 8-bit INTER_LINEAR like the other uint8 steps (parity with OpenCV
itself unpinned, oracle/cv_host.py)."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from . import ops
from .warp_learn.planes_utils import _geom_row, _u8desc, square_crop_geometry

IMAGENET_MEAN = (0.485, 0.456, 0.406)          # trajectory_inference.py:62-64
IMAGENET_STD = (0.229, 0.224, 0.225)


def box_geometry(image_hw: Tuple[int, int], bboxes: Sequence[Sequence[int]], device) -> torch.Tensor:
    """square_crop_from_bbox geometry rows (x0, y0, x1, y1, pad_x_before, pad_y_before, pad_x_after, pad_y_after) of
    host-known boxes (the detector's), as a device int32 [V, 8]."""
    rows = [_geom_row(*square_crop_geometry(image_hw, bb)) for bb in bboxes]
    return ops.h2d(rows, device, torch.int32)


def _into(out: Optional[torch.Tensor], b: int, c: int, h: int, w: int, dev) -> torch.Tensor:
    """`out` (a float32 NHWC-physical [b, c, h, w] buffer of an earlier call, e.g. a recorded pass's input: every real
    channel of every pixel is rewritten, the channel padding stays zero) or a fresh zeroed one."""
    if out is None:
        return ops.nhwc_empty(b, c, h, w, dev, zero=True)
    assert tuple(out.shape) == (b, c, h, w) and out.dtype == torch.float32 and out.stride(1) == 1 and out.device == dev
    return out


def crop_resize(src: torch.Tensor, geom: torch.Tensor, out_hw: Tuple[int, int], mode: int = 0,
                mean: Optional[Sequence[float]] = None, std: Optional[Sequence[float]] = None,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """src: CUDA uint8 [H, W, 3] (one image, V windows) or [V, H, W, 3] (one image per window); geom: device int32 [V, 8].
    mode 0 -> uint8 [V, h, w, 3]; mode 1 -> float32 [V, 3, h, w] NHWC-physical, (v/255 - mean)/std; mode 2 -> v/255*2-1.
    out (modes 1, 2): write into this buffer instead of a fresh one."""
    ops._require_gpu(src, "image")
    if src.dim() == 3:
        src = src[None]
    V = int(geom.shape[0])
    assert src.dtype == torch.uint8 and src.shape[0] in (1, V) and geom.dtype == torch.int32 and geom.is_contiguous()
    h, w = out_hw
    dev = src.device
    with torch.cuda.device(dev):
        if mode == 0:
            out = torch.empty((V, h, w, 3), dtype=torch.uint8, device=dev)
            d = _u8desc(out)
        else:
            out = _into(out, V, 3, h, w, dev)
            d = ops.desc(out)
        m = (C.c_float * 3)(*(mean if mean is not None else (0, 0, 0)))
        s = (C.c_float * 3)(*(std if std is not None else (1, 1, 1)))
        L.check(L.lib().fusg_crop_resize_u8(C.byref(_u8desc(src.contiguous())), geom.data_ptr(), C.byref(d), int(mode),
                                            C.cast(m, C.c_void_p) if mean is not None else None,
                                            C.cast(s, C.c_void_p) if std is not None else None, ops.stream_ptr()), "crop_resize_u8")
    return out


def central_crop(img_bbox: torch.Tensor) -> torch.Tensor:
    """get_central_crop's second half (warp_learn/vehicle_utils.py:49-52): the middle 2*int(0.1*w) square of each resized
    box image, resized back to the full size.  img_bbox: CUDA uint8 [V, h, w, 3]."""
    V, h, w, _ = img_bbox.shape
    off = int(w * 0.1)
    row = [w // 2 - off, h // 2 - off, w // 2 + off, h // 2 + off, 0, 0, 0, 0]
    geom = ops.h2d([row] * V, img_bbox.device, torch.int32)
    return crop_resize(img_bbox, geom, (h, w), 0)


def mask_bbox_geom(masks: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """masks: CUDA uint8 [V, H, W] (non-zero = vehicle) -> (bbox int32 [V, 4] = x_min, y_min, x_max, y_max; geom int32
    [V, 8]) without leaving the device."""
    ops._require_gpu(masks, "masks")
    assert masks.dtype == torch.uint8 and masks.dim() == 3
    V, H, W = masks.shape
    m = masks.contiguous()
    bbox = torch.empty((V, 4), dtype=torch.int32, device=m.device)
    geom = torch.empty((V, 8), dtype=torch.int32, device=m.device)
    with torch.cuda.device(m.device):
        L.check(L.lib().fusg_mask_bbox_geom(C.byref(ops.desc(m.view(V, 1, H, W))), bbox.data_ptr(), geom.data_ptr(),
                                            ops.stream_ptr()), "mask_bbox_geom")
    return bbox, geom


def vunet_inputs(frame: torch.Tensor, masks: torch.Tensor, src_sketch: torch.Tensor, dst_sketch: torch.Tensor,
                 geom: torch.Tensor, res: int = 256, out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
                 ) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (x float32 [V, 6, res, res], y_tilde float32 [V, 3, res, res]), NHWC-physical (trajectory_inference.py:203-228);
    out = (x, y) buffers to write into."""
    V, H, W = masks.shape
    dev = frame.device
    x = _into(out[0] if out else None, V, 6, res, res, dev)
    y = _into(out[1] if out else None, V, 3, res, res, dev)
    with torch.cuda.device(dev):
        L.check(L.lib().fusg_vunet_inputs(C.byref(_u8desc(frame.contiguous()[None])), C.byref(ops.desc(masks.contiguous().view(V, 1, H, W))),
                                          C.byref(_u8desc(src_sketch.contiguous())), C.byref(_u8desc(dst_sketch.contiguous())),
                                          geom.data_ptr(), C.byref(ops.desc(x)), C.byref(ops.desc(y)), ops.stream_ptr()), "vunet_inputs")
    return x, y


def keypoints_to_frame(idx: torch.Tensor, geom: torch.Tensor, hm_hw: Tuple[int, int]) -> torch.Tensor:
    """idx: int32 [V, K] heat-map argmax (ops.argmax_hw); geom: the detector boxes' crop rows -> float32 [V, K, 2]."""
    V, K = idx.shape
    out = torch.empty((V, K, 2), dtype=torch.float32, device=idx.device)
    with torch.cuda.device(idx.device):
        L.check(L.lib().fusg_keypoints_to_frame(idx.contiguous().data_ptr(), geom.data_ptr(), out.data_ptr(), V, K,
                                                int(hm_hw[1]), int(hm_hw[0]), ops.stream_ptr()), "keypoints_to_frame")
    return out
