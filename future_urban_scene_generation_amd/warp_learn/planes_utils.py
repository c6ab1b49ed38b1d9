"""Drop-in for the reference's ``warp_learn.planes_utils`` (+ ``get_icn_inputs`` of ``warp_learn.models``) with the
uint8 image work on the MI355X (§8f-1, §8f-2).

Same function names and argument meaning as the reference (``warp_learn/planes_utils.py:11-118``,
``warp_learn/models.py:323-366``).  Images may be numpy arrays (the reference's type: they are uploaded, processed by
the libfusg kernels of ``csrc/cvops.hip`` and downloaded again, so the functions stay call-compatible with
``trajectory_inference.py``) or CUDA ``torch.uint8`` tensors (then everything stays on the device and the returned
images are CUDA tensors - the point of moving these steps: 5 planes of 2.76 MB per vehicle and frame no longer cross
PCIe twice).  What stays on the host, as north_star says for the geometry: the homography fit
(``find_homography``: 4-6 point pairs) and the bounding-box arithmetic of ``square_crop_from_bbox``.

The arithmetic is OpenCV's (``warpPerspective``, ``fillPoly``, ``resize``, ``cvtColor`` Lab) in its published 8-bit
fixed-point form; the reference pins no OpenCV version and OpenCV is not available where this was built, so parity
with a particular OpenCV build is UNPINNED (oracle/cv_host.py states the algorithms; tests compare kernel and oracle
bit for bit and check OpenCV's documented invariants).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .. import _lib as L
from .. import ops

Image = Union[np.ndarray, torch.Tensor]

# which keypoints bound each texture plane of a car, in the reference's key order (warp_learn/online_visibility.py:9-23)
CAR_TEXTURE_PLANES = {
    "left": ["left_back_trunk", "left_back_wheel", "left_front_wheel", "left_front_light", "upper_left_windshield",
             "upper_left_rearwindow"],
    "right": ["right_back_trunk", "right_back_wheel", "right_front_wheel", "right_front_light", "upper_right_windshield",
              "upper_right_rearwindow"],
    "roof": ["upper_left_rearwindow", "upper_left_windshield", "upper_right_windshield", "upper_right_rearwindow"],
    "front": ["left_front_light", "right_front_light", "upper_right_windshield", "upper_left_windshield"],
    "back": ["left_back_trunk", "right_back_trunk", "upper_right_rearwindow", "upper_left_rearwindow"],
}
pascal_texture_planes = {"car": CAR_TEXTURE_PLANES, "chair": {}}
MAX_VERTS = 8


# ---------------------------------------------------------------------------------------------- plumbing
def _device(device=None) -> torch.device:
    return torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())


def _to_dev(img: Image, device=None) -> Tuple[torch.Tensor, bool]:
    """uint8 image(s) -> contiguous CUDA tensor; second value: was it numpy (then results go back as numpy)."""
    if isinstance(img, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(img)).to(_device(device)), True
    ops._require_gpu(img, "image")
    if img.dtype != torch.uint8:
        raise TypeError(f"expected uint8 images, got {img.dtype}")
    return img.contiguous(), False


def _back(t: torch.Tensor, as_numpy: bool) -> Image:
    return t.cpu().numpy() if as_numpy else t


def _u8desc(t: torch.Tensor) -> L.Tensor:
    """fusg_tensor of a uint8 [n, h, w, c] tensor: logical NCHW extents, HWC strides."""
    assert t.dim() == 4 and t.dtype == torch.uint8, (t.shape, t.dtype)
    return ops.desc(t.permute(0, 3, 1, 2))


# ---------------------------------------------------------------------------------------------- host geometry
def find_homography(src_points, dst_points) -> Optional[np.ndarray]:
    """``cv2.findHomography(src, dst)[0]`` with the default method 0 (all points, least squares): Hartley-normalised
    direct linear transform - the eigenvector of the smallest eigenvalue of A^T A, as OpenCV solves it - followed by
    Levenberg-Marquardt refinement of the reprojection error when there are more than four points.  Host side by design
    (4-6 point pairs).  Returns None where OpenCV returns None (degenerate configuration)."""
    s = np.asarray(src_points, dtype=np.float64).reshape(-1, 2)
    d = np.asarray(dst_points, dtype=np.float64).reshape(-1, 2)
    n = len(s)
    if n < 4 or len(d) != n:
        return None
    cs, cd = s.mean(0), d.mean(0)
    ss, sd = np.abs(s - cs).mean(0), np.abs(d - cd).mean(0)
    if min(ss.min(), sd.min()) < np.finfo(np.float64).eps:
        return None
    ss, sd = 1.0 / ss, 1.0 / sd
    xs, xd = (s - cs) * ss, (d - cd) * sd
    A = np.zeros((2 * n, 9))
    A[0::2, 0:2], A[0::2, 2] = xs, 1.0
    A[0::2, 6:8], A[0::2, 8] = -xd[:, 0:1] * xs, -xd[:, 0]
    A[1::2, 3:5], A[1::2, 5] = xs, 1.0
    A[1::2, 6:8], A[1::2, 8] = -xd[:, 1:2] * xs, -xd[:, 1]
    ev, evec = np.linalg.eigh(A.T @ A)
    if ev[1] < 1e-12 * max(ev[-1], 1e-300):                           # rank loss: no unique homography
        return None
    Hn = evec[:, 0].reshape(3, 3)
    Td_inv = np.array([[1 / sd[0], 0, cd[0]], [0, 1 / sd[1], cd[1]], [0, 0, 1.0]])
    Ts = np.array([[ss[0], 0, -cs[0] * ss[0]], [0, ss[1], -cs[1] * ss[1]], [0, 0, 1.0]])
    H = Td_inv @ Hn @ Ts
    if abs(H[2, 2]) < np.finfo(np.float64).tiny:
        return None
    H = H / H[2, 2]
    if n > 4:                                                          # LM on the 8 free parameters
        h = H.reshape(-1)[:8].copy()
        lam = 1e-3

        def resid(hv):
            Hc = np.append(hv, 1.0).reshape(3, 3)
            p = np.c_[s, np.ones(n)] @ Hc.T
            return (p[:, :2] / p[:, 2:3] - d).reshape(-1), p

        r, p = resid(h)
        for _ in range(10):
            J = np.zeros((2 * n, 8))
            w = p[:, 2]
            J[0::2, 0], J[0::2, 1], J[0::2, 2] = s[:, 0] / w, s[:, 1] / w, 1 / w
            J[0::2, 6], J[0::2, 7] = -p[:, 0] * s[:, 0] / w ** 2, -p[:, 0] * s[:, 1] / w ** 2
            J[1::2, 3], J[1::2, 4], J[1::2, 5] = s[:, 0] / w, s[:, 1] / w, 1 / w
            J[1::2, 6], J[1::2, 7] = -p[:, 1] * s[:, 0] / w ** 2, -p[:, 1] * s[:, 1] / w ** 2
            JtJ, g = J.T @ J, J.T @ r
            step = np.linalg.solve(JtJ + lam * np.diag(np.diag(JtJ)), -g)
            r2, p2 = resid(h + step)
            if r2 @ r2 <= r @ r:
                h, r, p, lam = h + step, r2, p2, lam * 0.1
                if np.abs(step).max() < 1e-13:
                    break
            else:
                lam *= 10.0
        H = np.append(h, 1.0).reshape(3, 3)
    return H


def find_homography_batch(pairs: Sequence[Tuple[np.ndarray, np.ndarray]]) -> List[Optional[np.ndarray]]:
    """`find_homography` for many point sets at once - the same arithmetic per problem (normalised DLT through the symmetric
    eigen-decomposition, then Levenberg-Marquardt with the same acceptance rule), vectorised over the problems of equal point
    count: a frame of 8 vehicles needs ~46 fits, and one Python call per fit (150 us of numpy overhead each) was the largest
    host cost of `VehiclePipeline.run_frame`.  Returns one 3x3 matrix (or None) per pair, in order.  The matrices equal
    `find_homography`'s bit for bit on well-posed fits (tests/test_cv_host_cpu.py); on a near-degenerate one (condition number
    ~1e9: unrelated quadrilaterals) LAPACK's batched and single eigen-decompositions differ in the last place."""
    out: List[Optional[np.ndarray]] = [None] * len(pairs)
    groups: Dict[int, List[int]] = {}
    for i, (sp, dp) in enumerate(pairs):
        sp = np.asarray(sp, dtype=np.float64).reshape(-1, 2)
        dp = np.asarray(dp, dtype=np.float64).reshape(-1, 2)
        if len(sp) >= 4 and len(sp) == len(dp):
            groups.setdefault(len(sp), []).append(i)
    eps = np.finfo(np.float64).eps
    for n, idx in groups.items():
        s = np.stack([np.asarray(pairs[i][0], dtype=np.float64).reshape(-1, 2) for i in idx])      # [N, n, 2]
        d = np.stack([np.asarray(pairs[i][1], dtype=np.float64).reshape(-1, 2) for i in idx])
        N = len(idx)
        cs, cd = s.mean(1), d.mean(1)
        ss, sd = np.abs(s - cs[:, None]).mean(1), np.abs(d - cd[:, None]).mean(1)
        ok = np.minimum(ss.min(1), sd.min(1)) >= eps
        ss = 1.0 / np.where(ss >= eps, ss, 1.0)
        sd = 1.0 / np.where(sd >= eps, sd, 1.0)
        xs, xd = (s - cs[:, None]) * ss[:, None], (d - cd[:, None]) * sd[:, None]
        A = np.zeros((N, 2 * n, 9))
        A[:, 0::2, 0:2], A[:, 0::2, 2] = xs, 1.0
        A[:, 0::2, 6:8], A[:, 0::2, 8] = -xd[:, :, 0:1] * xs, -xd[:, :, 0]
        A[:, 1::2, 3:5], A[:, 1::2, 5] = xs, 1.0
        A[:, 1::2, 6:8], A[:, 1::2, 8] = -xd[:, :, 1:2] * xs, -xd[:, :, 1]
        ev, evec = np.linalg.eigh(np.matmul(A.transpose(0, 2, 1), A))
        ok &= ev[:, 1] >= 1e-12 * np.maximum(ev[:, -1], 1e-300)
        Hn = evec[:, :, 0].reshape(N, 3, 3)
        Td_inv = np.zeros((N, 3, 3))
        Td_inv[:, 0, 0], Td_inv[:, 0, 2], Td_inv[:, 1, 1], Td_inv[:, 1, 2], Td_inv[:, 2, 2] = 1 / sd[:, 0], cd[:, 0], 1 / sd[:, 1], cd[:, 1], 1.0
        Ts = np.zeros((N, 3, 3))
        Ts[:, 0, 0], Ts[:, 0, 2], Ts[:, 1, 1], Ts[:, 1, 2], Ts[:, 2, 2] = ss[:, 0], -cs[:, 0] * ss[:, 0], ss[:, 1], -cs[:, 1] * ss[:, 1], 1.0
        H = np.matmul(np.matmul(Td_inv, Hn), Ts)
        ok &= np.abs(H[:, 2, 2]) >= np.finfo(np.float64).tiny
        H = H / np.where(ok, H[:, 2, 2], 1.0)[:, None, None]
        if n > 4:                                                      # LM on the 8 free parameters, per-problem acceptance
            h = H.reshape(N, 9)[:, :8].copy()
            lam = np.full(N, 1e-3)
            live = ok.copy()
            sh = np.concatenate([s, np.ones((N, n, 1))], 2)            # homogeneous source points

            def resid(hv):
                Hc = np.concatenate([hv, np.ones((len(hv), 1))], 1).reshape(-1, 3, 3)
                pp = np.matmul(sh, Hc.transpose(0, 2, 1))
                return (pp[:, :, :2] / pp[:, :, 2:3] - d).reshape(len(hv), -1), pp

            with np.errstate(all="ignore"):
                r, pp = resid(h)
                for _ in range(10):
                    if not live.any():
                        break
                    w = pp[:, :, 2]
                    J = np.zeros((N, 2 * n, 8))
                    J[:, 0::2, 0], J[:, 0::2, 1], J[:, 0::2, 2] = s[:, :, 0] / w, s[:, :, 1] / w, 1 / w
                    J[:, 0::2, 6], J[:, 0::2, 7] = -pp[:, :, 0] * s[:, :, 0] / w ** 2, -pp[:, :, 0] * s[:, :, 1] / w ** 2
                    J[:, 1::2, 3], J[:, 1::2, 4], J[:, 1::2, 5] = s[:, :, 0] / w, s[:, :, 1] / w, 1 / w
                    J[:, 1::2, 6], J[:, 1::2, 7] = -pp[:, :, 1] * s[:, :, 0] / w ** 2, -pp[:, :, 1] * s[:, :, 1] / w ** 2
                    JtJ = np.matmul(J.transpose(0, 2, 1), J)
                    g = np.matmul(J.transpose(0, 2, 1), r[:, :, None])[:, :, 0]
                    M = JtJ + lam[:, None, None] * (JtJ * np.eye(8)[None])
                    M[~live] = np.eye(8)                               # finished / rejected problems: a harmless system
                    step = np.linalg.solve(M, -g[:, :, None])[:, :, 0]
                    r2, pp2 = resid(h + step)
                    better = live & ((r2 * r2).sum(1) <= (r * r).sum(1))
                    h[better] += step[better]
                    r[better], pp[better] = r2[better], pp2[better]
                    done = better & (np.abs(step).max(1) < 1e-13)
                    lam = np.where(better, lam * 0.1, np.where(live, lam * 10.0, lam))
                    live &= ~done
            H = np.concatenate([h, np.ones((N, 1))], 1).reshape(N, 3, 3)
        for k, i in enumerate(idx):
            out[i] = H[k] if ok[k] else None
    return out


def square_crop_geometry(image_hw: Tuple[int, int], bbox: Sequence[int]):
    """The window ``square_crop_from_bbox`` (utils/crop_utils.py:4-52, 'pascal' branch) cuts out of the zero-padded
    image, without copying pixels: ((x0, y0, x1, y1), pad_xy_before, pad_xy_after)."""
    image_h, image_w = image_hw
    x_min, y_min, x_max, y_max = [int(v) for v in bbox]
    side_x, side_y = x_max - x_min, y_max - y_min
    major = max(side_x, side_y) * 1.1
    cx, cy = x_min + side_x / 2, y_min + side_y / 2
    pad = {"xb": 0, "xa": 0, "yb": 0, "ya": 0}
    x0 = int(cx - major / 2.0)
    if x0 < 0:
        pad["xb"], x0 = int(np.ceil(abs(x0))), 0
    x1 = int(cx + major / 2.0) + pad["xb"]
    if x1 > image_w:
        pad["xa"] = int(np.ceil(abs(x1 - image_w)))
        x1 = image_w + pad["xa"]
    y0 = int(cy - major / 2.0)
    if y0 < 0:
        pad["yb"], y0 = int(np.ceil(abs(y0))), 0
    y1 = int(cy + major / 2.0) + pad["yb"]
    if y1 > image_h:
        pad["ya"] = int(np.ceil(abs(y1 - image_h)))
        y1 = image_h + pad["ya"]
    return (x0, y0, x1, y1), (pad["xb"], pad["yb"]), (pad["xa"], pad["ya"])


def _geom_row(win, pb, pa) -> List[int]:
    return [win[0], win[1], win[2], win[3], pb[0], pb[1], pa[0], pa[1]]


# ---------------------------------------------------------------------------------------------- W-2: get_planes
def plane_polygons(image_hw: Tuple[int, int], src_kpoint_dict: Dict[str, Sequence[float]], pascal_class: str = "car",
                   texture_planes=None) -> List[np.ndarray]:
    """int32 polygons of the texture planes in pixels: normalised keypoints * (w, h), truncated (planes_utils.py:20-27)."""
    h, w = image_hw
    table = (texture_planes or pascal_texture_planes)[pascal_class]
    polys = []
    for names in table.values():
        p = np.asarray([list(map(float, src_kpoint_dict[k])) for k in names])
        p[:, 0] *= w
        p[:, 1] *= h
        polys.append(np.int32(p))
    return polys


def fill_planes(image: torch.Tensor, polygons: Sequence[np.ndarray]) -> torch.Tensor:
    """image [H, W, 3] CUDA uint8, polygons: int32 [n_i, 2] -> [P, H, W, 3] = image * fillPoly mask of each polygon."""
    P = len(polygons)
    if not 1 <= P <= 8 or any(len(p) > MAX_VERTS for p in polygons):
        raise ValueError("fill_planes: 1..8 polygons of at most 8 vertices")
    pts = np.zeros((P, MAX_VERTS, 2), dtype=np.int32)
    nv = np.zeros(P, dtype=np.int32)
    for i, p in enumerate(polygons):
        nv[i] = len(p)
        pts[i, :len(p)] = np.asarray(p, dtype=np.int32).reshape(-1, 2)
    H, W, _ = image.shape
    out = torch.empty((P, H, W, 3), dtype=torch.uint8, device=image.device)
    with torch.cuda.device(image.device):
        L.check(L.lib().fusg_fill_poly_planes_u8(C.byref(_u8desc(image[None])), pts.ctypes.data, nv.ctypes.data, P,
                                                 C.byref(_u8desc(out)), ops.stream_ptr()), "fill_poly_planes_u8")
    return out


def get_planes(image: Image, src_kpoint_dict, pascal_class: str, planes_visibility):
    """Reference signature (planes_utils.py:11-37): (planes [P, H, W, 3], polygon list, visibilities uint8 [P])."""
    img, as_np = _to_dev(image)
    polys = plane_polygons(tuple(img.shape[:2]), src_kpoint_dict, pascal_class)
    names = list(pascal_texture_planes[pascal_class].keys())
    vis = np.stack([planes_visibility[n] for n in names]).astype(np.uint8)
    return _back(fill_planes(img, polys), as_np), polys, vis


# ---------------------------------------------------------------------------------------------- W-1: warp
def warp_perspective(src: torch.Tensor, H: Sequence[np.ndarray], dsize: Tuple[int, int]) -> torch.Tensor:
    """``cv2.warpPerspective(src[i], H[i], dsize=(w, h))`` for a batch: src [N, h, w, 3] CUDA uint8, H: N 3x3 matrices
    (host, double; inverted here in double as OpenCV does) -> [N, dsize[1], dsize[0], 3]."""
    n = src.shape[0]
    minv = np.stack([np.linalg.inv(np.asarray(h, dtype=np.float64)) for h in H]).reshape(n, 9)
    minv_d = ops.h2d(minv, src.device)
    out = torch.empty((n, dsize[1], dsize[0], 3), dtype=torch.uint8, device=src.device)
    with torch.cuda.device(src.device):
        L.check(L.lib().fusg_warp_perspective_u8(C.byref(_u8desc(src)), minv_d.data_ptr(), C.byref(_u8desc(out)),
                                                 ops.stream_ptr()), "warp_perspective_u8")
    return out


def warp_jobs(src_planes_kpoints, dst_planes_kpoints, src_visibilities, dst_visibilities, pascal_class: str = "car",
              texture_planes=None):
    """The host half of warp_unwarp_planes (planes_utils.py:53-75): which planes are warped where, and their two
    homographies.  -> list of (source plane i, destination slot j, H12, H21)."""
    keys = list((texture_planes or pascal_texture_planes)[pascal_class].keys())
    sym = [keys.index("left"), keys.index("right")]
    jobs = []
    for i in range(len(keys)):
        if not src_visibilities[i]:
            continue
        if i not in sym and not dst_visibilities[i]:
            continue
        if i in sym and 1 not in [dst_visibilities[j] for j in sym]:
            continue
        j = i
        if i in sym and not dst_visibilities[i]:
            j = sym[0] if i == sym[1] else sym[1]
        H12 = find_homography(src_planes_kpoints[i], dst_planes_kpoints[j])
        H21 = find_homography(dst_planes_kpoints[j], src_planes_kpoints[i])
        if H12 is not None and H21 is not None:
            jobs.append((i, j, H12, H21))
    return jobs


def warp_jobs_frame(src_kp_v, dst_kp_v, src_vis_v, dst_vis_v, pascal_class: str = "car", texture_planes=None):
    """`warp_jobs` for every vehicle of a frame with ALL homographies fitted in one vectorised call
    (`find_homography_batch`): -> list (per vehicle) of lists of (source plane i, destination slot j, H12, H21)."""
    keys = list((texture_planes or pascal_texture_planes)[pascal_class].keys())
    sym = [keys.index("left"), keys.index("right")]
    want, pairs = [], []
    for v in range(len(src_kp_v)):
        sv, dv = src_vis_v[v], dst_vis_v[v]
        for i in range(len(keys)):
            if not sv[i]:
                continue
            if i not in sym and not dv[i]:
                continue
            if i in sym and 1 not in [dv[j] for j in sym]:
                continue
            j = i
            if i in sym and not dv[i]:
                j = sym[0] if i == sym[1] else sym[1]
            want.append((v, i, j))
            pairs.append((src_kp_v[v][i], dst_kp_v[v][j]))
            pairs.append((dst_kp_v[v][j], src_kp_v[v][i]))
    Hs = find_homography_batch(pairs)
    jobs = [[] for _ in range(len(src_kp_v))]
    for k, (v, i, j) in enumerate(want):
        H12, H21 = Hs[2 * k], Hs[2 * k + 1]
        if H12 is not None and H21 is not None:
            jobs[v].append((i, j, H12, H21))
    return jobs


def warp_planes_batch(src_planes: torch.Tensor, jobs_per_vehicle) -> torch.Tensor:
    """`planes_warped` of warp_unwarp_planes for every vehicle of a frame in ONE launch: src_planes CUDA uint8
    [V, P, H, W, 3], jobs_per_vehicle[v] = warp_jobs(...) of vehicle v -> [V, P, H, W, 3] (zeros where nothing is warped)."""
    V, P, H, W, _ = src_planes.shape
    flat = src_planes.reshape(V * P, H, W, 3)
    warped = torch.zeros_like(flat)
    src_idx, dst_idx, Hs = [], [], []
    for v, jobs in enumerate(jobs_per_vehicle):
        last = {}
        for k, (i, j, H12, _) in enumerate(jobs):                 # plane order: a later job on slot j overwrites an earlier one
            last[j] = (i, H12)
        for j, (i, H12) in last.items():
            src_idx.append(v * P + i)
            dst_idx.append(v * P + j)
            Hs.append(H12)
    if Hs:                                                        # every job reads its plane and writes its slot in place
        flat = flat.contiguous()
        minv = np.linalg.inv(np.asarray(Hs, dtype=np.float64).reshape(len(Hs), 3, 3)).reshape(len(Hs), 9)   # LAPACK getrf/getri per matrix, as one call
        minv_d = ops.h2d(minv, flat.device)
        index = ops.h2d(np.stack([src_idx, dst_idx], 1).astype(np.int32), flat.device)
        with torch.cuda.device(flat.device):
            L.check(L.lib().fusg_warp_perspective_indexed_u8(C.byref(_u8desc(flat)), minv_d.data_ptr(), index.data_ptr(), len(Hs),
                                                             C.byref(_u8desc(warped)), ops.stream_ptr()), "warp_perspective_indexed_u8")
    return warped.view(V, P, H, W, 3)


def warp_unwarp_planes(src_planes: Image, src_planes_kpoints: List[np.ndarray], dst_planes_kpoints: List[np.ndarray],
                       src_visibilities, dst_visibilities, pascal_class: str, pascal_texture_planes=pascal_texture_planes,
                       unwarp: bool = True):
    """Reference signature (planes_utils.py:40-82): visibility / symmetry gating and the homography fits on the host,
    both warps of all selected planes in two batched launches.  `unwarp=False` skips the second warp (its result is
    discarded by the reference's only caller, trajectory_inference.py:171) and returns None for it."""
    planes, as_np = _to_dev(src_planes)
    jobs = warp_jobs(src_planes_kpoints, dst_planes_kpoints, src_visibilities, dst_visibilities, pascal_class,
                     pascal_texture_planes)
    warped = torch.zeros_like(planes)
    unwarped = torch.zeros_like(planes) if unwarp else None
    if jobs:
        h, w = planes.shape[1:3]
        src_idx = torch.tensor([i for i, _, _, _ in jobs], device=planes.device)
        w1 = warp_perspective(planes[src_idx], [H12 for _, _, H12, _ in jobs], (w, h))
        # the reference assigns in plane order: a later job writing the same slot j overwrites an earlier one
        for k, (_, j, _, _) in enumerate(jobs):
            warped[j] = w1[k]
        if unwarp:
            w2 = warp_perspective(w1, [H21 for _, _, _, H21 in jobs], (w, h))
            for k, (i, _, _, _) in enumerate(jobs):
                unwarped[i] = w2[k]
    return _back(warped, as_np), (None if unwarped is None else _back(unwarped, as_np))


# ---------------------------------------------------------------------------------------------- W-3: ICN inputs
def icn_inputs_batch(planes: torch.Tensor, sketches: torch.Tensor, centrals: torch.Tensor, bboxes: Sequence[Sequence[int]],
                     icn_w: int = 256, icn_h: int = 256):
    """Batched, device-resident ICN input assembly: planes [B, P, H, W, 3] (BGR), sketches [B, H, W, 3] (RGB), centrals
    [B, icn_h, icn_w, 3] (RGB) CUDA uint8; bboxes: per vehicle [x_min, y_min, x_max, y_max] of its sketch mask.
    Returns (float32 [B, 3 * (P + 2), icn_h, icn_w] NHWC-physical - what G_Resnet's stem reads without a copy -,
    list of crop_info dicts)."""
    B, P, H, W, _ = planes.shape
    geom, infos = [], []
    for bb in bboxes:
        win, pb, pa = square_crop_geometry((H, W), bb)
        geom.append(_geom_row(win, pb, pa))
        infos.append({"crop_xy_min": (win[0], win[1]), "pad_xy_before": pb, "pad_xy_after": pa,
                      "crop_size_orig": (win[3] - win[1], win[2] - win[0])})
    geom_d = torch.tensor(geom, dtype=torch.int32, device=planes.device)
    out = ops.nhwc_empty(B, 3 * (P + 2), icn_h, icn_w, planes.device, zero=True)
    with torch.cuda.device(planes.device):
        L.check(L.lib().fusg_icn_inputs(C.byref(_u8desc(sketches.contiguous())), C.byref(_u8desc(centrals.contiguous())),
                                        C.byref(_u8desc(planes.reshape(B * P, H, W, 3).contiguous())), geom_d.data_ptr(),
                                        C.byref(ops.desc(out)), ops.stream_ptr()), "icn_inputs")
    return out, infos


def icn_inputs_device(planes: torch.Tensor, sketches: torch.Tensor, centrals: torch.Tensor, geom: torch.Tensor,
                      icn_w: int = 256, icn_h: int = 256, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """icn_inputs_batch with the crop geometry already on the device (frame_ops.mask_bbox_geom): nothing is read back.
    out: an earlier result's buffer to overwrite (every real channel of every pixel is written)."""
    B, P, H, W, _ = planes.shape
    if out is None:
        out = ops.nhwc_empty(B, 3 * (P + 2), icn_h, icn_w, planes.device, zero=True)
    else:
        assert tuple(out.shape) == (B, 3 * (P + 2), icn_h, icn_w) and out.dtype == torch.float32 and out.stride(1) == 1
    with torch.cuda.device(planes.device):
        L.check(L.lib().fusg_icn_inputs(C.byref(_u8desc(sketches.contiguous())), C.byref(_u8desc(centrals.contiguous())),
                                        C.byref(_u8desc(planes.reshape(B * P, H, W, 3).contiguous())), geom.data_ptr(),
                                        C.byref(ops.desc(out)), ops.stream_ptr()), "icn_inputs")
    return out


def get_icn_inputs(planes: Image, sketch_normal: Image, sketch_mask, central_crop: Image, icn_w: int, icn_h: int):
    """Reference signature (warp_learn/models.py:323-366): -> (float32 [1, 21, icn_h, icn_w] on the device, crop_info)."""
    pl, _ = _to_dev(planes)
    sk, _ = _to_dev(sketch_normal, pl.device)
    cc, _ = _to_dev(central_crop, pl.device)
    if isinstance(sketch_mask, torch.Tensor):
        ys, xs = torch.nonzero(sketch_mask, as_tuple=True)
        bbox = [int(xs.min()), int(ys.min()), int(xs.max()), int(ys.max())]
    else:
        ys, xs = np.nonzero(sketch_mask)
        bbox = [int(np.min(xs)), int(np.min(ys)), int(np.max(xs)), int(np.max(ys))]
    out, infos = icn_inputs_batch(pl[None], sk[None], cc[None], [bbox], icn_w, icn_h)
    return out, infos[0]


def planes_to_torch(planes: Image, to_LAB: bool) -> torch.Tensor:
    """Reference signature (planes_utils.py:85-93): uint8 [P, H, W, 3] (BGR) -> float32 [P, 3, H, W] in [-1, 1]."""
    pl, _ = _to_dev(planes)
    P, H, W, _ = pl.shape
    if not to_LAB:
        x = pl.permute(0, 3, 1, 2).to(torch.float32)
        return (x / 255.0 - 0.5) / 0.5
    # full-frame window: crop == frame, no resize; the sketch / central slots are computed and dropped
    geom = torch.tensor([[0, 0, W, H, 0, 0, 0, 0]], dtype=torch.int32, device=pl.device)
    dst = ops.nhwc_empty(1, 3 * (P + 2), H, W, pl.device, zero=True)
    with torch.cuda.device(pl.device):
        L.check(L.lib().fusg_icn_inputs(C.byref(_u8desc(pl[:1])), C.byref(_u8desc(pl[:1])), C.byref(_u8desc(pl)),
                                        geom.data_ptr(), C.byref(ops.desc(dst)), ops.stream_ptr()), "icn_inputs")
    return dst[0, 6:].reshape(P, 3, H, W)


# ---------------------------------------------------------------------------------------------- W-10 / f-2: way back
def lab2bgr(img: torch.Tensor) -> torch.Tensor:
    """``cv2.cvtColor(img, cv2.COLOR_LAB2BGR)`` on CUDA uint8 [N, H, W, 3]."""
    out = torch.empty_like(img)
    with torch.cuda.device(img.device):
        L.check(L.lib().fusg_lab2bgr_u8(C.byref(_u8desc(img)), C.byref(_u8desc(out)), ops.stream_ptr()), "lab2bgr_u8")
    return out


def to_image(x, from_LAB: bool):
    """Reference signature (planes_utils.py:96-118): [3, H, W] in [-1, 1] -> uint8 [H, W, 3] BGR, as numpy (what the
    reference returns).  `to_image_device` keeps the result on the GPU."""
    if isinstance(x, np.ndarray):                                    # the reference's "already numpy" branch: HWC in
        x = torch.from_numpy(np.ascontiguousarray(np.transpose(x, (2, 0, 1))))
        x = x.to(_device())
    assert x.dim() == 3, f"Unsupported image shape {tuple(x.shape)}"
    return to_image_device(x.detach()[None], from_LAB)[0].cpu().numpy()


def to_image_device(x: torch.Tensor, from_LAB: bool) -> torch.Tensor:
    """[B, 3, H, W] float32 in [-1, 1] (CUDA) -> uint8 [B, H, W, 3]: trunc(clip((x + 1) / 2 * 255)) [+ Lab -> BGR]."""
    u8 = ops.to_image_u8(x.to(torch.float32))
    return lab2bgr(u8) if from_LAB else u8


def paste_back_device(frame: torch.Tensor, net_images: torch.Tensor, geom: torch.Tensor, masks: torch.Tensor,
                      box_images: Optional[torch.Tensor] = None, box_geom: Optional[torch.Tensor] = None) -> torch.Tensor:
    """paste_back with device geometry rows (int32 [V, 8]) and device masks (uint8 [V, H, W]); returns a new frame tensor.
    box_images uint8 [V, h, w, 3] + box_geom int32 [V, 8] (x0, y0, x1, y1, ...): the --inpaint form - every vehicle first
    writes its inpainted box image, resized to its rectangle, then its masked network image (fusg_paste_layers_u8)."""
    fr = frame.contiguous().clone()
    V, H, W = masks.shape
    with torch.cuda.device(fr.device):
        if box_images is None:
            L.check(L.lib().fusg_paste_back_u8(C.byref(_u8desc(net_images.contiguous())), C.byref(ops.desc(masks.contiguous().view(V, 1, H, W))),
                                               geom.data_ptr(), C.byref(_u8desc(fr[None])), ops.stream_ptr()), "paste_back_u8")
        else:
            assert box_geom is not None and box_geom.dtype == torch.int32 and tuple(box_geom.shape) == (V, 8) and box_images.shape[0] == V
            L.check(L.lib().fusg_paste_layers_u8(C.byref(_u8desc(net_images.contiguous())), C.byref(ops.desc(masks.contiguous().view(V, 1, H, W))),
                                                 geom.data_ptr(), C.byref(_u8desc(box_images.contiguous())), box_geom.contiguous().data_ptr(),
                                                 C.byref(_u8desc(fr[None])), ops.stream_ptr()), "paste_layers_u8")
    return fr


def paste_back(frame: Image, net_images: torch.Tensor, crop_infos: Sequence[dict], paste_masks: Image) -> Image:
    """The "revert and stitch" of trajectory_inference.py:184-198 for V vehicles in one launch: net_images [V, h, w, 3]
    uint8 (CUDA), their crop_infos (from get_icn_inputs) and paste masks [V, H, W] (bool / uint8; the reference's
    `dst_sketch_mask`); later vehicles overwrite earlier ones, exactly as the reference's loop does.  Returns the
    updated frame (a new array / tensor)."""
    fr, as_np = _to_dev(frame, net_images.device)
    fr = fr.clone()
    if isinstance(paste_masks, np.ndarray):
        masks = torch.from_numpy(np.ascontiguousarray(paste_masks.astype(np.uint8))).to(fr.device)
    else:
        masks = paste_masks.to(torch.uint8).contiguous()
    geom = []
    for ci in crop_infos:
        (x0, y0), pb, pa = ci["crop_xy_min"], ci["pad_xy_before"], ci["pad_xy_after"]
        hh, ww = ci["crop_size_orig"]
        geom.append([x0, y0, x0 + ww, y0 + hh, pb[0], pb[1], pa[0], pa[1]])
    geom_d = torch.tensor(geom, dtype=torch.int32, device=fr.device)
    V, H, W = masks.shape
    with torch.cuda.device(fr.device):
        L.check(L.lib().fusg_paste_back_u8(C.byref(_u8desc(net_images.contiguous())), C.byref(ops.desc(masks.view(V, 1, H, W))),
                                           geom_d.data_ptr(), C.byref(_u8desc(fr[None])), ops.stream_ptr()), "paste_back_u8")
    return _back(fr, as_np)
