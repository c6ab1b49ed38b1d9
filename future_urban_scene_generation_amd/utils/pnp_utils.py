"""Drop-in for the pose fit of the reference's ``utils.pnp_utils`` on MI355X.

``cpc_rodr_4_angles(focals, centers, keypoints_pred, kpoints3D)`` has the reference's signature and return value
(utils/pnp_utils.py:43-130: ``(error, rvect [3, 1], tvect [3, 1])``); ``cpc_rodr_4_angles_batch`` fits every vehicle
of a frame in the same launch.  The four Levenberg-Marquardt runs per vehicle (utils/cpc.py:45-139) execute in the
``fusg_pnp_cpc`` kernel - one GPU thread per (vehicle, start rotation), analytic Jacobian, float32, the reference's
iteration and lambda policies - instead of ~2500 autograd backward passes on the host (5 s per vehicle in the
reference).  What stays on the host is the reference's epilogue: ``np.argmin`` over the four errors and the sign
flip through ``cv2.Rodrigues`` (pnp_utils.py:117-130), here with OpenCV's published Rodrigues formulas in numpy
(OpenCV is not a dependency of this package; parity with OpenCV itself is unpinned).
There is no CPU fallback for the fit: without a HIP device / libfusg.so the call raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np
import torch

from .. import _lib as L
from .. import ops

# start rotations of the four runs (utils/pnp_utils.py:57,72,87,102) and the common start translation (:54)
START_RVECS = ((1.1509305, -1.1552572, 1.2745042), (-0.12036987, 2.4503145, -2.0552557),
               (1.2133899, 1.1018114, -1.120625), (1.6997603, 0.19744678, -0.05384163))
START_TVEC = (0.0, 0.0, 10.0)
MAX_ITER = 50                                         # check_iteration: stop when iteration > 50 (pnp_utils.py:21)


def _skew(u):
    return np.array([[0.0, -u[2], u[1]], [u[2], 0.0, -u[0]], [-u[1], u[0], 0.0]])


def rodrigues(r) -> np.ndarray:
    """Rotation vector -> matrix (cv2.Rodrigues' formula), float64."""
    r = np.asarray(r, np.float64).reshape(3)
    th = float(np.linalg.norm(r))
    if th < 2.220446049250313e-16:
        return np.eye(3)
    u = r / th
    return np.cos(th) * np.eye(3) + (1.0 - np.cos(th)) * np.outer(u, u) + np.sin(th) * _skew(u)


def rodrigues_inv(R) -> np.ndarray:
    """Rotation matrix -> vector (cv2.Rodrigues' published branches: angle from trace and antisymmetric part, the
    axis from the diagonal when the angle is near pi), float64."""
    R = np.asarray(R, np.float64).reshape(3, 3)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = float(np.sqrt((v * v).sum() * 0.25))
    c = min(max((R[0, 0] + R[1, 1] + R[2, 2] - 1.0) * 0.5, -1.0), 1.0)
    th = float(np.arccos(c))
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        x = np.sqrt(max((R[0, 0] + 1.0) * 0.5, 0.0))
        y = np.sqrt(max((R[1, 1] + 1.0) * 0.5, 0.0)) * (-1.0 if R[0, 1] < 0 else 1.0)
        z = np.sqrt(max((R[2, 2] + 1.0) * 0.5, 0.0)) * (-1.0 if R[0, 2] < 0 else 1.0)
        if abs(x) < abs(y) and abs(x) < abs(z) and (R[1, 2] > 0) != (y * z > 0):
            z = -z
        a = np.array([x, y, z])
        return a * (th / np.linalg.norm(a))
    return v * (th / (2.0 * s))


def cpc_fit_device(focals: torch.Tensor, centers: torch.Tensor, points2d: torch.Tensor, points3d: torch.Tensor,
                   max_iter: int = MAX_ITER) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """The Levenberg-Marquardt runs only, device tensors in and out: focals / centers [B, 2], points2d [B, n, 2],
    points3d [B, n, 3] (float32, n <= 16) -> rvec [B, 4, 3], tvec [B, 4, 3], err [B, 4] on the same device."""
    ops._require_gpu(points3d, "points3d")
    args = [t.to(torch.float32).contiguous() for t in (points3d, points2d, focals, centers)]
    dev = points3d.device
    b, n = args[0].shape[0], args[0].shape[1]
    assert args[0].shape == (b, n, 3) and args[1].shape == (b, n, 2) and args[2].shape == (b, 2) and args[3].shape == (b, 2)
    with torch.cuda.device(dev):
        r0 = ops.h2d(START_RVECS, dev, torch.float32)
        t0 = ops.h2d(START_TVEC, dev, torch.float32)
        rvec = torch.empty((b, 4, 3), dtype=torch.float32, device=dev)
        tvec = torch.empty((b, 4, 3), dtype=torch.float32, device=dev)
        err = torch.empty((b, 4), dtype=torch.float32, device=dev)
        L.check(L.lib().fusg_pnp_cpc(args[0].data_ptr(), args[1].data_ptr(), args[2].data_ptr(), args[3].data_ptr(),
                                     r0.data_ptr(), t0.data_ptr(), b, n, 4, int(max_iter), rvec.data_ptr(), tvec.data_ptr(),
                                     err.data_ptr(), ops.stream_ptr()), "pnp_cpc")
    return rvec, tvec, err


def select_and_flip(rvecs: np.ndarray, tvecs: np.ndarray, errors: np.ndarray):
    """utils/pnp_utils.py:117-130 for one vehicle: the first start with the smallest error, rows 0 and 1 of its
    rotation matrix and its translation multiplied by sign(t_z)."""
    i = int(np.argmin(errors))
    rvec = np.asarray(rvecs[i], np.float32).reshape(3, 1)
    tvec = np.asarray(tvecs[i], np.float32).reshape(3, 1)
    sg = np.sign(tvec[2, 0])
    rm = rodrigues(rvec)
    rm[0] *= sg
    rm[1] *= sg
    return errors[i], rodrigues_inv(rm).astype(np.float32).reshape(3, 1), tvec * sg


def cpc_rodr_4_angles_batch(focals, centers, keypoints_pred, kpoints3D, device="cuda"):
    """All vehicles of a frame at once: focals / centers [2] or [B, 2], keypoints_pred [B, n, 2], kpoints3D [B, n, 3]
    (numpy) -> list of (error, rvect [3, 1], tvect [3, 1]) in vehicle order."""
    p2 = np.asarray(keypoints_pred, np.float32)
    p3 = np.asarray(kpoints3D, np.float32)
    b = p3.shape[0]
    f = np.broadcast_to(np.asarray(focals, np.float32).reshape(-1, 2), (b, 2))
    c = np.broadcast_to(np.asarray(centers, np.float32).reshape(-1, 2), (b, 2))
    dev = torch.device(device)
    rv, tv, er = cpc_fit_device(*(torch.from_numpy(np.array(a, dtype=np.float32, order="C")).to(dev) for a in (f, c, p2, p3)))
    rv, tv, er = rv.cpu().numpy(), tv.cpu().numpy(), er.cpu().numpy()
    return [select_and_flip(rv[i], tv[i], er[i]) for i in range(b)]


def cpc_rodr_4_angles(focals, centers, keypoints_pred, kpoints3D):
    """Reference signature (utils/pnp_utils.py:43): focals [2], centers [2], keypoints_pred [n, 2], kpoints3D [n, 3]
    numpy arrays -> (error, rvect [3, 1], tvect [3, 1])."""
    return cpc_rodr_4_angles_batch(focals, centers, np.asarray(keypoints_pred)[None], np.asarray(kpoints3D)[None])[0]
