"""CPU oracle of the reference's pose fit ("CamPoseCalib" Levenberg-Marquardt).  TEST INFRASTRUCTURE ONLY.

Restates, in float32 numpy with an analytic Jacobian, what ``utils/cpc.py:6-139`` (``CPC_R.forward``) computes with
autograd and ``utils/pnp_utils.py:8-41`` (``check_iteration``, ``check_lambda``) steer, plus the four-start wrapper
``cpc_rodr_4_angles`` (``utils/pnp_utils.py:43-130``).  Reference behaviour kept on purpose:

* parameters = Rodrigues vector + translation, start translation (0, 0, 10), four fixed start rotations
  (``pnp_utils.py:57,72,87,102``);
* the Jacobian loop runs over ``len(inputs)`` = 6 *points*, not 12 (``cpc.py:30``): only the first six keypoints ever
  enter J, so the normal equations see their residuals only - while the cost in the gain ratio and the returned error
  use all 12 points;
* every step is accepted; lambda starts at 1e-8 * max diag(J^T J) and follows ``check_lambda``; the loop ends on
  ``iteration > 50`` (the 1e-8 thresholds are below float32 resolution), i.e. after 52 body evaluations;
* the returned error is the mean squared residual of the LAST evaluated body, i.e. at the parameters before the last
  update (``cpc.py:139``).

Pin: ``tools/gen_golden.py`` runs the reference's own ``CPC_R`` (with its three holder parameters set to
``requires_grad=False``, which torch >= 1.x needs for the in-place fills of ``cpc.py:10-21``) on the seeded problems
of ``pnp_problem`` and stores its per-start results in ``tests/golden/pnp.npz``.  The restatement is not bit-identical
(autograd orders the Jacobian's arithmetic differently; LAPACK inverts the 6 x 6 system) but both run to the same fixed
point: ``tests/test_oracle_golden.py`` asserts rotation matrix / translation / error agreement at 1e-4 relative.

PARITY UNPINNED for the final sign flip (``pnp_utils.py:122-128``): it goes through ``cv2.Rodrigues`` twice (OpenCV is
absent); ``rodrigues`` / ``rodrigues_inv`` below restate OpenCV's published formulas.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
START_RVECS = np.array([[1.1509305, -1.1552572, 1.2745042],          # utils/pnp_utils.py:57 (0 deg)
                        [-0.12036987, 2.4503145, -2.0552557],       # :72 (90 deg)
                        [1.2133899, 1.1018114, -1.120625],          # :87 (180 deg)
                        [1.6997603, 0.19744678, -0.05384163]],      # :102 (270 deg)
                       dtype=np.float32)
START_TVEC = np.array([0.0, 0.0, 10.0], dtype=np.float32)           # :54


def pnp_problem(seed: int):
    """Seeded synthetic vehicle: 12 box-like 3-D keypoints (metres), a pose near one of the four start rotations,
    pin-hole projection with 1 px noise.  Returns focals[2], centers[2] (float64), points2d [12, 2] (float64),
    points3d [12, 3] (float32) - the argument types of cpc_rodr_4_angles (trajectory_inference.py:87,99-104)."""
    g = np.random.default_rng(seed)
    p = np.array([[sx * 0.8, sy * 0.7, sz * 2.0] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)] +
                 [[0.8, 0, 1.2], [-0.8, 0, 1.2], [0.8, 0, -1.2], [-0.8, 0, -1.2]], dtype=np.float64)
    p += g.normal(0, 0.05, p.shape)
    r = START_RVECS[seed % 4].astype(np.float64) + g.normal(0, 0.25, 3)
    t = np.array([g.uniform(-3, 3), g.uniform(-1, 1.5), g.uniform(8, 25)])
    f = np.array([1000.0 + g.uniform(-50, 50)] * 2)
    c = np.array([640.0, 360.0])
    pc = (rodrigues(r) @ p.T).T + t
    p2 = f * pc[:, :2] / pc[:, 2:] + c + g.normal(0, 1.0, (12, 2))
    return f, c, p2, p.astype(np.float32)


def _skew(u):
    z = u.dtype.type(0)
    return np.array([[z, -u[2], u[1]], [u[2], z, -u[0]], [-u[1], u[0], z]], dtype=u.dtype)


def _rot_and_derivs(r):
    """R(r) as cpc.py:84-93 builds it (cos I + (1 - cos) u u^T + sin [u]x, u = r / |r|) and dR/dr_k, float32."""
    th = np.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]).astype(F32)
    u = (r / th).astype(F32)
    c, s = np.cos(th).astype(F32), np.sin(th).astype(F32)
    eye = np.eye(3, dtype=F32)
    uu = np.outer(u, u).astype(F32)
    U = _skew(u)
    R = (eye * c + (F32(1) - c) * uu + U * s).astype(F32)
    dR = []
    for k in range(3):
        du = ((eye[k] - u * u[k]) / th).astype(F32)                       # d u / d r_k
        d = (-s * u[k]) * eye + (s * u[k]) * uu + (F32(1) - c) * (np.outer(du, u) + np.outer(u, du)) \
            + (c * u[k]) * U + s * _skew(du)
        dR.append(d.astype(F32))
    return R, dR


def cpc_solve(points3d, points2d, rvec0, tvec0, focals, centers, max_iter: int = 50):
    """One Levenberg-Marquardt run (utils/cpc.py:45-139 with the policies of utils/pnp_utils.py:8-41).
    Returns (rvec[3], tvec[3], error) in float32."""
    P = np.asarray(points3d, F32)
    p2 = np.asarray(points2d, F32)
    f = np.asarray(focals, F32)
    cen = np.asarray(centers, F32)
    prm = np.concatenate([np.asarray(rvec0, F32), np.asarray(tvec0, F32)]).astype(F32)
    n = P.shape[0]
    nj = min(6, n)                                    # cpc.py:30: range(len(inputs)) points enter the Jacobian
    J = np.zeros((2 * n, 6), F32)
    lam, factor = None, 2.0
    prev_err = cur_err = err = upd = None
    it = 0
    while True:
        if not (prev_err is None and cur_err is None):                    # check_iteration, pnp_utils.py:8-24
            prev_prm = prm.astype(np.float64) - upd.astype(np.float64)
            g = (J.T @ err.reshape(-1)).astype(F32)
            if np.abs(g).max() < 1e-8:
                break
            if float(np.sqrt((upd * upd).sum(dtype=F32))) < 1e-8 * (np.linalg.norm(prev_prm) + 1e-8):
                break
            if it > max_iter:
                break
        R, dR = _rot_and_derivs(prm[:3])
        pc = (prm[3:] + (R @ P.T).T).astype(F32)                           # cpc.py:95
        iz = (F32(1) / pc[:, 2]).astype(F32)
        pred = (f * pc[:, :2] * iz[:, None] + cen).astype(F32)             # :96-97
        err = (pred - p2).astype(F32)                                      # :99
        for i in range(nj):
            x, y, z = pc[i]
            dpx = np.array([f[0] / z, F32(0), -f[0] * x / (z * z)], F32)   # d pred_x / d pc
            dpy = np.array([F32(0), f[1] / z, -f[1] * y / (z * z)], F32)
            for k in range(3):
                dk = (dR[k] @ P[i]).astype(F32)
                J[2 * i, k] = dpx @ dk
                J[2 * i + 1, k] = dpy @ dk
            J[2 * i, 3:] = dpx
            J[2 * i + 1, 3:] = dpy
        JtJ = (J.T @ J).astype(F32)
        if JtJ.sum() < 1e-7:                                               # :105-106
            break
        if lam is None:
            lam = 1e-8 * float(np.diag(JtJ).max())                         # :110-112
        A = (JtJ + F32(lam) * np.eye(6, dtype=F32)).astype(F32)
        try:
            Ainv = np.linalg.inv(A).astype(F32)
        except np.linalg.LinAlgError:                                      # :116-117
            break
        upd = ((-Ainv @ J.T).astype(F32) @ err.reshape(-1)).astype(F32)    # :115
        prm = (prm + upd).astype(F32)
        prev_err, cur_err = cur_err, err
        it += 1
        if prev_err is not None:                                           # check_lambda, pnp_utils.py:27-41
            pe, ce = prev_err.reshape(-1), cur_err.reshape(-1)
            prev_cost, cur_cost = F32(0.5) * (pe @ pe), F32(0.5) * (ce @ ce)
            den = F32(0.5) * (upd @ (F32(lam) * upd - J.T @ ce).astype(F32))
            with np.errstate(divide="ignore", invalid="ignore"):
                rho = F32(prev_cost - cur_cost) / F32(den)
            if rho <= 0:
                lam, factor = lam * factor, factor * 2
            else:
                lam, factor = lam * max(1.0 / 3.0, float(1 - (2 * float(rho) - 1) ** 3)), 2.0
    error = float((err * err).mean(dtype=F32)) if err is not None else float("nan")
    return prm[:3].copy(), prm[3:].copy(), error


def rodrigues(r):
    """cv2.Rodrigues, vector -> matrix (float64): R = cos I + (1 - cos) u u^T + sin [u]x; identity below ~eps."""
    r = np.asarray(r, np.float64).reshape(3)
    th = np.linalg.norm(r)
    if th < 2.220446049250313e-16:
        return np.eye(3)
    u = r / th
    return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(u, u) + np.sin(th) * _skew(u)


def rodrigues_inv(R):
    """cv2.Rodrigues, matrix -> vector (float64), OpenCV's published branches (calibration module): angle from the
    trace and the antisymmetric part; near pi, the axis from the diagonal with signs fixed from the first row.  (OpenCV
    first re-orthogonalises R by an SVD: a no-op for the exactly orthonormal products this is used on.)"""
    R = np.asarray(R, np.float64).reshape(3, 3)
    rx, ry, rz = R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]
    s = np.sqrt((rx * rx + ry * ry + rz * rz) * 0.25)
    c = min(max((R[0, 0] + R[1, 1] + R[2, 2] - 1) * 0.5, -1.0), 1.0)
    th = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        x = np.sqrt(max((R[0, 0] + 1) * 0.5, 0.0))
        y = np.sqrt(max((R[1, 1] + 1) * 0.5, 0.0)) * (-1.0 if R[0, 1] < 0 else 1.0)
        z = np.sqrt(max((R[2, 2] + 1) * 0.5, 0.0)) * (-1.0 if R[0, 2] < 0 else 1.0)
        if abs(x) < abs(y) and abs(x) < abs(z) and (R[1, 2] > 0) != (y * z > 0):
            z = -z
        v = np.array([x, y, z])
        return v * (th / np.linalg.norm(v))
    return np.array([rx, ry, rz]) * (th / (2 * s))


def select_and_flip(rvecs, tvecs, errors):
    """utils/pnp_utils.py:117-130: the start with the smallest error (first on ties, np.argmin), then rows 0 and 1 of
    its rotation matrix and the translation multiplied by sign(t_z) so that the vehicle is in front of the camera.
    Returns (error, rvec [3, 1], tvec [3, 1]) like cpc_rodr_4_angles."""
    i = int(np.argmin(np.asarray(errors)))
    rvec = np.asarray(rvecs[i], np.float32).reshape(3, 1)
    tvec = np.asarray(tvecs[i], np.float32).reshape(3, 1)
    sg = np.sign(tvec[2, 0])
    Rm = rodrigues(rvec)
    Rm[0] *= sg
    Rm[1] *= sg
    return errors[i], rodrigues_inv(Rm).astype(np.float32).reshape(3, 1), tvec * sg


def cpc_rodr_4_angles(focals, centers, keypoints_pred, kpoints3D):
    """utils/pnp_utils.py:43-130.  Also returns the per-start results for the tests."""
    res = [cpc_solve(kpoints3D, keypoints_pred, r0, START_TVEC, focals, centers) for r0 in START_RVECS]
    rv = np.stack([r[0] for r in res])
    tv = np.stack([r[1] for r in res])
    er = np.array([r[2] for r in res])
    return select_and_flip(rv, tv, er) + (rv, tv, er)
