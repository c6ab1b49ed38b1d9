"""CPU: the pose-fit oracle (oracle/pnp.py) against the reference's own CPC_R runs stored in tests/golden/pnp.npz
(tools/gen_golden.py), and the host epilogue (best start + sign flip) of both the oracle and the product module."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import pnp as opnp
from future_urban_scene_generation_amd.utils import pnp_utils as prod


def _rot(rv):
    return np.stack([opnp.rodrigues(r) for r in np.asarray(rv).reshape(-1, 3)])


@pytest.mark.parametrize("case", range(6))
def test_oracle_matches_the_reference_runs(case):
    """Per start rotation: same rotation MATRIX (the Rodrigues vector itself is only defined up to 2 pi - the 90 deg
    start ends on |r| > pi), translation and reprojection error as the reference's float32 autograd iteration."""
    g = load_golden("pnp")
    f, c, p2, p3 = opnp.pnp_problem(case)
    np.testing.assert_array_equal(p2, g["points2d"][case])            # the fixture inputs are the seeded problems
    np.testing.assert_array_equal(p3, g["points3d"][case])
    _, _, _, rv, tv, er = opnp.cpc_rodr_4_angles(f, c, p2, p3)
    assert np.abs(_rot(rv) - _rot(g["rvec"][case])).max() < 1e-5
    np.testing.assert_allclose(tv, g["tvec"][case], rtol=1e-5, atol=5e-5)
    np.testing.assert_allclose(er, g["err"][case], rtol=2e-4)
    # every start reaches the same pose on these problems (as in the reference)
    assert np.abs(_rot(rv) - _rot(rv[:1])).max() < 1e-4


def test_reference_quirk_only_six_points_enter_the_jacobian():
    """utils/cpc.py:30 loops over len(inputs) = 6 points: moving the 2-D observations of keypoints 6..11 changes the
    returned error but not the fitted pose."""
    f, c, p2, p3 = opnp.pnp_problem(1)
    r0, t0, e0 = opnp.cpc_solve(p3, p2, opnp.START_RVECS[1], opnp.START_TVEC, f, c)
    q = p2.copy()
    q[6:] += 25.0
    r1, t1, e1 = opnp.cpc_solve(p3, q, opnp.START_RVECS[1], opnp.START_TVEC, f, c)
    assert np.abs(r0 - r1).max() < 1e-4 and np.abs(t0 - t1).max() < 1e-3 and e1 > e0 + 100


@pytest.mark.parametrize("mod", [opnp, prod])
def test_rodrigues_roundtrip_and_flip(mod):
    g = np.random.default_rng(0)
    for _ in range(50):
        r = g.normal(0, 1.5, 3)
        rm = mod.rodrigues(r)
        np.testing.assert_allclose(rm @ rm.T, np.eye(3), atol=1e-12)
        back = mod.rodrigues_inv(rm)
        assert np.linalg.norm(back) <= np.pi + 1e-9
        np.testing.assert_allclose(mod.rodrigues(back), rm, atol=1e-9)
    # angle pi about each axis and about a diagonal (the branch that takes the axis from the diagonal)
    for ax in ([1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [-1, 2, 3]):
        a = np.asarray(ax, float)
        r = a / np.linalg.norm(a) * np.pi
        np.testing.assert_allclose(mod.rodrigues(mod.rodrigues_inv(mod.rodrigues(r))), mod.rodrigues(r), atol=1e-7)
    assert np.array_equal(mod.rodrigues_inv(np.eye(3)), np.zeros(3))
    # selection: first minimum; a pose behind the camera (t_z < 0) comes back in front with rows 0, 1 of R negated
    rv = np.array([[0.3, -0.2, 0.1], [1.0, 0.5, -0.4], [0.0, 0.1, 0.2], [0.5, 0.5, 0.5]], np.float32)
    tv = np.array([[1, 2, 10], [0.5, -1, -12], [0, 0, 9], [1, 1, 1]], np.float32)
    e, r, t = mod.select_and_flip(rv, tv, np.array([3.0, 1.0, 1.0, 2.0]))
    assert e == 1.0 and r.shape == (3, 1) and t.shape == (3, 1) and r.dtype == np.float32
    np.testing.assert_allclose(t.ravel(), [-0.5, 1, 12])
    want = mod.rodrigues(rv[1])
    want[:2] *= -1
    np.testing.assert_allclose(mod.rodrigues(r), want, atol=1e-6)


def test_product_pose_fit_has_no_cpu_path():
    """The fit itself runs in libfusg only: CPU tensors are refused."""
    import torch
    with pytest.raises(RuntimeError):
        prod.cpc_fit_device(torch.zeros(1, 2), torch.zeros(1, 2), torch.zeros(1, 12, 2), torch.zeros(1, 12, 3))


def test_well_posed_kp3d_is_explained_by_a_pose():
    """tests' helper (oracle/frame.py): the 3-D points it builds for given 2-D keypoints are projected onto those keypoints by a
    pose the fit recovers - every start that converges reaches the same small residual - unlike random 3-D points."""
    import numpy as np
    import oracle
    from oracle import pnp
    g = np.random.default_rng(0)
    kp = np.stack([g.uniform([100, 80], [400, 300], (12, 2)) for _ in range(2)]).astype(np.float32)
    f, c = np.array([704.0, 704.0], np.float32), np.array([320.0, 180.0], np.float32)
    x3 = oracle.frame.well_posed_kp3d(kp, f, c, seed=2)
    assert x3.shape == (2, 12, 3) and x3.dtype == np.float32
    for v in range(2):
        e, rv, tv, _, _, ers = pnp.cpc_rodr_4_angles(f, c, kp[v], x3[v])
        assert 0 < float(e) < 2.0 and float(tv.ravel()[2]) > 0                 # a few tenths of a squared pixel (1 cm of noise)
        pc = (pnp.rodrigues(rv).astype(np.float64) @ x3[v].astype(np.float64).T).T + tv.ravel()
        p2 = f * pc[:, :2] / pc[:, 2:] + c
        assert np.abs(p2 - kp[v]).max() < 5.0
