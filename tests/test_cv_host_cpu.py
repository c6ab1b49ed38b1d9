"""Oracle of the OpenCV-defined host steps (oracle/cv_host.py) - PARITY UNPINNED (no OpenCV in the build container, no
fixtures in the reference).  What can be checked without OpenCV: the invariants its documentation and tests state for
these 8-bit paths (identity / integer translation / integer scale are exact; resize to the same size is a copy;
2x decimation of 2x2 blocks returns the blocks; the Lab values of the sRGB primaries and of the grey axis;
axis-aligned fills), and the host-side product code against the oracle (homography fit, crop geometry)."""
import numpy as np
import pytest

from oracle import cv_host as C


@pytest.fixture
def img():
    return np.random.default_rng(0).integers(0, 256, (60, 80, 3), dtype=np.uint8)


def test_warp_identity_translation_scale(img):
    assert np.array_equal(C.warp_perspective_u8(img, np.eye(3), (80, 60)), img)
    out = C.warp_perspective_u8(img, np.array([[1, 0, 5], [0, 1, -3], [0, 0, 1.0]]), (80, 60))
    ref = np.zeros_like(img)
    ref[0:57, 5:80] = img[3:60, 0:75]
    assert np.array_equal(out, ref)                              # integer shift: exact, constant-0 border
    up = C.warp_perspective_u8(img, np.diag([2.0, 2.0, 1.0]), (160, 120))
    assert np.array_equal(up[::2, ::2], img)                     # samples that fall on source pixels are exact
    mid = up[0, 1::2].astype(int)[:-1]                           # half-way samples: (a + b + 1) >> 1 in 15-bit arithmetic
    a, b = img[0, :-1].astype(int), img[0, 1:].astype(int)
    assert np.array_equal(mid, (a * 16384 + b * 16384 + 16384) >> 15)
    assert C.bilinear_tab_i().sum(axis=1).tolist() == [32768] * 1024 and C.bilinear_tab_i()[0].tolist() == [32767, 0, 0, 1]


def test_resize_invariants(img):
    assert np.array_equal(C.resize_linear_u8(img, (80, 60)), img)
    blocks = np.repeat(np.repeat(img, 2, 0), 2, 1)
    assert np.array_equal(C.resize_linear_u8(blocks, (80, 60)), img)
    flat = np.full((33, 47, 3), 173, np.uint8)
    assert (C.resize_linear_u8(flat, (256, 256)) == 173).all()   # constants are preserved at any scale
    up = C.resize_linear_u8(img, (160, 120))
    assert up.shape == (120, 160, 3) and abs(int(up.mean()) - int(img.mean())) <= 1


def test_lab_documented_values():
    prim = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0]]], dtype=np.uint8)
    # OpenCV's 8-bit Lab of the sRGB primaries, white and black (L*255/100, a+128, b+128)
    assert C.rgb2lab_u8(prim)[0].tolist() == [[136, 208, 195], [224, 42, 211], [82, 207, 20], [255, 128, 128], [0, 128, 128]]
    assert np.array_equal(C.rgb2lab_u8(prim[..., ::-1].copy(), bgr=True), C.rgb2lab_u8(prim))
    grey = np.stack([np.arange(256)] * 3, -1).astype(np.uint8)[None]
    lab = C.rgb2lab_u8(grey)
    assert (lab[..., 1] == 128).all() and (lab[..., 2] == 128).all() and (np.diff(lab[0, :, 0].astype(int)) >= 0).all()
    back = C.lab2bgr_u8(lab)
    assert np.abs(back.astype(int) - grey).max() <= 2
    rgb = np.random.default_rng(1).integers(0, 256, (32, 32, 3), dtype=np.uint8)
    err = np.abs(C.lab2bgr_u8(C.rgb2lab_u8(rgb, bgr=True)).astype(int) - rgb)
    assert err.mean() < 1.0                                      # 8-bit Lab is lossy, mostly in saturated blues


def test_fill_poly():
    m = C.fill_poly_mask((20, 30), np.array([[3, 4], [10, 4], [10, 9], [3, 9]]))
    r = np.zeros((20, 30), np.uint8)
    r[4:10, 3:11] = 1                                            # OpenCV fills rectangles including both borders
    assert np.array_equal(m, r)
    tri = C.fill_poly_mask((40, 40), np.array([[5, 5], [30, 10], [20, 35]]))
    assert tri[5, 5] and tri[10, 30] and tri[35, 20] and tri[15, 18] and not tri[5, 30] and not tri[34, 5]
    assert 337 <= int(tri.sum()) <= 337.5 + 50                   # area 337.5 + (part of) the outline
    clip = C.fill_poly_mask((10, 10), np.array([[-5, -5], [20, -5], [20, 20], [-5, 20]]))
    assert clip.all()
    assert np.array_equal(C.fill_poly_mask((20, 30), np.array([[3, 9], [10, 9], [10, 4], [3, 4]])), r)   # orientation-free


def test_homography_and_crop_geometry_match_the_product_host_code():
    from future_urban_scene_generation_amd.warp_learn import planes_utils as P
    rng = np.random.default_rng(3)
    for n in (4, 5, 6):
        s = rng.uniform(0, 600, (n, 2))
        Ht = np.array([[1.1, 0.1, 30], [0.05, 0.9, -20], [1e-4, 2e-4, 1]])
        p = np.c_[s, np.ones(n)] @ Ht.T
        d = p[:, :2] / p[:, 2:]
        assert np.abs(P.find_homography(s, d) - Ht).max() < 1e-8 and np.abs(C.find_homography(s, d) - Ht).max() < 1e-8
        dn = np.int32(d + rng.normal(0, 1.0, d.shape))          # the reference passes int32 pixel coordinates
        a, b = P.find_homography(np.int32(s), dn), C.find_homography(np.int32(s), dn)
        assert np.abs(a / np.abs(a).max() - b / np.abs(b).max()).max() < 1e-6
    assert P.find_homography([[0, 0], [1, 1], [2, 2], [3, 3]], [[0, 0], [1, 0], [1, 1], [0, 1]]) is None    # collinear
    assert C.find_homography([[0, 0], [1, 1], [2, 2], [3, 3]], [[0, 0], [1, 0], [1, 1], [0, 1]]) is None
    for bb in ([100, 200, 400, 350], [0, 0, 50, 700], [1200, 600, 1279, 719], [-5, 10, 30, 40], [600, 300, 640, 330]):
        assert P.square_crop_geometry((720, 1280), bb) == C.square_crop_geometry((720, 1280), bb)
    img = rng.integers(0, 256, (72, 128, 3), dtype=np.uint8)
    crop = C.square_crop(img, [110, 50, 127, 71])                # runs off the right edge: zero padding
    (x0, y0, x1, y1), pb, pa = C.square_crop_geometry((72, 128), [110, 50, 127, 71])
    assert crop.shape[:2] == (y1 - y0, x1 - x0) and pa[0] > 0 and (crop[:, -pa[0]:] == 0).all()


def test_reference_control_flow_of_warp_unwarp():
    """planes_utils.py:40-82: which planes are warped where (visibility and left/right symmetry gating)."""
    rng = np.random.default_rng(5)
    planes = rng.integers(0, 256, (5, 48, 64, 3), dtype=np.uint8)
    sq = [np.int32([[8, 8], [40, 8], [40, 30], [8, 30]])] * 5
    dst = [np.int32([[10, 9], [44, 10], [41, 33], [9, 30]])] * 5
    w, u = C.warp_unwarp_planes(planes, sq, dst, [1, 1, 1, 0, 1], [0, 1, 1, 1, 0])
    assert not w[0].any() and w[1].any() and w[2].any() and not w[3].any() and not w[4].any()
    # left (0) visible in src but not in dst -> warped into the right slot (1); right itself then overwrites slot 1
    assert u[0].any() and u[1].any() and u[2].any() and not u[3].any() and not u[4].any()
    w2, _ = C.warp_unwarp_planes(planes, sq, dst, [1, 0, 0, 0, 0], [0, 1, 0, 0, 0])
    assert w2[1].any() and not w2[0].any()                       # symmetry swap alone
    w3, _ = C.warp_unwarp_planes(planes, sq, dst, [1, 1, 0, 0, 0], [0, 0, 1, 1, 1])
    assert not w3.any()                                          # neither symmetric plane visible in dst


def test_find_homography_batch_equals_the_scalar_fit():
    """The vectorised fit used by VehiclePipeline.run_frame (one call for every plane of every vehicle) performs the scalar
    fit's arithmetic per problem: identical matrices, identical rejections (degenerate and collinear point sets)."""
    from future_urban_scene_generation_amd.warp_learn import planes_utils as pu
    g = np.random.default_rng(0)
    pairs = []
    for k in range(40):
        n = 4 if k % 2 else 6
        s = g.uniform(0, 500, (n, 2))
        Ht = np.eye(3) + g.normal(0, 0.05, (3, 3))
        Ht[2, :2] *= 1e-3
        Ht[2, 2] = 1
        p = np.c_[s, np.ones(n)] @ Ht.T
        d = p[:, :2] / p[:, 2:3] + g.normal(0, 0.5, (n, 2))
        pairs.append((np.int32(s), np.int32(d)))
    pairs.append((np.zeros((4, 2), np.int32), np.int32(g.uniform(0, 9, (4, 2)))))
    pairs.append((np.int32([[0, 0], [1, 1], [2, 2], [3, 3]]), np.int32([[0, 0], [1, 0], [0, 1], [1, 1]])))
    pairs.append((np.int32([[0, 0], [5, 0], [5, 5]]), np.int32([[0, 0], [5, 0], [5, 5]])))          # too few points
    got = pu.find_homography_batch(pairs)
    assert len(got) == len(pairs)
    for (s, d), hb in zip(pairs, got):
        hs = pu.find_homography(s, d)
        assert (hs is None) == (hb is None)
        if hs is not None:
            assert np.array_equal(hs, hb)
    assert got[-1] is None and got[-2] is None and got[-3] is None
