"""Oracle for the OpenCV-defined host steps either side of the ICN (§8f-1/2).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  **PARITY UNPINNED**: the reference delegates this arithmetic to
`opencv-python` (requirements.txt:5, no version pin), which is not vendored under /root/reference and is absent from
the build container, and the reference has no tests or golden vectors at these boundaries.  What
follows therefore restates OpenCV's *published* 8-bit algorithms (modules/imgproc: imgwarp.cpp, resize.cpp,
color_lab.cpp, drawing.cpp of the 3.x / 4.x line) in numpy, and the reference's own control flow line by line:

  warp_perspective_u8   cv2.warpPerspective(src, H, dsize)      INTER_LINEAR, BORDER_CONSTANT 0 (planes_utils.py:76-77):
                        M = inv(H) in double; per destination pixel X = rint(32 * x'/w'), Y likewise (INTER_BITS = 5);
                        bilinear weights from the 32 x 32 fixed-point table (INTER_REMAP_COEF_BITS = 15, entry (0, 0)
                        saturates to 32767 and receives OpenCV's +1 correction on its last weight); result
                        (sum + 2^14) >> 15; neighbours outside the image read the border value 0.
  resize_linear_u8      cv2.resize(src, (w, h))                 INTER_LINEAR (models.py:343,348; trajectory_inference.py:190):
                        half-pixel centres, 11-bit coefficients, the 8-bit vertical pass
                        ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
  rgb2lab_u8            cv2.cvtColor(..., COLOR_RGB2LAB / COLOR_BGR2LAB) on uint8 (models.py:355,358; planes_utils.py:88):
                        the integer path: sRGB gamma table (<< 3), 12-bit D65 matrix, cube-root table (<< 15),
                        L = (296 fY - Lshift) >> 15, a = (500 (fX - fY)) >> 15 + 128, b = (200 (fY - fZ)) >> 15 + 128.
  lab2bgr_u8            cv2.cvtColor(x, COLOR_LAB2BGR) on uint8 (planes_utils.py:117): the float path of the 3.x line
                        (L*100/255, a-128, b-128 -> XYZ -> linear RGB -> sRGB gamma) rounded to uint8.
  fill_poly_mask        cv2.fillPoly(zeros, [pts], (1,1,1))     (planes_utils.py:29): 8-connected outline
                        (left-to-right Bresenham of LineIterator) + scanline interior with 16.16 fixed-point edges.
  find_homography       cv2.findHomography(src, dst) method 0   (planes_utils.py:71-72): normalised DLT on all points
                        + Gauss-Newton refinement of the reprojection error (OpenCV: LM, <= 10 iterations).
Newer OpenCV builds interpolate RGB<->Lab through a trilinear LUT and use an integer Lab->RGB path; they differ from
the formulas above by at most a few LSB.  Property tests (identity / integer translation / integer scale exact,
axis-aligned fills exact, grey axis a = b = 128) anchor the restatement in tests/test_cv_host_cpu.py.
"""
from typing import Dict, List, Sequence, Tuple

import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
REMAP_COEF_BITS = 15
RESIZE_COEF_BITS = 11


# ---------------------------------------------------------------------------------------------- warpPerspective
def bilinear_tab_i() -> np.ndarray:
    """BilinearTab_i[ay*32+ax][4] (imgwarp.cpp initInterTab2D, fixed point): saturate_short(w * 2^15) and the sum
    correction, which only entry (0, 0) needs (32768 saturates to 32767; the correction lands on its last weight)."""
    t = np.arange(INTER_TAB_SIZE, dtype=np.float32) / INTER_TAB_SIZE
    w1 = np.stack([1.0 - t, t], axis=1)                                     # [32][2]
    tab = np.einsum("ik,jl->ijkl", w1, w1).reshape(INTER_TAB_SIZE * INTER_TAB_SIZE, 4)   # [ay][ax][ky][kx]
    itab = np.clip(np.rint(tab * (1 << REMAP_COEF_BITS)), -32768, 32767).astype(np.int32)
    assert (itab.sum(axis=1)[1:] == (1 << REMAP_COEF_BITS)).all()
    itab[0, 3] += (1 << REMAP_COEF_BITS) - itab[0].sum()
    return itab


def perspective_coords(Minv: np.ndarray, w: int, h: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(sx, sy, alpha) of every destination pixel: integer source cell and index into the weight table."""
    x = np.arange(w, dtype=np.float64)[None, :]
    y = np.arange(h, dtype=np.float64)[:, None]
    X0 = Minv[0, 0] * x + Minv[0, 1] * y + Minv[0, 2]
    Y0 = Minv[1, 0] * x + Minv[1, 1] * y + Minv[1, 2]
    W0 = Minv[2, 0] * x + Minv[2, 1] * y + Minv[2, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        W = np.where(W0 != 0, INTER_TAB_SIZE / W0, 0.0)
    lim = (-2147483648.0, 2147483647.0)
    X = np.rint(np.clip(X0 * W, *lim)).astype(np.int64)
    Y = np.rint(np.clip(Y0 * W, *lim)).astype(np.int64)
    sx = np.clip(X >> INTER_BITS, -32768, 32767)
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    alpha = (Y & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (X & (INTER_TAB_SIZE - 1))
    return sx, sy, alpha


def warp_perspective_u8(src: np.ndarray, H: np.ndarray, dsize: Tuple[int, int]) -> np.ndarray:
    """cv2.warpPerspective(src[h, w, c] uint8, H, dsize=(w, h)) with the defaults the reference uses."""
    w, h = dsize
    Minv = np.linalg.inv(np.asarray(H, dtype=np.float64))
    sx, sy, alpha = perspective_coords(Minv, w, h)
    wt = bilinear_tab_i()[alpha]                                               # [h, w, 4]
    sh, sw = src.shape[:2]
    acc = np.zeros((h, w, src.shape[2]), dtype=np.int64)
    for k, (dy, dx) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        yy, xx = sy + dy, sx + dx
        ok = (yy >= 0) & (yy < sh) & (xx >= 0) & (xx < sw)
        v = src[np.clip(yy, 0, sh - 1), np.clip(xx, 0, sw - 1)].astype(np.int64)
        acc += np.where(ok[..., None], v, 0) * wt[..., k:k + 1]
    return np.clip((acc + (1 << (REMAP_COEF_BITS - 1))) >> REMAP_COEF_BITS, 0, 255).astype(np.uint8)


# ---------------------------------------------------------------------------------------------- resize
def resize_coeffs(ssize: int, dsize: int) -> Tuple[np.ndarray, np.ndarray]:
    """(source index, 11-bit weight pair) of every destination coordinate along one axis (resize.cpp)."""
    scale = ssize / dsize
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= ssize - 1
    f[hi], s[hi] = 0.0, ssize - 1
    c = np.stack([np.float32(1.0) - f, f], axis=1) * np.float32(1 << RESIZE_COEF_BITS)
    return s, np.clip(np.rint(c), -32768, 32767).astype(np.int64)


def resize_linear_u8(src: np.ndarray, dsize: Tuple[int, int]) -> np.ndarray:
    """cv2.resize(src[h, w, c] uint8, (w, h)) with INTER_LINEAR."""
    dw, dh = dsize
    sh, sw = src.shape[:2]
    if (dw, dh) == (sw, sh):
        return src.copy()
    sx, ax = resize_coeffs(sw, dw)
    sy, ay = resize_coeffs(sh, dh)
    s = src.astype(np.int64)
    x1 = np.minimum(sx + 1, sw - 1)
    rows = s[:, sx] * ax[None, :, 0:1] + s[:, x1] * ax[None, :, 1:2]          # horizontal pass: [sh, dw, c], << 11
    y1 = np.minimum(sy + 1, sh - 1)
    S0, S1 = rows[sy], rows[y1]
    b0, b1 = ay[:, 0][:, None, None], ay[:, 1][:, None, None]
    out = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


# ---------------------------------------------------------------------------------------------- Lab
_LAB_SHIFT, _GAMMA_SHIFT = 12, 3
_LAB_SHIFT2 = _LAB_SHIFT + _GAMMA_SHIFT
_XYZ = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])
_XYZ_INV = np.array([[3.240479, -1.53715, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]])
_D65 = np.array([0.950456, 1.0, 1.088754])


def lab_tables() -> Dict[str, np.ndarray]:
    x = np.arange(256, dtype=np.float32) * np.float32(1.0 / 255.0)
    g = np.where(x <= np.float32(0.04045), x * np.float32(1.0 / 12.92),
                 np.power((x.astype(np.float64) + 0.055) * (1.0 / 1.055), 2.4).astype(np.float32))
    gamma = np.clip(np.rint(np.float32(255.0 * (1 << _GAMMA_SHIFT)) * g), 0, 65535).astype(np.int64)
    n = 256 * 3 // 2 * (1 << _GAMMA_SHIFT)
    t = np.arange(n, dtype=np.float32) * np.float32(1.0 / (255.0 * (1 << _GAMMA_SHIFT)))
    c = np.where(t < np.float32(0.008856), t * np.float32(7.787) + np.float32(0.13793103448275862),
                 np.cbrt(t.astype(np.float64)).astype(np.float32))
    cbrt = np.clip(np.rint(np.float32(1 << _LAB_SHIFT2) * c), 0, 65535).astype(np.int64)
    coef = np.rint((1 << _LAB_SHIFT) * _XYZ / _D65[:, None]).astype(np.int64)       # rows X, Y, Z over (R, G, B)
    return {"gamma": gamma, "cbrt": cbrt, "coef": coef}


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def rgb2lab_u8(img: np.ndarray, bgr: bool = False) -> np.ndarray:
    """cv2.cvtColor(img uint8 [.., 3], COLOR_RGB2LAB) (bgr=True: COLOR_BGR2LAB): L*255/100, a+128, b+128."""
    t = lab_tables()
    v = t["gamma"][img.astype(np.int64)]
    R, G, B = (v[..., 2], v[..., 1], v[..., 0]) if bgr else (v[..., 0], v[..., 1], v[..., 2])
    C = t["coef"]
    fX = t["cbrt"][_descale(R * C[0, 0] + G * C[0, 1] + B * C[0, 2], _LAB_SHIFT)]
    fY = t["cbrt"][_descale(R * C[1, 0] + G * C[1, 1] + B * C[1, 2], _LAB_SHIFT)]
    fZ = t["cbrt"][_descale(R * C[2, 0] + G * C[2, 1] + B * C[2, 2], _LAB_SHIFT)]
    Lscale = (116 * 255 + 50) // 100
    Lshift = -((16 * 255 * (1 << _LAB_SHIFT2) + 50) // 100)
    L = _descale(Lscale * fY + Lshift, _LAB_SHIFT2)
    a = _descale(500 * (fX - fY) + 128 * (1 << _LAB_SHIFT2), _LAB_SHIFT2)
    b = _descale(200 * (fY - fZ) + 128 * (1 << _LAB_SHIFT2), _LAB_SHIFT2)
    return np.clip(np.stack([L, a, b], axis=-1), 0, 255).astype(np.uint8)


def lab2bgr_u8(lab: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(lab uint8, COLOR_LAB2BGR): float path, sRGB gamma, rounded to uint8."""
    x = lab.astype(np.float32)
    L = x[..., 0] * np.float32(100.0 / 255.0)
    a = x[..., 1] - np.float32(128.0)
    b = x[..., 2] - np.float32(128.0)
    fy = (L + np.float32(16.0)) / np.float32(116.0)
    Y = np.where(L <= np.float32(903.3 * 0.008856), L / np.float32(903.3), fy * fy * fy)
    fy = np.where(L <= np.float32(903.3 * 0.008856), np.float32(7.787) * Y + np.float32(16.0 / 116.0), fy)
    fx = fy + a / np.float32(500.0)
    fz = fy - b / np.float32(200.0)

    def finv(f):
        return np.where(f <= np.float32(6.0 / 29.0), (f - np.float32(16.0 / 116.0)) / np.float32(7.787), f * f * f)

    X = finv(fx) * np.float32(_D65[0])
    Z = finv(fz) * np.float32(_D65[2])
    m = _XYZ_INV.astype(np.float32)
    rgb = [np.clip(m[i, 0] * X + m[i, 1] * Y + m[i, 2] * Z, 0.0, 1.0) for i in range(3)]

    def gam(v):
        return np.where(v <= np.float32(0.0031308), v * np.float32(12.92),
                        np.float32(1.055) * np.power(v.astype(np.float64), 1.0 / 2.4).astype(np.float32) - np.float32(0.055))

    out = np.stack([gam(rgb[2]), gam(rgb[1]), gam(rgb[0])], axis=-1) * np.float32(255.0)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


# ---------------------------------------------------------------------------------------------- fillPoly
def _line_mask(mask: np.ndarray, p0, p1) -> None:
    """cv::Line, 8-connected, drawn left to right (LineIterator, leftToRight = true), clipped to the image."""
    (x0, y0), (x1, y1) = (int(p0[0]), int(p0[1])), (int(p1[0]), int(p1[1]))
    if x1 < x0:
        x0, y0, x1, y1 = x1, y1, x0, y0
    dx, dy = x1 - x0, abs(y1 - y0)
    sy = 1 if y1 >= y0 else -1
    h, w = mask.shape
    if dx >= dy:                                        # x-major: after k steps y has moved floor((2 dy k + dx - 1) / (2 dx))
        k = np.arange(dx + 1)
        xs = x0 + k
        ys = y0 + sy * ((2 * dy * k + dx - 1) // (2 * dx) if dx else np.zeros_like(k))
    else:
        k = np.arange(dy + 1)
        ys = y0 + sy * k
        xs = x0 + (2 * dx * k + dy - 1) // (2 * dy)
    ok = (xs >= 0) & (xs < w) & (ys >= 0) & (ys < h)
    mask[ys[ok], xs[ok]] = 1


def fill_poly_mask(shape: Tuple[int, int], pts: np.ndarray) -> np.ndarray:
    """uint8 [h, w] = 1 inside or on the outline of the polygon `pts` (int32 [n, 2] as (x, y)); drawing.cpp
    CollectPolyEdges + FillEdgeCollection at shift 0, even-odd rule."""
    h, w = shape
    mask = np.zeros((h, w), dtype=np.uint8)
    pts = np.asarray(pts, dtype=np.int64)
    n = len(pts)
    edges = []
    for i in range(n):
        p0, p1 = pts[i - 1], pts[i]
        _line_mask(mask, p0, p1)
        if p0[1] == p1[1]:
            continue
        a, b = (p0, p1) if p0[1] < p1[1] else (p1, p0)
        num, den = (int(p1[0]) - int(p0[0])) << 16, int(p1[1]) - int(p0[1])
        dxf = abs(num) // abs(den) * (1 if (num >= 0) == (den > 0) else -1)      # C++ integer division truncates
        edges.append((int(a[1]), int(b[1]), int(a[0]) << 16, dxf))
    if not edges:
        return mask
    y_lo, y_hi = min(e[0] for e in edges), max(e[1] for e in edges)
    for y in range(max(y_lo, 0), min(y_hi, h)):
        xs = sorted(x0 + dxf * (y - y0) for (y0, y1, x0, dxf) in edges if y0 <= y < y1)
        for j in range(0, len(xs) - 1, 2):
            x1 = (xs[j] + (1 << 16) - 1) >> 16
            x2 = xs[j + 1] >> 16
            if x1 < w and x2 >= 0:
                mask[y, max(x1, 0):min(x2, w - 1) + 1] = 1
    return mask


# ---------------------------------------------------------------------------------------------- findHomography
def find_homography(src: np.ndarray, dst: np.ndarray):
    """cv2.findHomography(src, dst)[0] with method 0: Hartley-normalised DLT over all correspondences, refined by
    Gauss-Newton on the reprojection error; h33 = 1.  Returns None for degenerate input (< 4 points or rank loss)."""
    s = np.asarray(src, dtype=np.float64).reshape(-1, 2)
    d = np.asarray(dst, dtype=np.float64).reshape(-1, 2)
    if len(s) < 4 or len(s) != len(d):
        return None

    def norm(p):
        c = p.mean(axis=0)
        sc = np.abs(p - c).mean(axis=0)
        if (sc < 1e-12).any():
            return None, None
        sc = 1.0 / sc
        T = np.array([[sc[0], 0, -c[0] * sc[0]], [0, sc[1], -c[1] * sc[1]], [0, 0, 1.0]])
        return (p - c) * sc, T

    sn, Ts = norm(s)
    dn, Td = norm(d)
    if sn is None or dn is None:
        return None
    A = []
    for (x, y), (u, v) in zip(sn, dn):
        A.append([x, y, 1, 0, 0, 0, -u * x, -u * y, -u])
        A.append([0, 0, 0, x, y, 1, -v * x, -v * y, -v])
    A = np.asarray(A)
    _, sv, vt = np.linalg.svd(A)
    if sv[-2] < 1e-12:
        return None
    Hn = vt[-1].reshape(3, 3)
    H = np.linalg.inv(Td) @ Hn @ Ts
    if abs(H[2, 2]) < 1e-300:
        return None
    H = H / H[2, 2]
    if len(s) > 4:
        hv = H.reshape(-1)[:8].copy()
        for _ in range(10):
            Hc = np.append(hv, 1.0).reshape(3, 3)
            p = np.c_[s, np.ones(len(s))] @ Hc.T
            wv = p[:, 2:3]
            r = (p[:, :2] / wv - d).reshape(-1)
            J = np.zeros((2 * len(s), 8))
            for i, ((x, y), (px, py, pw)) in enumerate(zip(s, p)):
                J[2 * i] = [x / pw, y / pw, 1 / pw, 0, 0, 0, -px * x / pw ** 2, -px * y / pw ** 2]
                J[2 * i + 1] = [0, 0, 0, x / pw, y / pw, 1 / pw, -py * x / pw ** 2, -py * y / pw ** 2]
            step = np.linalg.lstsq(J, -r, rcond=None)[0]
            hv = hv + step
            if np.abs(step).max() < 1e-12:
                break
        H = np.append(hv, 1.0).reshape(3, 3)
    return H


# ---------------------------------------------------------------------------------------------- reference control flow
PASCAL_CAR_PLANES = ("left", "right", "roof", "front", "back")      # warp_learn/online_visibility.py:9-23 key order


def get_planes(image: np.ndarray, plane_points: Sequence[np.ndarray]) -> np.ndarray:
    """warp_learn/planes_utils.py:11-37 for already-scaled int32 polygons: image * fillPoly mask per plane."""
    return np.stack([image * fill_poly_mask(image.shape[:2], p)[..., None] for p in plane_points], 0)


def warp_unwarp_planes(src_planes: np.ndarray, src_kp: List[np.ndarray], dst_kp: List[np.ndarray],
                       src_vis: Sequence[int], dst_vis: Sequence[int], keys: Sequence[str] = PASCAL_CAR_PLANES):
    """warp_learn/planes_utils.py:40-82: visibility / symmetry gating, H12 and H21, warp then un-warp."""
    warped = np.zeros_like(src_planes)
    unwarped = np.zeros_like(src_planes)
    sym = [keys.index("left"), keys.index("right")]
    for i in range(len(keys)):
        if not src_vis[i]:
            continue
        if i not in sym and not dst_vis[i]:
            continue
        if i in sym and 1 not in [dst_vis[j] for j in sym]:
            continue
        j = i
        if i in sym and not dst_vis[i]:
            j = sym[0] if i == sym[1] else sym[1]
        H12 = find_homography(src_kp[i], dst_kp[j])
        H21 = find_homography(dst_kp[j], src_kp[i])
        if H12 is not None and H21 is not None:
            h, w = src_planes[0].shape[:2]
            sw = warp_perspective_u8(src_planes[i], H12, (w, h))
            warped[j] = sw
            unwarped[i] = warp_perspective_u8(sw, H21, (w, h))
    return warped, unwarped


def square_crop_geometry(image_hw: Tuple[int, int], bbox: Sequence[int]):
    """utils/crop_utils.py:4-52 ('pascal' branch) without the pixel copy: ((x0, y0, x1, y1) in the padded image,
    pad_before (x, y), pad_after (x, y))."""
    image_h, image_w = image_hw
    x_min, y_min, x_max, y_max = [int(v) for v in bbox]
    side_x, side_y = x_max - x_min, y_max - y_min
    major = max(side_x, side_y) * 1.1
    cx, cy = x_min + side_x / 2, y_min + side_y / 2
    pxb = pxa = pyb = pya = 0
    nx0 = int(cx - major / 2.0)
    if nx0 < 0:
        pxb, nx0 = int(np.ceil(abs(nx0))), 0
    nx1 = int(cx + major / 2.0) + pxb
    if nx1 > image_w:
        pxa = int(np.ceil(abs(nx1 - image_w)))
        nx1 = image_w + pxa
    ny0 = int(cy - major / 2.0)
    if ny0 < 0:
        pyb, ny0 = int(np.ceil(abs(ny0))), 0
    ny1 = int(cy + major / 2.0) + pyb
    if ny1 > image_h:
        pya = int(np.ceil(abs(ny1 - image_h)))
        ny1 = image_h + pya
    return (nx0, ny0, nx1, ny1), (pxb, pyb), (pxa, pya)


def square_crop(image: np.ndarray, bbox: Sequence[int]) -> np.ndarray:
    (x0, y0, x1, y1), (pxb, pyb), (pxa, pya) = square_crop_geometry(image.shape[:2], bbox)
    padded = np.pad(image, [(pyb, pya), (pxb, pxa), (0, 0)], mode="constant")
    return padded[y0:y1, x0:x1]


def get_icn_inputs(planes: np.ndarray, sketch_normal: np.ndarray, sketch_mask: np.ndarray, central_crop: np.ndarray,
                   icn_w: int, icn_h: int):
    """warp_learn/models.py:323-366: float32 [1, 21, icn_h, icn_w] in [-1, 1] + crop_info."""
    ys, xs = np.nonzero(sketch_mask)
    bbox = [int(xs.min()), int(ys.min()), int(xs.max()), int(ys.max())]
    (x0, y0, x1, y1), pb, pa = square_crop_geometry(sketch_normal.shape[:2], bbox)
    info = {"crop_xy_min": (x0, y0), "pad_xy_before": pb, "pad_xy_after": pa, "crop_size_orig": (y1 - y0, x1 - x0)}

    def prep(img, bgr):
        lab = rgb2lab_u8(img, bgr=bgr)
        return (np.transpose(lab.astype(np.float32) / np.float32(255.0), (2, 0, 1)) - np.float32(0.5)) / np.float32(0.5)

    sk = prep(resize_linear_u8(square_crop(sketch_normal, bbox), (icn_w, icn_h)), False)        # RGB2LAB, models.py:355
    cc = prep(central_crop, False)                                                              # models.py:358
    pl = [prep(resize_linear_u8(square_crop(p, bbox), (icn_w, icn_h)), True) for p in planes]   # BGR2LAB, planes_utils.py:88
    return np.concatenate([sk, cc] + pl, axis=0)[None], info


def paste_back(frame_out: np.ndarray, net_image: np.ndarray, crop_info: dict, paste_mask: np.ndarray) -> np.ndarray:
    """trajectory_inference.py:184-198: resize the network image back to the crop's size, drop the padding, place it at
    crop_xy_min in an empty frame and copy the masked pixels into `frame_out` (in place; later calls overwrite)."""
    hh, ww = crop_info["crop_size_orig"]
    inv = resize_linear_u8(net_image, (ww, hh))
    pb, pa = crop_info["pad_xy_before"], crop_info["pad_xy_after"]
    inv = inv[pb[1]:inv.shape[0] - pa[1], pb[0]:inv.shape[1] - pa[0]]
    x0, y0 = crop_info["crop_xy_min"]
    canvas = np.zeros_like(frame_out)
    canvas[y0:y0 + inv.shape[0], x0:x0 + inv.shape[1]] = inv
    frame_out[paste_mask] = canvas[paste_mask]
    return frame_out
