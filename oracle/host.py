"""Oracle: the small host-side tensor<->image helpers adjacent to the networks, plus the SSIM metric
the quality bar is stated in (the reference has no SSIM implementation; this one is build-owned)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def to_image_u8(x: torch.Tensor) -> np.ndarray:
    """to_image(x, from_LAB=False), warp_learn/planes_utils.py:96-118: (x+1)/2*255, clip to [0,255],
    astype(uint8) (= truncation), CHW -> HWC.  Accepts [3,H,W] or [B,3,H,W].  The LAB->BGR branch
    is OpenCV arithmetic and is not restated (parity unpinned)."""
    a = x.detach().to("cpu").numpy()
    a = np.transpose(a, (1, 2, 0)) if a.ndim == 3 else np.transpose(a, (0, 2, 3, 1))
    a = (a + 1.) / 2 * 255
    a = np.clip(a, 0, 255)
    return a.astype(np.uint8)


def to_tensor_pm1(image: np.ndarray, max_range: int = 255) -> torch.Tensor:
    """to_tensor, utils/misc_utils.py:35-49: uint8 HWC in [0,max_range] -> float CHW in [-1,1]."""
    image = np.float32(image)
    assert image.max() <= max_range
    image = image / max_range
    image = np.transpose(image, (2, 0, 1))
    image = image * 2. - 1.
    return torch.from_numpy(image)


def ssim(a: np.ndarray, b: np.ndarray, data_range: float = 255.0) -> float:
    """Mean SSIM (Wang et al. 2004: 11x11 Gaussian window sigma 1.5, K1=0.01, K2=0.03) between two
    uint8/float images [H,W,C] or batches [B,H,W,C]; channels and batch are averaged."""
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64)
    if a.dim() == 3:
        a, b = a[None], b[None]
    a = a.permute(0, 3, 1, 2)
    b = b.permute(0, 3, 1, 2)
    c = a.shape[1]
    g = torch.arange(11, dtype=torch.float64) - 5
    g = torch.exp(-(g ** 2) / (2 * 1.5 ** 2))
    g = (g / g.sum())
    win = (g[:, None] * g[None, :]).expand(c, 1, 11, 11).contiguous()
    mu_a = F.conv2d(a, win, groups=c)
    mu_b = F.conv2d(b, win, groups=c)
    s_aa = F.conv2d(a * a, win, groups=c) - mu_a ** 2
    s_bb = F.conv2d(b * b, win, groups=c) - mu_b ** 2
    s_ab = F.conv2d(a * b, win, groups=c) - mu_a * mu_b
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    m = ((2 * mu_a * mu_b + c1) * (2 * s_ab + c2)) / ((mu_a ** 2 + mu_b ** 2 + c1) * (s_aa + s_bb + c2))
    return float(m.mean())
