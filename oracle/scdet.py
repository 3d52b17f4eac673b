"""Oracle scene-cut detector — restates models/utils/tools.py:27-30 (check_scene) and
models/pytorch_msssim/__init__.py:83-136 (ssim_matlab).  (test infra)"""
from math import exp

import torch
import torch.nn.functional as F


def gaussian_window_3d(size=11, sigma=1.5):
    """pytorch_msssim/__init__.py:9-11, :21-26: normalised 1-D gaussian, outer-producted to 3-D."""
    g = torch.Tensor([exp(-(x - size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(size)])
    g = (g / g.sum()).unsqueeze(1)
    w2 = g.mm(g.t())
    w3 = w2.unsqueeze(2) @ g.t()
    return w3.expand(1, 1, size, size, size).contiguous()


def ssim_matlab(img1, img2, window_size=11):
    """SSIM of two [N,3,H,W] images treated as 3-D volumes (channel axis convolved too),
    replicate padding 5 on all three axes, val_range inferred from img1."""
    max_val = 255 if torch.max(img1) > 128 else 1
    min_val = -1 if torch.min(img1) < -0.5 else 0
    L = max_val - min_val
    _, _, h, w = img1.size()
    window = gaussian_window_3d(min(window_size, h, w))
    a = img1.unsqueeze(1)
    b = img2.unsqueeze(1)
    pad = (5, 5, 5, 5, 5, 5)

    def blur(v):
        return F.conv3d(F.pad(v, pad, mode="replicate"), window, padding=0, groups=1)

    mu1, mu2 = blur(a), blur(b)
    mu1_sq, mu2_sq, mu12 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = blur(a * a) - mu1_sq
    s2 = blur(b * b) - mu2_sq
    s12 = blur(a * b) - mu12
    C1 = (0.01 * L) ** 2
    C2 = (0.03 * L) ** 2
    v1 = 2.0 * s12 + C2
    v2 = s1 + s2 + C2
    return (((2 * mu12 + C1) * v1) / ((mu1_sq + mu2_sq + C1) * v2)).mean()


def check_scene(x1, x2, scdet_threshold=0.3):
    """tools.py:27-30: 32x32 bilinear thumbnails -> ssim_matlab < threshold (0-dim bool tensor)."""
    x1 = F.interpolate(x1, (32, 32), mode="bilinear", align_corners=False)
    x2 = F.interpolate(x2, (32, 32), mode="bilinear", align_corners=False)
    return ssim_matlab(x1, x2) < scdet_threshold
