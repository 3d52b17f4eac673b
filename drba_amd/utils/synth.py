"""Deterministic synthetic weights and frames.

The RIFE 4.26-heavy checkpoint is not shipped with the reference
(its .MISSING_LARGE_BLOBS lists weights/train_log_rife_426_heavy/flownet.pkl), and there is no network, so benchmarks and
parity tests run on seeded random weights of the exact architecture and on seeded
synthetic clips.  Everything here is pure CPU torch/numpy and regenerates bit-identically
from the seed on any box with the same torch build, so nothing large is committed.

State-dict key/shape inventory follows the reference module tree
(models/rife_426_heavy/IFNet_HDv3.py:28-47 Head, :50-59 ResConv, :62-82 IFBlock,
:99-106 IFNet).
"""
import zlib

import numpy as np
import torch

IFNET_BLOCK_C = (192, 128, 96, 64, 32)  # IFNet_HDv3.py:102-106
IFNET_BLOCK_IN = (7 + 32, 8 + 4 + 8 + 32, 8 + 4 + 8 + 32, 8 + 4 + 8 + 32, 8 + 4 + 8 + 32)


def ifnet_shapes():
    """Ordered {key: shape} of IFNet().state_dict() (158 tensors, 5 723 156 params)."""
    shapes = {}
    for i, (c, cin) in enumerate(zip(IFNET_BLOCK_C, IFNET_BLOCK_IN)):
        p = f"block{i}."
        shapes[p + "conv0.0.0.weight"] = (c // 2, cin, 3, 3)
        shapes[p + "conv0.0.0.bias"] = (c // 2,)
        shapes[p + "conv0.1.0.weight"] = (c, c // 2, 3, 3)
        shapes[p + "conv0.1.0.bias"] = (c,)
        for j in range(8):
            q = p + f"convblock.{j}."
            shapes[q + "beta"] = (1, c, 1, 1)
            shapes[q + "conv.weight"] = (c, c, 3, 3)
            shapes[q + "conv.bias"] = (c,)
        shapes[p + "lastconv.0.weight"] = (c, 4 * 13, 4, 4)  # ConvTranspose2d: [Cin, Cout, kh, kw]
        shapes[p + "lastconv.0.bias"] = (4 * 13,)
    shapes["encode.cnn0.weight"] = (16, 3, 3, 3)
    shapes["encode.cnn0.bias"] = (16,)
    for k in (1, 2):
        shapes[f"encode.cnn{k}.weight"] = (16, 16, 3, 3)
        shapes[f"encode.cnn{k}.bias"] = (16,)
    shapes["encode.cnn3.weight"] = (16, 16, 4, 4)  # ConvTranspose2d
    shapes["encode.cnn3.bias"] = (16,)
    return shapes


def _gen(key, seed):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def ifnet_state_dict(seed=0):
    """Seeded fp32 IFNet weights with activations that stay O(1) through the 5 stages.

    conv weights ~ N(0, (g/sqrt(fan_in))^2); ResConv beta ~ U(0.25, 0.75); the flow head
    (lastconv) is scaled so per-stage flow updates are a few pixels, which keeps the
    warps non-trivial but smooth (see SURVEY.md 'Hard parts' on discontinuities).
    """
    sd = {}
    for key, shape in ifnet_shapes().items():
        g = _gen(key, seed)
        if key.endswith("beta"):
            t = torch.rand(shape, generator=g) * 0.5 + 0.25
        elif key.endswith("bias"):
            t = torch.randn(shape, generator=g) * 0.02
        else:
            if "lastconv" in key or "cnn3" in key:  # ConvTranspose2d: each output gets 2x2 taps
                fan_in = shape[0] * 4
            else:
                fan_in = shape[1] * shape[2] * shape[3]
            gain = 1.0
            if "convblock" in key:
                gain = 0.7
            if "lastconv" in key:
                gain = 0.12
            t = torch.randn(shape, generator=g) * (gain / fan_in ** 0.5)
        sd[key] = t.contiguous()
    return sd


def _smooth_field(c, h, w, seed, cell=16):
    """Smooth random texture in [0,1]: bicubic upsample of a coarse uniform grid."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    coarse = torch.rand(1, c, h // cell + 3, w // cell + 3, generator=g)
    up = torch.nn.functional.interpolate(coarse, scale_factor=cell, mode="bicubic", align_corners=False)
    return up[:, :, cell:cell + h, cell:cell + w].clamp(0, 1)


def make_clip(n_frames, height, width, seed=1234, cut_at=None, offsets=None):
    """uint8 HWC 'BGR' frames of a two-octave smooth texture under non-uniform translation.

    Non-uniform per-frame offsets make the DistanceRatioMap non-trivial;
    the coarse octave keeps neighbouring frames similar at thumbnail scale (no false scene
    cuts), the fine octave gives the flow network local structure.
    `cut_at`: frame index where an independently seeded scene starts (scene-cut tests).
    Returns a list of np.uint8 arrays [H, W, 3].
    """
    unit = max(1, width // 240)
    if offsets is None:
        steps = [1, 2, 1, 3, 1, 2, 1, 2]  # never 0: consecutive frames are always distinct
        offsets, acc = [], 0
        for k in range(n_frames):
            offsets.append(acc)
            acc += steps[k % len(steps)] * unit
    margin = max(offsets) + 8
    coarse = max(16, (width // 8) // 16 * 16)

    def scene(sd):
        hh, ww = height + margin // 2 + 8, width + margin
        return 0.7 * _smooth_field(3, hh, ww, sd, cell=coarse) + 0.3 * _smooth_field(3, hh, ww, sd + 1, cell=16)

    base_a = scene(seed)
    base_b = scene(seed + 7919) if cut_at is not None else None
    frames = []
    for k in range(n_frames):
        base = base_b if (cut_at is not None and k >= cut_at) else base_a
        dx = offsets[k]
        dy = offsets[k] // 2
        crop = base[0, :, dy:dy + height, dx:dx + width]
        frames.append((crop.permute(1, 2, 0).numpy() * 255.0).astype(np.uint8).copy())
    return frames


def make_triplet_tensors(height, width, seed=1234, device="cpu"):
    """Three consecutive fp32 NCHW frames in [0,1] already at network size."""
    fr = make_clip(3, height, width, seed=seed)
    return [torch.from_numpy(f.transpose(2, 0, 1)).unsqueeze(0).float().div(255.0).to(device) for f in fr]
