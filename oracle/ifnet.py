"""Oracle RIFE 4.26-heavy IFNet, functional over a plain state dict.  (test infra)

Restates models/rife_426_heavy/IFNet_HDv3.py with torch.nn.functional calls only, so
the same code also serves as the per-layer reference for the HIP conv kernels.
"""
import torch
import torch.nn.functional as F

from .ops import backwarp

LRELU = 0.2


def lrelu(x):
    return F.leaky_relu(x, LRELU)


def head(sd, x, prefix="encode."):
    """Context encoder — IFNet_HDv3.py:28-47: conv3x3 s2 (3->16), two conv3x3 (16->16),
    each followed by LeakyReLU(0.2), then ConvTranspose2d 4x4 s2 p1 back to full size."""
    x = lrelu(F.conv2d(x, sd[prefix + "cnn0.weight"], sd[prefix + "cnn0.bias"], stride=2, padding=1))
    x = lrelu(F.conv2d(x, sd[prefix + "cnn1.weight"], sd[prefix + "cnn1.bias"], stride=1, padding=1))
    x = lrelu(F.conv2d(x, sd[prefix + "cnn2.weight"], sd[prefix + "cnn2.bias"], stride=1, padding=1))
    return F.conv_transpose2d(x, sd[prefix + "cnn3.weight"], sd[prefix + "cnn3.bias"], stride=2, padding=1)


def resconv(sd, prefix, x):
    """ResConv — IFNet_HDv3.py:50-59: lrelu(conv3x3(x) * beta + x)."""
    y = F.conv2d(x, sd[prefix + "conv.weight"], sd[prefix + "conv.bias"], stride=1, padding=1)
    return lrelu(y * sd[prefix + "beta"] + x)


def ifblock_core(sd, prefix, x):
    """conv0 (two stride-2 convs) -> 8 ResConv -> ConvTranspose2d(c, 52) -> PixelShuffle(2).
    IFNet_HDv3.py:65-82, :89-91."""
    x = lrelu(F.conv2d(x, sd[prefix + "conv0.0.0.weight"], sd[prefix + "conv0.0.0.bias"], stride=2, padding=1))
    x = lrelu(F.conv2d(x, sd[prefix + "conv0.1.0.weight"], sd[prefix + "conv0.1.0.bias"], stride=2, padding=1))
    for j in range(8):
        x = resconv(sd, prefix + f"convblock.{j}.", x)
    x = F.conv_transpose2d(x, sd[prefix + "lastconv.0.weight"], sd[prefix + "lastconv.0.bias"], stride=2, padding=1)
    return F.pixel_shuffle(x, 2)


def ifblock(sd, prefix, x, flow, scale):
    """IFBlock.forward — IFNet_HDv3.py:84-96."""
    x = F.interpolate(x, scale_factor=1.0 / scale, mode="bilinear", align_corners=False)
    if flow is not None:
        flow = F.interpolate(flow, scale_factor=1.0 / scale, mode="bilinear", align_corners=False) * 1.0 / scale
        x = torch.cat((x, flow), 1)
    tmp = ifblock_core(sd, prefix, x)
    tmp = F.interpolate(tmp, scale_factor=scale, mode="bilinear", align_corners=False)
    return tmp[:, :4] * scale, tmp[:, 4:5], tmp[:, 5:]


def ifnet(sd, x, timestep=0.5, scale_list=(8, 4, 2, 1), f0=None, f1=None, trace=None):
    """IFNet.forward (inference branch) — IFNet_HDv3.py:126-177.

    x = cat(img0, img1); timestep scalar or [1,1,H,W] map (used as-is); five
    coarse-to-fine stages; returns (merged frame, flow_list).  `trace`, if a dict, is
    filled with per-stage intermediates for kernel-level parity tests.
    """
    c = x.shape[1] // 2
    img0, img1 = x[:, :c], x[:, c:]
    if not torch.is_tensor(timestep):
        timestep = (x[:, :1].clone() * 0 + 1) * timestep
    f0 = head(sd, img0[:, :3]) if f0 is None else f0
    f1 = head(sd, img1[:, :3]) if f1 is None else f1
    flow_list = []
    w0, w1 = img0, img1
    flow = mask = feat = None
    for i in range(5):
        p = f"block{i}."
        if flow is None:
            xin = torch.cat((img0[:, :3], img1[:, :3], f0, f1, timestep), 1)
            flow, mask, feat = ifblock(sd, p, xin, None, scale_list[i])
        else:
            wf0 = backwarp(f0, flow[:, :2])
            wf1 = backwarp(f1, flow[:, 2:4])
            xin = torch.cat((w0[:, :3], w1[:, :3], wf0, wf1, timestep, mask, feat), 1)
            fd, mask, feat = ifblock(sd, p, xin, flow, scale_list[i])
            flow = flow + fd
        flow_list.append(flow)
        w0 = backwarp(img0, flow[:, :2])
        w1 = backwarp(img1, flow[:, 2:4])
        if trace is not None:
            trace[f"flow{i}"] = flow
            trace[f"mask{i}"] = mask
            trace[f"feat{i}"] = feat
    m = torch.sigmoid(mask)
    return w0 * m + w1 * (1 - m), flow_list
