"""Oracle operators: backward warp, forward splat, flow distance, resize.  (test infra)"""
import torch
import torch.nn.functional as F


def backwarp(x, flow):
    """Backward bilinear warp with border clamp.

    Follows models/rife_426_heavy/warplayer.py:8-22: base grid linspace(-1,1) per axis,
    flow normalised by (W-1)/2 and (H-1)/2, grid_sample(bilinear, border,
    align_corners=True).  out[c,y,x] = bilinear(in[c], x+fx, y+fy) clamped to the image.
    """
    n, _, h, w = flow.shape
    gx = torch.linspace(-1.0, 1.0, w).view(1, 1, 1, w).expand(n, 1, h, w)
    gy = torch.linspace(-1.0, 1.0, h).view(1, 1, h, 1).expand(n, 1, h, w)
    fx = flow[:, 0:1] / ((x.shape[3] - 1.0) / 2.0)
    fy = flow[:, 1:2] / ((x.shape[2] - 1.0) / 2.0)
    grid = (torch.cat([gx, gy], 1) + torch.cat([fx, fy], 1)).permute(0, 2, 3, 1)
    return F.grid_sample(x, grid, mode="bilinear", padding_mode="border", align_corners=True)


def backwarp_zeros(x, flow):
    """MetricNet's backwarp: same grid, zeros padding (models/model_gmfss_union/MetricNet.py:10-20)."""
    n, _, h, w = flow.shape
    gx = torch.linspace(-1.0, 1.0, w).view(1, 1, 1, w).expand(n, 1, h, w)
    gy = torch.linspace(-1.0, 1.0, h).view(1, 1, h, 1).expand(n, 1, h, w)
    fx = flow[:, 0:1] / ((x.shape[3] - 1.0) / 2.0)
    fy = flow[:, 1:2] / ((x.shape[2] - 1.0) / 2.0)
    grid = (torch.cat([gx, gy], 1) + torch.cat([fx, fy], 1)).permute(0, 2, 3, 1)
    return F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=True)


def splat_sum(src, flow):
    """Un-normalised forward splat (summation splatting).

    Follows models/softsplat/softsplat_torch.py:70-179 (== CUDA kernel
    models/softsplat/softsplat.py:306-367): every source pixel (y,x) lands at
    (x+fx, y+fy); non-finite targets are skipped; its value is spread over the four
    integer neighbours with bilinear weights, each corner bounds-checked independently;
    contributions are accumulated in corner order NW, NE, SW, SE with sequential
    index_add_ (that order is what makes the CPU result reproducible).
    """
    n, c, h, w = src.shape
    dt = src.dtype
    ys, xs = torch.meshgrid(torch.arange(h, dtype=dt), torch.arange(w, dtype=dt), indexing="ij")
    tx = xs.expand(n, h, w) + flow[:, 0]
    ty = ys.expand(n, h, w) + flow[:, 1]
    out = torch.zeros(n * h * w, c, dtype=dt)
    ok = torch.isfinite(tx) & torch.isfinite(ty)
    if not bool(ok.any()):
        return out.view(n, h, w, c).permute(0, 3, 1, 2).contiguous()
    vals = src.permute(0, 2, 3, 1)[ok]  # [M, C]
    bidx = torch.arange(n).view(n, 1, 1).expand(n, h, w)[ok]
    tx, ty = tx[ok], ty[ok]
    x0 = torch.floor(tx).to(torch.int32)
    y0 = torch.floor(ty).to(torch.int32)
    for dx, dy in ((0, 0), (1, 0), (0, 1), (1, 1)):
        wx = ((x0 + 1) - tx) if dx == 0 else (tx - x0)
        wy = ((y0 + 1) - ty) if dy == 0 else (ty - y0)
        wgt = wx * wy
        cx, cy = x0 + dx, y0 + dy
        inb = (cx >= 0) & (cx < w) & (cy >= 0) & (cy < h)
        if not bool(inb.any()):
            continue
        lin = bidx[inb] * h * w + cy[inb] * w + cx[inb]
        out.index_add_(0, lin, vals[inb] * wgt[inb].unsqueeze(1))
    # NCHW-contiguous result: that is the layout the reference's softsplat_func.apply hands back (observed), and
    # oneDNN convolutions downstream (GridNet) round differently for channels-last inputs
    return out.view(n, h, w, c).permute(0, 3, 1, 2).contiguous()


def softsplat(src, flow, metric, mode):
    """softsplat(tenIn, tenFlow, tenMetric, strMode) — models/softsplat/softsplat_torch.py:19-67.

    mode = main[-sub]; main in {sum, avg, linear, soft}; sub in {addeps, zeroeps, clipeps}.
    avg: append a ones channel; linear: [in*m, m]; soft: [in*exp(m), exp(m)] (no max
    subtraction); result = out[:-1] / norm(out[-1]) with norm = +1e-7 by default.
    """
    parts = mode.split("-")
    main, sub = parts[0], (parts[1] if len(parts) > 1 else None)
    assert main in ("sum", "avg", "linear", "soft")
    if main in ("sum", "avg"):
        assert metric is None
    else:
        assert metric is not None
    if main == "avg":
        src = torch.cat([src, src.new_ones(src.shape[0], 1, src.shape[2], src.shape[3])], 1)
    elif main == "linear":
        src = torch.cat([src * metric, metric], 1)
    elif main == "soft":
        e = metric.exp()
        src = torch.cat([src * e, e], 1)
    out = splat_sum(src, flow)
    if main == "sum":
        return out
    norm = out[:, -1:]
    if sub in (None, "addeps"):
        norm = norm + 0.0000001
    elif sub == "zeroeps":
        norm = torch.where(norm == 0.0, torch.tensor(1.0), norm)
    elif sub == "clipeps":
        norm = norm.clip(0.0000001, None)
    return out[:, :-1] / norm


def distance(flow):
    """|flow| per pixel, computed in fp32 — models/utils/tools.py:77-80."""
    dt = flow.dtype
    u, v = flow[:, 0:1].float(), flow[:, 1:].float()
    return torch.sqrt(u ** 2 + v ** 2).to(dt)


def resize(x, size):
    """Bilinear resize to an explicit size, align_corners=False — models/utils/tools.py:71-72."""
    return F.interpolate(x, size=size, mode="bilinear", align_corners=False)
