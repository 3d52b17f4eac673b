"""Oracle GMFlow (optical flow by global matching), functional over a plain state dict.  (test infra)

Restates models/gmflow/{gmflow,backbone,trident_conv,transformer,matching,geometry,utils,position}.py for the
configuration DRBA uses: num_scales=2, upsample_factor=4, feature_channels=128, swin attention, 6 layers,
1 head, ffn expansion 4; forward(attn_splits_list=[2,8], corr_radius_list=[-1,4], prop_radius_list=[-1,1]).
"""
import math

import torch
import torch.nn.functional as F

C = 128  # feature channels


# ----------------------------------------------------------------------------------------- CNN encoder
def _inorm(x):
    """nn.InstanceNorm2d defaults: eps 1e-5, no affine, no running stats (backbone.py:7,17-20)."""
    return F.instance_norm(x, eps=1e-5)


def _res_block(sd, p, x, stride):
    """backbone.py:5-36: conv3x3(stride) - IN - ReLU - conv3x3 - IN - ReLU; 1x1 strided conv + IN shortcut when the
    shape changes; ReLU(x + y)."""
    y = F.relu(_inorm(F.conv2d(x, sd[p + "conv1.weight"], None, stride=stride, padding=1)))
    y = F.relu(_inorm(F.conv2d(y, sd[p + "conv2.weight"], None, stride=1, padding=1)))
    if (p + "downsample.0.weight") in sd:
        x = _inorm(F.conv2d(x, sd[p + "downsample.0.weight"], sd[p + "downsample.0.bias"], stride=stride))
    return F.relu(x + y)


def encoder(sd, x, p="backbone."):
    """CNNEncoder.forward, num_output_scales=2 (backbone.py:39-117): 7x7 s2 conv, three stages (strides 1, 2, 1),
    1x1 conv, then the shared-weight trident conv at strides (1, 2) -> [1/4-res, 1/8-res] features."""
    x = F.relu(_inorm(F.conv2d(x, sd[p + "conv1.weight"], None, stride=2, padding=3)))
    for name, stride in (("layer1", 1), ("layer2", 2), ("layer3", 1)):
        x = _res_block(sd, p + name + ".0.", x, stride)
        x = _res_block(sd, p + name + ".1.", x, 1)
    x = F.conv2d(x, sd[p + "conv2.weight"], sd[p + "conv2.bias"])
    w = sd[p + "trident_conv.weight"]
    return [F.conv2d(x, w, None, stride=1, padding=1), F.conv2d(x, w, None, stride=2, padding=1)]


# ----------------------------------------------------------------------------------------- helpers (utils.py, position.py)
def split_feature(f, k, channel_last=False):
    """utils.py:5-31: [B,C,H,W] -> [B*k*k, C, H/k, W/k] (or the channel-last analogue)."""
    if channel_last:
        b, h, w, c = f.shape
        return f.view(b, k, h // k, k, w // k, c).permute(0, 1, 3, 2, 4, 5).reshape(b * k * k, h // k, w // k, c)
    b, c, h, w = f.shape
    return f.view(b, c, k, h // k, k, w // k).permute(0, 2, 4, 1, 3, 5).reshape(b * k * k, c, h // k, w // k)


def merge_splits(s, k, channel_last=False):
    """utils.py:34-54: inverse of split_feature."""
    if channel_last:
        b, h, w, c = s.shape
        nb = b // k // k
        return s.view(nb, k, k, h, w, c).permute(0, 1, 3, 2, 4, 5).contiguous().view(nb, k * h, k * w, c)
    b, c, h, w = s.shape
    nb = b // k // k
    return s.view(nb, k, k, c, h, w).permute(0, 3, 1, 4, 2, 5).contiguous().view(nb, c, k * h, k * w)


def position_embedding_sine(x, num_pos_feats=64, temperature=10000):
    """position.py:13-54 with normalize=True, scale=2*pi."""
    b, _, h, w = x.shape
    ones = torch.ones((b, h, w), dtype=x.dtype)
    y_embed, x_embed = ones.cumsum(1), ones.cumsum(2)
    eps, scale = 1e-6, 2 * math.pi
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=x.dtype)
    dim_t = temperature ** (2 * (dim_t // 2) / num_pos_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


def feature_add_position(f0, f1, attn_splits):
    """utils.py:72-94: sine position per attention window."""
    if attn_splits > 1:
        s0, s1 = split_feature(f0, attn_splits), split_feature(f1, attn_splits)
        pos = position_embedding_sine(s0, C // 2)
        return merge_splits(s0 + pos, attn_splits), merge_splits(s1 + pos, attn_splits)
    pos = position_embedding_sine(f0, C // 2)
    return f0 + pos, f1 + pos


def coords_grid(b, h, w, dtype=torch.float32):
    """geometry.py:5-24: [B,2,H,W] pixel coordinates (x, y)."""
    y, x = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    return torch.stack([x, y], 0)[None].repeat(b, 1, 1, 1).to(dtype)


def flow_warp(feature, flow):
    """geometry.py:53-84: grid_sample(bilinear, zeros, align_corners=True) at coords + flow, normalised 2c/(size-1)-1."""
    b, _, h, w = feature.shape
    g = coords_grid(b, h, w, flow.dtype) + flow
    xg = 2 * g[:, 0] / (w - 1) - 1
    yg = 2 * g[:, 1] / (h - 1) - 1
    return F.grid_sample(feature, torch.stack([xg, yg], -1), mode="bilinear", padding_mode="zeros", align_corners=True)


# ----------------------------------------------------------------------------------------- transformer
def shift_window_mask(h, w, wh, ww, sh, sw):
    """transformer.py:19-43: -100 between tokens of different shifted-window regions."""
    img = torch.zeros((1, h, w, 1))
    cnt = 0
    for hs in (slice(0, -wh), slice(-wh, -sh), slice(-sh, None)):
        for ws in (slice(0, -ww), slice(-ww, -sw), slice(-sw, None)):
            img[:, hs, ws, :] = cnt
            cnt += 1
    mw = split_feature(img, w // ww, channel_last=True).view(-1, wh * ww)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, float(-100.0)).masked_fill(m == 0, float(0.0))


def window_attention(q, k, v, num_splits, with_shift, h, w, mask):
    """transformer.py:46-105: single-head attention inside (optionally shifted) windows."""
    b, _, c = q.shape
    bn = b * num_splits * num_splits
    wh, ww = h // num_splits, w // num_splits
    q, k, v = q.view(b, h, w, c), k.view(b, h, w, c), v.view(b, h, w, c)
    if with_shift:
        sh, sw = wh // 2, ww // 2
        q, k, v = [torch.roll(t, shifts=(-sh, -sw), dims=(1, 2)) for t in (q, k, v)]
    q, k, v = [split_feature(t, num_splits, channel_last=True) for t in (q, k, v)]
    scores = torch.matmul(q.view(bn, -1, c), k.view(bn, -1, c).permute(0, 2, 1)) / (c ** 0.5)
    if with_shift:
        scores = scores + mask.repeat(b, 1, 1)
    out = torch.matmul(torch.softmax(scores, dim=-1), v.view(bn, -1, c))
    out = merge_splits(out.view(bn, wh, ww, c), num_splits, channel_last=True)
    if with_shift:
        out = torch.roll(out, shifts=(sh, sw), dims=(1, 2))
    return out.view(b, -1, c)


def transformer_layer(sd, p, source, target, h, w, mask, num_splits, with_shift, ffn):
    """TransformerLayer.forward (transformer.py:108-185)."""
    q = F.linear(source, sd[p + "q_proj.weight"])
    k = F.linear(target, sd[p + "k_proj.weight"])
    v = F.linear(target, sd[p + "v_proj.weight"])
    if num_splits > 1:
        msg = window_attention(q, k, v, num_splits, with_shift, h, w, mask)
    else:
        msg = torch.matmul(torch.softmax(torch.matmul(q, k.permute(0, 2, 1)) / (q.size(2) ** .5), dim=2), v)
    msg = F.linear(msg, sd[p + "merge.weight"])
    msg = F.layer_norm(msg, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    if ffn:
        msg = F.linear(F.gelu(F.linear(torch.cat([source, msg], dim=-1), sd[p + "mlp.0.weight"])), sd[p + "mlp.2.weight"])
        msg = F.layer_norm(msg, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    return source + msg


def feature_transformer(sd, f0, f1, num_splits, p="transformer."):
    """FeatureTransformer.forward (transformer.py:244-322): 6 x (self-attn, cross-attn + FFN), both directions batched."""
    b, c, h, w = f0.shape
    f0 = f0.flatten(-2).permute(0, 2, 1)
    f1 = f1.flatten(-2).permute(0, 2, 1)
    mask = None
    if num_splits > 1:
        wh, ww = h // num_splits, w // num_splits
        mask = shift_window_mask(h, w, wh, ww, wh // 2, ww // 2)
    c0, c1 = torch.cat((f0, f1), 0), torch.cat((f1, f0), 0)
    for i in range(6):
        lp = f"{p}layers.{i}."
        shift = (i % 2 == 1)
        c0 = transformer_layer(sd, lp + "self_attn.", c0, c0, h, w, mask, num_splits, shift, ffn=False)
        c0 = transformer_layer(sd, lp + "cross_attn_ffn.", c0, c1, h, w, mask, num_splits, shift, ffn=True)
        c1 = torch.cat(c0.chunk(2, 0)[::-1], 0)
    f0, f1 = c0.chunk(2, 0)
    f0 = f0.view(b, h, w, c).permute(0, 3, 1, 2).contiguous()
    f1 = f1.view(b, h, w, c).permute(0, 3, 1, 2).contiguous()
    return f0, f1


# ----------------------------------------------------------------------------------------- matching
def global_correlation_softmax(f0, f1):
    """matching.py:7-38: softmax over all target positions, flow = expected coordinate - own coordinate."""
    b, c, h, w = f0.shape
    corr = torch.matmul(f0.view(b, c, -1).permute(0, 2, 1), f1.view(b, c, -1)).view(b, h, w, h, w) / (c ** 0.5)
    init = coords_grid(b, h, w, f0.dtype)
    grid = init.view(b, 2, -1).permute(0, 2, 1)
    prob = F.softmax(corr.view(b, h * w, h * w), dim=-1)
    return torch.matmul(prob, grid).view(b, h, w, 2).permute(0, 3, 1, 2) - init


def local_correlation_softmax(f0, f1, r):
    """matching.py:41-89: softmax over a (2r+1)^2 window sampled with grid_sample(zeros), out-of-image taps -> -1e4."""
    b, c, h, w = f0.shape
    init = coords_grid(b, h, w, f0.dtype)
    coords = init.view(b, 2, -1).permute(0, 2, 1)
    n = 2 * r + 1
    gx, gy = torch.meshgrid([torch.linspace(-r, r, n), torch.linspace(-r, r, n)], indexing="ij")
    win = torch.stack((gx, gy), -1).transpose(0, 1).to(f0.dtype).reshape(-1, 2).repeat(b, 1, 1, 1)
    sc = coords.unsqueeze(-2) + win
    valid = (sc[..., 0] >= 0) & (sc[..., 0] < w) & (sc[..., 1] >= 0) & (sc[..., 1] < h)
    cc = torch.tensor([(w - 1) / 2., (h - 1) / 2.], dtype=sc.dtype)
    wf = F.grid_sample(f1.contiguous(), ((sc - cc) / cc).contiguous(), padding_mode="zeros", align_corners=True).permute(0, 2, 1, 3)
    corr = torch.matmul(f0.permute(0, 2, 3, 1).view(b, h * w, 1, c), wf).view(b, h * w, -1) / (c ** 0.5)
    corr[~valid] = -1e4
    prob = F.softmax(corr, -1)
    return torch.matmul(prob.unsqueeze(-2), sc).squeeze(-2).view(b, h, w, 2).permute(0, 3, 1, 2) - init


def flow_attention(sd, f0, flow, local, radius, p="feature_flow_attn."):
    """FeatureFlowAttention (transformer.py:325-409): flow propagated by feature self-similarity.
    NB (kept as written): key = k_proj(q_proj(feature)) in the global form, k_proj(feature) in the local form."""
    b, c, h, w = f0.shape
    tok = f0.view(b, c, h * w).permute(0, 2, 1)
    if not local:
        q = F.linear(tok, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"])
        k = F.linear(q, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"])
        v = flow.view(b, flow.size(1), h * w).permute(0, 2, 1)
        prob = torch.softmax(torch.matmul(q, k.permute(0, 2, 1)) / (c ** 0.5), dim=-1)
        return torch.matmul(prob, v).view(b, h, w, v.size(-1)).permute(0, 3, 1, 2)
    q = F.linear(tok, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"]).reshape(b * h * w, 1, c)
    ks = 2 * radius + 1
    kp = F.linear(tok, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"]).permute(0, 2, 1).reshape(b, c, h, w)
    kw = F.unfold(kp, kernel_size=ks, padding=radius).view(b, c, ks ** 2, h, w).permute(0, 3, 4, 1, 2).reshape(b * h * w, c, ks ** 2)
    fw = F.unfold(flow, kernel_size=ks, padding=radius).view(b, 2, ks ** 2, h, w).permute(0, 3, 4, 2, 1).reshape(b * h * w, ks ** 2, 2)
    prob = torch.softmax(torch.matmul(q, kw) / (c ** 0.5), dim=-1)
    return torch.matmul(prob, fw).view(b, h, w, 2).permute(0, 3, 1, 2).contiguous()


def upsample_flow(sd, flow, feature, factor=4, p="upsampler."):
    """GMFlow.upsample_flow, learned convex upsampling (gmflow.py:67-90)."""
    m = F.conv2d(torch.cat((flow, feature), 1), sd[p + "0.weight"], sd[p + "0.bias"], padding=1)
    m = F.conv2d(F.relu(m), sd[p + "2.weight"], sd[p + "2.bias"])
    b, fc, h, w = flow.shape
    m = torch.softmax(m.view(b, 1, 9, factor, factor, h, w), dim=2)
    up = F.unfold(factor * flow, [3, 3], padding=1).view(b, fc, 9, 1, 1, h, w)
    up = torch.sum(m * up, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(b, fc, factor * h, factor * w)


_MEAN = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
_STD = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)


def gmflow(sd, img0, img1):
    """GMFlow.forward (gmflow.py:92-185), inference path, unidirectional."""
    img0, img1 = (img0 - _MEAN) / _STD, (img1 - _MEAN) / _STD
    feats = encoder(sd, torch.cat((img0, img1), 0))[::-1]  # low -> high resolution
    f0s = [f.chunk(2, 0)[0] for f in feats]
    f1s = [f.chunk(2, 0)[1] for f in feats]
    flow = None
    for idx, (splits, corr_r, prop_r) in enumerate(((2, -1, -1), (8, 4, 1))):
        f0, f1 = f0s[idx], f1s[idx]
        if idx > 0:
            flow = F.interpolate(flow, scale_factor=2, mode="bilinear", align_corners=True) * 2
        if flow is not None:
            f1 = flow_warp(f1, flow)
        f0, f1 = feature_add_position(f0, f1, splits)
        f0, f1 = feature_transformer(sd, f0, f1, splits)
        pred = global_correlation_softmax(f0, f1) if corr_r == -1 else local_correlation_softmax(f0, f1, corr_r)
        flow = flow + pred if flow is not None else pred
        flow = flow_attention(sd, f0, flow, local=prop_r > 0, radius=prop_r)
        if idx == 1:
            return upsample_flow(sd, flow, f0)
