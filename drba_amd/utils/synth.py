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


def gmflow_shapes():
    """Ordered {key: shape} of GMFlow().state_dict() (124 tensors, 4 716 720 params; models/gmflow/gmflow.py:13-44)."""
    sh = {"backbone.conv1.weight": (64, 3, 7, 7)}
    cin = 64
    for name, c in (("layer1", 64), ("layer2", 96), ("layer3", 128)):
        for b in (0, 1):
            ci = cin if b == 0 else c
            sh[f"backbone.{name}.{b}.conv1.weight"] = (c, ci, 3, 3)
            sh[f"backbone.{name}.{b}.conv2.weight"] = (c, c, 3, 3)
            if b == 0 and ci != c:
                sh[f"backbone.{name}.{b}.downsample.0.weight"] = (c, ci, 1, 1)
                sh[f"backbone.{name}.{b}.downsample.0.bias"] = (c,)
        cin = c
    sh["backbone.conv2.weight"] = (128, 128, 1, 1)
    sh["backbone.conv2.bias"] = (128,)
    sh["backbone.trident_conv.weight"] = (128, 128, 3, 3)
    for i in range(6):
        for part, ffn in (("self_attn", False), ("cross_attn_ffn", True)):
            p = f"transformer.layers.{i}.{part}."
            for n in ("q_proj", "k_proj", "v_proj", "merge"):
                sh[p + n + ".weight"] = (128, 128)
            sh[p + "norm1.weight"] = (128,)
            sh[p + "norm1.bias"] = (128,)
            if ffn:
                sh[p + "mlp.0.weight"] = (1024, 256)
                sh[p + "mlp.2.weight"] = (128, 1024)
                sh[p + "norm2.weight"] = (128,)
                sh[p + "norm2.bias"] = (128,)
    for n in ("q_proj", "k_proj"):
        sh[f"feature_flow_attn.{n}.weight"] = (128, 128)
        sh[f"feature_flow_attn.{n}.bias"] = (128,)
    sh["upsampler.0.weight"] = (256, 130, 3, 3)
    sh["upsampler.0.bias"] = (256,)
    sh["upsampler.2.weight"] = (144, 256, 1, 1)
    sh["upsampler.2.bias"] = (144,)
    return sh


def metricnet_shapes():
    """MetricNet (model_gmfss_union/MetricNet.py:23-43): 14 tensors, 120 070 params."""
    sh = {"metric_in.weight": (64, 14, 3, 3), "metric_in.bias": (64,)}
    for k in (1, 2, 3):
        sh[f"metric_net{k}.0.weight"] = (1,)
        sh[f"metric_net{k}.1.weight"] = (64, 64, 3, 3)
        sh[f"metric_net{k}.1.bias"] = (64,)
    sh["metric_out.0.weight"] = (1,)
    sh["metric_out.1.weight"] = (2, 64, 3, 3)
    sh["metric_out.1.bias"] = (2,)
    return sh


def featurenet_shapes():
    """FeatureNet (FeatureNet.py:9-27): 18 tensors, 813 510 params."""
    sh = {}
    cin = 3
    for b, c in ((1, 64), (2, 128), (3, 192)):
        sh[f"block{b}.0.weight"] = (1,)
        sh[f"block{b}.1.weight"] = (c, cin, 3, 3)
        sh[f"block{b}.1.bias"] = (c,)
        sh[f"block{b}.2.weight"] = (1,)
        sh[f"block{b}.3.weight"] = (c, c, 3, 3)
        sh[f"block{b}.3.bias"] = (c,)
        cin = c
    return sh


def gridnet_shapes(in_channels=9, head0="head0"):
    """GridNet(in, 128, 256, 384, 3) (FusionNet.py:55-104); union: in=9, key 'head0'; gmfss: in=12, key 'head'."""
    sh = {}

    def two(name, ci, co, deconv=False):
        p = name + "."
        sh[p + "0.weight"] = (1,)
        sh[p + "1.weight"] = (ci, co, 4, 4) if deconv else (co, ci, 3, 3)
        sh[p + "1.bias"] = (co,)
        sh[p + "2.weight"] = (1,)
        sh[p + "3.weight"] = (co, co, 3, 3)
        sh[p + "3.bias"] = (co,)

    two("residual_model_" + head0, in_channels, 64)
    two("residual_model_head1", 128, 64)
    two("residual_model_head2", 256, 128)
    two("residual_model_head3", 384, 192)
    for n in ("01", "04", "05"):
        two("residual_model_" + n, 64, 64)
    sh["residual_model_tail.conv_before_upsample.0.weight"] = (64, 64, 3, 3)
    sh["residual_model_tail.conv_before_upsample.0.bias"] = (64,)
    sh["residual_model_tail.conv_before_upsample.1.weight"] = (1,)
    sh["residual_model_tail.upsample.0.weight"] = (256, 64, 3, 3)
    sh["residual_model_tail.upsample.0.bias"] = (256,)
    sh["residual_model_tail.conv_last.weight"] = (3, 64, 3, 3)
    sh["residual_model_tail.conv_last.bias"] = (3,)
    for n in ("11", "14", "15"):
        two("residual_model_" + n, 128, 128)
    for n in ("21", "24", "25"):
        two("residual_model_" + n, 192, 192)
    two("downsample_model_10", 64, 128)
    two("downsample_model_20", 128, 192)
    two("downsample_model_11", 64, 128)
    two("downsample_model_21", 128, 192)
    two("upsample_model_04", 128, 64, deconv=True)
    two("upsample_model_14", 192, 128, deconv=True)
    two("upsample_model_05", 128, 64, deconv=True)
    two("upsample_model_15", 192, 128, deconv=True)
    return sh


def seeded_state_dict(shapes, seed=0, tag="", damp_transformer=True):
    """Generic seeded weights: conv/linear ~ N(0, (g/sqrt(fan_in))^2), PReLU slopes ~ U(0.1, 0.4), norm weights ~ 1, biases small.
    damp_transformer=False keeps the transformer's LayerNorm gains at ~1 (the ill-conditioned variant, see below)."""
    sd = {}
    for key, shape in shapes.items():
        g = _gen(tag + key, seed)
        if len(shape) == 1 and shape[0] == 1:  # nn.PReLU() slope
            t = torch.rand(shape, generator=g) * 0.3 + 0.1
        elif key.endswith("bias"):
            t = torch.randn(shape, generator=g) * (0.002 if key == "backbone.conv2.bias" and damp_transformer else 0.02)
        elif len(shape) == 1:  # LayerNorm weight
            t = 1.0 + torch.randn(shape, generator=g) * 0.05
            if key.startswith("transformer.") and damp_transformer:
                # damped attention / FFN messages keep the matching features close to the CNN features, so the
                # correlation softmax is peaked as in a trained network; with unit gains the random transformer
                # makes it diffuse and a 1-ulp input change moves the reference's own flow by > 1e-2
                t = t * 0.1
        else:
            if len(shape) == 4 and shape[2] == 4:  # ConvTranspose2d [Cin, Cout, 4, 4]: 2x2 taps per output
                fan_in = shape[0] * 4
            elif len(shape) == 4:
                fan_in = shape[1] * shape[2] * shape[3]
            else:
                fan_in = shape[1]
            gain = 0.7 if ("residual_model" in key or "metric_net" in key) else 1.0
            if key == "backbone.conv2.weight" and damp_transformer:
                # GMFlow's matching features = CNN features + a weight-free sinusoidal position embedding.  Random CNN features
                # of low-texture synthetic frames match globally at random (flows of hundreds of pixels whose arg-max flips under
                # a 1-ulp input change: the oracle's own 1152x1920 frame moved by 4.5e-2, rounds 1-4); scaled down, the position
                # term dominates, matches stay near the identity as a trained network's do on small motion (flows <= 20 px) and the
                # oracle's self-sensitivity at 1152x1920 is 6e-5 on frames, 1.3e-4 on flows (tools/exp/union_floor_probe.py) --
                # the end-to-end 1e-3 bar is falsifiable at every size
                gain = 0.1
            t = torch.randn(shape, generator=g) * (gain / fan_in ** 0.5)
        sd[key] = t.contiguous()
    return sd


def gmfss_union_state_dicts(seed=0):
    """flownet (GMFlow), metric, feat, fusion (GridNet in=9) and the auxiliary RIFE for GMFSS_UNION."""
    return {"flownet": seeded_state_dict(gmflow_shapes(), seed, "gmflow."),
            "metric": seeded_state_dict(metricnet_shapes(), seed, "metric."),
            "feat": seeded_state_dict(featurenet_shapes(), seed, "feat."),
            "fusion": seeded_state_dict(gridnet_shapes(9, "head0"), seed, "grid."),
            "rife": ifnet_state_dict(seed + 1)}


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
