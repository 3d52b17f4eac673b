"""GMFlow on the MI355X (reference models/gmflow/{gmflow,backbone,trident_conv,transformer,matching,geometry,utils,
position}.py, same state-dict keys; configuration used by DRBA: 2 scales, swin attention, 6 layers, 1 head).

Division of labour: the 3x3 convolutions run on the MFMA implicit-GEMM kernels; the 7x7 / 1x1 convolutions, the norms,
local-window correlation / propagation, convex upsampling and warps are hand-written HIP kernels
(drba_amd/csrc/gmflow.hip); every linear layer is drba_linear_split (fp32 operands as two fp16 or three bf16 terms on the matrix cores, fp32-level error), the
window attention is one fused kernel (window_attn.hip) and the global correlation / propagation softmaxes are a
flash-style kernel that never forms the L x L score matrix (global_corr.hip).  torch does views, cat and nothing else:
no arithmetic and no vendor BLAS.
"""
import math

import torch

from drba_amd import ops as _ops

C = 128
_MEAN, _STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def _linear(x, w):
    return w(x)  # _ops.LinearSplit


class _ResBlock:
    """backbone.py:5-36."""

    def __init__(self, sd, p, stride, device):
        self.c1 = _ops.Conv3x3(sd[p + "conv1.weight"], None, stride=stride, act=None, device=device)
        self.c2 = _ops.Conv3x3(sd[p + "conv2.weight"], None, stride=1, act=None, device=device)
        self.stride = stride
        self.down = None
        if (p + "downsample.0.weight") in sd:
            self.down = (sd[p + "downsample.0.weight"].float().to(device).contiguous(),
                         sd[p + "downsample.0.bias"].float().to(device).contiguous())

    def __call__(self, x):
        y = _ops.instance_norm(self.c1(x), relu=True)
        y = _ops.instance_norm(self.c2(y), relu=True)
        if self.down is not None:
            x = _ops.instance_norm(_ops.conv_direct(x, self.down[0], self.down[1], self.stride, 0), relu=False)
        return _ops.add_act(x, y, relu=True)


class GMFlow:
    def __init__(self, sd=None, device=None):
        self.device = device
        if sd is not None:
            self.load_state_dict(sd)

    def to(self, device):
        self.device = device
        return self

    def eval(self):
        return self

    def load_state_dict(self, sd, strict=True):
        dev = self.device if self.device is not None else _ops.default_device()
        self.device = dev
        g = lambda k: sd[k].detach().float().to(dev).contiguous()  # noqa: E731
        self.conv1_w = g("backbone.conv1.weight")
        self.blocks = []
        for name, stride in (("layer1", 1), ("layer2", 2), ("layer3", 1)):
            self.blocks.append(_ResBlock(sd, f"backbone.{name}.0.", stride, dev))
            self.blocks.append(_ResBlock(sd, f"backbone.{name}.1.", 1, dev))
        self.conv2 = (g("backbone.conv2.weight"), g("backbone.conv2.bias"))
        self.trident = [_ops.Conv3x3(sd["backbone.trident_conv.weight"], None, stride=s, act=None, device=dev) for s in (1, 2)]
        self.layers = []
        for i in range(6):
            lay = {}
            for part, ffn in (("self_attn", False), ("cross_attn_ffn", True)):
                p = f"transformer.layers.{i}.{part}."
                d = {n: g(p + n + ".weight") for n in ("q_proj", "k_proj", "v_proj", "merge")}
                # fused projections: self-attention projects one tensor three times, cross-attention its target twice
                d["qkv_proj"] = torch.cat((d["q_proj"], d["k_proj"], d["v_proj"]), 0).contiguous()
                d["kv_proj"] = torch.cat((d["k_proj"], d["v_proj"]), 0).contiguous()
                for n in ("qkv_proj", "kv_proj", "q_proj", "merge"):
                    d[n] = _ops.LinearSplit(d[n], device=dev)
                del d["k_proj"], d["v_proj"]
                d["n1w"], d["n1b"] = g(p + "norm1.weight"), g(p + "norm1.bias")
                if ffn:
                    d["mlp0"] = _ops.LinearSplit(g(p + "mlp.0.weight"), gelu=True, device=dev)  # GELU in the epilogue
                    d["mlp2"] = _ops.LinearSplit(g(p + "mlp.2.weight"), device=dev)
                    d["n2w"], d["n2b"] = g(p + "norm2.weight"), g(p + "norm2.bias")
                lay[part] = d
            self.layers.append(lay)
        self.ffa_q = _ops.LinearSplit(g("feature_flow_attn.q_proj.weight"), g("feature_flow_attn.q_proj.bias"), device=dev)
        self.ffa_k = _ops.LinearSplit(g("feature_flow_attn.k_proj.weight"), g("feature_flow_attn.k_proj.bias"), device=dev)
        self.up0 = _ops.Conv3x3(sd["upsampler.0.weight"], sd["upsampler.0.bias"], stride=1, act="relu", device=dev)
        self.up2 = (g("upsampler.2.weight").view(144, 256, 1, 1), g("upsampler.2.bias"))
        self._pos, self._mask = {}, {}
        return self

    # ---------------------------------------------------------------- encoder (backbone.py:39-117)
    def encoder(self, x):
        x = _ops.instance_norm(_ops.conv_direct(x, self.conv1_w, None, 2, 3), relu=True)
        for b in self.blocks:
            x = b(x)
        x = _ops.conv_direct(x, self.conv2[0], self.conv2[1], 1, 0)
        return [self.trident[0](x), self.trident[1](x)]  # [1/4-res, 1/8-res]

    # ---------------------------------------------------------------- host-side tables
    def _position(self, b, h, w):
        """PositionEmbeddingSine (position.py:13-54), data independent: built once per size on the host."""
        key = (b, h, w)
        if key not in self._pos:
            npf, temp, scale, eps = C // 2, 10000, 2 * math.pi, 1e-6
            ones = torch.ones((b, h, w), dtype=torch.float32)
            ye, xe = ones.cumsum(1), ones.cumsum(2)
            ye = ye / (ye[:, -1:, :] + eps) * scale
            xe = xe / (xe[:, :, -1:] + eps) * scale
            dim_t = torch.arange(npf, dtype=torch.float32)
            dim_t = temp ** (2 * (dim_t // 2) / npf)
            px, py = xe[:, :, :, None] / dim_t, ye[:, :, :, None] / dim_t
            px = torch.stack((px[:, :, :, 0::2].sin(), px[:, :, :, 1::2].cos()), dim=4).flatten(3)
            py = torch.stack((py[:, :, :, 0::2].sin(), py[:, :, :, 1::2].cos()), dim=4).flatten(3)
            self._pos[key] = torch.cat((py, px), dim=3).permute(0, 3, 1, 2).contiguous().to(self.device)
        return self._pos[key]

    def _shift_mask(self, h, w, k):
        """generate_shift_window_attn_mask (transformer.py:19-43): host table, [k*k, L, L]."""
        key = (h, w, k)
        if key not in self._mask:
            wh, ww = h // k, w // k
            sh, sw = wh // 2, ww // 2
            img = torch.zeros((1, h, w, 1))
            cnt = 0
            for hs in (slice(0, -wh), slice(-wh, -sh), slice(-sh, None)):
                for ws in (slice(0, -ww), slice(-ww, -sw), slice(-sw, None)):
                    img[:, hs, ws, :] = cnt
                    cnt += 1
            mw = _split(img, k, True).view(-1, wh * ww)
            m = mw.unsqueeze(1) - mw.unsqueeze(2)
            m = m.masked_fill(m != 0, float(-100.0)).masked_fill(m == 0, float(0.0))
            self._mask[key] = m.contiguous().to(self.device)
        return self._mask[key]

    # ---------------------------------------------------------------- transformer (transformer.py)
    def _attention(self, q, k, v, h, w, splits, shift):
        b, _, c = q.shape
        k_ = max(splits, 1)
        shifted = bool(shift) and splits > 1
        if shifted and (h // k_ < 2 or w // k_ < 2):
            return self._attention_degenerate(q, k, v, h, w, splits)
        return _ops.window_attention(q, k, v, h, w, k_, shifted, c ** 0.5)

    def _attention_degenerate(self, q, k, v, h, w, splits):
        """Shifted windows one pixel wide or high (feature maps below 16 px, i.e. frames below 128 px): the reference's
        mask table is built with slice(-0, None) there (transformer.py:28-36), which the fused kernel's coordinate rule
        does not reproduce, so the reference's steps are replayed with the host-built table: roll, split, plain fp32
        products (drba_bmm), masked softmax kernel, merge, roll back."""
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()  # column slices of a fused projection output
        b, _, c = q.shape
        bn, wh, ww = b * splits * splits, h // splits, w // splits
        sh, sw = wh // 2, ww // 2
        q, k, v = [torch.roll(t.view(b, h, w, c), shifts=(-sh, -sw), dims=(1, 2)) for t in (q, k, v)]
        q, k, v = [_split(t, splits, True).reshape(bn, -1, c).contiguous() for t in (q, k, v)]
        scores = _ops.bmm(q, k, trans_b=True)
        _ops.softmax_rows_(scores, c ** 0.5, self._shift_mask(h, w, splits))
        out = _merge(_ops.bmm(scores, v, trans_b=False).view(bn, wh, ww, c), splits, True)
        return torch.roll(out, shifts=(sh, sw), dims=(1, 2)).reshape(b, -1, c)

    def _layer(self, d, source, target, h, w, splits, shift, ffn):
        if source is target:  # one [tokens, 3C] GEMM; the attention kernel reads column slices
            qkv = _linear(source, d["qkv_proj"])
            q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        else:
            q, kv = _linear(source, d["q_proj"]), _linear(target, d["kv_proj"])
            k, v = kv[..., :C], kv[..., C:]
        att = self._attention(q, k, v, h, w, splits, shift)
        if not ffn:  # LayerNorm (+ residual) in the GEMM's epilogue
            return d["merge"].layernorm(att, d["n1w"], d["n1b"], residual=source)
        msg = d["merge"].layernorm(att, d["n1w"], d["n1b"])
        hid = d["mlp0"].cat(source, msg)  # reads cat(source, msg) in place; GELU in the epilogue
        return d["mlp2"].layernorm(hid, d["n2w"], d["n2b"], residual=source)

    def transformer(self, f0, f1, splits):
        b, c, h, w = f0.shape
        t0 = f0.flatten(-2).permute(0, 2, 1)
        t1 = f1.flatten(-2).permute(0, 2, 1)
        c0, c1 = torch.cat((t0, t1), 0).contiguous(), torch.cat((t1, t0), 0).contiguous()
        for i, lay in enumerate(self.layers):
            shift = (i % 2 == 1) and splits > 1
            c0 = self._layer(lay["self_attn"], c0, c0, h, w, splits, shift, False)
            c0 = self._layer(lay["cross_attn_ffn"], c0, c1, h, w, splits, shift, True)
            c1 = torch.cat(c0.chunk(2, 0)[::-1], 0).contiguous()
        t0, t1 = c0.chunk(2, 0)
        self._tokens = (t0.reshape(h * w, c), t1.reshape(h * w, c))  # token-major views of the outputs (global matching reads them)
        return (t0.reshape(b, h, w, c).permute(0, 3, 1, 2).contiguous(),
                t1.reshape(b, h, w, c).permute(0, 3, 1, 2).contiguous())

    def _add_position(self, f0, f1, splits):
        if splits > 1:
            s0, s1 = _split(f0, splits, False), _split(f1, splits, False)
            pos = self._position(*[s0.shape[0], s0.shape[2], s0.shape[3]])
            return (_merge(_ops.add_act(s0.contiguous(), pos), splits, False),
                    _merge(_ops.add_act(s1.contiguous(), pos), splits, False))
        pos = self._position(f0.shape[0], f0.shape[2], f0.shape[3])
        return _ops.add_act(f0, pos), _ops.add_act(f1, pos)

    def _propagate(self, f0, flow, local, radius, tok=None):
        """FeatureFlowAttention (transformer.py:325-409).  `tok`: f0 token-major ([h*w, C]) if the caller has it."""
        b, c, h, w = f0.shape
        if tok is None:
            tok = f0.view(c, h * w).t().contiguous()
        q = self.ffa_q(tok)
        if not local:
            k = self.ffa_k(q)  # key from the projected query, as written (transformer.py:361-364)
            return _ops.global_expect2(q, k, flow.view(2, h * w), w, c ** 0.5).view(1, 2, h, w)
        k = self.ffa_k(tok)
        return _ops.local_attn_flow(q, k, flow, radius)

    # ---------------------------------------------------------------- forward (gmflow.py:92-185)
    def _match(self, f0, f1, flow, corr_r, prop_r, tok0=None, tok1=None):
        """Correlation softmax (global or local) + flow propagation of one direction at one scale.  tok0 / tok1: the
        token-major ([h*w, C]) form of f0 / f1 when the caller has it (the transformer's own layout)."""
        _, c, h, w = f0.shape
        if corr_r == -1:
            if tok0 is None:
                tok0, tok1 = f0.view(c, h * w).t().contiguous(), f1.view(c, h * w).t().contiguous()
            pred = _ops.global_expect2(tok0, tok1, None, w, c ** 0.5).view(1, 2, h, w)
        else:
            pred = _ops.local_corr_flow(f0, f1, corr_r)
        flow = pred if flow is None else _ops.add_act(flow, pred)
        return self._propagate(f0, flow, local=prop_r > 0, radius=prop_r, tok=tok0)

    def _upsample(self, flow, f0):
        """learned convex upsampling x4 (gmflow.py:67-90)"""
        m = self.up0(torch.cat((flow, f0), 1))
        m = _ops.conv_direct(m, self.up2[0], self.up2[1], 1, 0)  # 1x1 conv 256 -> 144 (9 x 16 convex weights)
        return _ops.convex_upsample(m, flow, 4)

    def _refine(self, fa, fb, flow, splits, corr_r, prop_r):
        """One finer scale of one direction: warp the other frame's features by the upsampled flow, transformer, match."""
        _, _, h, w = fa.shape
        flow = _ops.resize_bilinear_ac(flow, (h, w), 2.0)  # x2 bilinear (align_corners=True), * 2
        fb = _ops.flow_warp(fb, flow)
        fa, fb = self._add_position(fa, fb, splits)
        fa, fb = self.transformer(fa, fb, splits)
        ta, tb = self._tokens
        return self._match(fa, fb, flow, corr_r, prop_r, ta, tb), fa

    def __call__(self, img0, img1, attn_splits_list=(2, 8), corr_radius_list=(-1, 4), prop_radius_list=(-1, 1), **kw):
        x = _ops.channel_normalize3(torch.cat((img0, img1), 0).contiguous(), _MEAN, _STD)
        feats = self.encoder(x)[::-1]  # low -> high resolution
        flow = None
        for idx, (splits, corr_r, prop_r) in enumerate(zip(attn_splits_list, corr_radius_list, prop_radius_list)):
            f0, f1 = feats[idx][0:1].contiguous(), feats[idx][1:2].contiguous()
            if idx > 0:
                flow, f0 = self._refine(f0, f1, flow, splits, corr_r, prop_r)
            else:
                f0, f1 = self._add_position(f0, f1, splits)
                f0, f1 = self.transformer(f0, f1, splits)
                flow = self._match(f0, f1, None, corr_r, prop_r, *self._tokens)
        return self._upsample(flow, f0)

    def encode_frame(self, img):
        """CNN encoder outputs [1/8-res, 1/4-res] of ONE frame (normalisation included).  InstanceNorm keeps the samples
        of a batch independent (backbone.py), so a frame's features do not depend on its partner: Model.reuse caches them
        per frame instead of re-encoding the frame for both pairs it belongs to."""
        return self.encoder(_ops.channel_normalize3(img.contiguous(), _MEAN, _STD))[::-1]

    def bidirectional(self, img0, img1, attn_splits_list=(2, 8), corr_radius_list=(-1, 4), prop_radius_list=(-1, 1),
                      feats=None):
        """(self(img0, img1), self(img1, img0)) -- what Model.reuse needs (GMFSS.py:64-65) -- sharing what the two calls
        have in common: the CNN encoder of both frames and the coarsest transformer pass.  The transformer treats its
        two inputs symmetrically (transformer.py:236-302: both orders are concatenated along the batch), so the swapped
        call computes the same two feature maps in the other order; only from the first flow-dependent warp on do the
        directions differ.  Same values as two separate calls, about a third less work."""
        assert len(attn_splits_list) == 2, "two scales (the DRBA configuration)"
        if feats is None:
            x = _ops.channel_normalize3(torch.cat((img0, img1), 0).contiguous(), _MEAN, _STD)
            both = self.encoder(x)[::-1]
            c0, c1 = both[0][0:1].contiguous(), both[0][1:2].contiguous()
            h0, h1 = both[1][0:1].contiguous(), both[1][1:2].contiguous()
        else:  # (encode_frame(img0), encode_frame(img1))
            (c0, h0), (c1, h1) = feats
        t0, t1 = self._add_position(c0, c1, attn_splits_list[0])
        t0, t1 = self.transformer(t0, t1, attn_splits_list[0])
        k0, k1 = self._tokens
        flow_a = self._match(t0, t1, None, corr_radius_list[0], prop_radius_list[0], k0, k1)
        flow_b = self._match(t1, t0, None, corr_radius_list[0], prop_radius_list[0], k1, k0)
        flow_a, fa = self._refine(h0, h1, flow_a, attn_splits_list[1], corr_radius_list[1], prop_radius_list[1])
        flow_b, fb = self._refine(h1, h0, flow_b, attn_splits_list[1], corr_radius_list[1], prop_radius_list[1])
        return self._upsample(flow_a, fa), self._upsample(flow_b, fb)

    forward = __call__


def _split(f, k, channel_last):
    """utils.py:5-31 (data movement only)."""
    if channel_last:
        b, h, w, c = f.shape
        return f.reshape(b, k, h // k, k, w // k, c).permute(0, 1, 3, 2, 4, 5).reshape(b * k * k, h // k, w // k, c)
    b, c, h, w = f.shape
    return f.reshape(b, c, k, h // k, k, w // k).permute(0, 2, 4, 1, 3, 5).reshape(b * k * k, c, h // k, w // k)


def _merge(s, k, channel_last):
    """utils.py:34-54 (data movement only)."""
    if channel_last:
        b, h, w, c = s.shape
        nb = b // k // k
        return s.reshape(nb, k, k, h, w, c).permute(0, 1, 3, 2, 4, 5).contiguous().view(nb, k * h, k * w, c)
    b, c, h, w = s.shape
    nb = b // k // k
    return s.reshape(nb, k, k, c, h, w).permute(0, 3, 1, 4, 2, 5).contiguous().view(nb, c, k * h, k * w)
