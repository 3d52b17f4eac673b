"""RIFE v4.26-heavy IFNet on the HIP library.

Same module tree / state-dict keys as the reference network
(models/rife_426_heavy/IFNet_HDv3.py:28-47 Head, :50-59 ResConv, :62-96 IFBlock, :99-177 IFNet)
so reference checkpoints load unchanged, but no torch.nn compute: every layer is a call into
libdrba_hip.so.  What the reference does with cat / interpolate / grid_sample between the
convolutions is fused into three glue kernels (drba_amd/csrc/ifnet_glue.hip):

    stage input  = warp x4 + concat + 1/s bilinear downsample   (drba_ifblock_input)
    stage output = PixelShuffle (deconv epilogue) + x s upsample + flow accumulate (drba_ifblock_update);
                   mask/feat are re-derived from the low-res head output by the next stage's input kernel
    synthesis    = warp x2 + sigmoid blend                       (drba_warp_blend)
"""
import numpy as np
import torch

from drba_amd import _lib
from drba_amd import ops as _ops

BLOCK_C = (192, 128, 96, 64, 32)


class Head:
    """Context encoder (IFNet_HDv3.py:28-47): conv s2 -> conv -> conv (LeakyReLU 0.2 each) -> deconv 4x4 s2."""

    def __init__(self, sd, prefix, device):
        g = lambda k: sd[prefix + k]  # noqa: E731
        self.cnn0 = _ops.Conv3x3(g("cnn0.weight"), g("cnn0.bias"), stride=2, act=True, device=device)
        self.cnn1 = _ops.Conv3x3(g("cnn1.weight"), g("cnn1.bias"), stride=1, act=True, device=device)
        self.cnn2 = _ops.Conv3x3(g("cnn2.weight"), g("cnn2.bias"), stride=1, act=True, device=device)
        self.cnn3 = _ops.Deconv4x4(g("cnn3.weight"), g("cnn3.bias"), pixel_shuffle=False, device=device)
        self.chain = _ops.ConvChain([(self.cnn0, False), (self.cnn1, False), (self.cnn2, False), (self.cnn3, False)])

    def __call__(self, x, feat=False, planar=True):
        """planar=False (RIFE's own calls): the features in the pair-interleaved layout only (ops.head_fused), which is what
        every kernel of the pipeline reads; the default returns the reference's [1,16,H,W]."""
        if not feat:
            if _ops.HEAD_FUSED and x.is_cuda:
                f = _ops.head_fused(x, (self.cnn0, self.cnn1, self.cnn2, self.cnn3), self, planar=planar)  # one kernel
                if f is not None:
                    return f
            return self.chain(x)
        x0 = self.cnn0(x)
        x1 = self.cnn1(x0)
        x2 = self.cnn2(x1)
        x3 = self.cnn3(x2)
        return [x0, x1, x2, x3] if feat else x3  # NB: x0..x2 are post-activation, like the reference's in-place ReLU

    forward = __call__


class IFBlock:
    """IFNet_HDv3.py:62-96.  `core` is conv0 -> 8 x ResConv -> deconv(+PixelShuffle)."""

    def __init__(self, sd, prefix, device):
        g = lambda k: sd[prefix + k]  # noqa: E731
        self.conv0_0 = _ops.Conv3x3(g("conv0.0.0.weight"), g("conv0.0.0.bias"), stride=2, act=True, device=device)
        self.conv0_1 = _ops.Conv3x3(g("conv0.1.0.weight"), g("conv0.1.0.bias"), stride=2, act=True, device=device)
        self.convblock = [
            _ops.Conv3x3(g(f"convblock.{j}.conv.weight"), g(f"convblock.{j}.conv.bias"), stride=1, act=True,
                         beta=g(f"convblock.{j}.beta"), device=device) for j in range(8)]
        self.lastconv = _ops.Deconv4x4(g("lastconv.0.weight"), g("lastconv.0.bias"), pixel_shuffle=True, device=device)
        self.chain = _ops.ConvChain([(self.conv0_0, False), (self.conv0_1, False)] + [(rc, True) for rc in self.convblock]
                                    + [(self.lastconv, False)])
        # the core after conv0[0], for the stage whose input gather and first convolution are one kernel (ops.stage_conv0)
        self.chain_tail = _ops.ConvChain([(self.conv0_1, False)] + [(rc, True) for rc in self.convblock] + [(self.lastconv, False)])
        # the LAST stage's head output is only read for flow (4 channels) and mask (1): `feat`, its other 8 channels, feeds the
        # next stage and there is none (IFNet_HDv3.py:146-167) -- 20 of the transposed convolution's 52 pre-shuffle channels
        # (PixelShuffle(2): output channel c = channels 4c .. 4c+3) are computed and written, [N, 5, H/s, W/s]
        self.lastconv5 = _ops.Deconv4x4(g("lastconv.0.weight")[:, :20].contiguous(), g("lastconv.0.bias")[:20].contiguous(),
                                        pixel_shuffle=True, device=device)
        self.chain5 = _ops.ConvChain([(self.conv0_0, False), (self.conv0_1, False)] + [(rc, True) for rc in self.convblock]
                                     + [(self.lastconv5, False)])
        self.chain_tail5 = _ops.ConvChain([(self.conv0_1, False)] + [(rc, True) for rc in self.convblock] + [(self.lastconv5, False)])

    def core(self, x):
        """conv0 -> 8 x ResConv (lrelu(conv(x) * beta + x)) -> deconv + PixelShuffle: [N, 13, 4h, 4w]; one library call."""
        return self.chain(x)

    def __call__(self, x, flow=None, scale=1):
        """Reference call form: x is the already concatenated full-resolution input (IFNet_HDv3.py:84-96)."""
        _, _, H, W = x.shape
        h, w = int(H * (1.0 / scale)), int(W * (1.0 / scale))
        x = _ops.resize_bilinear_scale(x, (h, w), scale)
        if flow is not None:
            fl = _ops.affine(_ops.resize_bilinear_scale(flow, (h, w), scale), 1.0 / scale, 0.0)
            x = torch.cat((x, fl), 1)
        tmp = self.core(x)
        return _ops.ifblock_update(tmp, None, H, W, scale, want_mask_feat=True)

    forward = __call__


class IFNet:
    def __init__(self, device=None):
        self.device = device
        self.block = [None] * 5
        self.encode = None

    # --- torch.nn.Module-like surface used by models/rife.py:17-20
    def to(self, device):
        self.device = device
        return self

    def eval(self):
        return self

    def load_state_dict(self, sd, strict=False):
        dev = self.device if self.device is not None else _ops.default_device()
        self.device = dev
        for i in range(5):
            self.block[i] = IFBlock(sd, f"block{i}.", dev)
        self.block0, self.block1, self.block2, self.block3, self.block4 = self.block
        self.encode = Head(sd, "encode.", dev)
        return self

    def forward_pair(self, img0, img1, timestep=0.5, scale_list=(8, 4, 2, 1), f0=None, f1=None, want_flows=True):
        """IFNet.forward (IFNet_HDv3.py:126-177) on separate frames (no 6-channel concat) -> (frame, per-stage flows).
        want_flows=False: only the frame is needed -> the pipeline of forward_pairs (flow updates folded into their
        consumers, the per-stage full-resolution flows are not all materialised); returns (frame, None)."""
        _, _, H, W = img0.shape
        f0 = self.encode(img0, planar=False) if f0 is None else f0
        f1 = self.encode(img1, planar=False) if f1 is None else f1
        if not want_flows:
            return self.forward_pairs([(img0, img1, timestep, f0, f1)], scale_list)[0], None
        flow = tmp = None
        s_prev = 1.0
        flow_list = []
        for i in range(5):
            s = scale_list[i]
            xin = _ops.ifblock_input(img0, img1, f0, f1, timestep, flow, tmp, s_prev, s)
            tmp = self.block[i].core(xin)  # [1,13,H/s,W/s]: flow delta (4), mask (1), feat (8)
            flow = _ops.ifblock_update(tmp, flow, H, W, s)  # only the running flow exists at full resolution
            flow_list.append(flow)
            s_prev = s
        return _ops.warp_blend(img0, img1, flow, tmp, s_prev), flow_list

    @staticmethod
    def _lds_ok(s, s_prev):
        return _ops.LDS_STAGE_INPUT and s_prev == 2 * s and s in (1, 2, 4, 8, 16, 32)

    def forward_pairs(self, items, scale_list=(8, 4, 2, 1), first=0, last=5, state=None):
        """Several interpolations of one frame size in one pass: items = [(img0, img1, timestep, f0, f1), ...].
        The samples are independent (IFNet_HDv3.py:126-177 applied to each); stacking them makes every
        convolution of a stage one launch over the batch, which fills the 256 CUs better than the 1/16..1/64
        resolution maps of a single 1080p frame do and halves the launch count of a `-t 2` step.
        Stages [first, last) are run; last < 5 returns the carried state (per-sample flows, head output, its scale,
        pending) instead of frames, and `state` resumes from it (models/rife.py runs the low-resolution stages of the
        NEXT step on a side stream).
        `pending`: the flow update of the last stage run (flow += up(tmp[:4]) * s, IFNet_HDv3.py:92-95,160) has not been
        applied to `flows` yet -- it is folded into the consumer where that one visits every full-resolution pixel
        exactly once (next stage's input kernel at scale <= 2, the final warp_blend): no separate read-modify-write
        pass over the full-resolution flow for the two full-resolution stages."""
        B = len(items)
        M = _lib.MAX_STAGE_ITEMS
        wide = B > M and (state[3] == "lazy" if state is not None else self._lazy_ok(scale_list))
        if B > M and not wide:
            # the batched glue launches take at most M items (drba_hip.h DRBA_MAX_STAGE_ITEMS): a step with more frames to
            # synthesise (`-t 6`, 24 -> 144 fps, ...) runs as independent groups of M -- the samples do not interact
            parts = []
            lazy = state is not None and state[3] == "lazy"
            for a in range(0, B, M):
                if state is None:
                    st = None
                elif lazy:  # state[0]: the terms [(head output [B,13,h,w], scale), ...]
                    st = ([(t[a:a + M], s) for t, s in state[0]], state[1][a:a + M], state[2], state[3])
                else:
                    st = (list(state[0][a:a + M]), None if state[1] is None else state[1][a:a + M], state[2], state[3])
                parts.append(self.forward_pairs(items[a:a + M], scale_list, first, last, st))
            if last < 5:
                if parts[0][3] == "lazy":
                    terms = [(torch.cat([p[0][i][0] for p in parts], 0), parts[0][0][i][1]) for i in range(len(parts[0][0]))]
                    return (terms, torch.cat([p[1] for p in parts], 0), parts[0][2], "lazy")
                return ([f for p in parts for f in p[0]], torch.cat([p[1] for p in parts], 0), parts[0][2], parts[0][3])
            return [f for p in parts for f in p]
        _, _, H, W = items[0][0].shape
        if state is not None and state[3] == "lazy" or state is None and self._lazy_ok(scale_list):
            return self._forward_pairs_lazy(items, scale_list, first, last, state)
        flows, tmp, s_prev, pending = state if state is not None else ([None] * B, None, 1.0, False)
        flows = list(flows)
        dev = items[0][0].device
        for i in range(first, last):
            s = scale_list[i]
            h, w = int(np.floor(H * (1.0 / s))), int(np.floor(W * (1.0 / s)))
            lds = i > 0 and self._lds_ok(s, s_prev)
            fold = pending and lds and s <= 2
            final = i == 4  # the last stage: only flow and mask of its head output are read
            # every item's glue kernel of a stage is ONE launch (blockIdx.y = item): these launches are latency-bound on the
            # small maps, and each one costs the gap a dependent dispatch waits for its predecessor
            if pending and not fold:
                flows = _ops.flow_updates(tmp, flows, H, W, s_prev)
            if lds and s == 1 and _ops.stage_conv0_ok(self.block[i].conv0_0, H, W, s, s_prev):
                # scale 1: the 52-channel stage input (435 MB per 1080p sample) is consumed by conv0[0] inside the kernel
                # that gathers it and never written
                y0, fl = _ops.stage_conv0(items, flows, tmp, s_prev, self.block[i].conv0_0, fold=fold)
                flows = fl if fold else flows
                tmp = (self.block[i].chain_tail5 if final else self.block[i].chain_tail)(y0)
            else:
                xin = torch.empty((B, 52 if i else 39, h, w), dtype=torch.float32, device=dev)
                if fold:
                    flows = _ops.stage_inputs(items, flows, tmp, s_prev, s, xin, fold=True)
                elif lds:
                    _ops.stage_inputs(items, flows, tmp, s_prev, s, xin)
                else:
                    _ops.stage_inputs(items, flows, tmp, s_prev, s, xin, lds=False)
                tmp = (self.block[i].chain5 if final else self.block[i].chain)(xin)  # [B,13,H/s,W/s]: flow delta (4), mask (1), feat (8)
            s_prev = s
            # leave the update to the consumer if that one can fold it
            # (the final warp_blend_fold takes scale >= 1 only: with a model scale > 1 the last stage runs at s < 1 and
            # the plain update + warp_blend pair finishes the frame)
            nxt_folds = ((i + 1 < 5 and self._lds_ok(scale_list[i + 1], s) and scale_list[i + 1] <= 2)
                         or (i + 1 == 5 and s >= 1))
            pending = bool(nxt_folds)
            if not pending:
                flows = _ops.flow_updates(tmp, flows, H, W, s)
        if last < 5:
            return flows, tmp, s_prev, pending
        if pending:
            return [_ops.warp_blend_fold(it[0], it[1], flows[k], tmp[k:k + 1], s_prev) for k, it in enumerate(items)]
        return [_ops.warp_blend(it[0], it[1], flows[k], tmp[k:k + 1], s_prev) for k, it in enumerate(items)]

    def _lazy_ok(self, scale_list):
        """The running flow as terms (ops.LAZY_FLOW): every warped stage through the LDS gather (scales 32..1, each half the
        previous one) and a last stage at scale >= 1; anything else (a model scale > 1, ...) keeps the materialised flow."""
        sl = list(scale_list[:5])
        return bool(_ops.LAZY_FLOW and _ops.LDS_STAGE_INPUT and _ops.PAIR_FEATURES and len(sl) == 5 and sl[4] >= 1
                    and all(self._lds_ok(sl[i], sl[i - 1]) for i in range(1, 5)))

    def _forward_pairs_lazy(self, items, scale_list, first, last, state):
        """forward_pairs without a full-resolution flow tensor: IFNet_HDv3.py:146-160's flow = flow + up(tmp_i[:, :4]) * s_i is kept
        as the list of head outputs (`terms`, 1/32 .. 1/2 resolution) and evaluated by the kernels that need it, at their
        sample points -- no ifblock_update pass after a stage, no flow read or written by a gather, one warp_blend launch
        for all items.  State between calls: (terms, newest head output, its scale, "lazy").
        More items than one glue launch takes (DRBA_MAX_STAGE_ITEMS; round 6): the glue kernels -- stage inputs, fused stage
        convolutions, the final blend -- run in chunks of that many items on batch slices of the stage's tensors, the convolution
        chains run ONCE over the whole batch (their layers are latency- / tail-bound on the small maps: 64 ch 136x240 runs at 105
        TFLOP/s at N = 8 and 115 at N = 16, tools/exp/conv_batch_scaling.py)."""
        B = len(items)
        M = _lib.MAX_STAGE_ITEMS
        chunks = [(a, min(a + M, B)) for a in range(0, B, M)]
        _, _, H, W = items[0][0].shape
        dev = items[0][0].device
        terms, tmp, s_prev, _ = state if state is not None else ([], None, 1.0, "lazy")
        terms = list(terms)
        cut = lambda a, b: [(t[a:b], sc) for t, sc in terms]  # noqa: E731
        for i in range(first, last):
            s = scale_list[i]
            h, w = int(np.floor(H * (1.0 / s))), int(np.floor(W * (1.0 / s)))
            if i == 0:
                xin = torch.empty((B, 39, h, w), dtype=torch.float32, device=dev)
                for a, b in chunks:
                    _ops.stage_inputs(items[a:b], None, None, s_prev, s, xin[a:b], lds=False)
                tmp_new = self.block[i].core(xin)
            else:
                final = i == 4  # the last stage: only flow and mask of its head output are read (IFBlock.lastconv5)
                conv = self.block[i].conv0_0
                if s in (1, 2) and _ops.stage_conv0_ok(conv, H, W, s, s_prev, items=items):
                    # scale 1, and scale 2 where the two-term kernel takes it (the flow as terms, frames with their [H,W,4] copies)
                    if len(chunks) == 1:
                        y0, _ = _ops.stage_conv0(items, None, tmp, s_prev, conv, terms=terms, scale=s)
                    else:
                        hs, ws = H // int(s), W // int(s)
                        y0 = torch.empty((B, conv.cout, (hs - 1) // 2 + 1, (ws - 1) // 2 + 1), dtype=torch.float32, device=dev)
                        for a, b in chunks:
                            _ops.stage_conv0(items[a:b], None, tmp[a:b], s_prev, conv, terms=cut(a, b), scale=s, out=y0[a:b])
                    tmp_new = (self.block[i].chain_tail5 if final else self.block[i].chain_tail)(y0)
                else:
                    xin = torch.empty((B, 52, h, w), dtype=torch.float32, device=dev)
                    for a, b in chunks:
                        _ops.stage_inputs(items[a:b], None, tmp[a:b], s_prev, s, xin[a:b], terms=cut(a, b))
                    tmp_new = (self.block[i].chain5 if final else self.block[i].chain)(xin)
                terms.append((tmp, s_prev))
            tmp, s_prev = tmp_new, s
        if last < 5:
            return terms, tmp, s_prev, "lazy"
        frames = []
        for a, b in chunks:
            frames += _ops.warp_blend_lazy(items[a:b], cut(a, b), tmp[a:b], s_prev)
        return frames

    def __call__(self, x, timestep=0.5, scale_list=(8, 4, 2, 1), training=False, fastmode=True, ensemble=False,
                 f0=None, f1=None):
        c = x.shape[1] // 2
        return self.forward_pair(x[:, :3].contiguous(), x[:, c:c + 3].contiguous(), timestep, scale_list, f0, f1)

    forward = __call__
