"""RIFE wrapper with the DRBA call surface (reference models/rife.py:15-109), HIP path.

    RIFE(weights, scale, device).inference_ts(I0, I1, ts) -> [frames]
    RIFE(...).inference_ts_drba(I0, I1, I2, ts, reuse=None, linear=False) -> ([frames], reuse)

Everything runs in fp32 (the reference's CPU autocast path is bf16 and deviates ~2e-3 from
its own fp32 evaluation; parity is against the fp32 evaluation).
"""
import os
import weakref

import numpy as np
import torch

from drba_amd import ops as _ops
from drba_amd.models.lookahead import Lookahead, shared_stream
from drba_amd.models.lookahead import split as split_lookahead
from drba_amd.models.drm import calc_drm_rife
from drba_amd.models.rife_426_heavy.IFNet_HDv3 import IFNet
from drba_amd.models.utils.tools import convert, load_weights


class RIFE:
    supports_lookahead = True  # inference_ts_drba(..., lookahead=next frame): see prefetch_flow
    STAT_KEYS = ("groups_formed", "staged_groups", "group_collects", "groups_dropped", "single_steps", "encoder_prefetch_hits",
                 "encoder_prefetch_misses", "pairflow_prefetch_hits", "lookahead_hits")
    stats = None     # per instance: {key: count}, see __init__

    def _count(self, key):
        if self.stats is None:
            self.stats = dict.fromkeys(self.STAT_KEYS, 0)
        self.stats[key] += 1
    _look = None     # models/lookahead.Lookahead, created on first use

    def __init__(self, weights="weights/train_log_rife_426_heavy", scale=1.0, device=None):
        device = _ops.default_device() if device is None else torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("drba_amd RIFE runs on the MI355X HIP path only; there is no CPU fallback")
        if isinstance(weights, dict):  # an already-loaded state dict (no 'module.' prefix)
            sd = weights
        else:
            sd = convert(load_weights(os.path.join(weights, "flownet.pkl")))
        self.device = device
        _ops.status_init(device)  # the two-term fp16 kernels report an overflow from now on (checked once per call below)
        self.ifnet = IFNet().to(device).eval()
        self.ifnet.load_state_dict(sd, strict=False)
        self.scale = scale
        self.scale_list = [16 / scale, 8 / scale, 4 / scale, 2 / scale, 1 / scale]
        self.pad_size = 64
        # which path the calls took (not in the reference): a caller that clones / slices frames loses the identity-keyed
        # prefetches and groups silently -- correct results, slower -- so the hit counts are kept where bench.py and the
        # tests can assert them (`groups_formed`, `staged_groups`: _drba_group; `group_collects`: calls that only collected
        # a step computed ahead; `single_steps`: calls that ran the one-step path; `encoder_prefetch_hits` / `_misses`:
        # _encoded; `pairflow_prefetch_hits`: _pair_flow; `lookahead_hits`: one-step lookahead results picked up)
        self.stats = dict.fromkeys(self.STAT_KEYS, 0)

    def encode(self, img):
        return self.ifnet.encode(img[:, :3])

    def inference_ts(self, I0, I1, ts):
        """t == 0 / t == 1 return the input tensor objects themselves (reference rife.py:30-33)."""
        _ops.check_overflow(self.device)
        output, items = [], []
        f0 = f1 = None
        for t in ts:
            if t == 0:
                output.append(I0)
            elif t == 1:
                output.append(I1)
            else:
                if f0 is None:  # encode once per call, not once per t (same values); from prefetch_frame where the driver started it
                    f0, f1 = self._encoded(I0), self._encoded(I1)
                output.append(len(items))  # placeholder: index into the batched pass below
                items.append((I0, I1, float(t), f0, f1))
        return self._fill(output, items)

    def _fill(self, output, items):
        """Run the queued interpolations as one stacked IFNet pass and put the frames where their indices are."""
        if items:
            frames = self.ifnet.forward_pairs(items, self.scale_list)
            output = [frames[o] if isinstance(o, int) else o for o in output]
        return output

    def calc_flow(self, a, b, f0=None, f1=None):
        """Coarse bidirectional flow from block0 at t=0.5, reversed to the frame's own time by a
        forward splat (reference rife.py:41-75); one fused kernel pair per direction."""
        _, _, H, W = a.shape
        f0 = self._encoded(a) if f0 is None else f0
        f1 = self._encoded(b) if f1 is None else f1
        s = self.scale_list[0]
        xin = _ops.ifblock_input(a, b, f0, f1, 0.5, None, None, 1.0, s)
        flow = _ops.ifblock_update(self.ifnet.block[0].core(xin), None, H, W, s)
        # 2 * (-splat_avg(flow50)), holes -> 2*max(H, W); both directions in one launch pair: [1,4,H,W] is [2,2,H,W]
        rev = _ops.flow_reverse(flow.reshape(2, 2, H, W))
        flow01, flow10 = rev[0:1], rev[1:2]
        if _ops.PAIR_FEATURES and f0.is_cuda:  # the stages' pair-interleaved copies, made where the features are made
            _ops.pair_interleaved(f0)          # (with a lookahead this runs on the side stream, off the critical path)
            _ops.pair_interleaved(f1)
        return flow01, flow10, f0, f1

    def calc_flow_batch(self, pairs):
        """calc_flow for several frame pairs [(a, b, fa, fb)] at once: block 0 runs ONCE over the stacked pairs (its 11 launches
        are latency-bound on a 1/16-resolution map: four pairs cost little more than one), the flow reversal once over all
        directions.  -> [(flow01, flow10, fa, fb)], the values calc_flow returns pair by pair."""
        B = len(pairs)
        _, _, H, W = pairs[0][0].shape
        s = self.scale_list[0]
        h, w = int(np.floor(H * (1.0 / s))), int(np.floor(W * (1.0 / s)))
        xin = torch.empty((B, 39, h, w), dtype=torch.float32, device=pairs[0][0].device)
        _ops.stage_inputs([(a, b, 0.5, fa, fb) for a, b, fa, fb in pairs], None, None, 1.0, s, xin, lds=False)
        flow = _ops.flow_updates(self.ifnet.block[0].core(xin), [None] * B, H, W, s, whole=True)  # [B,4,H,W]
        rev = _ops.flow_reverse(flow.reshape(2 * B, 2, H, W))
        out = []
        for k, (a, b, fa, fb) in enumerate(pairs):
            if _ops.PAIR_FEATURES:
                _ops.pair_interleaved(fa)
                _ops.pair_interleaved(fb)
            out.append((rev[2 * k:2 * k + 1], rev[2 * k + 1:2 * k + 2], fa, fb))
        return out

    def _group_flows(self, F, n, fa):
        """calc_flow(F[j+1], F[j+2]) for j = 0 .. n-1 (fa: features of F[1] if known): taken from prefetch_pair where the driver
        started it, the rest in one batched pass."""
        P, todo = [None] * n, []
        for j in range(n):
            a, b = F[j + 1], F[j + 2]
            c = getattr(b, "_drba_pairflow", None)
            if c is not None and c[0]() is a and c[3] == id(self):
                P[j] = self._pair_flow(a, b, None)
            else:
                todo.append(j)
        if todo:
            feats = {}
            for j in todo:
                for x in (F[j + 1], F[j + 2]):
                    if id(x) not in feats:
                        feats[id(x)] = fa if (x is F[1] and fa is not None) else self._encoded(x)
            res = self.calc_flow_batch([(F[j + 1], F[j + 2], feats[id(F[j + 1])], feats[id(F[j + 2])]) for j in todo])
            for j, r in zip(todo, res):
                P[j] = r
        return P

    def intake_stream(self, device):
        """The stream a driver may run a newly read frame's to_inp and scene test on (the prefetch stream: the frame's encoder is
        started there anyway, so everything that depends only on the frame stays clear of the synthesis queue).  None: no such
        stream (the encoders run in the caller's stream)."""
        if self.ENC_ON_MAIN:
            return None
        if getattr(self, "_enc_stream", None) is None:
            self._enc_stream = shared_stream(device, "prefetch")
        return self._enc_stream

    def prefetch_frame(self, I):
        """Optional (not in the reference): start the context encoder of a frame the driver has just read -- it depends on
        nothing but the frame -- on its own HIP stream.  The lookahead's serial chain (encoder -> block0 -> flow reversal ->
        DRM -> low-resolution stages) is the critical path of a step; with the driver reading two frames ahead the
        encoder (4 full-chip launches + the pair-interleaved copy) leaves that chain and runs beside it.  calc_flow picks
        the result up by frame identity; without a prefetch it encodes in place as before."""
        if not I.is_cuda or getattr(I, "_drba_enc", None) is not None:
            return
        dev = I.device
        main = torch.cuda.current_stream(dev)
        if getattr(self, "_enc_stream", None) is None:
            self._enc_stream = main if self.ENC_ON_MAIN else shared_stream(dev, "prefetch")
        ready = torch.cuda.Event()
        ready.record(main)  # the frame was produced (to_inp) on the caller's stream
        with torch.cuda.stream(self._enc_stream):
            self._enc_stream.wait_event(ready)
            I.record_stream(self._enc_stream)
            f = self.ifnet.encode(I, planar=False)
            if _ops.PAIR_FEATURES:
                _ops.pair_interleaved(f)
            done = torch.cuda.Event()
            done.record(self._enc_stream)
        I._drba_enc = (f, done, id(self))

    def prefetch_pair(self, a, b):
        """Optional: calc_flow(a, b) -- block0 on the 1/16-resolution map and the flow reversal, ~25 serial launches -- on
        the prefetch stream, behind the two frames' encoders.  With the driver reading two frames ahead this is the pair
        the NEXT call's lookahead starts from, so the lookahead stream's chain begins at the DRM maps."""
        if not a.is_cuda or getattr(b, "_drba_pairflow", None) is not None:
            return
        if self.GROUP > 1 and self.BATCH_COARSE:  # the coarse flows of a group's new pairs are made in ONE batched pass where the group is staged
            self.prefetch_frame(a)
            self.prefetch_frame(b)
            return
        self.prefetch_frame(a)
        self.prefetch_frame(b)
        with torch.cuda.stream(self._enc_stream):
            res = self.calc_flow(a, b, f0=self._encoded(a), f1=self._encoded(b))
            done = torch.cuda.Event()
            done.record(self._enc_stream)
        # (a weak reference to `a`: a strong one would chain every frame -- with its features -- to its successor for the
        # length of the clip)
        b._drba_pairflow = (weakref.ref(a), res, done, id(self))

    def _pair_flow(self, a, b, fa=None):
        """calc_flow(a, b), from prefetch_pair if it was started there (the consumer's stream waits for it)."""
        c = getattr(b, "_drba_pairflow", None)
        if c is not None and c[0]() is a and c[3] == id(self):
            cur = torch.cuda.current_stream(a.device)
            cur.wait_event(c[2])
            self._count("pairflow_prefetch_hits")
            for t in c[1]:
                t.record_stream(cur)
                fp = getattr(t, "_drba_pair", None)
                if fp is not None:
                    fp.record_stream(cur)
            return c[1]
        return self.calc_flow(a, b, f0=fa)

    def _encoded(self, I):
        """encode(I), from prefetch_frame's stream if it was started there (the consumer's stream waits for it)."""
        c = getattr(I, "_drba_enc", None)
        if c is not None and c[2] == id(self):
            cur = torch.cuda.current_stream(I.device)
            cur.wait_event(c[1])
            self._count("encoder_prefetch_hits")
            c[0].record_stream(cur)
            fp = getattr(c[0], "_drba_pair", None)
            if fp is not None:
                fp.record_stream(cur)
            return c[0]
        self._count("encoder_prefetch_misses")
        return self.ifnet.encode(I, planar=False)  # the pair-interleaved layout only: what the kernels read (ops.head_fused)

    def warm_reuse(self, Ia, Ib):
        """The `reuse` a DRBA step ending on the pair (Ia, Ib) hands to the next step (rife.py:82-85,109)."""
        flow_ab, flow_ba, fa, fb = self.calc_flow(Ia, Ib)
        return (flow_ba, flow_ab, fb, fa)

    ENC_ON_MAIN = False  # A/B runs: the frames' encoders in the caller's stream (right behind to_inp) instead of the prefetch stream
    SIDE_STAGES = 2  # IFNet stages of the NEXT step / group run on the side stream (class attribute: A/B runs set it).  Round 5, same box, the
    #                  stage at scale 2 being one fused kernel now: 2 -> 1075-1080 frames/s at 1080p, 3 -> 1053, 1 -> 1056-1060, 0 -> 1050-1053; 4K 535-537 against 531

    def _items(self, I0, I1, I2, ts, linear, flow10, flow12, f0, f1, f2, defer=None):
        """DRM maps and the (img0, img1, timestep, f0, f1) work items of one step; output holds pass-through frames
        and, for the frames to synthesise, their index into items.  defer (a list, linear DRM only): the maps are not computed here --
        (flow_self, flow_other, t) is appended and the item carries its index into `defer` as the timestep; the caller computes
        the maps of all its steps in one launch (_ops.drm_rife_linear_many) and calls _fill_drm."""
        output, items = [], []

        def drm_of(fs, fo, t, key):
            if not linear:
                return calc_drm_rife(t, flow10, flow12, False)[key]
            if defer is None:
                return _ops.drm_rife_linear(fs, fo, t, 1e-4)
            defer.append((fs, fo, t))
            return len(defer) - 1

        for t in ts:
            if t == 0:
                output.append(I0)
            elif t == 1:
                output.append(I1)
            elif t == 2:
                output.append(I2)
            elif 0 < t < 1:
                t = 1 - t
                # only the map this frame consumes is computed (the reference builds both, drm.py:89-96)
                drm = drm_of(flow10, flow12, t, "drm_t1_t01")
                output.append(len(items))
                items.append((I1, I0, drm, f1, f0))
            elif 1 < t < 2:
                t = t - 1
                drm = drm_of(flow12, flow10, t, "drm_t1_t12")
                output.append(len(items))
                items.append((I1, I2, drm, f1, f2))
        return output, items

    def prefetch_flow(self, a, b, fa=None, then=None, also_reads=()):
        """Start calc_flow(a, b) on the side stream (a's features `fa` if already known): the encoder of the new
        frame, block0 on a 1/16-resolution map and the flow reversal are small, latency-bound launches that leave
        most of the chip idle, so they are overlapped with the current step's full-resolution stages.  The next
        inference_ts_drba(_, a, b, ...) picks the result up (models/lookahead.py).  `then(result)`, if given, runs
        right after on the same stream and its return value is kept alongside."""
        if self._look is None:
            self._look = Lookahead()

        def work():
            res = self._pair_flow(a, b, fa)
            return res, (then(res) if then is not None else None)
        self._look.start(a, b, work, inputs=tuple(t for t in (fa,) + tuple(also_reads) if t is not None))

    def _flow_pair(self, a, b, fa):
        """(calc_flow(a, b), staged low-resolution stages or None), from a matching lookahead if there is one."""
        got = self._look.take(a, b) if self._look is not None else None
        if got is not None:
            self._count("lookahead_hits")
        return got if got is not None else (self._pair_flow(a, b, fa), None)

    BATCH_COARSE = True  # with GROUP > 1: calc_flow of a group's new frame pairs in one batched pass (A/B runs)
    GROUP = 4          # consecutive steps per stacked IFNet pass when the driver announces enough frames (class attribute: A/B runs; 1: off)
    _group_out = ()    # the later steps of a group, computed by an earlier call: [(I0, I1, I2, ts, reuse_in, outputs, reuse_out)]
    _look2 = None      # side-stream staging of the NEXT group of steps

    def _group_items(self, F, ts_list, reuse0, P):
        """Work items of the steps (F[j], F[j+1], F[j+2]; ts_list[j]), j = 0 .. len(ts_list) - 1: P[j] = calc_flow(F[j+1], F[j+2]),
        reuse0 what the step before the first one handed on.  -> (per-step outputs with placeholders, per-step item counts,
        all items, per-step reuse: reuses[j] enters step j, reuses[j + 1] leaves it)."""
        outs, counts, items, reuses, jobs = [], [], [], [reuse0], []
        for j, ts in enumerate(ts_list):
            r, p = reuses[j], P[j]
            o, it = self._items(F[j], F[j + 1], F[j + 2], ts, True, r[0], p[0], r[3], r[2], p[3], defer=jobs)
            outs.append(o)
            counts.append(len(it))
            items += it
            reuses.append((p[1], p[0], p[3], p[2]))  # (flow21, flow12, f2, f1), reference rife.py:109
        maps = _ops.drm_rife_linear_many(jobs, 1e-4)  # the group's DRM maps: one launch pair (they were 2 launches per map)
        items = [(a, b, maps[t], fa, fb) for (a, b, t, fa, fb) in items]
        return outs, counts, items, reuses

    def _stage_group(self, F, ts_list, reuse0):
        """Side stream: the group of steps after the one being computed -- steps (F[j], F[j+1], F[j+2]; ts_list[j]).  Coarse
        flows of the new frame pairs (from prefetch_pair where the driver started them), the DRM maps and the first
        SIDE_STAGES IFNet stages of all their frames stacked."""
        if self._look2 is None:
            self._look2 = Lookahead()
        n = len(ts_list)

        def work():
            P = self._group_flows(F, n, reuse0[2])
            outs, counts, items, reuses = self._group_items(F, ts_list, reuse0, P)
            state = self.ifnet.forward_pairs(items, self.scale_list, 0, self.SIDE_STAGES) if items else None
            return {"F": tuple(F), "ts": ts_list, "flow10": reuse0[0], "P": P, "outs": outs, "counts": counts, "items": items,
                    "reuses": reuses, "state": state}
        self._look2.start(F[n], F[n + 1], work, inputs=tuple(t for t in tuple(F[:n]) + tuple(reuse0) if t is not None))

    def _drba_group(self, F, ts_list, reuse, more):
        """Steps (F[j], F[j+1], F[j+2]; ts_list[j]), j = 0 .. g - 1, in ONE stacked IFNet pass: the frames of g steps are 2 g instead
        of 2 samples per launch -- at 1080p 11 % less kernel time per frame for g = 2 and 17 % for g = 4 (the low-resolution
        stages are latency-bound: 18 % / 31 %), and 1 / g of the launches.  Returns the first step's result and keeps the others
        for the next calls.  `more` = ([frames], [ts]) of the group after this one (its frames continue F), whose
        low-resolution stages start on the side stream now, or None."""
        g = len(ts_list)
        staged = self._look2.take(F[g], F[g + 1]) if self._look2 is not None else None
        if not (staged is not None and len(staged["ts"]) == g and all(a is b for a, b in zip(staged["F"], F))
                and staged["flow10"] is reuse[0] and all(np.array_equal(a, b) for a, b in zip(staged["ts"], ts_list))):
            staged = None
        self._count("groups_formed")
        if staged is not None:
            self._count("staged_groups")
            P, outs, counts, items, reuses = (staged[k] for k in ("P", "outs", "counts", "items", "reuses"))
        else:
            got = self._look.take(F[1], F[2]) if self._look is not None else None  # a one-step lookahead of the call before
            P = [got[0]] + self._group_flows(F[1:], g - 1, got[0][3]) if got is not None else self._group_flows(F, g, reuse[2])
            outs, counts, items, reuses = self._group_items(F, ts_list, reuse, P)
        if more is not None:
            self._stage_group(list(F[g:]) + list(more[0]), list(more[1]), reuses[g])
        if staged is not None:
            frames = self.ifnet.forward_pairs(items, self.scale_list, self.SIDE_STAGES, 5, staged["state"]) if items else []
        else:
            frames = self.ifnet.forward_pairs(items, self.scale_list) if items else []
        res, base = [], 0
        for j in range(g):
            res.append([frames[base + o] if isinstance(o, int) else o for o in outs[j]])
            base += counts[j]
        self._group_out = [(F[j], F[j + 1], F[j + 2], ts_list[j], reuses[j], res[j], reuses[j + 1]) for j in range(1, g)]
        return res[0], reuses[1]

    def inference_ts_drba(self, I0, I1, I2, ts, reuse=None, linear=False, lookahead=None):
        """reference rife.py:77-109.  `lookahead` (not in the reference): the frame that will be I2 of the next call,
        or (that frame, the next call's ts).  calc_flow(I2, next) -- and, when the timesteps are known, the DRM maps
        and the first SIDE_STAGES low-resolution IFNet stages of the next step -- run on a side stream under this
        call's full-resolution stages; the next call resumes from there if its arguments match.
        (next, ts1, next2, ts2, ...) -- the driver reading further ahead and vouching that the calls it announces are plain DRBA
        steps -- lets this call compute the next GROUP - 1 steps together with this one (_drba_group): the following calls
        then only collect their results."""
        _ops.check_overflow(self.device)  # raises if a family-4 kernel of an earlier call stored inf / NaN (no synchronisation)
        if self._group_out:
            po, self._group_out = self._group_out[0], self._group_out[1:]
            if (reuse and po[0] is I0 and po[1] is I1 and po[2] is I2 and po[4][0] is reuse[0]
                    and np.array_equal(po[3], np.asarray(ts, dtype=np.float64))):
                self._count("group_collects")
                return po[5], po[6]  # a later step of the group an earlier call computed
            self._group_out = ()     # another call pattern than announced: what was computed ahead is dropped
            self._count("groups_dropped")
        if self.GROUP > 1 and linear and reuse and isinstance(lookahead, (tuple, list)) and len(lookahead) >= 4 and I0.is_cuda:
            # (frame, ts) of the following calls; two or more entries = the driver vouches that all of them are plain DRBA steps
            # (no scene cut up to the last frame it names); one entry alone is the one-step lookahead below
            ahead = []
            for j in range(0, len(lookahead) - 1, 2):
                if lookahead[j] is None or lookahead[j + 1] is None:
                    break
                ahead.append((lookahead[j], np.asarray(lookahead[j + 1], dtype=np.float64)))
            g = min(self.GROUP, len(ahead) + 1)  # a group of g steps needs the entries of the g - 1 calls after this one
            if g >= 2 and len(ahead) >= 2:
                F = [I0, I1, I2] + [a[0] for a in ahead[:g - 1]]
                ts_list = [np.asarray(ts, dtype=np.float64)] + [a[1] for a in ahead[:g - 1]]
                rest = ahead[g - 1:]
                more = ([a[0] for a in rest[:g]], [a[1] for a in rest[:g]]) if len(rest) >= g else None
                return self._drba_group(F, ts_list, reuse, more)
        self._count("single_steps")
        flow10, flow01, f1, f0 = self.calc_flow(I1, I0) if not reuse else reuse
        (flow12, flow21, f1, f2), staged = self._flow_pair(I1, I2, None if reuse is None else reuse[2])
        nxt, ts_nxt = split_lookahead(lookahead)
        if nxt is not None:
            then = None
            if ts_nxt is not None and linear and 0 < self.SIDE_STAGES < 5:
                ts_nxt = np.array(ts_nxt, dtype=np.float64)

                def then(res):  # the next step: (I0, I1, I2) = (I1, I2, nxt), its reuse = (flow21, flow12, f2, f1)
                    out_n, items_n = self._items(I1, I2, nxt, ts_nxt, True, flow21, res[0], f1, f2, res[3])
                    state = self.ifnet.forward_pairs(items_n, self.scale_list, 0, self.SIDE_STAGES) if items_n else None
                    return {"I0": I1, "flow10": flow21, "ts": ts_nxt, "output": out_n, "items": items_n, "state": state}
            self.prefetch_flow(I2, nxt, f2, then, also_reads=(I1, flow21, f1))
        if (staged is not None and linear and staged["I0"] is I0 and staged["flow10"] is flow10
                and np.array_equal(staged["ts"], np.asarray(ts, dtype=np.float64))):
            output, items = staged["output"], staged["items"]
            if items:
                frames = self.ifnet.forward_pairs(items, self.scale_list, self.SIDE_STAGES, 5, staged["state"])
                output = [frames[o] if isinstance(o, int) else o for o in output]
        else:
            output, items = self._items(I0, I1, I2, ts, linear, flow10, flow12, f0, f1, f2)
            output = self._fill(output, items)
        return output, (flow21, flow12, f2, f1)
