"""CPU: dry-run of the whole Python plumbing of the HIP path.

Every C-ABI call is replaced by a stub that only validates the argument list against the ctypes
prototype (count + convertibility) and returns 0; tensors stay on the CPU.  This catches NameErrors,
wrong argument counts/orders and shape bookkeeping bugs before a GPU run is spent on them.
No arithmetic is checked here (outputs are uninitialised)."""
import ctypes as C

import numpy as np
import pytest
import torch

from drba_amd import _lib, ops
from drba_amd.utils import synth


class _StubLib:
    def __init__(self, real):
        self._real = real
        self.calls = {}

    def __getattr__(self, name):
        real = getattr(self._real, name)
        if name.endswith(("_pick_cfg", "_packed_floats", "_pack", "_ws_floats", "_supported", "drba_abi_version", "drba_error_string")):
            return real  # pure host functions: run for real
        argtypes = real.argtypes

        def stub(*args):
            assert len(args) == len(argtypes), f"{name}: {len(args)} args, prototype has {len(argtypes)}"
            for a, t in zip(args, argtypes):
                if isinstance(a, t):
                    continue
                t(a)  # raises if not convertible (e.g. a float passed for c_int)
            self.calls[name] = self.calls.get(name, 0) + 1
            return 0
        return stub


@pytest.fixture()
def dry(monkeypatch):
    stub = _StubLib(_lib.load())
    monkeypatch.setattr(_lib, "load", lambda: stub)
    monkeypatch.setattr(ops, "_f32", lambda t, name="tensor": t.float().contiguous())
    monkeypatch.setattr(ops, "_stream", lambda: C.c_void_p(0))
    monkeypatch.setattr(ops, "default_device", lambda: torch.device("cpu"))
    monkeypatch.setattr(ops, "_workspace", lambda dev, n, keep_token=False: torch.empty(int(n), dtype=torch.float32))
    monkeypatch.setattr(ops, "_zero_workspace", lambda dev, n: torch.zeros(int(n), dtype=torch.float32))
    return stub


def test_rife_pipeline_plumbing(dry, monkeypatch):
    from drba_amd.models import rife as rife_mod
    monkeypatch.setattr(rife_mod.torch, "device", torch.device)
    m = rife_mod.RIFE.__new__(rife_mod.RIFE)
    from drba_amd.models.rife_426_heavy.IFNet_HDv3 import IFNet
    m.device = torch.device("cpu")
    m.ifnet = IFNet().to(m.device).eval()
    m.ifnet.load_state_dict(synth.ifnet_state_dict(0))
    m.scale, m.scale_list, m.pad_size = 1.0, [16, 8, 4, 2, 1], 64
    I = [torch.rand(1, 3, 64, 128) for _ in range(3)]
    out, reuse = m.inference_ts_drba(I[0], I[1], I[2], np.array([0.75, 1.0, 1.25]), None, True)
    assert len(out) == 3 and out[1] is I[1] and out[0].shape == (1, 3, 64, 128)
    out2, _ = m.inference_ts_drba(I[0], I[1], I[2], np.array([0.6]), reuse, False)  # non-linear DRM composition
    assert out2[0].shape == (1, 3, 64, 128)
    r = m.inference_ts(I[0], I[1], np.array([0.0, 0.5, 1.0]))
    assert r[0] is I[0] and r[2] is I[1]
    # default pipeline: the running flow as terms (no flow tensor), the scale-1 stage input fused with conv0[0]
    for k in ("drba_conv3x3", "drba_deconv4x4s2", "drba_ifblock_input", "drba_ifblock_input_batch", "drba_ifblock_input_lazy_batch",
              "drba_stage_conv16_batch", "drba_warp_blend_lazy_batch", "drba_ifblock_update", "drba_flow_reverse",
              "drba_drm_rife_linear", "drba_softsplat", "drba_drm_retime"):
        assert dry.calls.get(k, 0) > 0, k
    assert dry.calls.get("drba_ifblock_update_batch", 0) == 0 and dry.calls.get("drba_warp_blend_fold", 0) == 0
    # ... and the pipeline with the materialised flow (a model scale > 1 takes it; ops.LAZY_FLOW / STAGE_CONV_FUSED are its A/B switches)
    monkeypatch.setattr(ops, "LAZY_FLOW", False)
    out3, _ = m.inference_ts_drba(I[0], I[1], I[2], np.array([0.75, 1.25]), None, True)
    assert out3[0].shape == (1, 3, 64, 128)
    monkeypatch.setattr(ops, "STAGE_CONV_FUSED", False)
    m.inference_ts_drba(I[0], I[1], I[2], np.array([0.75, 1.25]), None, True)
    for k in ("drba_ifblock_input_lds_batch", "drba_ifblock_update_batch", "drba_warp_blend_fold"):
        assert dry.calls.get(k, 0) > 0, k


def test_operator_surface_plumbing(dry):
    from drba_amd.models import drm
    from drba_amd.models.rife_426_heavy.IFNet_HDv3 import IFBlock
    from drba_amd.models.rife_426_heavy.warplayer import warp
    from drba_amd.models.softsplat.softsplat import softsplat
    from drba_amd.models.utils import tools
    x, f, mt = torch.rand(1, 3, 16, 24), torch.rand(1, 2, 16, 24), torch.rand(1, 1, 16, 24)
    assert warp(x, f).shape == x.shape
    for mode, metric in (("sum", None), ("avg", None), ("linear", mt), ("soft-zeroeps", mt)):
        assert softsplat(x, f, metric, mode).shape == x.shape
    with pytest.raises(AssertionError):
        softsplat(x, f, None, "soft")
    with pytest.raises(AssertionError):
        softsplat(x, f, mt, "avg")
    for fn in (drm.calc_drm_gmfss, drm.calc_drm_rife_auxiliary):
        for lin in (True, False):
            for mm in ((mt, mt), (None, None)):
                r = fn(0.3, f, f, mm[0], mm[1], lin)
                assert all(v.shape == (1, 1, 16, 24) for v in r.values())
    assert set(drm.calc_drm_rife(0.3, f, f, True)) == {"drm_t1_t01", "drm_t1_t12"}
    assert tools.distance_calculator(f).shape == (1, 1, 16, 24)
    assert tools.resize(x, (20, 30)).shape == (1, 3, 20, 30)
    blk = IFBlock(synth.ifnet_state_dict(0), "block1.", torch.device("cpu"))
    fl, mk, ft = blk(torch.rand(1, 48, 64, 64), torch.rand(1, 4, 64, 64), scale=2)
    assert fl.shape == (1, 4, 64, 64) and mk.shape == (1, 1, 64, 64) and ft.shape == (1, 8, 64, 64)


@pytest.mark.parametrize("scale,size", ((1.0, (128, 256)), (0.5, (256, 512))))
def test_gmfss_union_pipeline_plumbing(dry, monkeypatch, scale, size):
    """GMFSS_UNION + GMFlow + GridNet plumbing: argument lists, shapes, window bookkeeping (no arithmetic)."""
    from drba_amd.models import gmfss_union as gu
    from drba_amd.models.model_gmfss_union.GMFSS import Model
    from drba_amd.models.rife_426_heavy.IFNet_HDv3 import IFNet
    sds = synth.gmfss_union_state_dicts(0)
    cpu = torch.device("cpu")
    m = gu.GMFSS_UNION.__new__(gu.GMFSS_UNION)
    m.model = Model(union=True)
    m.model.load_state_dicts(sds["flownet"], sds["metric"], sds["feat"], sds["fusion"], cpu)
    m.ifnet = IFNet().to(cpu).eval()
    m.ifnet.load_state_dict(sds["rife"])
    m.scale, m.pad_size = scale, 128
    m.scale_list = [16 / scale, 8 / scale, 4 / scale, 2 / scale, 1 / scale]
    H, W = size
    I = [torch.rand(1, 3, H, W) for _ in range(3)]
    out, reuse = m.inference_ts_drba(I[0], I[1], I[2], np.array([0.75, 1.0, 1.25]), None, True)
    assert out[1] is I[1] and out[0].shape == (1, 3, H, W) and out[2].shape == (1, 3, H, W)
    assert reuse[0].shape == (1, 2, H // 2, W // 2) and reuse[2].shape == (1, 1, H // 2, W // 2)
    assert [f.shape[1:] for f in reuse[4]] == [(64, H // 2, W // 2), (128, H // 4, W // 4), (192, H // 8, W // 8)]
    out2, _ = m.inference_ts_drba(I[0], I[1], I[2], np.array([1.4]), reuse, False)
    assert out2[0].shape == (1, 3, H, W)
    r = m.inference_ts(I[0], I[1], np.array([0.0, 0.5, 1.0]))
    assert r[0] is I[0] and r[2] is I[1] and r[1].shape == (1, 3, H, W)
    for k in ("drba_conv_direct", "drba_instance_norm", "drba_linear_split_layernorm", "drba_linear_split", "drba_window_attention",
              "drba_global_expect2", "drba_local_corr_flow", "drba_local_attn_flow", "drba_convex_upsample",
              "drba_flow_warp", "drba_resize_bilinear_ac", "drba_metric_input", "drba_pixel_shuffle2",
              "drba_timestep_fix", "drba_swap_select", "drba_clamp", "drba_channel_normalize3", "drba_add_act",
              "drba_quad_interleave", "drba_softsplat_index", "drba_softsplat_gather_quad"):
        assert dry.calls.get(k, 0) > 0, k


def test_gmfss_pipeline_plumbing(dry):
    from drba_amd.models import gmfss as g
    from drba_amd.models.model_gmfss_union.GMFSS import Model
    sds = synth.gmfss_union_state_dicts(0)
    fusion = synth.seeded_state_dict(synth.gridnet_shapes(12, "head"), 0, "grid.")
    m = g.GMFSS.__new__(g.GMFSS)
    m.model = Model(union=False)
    m.model.load_state_dicts(sds["flownet"], sds["metric"], sds["feat"], fusion, torch.device("cpu"))
    m.scale, m.pad_size = 1.0, 64
    I = [torch.rand(1, 3, 128, 256) for _ in range(3)]
    out, reuse = m.inference_ts_drba(I[0], I[1], I[2], np.array([0.75, 1.25]), None, True)
    assert out[0].shape == (1, 3, 128, 256) and len(reuse) == 6
    assert m.inference_ts(I[0], I[1], np.array([0.5]))[0].shape == (1, 3, 128, 256)


@pytest.mark.parametrize("scale,n_items", ((1.0, 7), (2.0, 2), (0.5, 9)))
def test_rife_many_items_and_scale_plumbing(dry, scale, n_items):
    """More synthesised frames per step than one batched glue launch takes (DRBA_MAX_STAGE_ITEMS) run as groups, also when
    the step is split at a stage boundary (the lookahead's carried state); model scale 2 ends on a stage above the frame
    resolution, which must take the plain update + warp_blend pair (drba_warp_blend_fold refuses scale < 1)."""
    from drba_amd.models.rife_426_heavy.IFNet_HDv3 import IFNet
    net = IFNet().to(torch.device("cpu")).eval()
    net.load_state_dict(synth.ifnet_state_dict(0))
    sl = [16 / scale, 8 / scale, 4 / scale, 2 / scale, 1 / scale]
    H, W = 128, 128
    img = [torch.rand(1, 3, H, W) for _ in range(2)]
    f = [torch.rand(1, 16, H, W) for _ in range(2)]
    items = [(img[0], img[1], 0.1 * (k + 1), f[0], f[1]) for k in range(n_items)]
    frames = net.forward_pairs(items, sl)
    assert len(frames) == n_items and all(x.shape == (1, 3, H, W) for x in frames)
    state = net.forward_pairs(items, sl, 0, 3)
    lazy = state[3] == "lazy"  # the running flow as terms: state[0] = [(head output [B,13,h,w], scale)] of stages 0, 1
    assert lazy == (scale <= 1)
    if lazy:
        assert len(state[0]) == 2 and all(t.shape[0] == n_items for t, _ in state[0]) and state[1].shape[0] == n_items
    else:
        assert len(state[0]) == n_items and state[1].shape[0] == n_items
    frames = net.forward_pairs(items, sl, 3, 5, state)
    assert len(frames) == n_items
    groups = -(-n_items // _lib.MAX_STAGE_ITEMS)
    if n_items > _lib.MAX_STAGE_ITEMS:
        assert dry.calls["drba_ifblock_input_batch"] >= 2 * groups
    if scale > 1:
        assert dry.calls.get("drba_warp_blend", 0) == 2 * n_items and dry.calls.get("drba_warp_blend_fold", 0) == 0
    else:  # one launch per group of items, no flow update pass at all
        assert dry.calls.get("drba_warp_blend_lazy_batch", 0) == 2 * groups and dry.calls.get("drba_ifblock_update_batch", 0) == 0
        assert dry.calls.get("drba_ifblock_input_lazy_batch", 0) > 0


def test_non_union_module_paths_build_the_non_union_network(dry):
    """The reference has TWO module trees: models/model_gmfss (GMFSS.py:19-24: GridNet(6*2, ...), MetricNet.py:23-44: no Tanh()*10)
    and models/model_gmfss_union (GridNet(9, ...), Tanh()*10).  Every import path a reference user has resolves here, and the
    non-union path builds the non-union network by DEFAULT (it used to alias the union classes with the union defaults)."""
    import importlib
    for mod, names in (("models.model_gmfss.GMFSS", ["Model"]), ("models.model_gmfss.MetricNet", ["MetricNet", "backwarp"]),
                       ("models.model_gmfss.FeatureNet", ["FeatureNet"]), ("models.model_gmfss.FusionNet", ["GridNet"]),
                       ("models.model_gmfss_union.GMFSS", ["Model"]), ("models.model_gmfss_union.MetricNet", ["MetricNet"]),
                       ("models.model_gmfss_union.FeatureNet", ["FeatureNet"]), ("models.model_gmfss_union.FusionNet", ["GridNet"]),
                       ("models.gmflow.gmflow", ["GMFlow"]), ("models.gmfss", ["GMFSS"]), ("models.gmfss_union", ["GMFSS_UNION"]),
                       ("models.rife", ["RIFE"]), ("models.rife_426_heavy.IFNet_HDv3", ["IFNet"]),
                       ("models.rife_426_heavy.warplayer", ["warp"]), ("models.softsplat.softsplat", ["softsplat"]),
                       ("models.softsplat.softsplat_torch", ["softsplat"]), ("models.drm", ["calc_drm_rife", "calc_drm_gmfss", "get_drm_t"]),
                       ("models.utils.tools", ["VideoFI_IO", "to_inp", "to_out", "check_scene", "TMapper"])):
        m = importlib.import_module(mod)
        for n in names:
            assert hasattr(m, n), (mod, n)
    from models.model_gmfss.GMFSS import Model
    from models.model_gmfss.MetricNet import MetricNet
    from models.model_gmfss_union.GMFSS import Model as UnionModel
    from models.model_gmfss_union.MetricNet import MetricNet as UnionMetricNet
    cpu = torch.device("cpu")
    sds = synth.gmfss_union_state_dicts(0)
    fusion12 = synth.seeded_state_dict(synth.gridnet_shapes(12, "head"), 0, "grid.")
    m = Model()  # no arguments, as in the reference
    assert m.union is False and UnionModel().union is True
    m.load_state_dicts(sds["flownet"], sds["metric"], sds["feat"], fusion12, cpu)
    assert m.fusionnet.head[0].first.cin == 12          # GridNet(6 * 2, ...): img0, I1t, I2t, img1
    assert m.metricnet.conv_out.act == ops.Conv3x3.ACTS[None]          # metric_out ends on the convolution
    assert MetricNet(sds["metric"], cpu).conv_out.act == ops.Conv3x3.ACTS[None]
    assert UnionMetricNet(sds["metric"], cpu).conv_out.act == ops.Conv3x3.ACTS["tanh10"]
    u = UnionModel()
    u.load_state_dicts(sds["flownet"], sds["metric"], sds["feat"], sds["fusion"], cpu)
    assert u.fusionnet.head[0].first.cin == 9 and u.metricnet.conv_out.act == ops.Conv3x3.ACTS["tanh10"]
    # the splat stage hands the non-union GridNet its 12-channel input: [img0, I1t, I2t, img1]
    I0, I1 = torch.rand(1, 3, 128, 256), torch.rand(1, 3, 128, 256)
    bufs = m.fusion_inputs(I0, I1, m.reuse(I0, I1, 1.0), 0.4, 0.6)
    assert bufs[0].shape == (1, 12, 64, 128) and [b.shape[1] for b in bufs[1:]] == [128, 256, 384]
