"""Executors for tests/cases.py: the CPU oracle and the HIP product path."""
import torch


class OracleBackend:
    """CPU fp32 oracle (oracle/) — the checker."""
    name = "oracle"
    dev = torch.device("cpu")

    def __init__(self):
        import oracle
        self._o = oracle
        self.warp = oracle.ops.backwarp
        self.softsplat = oracle.ops.softsplat
        self.distance = oracle.ops.distance
        self.resize = oracle.ops.resize
        self.calc_drm_rife = oracle.drm.calc_drm_rife
        self.calc_drm_gmfss = oracle.drm.calc_drm_gmfss
        self.calc_drm_rife_auxiliary = oracle.drm.calc_drm_rife_auxiliary
        self.get_drm_t = oracle.drm.drm_to_t
        self.ssim_matlab = oracle.scdet.ssim_matlab
        self.check_scene = oracle.scdet.check_scene

    def make_rife(self, sd, scale):
        return self._o.rife.RifeOracle(sd, scale)

    def make_gmfss_union(self, sds, scale):
        return self._o.gmfss.GmfssUnionOracle(sds["flownet"], sds["metric"], sds["feat"], sds["fusion"], sds["rife"], scale)

    def make_gmfss(self, sds, scale):
        return self._o.gmfss.GmfssOracle(sds["flownet"], sds["metric"], sds["feat"], sds["fusion"], scale)

    def featurenet(self, sd, x):
        return self._o.gmfss.featurenet(sd, x)

    def metricnet(self, sd, h0, h1, f01, f10, union=True):
        return self._o.gmfss.metricnet(sd, h0, h1, f01, f10, union)

    def gmflow(self, sd, a, b):
        return self._o.gmflow.gmflow(sd, a, b)


class HipBackend:
    """The product: drba_amd's reference-named call surface running on the HIP library."""
    name = "hip"

    def __init__(self):
        from drba_amd.models import drm
        from drba_amd.models.rife import RIFE
        from drba_amd.models.rife_426_heavy.warplayer import warp
        from drba_amd.models.softsplat.softsplat import softsplat
        from drba_amd.models.utils import tools
        self.dev = torch.device("cuda:0")
        self._RIFE = RIFE
        self.warp = warp
        self.softsplat = softsplat
        self.distance = tools.distance_calculator
        self.resize = tools.resize
        self.calc_drm_rife = drm.calc_drm_rife
        self.calc_drm_gmfss = drm.calc_drm_gmfss
        self.calc_drm_rife_auxiliary = drm.calc_drm_rife_auxiliary
        self.get_drm_t = drm.get_drm_t
        self.check_scene = tools.check_scene

    def make_rife(self, sd, scale):
        return self._RIFE(weights=sd, scale=scale, device=self.dev)

    def make_gmfss_union(self, sds, scale):
        from drba_amd.models.gmfss_union import GMFSS_UNION
        return GMFSS_UNION(weights=sds, scale=scale, device=self.dev)

    def make_gmfss(self, sds, scale):
        from drba_amd.models.gmfss import GMFSS
        return GMFSS(weights=sds, scale=scale, device=self.dev)

    def featurenet(self, sd, x):
        from drba_amd.models.model_gmfss_union.FeatureNet import FeatureNet
        return FeatureNet(sd, self.dev)(x)

    def metricnet(self, sd, h0, h1, f01, f10, union=True):
        from drba_amd.models.model_gmfss_union.MetricNet import MetricNet
        return MetricNet(sd, self.dev, tanh10=union)(h0, h1, f01, f10)

    def gmflow(self, sd, a, b):
        from drba_amd.models.gmflow.gmflow import GMFlow
        return GMFlow(sd, self.dev)(a, b)
