"""GMFSS (non-union) Model on the HIP path (reference models/model_gmfss/GMFSS.py:19-24: GridNet(6*2, 64*2, 128*2, 192*2, 3),
a MetricNet without Tanh()*10, no auxiliary RIFE frame, no swap masks).  The union and non-union networks share one
implementation (drba_amd/models/model_gmfss_union/GMFSS.py, `union` flag); this module is the non-union import path with the
non-union defaults, so `from models.model_gmfss.GMFSS import Model; Model()` builds the network the reference's path builds."""
from drba_amd.models.model_gmfss_union.GMFSS import Model as _Model


class Model(_Model):
    def __init__(self, union=False):
        super().__init__(union=union)
