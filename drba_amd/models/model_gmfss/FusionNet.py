"""reference models/model_gmfss/FusionNet.py: the same GridNet class as model_gmfss_union/FusionNet.py (the non-union model
builds it with 12 input channels, GMFSS.py:23; here the channel counts come from the state dict)."""
from drba_amd.models.model_gmfss_union.FusionNet import GridNet  # noqa: F401
