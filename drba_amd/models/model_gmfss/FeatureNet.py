"""reference models/model_gmfss/FeatureNet.py: the same network as model_gmfss_union/FeatureNet.py."""
from drba_amd.models.model_gmfss_union.FeatureNet import FeatureNet  # noqa: F401
