from drba_amd.models.model_gmfss.FeatureNet import FeatureNet  # noqa: F401
