from drba_amd.models.model_gmfss_union.FeatureNet import FeatureNet  # noqa: F401
