"""Import-path compatibility with the reference tree (models.rife, models.drm, ...): thin re-exports of drba_amd.models."""
