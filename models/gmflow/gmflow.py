from drba_amd.models.gmflow.gmflow import GMFlow  # noqa: F401
