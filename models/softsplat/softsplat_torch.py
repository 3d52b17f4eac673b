from drba_amd.models.softsplat.softsplat_torch import softsplat  # noqa: F401
