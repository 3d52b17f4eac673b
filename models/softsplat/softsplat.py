from drba_amd.models.softsplat.softsplat import softsplat  # noqa: F401
