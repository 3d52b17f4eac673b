"""Name kept for import compatibility with the reference (models/softsplat/softsplat_torch.py);
it resolves to the same HIP operator, not to a torch fallback."""
from drba_amd.ops import softsplat  # noqa: F401

__all__ = ["softsplat"]
