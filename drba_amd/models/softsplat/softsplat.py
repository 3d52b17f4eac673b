"""Forward (splatting) warp on the HIP library.

Same signature and mode grammar as the reference operator
(models/softsplat/softsplat.py:248 / models/softsplat/softsplat_torch.py:19-22):
softsplat(tenIn, tenFlow, tenMetric, strMode), strMode = {sum,avg,linear,soft}[-{addeps,zeroeps,clipeps}].
The reference picks a cupy/CUDA or a torch implementation at import time; here there is
one implementation, the hand-written gfx950 scatter kernel (drba_amd/csrc/splat_warp.hip).
"""
from drba_amd.ops import softsplat  # noqa: F401

__all__ = ["softsplat"]
