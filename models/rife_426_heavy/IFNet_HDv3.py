from drba_amd.models.rife_426_heavy.IFNet_HDv3 import Head, IFBlock, IFNet  # noqa: F401
