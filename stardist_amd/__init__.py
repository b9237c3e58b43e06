"""stardist_amd: MI355X-native StarDist prediction path.

Public names follow the reference package (`stardist`): StarDist2D, Config2D, nms, geometry.
"""
import os as _os

__version__ = "0.1.0"

# MIOpen's find mode (torch.backends.cudnn.benchmark, enabled by the models) times EVERY applicable solver once per
# convolution shape, including its naive reference kernels, which need up to 4.6 s per call on a 256^3 volume (~140 s of
# warm-up for the 3D network).  They can never win; switching that solver off cuts the first prediction from minutes to
# seconds and changes nothing else (measured: same solvers chosen, same throughput).  Must be set before MIOpen initialises.
_os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "0")


def __getattr__(name):
    # lazy so that `import stardist_amd` (and the C-ABI symbol checks) work without torch
    if name in ("StarDist2D", "StarDist3D", "Config2D", "Config3D"):
        from . import models
        return getattr(models, name)
    raise AttributeError(name)
