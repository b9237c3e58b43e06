"""stardist_amd: MI355X-native StarDist prediction path.

Public names follow the reference package (`stardist`): StarDist2D, Config2D, nms, geometry.
"""
__version__ = "0.1.0"


def __getattr__(name):
    # lazy so that `import stardist_amd` (and the C-ABI symbol checks) work without torch
    if name in ("StarDist2D", "StarDist3D", "Config2D", "Config3D"):
        from . import models
        return getattr(models, name)
    raise AttributeError(name)
