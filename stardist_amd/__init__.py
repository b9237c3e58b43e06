"""stardist_amd: MI355X-native StarDist prediction path.

Public names follow the reference package (`stardist/__init__.py`): what `from stardist import ...` offers on the prediction path is
offered here under the same names (models, NMS entry points, geometry, ray sets, `edt_prob`, the ImageJ ROI export, the label-image helpers).
"""
__version__ = "0.1.0"

# name -> submodule, resolved lazily so that `import stardist_amd` (and the C-ABI symbol checks) work without torch
_LAZY = {
    "StarDist2D": "models", "StarDist3D": "models", "Config2D": "models", "Config3D": "models",
    "non_maximum_suppression": "nms", "non_maximum_suppression_3d": "nms", "non_maximum_suppression_3d_sparse": "nms",
    "edt_prob": "utils", "export_imagej_rois": "utils", "gputools_available": "utils",
    "fill_label_holes": "utils", "sample_points": "utils", "calculate_extents": "utils",
    "star_dist": "geometry", "polygons_to_label": "geometry", "relabel_image_stardist": "geometry", "ray_angles": "geometry",
    "dist_to_coord": "geometry", "star_dist3D": "geometry", "polyhedron_to_label": "geometry", "relabel_image_stardist3D": "geometry",
    "rays_from_json": "rays3d", "Rays_Cartesian": "rays3d", "Rays_SubDivide": "rays3d", "Rays_Tetra": "rays3d", "Rays_Octo": "rays3d",
    "Rays_GoldenSpiral": "rays3d", "Rays_Explicit": "rays3d",
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        return getattr(importlib.import_module("." + _LAZY[name], __name__), name)
    raise AttributeError(name)


def __dir__():
    return sorted(list(globals()) + list(_LAZY))
