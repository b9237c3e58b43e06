"""Golden outputs of the reference's BlockND.cover + crop_context + filter_objects (stardist/big.py:282-425) on random label
images, produced with the reference's own classes (exec'd from the source text; csbdeep helpers stubbed) and the REAL
skimage.measure.regionprops.  Needs scikit-image:

    /opt/conda/bin/python3.9 tests/golden/make_big_filter_golden.py     -> tests/golden/big_filter.npz"""
import math
import os
import warnings
from itertools import product

import numpy as np
from skimage.measure import regionprops

HERE = os.path.dirname(os.path.abspath(__file__))
src = open("/root/reference/stardist/big.py").read()
a = src.index("OBJECT_KEYS = "); b = src.index("class Polygon:")


def _raise(e):
    raise e


def axes_check_and_normalize(axes, length=None, disallowed=None, return_allowed=False):
    axes = str(axes).upper()
    assert length is None or len(axes) == length
    return axes


def axes_dict(axes):
    return {a: (axes.find(a) if a in axes else None) for a in "STCZYX"}


ns = {"np": np, "math": math, "warnings": warnings, "product": product, "regionprops": regionprops, "_raise": _raise,
      "axes_check_and_normalize": axes_check_and_normalize, "axes_dict": axes_dict, "tqdm": lambda x, **k: x}
g0 = src.index("def _grid_divisible"); g1 = src.index("# def render_polygons")
exec(src[g0:g1] + "\n" + src[a:b], ns)
BlockND = ns["BlockND"]


def discs(shape, n, r, seed):
    rng = np.random.RandomState(seed)
    lbl = np.zeros(shape, np.int32)
    grids = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    k = 0
    for _ in range(n):
        c = [rng.uniform(0, s) for s in shape]; rad = rng.uniform(*r)
        m = sum((g - ci) ** 2 for g, ci in zip(grids, c)) <= rad * rad
        if (lbl[m] == 0).all() and m.any():
            k += 1; lbl[m] = k
    return lbl


out = {}
cases = {"2d": ((160, 200), "YX", (64, 64), (16, 16), (8, 8), 1, 70, (2, 5)),
         "2dg": ((150, 132), "YX", (48, 64), (16, 16), (4, 8), 2, 60, (2, 5)),
         "3d": ((40, 80, 72), "ZYX", (32, 40, 40), (12, 12, 12), (2, 2, 2), 1, 50, (1.5, 4))}   # covers also used by tests/test_cpu_big.py
for name, (shape, axes, bs, mo, ctx, grid, n, r) in cases.items():
    gt = discs(shape, n, r, seed=len(name))
    blocks = BlockND.cover(shape, axes, bs, mo, ctx, grid)
    out[name + "_gt"] = gt
    out[name + "_args"] = np.array([bs, mo, ctx, (grid,) * len(shape)])
    out[name + "_nblocks"] = np.array(len(blocks))
    for bi, block in enumerate(blocks):
        # what predict_instances would return for the block: sequential labels of the visible objects + their centres
        sub = block.read(gt, axes=axes)
        ids = np.unique(sub); ids = ids[ids > 0]
        lab = np.zeros_like(sub)
        for j, v in enumerate(ids, 1):
            lab[sub == v] = j
        pts = np.array([np.mean(np.nonzero(lab == j), axis=1) for j in range(1, len(ids) + 1)]).reshape(len(ids), len(shape))
        polys = dict(points=pts, prob=np.linspace(1, 0.5, len(ids)))
        labc = block.crop_context(lab, axes=axes)
        try:
            lf, pf = block.filter_objects(labc, polys, axes=axes)
            out["%s_b%d_labels" % (name, bi)] = lf
            out["%s_b%d_points" % (name, bi)] = pf["points"]
            out["%s_b%d_prob" % (name, bi)] = pf["prob"]
        except RuntimeError:
            out["%s_b%d_error" % (name, bi)] = np.array(1)
np.savez_compressed(os.path.join(HERE, "big_filter.npz"), **out)
print("wrote big_filter.npz:", {k: int(out[k + "_nblocks"]) for k in cases}, "errors:", [k for k in out if k.endswith("_error")])
