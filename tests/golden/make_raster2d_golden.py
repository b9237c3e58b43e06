"""Golden vectors for the 2D label rasteriser, produced by the REFERENCE's own code (stardist/geometry/geom2d.py:
ray_angles, dist_to_coord, polygons_to_label_coord, polygons_to_label) executed with the real scikit-image
`skimage.draw.polygon` (the third-party routine whose arithmetic is not under /root/reference).

The default interpreter of the build image has no scikit-image; an Anaconda python that has it ships in the same image:

    /opt/conda/bin/python3.9 tests/golden/make_raster2d_golden.py        # scikit-image 0.18.3, numpy 1.26.4

The reference functions are taken verbatim from the reference file at run time (ast -> exec; nothing is copied into this
repo); only their numpy inputs/outputs are stored in tests/golden/raster2d_reference.npz."""
import ast
import os
import sys

import numpy as np
import skimage
from skimage.draw import polygon

REF = "/root/reference/stardist/geometry/geom2d.py"
HERE = os.path.dirname(os.path.abspath(__file__))

src = open(REF).read()
tree = ast.parse(src)
want = {"ray_angles", "dist_to_coord", "polygons_to_label_coord", "polygons_to_label"}
ns = {"np": np, "polygon": polygon, "_check_label_array": lambda *a, **k: None}
for node in tree.body:
    if isinstance(node, ast.FunctionDef) and node.name in want:
        exec(compile(ast.Module([node], []), REF, "exec"), ns)
assert want <= set(ns), sorted(want - set(ns))

out = {"skimage_version": np.array(skimage.__version__), "numpy_version": np.array(np.__version__)}
rng = np.random.RandomState(7)


def star_case(name, shape, n, n_rays, radius, noise, margin):
    pts = np.stack([rng.uniform(-margin, shape[0] + margin, n), rng.uniform(-margin, shape[1] + margin, n)], 1).astype(np.float32)
    dist = (radius * (1 + noise * rng.uniform(-1, 1, (n, n_rays)))).astype(np.float32)
    prob = rng.uniform(0, 1, n).astype(np.float32)
    lbl = ns["polygons_to_label"](dist, pts, shape, prob=prob, thr=0.2)
    out[name + "_dist"], out[name + "_points"], out[name + "_prob"] = dist, pts, prob
    out[name + "_shape"], out[name + "_labels"] = np.array(shape), lbl.astype(np.int32)


star_case("stars32", (120, 150), 160, 32, 9.0, 0.3, 6)            # overlapping, partly outside the image
star_case("stars8_small", (64, 64), 120, 8, 1.6, 0.6, 2)          # tiny polygons around single pixels
star_case("stars64_big", (100, 90), 12, 64, 30.0, 0.2, 10)        # large polygons clipped on all sides
star_case("stars5_int", (48, 56), 60, 5, 6.0, 0.0, 0)             # few rays

# explicit coordinate cases through polygons_to_label_coord: vertices / edges exactly on pixel centres, degenerate shapes
coords = []
coords.append(np.array([[2, 2, 8, 8], [3, 9, 9, 3]], np.float32))                 # axis-aligned rectangle on integer coordinates
coords.append(np.array([[10, 14, 18, 14], [10, 6, 10, 14]], np.float32))          # diamond with vertices on pixel centres
coords.append(np.array([[20.5, 20.5, 26.5, 26.5], [3.5, 9.5, 9.5, 3.5]], np.float32))   # rectangle on half-integers
coords.append(np.array([[30, 30, 30, 30], [2, 6, 10, 6]], np.float32))            # zero-area (collinear)
coords.append(np.array([[34, 38, 34, 38], [2, 2, 8, 8]], np.float32))             # self-intersecting bow-tie
coords.append(np.array([[-3, -3, 4, 4], [-2, 5, 5, -2]], np.float32))             # partly outside (negative coordinates)
coords.append(np.array([[40, 47.9, 47.9, 40], [50, 50, 70, 70]], np.float32))     # beyond the right/bottom border
coords.append(np.array([[12.25, 12.25, 12.75, 12.75], [20.25, 20.75, 20.75, 20.25]], np.float32))  # sub-pixel square
coord = np.stack(coords)
out["explicit_coord"] = coord
out["explicit_shape"] = np.array((48, 64))
out["explicit_labels"] = ns["polygons_to_label_coord"](coord, (48, 64)).astype(np.int32)
# per-polygon masks (rr, cc) of the raw skimage routine for the explicit cases
for i, c in enumerate(coord):
    rr, cc = polygon(*c, (48, 64))
    m = np.zeros((48, 64), bool); m[rr, cc] = True
    out["explicit_mask%d" % i] = m

np.savez_compressed(os.path.join(HERE, "raster2d_reference.npz"), **out)
print("wrote", os.path.join(HERE, "raster2d_reference.npz"), "with", len(out), "arrays; skimage", skimage.__version__)
