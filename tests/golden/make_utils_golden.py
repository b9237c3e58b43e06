"""Golden vectors for the training-target / exporter helpers, produced by the reference's own functions (taken from
stardist/utils.py at run time via ast -> exec; they only need numpy + scipy):
  _edt_prob_scipy (:100-125), polyroi_bytearray (:195-251), export_imagej_rois (:254-268)
usage: python tests/golden/make_utils_golden.py   -> tests/golden/utils_reference.npz"""
import ast
import io
import os
import zipfile

import numpy as np
from scipy.ndimage import distance_transform_edt, find_objects

HERE = os.path.dirname(os.path.abspath(__file__))


def grab(path, names, ns):
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module([node], []), path, "exec"), ns)
    assert set(names) <= set(ns)
    return ns


import warnings
from pathlib import Path
from zipfile import ZIP_DEFLATED, ZipFile
ns = {"np": np, "distance_transform_edt": distance_transform_edt, "find_objects": find_objects, "warnings": warnings, "Path": Path,
      "ZipFile": ZipFile, "ZIP_DEFLATED": ZIP_DEFLATED}
grab("/root/reference/stardist/utils.py", ["_edt_prob_scipy", "polyroi_bytearray", "export_imagej_rois"], ns)

out = {}
rng = np.random.RandomState(5)
lab2 = np.zeros((40, 50), np.int32)
for k in range(1, 9):
    y, x = rng.randint(0, 34), rng.randint(0, 44)
    lab2[y:y + rng.randint(3, 9), x:x + rng.randint(3, 9)] = k
lab2[0:4, 0:5] = 9; lab2[36:40, 44:50] = 11                      # touching the borders, a missing id (10)
lab3 = np.zeros((12, 20, 22), np.int32)
for k in range(1, 6):
    z, y, x = rng.randint(0, 8), rng.randint(0, 14), rng.randint(0, 16)
    lab3[z:z + rng.randint(2, 5), y:y + rng.randint(3, 7), x:x + rng.randint(3, 7)] = k
out["edt_lab2"], out["edt_prob2"] = lab2, ns["_edt_prob_scipy"](lab2)
out["edt_lab3"], out["edt_prob3"] = lab3, ns["_edt_prob_scipy"](lab3, anisotropy=(2.0, 1.0, 1.0))
const = np.full((6, 7), 3, np.int32)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    out["edt_const"], out["edt_prob_const"] = const, ns["_edt_prob_scipy"](const)
polys = (rng.uniform(5, 60, (4, 2, 32))).astype(np.float32)
out["roi_polys"] = polys
out["roi_bytes_sub"] = np.frombuffer(bytes(ns["polyroi_bytearray"](polys[0][1], polys[0][0], pos=3, subpixel=True)), np.uint8)
out["roi_bytes_int"] = np.frombuffer(bytes(ns["polyroi_bytearray"](polys[1][1], polys[1][0], pos=None, subpixel=False)), np.uint8)
tmp = os.path.join(HERE, "_tmp_rois")
ns["export_imagej_rois"](tmp, [polys[:2], polys[2:]])
with zipfile.ZipFile(tmp + ".zip") as z:
    names = sorted(z.namelist())
    out["roi_zip_names"] = np.array(names)
    out["roi_zip_concat"] = np.frombuffer(b"".join(z.read(n) for n in names), np.uint8)
os.remove(tmp + ".zip")
np.savez_compressed(os.path.join(HERE, "utils_reference.npz"), **out)
print("written", {k: getattr(v, "shape", None) for k, v in out.items()})
