"""Golden .obj text from the reference's own exporter (stardist/geometry/geom3d.py: dist_to_coord3D, export_to_obj_file3D,
taken from the reference file at run time).  usage: python tests/golden/make_export_golden.py -> tests/golden/export_obj_reference.npz"""
import ast
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/stardist/geometry/geom3d.py"
ns = {"np": np, "tqdm": lambda x, **k: x}
for node in ast.parse(open(REF).read()).body:
    if isinstance(node, ast.FunctionDef) and node.name in ("dist_to_coord3D", "export_to_obj_file3D"):
        exec(compile(ast.Module([node], []), REF, "exec"), ns)
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from stardist_amd.rays3d import Rays_GoldenSpiral      # vertices/faces pinned against the reference's rays (rays_reference.npz)

rays = Rays_GoldenSpiral(12, anisotropy=(2, 1, 1))
rng = np.random.RandomState(5)
polys = dict(dist=(3 + 2 * rng.rand(3, 12)).astype(np.float32), points=rng.uniform(5, 40, (3, 3)).astype(np.float32),
             rays_vertices=rays.vertices, rays_faces=rays.faces)
out = {k: np.asarray(v) for k, v in polys.items()}
for tag, kw in {"default": {}, "multi_uv": dict(single_mesh=False, uv_map=True, name="cell"), "scaled": dict(scale=(0.05, 0.2, 0.2))}.items():
    out["obj_" + tag] = np.array(ns["export_to_obj_file3D"](dict((k, np.array(v)) for k, v in polys.items()), **kw))
np.savez_compressed(os.path.join(HERE, "export_obj_reference.npz"), **out)
print("wrote export_obj_reference.npz")
