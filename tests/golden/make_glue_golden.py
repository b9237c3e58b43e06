"""Golden vectors for the pure-numpy glue of the prediction path, produced by the reference's own functions
(taken from the reference files at run time via ast -> exec; they only need numpy):
  stardist/matching.py  relabel_sequential  (:319-408)
  stardist/nms.py       _ind_prob_thresh    (:6-17)
usage: python tests/golden/make_glue_golden.py   -> tests/golden/glue_reference.npz"""
import ast
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def grab(path, names, ns):
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module([node], []), path, "exec"), ns)
    assert set(names) <= set(ns), sorted(set(names) - set(ns))
    return ns


def _raise(e):
    raise e


ns = {"np": np, "_raise": _raise}
grab("/root/reference/stardist/matching.py", ["relabel_sequential"], ns)
grab("/root/reference/stardist/nms.py", ["_ind_prob_thresh"], ns)

out = {}
rng = np.random.RandomState(3)
cases = {
    "a": (rng.randint(0, 40, (17, 23)).astype(np.int32) * (rng.rand(17, 23) > 0.3), 1),
    "b": (np.array([[0, 7, 7, 0], [3, 3, 99, 99], [0, 0, 5, 5]], np.int32), 1),
    "c": (rng.randint(0, 9, (5, 6, 7)).astype(np.int64) * 1000, 5),
    "d": (np.zeros((4, 4), np.int32), 1),
    "e": (np.arange(1, 13, dtype=np.int32).reshape(3, 4), 3),
}
for k, (lab, off) in cases.items():
    lab = np.asarray(lab)
    relab, fw, inv = ns["relabel_sequential"](lab.copy(), off)
    out["relabel_%s_in" % k], out["relabel_%s_offset" % k] = lab, np.array(off)
    out["relabel_%s_out" % k], out["relabel_%s_fw" % k], out["relabel_%s_inv" % k] = np.asarray(relab), np.asarray(fw), np.asarray(inv)
for k, (shape, thr, b) in {"2d": ((9, 11), 0.4, 2), "3d": ((6, 7, 8), 0.6, 1), "b0": ((5, 5), 0.5, None),
                           "bt": ((8, 9), 0.3, ((1, 2), (0, 3)))}.items():
    prob = rng.rand(*shape).astype(np.float32)
    out["thresh_%s_prob" % k] = prob
    out["thresh_%s_args" % k] = np.array([thr], np.float64)
    out["thresh_%s_b" % k] = np.array(-1 if b is None else b)
    out["thresh_%s_mask" % k] = ns["_ind_prob_thresh"](prob, thr, b=b)
np.savez_compressed(os.path.join(HERE, "glue_reference.npz"), **out)
print("wrote glue_reference.npz with", len(out), "arrays")
