"""Golden label images of `relabel_image_stardist` (stardist/geometry/geom2d.py:200-211) and `relabel_image_stardist3D`
(stardist/geometry/geom3d.py:201-217), produced by the REFERENCE's own functions -- taken from the reference files at run time
(ast -> exec; nothing is copied into this repo) -- with the pieces they call supplied by the real thing:

  * `star_dist` / `star_dist3D`, `c_polyhedron_to_label`: the compiled reference natives (oracle/_ref, this interpreter);
  * `skimage.measure.regionprops`, `skimage.draw.polygon`: the real scikit-image 0.18.3, which only the image's Anaconda interpreter
    has -- the script re-runs itself there for those steps (`--conda-stage`), handing arrays over through a temporary .npz.

    python tests/golden/make_relabel_golden.py            # build container (needs /root/reference, oracle/_ref, /opt/conda/bin/python3.9)

Inputs: the label images of the reference's own tests (tests/utils.py circle_image as used by tests/test_stardist2D.py:46-56 and
tests/test_stardist3D.py:55-66) and two multi-object images with non-sequential ids, a region that touches the border and one that is
not star-convex.  Stored: inputs, the region centroids regionprops reports, and the relabelled images (tests/golden/relabel_reference.npz)."""
import ast
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CONDA = "/opt/conda/bin/python3.9"
GEOM2D = "/root/reference/stardist/geometry/geom2d.py"
GEOM3D = "/root/reference/stardist/geometry/geom3d.py"


def ref_functions(path, want, ns):
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in want:
            exec(compile(ast.Module([node], []), path, "exec"), ns)
    assert set(want) <= set(ns), sorted(set(want) - set(ns))
    return ns


def circle_image(shape, radius, eps):
    """the reference tests' generator (tests/utils.py:52-65), restated: an axis-scaled ball around the centre pixel"""
    xs = tuple(np.arange(s) - s // 2 for s in shape)
    Xs = np.meshgrid(*xs, indexing="ij")
    R = np.sqrt(np.sum([X ** 2 / e ** 2 for X, e in zip(Xs, eps)], axis=0))
    return (R < radius).astype(np.uint16)


def multi2d():
    lbl = np.zeros((96, 120), np.uint16)
    yy, xx = np.mgrid[:96, :120]
    for lab, (cy, cx, ry, rx) in zip((3, 4, 7, 12, 13, 20, 21, 30), [(14, 16, 9, 12), (20, 60, 12, 7), (16, 100, 10, 10), (52, 24, 13, 13),
                                                                       (56, 70, 8, 15), (50, 112, 14, 11), (84, 8, 9, 9), (82, 52, 7, 7)]):
        lbl[((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1] = lab
    lbl[70:92, 86:94] = 31; lbl[84:92, 86:112] = 31                 # an L: not star-convex from its centroid's pixel
    return lbl


def multi3d():
    lbl = np.zeros((40, 48, 56), np.uint16)
    zz, yy, xx = np.mgrid[:40, :48, :56]
    for lab, (c, r) in zip((2, 5, 6, 11, 40, 41), [((10, 12, 12), (6, 8, 8)), ((10, 34, 40), (7, 6, 10)), ((28, 12, 42), (8, 8, 7)),
                                                  ((30, 36, 14), (6, 9, 9)), ((20, 24, 27), (5, 5, 5)), ((35, 44, 50), (7, 6, 8))]):
        lbl[((zz - c[0]) / r[0]) ** 2 + ((yy - c[1]) / r[1]) ** 2 + ((xx - c[2]) / r[2]) ** 2 < 1] = lab
    return lbl


def cases2d():
    return [("circle_iso", circle_image((32, 32), 8, (1, 1)), 32), ("circle_iso", circle_image((32, 32), 8, (1, 1)), 64),
            ("circle_aniso", circle_image((32, 32), 8, (.4, 1.3)), 32), ("circle_aniso", circle_image((32, 32), 8, (.4, 1.3)), 64),
            ("multi", multi2d(), 32), ("empty", np.zeros((20, 24), np.uint16), 32)]


def cases3d():
    return [("ball_iso", circle_image((32, 32, 32), 8, (1, 1, 1)), 64, (1, 1, 1)), ("ball_iso", circle_image((32, 32, 32), 8, (1, 1, 1)), 128, (1, 1, 1)),
            ("ball_aniso", circle_image((32, 32, 32), 8, (.4, 1.3, .7)), 64, (.4, 1.3, .7)), ("multi", multi3d(), 96, (1, 1, 1))]


def conda_stage(tmp_in, tmp_out):
    """under the Anaconda interpreter: the reference's 2D function end to end (star_dist answers with the compiled reference's array),
    and regionprops' (label, centroid) for the 3D inputs"""
    import skimage
    from skimage.draw import polygon
    from skimage.measure import regionprops
    G = np.load(tmp_in)
    out = {"skimage_version": np.array(skimage.__version__)}
    state = {}
    ns = {"np": np, "polygon": polygon, "regionprops": regionprops, "_check_label_array": lambda *a, **k: True,
          "star_dist": lambda lbl, n_rays, **kw: state["dist"]}
    ref_functions(GEOM2D, {"relabel_image_stardist", "polygons_to_label", "polygons_to_label_coord", "dist_to_coord", "ray_angles"}, ns)
    k = 0
    while "in2d_%d" % k in G:
        lbl, state["dist"] = G["in2d_%d" % k], G["dist2d_%d" % k]
        out["out2d_%d" % k] = ns["relabel_image_stardist"](lbl, int(G["rays2d_%d" % k])).astype(np.int32)
        regs = regionprops(lbl)
        out["cen2d_%d" % k] = np.array([r.centroid for r in regs], np.float64).reshape(len(regs), 2)
        out["lab2d_%d" % k] = np.array([r.label for r in regs], np.int64)
        k += 1
    k = 0
    while "in3d_%d" % k in G:
        regs = regionprops(G["in3d_%d" % k])
        out["cen3d_%d" % k] = np.array([r.centroid for r in regs], np.float64).reshape(len(regs), 3)
        out["lab3d_%d" % k] = np.array([r.label for r in regs], np.int64)
        k += 1
    np.savez(tmp_out, **out)


def main():
    sys.path.insert(0, ROOT)
    from oracle import port, ref
    from stardist_amd.rays3d import Rays_GoldenSpiral            # vertices / faces pinned to the reference's own (tests/golden/rays_*.npz)
    m2, m3 = ref.stardist2d(), ref.stardist3d()
    ref.set_threads(1)
    c2, c3 = cases2d(), cases3d()
    hand = {}
    for k, (name, lbl, R) in enumerate(c2):
        hand["in2d_%d" % k], hand["rays2d_%d" % k] = lbl, np.array(R)
        hand["dist2d_%d" % k] = m2.c_star_dist(lbl, np.int32(R), np.int32(1), np.int32(1))      # geom2d.py:29-31 _cpp_star_dist
    for k, (name, lbl, R, eps) in enumerate(c3):
        hand["in3d_%d" % k] = lbl
    with tempfile.TemporaryDirectory() as td:
        a, b = os.path.join(td, "in.npz"), os.path.join(td, "out.npz")
        np.savez(a, **hand)
        subprocess.run([CONDA, os.path.abspath(__file__), "--conda-stage", a, b], check=True)
        C = dict(np.load(b))
    out = {"skimage_version": C["skimage_version"], "n2d": np.array(len(c2)), "n3d": np.array(len(c3))}
    for k, (name, lbl, R) in enumerate(c2):
        out["name2d_%d" % k], out["in2d_%d" % k], out["rays2d_%d" % k] = np.array(name), lbl, np.array(R)
        out["out2d_%d" % k], out["cen2d_%d" % k], out["lab2d_%d" % k] = C["out2d_%d" % k], C["cen2d_%d" % k], C["lab2d_%d" % k]

    # 3D: the reference's function in this interpreter; regionprops answers with what the real one reported above
    class Reg(object):
        def __init__(self, label, centroid): self.label, self.centroid = int(label), tuple(centroid)
    state = {}
    ns = {"np": np, "regionprops": lambda lbl: state["regs"], "_check_label_array": lambda *a, **k: True,
          "c_polyhedron_to_label": m3.c_polyhedron_to_label,
          "star_dist3D": lambda lbl, rays, **kw: port.star_dist3D(lbl, rays.vertices, grid=(1, 1, 1))}   # geom3d.py:16-24 _cpp_star_dist3D
    ref_functions(GEOM3D, {"relabel_image_stardist3D", "polyhedron_to_label"}, ns)
    for k, (name, lbl, R, eps) in enumerate(c3):
        rays = Rays_GoldenSpiral(R, anisotropy=tuple(1.0 / np.array(eps)))
        state["regs"] = [Reg(l, c) for l, c in zip(C["lab3d_%d" % k], C["cen3d_%d" % k])]
        res = ns["relabel_image_stardist3D"](lbl, rays)
        out["name3d_%d" % k], out["in3d_%d" % k], out["rays3d_%d" % k], out["eps3d_%d" % k] = np.array(name), lbl, np.array(R), np.array(eps, np.float64)
        out["out3d_%d" % k], out["cen3d_%d" % k], out["lab3d_%d" % k] = res.astype(np.int32), C["cen3d_%d" % k], C["lab3d_%d" % k]
        # the reference's own consistency test (tests/test_stardist3D.py:55-66) on what it just produced
        if name.startswith("ball"):
            err = 1 - np.count_nonzero((lbl > 0) & (res > 0)) / np.count_nonzero(lbl > 0)
            assert err < 1e-1, (name, R, err)
    for k, (name, lbl, R) in enumerate(c2):
        if name.startswith("circle"):                                # tests/test_stardist2D.py:46-56
            err = 1 - np.count_nonzero((lbl > 0) & (out["out2d_%d" % k] > 0)) / np.count_nonzero(lbl > 0)
            assert err < 1e-1, (name, R, err)
    np.savez_compressed(os.path.join(HERE, "relabel_reference.npz"), **out)
    print("wrote relabel_reference.npz:", len(c2), "2D cases,", len(c3), "3D cases; skimage", str(C["skimage_version"]))


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "--conda-stage":
        conda_stage(sys.argv[2], sys.argv[3])
    else:
        main()
