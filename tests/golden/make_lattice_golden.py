"""Golden answers of the COMPILED REFERENCE (oracle/_ref) -- and, for the 2D rasteriser, of the reference's own Python loop on the real
scikit-image -- on LATTICE-ALIGNED inputs (oracle/synth.py lattice_candidates_2d / _3d): integer centres, integer or half-integer ray
lengths, few rays; coincident and one-pixel-shifted shapes.  These are the tie cases random float inputs never produce.

    python tests/golden/make_lattice_golden.py       # build container: /root/reference, oracle/_ref, /opt/conda/bin/python3.9 (scikit-image 0.18.3)

Stored (tests/golden/lattice_reference.npz): keep flags of c_non_max_suppression_inds 2D / 3D (bit-packed), label images of
polygons_to_label (stardist/geometry/geom2d.py:169-197, real skimage.draw.polygon) and c_polyhedron_to_label (modes full and kernel)."""
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
CASES2D = [(R, fam, 0) for R in (4, 8, 16, 32) for fam in ("const", "int", "half")]
THR2D = (0.3, 0.5)
RAYS3D = ("cartesian_8_5", "octo", "golden32", "golden32_aniso")
THR3D = (0.2, 0.4)
SHAPE2D, SIZE3D = (96, 96), 48


def rays_of(name):
    from stardist_amd.rays3d import Rays_Cartesian, Rays_GoldenSpiral, Rays_Octo      # vertices / faces pinned to the reference's own
    return {"cartesian_8_5": lambda: Rays_Cartesian(8, 5), "octo": Rays_Octo, "golden32": lambda: Rays_GoldenSpiral(32),
            "golden32_aniso": lambda: Rays_GoldenSpiral(32, anisotropy=(2, 1, 1))}[name]()


def conda_stage(a, b):
    """the reference's own polygons_to_label on the real scikit-image"""
    from skimage.draw import polygon
    ns = {"np": np, "polygon": polygon, "_check_label_array": lambda *x, **k: True}
    tree = ast.parse(open(GEOM2D).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in {"ray_angles", "dist_to_coord", "polygons_to_label_coord", "polygons_to_label"}:
            exec(compile(ast.Module([node], []), GEOM2D, "exec"), ns)
    G = np.load(a)
    out = {}
    k = 0
    while "d_%d" % k in G:
        out["lab_%d" % k] = ns["polygons_to_label"](G["d_%d" % k], G["p_%d" % k], tuple(G["shape"]), prob=G["s_%d" % k]).astype(np.int32)
        k += 1
    np.savez(b, **out)


def main():
    sys.path.insert(0, ROOT)
    from oracle import port, ref, synth
    m2, m3 = ref.stardist2d(), ref.stardist3d()
    ref.set_threads(1)
    out, hand = {}, {"shape": np.array(SHAPE2D)}
    for k, (R, fam, seed) in enumerate(CASES2D):
        d, p, s = synth.lattice_candidates_2d(R, fam, seed, shape=SHAPE2D)
        for thr in THR2D:
            keep = m2.c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(thr)).astype(bool)
            out["nms2d_%d_%s_%d_%.1f" % (R, fam, seed, thr)] = np.packbits(keep)
        keep = m2.c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(THR2D[0])).astype(bool)
        hand["d_%d" % k], hand["p_%d" % k], hand["s_%d" % k] = d[keep], p[keep], s[keep]
    with tempfile.TemporaryDirectory() as td:
        a, b = os.path.join(td, "in.npz"), os.path.join(td, "out.npz")
        np.savez(a, **hand)
        subprocess.run([CONDA, os.path.abspath(__file__), "--conda-stage", a, b], check=True)
        C = dict(np.load(b))
    for k, (R, fam, seed) in enumerate(CASES2D):
        lab = C["lab_%d" % k]
        mine = port.polygons_to_label(hand["d_%d" % k], hand["p_%d" % k], SHAPE2D, prob=hand["s_%d" % k])
        assert np.array_equal(lab, mine), ("the numpy restatement of the rasteriser differs from the real scikit-image", R, fam)
        assert lab.max() < 65536
        out["raster2d_%d_%s_%d" % (R, fam, seed)] = lab.astype(np.uint16)
    for name in RAYS3D:
        rays = rays_of(name)
        V, F = rays.vertices.astype(np.float32), rays.faces.astype(np.int32)
        for fam in ("const", "int"):
            d, p, s = synth.lattice_candidates_3d(len(V), fam, size=SIZE3D)
            for thr in THR3D:
                keep = m3.c_non_max_suppression_inds(d, p, V, F, s, 1, 1, 0, np.float32(thr)).astype(bool)
                out["nms3d_%s_%s_%.1f" % (name, fam, thr)] = np.packbits(keep)
            keep = m3.c_non_max_suppression_inds(d, p, V, F, s, 1, 1, 0, np.float32(THR3D[0])).astype(bool)
            lab = np.arange(1, keep.sum() + 1, dtype=np.int32)
            for mode, mname in ((0, "full"), (1, "kernel")):
                vol = m3.c_polyhedron_to_label(d[keep], p[keep], V, F, lab, mode, 0, 0, 0, (SIZE3D,) * 3)
                assert vol.max() < 65536
                out["raster3d_%s_%s_%s" % (name, fam, mname)] = vol.astype(np.uint16)
    np.savez_compressed(os.path.join(HERE, "lattice_reference.npz"), **out)
    print("wrote lattice_reference.npz:", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "lattice_reference.npz")), "bytes")


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "--conda-stage":
        conda_stage(sys.argv[2], sys.argv[3])
    else:
        main()
