"""More lattice goldens of the COMPILED REFERENCE (oracle/_ref) for the 3D natives, added in round 6 after the lattice sets had exposed
three defects: highly SYMMETRIC ray sets -- the plain octahedron (6 rays), its first subdivision (18), a
subdivided tetrahedron (34), the reference's default Rays_Cartesian (11 x 5) and GoldenSpiral(96) -- whose lattice copies share facet
planes bit for bit (coincident half-spaces) and put voxels exactly on faces.  Same candidates as make_lattice_golden.py
(oracle/synth.py lattice_candidates_3d), same thresholds, one OpenMP thread.

    python tests/golden/make_lattice_golden_more.py      # build container: /root/reference, oracle/_ref

Stored (tests/golden/lattice_reference_more.npz): keep flags of c_non_max_suppression_inds (bit-packed), label volumes of
c_polyhedron_to_label in modes full and kernel."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
RAYS3D_MORE = ("octo1", "octo2", "tetra3", "cartesian_11_5", "golden96")
THR3D = (0.2, 0.4)
SIZE3D = 48


def rays_of_more(name):
    from stardist_amd.rays3d import Rays_Cartesian, Rays_GoldenSpiral, Rays_Octo, Rays_Tetra
    return {"octo1": lambda: Rays_Octo(1), "octo2": lambda: Rays_Octo(2), "tetra3": lambda: Rays_Tetra(3),
            "cartesian_11_5": lambda: Rays_Cartesian(11, 5), "golden96": lambda: Rays_GoldenSpiral(96)}[name]()


def main():
    sys.path.insert(0, ROOT)
    from oracle import ref, synth
    m3 = ref.stardist3d()
    ref.set_threads(1)
    out = {}
    for name in RAYS3D_MORE:
        rays = rays_of_more(name)
        V, F = rays.vertices.astype(np.float32), rays.faces.astype(np.int32)
        for fam in ("const", "int"):
            d, p, s = synth.lattice_candidates_3d(len(V), fam, size=SIZE3D)
            for thr in THR3D:
                keep = m3.c_non_max_suppression_inds(d, p, V, F, s, 1, 1, 0, np.float32(thr)).astype(bool)
                out["nms3d_%s_%s_%.1f" % (name, fam, thr)] = np.packbits(keep)
                print(name, fam, thr, len(d), "->", int(keep.sum()), flush=True)
            keep = m3.c_non_max_suppression_inds(d, p, V, F, s, 1, 1, 0, np.float32(THR3D[0])).astype(bool)
            lab = np.arange(1, keep.sum() + 1, dtype=np.int32)
            for mode, mname in ((0, "full"), (1, "kernel")):
                vol = m3.c_polyhedron_to_label(d[keep], p[keep], V, F, lab, mode, 0, 0, 0, (SIZE3D,) * 3)
                assert vol.max() < 65536
                out["raster3d_%s_%s_%s" % (name, fam, mname)] = vol.astype(np.uint16)
    path = os.path.join(HERE, "lattice_reference_more.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays,", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
