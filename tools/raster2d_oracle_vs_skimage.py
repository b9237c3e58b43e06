"""The numpy restatement of skimage.draw.polygon the parity suite judges the 2D rasteriser with (oracle/port.py polygon) against the REAL
scikit-image 0.18.3 on a large random + lattice sample (CPU only; the real library lives in the image's Anaconda interpreter):

    python tools/raster2d_oracle_vs_skimage.py [n_polygons]      ->  profiles/r05_raster2d_oracle_vs_skimage.txt

Families: float vertices (star polygons of 3-64 rays, partly outside the image), integer vertices, half-integer vertices, vertices a
float32 ulp away from the lattice, degenerate (repeated / collinear vertices), self-intersecting (random vertex order)."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONDA = "/opt/conda/bin/python3.9"
SHAPE = (72, 88)


def polygons(n, seed=0):
    rng = np.random.RandomState(seed)
    out = []
    for i in range(n):
        fam = i % 6
        R = int(rng.choice([3, 4, 5, 8, 16, 32, 64]))
        c = np.array([rng.uniform(-6, SHAPE[0] + 6), rng.uniform(-6, SHAPE[1] + 6)])
        phi = np.linspace(0, 2 * np.pi, R, endpoint=False)
        rad = rng.uniform(1, 25) * (1 + rng.uniform(0, 0.6) * rng.uniform(-1, 1, R))
        v = np.stack([c[0] + rad * np.sin(phi), c[1] + rad * np.cos(phi)]).astype(np.float32)
        if fam == 1:
            v = np.round(v)
        elif fam == 2:
            v = np.round(2 * v) / 2
        elif fam == 3:
            v = np.nextafter(np.round(v).astype(np.float32), np.float32(rng.choice([-1e9, 1e9])))
        elif fam == 4:
            v = np.round(v); v[:, rng.randint(0, R)] = v[:, rng.randint(0, R)]
            if R > 4: v[:, 1] = (v[:, 0] + v[:, 2]) / 2
        elif fam == 5:
            v = np.round(2 * v[:, rng.permutation(R)]) / 2
        out.append(v.astype(np.float32))
    return out


def conda_stage(a, b):
    from skimage.draw import polygon
    import skimage
    G = np.load(a)
    n = int(G["n"])
    sums = np.zeros((n, 3), np.int64)
    for i in range(n):
        v = G["p%d" % i]
        rr, cc = polygon(v[0], v[1], SHAPE)
        sums[i] = (len(rr), int((rr.astype(np.int64) * 131 + cc).sum()), int(((rr.astype(np.int64) * SHAPE[1] + cc) ** 2 % 1000003).sum()))
    np.savez(b, sums=sums, version=np.array(skimage.__version__))


def main():
    sys.path.insert(0, ROOT)
    from oracle import port
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
    polys = polygons(n)
    t0 = time.time()
    mine = np.zeros((n, 3), np.int64)
    for i, v in enumerate(polys):
        rr, cc = port.polygon(v[0], v[1], SHAPE)
        mine[i] = (len(rr), int((rr.astype(np.int64) * 131 + cc).sum()), int(((rr.astype(np.int64) * SHAPE[1] + cc) ** 2 % 1000003).sum()))
    t1 = time.time()
    with tempfile.TemporaryDirectory() as td:
        a, b = os.path.join(td, "in.npz"), os.path.join(td, "out.npz")
        np.savez(a, n=np.array(n), **{"p%d" % i: v for i, v in enumerate(polys)})
        subprocess.run([CONDA, os.path.abspath(__file__), "--conda-stage", a, b], check=True)
        C = np.load(b)
        real, ver = C["sums"], str(C["version"])
    bad = np.flatnonzero((mine != real).any(1))
    fams = ["float vertices", "integer vertices", "half-integer vertices", "one float32 ulp off the lattice", "repeated / collinear vertices", "self-intersecting"]
    lines = ["oracle/port.py polygon (the restatement the parity suite judges the 2D rasteriser with) vs skimage.draw.polygon of scikit-image %s" % ver,
             "%d polygons on a %dx%d image (tools/raster2d_oracle_vs_skimage.py), pixel sets compared by (count, two checksums); restatement %.1f s" % (n, SHAPE[0], SHAPE[1], t1 - t0)]
    for f, name in enumerate(fams):
        idx = np.arange(f, n, 6)
        lines.append("  %-32s %6d polygons, %8d pixels painted, %d with a different pixel set" % (name, len(idx), int(real[idx, 0].sum()), int(np.isin(idx, bad).sum())))
    lines.append("TOTAL: %d of %d polygons differ" % (len(bad), n))
    for i in bad[:5]:
        lines.append("  differing polygon %d: %s" % (i, polys[i].tolist()))
    txt = "\n".join(lines)
    print(txt)
    open(os.path.join(ROOT, "profiles", "r05_raster2d_oracle_vs_skimage.txt"), "w").write(txt + "\n")


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "--conda-stage":
        conda_stage(sys.argv[2], sys.argv[3])
    else:
        main()
