"""Seeded synthetic inputs of SURVEY.md section 8(d) (TEST INFRASTRUCTURE + bench inputs).

Pure numpy; shared by the oracle and the GPU path so both see identical bytes.
The generator definitions are normative (SURVEY.md "Generator specifications").
"""
import numpy as np


def s2d_uniform(H, W, n_rays=32, radius=10, noise=0.1, prob_thresh=0.9, b=2, seed=42, dense=False):
    """S2D-uniform, after the reference's tests/test_nms2D.py:9-15 (create_random_data).

    Returns (dist (N,R) f32, points (N,2) f32, prob (N,) f32) sorted by prob descending with
    the reference's ``argsort[::-1]`` (stardist/nms.py:114), or the dense (dist, prob) maps.
    """
    rng = np.random.RandomState(seed)
    dist = (radius * np.ones((H, W, n_rays))).astype(np.float32) * \
        (1 + noise * rng.uniform(-1, 1, (H, W, n_rays))).astype(np.float32)
    prob = rng.uniform(0, 1, (H, W)).astype(np.float32)
    if dense:
        return dist, prob
    mask = prob > np.float32(prob_thresh)
    if b:
        m2 = np.zeros_like(mask)
        m2[b:-b, b:-b] = True
        mask &= m2
    pts = np.stack(np.where(mask), 1)
    d = dist[mask]
    s = prob[mask]
    ind = np.argsort(s)[::-1]
    return np.ascontiguousarray(d[ind]), np.ascontiguousarray(pts[ind].astype(np.float32)), np.ascontiguousarray(s[ind])


def s3d_nuclei(N, rays_vertices, spacing=24, R=(7, 10), rc=3, seed=0):
    """S3D-nuclei: spheres on a jittered lattice, candidates within rc voxels of each centre.

    rays_vertices: (n_rays,3) unit vectors (z,y,x) of the ray set (Rays_GoldenSpiral(96)).
    Returns (dist (N,R) f32, points (N,3) f32, prob (N,) f32) sorted by prob descending.
    """
    V = np.asarray(rays_vertices, np.float64)
    rng = np.random.RandomState(seed)
    g = np.arange(spacing // 2, N, spacing)
    C = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
    C += rng.uniform(-4, 4, C.shape)
    Rs = rng.uniform(R[0], R[1], len(C))
    o = np.arange(-rc, rc + 1)
    off = np.stack(np.meshgrid(o, o, o, indexing="ij"), -1).reshape(-1, 3)
    off = off[(off ** 2).sum(1) <= rc * rc]
    P, D, S = [], [], []
    for c, Rr in zip(C, Rs):
        p = np.round(c).astype(np.int64) + off
        p = p[np.all((p >= 2) & (p < N - 2), 1)]
        q = p - c
        bq = q @ V.T
        t = -bq + np.sqrt(bq * bq - ((q * q).sum(1)[:, None] - Rr * Rr))
        pr = 0.5 * np.clip(1 - np.sqrt((q * q).sum(1)) / Rr, 0, 1) + 0.5 * rng.uniform(0.9, 1, len(p))
        P.append(p); D.append(t); S.append(pr)
    P = np.concatenate(P).astype(np.float32)
    D = np.maximum(np.concatenate(D), 1e-3).astype(np.float32)
    S = np.concatenate(S).astype(np.float32)
    ind = np.argsort(S)[::-1]
    return np.ascontiguousarray(D[ind]), np.ascontiguousarray(P[ind]), np.ascontiguousarray(S[ind]), len(C)


def s2d_nuclei_labels(H, W, spacing=32, R=(8, 14), seed=0):
    """Label image of discs on a jittered lattice (S2D-nuclei). uint16, ids 1..K."""
    rng = np.random.RandomState(seed)
    g = np.arange(spacing // 2, max(H, W), spacing)
    cy, cx = np.meshgrid(g[g < H], g[g < W], indexing="ij")
    C = np.stack([cy.ravel(), cx.ravel()], 1).astype(np.float64)
    C += rng.uniform(-6, 6, C.shape)
    Rs = rng.uniform(R[0], R[1], len(C))
    lbl = np.zeros((H, W), np.uint16)
    for k, ((y, x), r) in enumerate(zip(C, Rs)):
        y0, y1 = max(0, int(y - r - 1)), min(H, int(y + r + 2))
        x0, x1 = max(0, int(x - r - 1)), min(W, int(x + r + 2))
        yy, xx = np.mgrid[y0:y1, x0:x1]
        m = (yy - y) ** 2 + (xx - x) ** 2 <= r * r
        sub = lbl[y0:y1, x0:x1]
        sub[m & (sub == 0)] = (k % 65535) + 1
    return lbl, C, Rs


def s2d_nuclei_image(H, W, seed=0, channels=1):
    """Synthetic 'fluo' image for the U-Net leg: blurred disc mask + gaussian noise, float32 in ~[0,1]."""
    lbl, _, _ = s2d_nuclei_labels(H, W, seed=seed)
    rng = np.random.RandomState(seed + 1)
    img = (lbl > 0).astype(np.float32)
    # separable 5-tap box blur (no scipy dependency on the GPU box path)
    k = np.ones(5, np.float32) / 5
    img = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), 0, img)
    img = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), 1, img)
    img = img + rng.normal(0, 0.05, img.shape).astype(np.float32)
    if channels > 1:
        img = np.stack([img * (0.6 + 0.2 * c) for c in range(channels)], -1)
    return img.astype(np.float32)


def s3d_nuclei_image(N, spacing=24, R=(7, 10), seed=0):
    """Synthetic 3D 'fluo' volume for the U-Net leg: spheres on a jittered lattice + gaussian noise, float32."""
    rng = np.random.RandomState(seed)
    g = np.arange(spacing // 2, N, spacing)
    C = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
    C += rng.uniform(-4, 4, C.shape)
    Rs = rng.uniform(R[0], R[1], len(C))
    vol = np.zeros((N, N, N), np.float32)
    for c, r in zip(C, Rs):
        lo = np.maximum(0, np.floor(c - r - 1).astype(int)); hi = np.minimum(N, np.ceil(c + r + 2).astype(int))
        zz, yy, xx = np.mgrid[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
        m = (zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2 <= r * r
        vol[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]][m] = 1.0
    vol += rng.normal(0, 0.05, vol.shape).astype(np.float32)
    return vol


def lattice_candidates_2d(R, family, seed, n=1800, shape=(96, 96)):
    """Candidates whose geometry sits EXACTLY on the pixel lattice (tests/golden/make_lattice_golden.py, tests/test_gpu_lattice.py):
    integer centres and `const` = regular R-gons of integer radius (R = 4: diamonds with lattice vertices), `int` = every ray its own
    integer length, `half` = half-integer lengths.  Many shapes coincide or are one-pixel shifts of each other: coincident edges and
    vertices for Clipper, polygon vertices on pixel centres for the rasteriser.  Sorted by score, best first."""
    rng = np.random.RandomState(1000 * R + seed)
    pts = np.stack([rng.randint(2, shape[0] - 2, n), rng.randint(2, shape[1] - 2, n)], 1).astype(np.float32)
    if family == "const":
        d = np.repeat(rng.randint(2, 9, (n, 1)), R, 1)
    elif family == "int":
        d = rng.randint(2, 9, (n, R))
    elif family == "half":
        d = rng.randint(4, 18, (n, R)) * 0.5
    else:
        raise ValueError(family)
    s = rng.uniform(0, 1, n).astype(np.float32)
    ind = np.argsort(s, kind="stable")[::-1]
    return np.ascontiguousarray(d[ind].astype(np.float32)), np.ascontiguousarray(pts[ind]), np.ascontiguousarray(s[ind])


def lattice_candidates_3d(n_rays, family, n=500, size=48):
    """3D counterpart: integer centres, one integer radius per polyhedron (`const`) or per ray (`int`)."""
    rng = np.random.RandomState(n_rays * 7 + (family == "int"))
    pts = np.stack([rng.randint(4, size - 4, n) for _ in range(3)], 1).astype(np.float32)
    d = (np.repeat(rng.randint(3, 8, (n, 1)), n_rays, 1) if family == "const" else rng.randint(3, 8, (n, n_rays))).astype(np.float32)
    s = rng.uniform(0, 1, n).astype(np.float32)
    ind = np.argsort(s, kind="stable")[::-1]
    return np.ascontiguousarray(d[ind]), np.ascontiguousarray(pts[ind]), np.ascontiguousarray(s[ind])
