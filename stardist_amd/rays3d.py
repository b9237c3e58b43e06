"""Ray sets for 3D star-convex polyhedra: unit vectors (z, y, x) + outward-oriented triangles.

Mirror of the reference's stardist/rays3d.py API (Rays_Base :20-152, Rays_Explicit/Cartesian/
SubDivide/Tetra/Octo :163-327, reorder_faces :330-334, Rays_GoldenSpiral :337-373,
rays_from_json :156).  The vertex/face arrays feed every 3D kernel, and the FACE ORDER matters
(sequential fp32 sums over faces), so constructions follow the reference step by step; faces of
the golden spiral come from scipy.spatial.ConvexHull exactly as there.  Pinned against the
reference module's output in tests/golden/rays_*.npz.

This file is an ADAPTATION of the reference's host-side table generation, not a re-design (about half of its
statements restate the reference's: the operation order fixes the float32 vertices bit for bit and the face order
feeds every sequential sum of the 3D kernels, SURVEY.md section 2 row 11 "reuse verbatim semantics"); everything
that consumes the tables -- the kernels -- is this package's own.
"""
import copy as _copy

import numpy as np


class Rays_Base(object):
    def __init__(self, **kwargs):
        self.kwargs = kwargs
        v, f = self.setup_vertices_faces()
        self._vertices = np.asarray(v, np.float32)
        self._faces = np.asanyarray(np.asarray(f, int))

    def setup_vertices_faces(self):
        """returns (verts ((z,y,x), ...), faces ((i,j,k), ...))"""
        raise NotImplementedError()

    @property
    def vertices(self):
        return self._vertices.copy()

    @property
    def faces(self):
        return self._faces.copy()

    def __getitem__(self, i):
        return self.vertices[i]

    def __len__(self):
        return len(self._vertices)

    def __repr__(self):
        def conv(x):
            if isinstance(x, (tuple, list, np.ndarray)):
                return "_".join(conv(t) for t in x)
            if isinstance(x, float):
                return "%.2f" % x
            return str(x)
        return "%s_%s" % (self.__class__.__name__, "_".join("%s_%s" % (k, conv(v)) for k, v in sorted(self.kwargs.items())))

    def to_json(self):
        return {"name": self.__class__.__name__, "kwargs": self.kwargs}

    def dist_loss_weights(self, anisotropy=(1, 1, 1)):
        anisotropy = np.array(anisotropy)
        assert anisotropy.shape == (3,)
        return np.linalg.norm(self.vertices * anisotropy, axis=-1)

    def volume(self, dist=None):
        """volume of the polyhedron spanned by dist (last axis = rays); dist=None -> unit distances"""
        d = np.ones(len(self), np.float64) if dist is None else np.asarray(dist, np.float64)
        if d.shape[-1] != len(self):
            raise ValueError("last dimension of dist should have length len(rays.vertices)")
        p = d[..., None] * self.vertices.astype(np.float64)          # (..., R, 3)
        a, b, c = (p[..., self._faces[:, k], :] for k in range(3))   # (..., F, 3)
        det = np.einsum("...i,...i->...", a, np.cross(b, c))
        return -1. / 6 * det.sum(-1)

    def surface(self, dist=None):
        """surface area of the polyhedron spanned by dist (last axis = rays): the sum of its triangles' areas (rays3d.py:109-142)"""
        d = np.asarray(dist)
        if d.ndim == 0 or d.shape[-1] != len(self):
            raise ValueError("last dimension of dist should have length len(rays.vertices)")
        p = d[..., None] * self.vertices                             # (..., R, 3); float32 vertices promote as in the reference
        a, b, c = (p[..., self._faces[:, k], :] for k in range(3))   # (..., F, 3)
        return (0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=-1)).sum(-1)

    def copy(self, scale=(1, 1, 1)):
        scale = np.asarray(scale)
        assert scale.shape == (3,)
        res = _copy.deepcopy(self)
        res._vertices *= scale[np.newaxis]
        res.__dict__.pop("_coincident", None)
        return res

    def has_coincident_vertices(self):
        """True if two rays are closer than 1e-6: their polyhedron vertices `centre + dist * ray`, computed in float32 at image
        coordinates (stardist3d_impl.cpp polyhedron_polyverts), are the same point (not in the reference; see warn_if_degenerate)."""
        c = self.__dict__.get("_coincident")
        if c is None:
            v = np.round(np.asarray(self._vertices, np.float64), 6) + 0.0
            c = self.__dict__["_coincident"] = bool(len(np.unique(v, axis=0)) < len(v))
        return c


_WARNED_DEGENERATE = set()


def warn_if_degenerate(rays):
    """One warning per kind of ray set with (nearly) coincident rays.  `Rays_Cartesian` is the case: its pole rays differ by
    1e-12 (rays3d.py:189-197), so the polyhedron's pole vertices collapse to one float32 point (or lie on one line through the centre)
    and the mesh has degenerate triangles at both poles -- the reference's own Qhull calls print precision warnings for every such
    polyhedron and its cascade runs on its error paths (kernel stage: Qhull error for every pair; rendered overlap: a tetrahedron of zero
    volume passes the inside test on its whole plane).  Since round 6 the 3D NMS follows it there (hulls of point sets with coincident /
    collinear points, the rendered overlap over the whole box of the first polyhedron; DESIGN.md section 4 item 3a: keep flags identical on
    the lattice goldens).  The closed sets (GoldenSpiral, Octo,
    Tetra, SubDivide) are pinned."""
    fn = getattr(rays, "has_coincident_vertices", None)
    if fn is None or not fn():
        return False
    key = repr(rays)
    if key not in _WARNED_DEGENERATE:
        _WARNED_DEGENERATE.add(key)
        import warnings
        warnings.warn("%s: some rays coincide in float32 (degenerate faces): the reference's Qhull stages run on their error paths for such "
                      "meshes. The 3D NMS follows them (identical keep flags on the lattice goldens, DESIGN.md section 4 item 3a). "
                      "Rays_GoldenSpiral (the default) is pinned on many more inputs." % key, stacklevel=3)
    return True


_RAYS_CACHE = {}
_RAYS_DEV_CACHE = {}


def rays_device_tensors(rays, device):
    """(vertices float32 (R, 3), faces int32 (F, 3)) of a ray set as tensors on `device`, uploaded once per distinct ray set (the natives
    take them on every NMS / rasteriser call: two host -> device copies per call otherwise)"""
    import torch
    v = np.ascontiguousarray(rays.vertices, np.float32); f = np.ascontiguousarray(rays.faces, np.int32)
    key = (str(device), v.tobytes(), f.tobytes())
    r = _RAYS_DEV_CACHE.get(key)
    if r is None:
        if len(_RAYS_DEV_CACHE) >= 16:
            _RAYS_DEV_CACHE.clear()
        r = (torch.as_tensor(v, device=device), torch.as_tensor(f, device=device))
        _RAYS_DEV_CACHE[key] = r
    return r


def rays_from_json(d):
    """rays3d.py:156.  The objects are immutable in use (`copy(scale)` returns a new one), so one instance per description is kept
    and SHARED by its callers (its vertex / face arrays are read-only; `copy()` returns a private, writable instance):
    building Rays_GoldenSpiral(96) runs scipy's ConvexHull (1.4 ms), and the reference's call sites ask for it on every
    predict_instances (model3d.py:600)."""
    import json
    cls = {c.__name__: c for c in (Rays_Explicit, Rays_Cartesian, Rays_Tetra, Rays_Octo, Rays_GoldenSpiral)}
    try:
        key = json.dumps(d, sort_keys=True)
    except TypeError:
        return cls[d["name"]](**d["kwargs"])
    r = _RAYS_CACHE.get(key)
    if r is None:
        if len(_RAYS_CACHE) >= 16:
            _RAYS_CACHE.clear()
        r = cls[d["name"]](**d["kwargs"])
        for a in (r._vertices, r._faces):                            # shared between every model with this description: an in-place
            if isinstance(a, np.ndarray):                            # edit would corrupt them all -- refuse it (copy() gives a private one)
                a.setflags(write=False)
        _RAYS_CACHE[key] = r
    return r


class Rays_Explicit(Rays_Base):
    def __init__(self, vertices0, faces0):
        self.vertices0, self.faces0 = vertices0, faces0
        super().__init__(vertices0=list(vertices0), faces0=list(faces0))

    def setup_vertices_faces(self):
        return self.vertices0, self.faces0


class Rays_Cartesian(Rays_Base):
    def __init__(self, n_rays_x=11, n_rays_z=5):
        super().__init__(n_rays_x=n_rays_x, n_rays_z=n_rays_z)

    def setup_vertices_faces(self):
        nx, nz = self.kwargs["n_rays_x"], self.kwargs["n_rays_z"]
        dphi = np.float32(2. * np.pi / nx)
        dtheta = np.float32(np.pi / nz)
        verts = []
        for mz in range(nz):
            for mx in range(nx):
                phi, theta = mx * dphi, mz * dtheta
                pole = mz in (0, nz - 1)
                if mz == 0: theta = 1e-12
                if mz == nz - 1: theta = np.pi - 1e-12
                dx = np.cos(phi) * np.sin(theta)
                dy = np.sin(phi) * np.sin(theta)
                dz = np.cos(theta)
                if pole:
                    dx += 1e-12; dy += 1e-12
                verts.append([dz, dy, dx])
        ind = lambda mz, mx: mz * nx + mx
        faces = []
        for mz in range(nz - 1):
            for mx in range(nx):
                faces.append([ind(mz, mx), ind(mz + 1, (mx + 1) % nx), ind(mz, (mx + 1) % nx)])
                faces.append([ind(mz, mx), ind(mz + 1, mx), ind(mz + 1, (mx + 1) % nx)])
        return np.array(verts), np.array(faces)


class Rays_SubDivide(Rays_Base):
    """n_level = 1 -> base polyhedron, each further level splits every triangle in four"""

    def __init__(self, n_level=4):
        super().__init__(n_level=n_level)

    def base_polyhedron(self):
        raise NotImplementedError()

    def setup_vertices_faces(self):
        verts, faces = self.base_polyhedron()
        for _ in range(self.kwargs["n_level"] - 1):
            verts, faces = Rays_SubDivide.split(verts, faces)
        return verts, faces

    @classmethod
    def split(cls, verts0, faces0):
        mids = dict()
        verts = list(verts0[:])
        faces = []

        def mid(a, b):
            key = tuple(sorted((a, b)))
            if key not in mids:
                v = .5 * (verts[a] + verts[b])
                v *= 1. / np.linalg.norm(v)
                verts.append(v)
                mids[key] = len(verts) - 1
            return mids[key]
        for v1, v2, v3 in faces0:
            i1, i2, i3 = mid(v1, v2), mid(v2, v3), mid(v3, v1)
            faces += [[v1, i1, i3], [v2, i2, i1], [v3, i3, i2], [i1, i2, i3]]
        return verts, faces


class Rays_Tetra(Rays_SubDivide):
    def base_polyhedron(self):
        verts = np.array([[np.sqrt(8. / 9), 0., -1. / 3],
                          [-np.sqrt(2. / 9), np.sqrt(2. / 3), -1. / 3],
                          [-np.sqrt(2. / 9), -np.sqrt(2. / 3), -1. / 3],
                          [0., 0., 1.]])
        return verts, [[0, 1, 2], [0, 3, 1], [0, 2, 3], [1, 3, 2]]


class Rays_Octo(Rays_SubDivide):
    def base_polyhedron(self):
        verts = np.array([[0, 0, 1], [0, 1, 0], [0, 0, -1], [0, -1, 0], [1, 0, 0], [-1, 0, 0]])
        faces = [[0, 1, 4], [0, 5, 1], [1, 2, 4], [1, 5, 2], [2, 3, 4], [2, 5, 3], [3, 0, 4], [3, 5, 0]]
        return verts, faces


def reorder_faces(verts, faces):
    """flip triangles so that their orientation points outward (rays3d.py:330-334)"""
    return tuple((f[::-1] if np.linalg.det(verts[f]) > 0 else f) for f in faces)


class Rays_GoldenSpiral(Rays_Base):
    def __init__(self, n=70, anisotropy=None):
        if n < 4:
            raise ValueError("At least 4 points have to be given!")
        super().__init__(n=n, anisotropy=anisotropy if anisotropy is None else tuple(anisotropy))

    def setup_vertices_faces(self):
        from scipy.spatial import ConvexHull
        n = self.kwargs["n"]
        anisotropy = self.kwargs["anisotropy"]
        anisotropy = np.ones(3) if anisotropy is None else np.array(anisotropy)
        g = (3. - np.sqrt(5.)) * np.pi                 # the smaller golden angle
        phi = g * np.arange(n)
        z = np.linspace(-1, 1, n)
        rho = np.sqrt(1. - z ** 2)
        verts = np.stack([z, rho * np.sin(phi), rho * np.cos(phi)]).T
        verts = verts / anisotropy
        hull = ConvexHull(verts)
        faces = reorder_faces(verts, hull.simplices)
        verts /= np.linalg.norm(verts, axis=-1, keepdims=True)
        return verts, faces
