"""CPU: the REFERENCE'S OWN TEST FILES, run as they are against this package -- `import stardist` resolves to `stardist_amd`.

tests/test_stardist2D.py, test_stardist3D.py, test_nms2D.py, test_nms3D.py, test_big.py are imported from /root/reference/tests at run
time (nothing copied) with `sys.modules["stardist"]` = stardist_amd, and every test function in them that needs neither a trained model
(fixtures `model2d` / `model3d`: weights are absent, .MISSING_LARGE_BLOBS) nor OpenCL (`@pytest.mark.gpu` there) is called with its own
`parametrize` sets.  A user of the
reference who switches packages meets exactly these call sites: the top-level names, their signatures, dtypes, grids, the old / new NMS
pair, the polyhedron rasteriser against the NMS (`test_nms_accuracy`), the block cover / filter / reassemble identity.

There is no GPU here and the product has no CPU path: the natives behind the package's `lib` modules are stood in for by the compiled
reference natives (oracle/_ref) -- the replay pins the PACKAGE SURFACE and its Python glue under the reference's own assertions; the HIP
natives are held to the compiled reference on the GPU (tests/test_gpu_*).  Third-party imports of the reference's test helpers that are
absent offline are stood in for: tifffile.imread (PIL), skimage.measure.label (scipy.ndimage.label, full connectivity),
csbdeep.utils.normalize (the package's restatement), csbdeep.utils.tf.keras_import.  Build container only."""
import importlib.util
import inspect
import itertools
import os
import sys
import types

import numpy as np
import pytest

REF_TESTS = "/root/reference/tests"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_TESTS), reason="needs the reference sources (build container only)")
f32, i32 = (lambda a: np.ascontiguousarray(a, np.float32)), (lambda a: np.ascontiguousarray(a, np.int32))


def _imread(path):
    from PIL import Image
    im = Image.open(path)
    frames = []
    for k in range(getattr(im, "n_frames", 1)):
        im.seek(k)
        frames.append(np.array(im))
    return frames[0] if len(frames) == 1 else np.stack(frames)


def _label(img, **kw):
    from scipy import ndimage as ndi
    return ndi.label(img, structure=np.ones((3,) * np.ndim(img)))[0]


def _disk(center, radius, shape=None):
    """skimage.draw.disk: the pixels strictly inside the circle, clipped to `shape`"""
    r0, c0 = center
    lo_r, hi_r = int(np.floor(r0 - radius)), int(np.ceil(r0 + radius)) + 1
    lo_c, hi_c = int(np.floor(c0 - radius)), int(np.ceil(c0 + radius)) + 1
    if shape is not None:
        lo_r, lo_c, hi_r, hi_c = max(lo_r, 0), max(lo_c, 0), min(hi_r, shape[0]), min(hi_c, shape[1])
    rr, cc = np.mgrid[lo_r:hi_r, lo_c:hi_c]
    inside = ((rr - r0) / radius) ** 2 + ((cc - c0) / radius) ** 2 < 1
    return rr[inside], cc[inside]


def _nuclei_2d(return_mask=False):
    """stardist.data.test_image_nuclei_2d: the sample image (and mask) the reference package ships, read where it lies"""
    d = "/root/reference/stardist/data/images"
    img, mask = _imread(os.path.join(d, "img2d.tif")), _imread(os.path.join(d, "mask2d.tif"))
    return (img, mask) if return_mask else img


@pytest.fixture()
def as_stardist(monkeypatch, refmods):
    """`import stardist` -> stardist_amd (natives -> compiled reference), the absent third-party helpers, the reference's tests dir on the path"""
    import stardist_amd
    import stardist_amd.big
    import stardist_amd.geometry
    import stardist_amd.matching
    from oracle import port
    from stardist_amd.lib import stardist2d as sd2, stardist3d as sd3
    from stardist_amd.utils import normalize
    m2, m3 = refmods.stardist2d(), refmods.stardist3d()
    refmods.set_threads(4)
    monkeypatch.setattr(sd2, "c_non_max_suppression_inds", lambda d, p, a, b, c, t, **k: m2.c_non_max_suppression_inds(f32(d), f32(p), int(a), int(b), int(c), np.float32(t)).astype(bool))
    monkeypatch.setattr(sd2, "c_non_max_suppression_inds_old", lambda polys, mapping, t, mb, gy, gx, v: m2.c_non_max_suppression_inds_old(i32(polys), i32(mapping), np.float32(t), int(mb), int(gy), int(gx), int(v)).astype(bool))
    monkeypatch.setattr(sd2, "c_star_dist", lambda src, n, gy, gx: m2.c_star_dist(np.ascontiguousarray(src, np.uint16), int(n), int(gy), int(gx)))
    monkeypatch.setattr(sd2, "c_polygons_to_label", lambda coord, labels, shape, window=None: port.polygons_to_label_coord(coord, shape, labels=labels))
    monkeypatch.setattr(sd3, "c_non_max_suppression_inds", lambda d, p, V, F, s, a, b, c, t, **k: m3.c_non_max_suppression_inds(f32(d), f32(p), f32(V), i32(F), f32(s), int(a), int(b), int(c), np.float32(t)).astype(bool))
    monkeypatch.setattr(sd3, "c_polyhedron_to_label", lambda d, p, V, F, l, mode, vb, uo, ol, shape, window=None: m3.c_polyhedron_to_label(f32(d), f32(p), f32(V), i32(F), i32(l), int(mode), int(vb), int(uo), int(ol), tuple(int(s) for s in shape)))
    monkeypatch.setattr(sd3, "c_star_dist3d", lambda src, dz, dy, dx, n, gz, gy, gx: m3.c_star_dist3d(np.ascontiguousarray(src, np.uint16), f32(dz), f32(dy), f32(dx), int(n), int(gz), int(gy), int(gx)))

    # edt_prob is a HIP kernel in this package: stood in for by the reference's own scipy form (stardist/utils.py:99-125), taken from its file
    import ast
    import warnings
    from scipy.ndimage import distance_transform_edt, find_objects
    ns = dict(np=np, warnings=warnings, distance_transform_edt=distance_transform_edt, find_objects=find_objects)
    upath = "/root/reference/stardist/utils.py"
    for node in ast.parse(open(upath).read()).body:
        if isinstance(node, ast.FunctionDef) and node.name == "_edt_prob_scipy":
            exec(compile(ast.Module([node], []), upath, "exec"), ns)
    import stardist_amd.utils
    monkeypatch.setattr(stardist_amd.utils, "edt_prob", ns["_edt_prob_scipy"])

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__path__ = []
        return m

    installed = {"stardist": stardist_amd, "stardist.matching": stardist_amd.matching, "stardist.geometry": stardist_amd.geometry, "stardist.big": stardist_amd.big,
                 "csbdeep": mod("csbdeep"), "csbdeep.utils": mod("csbdeep.utils", normalize=normalize),
                 "csbdeep.utils.tf": mod("csbdeep.utils.tf", keras_import=lambda *a, **k: object),
                 "tifffile": mod("tifffile", imread=_imread), "skimage": mod("skimage"), "skimage.measure": mod("skimage.measure", label=_label),
                 "skimage.draw": mod("skimage.draw", disk=_disk), "stardist.data": mod("stardist.data", test_image_nuclei_2d=_nuclei_2d)}
    # every sub-module under its `stardist.` name as the SAME object (a second import under the alias would make copies the stand-ins miss)
    import stardist_amd.nms, stardist_amd.rays3d, stardist_amd.geometry.geom2d, stardist_amd.geometry.geom3d  # noqa: E401,F401
    for k in [k for k in sys.modules if k.startswith("stardist_amd.")]:
        installed.setdefault("stardist." + k[len("stardist_amd."):], sys.modules[k])
    for k, v in installed.items():
        monkeypatch.setitem(sys.modules, k, v)
    monkeypatch.syspath_prepend(REF_TESTS)
    monkeypatch.delitem(sys.modules, "utils", raising=False)
    yield
    sys.modules.pop("utils", None)


def _load(name):
    spec = importlib.util.spec_from_file_location("_ref_tests_" + name, os.path.join(REF_TESTS, name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _cases(fn):
    """the parameter sets of a test function, from its own pytest.mark.parametrize decorators; None if it needs a fixture or is marked gpu"""
    marks = list(getattr(fn, "pytestmark", []))
    if any(mk.name == "gpu" for mk in marks):
        return None
    axes = []
    for mk in marks:
        if mk.name != "parametrize":
            continue
        names = [n.strip() for n in mk.args[0].split(",")] if isinstance(mk.args[0], str) else list(mk.args[0])
        vals = [tuple(getattr(v, "values", v)) if len(names) > 1 else (getattr(v, "values", (v,))[0] if hasattr(v, "values") else v,) for v in mk.args[1]]
        axes.append([dict(zip(names, v)) for v in vals])
    required = [p.name for p in inspect.signature(fn).parameters.values() if p.default is inspect.Parameter.empty]
    combos = [dict(itertools.chain.from_iterable(d.items() for d in c)) for c in itertools.product(*axes)] if axes else [dict()]
    if any(set(required) - set(c) for c in combos):
        return None                                                  # a fixture (model2d / model3d)
    return combos


def _replay(module_name, skip=()):
    mod = _load(module_name)
    ran, skipped = {}, {}
    for name, fn in sorted(vars(mod).items()):
        if not (name.startswith("test_") and callable(fn) and getattr(fn, "__module__", None) == mod.__name__):
            continue                                                   # (imported helpers such as stardist.data.test_image_nuclei_2d are no tests)
        if name in skip:
            skipped[name] = skip[name]
            continue
        cases = _cases(fn)
        if cases is None:
            skipped[name] = "needs a trained model or OpenCL"
            continue
        n = 0
        for kw in cases:
            try:
                np.random.seed(1234 + n)
                fn(**kw)
                n += 1
            except pytest.skip.Exception as e:
                skipped[name] = str(e)
                break
        if n:
            ran[name] = n
    return ran, skipped


def test_reference_test_stardist2D_runs_against_this_package(as_stardist):
    ran, skipped = _replay("test_stardist2D")
    assert ran == {"test_types": 6, "test_relabel_consistency": 4, "test_grid": 4}, (ran, skipped)
    assert set(skipped) == {"test_types_gpu", "test_cpu_gpu"}


def test_reference_test_stardist3D_runs_against_this_package(as_stardist):
    ran, skipped = _replay("test_stardist3D")
    assert ran == {"test_types": 12, "test_relabel_consistency": 4, "test_grid": 4}, (ran, skipped)
    assert set(skipped) == {"test_types_gpu", "test_cpu_gpu"}


def test_reference_test_nms2D_runs_against_this_package(as_stardist):
    ran, skipped = _replay("test_nms2D", skip={"test_speed": "prints timings only", "test_large": "2000 x 2007 smoke run: minutes on the CPU natives"})
    assert ran == {"test_bbox_search_old": 2, "test_old_new": 8, "test_acc": 2, "test_acc_old": 2}, (ran, skipped)       # test_acc*: matching accuracy > 0.9
    assert set(skipped) == {"test_speed", "test_large"}


def test_reference_test_nms3D_runs_against_this_package(as_stardist):
    ran, skipped = _replay("test_nms3D", skip={"test_speed": "prints timings only", "test_rays_volume_area": "needs skimage.measure.regionprops; prints only"})
    assert ran == {"test_nms": 4, "test_label": 1, "test_nms_kdtree": 1, "test_nms_accuracy": 12}, (ran, skipped)


def test_reference_test_big_runs_against_this_package(as_stardist):
    ran, skipped = _replay("test_big")
    assert ran == {"test_cover2D": 18, "test_cover3D": 6, "test_edgecases": 1}, (ran, skipped)
    assert set(skipped) == {"test_predict2D", "test_predict3D", "test_polygon_order_2D", "test_polyhedron_order_3D"}


def test_polygon_and_polyhedron_helpers_equal_the_reference_classes(as_stardist):
    """stardist.big.Polygon / Polyhedron (big.py:452-498) -- bbox, slice, shape, relative coordinates, mask, the common box of several
    objects, clipping to the image -- against the reference's own classes (rasterisers: the same stand-ins on both sides)"""
    from oracle import port
    from test_cpu_vs_reference_source import ref_functions
    import stardist_amd.geometry
    from stardist_amd.big import Polygon, Polyhedron
    from stardist_amd.rays3d import Rays_GoldenSpiral
    ns = ref_functions("big.py", {"Polygon", "Polyhedron"}, {"np": np, "polygon": port.polygon, "polyhedron_to_label": stardist_amd.geometry.polyhedron_to_label})
    RPolygon, RPolyhedron = ns["Polygon"], ns["Polyhedron"]
    rng = np.random.RandomState(21)
    phi = np.linspace(0, 2 * np.pi, 16, endpoint=False)
    for it in range(40):
        kw = {} if it % 3 else dict(shape_max=(48, 52))
        c = rng.uniform(5, 44 if kw else 60, 2)                       # (a polygon wholly outside shape_max has no box on either side)
        r = rng.uniform(2, 12, 16)
        coord = np.stack([c[0] + r * np.sin(phi), c[1] + r * np.cos(phi)])
        a, b = Polygon(coord, **kw), RPolygon(coord, **kw)
        assert a.bbox == b.bbox and a.slice == b.slice and a.shape == b.shape and np.array_equal(a.coord, b.coord)
        assert a.mask.dtype == b.mask.dtype and np.array_equal(a.mask, b.mask)
        other = coord + rng.uniform(-6, 6, (2, 1))
        union = Polygon.coords_bbox(coord, other, **kw)
        assert union == RPolygon.coords_bbox(coord, other, **kw)
        assert np.array_equal(Polygon(coord, bbox=union).mask, RPolygon(coord, bbox=union).mask)
    rays = Rays_GoldenSpiral(24, anisotropy=(2, 1, 1))
    for it in range(10):
        dist = rng.uniform(3, 7, 24).astype(np.float32)
        origin = rng.uniform(6, 20, 3)
        kw = {} if it % 2 else dict(shape_max=(18, 22, 24))
        a, b = Polyhedron(dist, origin, rays, **kw), RPolyhedron(dist, origin, rays, **kw)
        assert a.bbox == b.bbox and a.slice == b.slice and a.shape == b.shape and np.array_equal(a.mask, b.mask) and a.mask.any()
        d2, o2 = rng.uniform(3, 7, 24).astype(np.float32), origin + rng.uniform(-3, 3, 3)
        assert Polyhedron.coords_bbox((dist, origin), (d2, o2), rays=rays, **kw) == RPolyhedron.coords_bbox((dist, origin), (d2, o2), rays=rays, **kw)


def test_predict_big_tells_where_the_function_moved():
    """stardist.big.predict_big (big.py:596-602)"""
    from stardist_amd.big import predict_big
    from stardist_amd.models import Config2D, StarDist2D
    with pytest.raises(RuntimeError, match=r"moved to \{StarDist2D, StarDist3D\}\.predict_instances_big"):
        predict_big(object())
    m = StarDist2D(Config2D(n_rays=8, unet_n_depth=1, unet_n_filter_base=4), basedir=None, device="cpu")
    with pytest.raises(RuntimeError, match=r"moved to StarDist2D\.predict_instances_big"):
        predict_big(m, None)


def test_reference_test_matching_runs_against_this_package(as_stardist):
    """tests/test_matching.py: matching on shifted discs, _shuffle_labels and group_matching_labels on the package's sample mask with the
    reference's own expected label maxima [183, 199, 215, 231, 247, 263]"""
    ran, skipped = _replay("test_matching")
    assert ran == {"test_matching": 1, "test_grouping": 1} and not skipped, (ran, skipped)
