"""Where the 2D predict_instances step spends its time outside the network: every stage wrapped in synchronize + perf_counter
(serialises the stream, so the sum is an upper bound of the pipelined step).  usage: python tools/time_predict_sections.py [--size 2048]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import synth  # noqa: E402
from stardist_amd.models import Config2D, StarDist2D  # noqa: E402
import stardist_amd.models.model2d as M2  # noqa: E402
import stardist_amd.nms as NMS  # noqa: E402
import stardist_amd.geometry.geom2d as G2  # noqa: E402

acc = {}


def wrap(mod, name, label):
    orig = getattr(mod, name)

    def f(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = orig(*a, **k)
        torch.cuda.synchronize(); acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
        return r
    setattr(mod, name, f)
    return orig


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--size3d", type=int, default=256)
    ap.add_argument("--host-input", action="store_true", help="hand the image over as a host numpy array (SURVEY.md 8d's definition of the metric)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    img = torch.from_numpy(synth.s2d_nuclei_image(a.size, a.size, seed=0)).to(dev)
    model = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
    bench.calibrate_heads(model, img)
    if a.host_input:
        import stardist_amd.models.base as MB
        img = img.cpu().numpy()
        wrap(MB, "to_device", "to_device (H2D through the pinned stage)")
        a.size3d = 0
    for _ in range(3):
        model.predict_instances(img)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps):
        model.predict_instances(img)
    torch.cuda.synchronize(); whole = (time.perf_counter() - t0) / a.steps
    import stardist_amd.lib.stardist2d as L2
    import stardist_amd.utils as UT
    wrap(model, "_net_forward", "net_forward")
    wrap(model, "_select_sorted", "select + sort + distance head on the sorted rows")
    wrap(NMS, "nms_keep_sorted", "nms (sd_nms2d_device)")
    wrap(L2, "survivors_of_sorted", "survivors: positions, rows, coord, painting order (csrc/survivors.hip)")
    wrap(L2, "c_polygons_to_label", "raster")
    wrap(UT, "to_host_many", "results_to_host (labels, coord, points, prob)")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps):
        model.predict_instances(img)
    torch.cuda.synchronize(); ser = (time.perf_counter() - t0) / a.steps
    print("pipelined step %.3f ms; serialised step %.3f ms" % (1e3 * whole, 1e3 * ser))
    for k, v in acc.items():
        print("  %-42s %8.3f ms" % (k, 1e3 * v / a.steps))
    print("  %-42s %8.3f ms" % ("unaccounted (python glue, to-numpy copies)", 1e3 * (ser - sum(v for k, v in acc.items() if not k.startswith("  ")) / a.steps)))
    if a.size3d:
        sections_3d(a, dev)


def sections_3d(a, dev):
    from stardist_amd.models import Config3D, StarDist3D
    import stardist_amd.models.model3d as M3
    acc.clear()
    vol = torch.from_numpy(synth.s3d_nuclei_image(a.size3d, seed=0)).to(dev)
    m3 = StarDist3D(Config3D(rays=96), basedir=None, device=dev, seed=0)
    m3.thresholds = dict(prob=0.5, nms=0.3)
    bench.calibrate_heads(m3, vol, frac=0.009, radius=8.5, noise=0.03)
    for _ in range(2):
        m3.predict_instances(vol)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        m3.predict_instances(vol)
    torch.cuda.synchronize(); whole = (time.perf_counter() - t0) / 3
    import stardist_amd.lib.stardist3d as L3
    import stardist_amd.utils as UT
    wrap(m3, "_net_forward", "net_forward")
    wrap(m3, "_select_sorted", "select + sort + distance head on the sorted rows")
    wrap(NMS, "non_maximum_suppression_3d_sparse_sorted", "nms (native + survivor positions)")
    wrap(L3, "c_non_max_suppression_inds", "  nms_3d_inds(native)")
    wrap(M3, "polyhedron_to_label", "raster")
    wrap(M3, "relabel_sequential", "relabel_sequential")
    wrap(M3, "to_host", "labels_to_host")
    wrap(UT, "to_host_many", "dict_to_host (dist, points, prob)")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        m3.predict_instances(vol)
    torch.cuda.synchronize(); ser = (time.perf_counter() - t0) / 3
    print("3D: pipelined step %.3f ms; serialised step %.3f ms" % (1e3 * whole, 1e3 * ser))
    for k, v in acc.items():
        print("  %-42s %8.3f ms" % (k, 1e3 * v / 3))
    print("  %-42s %8.3f ms" % ("unaccounted", 1e3 * (ser - sum(v for k, v in acc.items() if not k.startswith("  ")) / 3)))
    from stardist_amd.lib import _native
    print("  nms3d native stats:", [int(v) for v in _native.last_stats.get("nms3d", [])])


if __name__ == "__main__":
    main()
