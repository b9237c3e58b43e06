"""Pin the network translation against the REAL Keras graph and the REAL pretrained weights -- to be run by a maintainer on a host that
has TensorFlow, csbdeep, the reference `stardist` package and network access (none of which exist in the build container: SURVEY.md 8c,
`.MISSING_LARGE_BLOBS`).  It closes the one parity row this repository cannot close offline.

What it checks, for a registered 2D model (default `2D_versatile_fluo`, BASELINE.json config 1):
  1. dense prediction: stardist_amd's `predict(img)` (MI355X kernels) against `stardist.models.StarDist2D.from_pretrained(key).predict(img)`
     (TensorFlow): max |d prob| <= 1e-5, max |d dist| / max(1, |dist|) <= 1e-5 (the north star's tolerance);
  2. the reference's own weight-dependent golden, tests/test_model2D.py:17-23: on tests/data/img2d.tif normalised with
     `normalize(img, 1, 99.8)`, `predict_instances` finds 119 objects whose label image has 55985 +- 10 foreground pixels;
  3. instance parity: same number of instances, same points, labels identical (the NMS / rasteriser natives are pinned offline
     against the compiled reference; this is the end-to-end statement on real weights).

  4. (`--layer-order`, needs no weights) builds the reference's 3D ResNet (`Config3D(backbone="resnet", grid=(1,2,2))`, the 3D_demo
     topology) in real Keras and prints the convolution layers in `model.layers` order -- the order `save_weights` writes: the offline
     stand-in (tests/_mini_keras.py) derives that a strided block's 1x1x1 projection precedes the block's last body convolution; the
     weight loader reads either order, this confirms which one real Keras produces.

usage:   python tools/verify_with_tf.py [--model 2D_versatile_fluo] [--image path.tif] [--device cuda:0] [--layer-order]
exit code 0 = all checks passed.  Nothing here is imported by the product or by the test suite."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def layer_order():
    try:
        from stardist.models import Config3D, StarDist3D as RefStarDist3D
    except ImportError as e:
        print("this needs the reference stack (tensorflow, csbdeep, stardist): %r" % (e,))
        return 2
    ref = RefStarDist3D(Config3D(backbone="resnet", grid=(1, 2, 2), resnet_n_blocks=2, rays=16), name=None, basedir=None)
    convs = [(l.name, tuple(l.kernel_size), tuple(l.strides)) for l in ref.keras_model.layers if hasattr(l, "kernel_size")]
    for name, k, st in convs:
        print("  %-16s kernel %s strides %s" % (name, k, st))
    idx = {n: i for i, (n, _, _) in enumerate(convs)}
    proj = [n for n, k, st in convs if k == (1, 1, 1) and st != (1, 1, 1)]
    if proj:
        body = [n for n, k, st in convs if k != (1, 1, 1) and idx[n] in (idx[proj[0]] - 1, idx[proj[0]] + 1)]
        print("projection %s is listed %s its neighbour %s (tests/_mini_keras.py derives: BEFORE the block's last body convolution)"
              % (proj[0], "before" if body and idx[proj[0]] < idx[body[-1]] else "after", body))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="2D_versatile_fluo")
    ap.add_argument("--image", default=None, help="default: the reference's tests/data/img2d.tif (stardist.data.test_image_nuclei_2d)")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--tol", type=float, default=1e-5)
    ap.add_argument("--layer-order", action="store_true", help="print model.layers' convolution order of a strided 3D ResNet and exit")
    args = ap.parse_args()
    if args.layer_order:
        return layer_order()

    try:
        from csbdeep.utils import normalize
        from stardist.models import StarDist2D as RefStarDist2D
        from stardist.data import test_image_nuclei_2d
    except ImportError as e:
        print("this script needs the reference stack (tensorflow, csbdeep, stardist): %r" % (e,))
        return 2
    if args.image:
        from tifffile import imread
        img = imread(args.image)
    else:
        img = test_image_nuclei_2d()
    x = normalize(img, 1, 99.8).astype(np.float32)          # tests/test_model2D.py:19

    ref = RefStarDist2D.from_pretrained(args.model)          # downloads the registered zip (models/__init__.py:19-27)
    folder = os.path.join(str(ref.basedir), ref.name)
    print("reference model folder:", folder)

    # the same folder through this repository: config.json, thresholds.json, weights_best.h5 (models/pretrained.py, models/hdf5_min.py)
    from stardist_amd.models import StarDist2D
    mine = StarDist2D(None, name=ref.name, basedir=str(ref.basedir), device=args.device)
    assert mine.config.n_rays == ref.config.n_rays and tuple(mine.config.grid) == tuple(ref.config.grid)
    mine.thresholds = dict(prob=ref.thresholds.prob, nms=ref.thresholds.nms)

    ok = True
    # 1. dense prediction
    p_ref, d_ref = ref.predict(x)
    p_my, d_my = mine.predict(x)
    e_p = float(np.abs(p_my - p_ref).max())
    e_d = float((np.abs(d_my - d_ref) / np.maximum(1.0, np.abs(d_ref))).max())
    print("dense prediction: max |d prob| = %.3g, max rel |d dist| = %.3g (tolerance %.1g)" % (e_p, e_d, args.tol))
    ok &= e_p <= args.tol and e_d <= args.tol

    # 2. the reference's weight-dependent golden (only meaningful for the default model on the default image)
    l_ref, r_ref = ref.predict_instances(x)
    l_my, r_my = mine.predict_instances(x)
    n_ref, n_my = len(r_ref["points"]), len(r_my["points"])
    fg_ref, fg_my = int((l_ref > 0).sum()), int((l_my > 0).sum())
    print("predict_instances: reference %d objects / %d foreground pixels, stardist_amd %d / %d" % (n_ref, fg_ref, n_my, fg_my))
    if args.model == "2D_versatile_fluo" and args.image is None:
        golden = n_my == 119 and abs(fg_my - 55985) <= 10     # tests/test_model2D.py:21-23
        print("golden of tests/test_model2D.py:17-23 (119 objects, 55985 +- 10 px): %s" % ("ok" if golden else "MISSED"))
        ok &= golden

    # 3. instance parity
    same_n = n_ref == n_my
    same_pts = same_n and np.array_equal(np.asarray(r_ref["points"]), np.asarray(r_my["points"]))
    same_lbl = np.array_equal(l_ref, l_my)
    print("instances: same count %s, same points %s, identical label image %s" % (same_n, same_pts, same_lbl))
    if not same_pts and same_n:
        # probabilities within 1e-5 can reorder exact ties: compare as sets
        a = set(map(tuple, np.asarray(r_ref["points"]).tolist())); b = set(map(tuple, np.asarray(r_my["points"]).tolist()))
        print("  points as sets: %d common, %d only reference, %d only stardist_amd" % (len(a & b), len(a - b), len(b - a)))
    ok &= same_n and same_lbl
    print("RESULT:", "PASS" if ok else "FAIL")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
