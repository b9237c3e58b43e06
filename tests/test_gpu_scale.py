"""GPU: `predict_instances(img, scale=...)` end to end -- the reference's own test of the option (tests/test_model2D.py:526-555,
tests/test_model3D.py test_predict_with_scale): predicting with `scale` equals predicting the zoomed image, with centres and
coordinates brought back to the input's grid, and the label image has the INPUT's shape.

The host logic of the option is also pinned on the CPU (tests/test_cpu_host_logic.py: the zoom, the per-axis dict, the rescaled centres /
coordinates / rays on the numpy and the tensor path)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("scale", (0.5, 2.0, (0.34, 1.47)))
def test_predict_with_scale_2d(scale):
    import torch
    from scipy.ndimage import zoom
    sys.path.insert(0, ROOT)
    import bench
    from oracle import synth
    from stardist_amd.models import Config2D, StarDist2D
    scale = (scale, scale) if np.isscalar(scale) else tuple(scale)
    dev = torch.device("cuda:0")
    x = synth.s2d_nuclei_image(192, 256, seed=2)
    model = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
    bench.calibrate_heads(model, torch.from_numpy(synth.s2d_nuclei_image(256, 256, seed=1)).to(dev))
    labels, res = model.predict_instances(x, scale=scale)
    labels_scaled, res_scaled = model.predict_instances(zoom(x, scale, order=1))
    assert labels.shape == x.shape[:2] and len(res["prob"]) == len(res_scaled["prob"]) > 0
    assert np.allclose(res["points"] * np.asarray(scale).reshape(1, 2), res_scaled["points"])
    assert np.allclose(res["coord"] * np.asarray(scale).reshape(1, 2, 1), res_scaled["coord"], atol=1e-4)
    assert np.allclose(res["prob"], res_scaled["prob"])


@pytest.mark.parametrize("scale", (0.5, (1.0, 1.5, 0.75)))
def test_predict_with_scale_3d(scale):
    import torch
    from scipy.ndimage import zoom
    sys.path.insert(0, ROOT)
    import bench
    from oracle import synth
    from stardist_amd.models import Config3D, StarDist3D
    scale = (scale,) * 3 if np.isscalar(scale) else tuple(scale)
    dev = torch.device("cuda:0")
    x = synth.s3d_nuclei_image(64, seed=1)
    model = StarDist3D(Config3D(rays=32), basedir=None, device=dev, seed=0)
    model.thresholds = dict(prob=0.5, nms=0.3)
    bench.calibrate_heads(model, torch.from_numpy(synth.s3d_nuclei_image(64, seed=0)).to(dev), frac=0.009, radius=8.5, noise=0.03)
    labels, res = model.predict_instances(x, scale=scale)
    labels_scaled, res_scaled = model.predict_instances(zoom(x, scale, order=1))
    assert labels.shape == x.shape and len(res["prob"]) == len(res_scaled["prob"])
    assert np.allclose(res["points"] * np.asarray(scale).reshape(1, 3), res_scaled["points"])
    assert np.allclose(res["dist"], res_scaled["dist"]) and np.allclose(res["prob"], res_scaled["prob"])
    assert np.allclose(res["rays_vertices"] * np.asarray(scale).reshape(1, 3), res_scaled["rays_vertices"], atol=1e-6)
