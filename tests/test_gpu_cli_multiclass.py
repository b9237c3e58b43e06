"""GPU: the command line script end to end (model folder -> tiff in -> label tiff out) and the multi-class head
(stardist/models/model2d.py:339-347, base.py:595-614: class_prob / class_id in the result dict)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_cli_predict2d_equals_api(tmp_path):
    import torch
    import bench
    from oracle import synth
    from stardist_amd.models import Config2D, StarDist2D
    from stardist_amd.scripts import predict2d
    from stardist_amd.scripts._io import imread, imwrite
    from stardist_amd.utils import normalize
    dev = torch.device("cuda:0")
    img = synth.s2d_nuclei_image(256, 256, seed=4)
    model = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
    bench.calibrate_heads(model, torch.from_numpy(normalize(img, 1, 99.8)).to(dev), frac=0.03)
    folder = tmp_path / "my_model"
    folder.mkdir()
    (folder / "config.json").write_text(model.config.to_json())
    (folder / "thresholds.json").write_text(json.dumps(dict(prob=0.5, nms=0.4)))
    model.save_weights_npz(str(folder / "weights_best.npz"))
    imwrite(str(tmp_path / "in.tif"), img)
    rc = predict2d.main(["-i", str(tmp_path / "in.tif"), "-m", str(folder), "-o", str(tmp_path / "out"), "--n_tiles", "2", "1"])
    assert rc == 0
    got = imread(str(tmp_path / "out" / "in.stardist.tif"))
    model.thresholds = dict(prob=0.5, nms=0.4)
    want, _ = model.predict_instances(normalize(imread(str(tmp_path / "in.tif")), 1, 99.8), n_tiles=(2, 1))
    assert got.shape == want.shape and np.array_equal(got, want) and want.max() > 5


def test_multiclass_head_sparse_equals_dense():
    import torch
    import bench
    from oracle import synth
    from stardist_amd.models import Config2D, StarDist2D
    dev = torch.device("cuda:0")
    img = synth.s2d_nuclei_image(256, 256, seed=6)
    model = StarDist2D(Config2D(n_rays=32, n_classes=3), basedir=None, device=dev, seed=0)
    bench.calibrate_heads(model, torch.from_numpy(img).to(dev), frac=0.03)
    l1, r1 = model.predict_instances(img, sparse=True)
    l2, r2 = model.predict_instances(img, sparse=False)
    assert np.array_equal(l1, l2) and np.array_equal(r1["points"], r2["points"])
    for r in (r1, r2):
        assert r["class_prob"].shape == (len(r["prob"]), 4) and r["class_id"].shape == (len(r["prob"]),)
        assert np.allclose(r["class_prob"].sum(1), 1, atol=1e-5) and np.array_equal(r["class_id"], r["class_prob"].argmax(1))
    assert np.array_equal(r1["class_prob"], r2["class_prob"])
    # the class probabilities are the softmax head sampled at the instance centres
    prob, dist, prob_class = model.predict(img)
    pts = np.asarray(r1["points"]).astype(int)
    assert np.allclose(r1["class_prob"], prob_class[pts[:, 0], pts[:, 1]], atol=1e-6)
