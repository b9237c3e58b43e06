"""CPU: the pre-trained model registry and from_pretrained (csbdeep BaseModel.from_pretrained + stardist/models/__init__.py:19-27),
on the reference's own example model folders (config.json / thresholds.json fixtures, tests/golden/make_pretrained_fixture.py) with
seeded weights written in the Keras .npz layout (the real weight files are not available offline)."""
import os
import shutil

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_registry_matches_reference_registrations():
    from stardist_amd.models import pretrained
    keys, aliases = pretrained.get_registered_models("StarDist2D")
    assert keys == ("2D_versatile_fluo", "2D_versatile_he", "2D_paper_dsb2018", "2D_demo")
    assert pretrained.get_registered_models("StarDist3D", return_aliases=False) == ("3D_demo",)
    assert pretrained.resolve("StarDist2D", "Versatile (fluorescent nuclei)") == "2D_versatile_fluo"
    assert pretrained.resolve("StarDist2D", "Versatile (H&E nuclei)") == "2D_versatile_he"
    assert pretrained.resolve("StarDist2D", "DSB 2018 (from StarDist 2D paper)") == "2D_paper_dsb2018"
    assert pretrained._MODELS["StarDist2D"]["2D_versatile_fluo"]["md5"] == "8db40dacb5a1311b8d2c447ad934fb8a"
    with pytest.raises(ValueError):
        pretrained.resolve("StarDist2D", "3D_demo")


@pytest.mark.parametrize("cls_name,key", [("StarDist2D", "2D_demo"), ("StarDist3D", "3D_demo")])
def test_from_pretrained_loads_config_thresholds_and_weights(cls_name, key, tmp_path, monkeypatch, capsys):
    import torch
    import stardist_amd.models as M
    cls = getattr(M, cls_name)
    cache = tmp_path / "cache"
    shutil.copytree(os.path.join(ROOT, "tests", "golden", "pretrained"), str(cache))
    monkeypatch.setenv("STARDIST_AMD_MODELS", str(cache))
    monkeypatch.setenv("HOME", str(tmp_path / "nohome"))
    # 1) no weights in the folder: seeded init, config and thresholds from the folder
    m0 = cls.from_pretrained(key, device="cpu")
    assert "Found model '%s'" % key in capsys.readouterr().out
    if key == "2D_demo":
        assert m0.config.n_rays == 32 and tuple(m0.config.grid) == (2, 2) and m0.config.net_conv_after_unet == 128
        assert abs(m0.thresholds.prob - 0.4861655269131771) < 1e-12 and m0.thresholds.nms == 0.5
    else:
        assert m0.config.n_dim == 3 and m0.config.backbone == "resnet" and tuple(m0.config.grid) == (1, 2, 2) and m0.config.n_rays == 96
    # 2) weights written in the Keras layout by another instance are picked up by from_pretrained (round trip)
    src = cls(m0.config, basedir=None, device="cpu", seed=123)
    src.save_weights_npz(str(cache / cls_name / key / "weights_best.npz"))
    m1 = cls.from_pretrained(key, device="cpu")
    for (n1, p1), (n2, p2) in zip(src.net.state_dict().items(), m1.net.state_dict().items()):
        assert n1 == n2 and torch.equal(p1, p2), n1
    # the loaded model predicts like the source model
    shape = (32, 48) if m0.config.n_dim == 2 else (8, 32, 32)
    x = np.random.RandomState(0).rand(*shape).astype(np.float32)
    a, b = src.predict(x), m1.predict(x)
    assert all(np.array_equal(u, v) for u, v in zip(a, b))
    # 3) aliases, listing, unknown names, missing folders
    assert cls.from_pretrained(None) is None and key in capsys.readouterr().out
    assert cls.from_pretrained("no such model") is None            # csbdeep: message on stderr, registry printed, None returned
    cap = capsys.readouterr()
    assert "Could not find model with name or alias 'no such model'" in cap.err and key in cap.out
    if cls_name == "StarDist2D":
        with pytest.raises(FileNotFoundError) as e:
            cls.from_pretrained("Versatile (fluorescent nuclei)")          # registered, but neither cached nor downloadable offline
        assert "python_2D_versatile_fluo.zip" in str(e.value)


@pytest.mark.parametrize("cls_name,key", [("StarDist2D", "fixture2d"), ("StarDist3D", "fixture3d")])
def test_keras_layout_fixture_written_by_an_independent_script_loads_and_predicts(cls_name, key):
    """tests/golden/make_keras_fixture.py (no stardist_amd import) writes Keras-named, Keras-layout weights in Keras' layer order and
    the outputs of its own numpy forward pass (grid (2,2) U-Net: unnamed pre-convolutions + csbdeep block names; ResNet: 7x7x7 stem,
    strided SAME-padded block convolutions, shortcut projection stored after the block's last convolution).  The loader maps them by
    name / graph order with kernel transposition; the module must reproduce the independent evaluation."""
    import stardist_amd.models as M
    d = os.path.join(ROOT, "tests", "golden", "keras_fixture", cls_name)
    m = getattr(M, cls_name)(config=None, name=key, basedir=d, device="cpu")
    e = np.load(os.path.join(d, key, "expected.npz"))
    prob, dist = m.predict(e["x"])[:2]
    assert prob.shape == e["prob"].shape and dist.shape == e["dist"].shape
    assert np.abs(prob - e["prob"]).max() <= 2e-6
    assert np.abs(dist - np.maximum(e["dist"], 1e-3)).max() <= 2e-5 * max(1.0, float(np.abs(e["dist"]).max()))


@pytest.mark.parametrize("cls_name,key", [("StarDist2D", "fixture2d"), ("StarDist3D", "fixture3d")])
def test_keras_h5_weights_read_without_h5py(cls_name, key, tmp_path):
    """the Keras HDF5 weight files themselves (tests/golden/make_keras_h5_fixture.py: written by the real HDF5 library, h5py 3.3, in
    keras.Model.save_weights' layout -- and a model.save()-style file with /model_weights and a chunked dataset) through the package's
    own minimal HDF5 reader (models/hdf5_min.py): every variable bit-identical to the .npz, in Keras' order; a model folder that holds
    ONLY weights_best.h5 loads and reproduces the independent evaluation (csbdeep _find_and_load_weights, stardist/models/base.py:232-252)"""
    import stardist_amd.models as M
    from stardist_amd.models import hdf5_min
    src = os.path.join(ROOT, "tests", "golden", "keras_fixture", cls_name, key)
    z = np.load(os.path.join(src, "weights_best.npz"))
    for fn in ("weights_best.h5", "model_saved.h5"):
        w = hdf5_min.read_keras_weights(os.path.join(src, fn))
        assert list(w) == list(z.files)
        assert all(w[k].dtype == z[k].dtype and np.array_equal(w[k], z[k]) for k in z.files)
    f = hdf5_min.File(os.path.join(src, "model_saved.h5"))
    assert np.array_equal(f.root["model_weights"]["chunked_copy"].read(), z[z.files[0]])            # unfiltered chunks, v1 B-tree
    assert bytes(f.root["model_weights"].attrs["backend"]) == b"tensorflow"
    with pytest.raises(KeyError):
        f.root["model_weights"]["no_such_layer"]
    with pytest.raises(hdf5_min.HDF5Error):
        hdf5_min.File(b"not an hdf5 file" * 100)
    folder = tmp_path / cls_name / key
    os.makedirs(str(folder))
    for fn in ("config.json", "thresholds.json", "weights_best.h5"):
        shutil.copy(os.path.join(src, fn), str(folder / fn))
    m = getattr(M, cls_name)(config=None, name=key, basedir=str(tmp_path / cls_name), device="cpu")
    e = np.load(os.path.join(src, "expected.npz"))
    prob, dist = m.predict(e["x"])[:2]
    assert np.abs(prob - e["prob"]).max() <= 2e-6
    assert np.abs(dist - np.maximum(e["dist"], 1e-3)).max() <= 2e-5 * max(1.0, float(np.abs(e["dist"]).max()))


def test_model_folder_semantics_follow_csbdeep_basemodel(tmp_path, capsys):
    """csbdeep BaseModel.__init__ / _set_logdir / _find_and_load_weights / load_weights + stardist/models/base.py:230-253 (restated: csbdeep is
    absent): a model built from a configuration creates <basedir>/<name>/config.json and reads no weights; built with config=None it
    reads config.json, thresholds.json and the preferred ('best', else the newest) weights file; load_weights(name) switches files"""
    import json
    import time
    import warnings
    import torch
    from stardist_amd.models import Config2D, StarDist2D
    cfg = Config2D(n_rays=8, unet_n_depth=1, unet_n_filter_base=4, net_conv_after_unet=8, grid=(2, 2))
    base = str(tmp_path)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                  # a fresh folder: no warning
        a = StarDist2D(cfg, name="m", basedir=base, device="cpu", seed=1)
    assert a.name == "m" and os.path.samefile(a.logdir, os.path.join(base, "m"))
    assert json.load(open(os.path.join(base, "m", "config.json"))) == json.loads(cfg.to_json())
    a.save_weights_npz(os.path.join(a.logdir, "weights_now.npz"))
    time.sleep(0.02)
    b = StarDist2D(cfg, name=None, basedir=base, device="cpu", seed=2)      # name None: a time stamp (csbdeep), its own folder
    assert b.name is not None and b.name != "m" and os.path.exists(os.path.join(base, b.name, "config.json"))
    b.save_weights_npz(os.path.join(a.logdir, "weights_last.npz"))          # the NEWER file of folder m holds b's weights
    with pytest.warns(UserWarning, match="already exists"):
        c = StarDist2D(cfg, name="m", basedir=base, device="cpu", seed=3)   # config given: the folder's weights are NOT read
    wa, wb, wc = (m.net.prob.weight.detach().clone() for m in (a, b, c))
    assert not torch.equal(wc, wa) and not torch.equal(wc, wb)
    capsys.readouterr()
    d = StarDist2D(None, name="m", basedir=base, device="cpu", seed=4)      # no 'best' file: the newest one
    out = capsys.readouterr().out
    assert "Loading network weights from 'weights_last.npz'." in out and "Using default values: prob_thresh=0.5, nms_thresh=0.4." in out
    assert torch.equal(d.net.prob.weight, wb) and tuple(d.config.grid) == (2, 2)
    d.load_weights("weights_now.npz")
    assert torch.equal(d.net.prob.weight, wa)
    with pytest.raises(FileNotFoundError):
        d.load_weights("weights_best.h5")
    c.save_weights_npz(os.path.join(a.logdir, "weights_best.npz"))
    os.utime(os.path.join(a.logdir, "weights_best.npz"), (1, 1))            # the oldest file -- but the preferred name
    json.dump(dict(prob=0.62, nms=1.5), open(os.path.join(a.logdir, "thresholds.json"), "w"))
    e = StarDist2D(None, name="m", basedir=base, device="cpu")
    out = capsys.readouterr().out
    assert "Loading network weights from 'weights_best.npz'." in out and "Loading thresholds from 'thresholds.json'." in out
    assert "- Invalid 'nms' threshold (1.5), using default value." in out and "prob_thresh=0.62, nms_thresh=0.4" in out
    assert torch.equal(e.net.prob.weight, wc) and e.thresholds.prob == 0.62 and e.thresholds.nms == 0.4
    # no folder: nothing is written, load_weights only warns; no configuration and no folder: an error
    f = StarDist2D(cfg, basedir=None, device="cpu")
    assert f.logdir is None
    with pytest.warns(UserWarning, match="basedir=None"):
        f.load_weights()
    with pytest.raises(FileNotFoundError):
        StarDist2D(None, name="absent", basedir=base, device="cpu")
    with pytest.raises(ValueError):
        StarDist2D(cfg, name="", basedir=None, device="cpu")
    assert sorted(os.listdir(base)) == sorted(["m", b.name])


def test_a_converted_npz_stands_for_its_keras_file_only_while_it_is_not_older(tmp_path, monkeypatch):
    """weights_best.npz next to weights_best.h5 is the converted copy and is loaded in its place -- unless the Keras file is newer
    (fresh weights dropped into the folder): then the Keras file is loaded, as csbdeep's newest-file rule would, with a warning that names the stale copy"""
    import warnings
    from stardist_amd.models import Config2D, StarDist2D
    cfg = Config2D(n_rays=8, unet_n_depth=1, unet_n_filter_base=4, net_conv_after_unet=8)
    m = StarDist2D(cfg, name="m", basedir=str(tmp_path), device="cpu", seed=1)
    npz, h5 = os.path.join(m.logdir, "weights_best.npz"), os.path.join(m.logdir, "weights_best.h5")
    m.save_weights_npz(npz)
    open(h5, "wb").write(b"not read in this test")
    loaded = []
    monkeypatch.setattr(StarDist2D, "load_weights_h5", lambda self, p: loaded.append(os.path.basename(p)))
    monkeypatch.setattr(StarDist2D, "load_weights_npz", lambda self, p: loaded.append(os.path.basename(p)))
    os.utime(npz, (2000, 2000)); os.utime(h5, (1000, 1000))             # the copy is newer than the Keras file: the copy
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        m._find_and_load_weights()
    os.utime(npz, (1000, 1000)); os.utime(h5, (2000, 2000))             # the Keras file is newer: the Keras file, loudly
    with pytest.warns(UserWarning, match="out of date"):
        m._find_and_load_weights()
    assert loaded == ["weights_best.npz", "weights_best.h5"]


def test_sample_image_accessors_return_the_reference_images():
    """stardist/data/__init__.py:7-39: the arrays the reference's accessors read with tifffile / imageio (here: Pillow on the reference's files,
    build container only) == what stardist_amd.data serves from its images.npz"""
    from stardist_amd.data import test_image_he_2d, test_image_nuclei_2d, test_image_nuclei_3d
    img, mask = test_image_nuclei_2d(return_mask=True)
    assert img.shape == mask.shape == (512, 512) and test_image_nuclei_2d().shape == (512, 512)
    v, vm = test_image_nuclei_3d(return_mask=True)
    assert v.shape == vm.shape == (31, 61, 57) and int(vm.max()) > 10
    assert test_image_he_2d().shape == (300, 500, 3) and test_image_he_2d().dtype == np.uint8
    src = "/root/reference/stardist/data/images"
    if os.path.isdir(src):
        from PIL import Image

        def read(p):
            im = Image.open(p); pages = []
            for k in range(getattr(im, "n_frames", 1)):
                im.seek(k); pages.append(np.array(im))
            return pages[0] if len(pages) == 1 else np.stack(pages)
        assert np.array_equal(img, read(os.path.join(src, "img2d.tif"))) and np.array_equal(mask, read(os.path.join(src, "mask2d.tif")))
        assert np.array_equal(v, read(os.path.join(src, "img3d.tif"))) and np.array_equal(vm, read(os.path.join(src, "mask3d.tif")))
