"""Keras HDF5 weight files written by the REAL HDF5 library (h5py 3.3 / HDF5 1.10 of the image's conda python:
/opt/conda/bin/python3.9 -- the system python has no h5py), in exactly the layout of keras.Model.save_weights
(keras/saving/hdf5_format.py: save_weights_to_hdf5_group): root attributes layer_names / backend / keras_version, one group per
layer with the attribute weight_names, one contiguous dataset per variable at <layer>/<variable name> (the variable name repeats the
layer name, so the dataset sits in a nested group: /conv2d_1/conv2d_1/kernel:0).

Source of the values: the independent Keras-layout fixtures tests/golden/keras_fixture/*/weights_best.npz (make_keras_fixture.py).
Written next to them as weights_best.h5; stardist_amd/models/hdf5_min.py must read them back bit for bit (tests/test_cpu_pretrained.py).
A second file per model uses `model.save()` nesting (group /model_weights) and a chunked, gzip-free dataset to cover those branches.

usage: /opt/conda/bin/python3.9 tests/golden/make_keras_h5_fixture.py"""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def save_weights(f, arrays, order):
    """keras.saving.hdf5_format.save_weights_to_hdf5_group"""
    layers = []
    for name in order:
        ln = name.split("/")[0]
        if ln not in layers:
            layers.append(ln)
    f.attrs["layer_names"] = [n.encode("utf8") for n in layers]
    f.attrs["backend"] = "tensorflow".encode("utf8")
    f.attrs["keras_version"] = "2.3.1".encode("utf8")
    for ln in layers:
        g = f.create_group(ln)
        names = [n for n in order if n.split("/")[0] == ln]
        g.attrs["weight_names"] = [n.encode("utf8") for n in names]
        for n in names:
            val = arrays[n]
            d = g.create_dataset(n, val.shape, dtype=val.dtype)
            if not val.shape:
                d[()] = val
            else:
                d[:] = val


for cls, name in (("StarDist2D", "fixture2d"), ("StarDist3D", "fixture3d")):
    folder = os.path.join(HERE, "keras_fixture", cls, name)
    z = np.load(os.path.join(folder, "weights_best.npz"))
    arrays = {k: z[k] for k in z.files}
    with h5py.File(os.path.join(folder, "weights_best.h5"), "w") as f:
        save_weights(f, arrays, list(z.files))
    with h5py.File(os.path.join(folder, "model_saved.h5"), "w") as f:            # keras model.save(): weights under /model_weights
        f.attrs["model_config"] = "{}".encode("utf8")
        save_weights(f.create_group("model_weights"), arrays, list(z.files))
        k0 = z.files[0]
        f["model_weights"].create_dataset("chunked_copy", data=arrays[k0], chunks=tuple(max(1, s // 2) for s in arrays[k0].shape))
    print(folder, len(arrays), "variables")
