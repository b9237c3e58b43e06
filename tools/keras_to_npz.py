"""Convert a StarDist Keras weight file (weights_best.h5 / weights_last.h5 of a csbdeep model folder) into the .npz consumed by
`StarDistBase.load_weights_npz` -- to be run with an interpreter that has h5py (in the build image: /opt/conda/bin/python3.9;
checked there on a synthetic Keras-style file only, real StarDist weight files are not available offline).

usage: python tools/keras_to_npz.py <model_dir or weights.h5> <out.npz>

The .npz holds one entry per variable, named "<layer>/<variable>" (e.g. "conv2d_1/kernel:0", "conv2d_1/bias:0"), in the order of
the file's `layer_names` attribute, i.e. the Keras graph order `load_weights_npz` expects (kernels (k..., cin, cout) are
transposed to torch's (cout, cin, k...) by the loader)."""
import os
import sys

import numpy as np


def convert(src, dst):
    import h5py
    if os.path.isdir(src):
        for name in ("weights_best.h5", "weights_last.h5", "weights_now.h5"):
            if os.path.exists(os.path.join(src, name)):
                src = os.path.join(src, name)
                break
        else:
            raise FileNotFoundError("no weights_*.h5 in %s" % src)
    out = {}
    with h5py.File(src, "r") as f:
        g = f["model_weights"] if "model_weights" in f else f
        layer_names = [n.decode() if isinstance(n, bytes) else n for n in g.attrs["layer_names"]]
        for ln in layer_names:
            lg = g[ln]
            for wn in [n.decode() if isinstance(n, bytes) else n for n in lg.attrs.get("weight_names", [])]:
                key = wn if wn.startswith(ln) else ln + "/" + wn.split("/")[-1]
                out[key] = np.asarray(lg[wn])
    np.savez(dst, **out)
    return list(out)


if __name__ == "__main__":
    if len(sys.argv) != 3:
        sys.exit(__doc__)
    keys = convert(sys.argv[1], sys.argv[2])
    print("wrote %d arrays to %s" % (len(keys), sys.argv[2]))
