"""Convert a StarDist Keras weight file (weights_best.h5 / weights_last.h5 of a csbdeep model folder) into the .npz consumed by
`StarDistBase.load_weights_npz` -- to be run with an interpreter that has h5py.

usage: python tools/keras_to_npz.py <model_dir or weights.h5> <out.npz>

(thin command-line wrapper of stardist_amd.models.pretrained.keras_h5_to_npz)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

if __name__ == "__main__":
    if len(sys.argv) != 3:
        sys.exit(__doc__)
    from stardist_amd.models.pretrained import keras_h5_to_npz
    keys = keras_h5_to_npz(sys.argv[1], sys.argv[2])
    print("wrote %d arrays to %s" % (len(keys), sys.argv[2]))
