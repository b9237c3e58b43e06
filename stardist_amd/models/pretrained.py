"""Registry of pre-trained models and `from_pretrained` (csbdeep.models.pretrained + the registrations of
stardist/models/__init__.py:19-27, same keys / aliases / URLs / md5 sums).

The reference downloads and unpacks `<url>` into Keras' cache (`~/.keras/models/<ClassName>/<key>/`) with
`keras.utils.get_file` and then constructs `cls(config=None, name=key, basedir=<cache>/<ClassName>)`.  Here the same folder
layout is used; the folder is looked up in, in order: $STARDIST_AMD_MODELS/<ClassName>/<key>, ~/.keras/models/<ClassName>/<key>.
If it is missing a download is attempted (urllib + md5 check + unzip); without network that fails with a message that says
where to put the unpacked folder.  Weights: `weights.npz` / `weights_best.npz` (tools/keras_to_npz.py output) or the Keras
`weights_best.h5` itself (h5py if importable, else the package's own minimal HDF5 reader, models/hdf5_min.py).
"""
import hashlib
import os
import zipfile
from collections import OrderedDict

_MODELS = {}      # class name -> OrderedDict(key -> dict(url, md5))
_ALIASES = {}     # class name -> dict(alias -> key)


def register_model(cls_name, key, url, md5):
    _MODELS.setdefault(cls_name, OrderedDict())[key] = dict(url=url, md5=md5)


def register_aliases(cls_name, key, *aliases):
    if key not in _MODELS.get(cls_name, {}):
        raise ValueError("model key '%s' unknown" % key)
    for a in aliases:
        _ALIASES.setdefault(cls_name, {})[a] = key


# stardist/models/__init__.py:19-27
_REL = "https://github.com/stardist/stardist-models/releases/download/v0.1/"
register_model("StarDist2D", "2D_versatile_fluo", _REL + "python_2D_versatile_fluo.zip", "8db40dacb5a1311b8d2c447ad934fb8a")
register_model("StarDist2D", "2D_versatile_he", _REL + "python_2D_versatile_he.zip", "bf34cb3c0e5b3435971e18d66778a4ec")
register_model("StarDist2D", "2D_paper_dsb2018", _REL + "python_2D_paper_dsb2018.zip", "6287bf283f85c058ec3e7094b41039b5")
register_model("StarDist2D", "2D_demo", _REL + "python_2D_demo.zip", "31f70402f58c50dd231ec31b4375ea2c")
register_model("StarDist3D", "3D_demo", _REL + "python_3D_demo.zip", "f481c16c1ee9f28a8dcfa1e7aae3dc83")
register_aliases("StarDist2D", "2D_paper_dsb2018", "DSB 2018 (from StarDist 2D paper)")
register_aliases("StarDist2D", "2D_versatile_fluo", "Versatile (fluorescent nuclei)")
register_aliases("StarDist2D", "2D_versatile_he", "Versatile (H&E nuclei)")


def get_registered_models(cls_name, return_aliases=True):
    keys = tuple(_MODELS.get(cls_name, {}).keys())
    aliases = {k: tuple(a for a, kk in _ALIASES.get(cls_name, {}).items() if kk == k) for k in keys}
    return (keys, aliases) if return_aliases else keys


def print_registered(cls_name):
    keys, aliases = get_registered_models(cls_name)
    if not keys:
        print("There are no registered models for '%s'" % cls_name)
        return
    width = max(len(k) for k in keys)
    print("There are %d registered models for '%s':\n" % (len(keys), cls_name))
    print("%s | Alias(es)\n%s + %s" % ("Name".ljust(width), "-" * width, "-" * 20))
    for k in keys:
        print("%s | %s" % (("'%s'" % k).ljust(width), ", ".join("'%s'" % a for a in aliases[k]) or "None"))


def resolve(cls_name, name_or_alias):
    models, aliases = _MODELS.get(cls_name, {}), _ALIASES.get(cls_name, {})
    if name_or_alias in models:
        return name_or_alias
    if name_or_alias in aliases:
        return aliases[name_or_alias]
    raise ValueError("'%s' is neither a key nor an alias for '%s'; registered: %s" % (name_or_alias, cls_name, ", ".join(models) or "none"))


def _cache_roots():
    roots = []
    if os.environ.get("STARDIST_AMD_MODELS"):
        roots.append(os.environ["STARDIST_AMD_MODELS"])
    roots.append(os.path.join(os.path.expanduser("~"), ".keras", "models"))
    return roots


def get_model_folder(cls_name, key_or_alias):
    """folder of the unpacked model (downloaded on first use, like keras.utils.get_file(..., extract=True))"""
    key = resolve(cls_name, key_or_alias)
    for root in _cache_roots():
        d = os.path.join(root, cls_name, key)
        if os.path.exists(os.path.join(d, "config.json")):
            return d
    info = _MODELS[cls_name][key]
    target_root = os.path.join(_cache_roots()[0], cls_name)
    os.makedirs(target_root, exist_ok=True)
    zpath = os.path.join(target_root, key + ".zip")
    try:
        import urllib.request
        urllib.request.urlretrieve(info["url"], zpath)
    except Exception as e:
        raise FileNotFoundError("pre-trained model '%s' is not in %s and could not be downloaded from %s (%s); unpack the archive to %s"
                                % (key, [os.path.join(r, cls_name, key) for r in _cache_roots()], info["url"], e, os.path.join(target_root, key)))
    if hashlib.md5(open(zpath, "rb").read()).hexdigest() != info["md5"]:
        raise IOError("md5 mismatch for %s" % zpath)
    with zipfile.ZipFile(zpath) as z:
        z.extractall(target_root)
    d = os.path.join(target_root, key)
    if not os.path.exists(os.path.join(d, "config.json")):
        raise FileNotFoundError("archive %s did not contain %s/config.json" % (zpath, key))
    return d


def keras_h5_to_npz(src, dst):
    """Keras HDF5 weight file (weights_best.h5 / weights_last.h5 of a csbdeep model folder, or the folder itself) -> the .npz that
    StarDistBase.load_weights_npz reads: one entry per variable, named "<layer>/<variable>" ("conv2d_1/kernel:0"), in the order of the
    file's `layer_names` attribute (= Keras graph order); kernels stay in Keras layout (k..., cin, cout).  Read with h5py when it is
    importable, else with the package's own minimal HDF5 reader (models/hdf5_min.py: superblock 0-3, old- and new-style groups,
    contiguous / compact / unfiltered chunked float datasets -- everything keras.Model.save_weights writes)."""
    import numpy as np
    if not hasattr(src, "read") and os.path.isdir(src):
        for name in ("weights_best.h5", "weights_last.h5", "weights_now.h5"):
            if os.path.exists(os.path.join(src, name)):
                src = os.path.join(src, name)
                break
        else:
            raise FileNotFoundError("no weights_*.h5 in %s" % src)
    try:
        import h5py
    except ImportError:
        h5py = None
    if h5py is None:
        from .hdf5_min import read_keras_weights
        out = read_keras_weights(src)
    else:
        out = {}
        text = lambda n: n.decode() if isinstance(n, bytes) else n
        with h5py.File(src, "r") as f:
            g = f["model_weights"] if "model_weights" in f else f
            for ln in map(text, g.attrs["layer_names"]):
                lg = g[ln]
                for wn in map(text, lg.attrs.get("weight_names", [])):
                    out[wn if wn.startswith(ln) else ln + "/" + wn.split("/")[-1]] = np.asarray(lg[wn])
    np.savez(dst, **out)
    return list(out)
