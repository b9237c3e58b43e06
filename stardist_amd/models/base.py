"""StarDistBase: the prediction API of the reference's model classes, MI355X-native.

Mirrors stardist/models/base.py for the hot path only:
  predict / predict_sparse / predict_instances (+ the generator variants used by napari),
  _predict_setup :371-443, _predict_sparse_generator :541-633, _predict_generator :446-529,
  _predict_instances_generator :645-772, StarDistPadAndCropResizer :1162-1211.
Device-resident: the image goes to HBM once; network heads, `max(1e-3, dist)`, threshold +
border mask + compaction (select.hip), score sort, NMS and rasterisation all stay on the GPU;
only the label image and the survivor dict return to the host.
Training and export are out of scope; optimize_thresholds (the producer of thresholds.json) is a host-side tool
around the prediction natives.
"""
import json
import math
import numbers
import os
import warnings
from collections import namedtuple

import numpy as np

from ..lib import _native as N
from ..nms import _ind_prob_thresh
from ..utils import to_device


def axes_check_and_normalize(axes, length=None):
    """csbdeep.utils.axes_check_and_normalize (subset)"""
    allowed = "STCZYX"
    axes = str(axes).upper()
    if any(a not in allowed for a in axes) or any(axes.count(a) > 1 for a in axes):
        raise ValueError("invalid axes '%s'" % axes)
    if length is not None and len(axes) != length:
        raise ValueError("axes (%s) must be of length %d." % (axes, length))
    return axes


def axes_dict(axes):
    axes = axes_check_and_normalize(axes)
    return {a: (None if axes.find(a) == -1 else axes.find(a)) for a in "STCZYX"}


class StarDistPadAndCropResizer(object):
    """base.py:1162-1211: reflect-pad at the END of each axis to a multiple of div_by."""

    def __init__(self, grid, mode="reflect"):
        assert isinstance(grid, dict)
        self.mode, self.grid = mode, grid

    def before(self, x, axes, axes_div_by):
        """x: torch tensor laid out as `axes`"""
        import torch
        assert all(a % g == 0 for g, a in zip((self.grid.get(a, 1) for a in axes), axes_div_by))
        self.pad = {a: (0, (div_n - s % div_n) % div_n) for a, div_n, s in zip(axes, axes_div_by, x.shape)}
        pads = [self.pad[a][1] for a in axes]
        if any(pads):
            xn = x
            # np.pad(mode='reflect') one axis at a time (reflect without repeating the edge sample)
            for d, p in enumerate(pads):
                if p:
                    n = xn.shape[d]
                    if n == 1:                                    # numpy: a length-1 axis is repeated
                        idx = torch.zeros(p, dtype=torch.long, device=xn.device)
                    else:                                         # periodic reflection (period 2n-2), as np.pad does for p > n-1
                        k = torch.arange(n, n + p, device=xn.device) % (2 * n - 2)
                        idx = torch.where(k < n, k, 2 * n - 2 - k)
                    xn = torch.cat([xn, xn.index_select(d, idx)], dim=d)
            x = xn
        self.padded_shape = dict(zip(axes, x.shape))
        self.padded_shape.pop("C", None)
        return x

    def after(self, x, axes):
        crop = tuple(slice(0, -(math.floor(p[1] / g)) if p[1] >= g else None)
                     for p, g in zip((self.pad.get(a, (0, 0)) for a in axes), (self.grid.get(a, 1) for a in axes)))
        return x[crop]

    def filter_points(self, ndim, points, axes):
        """indices of points inside the un-padded region (base.py:1204-1211)"""
        bounds = tuple(self.padded_shape[a] - self.pad[a][1] for a in axes if a.lower() in ("z", "y", "x"))
        if N.is_torch(points):
            import torch
            b = torch.tensor(bounds, device=points.device, dtype=points.dtype)
            return torch.where(torch.all(points < b, dim=1))[0]
        return np.where(np.all(points < np.array(bounds), 1))[0]


class SortedCandidates(object):
    """candidates of one (untiled) prediction in SCORE order (descending, `np.argsort(prob)[::-1]` of stardist/nms.py:167 already applied),
    all on the device: prob (n,), dist (n, R), points (n, nd) int64 pixel coordinates, points_f32 the same as float32 (what the NMS
    natives take).  What _predict_instances_generator hands from the selection straight to the NMS natives."""
    __slots__ = ("prob", "dist", "points", "points_f32")

    def __init__(self, prob, dist, points, points_f32):
        self.prob, self.dist, self.points, self.points_f32 = prob, dist, points, points_f32


class StarDistBase(object):

    def __init__(self, config, name=None, basedir=".", device=None, seed=0, compute_dtype="float32"):
        import torch
        # the model folder, as csbdeep's BaseModel keeps it (csbdeep/models/base_model.py __init__ / _set_logdir; stardist/models/base.py:230-253):
        #   config given + basedir: <basedir>/<name> is created (a warning if it exists) and config.json written -- no weights are read;
        #   config None: config.json, thresholds.json and the weights of <basedir>/<name> are loaded; name None: a time stamp;
        #   basedir None: nothing touches the disk.
        if not (name is None or (isinstance(name, str) and len(name) > 0)):
            raise ValueError("No valid name: '%s'" % str(name))
        from_disk = config is None
        if name is None and (from_disk or basedir is None):
            self.name = None
        else:
            import datetime
            self.name = name if name is not None else datetime.datetime.now().strftime("%Y-%m-%d-%H-%M-%S.%f")
        self.basedir = basedir
        self.logdir = None if basedir is None else os.path.join(str(basedir), self.name if self.name is not None else "stardist_amd")
        if from_disk:
            if self.logdir is None or not os.path.exists(os.path.join(self.logdir, "config.json")):
                raise FileNotFoundError("config file doesn't exist: %s" % (None if self.logdir is None else os.path.abspath(os.path.join(self.logdir, "config.json"))))
            config = self._config_class.from_json(os.path.join(self.logdir, "config.json"))
        elif self.logdir is not None:
            if os.path.exists(self.logdir):
                warnings.warn("output path for model already exists, files may be overwritten: %s" % os.path.abspath(self.logdir))
            os.makedirs(self.logdir, exist_ok=True)
            with open(os.path.join(self.logdir, "config.json"), "w") as fh:
                fh.write(config.to_json())
        self.config = config
        threshs = dict(prob=None, nms=None)
        if self.logdir is not None:
            try:
                with open(os.path.join(self.logdir, "thresholds.json")) as fh:
                    threshs = json.load(fh)
                print("Loading thresholds from 'thresholds.json'.")
                if threshs.get("prob") is None or not (0 < threshs.get("prob") < 1):
                    print("- Invalid 'prob' threshold (%s), using default value." % str(threshs.get("prob")))
                    threshs["prob"] = None
                if threshs.get("nms") is None or not (0 < threshs.get("nms") < 1):
                    print("- Invalid 'nms' threshold (%s), using default value." % str(threshs.get("nms")))
                    threshs["nms"] = None
            except FileNotFoundError:
                import glob
                if from_disk and len(glob.glob(os.path.join(self.logdir, "*.h5"))) > 0:
                    print("Couldn't load thresholds from 'thresholds.json', using default values. "
                          "(Call 'optimize_thresholds' to change that.)")
        self.thresholds = dict(prob=0.5 if threshs["prob"] is None else threshs["prob"],
                               nms=0.4 if threshs["nms"] is None else threshs["nms"])
        if self.logdir is not None:      # (the reference prints this line for every model; kept to folder models: bench / library use stays silent)
            print("Using default values: prob_thresh={prob:g}, nms_thresh={nms:g}.".format(prob=self.thresholds.prob, nms=self.thresholds.nms))
        self.device = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        if compute_dtype != "float32":
            # (rounds 1-3 offered bfloat16 / float16 autocast through the framework's library kernels; the prediction path is float32
            # data on the library's own kernels -- whose matrix-core forms are already f32-accurate evaluations on the fp16 / bf16 pipes)
            raise ValueError("compute_dtype=%r: the prediction path computes in float32" % (compute_dtype,))
        self.compute_dtype = torch.float32
        self.net = self._build()
        from .unet import init_he_normal_
        init_he_normal_(self.net, seed)
        if from_disk:
            self._find_and_load_weights()
        self.net = self.net.to(self.device).eval()
        if self.device.type == "cuda":
            self.net = self.net.to(memory_format=torch.channels_last if config.n_dim == 2 else torch.channels_last_3d)

    # the seam where the reference calls keras_model.predict (base.py:408-410)
    @property
    def keras_model(self):
        return self.net

    @property
    def thresholds(self):
        return self._thresholds

    @thresholds.setter
    def thresholds(self, d):
        self._thresholds = namedtuple("Thresholds", d.keys())(*d.values())

    def _is_multiclass(self):
        return self.config.n_classes is not None

    @classmethod
    def from_pretrained(cls, name_or_alias=None, **kwargs):
        """csbdeep BaseModel.from_pretrained: `StarDist2D.from_pretrained('2D_versatile_fluo')`; without an argument the
        registered models are printed and None is returned (registry: stardist/models/__init__.py:19-27)."""
        from . import pretrained
        if name_or_alias is None:
            pretrained.print_registered(cls.__name__)
            return None
        try:
            pretrained.resolve(cls.__name__, name_or_alias)
        except ValueError:
            # csbdeep BaseModel.from_pretrained: an unknown name is reported on stderr, the registry is printed, None is returned
            import sys as _sys
            print("Could not find model with name or alias '%s'" % (name_or_alias,), file=_sys.stderr)
            _sys.stderr.flush()
            pretrained.print_registered(cls.__name__)
            return None
        folder = pretrained.get_model_folder(cls.__name__, name_or_alias)
        print("Found model '%s' for '%s'." % (os.path.basename(folder), cls.__name__))
        return cls(config=None, name=os.path.basename(folder), basedir=os.path.dirname(folder), **kwargs)

    def optimize_thresholds(self, X_val, Y_val, nms_threshs=[0.3, 0.4, 0.5], iou_threshs=[0.3, 0.5, 0.7], predict_kwargs=None, optimize_kwargs=None,
                            save_to_json=True):
        """Tune (prob_thresh, nms_thresh) on validation images X_val (normalised) with label images Y_val: for every nms threshold the
        prob threshold that maximises the matching score (stardist_amd.utils.optimize_threshold), the best pair becomes self.thresholds and
        is written to the model folder's thresholds.json -- the file later constructions read (base.py:986-1044).  Host-side tool around
        the prediction natives; returns dict(prob=..., nms=...)."""
        import sys
        from ..utils import optimize_threshold
        predict_kwargs = {} if predict_kwargs is None else predict_kwargs
        optimize_kwargs = {} if optimize_kwargs is None else optimize_kwargs

        def kwargs_for(x):
            if "n_tiles" in predict_kwargs:
                return predict_kwargs
            return dict(predict_kwargs, n_tiles=self._guess_n_tiles(x), show_tile_progress=False)
        Yhat_val = [self.predict(x, **kwargs_for(x))[:2] for x in X_val]           # (prob, dist) also of a multi-class model
        best = (None, -np.inf, None)
        for nms in nms_threshs:
            prob, value = optimize_threshold(Y_val, Yhat_val, model=self, nms_thresh=nms, iou_threshs=iou_threshs, **optimize_kwargs)
            if value > best[1]:
                best = (prob, value, nms)
        opt_threshs = dict(prob=best[0], nms=best[2])
        self.thresholds = opt_threshs
        print(end="", file=sys.stderr, flush=True)
        print("Using optimized values: prob_thresh={prob:g}, nms_thresh={nms:g}.".format(prob=self.thresholds.prob, nms=self.thresholds.nms))
        if save_to_json and self.basedir is not None:
            print("Saving to 'thresholds.json'.")
            with open(os.path.join(self.logdir, "thresholds.json"), "w") as fh:
                json.dump({k: float(v) for k, v in opt_threshs.items()}, fh)      # (numpy 2 keeps the search in float32: not serialisable as it is)
        return opt_threshs

    def _find_and_load_weights(self, prefer="best"):
        """csbdeep BaseModel._find_and_load_weights: of the weight files in the model folder (*.h5 / *.hdf5 Keras files, and the *.npz this
        package converts them to: tools/keras_to_npz.py, save_weights_npz) the newest one whose name contains `prefer`, else the newest"""
        import glob
        files = [f for ext in ("*.h5", "*.hdf5", "weights*.npz") for f in glob.glob(os.path.join(self.logdir, ext))]
        # newest first; of two files of one time stamp the converted .npz goes first (it needs no HDF5 reader)
        files.sort(key=lambda f: (-os.stat(f).st_mtime, not f.endswith(".npz"), f))
        if not files:
            warnings.warn("Couldn't find any network weights (*.h5, *.hdf5) to load.")
            return
        preferred = [f for f in files if prefer in os.path.basename(f)]
        # a converted copy stands for its Keras file -- weights_best.npz next to weights_best.h5 -- unless the Keras file is NEWER than the copy
        # (someone dropped fresh weights into the folder): then the Keras file is loaded, as csbdeep would, and the stale copy is named
        chosen = preferred[0] if preferred else files[0]
        twin = os.path.splitext(chosen)[0] + ".npz"
        if not chosen.endswith(".npz") and os.path.exists(twin):
            if os.stat(twin).st_mtime >= os.stat(chosen).st_mtime:
                chosen = twin
            else:
                warnings.warn("'%s' is older than '%s': loading the Keras file (the converted copy is out of date; tools/keras_to_npz.py renews it)"
                              % (os.path.basename(twin), os.path.basename(chosen)))
        print("Loading network weights from '%s'." % os.path.basename(chosen))
        self.load_weights(os.path.basename(chosen))

    def load_weights(self, name="weights_best.h5"):
        """csbdeep BaseModel.load_weights: the weights file `name` of the model folder -- Keras HDF5 (`weights_best.h5`, ...) or the
        converted `.npz`.  Without a model folder (basedir=None) a warning is given and nothing happens, as there."""
        if self.logdir is None:
            warnings.warn("Suppressing call of 'load_weights' (due to basedir=None).")
            return
        path = os.path.join(self.logdir, str(name))
        if not os.path.exists(path):
            raise FileNotFoundError("weights file doesn't exist: %s" % os.path.abspath(path))
        if path.endswith(".npz"):
            self.load_weights_npz(path)
        else:
            self.load_weights_h5(path)

    def load_weights_h5(self, path):
        """Keras HDF5 weights of a csbdeep model folder (needs h5py): converted in memory to the .npz layout, then loaded by name"""
        import io
        from .pretrained import keras_h5_to_npz
        buf = io.BytesIO()
        keras_h5_to_npz(path, buf)
        buf.seek(0)
        self.load_weights_npz(buf)

    def save_weights_npz(self, path):
        """inverse of load_weights_npz: Keras variable names and layouts ((k..., cin, cout) kernels), heads under their layer names"""
        import torch.nn as nn
        named = {id(self.net.prob): "prob", id(self.net.dist): "dist"}
        if isinstance(self.net.features, nn.Sequential):
            named[id(self.net.features[0])] = "features"
        if self.net.n_classes is not None:
            named[id(self.net.prob_class)] = "prob_class"
            if isinstance(self.net.features_class, nn.Sequential):
                named[id(self.net.features_class[0])] = "features_class"
        out, k = {}, 0
        for m in self.net.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv3d)):
                if id(m) in named:
                    name = named[id(m)]
                else:
                    name = "conv%dd_%d" % (self.config.n_dim, k); k += 1
                w = m.weight.detach().cpu().numpy()
                nd = w.ndim - 2
                out[name + "/kernel:0"] = np.ascontiguousarray(np.transpose(w, tuple(range(2, 2 + nd)) + (1, 0)))
                if m.bias is not None:
                    out[name + "/bias:0"] = m.bias.detach().cpu().numpy()
        for j, m in enumerate(mm for mm in self.net.modules() if isinstance(mm, (nn.BatchNorm2d, nn.BatchNorm3d))):
            base = "batch_normalization_%d/" % j
            out[base + "gamma:0"] = m.weight.detach().cpu().numpy(); out[base + "beta:0"] = m.bias.detach().cpu().numpy()
            out[base + "moving_mean:0"] = m.running_mean.cpu().numpy(); out[base + "moving_variance:0"] = m.running_var.cpu().numpy()
        np.savez(path, **out)

    def load_weights_npz(self, path):
        """weights exported from Keras (tools/keras_to_npz.py) as {layer_name/kernel:0, layer_name/bias:0, ...}.
        The output heads and the feature convolutions are matched BY NAME (prob, dist, features, features_class, prob_class:
        Keras orders layers by graph depth, which for a multi-class model differs from this module's order); the backbone
        convolutions (csbdeep block names vary between versions) are matched in graph order -- file order for a named backbone, the
        creation order of Keras' automatic names for an unnamed one (ResNet).  Every kernel's shape is checked."""
        import torch
        import torch.nn as nn
        data = np.load(path)
        kernels = [k for k in data.files if "kernel" in k]

        def lname(k):
            return k.split("/")[0]
        named = {"prob": self.net.prob, "dist": self.net.dist}
        if isinstance(self.net.features, nn.Sequential):
            named["features"] = self.net.features[0]
        if self.net.n_classes is not None:
            named["prob_class"] = self.net.prob_class
            if isinstance(self.net.features_class, nn.Sequential):
                named["features_class"] = self.net.features_class[0]
        head_ids = {id(m) for m in named.values()}
        backbone = [m for m in self.net.modules() if isinstance(m, (nn.Conv2d, nn.Conv3d)) and id(m) not in head_ids]
        by_name = {lname(k): k for k in kernels if lname(k) in named}
        missing = [n for n in named if n not in by_name]
        if missing:
            raise ValueError("weight file has no kernels for layer(s) %s" % ", ".join(missing))
        rest = [k for k in kernels if lname(k) not in named]
        # A backbone whose convolutions ALL carry Keras' automatic names (conv3d, conv3d_1, ...: the ResNet backbone -- csbdeep's
        # resnet_block names nothing, model3d.py:414-428) is matched in CREATION order, which the numeric suffix records and which is
        # this module's order (stem, then per block: strided convolution, body, shortcut projection).  The FILE order is model.layers'
        # -- by graph depth, and a block's 1x1 projection and its last body convolution have EQUAL depth: which of the two Keras lists
        # first depends on the operand order of the residual Add (tests/test_cpu_reference_build.py).  Named backbones (csbdeep's
        # unet_block: down_level_* / middle_* / up_level_*, after the unnamed grid stem) are chains without ties: file order.
        import re
        auto = [re.match(r"^conv\dd(?:_(\d+))?$", lname(k)) for k in rest]
        if rest and all(auto):
            rest = [k for _, k in sorted(zip((int(a.group(1) or 0) for a in auto), rest))]
        if len(rest) != len(backbone):
            raise ValueError("weight file has %d backbone conv kernels, network has %d" % (len(rest), len(backbone)))

        def put(m, kn):
            w = data[kn]
            nd = w.ndim - 2
            wt = np.ascontiguousarray(np.transpose(w, (nd + 1, nd) + tuple(range(nd))))
            if tuple(wt.shape) != tuple(m.weight.shape):
                raise ValueError("kernel %s has shape %s (torch layout %s), layer expects %s" % (kn, w.shape, wt.shape, tuple(m.weight.shape)))
            with torch.no_grad():
                m.weight.copy_(torch.from_numpy(wt))
                bn = kn.replace("kernel", "bias")
                if bn in data.files:
                    if m.bias is None or tuple(data[bn].shape) != tuple(m.bias.shape):
                        raise ValueError("bias %s does not fit its layer" % bn)
                    m.bias.copy_(torch.from_numpy(data[bn]))
        for n, m in named.items():
            put(m, by_name[n])
        for m, kn in zip(backbone, rest):
            put(m, kn)
        # batch normalisation layers (unet_batch_norm=True), in graph order: gamma, beta, moving_mean, moving_variance
        bns = [m for m in self.net.modules() if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d))]
        gam = [k for k in data.files if k.endswith("gamma:0")]
        if len(gam) != len(bns):
            raise ValueError("weight file has %d batch-normalisation layers, network has %d" % (len(gam), len(bns)))
        for m, gk in zip(bns, gam):
            base = gk[:-len("gamma:0")]
            with torch.no_grad():
                for attr, suffix in ((m.weight, "gamma:0"), (m.bias, "beta:0"), (m.running_mean, "moving_mean:0"), (m.running_var, "moving_variance:0")):
                    v = data[base + suffix]
                    if tuple(v.shape) != tuple(attr.shape):
                        raise ValueError("%s%s does not fit its layer" % (base, suffix))
                    attr.copy_(torch.from_numpy(v))
        # captured forward passes hold the packed form of the OLD kernels (the packed tensors are re-made when a parameter changes,
        # models/unet.py _packed_conv_weights); the layers' range fallbacks were decided on the old weights as well
        self.__dict__.pop("_graphs", None)
        for mod in self.net.modules():
            mod.__dict__.pop("_sd_force_form", None)

    # ------------------------------------------------------------------ helpers
    def _guess_n_tiles(self, img):
        """base.py:1046-1055: tiles of about one training batch of training patches each (1 for a channel axis)"""
        axes = self._normalize_axes(img, axes=None)
        shape = list(img.shape)
        if "C" in axes:
            del shape[axes_dict(axes)["C"]]
        b = self.config.train_batch_size ** (1.0 / self.config.n_dim)
        n_tiles = [int(np.ceil(s / (p * b))) for s, p in zip(shape, self.config.train_patch_size)]
        if "C" in axes:
            n_tiles.insert(axes_dict(axes)["C"], 1)
        return tuple(n_tiles)

    def _normalize_axes(self, img, axes):
        if axes is None:
            axes = self.config.axes
            assert "C" in axes
            if img.ndim == len(axes) - 1 and self.config.n_channel_in == 1:
                axes = axes.replace("C", "")
        return axes_check_and_normalize(axes, img.ndim)

    def _make_permute_axes(self, img_axes_in, net_axes_in, net_axes_out=None, img_axes_out=None):
        """csbdeep BaseModel._make_permute_axes (forward direction only)"""
        if net_axes_out is None: net_axes_out = net_axes_in
        if img_axes_out is None: img_axes_out = img_axes_in
        assert "C" in net_axes_in and "C" in net_axes_out
        if not ("C" in img_axes_in or "C" in img_axes_out):
            pass

        def _permute_axes(data, undo=False):
            assert not undo
            if "C" not in img_axes_in:
                data = data[..., None] if not N.is_torch(data) else data.unsqueeze(-1)
                src = img_axes_in + "C"
            else:
                src = img_axes_in
            perm = [src.index(a) for a in net_axes_in]
            if N.is_torch(data):
                return data.permute(*perm)
            return np.transpose(data, perm)
        return _permute_axes

    def _net_forward(self, x, sparse_head=False):
        """_net_forward_once under the range guard of the default split-fp16 convolutions (models/unet.py conv_mode): every such layer ORs
        ITS word of this model's flag tensor with 1 when an f32 activation it reads lies outside the fp16 range (|x| > 65504 or infinite),
        with 2 when a value of the split16 tensor it WRITES does (models/unet.py "split16": the producer makes the reader's fp16 terms).
        The words are read back after the pass (one 1-KiB copy); when one is set the outputs are discarded, a warning names the layers
        and the magnitude limit, exactly the layers that read the offending activation are moved to the six-product bf16 form (f32 range)
        for the rest of the model's life (its producer writes f32 again), and the pass is repeated -- the other layers stay on the fp16 form."""
        from . import unet
        if self.device.type != "cuda" or unet.conv_mode() != "f16x3":
            return self._net_forward_once(x, sparse_head)
        import torch
        flags = self.__dict__.get("_range_flags")
        if flags is None:
            flags = self._range_flags = torch.zeros(unet.N_FLAG_SLOTS, dtype=torch.int32, device=self.device)
        for _ in range(64):
            flags.zero_()
            unet.split16_replan(False)
            with unet.use_range_flags(flags):
                ys = self._net_forward_once(x, sparse_head)
            h = flags.cpu().numpy()
            if unet.split16_replan() and not h.any():
                # a layer met a split16 tensor it cannot read (models/unet.py _unpack_for): its producer writes f32 from now on; the pass is
                # repeated so that the result does not depend on which form carried the activation
                self.__dict__.pop("_graphs", None)
                continue
            if not h.any():
                return ys
            bad_in = set(int(k) for k in np.flatnonzero(h & 1))          # layers that READ an f32 activation outside the range
            bad_out = set(int(k) for k in np.flatnonzero(h & 2))         # layers whose split16 OUTPUT could not hold a value
            names, pinned = [], set()
            for name, mod in self.net.named_modules():
                slot = mod.__dict__.get("_sd_flag_slot")
                if slot in bad_in and mod.__dict__.get("_sd_force_form") != "bf16x6":
                    pinned.add(mod)
                if slot in bad_out:
                    # its readers (recorded by _hand_conv) move to the bf16x6 form, and it writes f32 again
                    mod.__dict__["_sd_split_out"] = False
                    for c in mod.__dict__.get("_sd_consumers", ()):
                        if c.__dict__.get("_sd_force_form") != "bf16x6":
                            pinned.add(c)
            for name, mod in self.net.named_modules():
                if mod in pinned:
                    mod.__dict__["_sd_force_form"] = "bf16x6"
                    names.append(name)
            if not names:                                    # (a word no layer of this model owns: cannot happen)
                raise RuntimeError("fp16 range flag set by an unknown layer")
            self.__dict__.setdefault("_fp16_range_layers", []).extend(names)
            self.__dict__.pop("_graphs", None)               # the captured passes launch the fp16 form of those layers
            import warnings
            warnings.warn("an activation read by layer(s) %s lies outside the fp16 range (|x| > 65504 or not finite): these layers use the "
                          "bf16x6 convolution kernels from now on (the others keep the fp16 form)" % ", ".join(names))
        raise RuntimeError("fp16 range fallback did not converge")

    def _net_forward_once(self, x, sparse_head=False):
        """x: torch tensor with axes_net semantics (channels last) -> tuple of channels-last outputs (prob, dist[, prob_class]).
        sparse_head=True (GPU, fused heads: models/unet.py): (prob, features[, prob_class]) when self._head_mode in ("sparse", "sparse_lazy") after
        the call -- the distance head is then evaluated on the selected rows only (_select_rows); otherwise as above.

        On the GPU the forward pass is captured once per input shape into a HIP graph (torch.cuda.CUDAGraph) and
        replayed: the network is a few dozen launches whose host-side dispatch through ctypes otherwise leaves the device idle
        between them.  Outputs of a replay are valid until the next call."""
        import torch
        nd = self.config.n_dim
        xc = x.permute(*([nd] + list(range(nd)))).unsqueeze(0)      # (1,C,...)
        mf = torch.channels_last if nd == 2 else torch.channels_last_3d
        use_graph = (self.device.type == "cuda") and getattr(self, "use_hip_graph", True)
        if not use_graph:
            ys = self._net_eager(xc.contiguous(memory_format=mf), sparse_head)
            self._head_mode = getattr(self.net, "head_mode", "dense")
        else:
            from .unet import conv_mode
            # everything the captured kernels depend on besides the weights: shape, head form, convolution kernel family
            from .unet import split16_enabled
            key = (tuple(xc.shape), xc.dtype, bool(sparse_head), conv_mode(), bool(getattr(self.net, "fused_heads", True)), split16_enabled())
            cache = self.__dict__.setdefault("_graphs", {})
            if key not in cache:
                if len(cache) >= 8:                                  # bounded: tiles of a big image share few shapes
                    cache.pop(next(iter(cache)))
                static_in = torch.empty(xc.shape, dtype=torch.float32, device=self.device).contiguous(memory_format=mf)
                static_in.copy_(xc)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):                        # warm-up outside capture (weight packing, allocations)
                    for _ in range(2):
                        self._net_eager(static_in, sparse_head)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                try:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        static_out = self._net_eager(static_in, sparse_head)
                    cache[key] = (g, static_in, static_out, getattr(self.net, "head_mode", "dense"))
                except Exception as e:                               # capture unsupported for some solver: stay eager
                    import warnings
                    warnings.warn("HIP graph capture of the network failed (%r); running eagerly" % (e,))
                    self.use_hip_graph = False
                    return self._net_forward_once(x, sparse_head)
            g, static_in, static_out, self._head_mode = cache[key]
            static_in.copy_(xc)
            g.replay()
            ys = static_out
        return tuple(y[0].permute(*(list(range(1, nd + 1)) + [0])) for y in ys)   # (...,C) views

    def _net_eager(self, xc, sparse_head=False):
        import torch
        kw = dict(sparse_head=True) if sparse_head else {}
        with torch.no_grad():
            return tuple(self.net(xc.float(), **kw))

    def _predict_setup(self, img, axes, normalizer, n_tiles):
        import torch
        if n_tiles is None:
            n_tiles = [1] * img.ndim
        try:
            n_tiles = tuple(n_tiles)
            img.ndim == len(n_tiles) or (_ for _ in ()).throw(TypeError())
        except TypeError:
            raise ValueError("n_tiles must be an iterable of length %d" % img.ndim)
        if not all(np.isscalar(t) and 1 <= t and int(t) == t for t in n_tiles):
            raise ValueError("all values of n_tiles must be integer values >= 1")
        n_tiles = tuple(map(int, n_tiles))
        axes = self._normalize_axes(img, axes)
        axes_net = self.config.axes
        # n_tiles is given per IMAGE axis (base.py:418 permutes it like the data): re-order to the net's axes, 1 for an added C
        nt = dict(zip(axes, n_tiles))
        if nt.get("C", 1) != 1:
            raise ValueError("entry of n_tiles > 1 only allowed for axes '%s'" % axes_net.replace("C", ""))
        n_tiles = tuple(nt.get(a, 1) for a in axes_net)
        _permute_axes = self._make_permute_axes(axes, axes_net)
        if normalizer is not None:
            # csbdeep Normalizer protocol: .before(x, axes) on the host array
            x_host = _permute_axes(np.asarray(img))
            x_host = normalizer.before(x_host, axes_net)
            x = to_device(x_host, self.device)
        else:
            x = img if N.is_torch(img) else to_device(img, self.device)
            x = _permute_axes(x.to(self.device))
        channel = axes_dict(axes_net)["C"]
        if self.config.n_channel_in != x.shape[channel]:
            raise ValueError("image has %d channels, model expects %d" % (x.shape[channel], self.config.n_channel_in))
        axes_net_div_by = self._axes_div_by(axes_net)
        grid = tuple(self.config.grid)
        grid_dict = dict(zip(axes_net.replace("C", ""), grid))
        resizer = StarDistPadAndCropResizer(grid=grid_dict)
        if not x.dtype.is_floating_point:
            warnings.warn("Predicting on non-float input... ( forgot to normalize? )")
            x = x.float()
        x = resizer.before(x.float(), axes_net, axes_net_div_by)
        return x, axes, axes_net, axes_net_div_by, resizer, n_tiles, grid, grid_dict, channel

    def _tile_slices(self, x, n_tiles, axes_net, axes_net_div_by):
        """Overlapping tiles (csbdeep tile_iterator role, base.py:412-441): per spatial axis split the padded
        image into n blocks of div_by-aligned size; each tile carries `overlap` context on both sides."""
        overlaps = self._axes_tile_overlap(axes_net)
        per_axis = []
        for a, n, s, db, ov in zip(axes_net, n_tiles, x.shape, axes_net_div_by, overlaps):
            if a == "C" or n == 1:
                per_axis.append([(slice(0, s), slice(0, s), slice(0, s))])
                continue
            nblocks = s // db
            n = min(n, nblocks)
            ovb = int(np.ceil(ov / db)) * db
            edges = [int(round(i * nblocks / n)) * db for i in range(n + 1)]
            lst = []
            for i in range(n):
                d0, d1 = edges[i], edges[i + 1]
                t0, t1 = max(0, d0 - ovb), min(s, d1 + ovb)
                lst.append((slice(t0, t1), slice(d0 - t0, d1 - t0), slice(d0, d1)))   # tile, src (in tile), dst (in image)
            per_axis.append(lst)
        import itertools
        for combo in itertools.product(*per_axis):
            yield tuple(c[0] for c in combo), tuple(c[1] for c in combo), tuple(c[2] for c in combo)

    def _axes_tile_overlap(self, query_axes):
        """base.py:1100-1110 derives this empirically from an impulse response; the analytic receptive field of
        the conv stack (upper bound of the empirical one) is used instead.  tests/test_cpu_reference_predict.py runs the reference's
        own _compute_receptive_field on the graph its own _build makes and holds this bound to it (never smaller; the comparison
        found the stem convolutions of anisotropic grids missing from the bound on the axes a stem round does not pool)."""
        rf = self._receptive_field_radius()
        d = dict(zip(self.config.axes.replace("C", ""), rf))
        return tuple(d.get(a, 0) for a in query_axes)

    def _receptive_field_radius(self):
        cfg = self.config
        nd = cfg.n_dim
        if cfg.backbone == "unet":
            k = cfg.unet_kernel_size; pool = cfg.unet_pool; depth = cfg.unet_n_depth; ncv = cfg.unet_n_conv_per_depth
            out = []
            # the grid stem (model2d.py:317-325, model3d.py:367-375) runs log2(max(grid)) rounds; the convolutions of EVERY round act on
            # every axis, also on the axes a round does not pool (anisotropic grids: (1, 2, 2), (4, 2))
            rounds = int(np.log2(max(cfg.grid))) if max(cfg.grid) > 1 else 0
            for d in range(nd):
                r, scale = 0, 1
                g = cfg.grid[d]
                for _ in range(rounds):                           # pre-pooling stages
                    r += ncv * (k[d] // 2) * scale
                    if scale < g:
                        scale *= 2
                for n in range(depth):
                    r += ncv * (k[d] // 2) * scale; scale *= pool[d]
                r += ncv * (k[d] // 2) * scale
                for n in reversed(range(depth)):
                    scale //= pool[d]; r += ncv * (k[d] // 2) * scale + scale
                r += (k[d] // 2) * scale                          # features conv
                out.append(int(r))
            return tuple(out)
        else:
            k = cfg.resnet_kernel_size
            out = []
            for d in range(nd):
                r, scale = 3 + 1, 1
                g = cfg.grid[d]
                for n in range(cfg.resnet_n_blocks):
                    if scale < g:
                        scale *= 2
                    r += cfg.resnet_n_conv_per_block * (k[d] // 2) * scale
                r += (k[d] // 2) * scale
                out.append(int(r))
            return tuple(out)

    # ------------------------------------------------------------------ predict (dense)   base.py:446-529
    def _predict_generator(self, img, axes=None, normalizer=None, n_tiles=None, show_tile_progress=True, **predict_kwargs):
        import torch
        x, axes, axes_net, axes_net_div_by, resizer, n_tiles, grid, grid_dict, channel = self._predict_setup(img, axes, normalizer, n_tiles)
        if np.prod(n_tiles) > 1:
            sh = [s // grid_dict.get(a, 1) for a, s in zip(axes_net, x.shape)]
            outs = None
            for s_tile, s_src, s_dst in self._tile_slices(x, n_tiles, axes_net, axes_net_div_by):
                res = self._net_forward(x[s_tile])
                g = lambda sl: tuple(slice(None) if a == "C" else slice(s.start // grid_dict.get(a, 1), s.stop // grid_dict.get(a, 1)) for s, a in zip(sl, axes_net))
                if outs is None:
                    outs = [torch.empty(tuple(sh[:channel]) + (r.shape[-1],), dtype=torch.float32, device=self.device) for r in res]
                for o, r in zip(outs, res):
                    o[g(s_dst)] = r[g(s_src)]
                yield
            results = outs
        else:
            results = list(self._net_forward(x))
        prob = results[0][..., 0]
        dist = torch.clamp_min(results[1], 1e-3)           # base.py:512-513
        prob = resizer.after(prob, axes_net.replace("C", ""))
        dist = resizer.after(dist, axes_net)
        if self._is_multiclass():
            yield prob, dist, resizer.after(results[2], axes_net)
        else:
            yield prob, dist

    def predict(self, *args, **kwargs):
        """dense prediction -> (prob, dist[, prob_class]) as numpy arrays (base.py:531-538)"""
        r = None
        for r in self._predict_generator(*args, **kwargs):
            pass
        return tuple(t.cpu().numpy() for t in r)

    # ------------------------------------------------------------------ predict_sparse   base.py:541-633
    def _select(self, prob, dist, prob_thresh, bs):
        """threshold + border + ordered compaction on device (select.hip).  prob (...), dist (..., R) contiguous."""
        import torch
        prob = prob.contiguous(); dist = dist.contiguous()
        nd = prob.dim()
        shape = np.asarray(prob.shape, np.int32)
        b = np.asarray([v for pair in bs for v in pair], np.int32)
        cnt = torch.zeros(1, dtype=torch.int32, device=prob.device)
        R = dist.shape[-1]
        # first pass with a capacity guess, second pass only if it overflowed
        cap = max(1024, int(prob.numel() // 64))
        while True:
            oprob = torch.empty(cap, dtype=torch.float32, device=prob.device)
            odist = torch.empty((cap, R), dtype=torch.float32, device=prob.device)
            opts = torch.empty((cap, nd), dtype=torch.int32, device=prob.device)
            N.dcall(prob, "sd_select_candidates_device", N.tptr(prob), N.tptr(dist), nd, N.ptr(shape), N.ptr(b), R,
                                                        float(np.float32(prob_thresh)), cap, N.tptr(oprob), N.tptr(odist),
                                                        N.tptr(opts), N.tptr(cnt))
            n = int(cnt.item())
            if n <= cap:
                return oprob[:n], odist[:n], opts[:n].to(torch.int64)
            cap = n

    def _select_raw(self, prob, prob_thresh, shape, b):
        """threshold + border + ordered compaction of `prob` alone (select.hip): (n, prob (cap,), grid indices (cap, nd) int32), the first n
        entries valid, in np.where order.  The capacity of the first pass is the previous call's count for this grid size + 25 % (a
        step of a stream of similar images never runs the second pass; the first guess is numel / 64)."""
        import torch
        nd = prob.dim()
        hints = self.__dict__.setdefault("_sel_cap_hint", {})
        cap = hints.get(int(prob.numel()), max(1024, int(prob.numel() // 64)))
        cnt = torch.zeros(1, dtype=torch.int32, device=prob.device)
        while True:
            oprob = torch.empty(cap, dtype=torch.float32, device=prob.device)
            opts = torch.empty((cap, nd), dtype=torch.int32, device=prob.device)
            N.dcall(prob, "sd_select_candidates_device", N.tptr(prob), None, nd, N.ptr(shape), N.ptr(b), 0,
                    float(np.float32(prob_thresh)), cap, N.tptr(oprob), None, N.tptr(opts), N.tptr(cnt))
            n = int(cnt.item())
            if n <= cap:
                break
            cap = n
        hints[int(prob.numel())] = max(1024, n + n // 4)
        return n, oprob, opts

    def _select_sorted(self, prob, feat, prob_thresh, bs):
        """selection -> score order -> distance head, for the step that goes on to the NMS: candidates of `prob` (the whole grid the
        channels-last feature tensor `feat` lives on) above the threshold, sorted by score (stable ascending, reversed: nms.py:167 as
        _argsort_desc states it), with the distance head evaluated on the candidate rows IN THAT ORDER -- the (n, R) distance matrix and
        the point arrays are written once, sorted, instead of being written in np.where order and gathered again behind the sort."""
        import torch
        prob = prob.contiguous()
        nd = prob.dim()
        shape = np.asarray(prob.shape, np.int32)
        b = np.asarray([v for pair in bs for v in pair], np.int32)
        n, oprob, opts = self._select_raw(prob, prob_thresh, shape, b)
        from ..nms import _sort_desc
        sp, order = _sort_desc(oprob[:n])
        rows = torch.empty(n, dtype=torch.int64, device=prob.device)
        pf = torch.empty((n, nd), dtype=torch.float32, device=prob.device)
        pi = torch.empty((n, nd), dtype=torch.int64, device=prob.device)
        if n:
            full = np.asarray(feat.shape[:-1], np.int32)
            grid = np.asarray(self.config.grid, np.int32)
            N.dcall(prob, "sd_sorted_rows_device", N.tptr(opts), N.tptr(order), n, nd, N.ptr(full), None, N.ptr(grid), N.tptr(rows), N.tptr(pf), N.tptr(pi))
        if self._head_mode == "sparse_lazy" and n:
            # the features of the candidates are evaluated in their SPATIAL order (np.where order: neighbouring candidates share the cache lines
            # of their 3x3 neighbourhoods), the distance head reads them in score order
            rows_sp = torch.empty_like(rows)
            rows_sp[order] = rows
            dist = self.net.dist_rows(feat, rows_sp, 1e-3, lazy=True, order=order)
        else:
            dist = self.net.dist_rows(feat, rows, 1e-3, lazy=self._head_mode == "sparse_lazy")           # max(dist, 1e-3): base.py:512-513 / select.hip
        return SortedCandidates(sp, dist, pi, pf)

    def _select_rows(self, prob, feat, origin, prob_thresh, bs):
        """_select for the sparse head: threshold + border + ordered compaction on `prob` (a crop, starting at `origin`, of the
        grid the channels-last feature tensor `feat` (..., C) lives on), then the distance head on the selected rows of feat."""
        import torch
        prob = prob.contiguous()
        nd = prob.dim()
        shape = np.asarray(prob.shape, np.int32)
        b = np.asarray([v for pair in bs for v in pair], np.int32)
        n, oprob, opts = self._select_raw(prob, prob_thresh, shape, b)
        pts = opts[:n].to(torch.int64)
        full = feat.shape[:-1]
        rows = torch.zeros(n, dtype=torch.int64, device=prob.device)
        for d in range(nd):
            rows = rows * int(full[d]) + (pts[:, d] + int(origin[d]))
        odist = self.net.dist_rows(feat, rows, 1e-3, lazy=self._head_mode == "sparse_lazy")          # max(dist, 1e-3): base.py:512-513 / select.hip
        return oprob[:n], odist, pts

    def _predict_sparse_generator(self, img, prob_thresh=None, axes=None, normalizer=None, n_tiles=None,
                                  show_tile_progress=True, b=2, _presort=False, **predict_kwargs):
        import torch
        if prob_thresh is None: prob_thresh = self.thresholds.prob
        x, axes, axes_net, axes_net_div_by, resizer, n_tiles, grid, grid_dict, channel = self._predict_setup(img, axes, normalizer, n_tiles)
        nd = self.config.n_dim
        gridt = torch.tensor(self.config.grid, device=self.device, dtype=torch.int64).reshape(1, nd)
        prob_classa = None
        if np.prod(n_tiles) > 1:
            sh = [s // grid_dict.get(a, 1) for a, s in zip(axes_net, x.shape)]
            pl, dl, ptl, pcl = [], [], [], []
            for s_tile, s_src, s_dst in self._tile_slices(x, n_tiles, axes_net, axes_net_div_by):
                res = self._net_forward(x[s_tile], sparse_head=True)
                g = lambda sl: [slice(s.start // grid_dict.get(a, 1), s.stop // grid_dict.get(a, 1)) for s, a in zip(sl, axes_net) if a != "C"]
                gsrc, gdst = g(s_src), g(s_dst)
                prob_tile = res[0][..., 0][tuple(gsrc)]
                bs = [(b if s.start == 0 else 0, b if s.stop == _sh else 0) for s, _sh in zip(gdst, [v for v, a in zip(sh, axes_net) if a != "C"])]   # base.py:583
                if self._head_mode in ("sparse", "sparse_lazy"):
                    p_, d_, pt_ = self._select_rows(prob_tile, res[1], [s.start for s in gsrc], prob_thresh, bs)
                else:
                    p_, d_, pt_ = self._select(prob_tile, res[1][tuple(gsrc)], prob_thresh, bs)
                off = torch.tensor([s.start for s in gdst], device=self.device, dtype=torch.int64).reshape(1, nd)
                pl.append(p_); dl.append(d_); ptl.append((pt_ + off) * gridt)
                if self._is_multiclass():
                    pc = res[2][tuple(gsrc)].reshape(-1, res[2].shape[-1])
                    lin = pt_[:, 0]
                    for d in range(1, nd): lin = lin * prob_tile.shape[d] + pt_[:, d]
                    pcl.append(pc[lin])
                yield
            proba, dista, pointsa = torch.cat(pl), torch.cat(dl), torch.cat(ptl)
            if self._is_multiclass(): prob_classa = torch.cat(pcl)
        else:
            res = self._net_forward(x, sparse_head=True)
            if (_presort and self._head_mode in ("sparse", "sparse_lazy") and not self._is_multiclass()
                    and not any(p[1] for p in resizer.pad.values())):
                # predict_instances, untiled, nothing padded (filter_points keeps every point): hand the candidates over in score order
                bs = [(b, b)] * self.config.n_dim if np.isscalar(b) else list(b)
                yield self._select_sorted(res[0][..., 0], res[1], prob_thresh, bs)
                return
            yield self._sparse_finish(res, self._head_mode, x, resizer, axes_net, prob_thresh, b)
            return
        idx = resizer.filter_points(x.dim(), pointsa, axes_net)
        proba, dista, pointsa = proba[idx], dista[idx], pointsa[idx]
        if self._is_multiclass():
            yield proba, dista, prob_classa[idx], pointsa
        else:
            yield proba, dista, pointsa

    def _sparse_finish(self, res, head_mode, x, resizer, axes_net, prob_thresh, b):
        """candidate selection behind an (untiled) forward pass: threshold + border + ordered compaction, distance head on the selected rows"""
        import torch
        nd = self.config.n_dim
        gridt = torch.tensor(self.config.grid, device=self.device, dtype=torch.int64).reshape(1, nd)
        prob = res[0][..., 0]
        bs = [(b, b)] * nd if np.isscalar(b) else list(b)
        if head_mode in ("sparse", "sparse_lazy"):
            proba, dista, pts = self._select_rows(prob, res[1], [0] * nd, prob_thresh, bs)
        else:
            proba, dista, pts = self._select(prob, res[1], prob_thresh, bs)
        pointsa = pts * gridt
        prob_classa = None
        if self._is_multiclass():
            pc = res[2].reshape(-1, res[2].shape[-1])
            lin = pts[:, 0]
            for d in range(1, nd): lin = lin * prob.shape[d] + pts[:, d]
            prob_classa = pc[lin]
        idx = resizer.filter_points(x.dim(), pointsa, axes_net)
        proba, dista, pointsa = proba[idx], dista[idx], pointsa[idx]
        if self._is_multiclass():
            return proba, dista, prob_classa[idx], pointsa
        return proba, dista, pointsa

    def predict_sparse(self, *args, **kwargs):
        r = None
        for r in self._predict_sparse_generator(*args, **kwargs):
            pass
        return tuple(t.cpu().numpy() for t in r)

    def predict_sparse_device(self, *args, **kwargs):
        """predict_sparse without the trip to the host: (prob, dist[, prob_class], points) as tensors on self.device"""
        r = None
        for r in self._predict_sparse_generator(*args, **kwargs):
            pass
        return tuple(r)

    # ------------------------------------------------------------------ predict_instances   base.py:645-790
    def _predict_instances_generator(self, img, axes=None, normalizer=None, sparse=True, prob_thresh=None, nms_thresh=None,
                                     scale=None, n_tiles=None, show_tile_progress=True, verbose=False, return_labels=True,
                                     predict_kwargs=None, nms_kwargs=None, overlap_label=None, return_predict=False):
        import torch
        if predict_kwargs is None: predict_kwargs = {}
        if nms_kwargs is None: nms_kwargs = {}
        if return_predict and sparse:
            sparse = False
            warnings.warn("Setting sparse to False because return_predict is True")
        nms_kwargs.setdefault("verbose", verbose)
        _axes = self._normalize_axes(img, axes)
        _axes_net = self.config.axes
        _permute = self._make_permute_axes(_axes, _axes_net)
        _shape_inst = tuple(s for s, a in zip(_permute(np.empty(img.shape, bool) if not N.is_torch(img) else img).shape, _axes_net) if a != "C")
        if scale is not None:
            if isinstance(scale, numbers.Number):
                scale = tuple(scale if a in "XYZ" else 1 for a in _axes)
            scale = tuple(scale)
            if len(scale) != len(_axes):
                raise ValueError("scale %s must be of length %d, i.e. one value for each of the axes %s" % (scale, len(_axes), _axes))
            for s, a in zip(scale, _axes):
                if not s > 0: raise ValueError("scale values must be greater than 0")
            scale = tuple(s if a in "XYZ" else 1 for s, a in zip(scale, _axes))
            from scipy import ndimage as ndi
            img = ndi.zoom(np.asarray(img.cpu() if N.is_torch(img) else img), scale, order=1)     # base.py:725-735
        yield "predict"
        res = None
        if sparse:
            for res in self._predict_sparse_generator(img, axes=axes, normalizer=normalizer, n_tiles=n_tiles,
                                                      prob_thresh=prob_thresh, show_tile_progress=show_tile_progress,
                                                      _presort=(self.device.type == "cuda"), **predict_kwargs):
                if res is None:
                    yield "tile"
            if isinstance(res, SortedCandidates):
                yield "nms"
                yield self._instances_from_sorted(_shape_inst, res, nms_thresh=nms_thresh,
                                                  scale=(None if scale is None else dict(zip(_axes, scale))),
                                                  return_labels=return_labels, overlap_label=overlap_label, **nms_kwargs)
                return
        else:
            for res in self._predict_generator(img, axes=axes, normalizer=normalizer, n_tiles=n_tiles,
                                               show_tile_progress=show_tile_progress, **predict_kwargs):
                if res is None:
                    yield "tile"
            res = tuple(res) + (None,)
        if self._is_multiclass():
            prob, dist, prob_class, points = res
        else:
            prob, dist, points = res
            prob_class = None
        yield "nms"
        res_instances = self._instances_from_prediction(_shape_inst, prob, dist, points=points, prob_class=prob_class,
                                                        prob_thresh=prob_thresh, nms_thresh=nms_thresh,
                                                        scale=(None if scale is None else dict(zip(_axes, scale))),
                                                        return_labels=return_labels, overlap_label=overlap_label, **nms_kwargs)
        if return_predict:
            yield res_instances, tuple(t.cpu().numpy() for t in res[:-1])
        else:
            yield res_instances

    def predict_instances_big(self, img, axes, block_size, min_overlap, context=None, labels_out=None, labels_out_dtype=np.int32,
                              show_progress=True, **kwargs):
        """Predict instances of very large inputs block by block (base.py:838-983).  When torch.distributed is
        initialised the blocks are sharded round-robin over the ranks (one process per GPU); see stardist_amd/big.py."""
        from ..big import predict_instances_big
        return predict_instances_big(self, img, axes, block_size, min_overlap, context=context, labels_out=labels_out,
                                     labels_out_dtype=labels_out_dtype, show_progress=show_progress, **kwargs)

    def predict_instances_sharded(self, img, axes, block_size, min_overlap, context=None, **kwargs):
        """Block-sharded prediction over the ranks of torch.distributed with a final cross-tile NMS on rank 0
        (SURVEY.md 8e design A); see stardist_amd/big.py::predict_instances_sharded."""
        from ..big import predict_instances_sharded
        return predict_instances_sharded(self, img, axes, block_size, min_overlap, context=context, **kwargs)

    def predict_instances_iter(self, imgs, prefetch=1, **kwargs):
        """predict_instances over a sequence of HOST arrays, yielding (labels, dict) per image in order, with the upload of image k + 1
        overlapped with the step on image k (the reference reads block k + 1 while it works on block k only in its big-image loop,
        stardist/big.py:312-326; a plain loop over predict_instances pays the host -> device copy of every input in front of its step).
        A helper thread copies the next array into page-locked memory (numpy copy: releases the GIL) and enqueues the device copy on
        its own stream; the step waits for that copy's event only.  Images that need host-side preparation (a `normalizer`, `scale`)
        and device tensors are passed through unchanged.  Results are those of predict_instances(img, **kwargs), image by image."""
        import queue
        import threading
        import torch
        if self.device.type != "cuda" or kwargs.get("normalizer") is not None or kwargs.get("scale") is not None:
            for img in imgs:
                yield self.predict_instances(img, **kwargs)
            return
        dev = self.device
        copy_stream = torch.cuda.Stream(device=dev)
        q = queue.Queue(maxsize=max(1, int(prefetch)))
        stop = threading.Event()

        def put(item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    continue
            return False

        def uploader():
            try:
                torch.cuda.set_device(dev)              # HIP's current device is per thread: page-locked blocks belong to THIS model's GPU
                for img in imgs:
                    if stop.is_set():
                        return
                    if N.is_torch(img) or not isinstance(img, np.ndarray) or img.dtype == object:
                        if not put((img, None)):
                            return
                        continue
                    a = np.ascontiguousarray(img)
                    # a ring of page-locked staging blocks per (shape, dtype), kept on the model: allocating page-locked memory per image
                    # costs milliseconds and synchronises the device; a block is re-used once the copy that read it has completed
                    ring = self.__dict__.setdefault("_upload_ring", {}).setdefault((a.shape, a.dtype.str), [])
                    slot = None
                    for cand in ring:
                        if cand[1] is None or cand[1].query():
                            slot = cand
                            break
                    if slot is None:
                        if len(ring) >= int(prefetch) + 2:
                            slot = ring[0]
                            slot[1].synchronize()
                        else:
                            slot = [torch.empty(a.shape, dtype=torch.from_numpy(a[:0].reshape(-1)).dtype, pin_memory=True), None]
                            ring.append(slot)
                    ring[:] = [c for c in ring if c is not slot] + [slot]     # least recently used first
                    np.copyto(slot[0].numpy(), a)
                    with torch.cuda.device(dev), torch.cuda.stream(copy_stream):
                        t = slot[0].to(dev, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(copy_stream)
                    slot[1] = ev
                    if not put((t, (ev, slot))):
                        return
                put((None, "end"))
            except BaseException as e:                      # noqa: BLE001 -- re-raised in the consumer
                put((e, "error"))

        th = threading.Thread(target=uploader, name="stardist_amd-upload", daemon=True)
        th.start()
        try:
            while True:
                item, tag = q.get()
                if tag == "end":
                    break
                if tag == "error":
                    raise item
                if tag is not None:
                    ev, _stage = tag
                    torch.cuda.current_stream(dev).wait_event(ev)
                    item.record_stream(torch.cuda.current_stream(dev))
                yield self.predict_instances(item, **kwargs)
        finally:
            stop.set()
            th.join(timeout=5.0)

    def predict_instances(self, *args, **kwargs):
        """Predict instance segmentation: returns (labels, dict) exactly like the reference (base.py:775-790)."""
        r = None
        for r in self._predict_instances_generator(*args, **kwargs):
            pass
        return r
