"""Config2D / Config3D: the prediction-relevant part of the reference's configuration schema
(stardist/models/model2d.py:123-269, model3d.py:129-311), json-compatible with the
reference's `config.json` (extra training keys are kept verbatim but unused)."""
import json

import numpy as np

from ..utils import _normalize_grid


class BaseConfig(object):
    """csbdeep.models.BaseConfig surface used by the reference: attribute bag + update_parameters."""

    def __init__(self, axes, n_channel_in, n_channel_out):
        # csbdeep BaseConfig.__init__: axes out of STCZYX without repeats, X and Y present, Z and T not together, a sample axis first
        # (and dropped), the channel axis last (channels_last is the one backend layout: models/__init__.py:8-14) and added when absent
        axes = str(axes).upper()
        if any(a not in "STCZYX" for a in axes) or any(axes.count(a) > 1 for a in axes):
            raise ValueError("invalid axes '%s': every axis must be one of 'STCZYX' and occur once" % axes)
        if not ("X" in axes and "Y" in axes):
            raise ValueError("lateral axes X and Y must be present.")
        if "Z" in axes and "T" in axes:
            raise ValueError("using Z and T axes together not supported.")
        if "S" in axes and not axes.startswith("S"):
            raise ValueError("sample axis S must be first.")
        axes = axes.replace("S", "")
        if "C" in axes:
            if axes[-1] != "C":
                raise ValueError("channel axis must be last for backend (channels_last).")
        else:
            axes = axes + "C"
        self.n_dim = len(axes) - 1
        self.axes = axes
        self.n_channel_in = int(max(1, n_channel_in))
        self.n_channel_out = int(max(1, n_channel_out))
        # csbdeep's BaseConfig keys as every config.json of the reference carries them (models/examples/*/config.json)
        self.train_checkpoint = "weights_best.h5"
        self.train_checkpoint_last = "weights_last.h5"
        self.train_checkpoint_epoch = "weights_now.h5"

    def _training_schema(self, patch_size, batch_size, shape_completion):
        """The training keys of the reference's schema with its defaults (model2d.py:235-257, model3d.py:274-294): unused by the
        prediction path, kept so that a configuration built with the reference's keyword arguments is accepted, validated and
        written to config.json exactly as the reference writes it."""
        if shape_completion:
            self.train_shape_completion = False
            self.train_completion_crop = 32
        self.train_patch_size = patch_size
        self.train_background_reg = 1e-4
        self.train_foreground_only = 0.9
        self.train_sample_cache = True
        self.train_dist_loss = "mae"
        self.train_loss_weights = (1, 0.2) if self.n_classes is None else (1, 0.2, 1)
        self.train_class_weights = (1, 1) if self.n_classes is None else (1,) * (self.n_classes + 1)
        self.train_epochs = 400
        self.train_steps_per_epoch = 100
        self.train_learning_rate = 0.0003
        self.train_batch_size = batch_size
        self.train_n_val_patches = None
        self.train_tensorboard = True
        self.train_reduce_lr = {"factor": 0.5, "patience": 40, "min_delta": 0}
        self.use_gpu = False

    def _check_training_weights(self):
        """model2d.py:266-270, model3d.py:306-310"""
        if not len(self.train_loss_weights) == (2 if self.n_classes is None else 3):
            raise ValueError("train_loss_weights %s not compatible with n_classes (%s): must be 3 weights if n_classes is not None, otherwise 2"
                             % (self.train_loss_weights, self.n_classes))
        if not len(self.train_class_weights) == (2 if self.n_classes is None else self.n_classes + 1):
            raise ValueError("train_class_weights %s not compatible with n_classes (%s): must be 'n_classes + 1' weights if n_classes is not None, otherwise 2"
                             % (self.train_class_weights, self.n_classes))

    def is_valid(self, return_invalid=False):
        """csbdeep BaseConfig.is_valid: the base schema has nothing to refuse (the constructors above raise instead)"""
        return (True, tuple()) if return_invalid else True

    def __repr__(self):
        """argparse.Namespace's, which csbdeep's BaseConfig inherits: ClassName(key=value, ...) in attribute order"""
        return "%s(%s)" % (type(self).__name__, ", ".join("%s=%r" % kv for kv in vars(self).items()))

    def __eq__(self, other):
        return isinstance(other, BaseConfig) and vars(self) == vars(other)

    __hash__ = None

    def update_parameters(self, allow_new=False, **kwargs):
        if not allow_new:
            unknown = [k for k in kwargs if not hasattr(self, k)]
            # the reference raises AttributeError for unknown keys; json files of other versions may
            # carry extra training keys, which are accepted silently there too (allow_new on load)
            if unknown:
                raise AttributeError("Not allowed to add new parameters (%s)" % ", ".join(unknown))
        for k, v in kwargs.items():
            setattr(self, k, v)

    def to_json(self):
        def conv(v):
            if isinstance(v, (np.integer,)): return int(v)
            if isinstance(v, (np.floating,)): return float(v)
            if isinstance(v, (tuple, list)): return [conv(x) for x in v]
            return v
        return json.dumps({k: conv(v) for k, v in vars(self).items()})

    @classmethod
    def from_json(cls, path_or_dict):
        if isinstance(path_or_dict, dict):
            d = dict(path_or_dict)
        else:
            with open(path_or_dict) as fh:
                d = dict(json.load(fh))
        init_keys = {k: d.pop(k) for k in list(d) if k in ("axes", "n_rays", "n_channel_in", "grid", "n_classes", "backbone", "rays", "anisotropy")}
        if "rays_json" in d and "rays" not in init_keys and cls.__name__ == "Config3D":
            from ..rays3d import rays_from_json
            init_keys["rays"] = rays_from_json(d["rays_json"])
        ax = init_keys.get("axes")
        if ax is not None:
            init_keys["axes"] = ax.replace("C", "")
        cfg = cls(**init_keys)
        for k in ("n_dim", "n_channel_out", "axes"):
            d.pop(k, None)
        cfg.update_parameters(True, **d)
        for k in ("grid", "unet_kernel_size", "unet_pool", "resnet_kernel_size"):
            if hasattr(cfg, k) and isinstance(getattr(cfg, k), list):
                setattr(cfg, k, tuple(getattr(cfg, k)))
        return cfg


class Config2D(BaseConfig):
    """model2d.py:198-269 (prediction-relevant defaults)."""

    def __init__(self, axes="YX", n_rays=32, n_channel_in=1, grid=(1, 1), n_classes=None, backbone="unet", **kwargs):
        super().__init__(axes=axes, n_channel_in=n_channel_in, n_channel_out=1 + n_rays)
        self.n_rays = int(n_rays)
        self.grid = _normalize_grid(grid, 2)
        self.backbone = str(backbone).lower()
        self.n_classes = None if n_classes is None else int(n_classes)
        if self.backbone == "unet":
            self.unet_n_depth = 3
            self.unet_kernel_size = 3, 3
            self.unet_n_filter_base = 32
            self.unet_n_conv_per_depth = 2
            self.unet_pool = 2, 2
            self.unet_activation = "relu"
            self.unet_last_activation = "relu"
            self.unet_batch_norm = False
            self.unet_dropout = 0.0
            self.unet_prefix = ""
            self.net_conv_after_unet = 128
        else:
            raise ValueError("backbone '%s' not supported." % self.backbone)
        self.net_input_shape = None, None, self.n_channel_in
        self.net_mask_shape = None, None, 1
        self._training_schema((256, 256), 4, shape_completion=True)
        for k in ("n_dim", "n_channel_out"):
            kwargs.pop(k, None)
        self.update_parameters(False, **kwargs)
        self._check_training_weights()


class Config3D(BaseConfig):
    """model3d.py:207-311 (prediction-relevant defaults)."""

    def __init__(self, axes="ZYX", rays=None, n_channel_in=1, grid=(1, 1, 1), n_classes=None, anisotropy=None,
                 backbone="unet", **kwargs):
        from ..rays3d import Rays_GoldenSpiral, rays_from_json
        if rays is None:                                  # model3d.py:209-217 (a configuration dictionary read back: Config3D(**config_dict))
            if "rays_json" in kwargs:
                rays = rays_from_json(kwargs["rays_json"])
            elif "n_rays" in kwargs:
                rays = Rays_GoldenSpiral(kwargs["n_rays"])
            else:
                rays = Rays_GoldenSpiral(96)
        elif np.isscalar(rays):
            rays = Rays_GoldenSpiral(rays)
        super().__init__(axes=axes, n_channel_in=n_channel_in, n_channel_out=1 + len(rays))
        self.n_rays = len(rays)
        self.grid = _normalize_grid(grid, 3)
        self.anisotropy = anisotropy if anisotropy is None else tuple(anisotropy)
        self.backbone = str(backbone).lower()
        import copy
        self.rays_json = copy.deepcopy(rays.to_json())            # (to_json hands out the ray set's own kwargs: the edit below must not reach a shared instance)
        self.n_classes = None if n_classes is None else int(n_classes)
        if "anisotropy" in self.rays_json["kwargs"]:
            if self.rays_json["kwargs"]["anisotropy"] is None and self.anisotropy is not None:
                self.rays_json["kwargs"]["anisotropy"] = self.anisotropy
        if self.backbone == "unet":
            self.unet_n_depth = 2
            self.unet_kernel_size = 3, 3, 3
            self.unet_n_filter_base = 32
            self.unet_n_conv_per_depth = 2
            self.unet_pool = 2, 2, 2
            self.unet_activation = "relu"
            self.unet_last_activation = "relu"
            self.unet_batch_norm = False
            self.unet_dropout = 0.0
            self.unet_prefix = ""
            self.net_conv_after_unet = 128
        elif self.backbone == "resnet":
            self.resnet_n_blocks = 4
            self.resnet_kernel_size = 3, 3, 3
            self.resnet_kernel_init = "he_normal"
            self.resnet_n_filter_base = 32
            self.resnet_n_conv_per_block = 3
            self.resnet_activation = "relu"
            self.resnet_batch_norm = False
            self.net_conv_after_resnet = 128
        else:
            raise ValueError("backbone '%s' not supported." % self.backbone)
        self.net_input_shape = None, None, None, self.n_channel_in
        self.net_mask_shape = None, None, None, 1
        self._training_schema((128, 128, 128), 1, shape_completion=False)
        for k in ("n_dim", "n_channel_out", "n_rays", "rays_json"):
            kwargs.pop(k, None)
        self.update_parameters(False, **kwargs)
        self._check_training_weights()
