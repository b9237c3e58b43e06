"""StarDist3D (prediction API) -- mirror of stardist/models/model3d.py:314-695 for the hot path."""
import numpy as np

from ..utils import to_host

from ..geometry.geom3d import polyhedron_to_label
from ..lib import _native as N
from ..matching import relabel_sequential
from ..nms import non_maximum_suppression_3d, non_maximum_suppression_3d_sparse
from ..rays3d import rays_from_json
from .base import StarDistBase, axes_check_and_normalize
from .config import Config3D


class StarDist3D(StarDistBase):
    """StarDist3D model: `predict_instances(vol)` -> (labels (Z,Y,X), dict(dist, points, prob, rays, ...))."""

    def __init__(self, config=Config3D(), name=None, basedir=".", **kwargs):
        if config is None and (basedir is None):                 # (nothing to load a configuration from)
            config = Config3D()
        super().__init__(config, name=name, basedir=basedir, **kwargs)

    def _build(self):
        from .unet import StarDistNet
        if self.config.backbone not in ("unet", "resnet"):
            raise NotImplementedError(self.config.backbone)
        return StarDistNet(self.config)

    def _instances_from_prediction(self, img_shape, prob, dist, points=None, prob_class=None, prob_thresh=None,
                                   nms_thresh=None, overlap_label=None, return_labels=True, scale=None, **nms_kwargs):
        """model3d.py:589-674"""
        if prob_thresh is None: prob_thresh = self.thresholds.prob
        if nms_thresh is None: nms_thresh = self.thresholds.nms
        rays = rays_from_json(self.config.rays_json)
        if points is not None:
            points, probi, disti, indsi = non_maximum_suppression_3d_sparse(dist, prob, points, rays, nms_thresh=nms_thresh, **nms_kwargs)
            if prob_class is not None:
                prob_class = prob_class[indsi]
        else:
            tonp = (lambda t: t.cpu().numpy()) if N.is_torch(prob) else (lambda t: t)
            points, probi, disti = non_maximum_suppression_3d(tonp(dist), tonp(prob), rays, grid=self.config.grid,
                                                              prob_thresh=prob_thresh, nms_thresh=nms_thresh, **nms_kwargs)
            if prob_class is not None:
                inds = tuple(p // g for p, g in zip(points.T, self.config.grid))
                prob_class = tonp(prob_class)[inds]
        verbose = nms_kwargs.get("verbose", False)
        verbose and print("render polygons...")
        return self._instances_from_survivors(img_shape, points, probi, disti, prob_class=prob_class, return_labels=return_labels, scale=scale,
                                              overlap_label=overlap_label, verbose=verbose, rays=rays)

    def _instances_from_sorted(self, img_shape, cand, nms_thresh=None, overlap_label=None, return_labels=True, scale=None, **nms_kwargs):
        """_instances_from_prediction for candidates the selection already delivers in score order on the device (base.SortedCandidates)"""
        from ..nms import non_maximum_suppression_3d_sparse_sorted
        if nms_thresh is None: nms_thresh = self.thresholds.nms
        rays = rays_from_json(self.config.rays_json)
        idx = non_maximum_suppression_3d_sparse_sorted(cand.dist, cand.prob, cand.points_f32, rays, nms_thresh=nms_thresh, **nms_kwargs)
        verbose = nms_kwargs.get("verbose", False)
        verbose and print("render polygons...")
        return self._instances_from_survivors(img_shape, cand.points.index_select(0, idx), cand.prob.index_select(0, idx),
                                              cand.dist.index_select(0, idx), return_labels=return_labels, scale=scale,
                                              overlap_label=overlap_label, verbose=verbose, rays=rays)

    def _instances_from_survivors(self, img_shape, points, probi, disti, prob_class=None, return_labels=True, scale=None, overlap_label=None,
                                  verbose=False, rays=None, window=None):
        """the part of model3d.py:616-674 behind the NMS: label volume (polyhedron rasteriser + relabel_sequential) and result dict of
        survivors given best score first.  window = ((z0, y0, x0), (nz, ny, nx)): only that part of the volume is rendered, with the
        polyhedra's running numbers as labels (the block-sharded prediction compacts them globally, stardist_amd/big.py)."""
        if rays is None:
            rays = rays_from_json(self.config.rays_json)
        if scale is not None:
            if not (isinstance(scale, dict) and "X" in scale and "Y" in scale and "Z" in scale):
                raise ValueError("scale must be a dictionary with entries for 'X', 'Y', and 'Z'")
            rescale = (1 / scale["Z"], 1 / scale["Y"], 1 / scale["X"])
            if N.is_torch(points):
                import torch
                points = points * torch.tensor(rescale, device=points.device, dtype=torch.float64).reshape(1, 3)
            else:
                points = points * np.array(rescale).reshape(1, 3)
            rays = rays.copy(scale=rescale)
        if window is not None:
            return polyhedron_to_label(disti, points, rays=rays, prob=probi, shape=img_shape, overlap_label=overlap_label, verbose=verbose,
                                       window=window), None
        if return_labels:
            labels = polyhedron_to_label(disti, points, rays=rays, prob=probi, shape=img_shape, overlap_label=overlap_label, verbose=verbose)
            if N.is_torch(labels):
                if overlap_label is not None and overlap_label < 0 and bool((labels == overlap_label).any()):
                    overlap_mask = (labels == overlap_label)
                    overlap_label2 = int(labels.max()) + 1
                    labels[overlap_mask] = overlap_label2
                    labels, fwd, bwd = relabel_sequential(labels)
                    labels[labels == fwd[overlap_label2]] = overlap_label
                elif overlap_label is None:
                    labels, _, _ = relabel_sequential(labels, _known_max=len(points))     # the rasteriser wrote ids 1..M (geom3d.py:141)
                else:
                    labels, _, _ = relabel_sequential(labels)
                labels = to_host(labels)
            else:
                if overlap_label is not None and overlap_label < 0 and (overlap_label in labels):
                    overlap_mask = (labels == overlap_label)
                    overlap_label2 = max(set(np.unique(labels)) - {overlap_label}) + 1
                    labels[overlap_mask] = overlap_label2
                    labels, fwd, bwd = relabel_sequential(labels)
                    labels[labels == fwd[overlap_label2]] = overlap_label
                else:
                    labels, _, _ = relabel_sequential(labels)
        else:
            labels = None
        if N.is_torch(disti) and N.is_torch(points) and N.is_torch(probi):
            from ..utils import to_host_many
            dist_h, points_h, prob_h = to_host_many([disti, points, probi])          # one synchronisation for the three
            to_np = lambda t: t.cpu().numpy()
        else:
            to_np = (lambda t: t.cpu().numpy()) if N.is_torch(disti) else (lambda t: t)
            dist_h, points_h, prob_h = to_np(disti), to_np(points), to_np(probi)
        res_dict = dict(dist=dist_h, points=points_h, prob=prob_h, rays=rays, rays_vertices=rays.vertices,
                        rays_faces=rays.faces)
        if prob_class is not None:
            prob_class = np.asarray(to_np(prob_class) if N.is_torch(prob_class) else prob_class)
            res_dict.update(dict(class_prob=prob_class, class_id=np.argmax(prob_class, axis=-1)))
        return labels, res_dict

    def _nms_sparse(self, dist, prob, points, nms_thresh=None, **nms_kwargs):
        """indices (into the given candidates) of the NMS survivors, best score first (used by the sharded predictor)"""
        if nms_thresh is None: nms_thresh = self.thresholds.nms
        rays = rays_from_json(self.config.rays_json)
        inds = non_maximum_suppression_3d_sparse(dist, prob, points, rays, nms_thresh=nms_thresh, **nms_kwargs)[3]
        return inds.cpu().numpy() if N.is_torch(inds) else np.asarray(inds)

    def _nms_sparse_device(self, dist, prob, points, nms_thresh=None, **nms_kwargs):
        """NMS of a candidate list given as device tensors: (points, prob, dist) of the survivors, best score first, still on
        the device (used by the sharded predictor: no host round trip between selection, local NMS and the RCCL gather)"""
        if nms_thresh is None: nms_thresh = self.thresholds.nms
        rays = rays_from_json(self.config.rays_json)
        r = non_maximum_suppression_3d_sparse(dist, prob, points, rays, nms_thresh=nms_thresh, **nms_kwargs)
        return r[0], r[1], r[2], r[3]

    def _axes_div_by(self, query_axes):
        """model3d.py:677-690"""
        if self.config.backbone == "unet":
            query_axes = axes_check_and_normalize(query_axes)
            div_by = dict(zip(self.config.axes.replace("C", ""),
                              tuple(p ** self.config.unet_n_depth * g for p, g in zip(self.config.unet_pool, self.config.grid))))
            return tuple(div_by.get(a, 1) for a in query_axes)
        elif self.config.backbone == "resnet":
            grid_dict = dict(zip(self.config.axes.replace("C", ""), self.config.grid))
            return tuple(grid_dict.get(a, 1) for a in query_axes)
        raise NotImplementedError()

    @property
    def _config_class(self):
        return Config3D
