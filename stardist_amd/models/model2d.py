"""StarDist2D (prediction API) -- mirror of stardist/models/model2d.py:272-593 for the hot path."""
import numpy as np

from ..utils import to_host

from ..geometry.geom2d import dist_to_coord, polygons_to_label
from ..lib import _native as N
from ..nms import non_maximum_suppression, non_maximum_suppression_sparse
from .base import StarDistBase, axes_check_and_normalize
from .config import Config2D


class StarDist2D(StarDistBase):
    """StarDist2D model: `predict_instances(img)` -> (labels int32 (H,W), dict(coord, points, prob))."""

    def __init__(self, config=Config2D(), name=None, basedir=".", **kwargs):
        super().__init__(config, name=name, basedir=basedir, **kwargs)

    def _build(self):
        from .unet import StarDistNet
        if self.config.backbone != "unet":
            raise NotImplementedError()
        return StarDistNet(self.config)

    def _instances_from_prediction(self, img_shape, prob, dist, points=None, prob_class=None, prob_thresh=None,
                                   nms_thresh=None, overlap_label=None, return_labels=True, scale=None, **nms_kwargs):
        """model2d.py:512-563. Works on device tensors (from predict*) or numpy arrays; returns numpy."""
        if prob_thresh is None: prob_thresh = self.thresholds.prob
        if nms_thresh is None: nms_thresh = self.thresholds.nms
        if overlap_label is not None:
            raise NotImplementedError("overlap_label not supported for 2D yet!")
        if points is not None:
            points, probi, disti, indsi = non_maximum_suppression_sparse(dist, prob, points, nms_thresh=nms_thresh, **nms_kwargs)
            if prob_class is not None:
                prob_class = prob_class[indsi]
        else:
            points, probi, disti = non_maximum_suppression(dist, prob, grid=self.config.grid, prob_thresh=prob_thresh,
                                                           nms_thresh=nms_thresh, **nms_kwargs)
            if prob_class is not None:
                inds = tuple(p // g for p, g in zip(points.T, self.config.grid))
                prob_class = prob_class[inds]
        return self._instances_from_survivors(img_shape, points, probi, disti, prob_class=prob_class, return_labels=return_labels, scale=scale)

    def _instances_from_sorted(self, img_shape, cand, nms_thresh=None, overlap_label=None, return_labels=True, scale=None, **nms_kwargs):
        """_instances_from_prediction for candidates the selection already delivers in score order on the device (base.SortedCandidates):
        the NMS natives take them as they are, the survivors' rows are gathered once"""
        from ..nms import non_maximum_suppression_sparse_sorted
        if nms_thresh is None: nms_thresh = self.thresholds.nms
        if overlap_label is not None:
            raise NotImplementedError("overlap_label not supported for 2D yet!")
        import torch
        if scale is None and cand.dist.is_cuda and cand.dist.dtype == torch.float32 and cand.points.dtype == torch.int64 and len(cand.prob):
            # the keep flags straight into the survivors' rows, coordinates and painting order (csrc/survivors.hip: two native calls and one
            # read-back where the generic path below issues a dozen framework launches), then the rasteriser and ONE transfer to the host
            from ..lib.stardist2d import survivors_of_sorted, c_polygons_to_label
            from ..nms import nms_keep_sorted
            from ..utils import to_host_many
            keep = nms_keep_sorted(cand.dist, cand.prob, cand.points_f32, nms_thresh=nms_thresh, **nms_kwargs)
            probi, points, coord, cpaint, lpaint = survivors_of_sorted(keep, cand.prob, cand.points, cand.dist, want_paint=return_labels)
            labels = c_polygons_to_label(cpaint, lpaint, img_shape) if return_labels else None
            labels, coord_h, points_h, prob_h = to_host_many([labels, coord, points, probi])
            return labels, dict(coord=coord_h, points=points_h, prob=prob_h)
        idx = non_maximum_suppression_sparse_sorted(cand.dist, cand.prob, cand.points_f32, nms_thresh=nms_thresh, **nms_kwargs)
        return self._instances_from_survivors(img_shape, cand.points.index_select(0, idx), cand.prob.index_select(0, idx),
                                              cand.dist.index_select(0, idx), return_labels=return_labels, scale=scale)

    def _instances_from_survivors(self, img_shape, points, probi, disti, prob_class=None, return_labels=True, scale=None, window=None):
        """the part of model2d.py:536-563 behind the NMS: label image and result dict of survivors given best score first.
        window = ((y0, x0), (h, w)): only that part of the label image is rendered (block-sharded prediction, stardist_amd/big.py)."""
        if scale is not None:
            if not (isinstance(scale, dict) and "X" in scale and "Y" in scale):
                raise ValueError("scale must be a dictionary with entries for 'X' and 'Y'")
            rescale = (1 / scale["Y"], 1 / scale["X"])
            if N.is_torch(points):
                import torch
                points = points * torch.tensor(rescale, device=points.device, dtype=torch.float64).reshape(1, 2)
            else:
                points = points * np.array(rescale).reshape(1, 2)
        else:
            rescale = (1, 1)
        if window is None and N.is_torch(disti) and N.is_torch(points) and N.is_torch(probi) and disti.is_cuda:
            # device tensors, whole image: the polygons' coordinates are computed ONCE (polygons_to_label would compute them again in
            # painting order) and everything returns to the host behind one synchronisation
            import torch
            from ..geometry.geom2d import polygons_to_label_coord
            from ..utils import to_host_many
            coord = dist_to_coord(disti, points, scale_dist=rescale)
            labels = None
            if return_labels:
                # geom2d.py:186-197 with prob given and thr = -inf: `prob > thr` holds for every finite score; paint in ascending
                # score order (stable), label id = position in the given (NMS) order + 1
                ind = torch.sort(probi, stable=True)[1]
                labels = polygons_to_label_coord(coord.index_select(0, ind), shape=img_shape, labels=ind)
            pc = prob_class if (prob_class is not None and N.is_torch(prob_class)) else None
            labels, coord_h, points_h, prob_h, pc_h = to_host_many([labels, coord, points, probi, pc])
            res_dict = dict(coord=coord_h, points=points_h, prob=prob_h)
            if prob_class is not None:
                prob_class = np.asarray(pc_h if pc is not None else prob_class)
                res_dict.update(dict(class_prob=prob_class, class_id=np.argmax(prob_class, axis=-1)))
            return labels, res_dict
        if return_labels:
            labels = polygons_to_label(disti, points, prob=probi, shape=img_shape, scale_dist=rescale, window=window)
        else:
            labels = None
        if window is not None:
            return labels, None
        coord = dist_to_coord(disti, points, scale_dist=rescale)
        to_np = (lambda t: t.cpu().numpy()) if N.is_torch(coord) else (lambda t: t)
        if labels is not None and N.is_torch(labels):
            labels = to_host(labels)
        res_dict = dict(coord=to_np(coord), points=to_np(points), prob=to_np(probi))
        if prob_class is not None:
            prob_class = np.asarray(to_np(prob_class) if N.is_torch(prob_class) else prob_class)
            class_id = np.argmax(prob_class, axis=-1)
            res_dict.update(dict(class_prob=prob_class, class_id=class_id))
        return labels, res_dict

    def _nms_sparse(self, dist, prob, points, nms_thresh=None, **nms_kwargs):
        """indices (into the given candidates) of the NMS survivors, best score first (used by the sharded predictor)"""
        if nms_thresh is None: nms_thresh = self.thresholds.nms
        inds = non_maximum_suppression_sparse(dist, prob, points, nms_thresh=nms_thresh, **nms_kwargs)[3]
        return inds.cpu().numpy() if N.is_torch(inds) else np.asarray(inds)

    def _nms_sparse_device(self, dist, prob, points, nms_thresh=None, **nms_kwargs):
        """NMS of a candidate list given as device tensors: (points, prob, dist) of the survivors, best score first, still on
        the device (used by the sharded predictor: no host round trip between selection, local NMS and the RCCL gather)"""
        if nms_thresh is None: nms_thresh = self.thresholds.nms
        r = non_maximum_suppression_sparse(dist, prob, points, nms_thresh=nms_thresh, **nms_kwargs)
        return r[0], r[1], r[2], r[3]

    def _axes_div_by(self, query_axes):
        """model2d.py:566-574"""
        query_axes = axes_check_and_normalize(query_axes)
        div_by = dict(zip(self.config.axes.replace("C", ""),
                          tuple(p ** self.config.unet_n_depth * g for p, g in zip(self.config.unet_pool, self.config.grid))))
        return tuple(div_by.get(a, 1) for a in query_axes)

    @property
    def _config_class(self):
        return Config2D
