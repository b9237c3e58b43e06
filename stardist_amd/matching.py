"""stardist/matching.py.  relabel_sequential is the one function of it that sits on the prediction path (model3d.py:634-645,
base.py:959): device version for torch label volumes, numpy version otherwise.  The detection metrics (matching, matching_dataset, ...)
follow further down: host-side numpy / scipy, off the hot path, equal to the reference's field by field (tests/test_cpu_vs_reference_source.py)."""
import numpy as np

from .lib import _native as N


def relabel_sequential(label_field, offset=1, _known_max=None):
    """matching.py:319-408: relabel arbitrary labels to {offset, ..., offset+n_labels-1}; 0 stays background.
    Returns (relabeled, forward_map, inverse_map).
    _known_max (device tensors, internal): the caller guarantees 0 <= labels <= _known_max (the rasteriser's output for M polyhedra
    without an overlap label); the relabelled volume is then computed without any read-back and the maps are returned as None."""
    offset = int(offset)
    if offset <= 0:
        raise ValueError("Offset must be strictly positive.")
    if N.is_torch(label_field) and _known_max is not None:
        import torch
        li = label_field.reshape(-1).long()
        present = torch.zeros(int(_known_max) + 1, dtype=torch.bool, device=label_field.device)
        present[li] = True
        present[0] = False
        fwd = torch.cumsum(present, 0, dtype=label_field.dtype) + (offset - 1)
        fwd = torch.where(present, fwd, torch.zeros_like(fwd))
        return fwd[li].reshape(label_field.shape), None, None
    if N.is_torch(label_field):
        import torch
        if int(label_field.min()) < 0:
            raise ValueError("Cannot relabel array that contains negative values.")
        max_label = int(label_field.max())
        present = torch.zeros(max_label + 1, dtype=torch.bool, device=label_field.device)
        present[label_field.reshape(-1).long()] = True
        present[0] = False
        labels0 = torch.where(present)[0]
        forward_map = torch.zeros(max_label + 1, dtype=label_field.dtype, device=label_field.device)
        forward_map[labels0] = torch.arange(offset, offset + len(labels0), device=label_field.device, dtype=label_field.dtype)
        inverse_map = torch.zeros(offset + len(labels0), dtype=label_field.dtype, device=label_field.device)
        inverse_map[offset:] = labels0.to(label_field.dtype)
        return forward_map[label_field.long()], forward_map, inverse_map
    if np.min(label_field) < 0:
        raise ValueError("Cannot relabel array that contains negative values.")
    max_label = int(label_field.max())
    if not np.issubdtype(label_field.dtype, np.integer):
        label_field = label_field.astype(np.min_scalar_type(max_label))
    labels0 = np.unique(label_field)
    labels0 = labels0[labels0 != 0]
    new_max = offset - 1 + len(labels0)
    out_type = label_field.dtype
    need = np.min_scalar_type(new_max)
    if np.dtype(need).itemsize > np.dtype(out_type).itemsize:
        out_type = need
    forward_map = np.zeros(max_label + 1, dtype=out_type)
    forward_map[labels0] = np.arange(offset, new_max + 1)
    inverse_map = np.zeros(new_max + 1, dtype=out_type)
    inverse_map[offset:] = labels0
    return forward_map[label_field], forward_map, inverse_map


# ----------------------------------------------------------------------------------------------------------------------------------
# Detection / segmentation metrics (stardist/matching.py:13-316, 409-483).  Host-side numpy + scipy, NOT part of the prediction path
# (SURVEY.md section 2 lists them out of scope); kept so that code written against `stardist.matching` -- the reference's own tests, threshold
# optimisation scripts -- finds the names.  The overlap table is one bincount over paired label ids (the reference loops under numba).

def label_are_sequential(y):
    """True when the labels of y other than 0 are exactly 1 ... max (matching.py:13-16)"""
    present = np.unique(y)
    present = present[present != 0]
    return len(present) == 0 or bool(present[0] == 1 and present[-1] == len(present))


def is_array_of_integers(y):
    return isinstance(y, np.ndarray) and np.issubdtype(y.dtype, np.integer)


def _check_label_array(y, name=None, check_sequential=False):
    """matching.py:23-35"""
    err = ValueError("%s must be an array of %snon-negative integers." % ("labels" if name is None else name, "sequential " if check_sequential else ""))
    if not is_array_of_integers(y):
        raise err
    if len(y) == 0:
        return True
    if check_sequential:
        if not label_are_sequential(y):
            raise err
    elif y.min() < 0:
        raise err
    return True


def _label_overlap(x, y):
    """overlap[i, j] = number of pixels with label i in x and j in y (matching.py:45-52)"""
    x = np.asarray(x).ravel().astype(np.int64)
    y = np.asarray(y).ravel().astype(np.int64)
    nx, ny = int(x.max()) + 1, int(y.max()) + 1
    return np.bincount(x * ny + y, minlength=nx * ny).reshape(nx, ny).astype(np.uint)


def label_overlap(x, y, check=True):
    if check:
        _check_label_array(x, "x", True)
        _check_label_array(y, "y", True)
        if x.shape != y.shape:
            raise ValueError("x and y must have the same shape")
    return _label_overlap(x, y)


def _safe_divide(x, y, eps=1e-10):
    """x / y, 0 where |y| <= eps; float32 for arrays (matching.py:55-62)"""
    if np.isscalar(x) and np.isscalar(y):
        return x / y if np.abs(y) > eps else 0.0
    out = np.zeros(np.broadcast(x, y).shape, np.float32)
    np.divide(x, y, out=out, where=np.abs(y) > eps)
    return out


def _criterion(kind):
    def score(overlap):
        _check_label_array(overlap, "overlap")
        if np.sum(overlap) == 0:
            return overlap
        per_pred = np.sum(overlap, axis=0, keepdims=True)
        per_true = np.sum(overlap, axis=1, keepdims=True)
        return _safe_divide(overlap, {"iou": per_pred + per_true - overlap, "iot": per_true, "iop": per_pred}[kind])
    return score


intersection_over_union, intersection_over_true, intersection_over_pred = _criterion("iou"), _criterion("iot"), _criterion("iop")
matching_criteria = dict(iou=intersection_over_union, iot=intersection_over_true, iop=intersection_over_pred)


def precision(tp, fp, fn):
    return tp / (tp + fp) if tp > 0 else 0


def recall(tp, fp, fn):
    return tp / (tp + fn) if tp > 0 else 0


def accuracy(tp, fp, fn):
    return tp / (tp + fp + fn) if tp > 0 else 0


def f1(tp, fp, fn):
    return (2 * tp) / (2 * tp + fp + fn) if tp > 0 else 0


_METRIC_KEYS = ("criterion", "thresh", "fp", "tp", "fn", "precision", "recall", "accuracy", "f1", "n_true", "n_pred", "mean_true_score",
                "mean_matched_score", "panoptic_quality")


def matching(y_true, y_pred, thresh=0.5, criterion="iou", report_matches=False):
    """Detection metrics between a ground-truth and a predicted label image (matching.py:109-230): objects are paired one to one by an
    optimal assignment that maximises the number of pairs whose score reaches `thresh` (the scores break ties); tp / fp / fn, precision,
    recall, accuracy, f1, mean scores and panoptic quality come back as a namedtuple `Matching` (a tuple of them for a sequence of
    thresholds).  report_matches adds matched_pairs (original label ids), matched_scores and matched_tps."""
    from collections import namedtuple
    from scipy.optimize import linear_sum_assignment
    _check_label_array(y_true, "y_true")
    _check_label_array(y_pred, "y_pred")
    if y_true.shape != y_pred.shape:
        raise ValueError("y_true (%s) and y_pred (%s) have different shapes" % (y_true.shape, y_pred.shape))
    if criterion not in matching_criteria:
        raise ValueError("Matching criterion '%s' not supported." % criterion)
    if thresh is None:
        thresh = 0
    thresh = float(thresh) if np.isscalar(thresh) else [float(t) for t in thresh]
    y_true, _, back_true = relabel_sequential(y_true)
    y_pred, _, back_pred = relabel_sequential(y_pred)
    scores = matching_criteria[criterion](label_overlap(y_true, y_pred, check=False))
    assert 0 <= np.min(scores) <= np.max(scores) <= 1
    scores = scores[1:, 1:]                                           # without the background row / column
    n_true, n_pred = scores.shape
    n_matched = min(n_true, n_pred)

    def at(thr):
        tp, total = 0, 0.0
        ti = pi = ok = None
        if n_matched > 0:
            ti, pi = linear_sum_assignment(-(scores >= thr).astype(float) - scores / (2 * n_matched))
            assert n_matched == len(ti) == len(pi)
            ok = scores[ti, pi] >= thr
            tp = int(np.count_nonzero(ok))
            total = np.sum(scores[ti, pi][ok])
        fp, fn = n_pred - tp, n_true - tp
        vals = dict(criterion=criterion, thresh=thr, fp=fp, tp=tp, fn=fn, precision=precision(tp, fp, fn), recall=recall(tp, fp, fn),
                    accuracy=accuracy(tp, fp, fn), f1=f1(tp, fp, fn), n_true=n_true, n_pred=n_pred, mean_true_score=_safe_divide(total, n_true),
                    mean_matched_score=_safe_divide(total, tp), panoptic_quality=_safe_divide(total, tp + fp / 2 + fn / 2))
        if bool(report_matches):
            if n_matched > 0:
                vals.update(matched_pairs=tuple((int(back_true[i]), int(back_pred[j])) for i, j in zip(1 + ti, 1 + pi)),
                            matched_scores=tuple(scores[ti, pi]), matched_tps=tuple(map(int, np.flatnonzero(ok))))
            else:
                vals.update(matched_pairs=(), matched_scores=(), matched_tps=())
        return namedtuple("Matching", vals.keys())(*vals.values())
    return at(thresh) if np.isscalar(thresh) else tuple(at(t) for t in thresh)


def matching_dataset_lazy(y_gen, thresh=0.5, criterion="iou", by_image=False, show_progress=True, parallel=False):
    """matching() over (y_true, y_pred) pairs, accumulated per threshold: counts are summed, the derived metrics recomputed from the
    sums -- or, with by_image, averaged over the images (matching.py:244-315).  show_progress is accepted and ignored (no progress bar)."""
    from collections import namedtuple
    single = np.isscalar(thresh)
    threshs = (thresh,) if single else tuple(thresh)
    run = lambda pair: matching(pair[0], pair[1], thresh=threshs, criterion=criterion, report_matches=False)
    if parallel:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor() as pool:
            per_image = tuple(pool.map(run, y_gen))
    else:
        per_image = tuple(run(pair) for pair in y_gen)
    n_images = len(per_image)
    out = []
    for k, thr in enumerate(threshs):
        acc = {}
        for stats in per_image:
            s = stats[k]
            for key, v in s._asdict().items():
                if key in ("criterion",):
                    continue
                acc[key] = acc.get(key, 0) + (v * s.n_true if (key == "mean_true_score" and not by_image) else v)
        if set(acc) | {"criterion"} != set(_METRIC_KEYS):
            raise ValueError("unexpected keys")
        acc.update(criterion=criterion, thresh=thr, by_image=bool(by_image))
        if by_image:
            for key in ("precision", "recall", "accuracy", "f1", "mean_true_score", "mean_matched_score", "panoptic_quality"):
                acc[key] /= n_images
        else:
            tp, fp, fn, total = acc["tp"], acc["fp"], acc["fn"], acc["mean_true_score"]
            acc.update(precision=precision(tp, fp, fn), recall=recall(tp, fp, fn), accuracy=accuracy(tp, fp, fn), f1=f1(tp, fp, fn),
                       mean_true_score=_safe_divide(total, acc["n_true"]), mean_matched_score=_safe_divide(total, tp),
                       panoptic_quality=_safe_divide(total, tp + fp / 2 + fn / 2))
        ordered = {key: acc[key] for key in _METRIC_KEYS}               # the fields of Matching in their order, then by_image
        ordered["by_image"] = acc["by_image"]
        out.append(namedtuple("DatasetMatching", ordered.keys())(*ordered.values()))
    return out[0] if single else tuple(out)


def matching_dataset(y_true, y_pred, thresh=0.5, criterion="iou", by_image=False, show_progress=True, parallel=False):
    """matching.py:234-241"""
    if len(y_true) != len(y_pred):
        raise ValueError("y_true and y_pred must have the same length.")
    return matching_dataset_lazy(tuple(zip(y_true, y_pred)), thresh=thresh, criterion=criterion, by_image=by_image, show_progress=show_progress, parallel=parallel)


def _objects(y):
    """(label id, bounding-box slices) of the labels present, ascending -- what the reference takes from skimage's regionprops"""
    from scipy.ndimage import find_objects
    return [(i, sl) for i, sl in enumerate(find_objects(y), 1) if sl is not None]


def group_matching_labels(ys, thresh=1e-10, criterion="iou"):
    """Give matching objects of consecutive label images (frames of a time lapse) the same id (matching.py:409-472): frame k + 1 is matched
    against the already grouped frame k; a matched object takes its partner's id, an unmatched one the next free id.  Returns an int32
    stack, the inputs stay untouched."""
    if len(ys) <= 1:
        raise ValueError("'ys' must have 2 or more entries")
    if isinstance(ys, np.ndarray):
        _check_label_array(ys, "ys")
        if ys.ndim <= 1:
            raise ValueError("'ys' must be at least 2-dimensional")
        out = np.empty_like(ys, dtype=np.int32)
    else:
        if not all(_check_label_array(y, "ys") for y in ys):
            raise ValueError("'ys' must be a list of label images")
        if not all(y.shape == ys[0].shape for y in ys):
            raise ValueError("all label images must have the same shape")
        out = np.empty((len(ys),) + ys[0].shape, dtype=np.int32)
    out[0] = ys[0]
    next_id = out[0].max() + 1
    for k in range(len(ys) - 1):
        y = ys[k + 1].astype(np.int32, copy=False)
        res = matching(out[k], y, report_matches=True, thresh=thresh, criterion=criterion)
        partner = dict(reversed(res.matched_pairs[i]) for i in res.matched_tps)          # id in y -> id in the grouped previous frame
        grouped = np.zeros_like(y)
        for lab, sl in _objects(y):
            m = y[sl] == lab
            if lab in partner:
                grouped[sl][m] = partner[lab]
            else:
                grouped[sl][m] = next_id
                next_id += 1
        out[k + 1] = grouped
    return out


def _shuffle_labels(y):
    """the same objects under randomly permuted ids (numpy's global random state; matching.py:475-483)"""
    _check_label_array(y, "y")
    out = np.zeros_like(y)
    ids = tuple(set(np.unique(y)) - {0})
    new = dict(zip(ids, np.random.permutation(ids)))
    for lab, sl in _objects(y):
        out[sl][y[sl] == lab] = new[lab]
    return out
