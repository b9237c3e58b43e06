"""relabel_sequential: the one function of stardist/matching.py that sits on the prediction path
(model3d.py:634-645, base.py:959).  Device version for torch label volumes, numpy version otherwise."""
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
