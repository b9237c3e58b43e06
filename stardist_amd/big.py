"""Block tiling for very large inputs + its multi-GPU sharding.

Mirror of the reference's stardist/big.py (Block :19-279, BlockND :283-450) and of
StarDistBase.predict_instances_big (stardist/models/base.py:838-983), re-expressed with flat arrays instead of a
linked chain.  Semantics kept: grid-aligned overlapping blocks with a read region (whole block), a write region
(block minus context), and the "responsibility" rule that assigns every object smaller than `min_overlap` to exactly
one block (big.py:89-122); per-block results are relabelled with a running label offset in block order.

Multi-GPU (not in the reference): blocks are dealt round-robin to the ranks of the default
torch.distributed process group (one process per GPU, backend nccl = RCCL on ROCm, gloo on CPU).  The only
exchanges are (1) an all_reduce(SUM) of the per-block survivor counts -> label offsets identical to the sequential
loop, (2) the relabelled label blocks: written in block order straight into a shared np.memmap / zarr array when the caller
passes one (block.write, big.py:319-326), otherwise sent point-to-point to rank 0, which writes them in block order -- never an
all_reduce of the whole label image, (3) an all_gather of the per-object records (400 B per 3D object).  No collective touches
the per-block data path.
"""
import math
from itertools import product

import numpy as np

OBJECT_KEYS = set(("prob", "points", "coord", "dist", "class_prob", "class_id"))
COORD_KEYS = set(("points", "coord"))


class NotFullyVisible(Exception):
    pass


def _grid_divisible(grid, size, name=None, verbose=True):
    """big.py:611-619"""
    if size % grid == 0:
        return size
    _size = size
    size = math.ceil(size / grid) * grid
    if bool(verbose):
        print("increasing '%s' from %d to %d to be evenly divisible by %d (grid)" % ("value" if name is None else name, _size, size, grid), flush=True)
    return size


class Block(object):
    """One block of a 1-D chain (frozen): start/end of the read region, context on both sides, write region."""

    def __init__(self, start, size, min_overlap, context_start, context_end, at_begin, at_end, r_start):
        self.start, self.size, self.min_overlap = int(start), int(size), int(min_overlap)
        self.context_start, self.context_end = int(context_start), int(context_end)
        self.at_begin, self.at_end = at_begin, at_end
        self._r_start = int(r_start)

    @property
    def end(self): return self.start + self.size

    @property
    def slice_read(self): return slice(self.start, self.end)

    @property
    def slice_crop_context(self): return slice(self.context_start, self.size - self.context_end)

    @property
    def slice_write(self): return slice(self.start + self.context_start, self.end - self.context_end)

    def is_responsible(self, bbox):
        """big.py:89-122: bbox = (min, max) relative to the block without context"""
        bmin, bmax = bbox
        r_start = self._r_start
        r_end = self.size - self.context_start - self.context_end
        assert 0 <= bmin < bmax <= r_end
        if bmin == 0 and bmax >= r_start:
            if bmax == r_end:
                raise NotFullyVisible(True)
            if not self.at_begin:
                raise NotFullyVisible(False)
        if bmax < r_start: return False
        if bmax == r_end and not self.at_end: return False
        return True

    def __repr__(self):
        return "Block(%03d:%03d, write=%03d:%03d)" % (self.start, self.end, self.slice_write.start, self.slice_write.stop)

    @staticmethod
    def cover(size, block_size, min_overlap, context, grid=1, verbose=True):
        """Chain of grid-aligned blocks covering [0, size] (big.py:170-279)."""
        assert 0 <= min_overlap + 2 * context < block_size <= size
        assert 0 < grid <= block_size
        block_size = _grid_divisible(grid, block_size, name="block_size", verbose=verbose)
        min_overlap = _grid_divisible(grid, min_overlap, name="min_overlap", verbose=verbose)
        context = _grid_divisible(grid, context, name="context", verbose=verbose)
        size_orig = size
        size = _grid_divisible(grid, size, name="size", verbose=False)
        # everything in grid multiples
        S, B, O, C = size // grid, block_size // grid, min_overlap // grid, context // grid
        stride0 = B - (O + 2 * C)
        n = 1
        while (n - 1) * stride0 + B < S:
            n += 1
        strides = [stride0] * n
        # move blocks to make the last one end at S: decrease strides round-robin over all but the last block
        excess = (n - 1) * stride0 + B - S
        t = 0
        while excess > 0:
            assert 0 <= 1 < strides[t]
            strides[t] -= 1
            excess -= 1
            t += 1
            if t == n - 1: t = 0
        starts = [0] * n
        for i in range(1, n):
            starts[i] = starts[i - 1] + strides[i - 1]
        extra_s, extra_e = [0] * n, [0] * n
        ctx_s = lambda i: 0 if i == 0 else C + extra_s[i]
        ctx_e = lambda i: 0 if i == n - 1 else C + extra_e[i]
        w_start = lambda i: starts[i] + ctx_s(i)
        w_stop = lambda i: starts[i] + B - ctx_e(i)
        # extra context so that only neighbouring write regions overlap
        for i in range(n - 1):
            if i + 2 <= n - 1:
                ow = w_stop(i) - w_start(i + 2)
                if ow > 0:
                    extra_e[i] += ow // 2
                    extra_s[i + 2] += ow - ow // 2
        blocks = []
        for i in range(n):
            sz = B * grid
            if i == n - 1:
                sz -= (size - size_orig)              # last block is shorter if size is not grid-divisible
            cs, ce = ctx_s(i) * grid, ctx_e(i) * grid
            if i == 0:
                r_start = 0
            else:
                overlap_prev = B - strides[i - 1]     # pred.overlap
                r_start = (overlap_prev - ctx_e(i - 1) - ctx_s(i)) * grid
            blocks.append(Block(starts[i] * grid, sz, O * grid, cs, ce, i == 0, i == n - 1, r_start))
        assert blocks[0].start == 0 and blocks[-1].end == size_orig
        for i in range(n - 1):
            assert blocks[i].slice_write.stop - blocks[i + 1].slice_write.start >= O * grid
        for i in range(n - 2):
            assert blocks[i].slice_write.stop <= blocks[i + 2].slice_write.start
        return blocks


class BlockND(object):
    """N-dimensional block = one 1-D Block per axis (big.py:283-450)."""

    def __init__(self, id, blocks, axes):
        self.id = id
        self.blocks = tuple(blocks)
        self.axes = str(axes).upper()
        assert len(self.axes) == len(self.blocks)
        self.axis_to_block = dict(zip(self.axes, self.blocks))

    def __iter__(self):
        return iter(self.blocks)

    def blocks_for_axes(self, axes=None):
        axes = self.axes if axes is None else str(axes).upper()
        return tuple(self.axis_to_block[a] for a in axes)

    def slice_read(self, axes=None): return tuple(t.slice_read for t in self.blocks_for_axes(axes))

    def slice_crop_context(self, axes=None): return tuple(t.slice_crop_context for t in self.blocks_for_axes(axes))

    def slice_write(self, axes=None): return tuple(t.slice_write for t in self.blocks_for_axes(axes))

    def read(self, x, axes=None): return x[self.slice_read(axes)]

    def crop_context(self, labels, axes=None): return labels[self.slice_crop_context(axes)]

    def write(self, x, labels, axes=None):
        """write (only entries > 0 of) labels to the block's write region of x"""
        s = self.slice_write(axes)
        mask = labels > 0
        region = x[s]
        region[mask] = labels[mask]
        x[s] = region

    def is_responsible(self, slices, axes=None):
        return all(t.is_responsible((s.start, s.stop)) for t, s in zip(self.blocks_for_axes(axes), slices))

    def __repr__(self):
        return "BlockND(%s|%s)" % (self.id, ",".join("%s=%03d:%03d" % (a, t.start, t.end) for t, a in zip(self.blocks, self.axes)))

    def _responsible(self, bmin, bmax, axes=None):
        """Block.is_responsible (big.py:89-122) for all objects at once.  bmin, bmax: (n_objects, ndim) integer arrays of bounding
        boxes [min, max) relative to the block without context.  Returns (responsible bool (n,), first_violation): the index of the
        first object (in array order) for which the reference raises NotFullyVisible, or -1.  `all(...)` in the reference stops at
        the first axis that answers False, so an axis is only consulted for objects every earlier axis accepted."""
        bl = self.blocks_for_axes(axes)
        n = len(bmin)
        alive = np.ones(n, bool)          # every earlier axis answered True
        raised = np.zeros(n, bool)
        for d, t in enumerate(bl):
            lo, hi = bmin[:, d], bmax[:, d]
            r_start = t._r_start
            r_end = t.size - t.context_start - t.context_end
            assert np.all((0 <= lo) & (lo < hi) & (hi <= r_end))
            crit = (lo == 0) & (hi >= r_start)
            viol = crit & ((hi == r_end) | (not t.at_begin))
            raised |= alive & viol
            resp = ~(hi < r_start) & ~((hi == r_end) & (not t.at_end))
            alive &= resp & ~viol
        bad = np.flatnonzero(raised)
        return alive, (int(bad[0]) if len(bad) else -1)

    def filter_objects(self, labels, polys, axes=None):
        """keep only the objects this block is responsible for (big.py:340-413); returns copies.  Vectorised: bounding boxes of all
        labels at once (scipy find_objects on the host, scatter-min/max on the device for torch tensors), the responsibility rule on
        the (n, ndim) box arrays, one look-up table pass over the label image -- no Python loop over objects."""
        ndim = len(self.blocks_for_axes(axes))
        assert ndim in (2, 3)
        shape_expected = tuple(s.stop - s.start for s in self.slice_crop_context(axes))
        is_t = type(labels).__module__.startswith("torch")
        if is_t:
            import torch
            assert not labels.dtype.is_floating_point
            assert labels.dim() == ndim and tuple(labels.shape) == shape_expected
            n = int(labels.max()) if labels.numel() else 0
            big = int(max(labels.shape)) + 1
            lin = labels.reshape(-1).to(torch.int64)
            fg = torch.nonzero(lin > 0).reshape(-1)
            lab = lin[fg]
            bmin_t = torch.full((n + 1, ndim), big, dtype=torch.int64, device=labels.device)
            bmax_t = torch.full((n + 1, ndim), -1, dtype=torch.int64, device=labels.device)
            rem = fg
            for d in reversed(range(ndim)):
                c = rem % int(labels.shape[d]); rem = rem // int(labels.shape[d])
                bmin_t[:, d].scatter_reduce_(0, lab, c, "amin"); bmax_t[:, d].scatter_reduce_(0, lab, c, "amax")
            present = (bmax_t[:, 0] >= 0).cpu().numpy()
            present[0] = False
            ids = np.flatnonzero(present)
            bmin, bmax = bmin_t.cpu().numpy()[ids], bmax_t.cpu().numpy()[ids] + 1
        else:
            from scipy import ndimage as ndi
            assert np.issubdtype(labels.dtype, np.integer)
            assert labels.ndim == ndim and labels.shape == shape_expected
            objs = ndi.find_objects(labels)                                        # regionprops bbox == find_objects slices
            ids = np.array([k + 1 for k, sl in enumerate(objs) if sl is not None], np.int64)
            bmin = np.array([[s.start for s in objs[k - 1]] for k in ids], np.int64).reshape(len(ids), ndim)
            bmax = np.array([[s.stop for s in objs[k - 1]] for k in ids], np.int64).reshape(len(ids), ndim)
            n = len(objs)
        resp, bad = self._responsible(bmin, bmax, axes)
        if bad >= 0:
            shape_object = tuple(int(v) for v in (bmax[bad] - bmin[bad]))
            shape_min_overlap = tuple(t.min_overlap for t in self.blocks_for_axes(axes))
            raise RuntimeError("Found object of shape %s, which violates the assumption of being smaller than 'min_overlap' %s. "
                               "Increase 'min_overlap' to avoid this problem." % (shape_object, shape_min_overlap))
        lut = np.zeros(n + 1, np.int64)
        lut[ids[resp]] = ids[resp]
        if is_t:
            labels_filtered = torch.from_numpy(lut).to(labels.device).to(labels.dtype)[labels.to(torch.int64)]
        else:
            labels_filtered = lut.astype(labels.dtype)[labels]
        if polys is None:
            return labels_filtered
        assert isinstance(polys, dict) and any(k in polys for k in COORD_KEYS)
        filtered_ind = ids[resp] - 1
        polys_out = {k: (v[filtered_ind] if k in OBJECT_KEYS else v) for k, v in polys.items()}
        for k in COORD_KEYS:
            if k in polys_out:
                polys_out[k] = self.translate_coordinates(polys_out[k], axes=axes)
        return labels_filtered, polys_out

    def translate_coordinates(self, coordinates, axes=None):
        """local (read region) -> global coordinates (big.py:415-422)"""
        ndim = len(self.blocks_for_axes(axes))
        assert isinstance(coordinates, np.ndarray) and coordinates.ndim >= 2 and coordinates.shape[1] == ndim
        start = [s.start for s in self.slice_read(axes)]
        shape = tuple(1 if d != 1 else ndim for d in range(coordinates.ndim))
        return coordinates + np.array(start).reshape(shape)

    @staticmethod
    def cover(shape, axes, block_size, min_overlap, context, grid=1):
        """big.py:426-450"""
        shape = tuple(shape)
        n = len(shape)
        axes = str(axes).upper()
        assert len(axes) == n
        if np.isscalar(block_size): block_size = n * [block_size]
        if np.isscalar(min_overlap): min_overlap = n * [min_overlap]
        if np.isscalar(context): context = n * [context]
        if np.isscalar(grid): grid = n * [grid]
        assert n == len(block_size) == len(min_overlap) == len(context) == len(grid)
        cover_1d = [Block.cover(*args, verbose=False) for args in zip(shape, block_size, min_overlap, context, grid)]
        return tuple(BlockND(i, blocks, axes) for i, blocks in enumerate(product(*cover_1d)))


def _clipped_box(lo, hi, shape_max):
    """integer bounding intervals [floor(lo), ceil(hi)) per axis, clipped to [0, shape_max]: ((start, stop), ...)"""
    lo = np.maximum(0, np.floor(lo)).astype(int)
    hi = np.minimum(shape_max, np.ceil(hi)).astype(int)
    return tuple((a, b) for a, b in zip(tuple(lo), tuple(hi)))


class Polygon(object):
    """One predicted polygon on its own bounding box (big.py:452-472): `bbox` ((y0, y1), (x0, x1)), `slice` into the image, `shape` of
    the box, `coord` relative to the box, `mask` = the pixels the 2D rasteriser paints for it (the package's rasteriser native, pinned
    to skimage.draw.polygon: DESIGN.md section 5)."""

    def __init__(self, coord, bbox=None, shape_max=None):
        self.bbox = self.coords_bbox(coord, shape_max=shape_max) if bbox is None else bbox
        self.coord = coord - np.array([r[0] for r in self.bbox]).reshape(2, 1)
        self.slice = tuple(slice(*r) for r in self.bbox)
        self.shape = tuple(r[1] - r[0] for r in self.bbox)
        from .geometry import polygons_to_label_coord
        if min(self.shape) > 0:
            self.mask = np.asarray(polygons_to_label_coord(np.ascontiguousarray(self.coord[np.newaxis]), self.shape)) > 0
        else:
            self.mask = np.zeros(self.shape, bool)

    @staticmethod
    def coords_bbox(*coords, shape_max=None):
        """common bounding box of polygons given as (2, n) coordinate arrays"""
        assert all(isinstance(c, np.ndarray) and c.ndim == 2 and c.shape[0] == 2 for c in coords)
        every = np.concatenate(coords, axis=1)
        return _clipped_box(every.min(axis=1), every.max(axis=1), (np.inf, np.inf) if shape_max is None else shape_max)


class Polyhedron(object):
    """One predicted polyhedron on its own bounding box (big.py:476-498): as Polygon, the mask by the 3D rasteriser."""

    def __init__(self, dist, origin, rays, bbox=None, shape_max=None):
        self.bbox = self.coords_bbox((dist, origin), rays=rays, shape_max=shape_max) if bbox is None else bbox
        self.slice = tuple(slice(*r) for r in self.bbox)
        self.shape = tuple(r[1] - r[0] for r in self.bbox)
        from .geometry import polyhedron_to_label
        local = origin.reshape(1, 3) - np.array([r[0] for r in self.bbox]).reshape(1, 3)
        self.mask = np.asarray(polyhedron_to_label(dist[np.newaxis], local, rays, shape=self.shape, verbose=False)).astype(bool)

    @staticmethod
    def coords_bbox(*dist_origin, rays, shape_max=None):
        """common bounding box of polyhedra given as (dist (n_rays,), origin (3,)) pairs"""
        dists, points = zip(*dist_origin)
        assert all(isinstance(d, np.ndarray) and d.ndim == 1 and len(d) == len(rays) for d in dists)
        assert all(isinstance(p, np.ndarray) and p.ndim == 1 and len(p) == 3 for p in points)
        verts = np.stack(dists)[..., np.newaxis] * rays.vertices[np.newaxis] + np.stack(points)[:, np.newaxis]      # (m, n_rays, 3)
        verts = verts.reshape(-1, 3)
        return _clipped_box(verts.min(axis=0), verts.max(axis=0), (np.inf, np.inf, np.inf) if shape_max is None else shape_max)


def predict_big(model, *args, **kwargs):
    """big.py:596-602: the old entry point only tells where the function went"""
    from .models import StarDist2D, StarDist3D
    dst = type(model).__name__ if isinstance(model, (StarDist2D, StarDist3D)) else "{StarDist2D, StarDist3D}"
    raise RuntimeError("This function has moved to %s.predict_instances_big." % dst)


def relabel_with_offset(labels, offset):
    from .matching import relabel_sequential
    return relabel_sequential(labels, offset)[0]


def predict_instances_big(model, img, axes, block_size, min_overlap, context=None, labels_out=None, labels_out_dtype=np.int32,
                          show_progress=True, distributed=None, **kwargs):
    """StarDistBase.predict_instances_big (base.py:838-983) + round-robin sharding of the blocks over the ranks of the
    default torch.distributed group (when initialised, or distributed=True).  Every rank returns the full object dict.  The label
    image is NOT replicated: it is returned by rank 0 only (the other ranks return None for it and do not allocate it), or lives in
    the shared memmap / zarr array passed as labels_out (returned by every rank)."""
    from .models.base import axes_check_and_normalize, axes_dict
    n = img.ndim
    axes = axes_check_and_normalize(axes, length=n)
    grid = model._axes_div_by(axes)
    axes_out = model.config.axes.replace("C", "")
    shape_dict = dict(zip(axes, img.shape))
    shape_out = tuple(shape_dict[a] for a in axes_out)
    if context is None:
        context = model._axes_tile_overlap(axes)
    if np.isscalar(block_size): block_size = n * [block_size]
    if np.isscalar(min_overlap): min_overlap = n * [min_overlap]
    if np.isscalar(context): context = n * [context]
    block_size, min_overlap, context = list(block_size), list(min_overlap), list(context)
    assert n == len(block_size) == len(min_overlap) == len(context)
    if "C" in axes:
        i = axes_dict(axes)["C"]
        block_size[i] = img.shape[i]
        min_overlap[i] = context[i] = 0
    block_size = tuple(_grid_divisible(g, v, name="block_size", verbose=False) for v, g in zip(block_size, grid))
    min_overlap = tuple(_grid_divisible(g, v, name="min_overlap", verbose=False) for v, g in zip(min_overlap, grid))
    context = tuple(_grid_divisible(g, v, name="context", verbose=False) for v, g in zip(context, grid))
    if show_progress:
        print("effective: block_size=%s, min_overlap=%s, context=%s" % (block_size, min_overlap, context), flush=True)
    blocks = BlockND.cover(img.shape, axes, block_size, min_overlap, context, grid)

    dist_ = None
    rank, world = 0, 1
    try:
        import torch
        import torch.distributed as td
        if (distributed is None and td.is_available() and td.is_initialized()) or distributed:
            dist_ = td
            rank, world = td.get_rank(), td.get_world_size()
    except ImportError:
        pass

    want_labels = not (np.isscalar(labels_out) and bool(labels_out) is False)
    owns_labels = True              # False on ranks != 0 of a multi-rank run without a shared labels_out: rank 0 alone holds the image
    if want_labels:
        if labels_out is None:
            owns_labels = rank == 0 or world == 1
            labels_out = np.zeros(shape_out, dtype=labels_out_dtype) if owns_labels else None
        elif tuple(labels_out.shape) != tuple(shape_out):
            raise ValueError("'labels_out' must have shape %s (axes %s)." % (shape_out, axes_out))
    else:
        labels_out = None
    out_dtype = np.dtype(labels_out.dtype if labels_out is not None else labels_out_dtype)

    kwargs_override = dict(axes=axes, overlap_label=None, return_labels=True, return_predict=False)
    if show_progress:
        kwargs_override["show_tile_progress"] = False
    for k, v in kwargs_override.items():
        if k in kwargs and show_progress:
            print("changing '%s' from %s to %s" % (k, kwargs[k], v), flush=True)
        kwargs[k] = v

    # ---- phase 1: every rank processes its blocks (local label ids 1..k)
    mine = {}
    counts = np.zeros(len(blocks), np.int64)
    for bi, block in enumerate(blocks):
        if bi % world != rank:
            continue
        labels, polys = model.predict_instances(block.read(img, axes=axes), **kwargs)
        labels = block.crop_context(labels, axes=axes_out)
        labels, polys = block.filter_objects(labels, polys, axes=axes_out)
        counts[bi] = len(polys["prob"])
        mine[bi] = (labels, polys)

    # ---- phase 2: label offsets in block order (exclusive scan of the per-block survivor counts, base.py:942,972)
    if dist_ is not None and world > 1:
        import torch
        dev = torch.device("cuda", torch.cuda.current_device()) if dist_.get_backend() == "nccl" else torch.device("cpu")
        tc = torch.from_numpy(counts).to(dev)
        dist_.all_reduce(tc, op=dist_.ReduceOp.SUM)
        counts = tc.cpu().numpy()
    offsets = 1 + np.concatenate([[0], np.cumsum(counts)[:-1]])

    # ---- phase 3: write my blocks; merge.  The label image is NOT reduced across ranks (1 GiB at 16384^2, 4 GiB at 1024^3):
    #   * labels_out backed by shared storage (np.memmap on a file every rank opened, zarr-like objects with a `store`): every rank
    #     writes the write regions of ITS blocks in place, as block.write does (big.py:319-326), in block order (a barrier per
    #     block keeps the sequential loop's "later block overwrites" rule where the objects of two blocks overlap);
    #   * otherwise rank 0 owns the result: the other ranks send the (cropped, relabelled) label block of each of their blocks
    #     point-to-point, rank 0 writes them in block order.  Total traffic = one image, to one rank.
    polys_blocks = {}
    shared_out = labels_out is not None and (isinstance(labels_out, np.memmap) or hasattr(labels_out, "store"))
    # wire type of a label block: the output dtype (never narrower: label ids may exceed 2^31 in an int64 output)
    import torch as _torch
    wire = {np.dtype(np.int64): _torch.int64, np.dtype(np.uint32): _torch.int64, np.dtype(np.uint64): _torch.int64,
            np.dtype(np.int16): _torch.int16, np.dtype(np.uint8): _torch.uint8}.get(out_dtype, _torch.int32)
    if wire == _torch.int32 and out_dtype.itemsize <= 4 and len(offsets) and int(offsets[-1] + counts[-1]) >= 2 ** 31:
        raise OverflowError("label ids exceed the int32 range of labels_out_dtype=%s" % out_dtype)
    for bi in sorted(mine):
        labels, polys = mine[bi]
        labels = relabel_with_offset(labels, int(offsets[bi]))
        mine[bi] = (labels, polys)
        polys_blocks[bi] = polys
    if dist_ is not None and world > 1:
        import torch
        if labels_out is not None and shared_out:
            for bi in range(len(blocks)):
                if bi in mine:
                    blocks[bi].write(labels_out, mine[bi][0].astype(labels_out.dtype, copy=False), axes=axes_out)
                    if hasattr(labels_out, "flush"):
                        labels_out.flush()
                dist_.barrier()
        elif want_labels:
            np_wire = {_torch.int64: np.int64, _torch.int32: np.int32, _torch.int16: np.int16, _torch.uint8: np.uint8}[wire]
            for bi, block in enumerate(blocks):
                owner = bi % world
                shp = tuple(sl.stop - sl.start for sl in block.slice_crop_context(axes_out))
                if owner == 0:
                    if rank == 0:
                        block.write(labels_out, mine[bi][0].astype(labels_out.dtype, copy=False), axes=axes_out)
                elif rank == owner:
                    dist_.send(torch.from_numpy(np.ascontiguousarray(mine[bi][0].astype(np_wire))).to(dev), dst=0)
                elif rank == 0:
                    buf = torch.empty(shp, dtype=wire, device=dev)
                    dist_.recv(buf, src=owner)
                    block.write(labels_out, buf.cpu().numpy().astype(labels_out.dtype, copy=False), axes=axes_out)
        gathered = [None] * world
        dist_.all_gather_object(gathered, polys_blocks)
        polys_blocks = {}
        for g in gathered:
            polys_blocks.update(g)
    elif labels_out is not None:
        for bi in sorted(mine):
            blocks[bi].write(labels_out, mine[bi][0].astype(labels_out.dtype, copy=False), axes=axes_out)
    polys_all = {}
    for bi in sorted(polys_blocks):
        for k, v in polys_blocks[bi].items():
            polys_all.setdefault(k, []).append(v)
    polys_all = {k: (np.concatenate(v) if k in OBJECT_KEYS else v[0]) for k, v in polys_all.items()}
    if want_labels and not owns_labels:
        return None, polys_all                                      # the label image lives on rank 0
    return (labels_out if labels_out is not None else False), polys_all


def _exclusive_intervals(blocks, axes_out):
    """per block and output axis: [lo, hi) = the part of the block's write region no other block's write region covers
    (neighbouring write regions overlap by >= min_overlap, non-neighbouring ones never: Block.cover)"""
    nd = len(axes_out)
    per_axis = [sorted(set((t.start + t.context_start, t.end - t.context_end) for b in blocks for t in [b.blocks_for_axes(axes_out)[a]])) for a in range(nd)]
    out = np.empty((len(blocks), nd, 2), np.float64)
    for bi, b in enumerate(blocks):
        for a, t in enumerate(b.blocks_for_axes(axes_out)):
            ws = per_axis[a]
            k = ws.index((t.start + t.context_start, t.end - t.context_end))
            out[bi, a, 0] = ws[k - 1][1] if k > 0 else -np.inf
            out[bi, a, 1] = ws[k + 1][0] if k + 1 < len(ws) else np.inf
    return out


def sharded_cover(model, shape, axes, block_size, min_overlap, context=None):
    """the blocks predict_instances_sharded deals to the ranks for an input of `shape` (BlockND.cover after the same normalisation of
    block_size / min_overlap / context: grid-divisible, the channel axis whole): (blocks, normalised axes, (block_size, min_overlap, context))"""
    from .models.base import axes_check_and_normalize, axes_dict
    n = len(shape)
    axes = axes_check_and_normalize(axes, length=n)
    grid = model._axes_div_by(axes)
    if context is None:
        context = model._axes_tile_overlap(axes)
    if np.isscalar(block_size): block_size = n * [block_size]
    if np.isscalar(min_overlap): min_overlap = n * [min_overlap]
    if np.isscalar(context): context = n * [context]
    block_size, min_overlap, context = list(block_size), list(min_overlap), list(context)
    if "C" in axes:
        i = axes_dict(axes)["C"]
        block_size[i] = shape[i]
        min_overlap[i] = context[i] = 0
    block_size = tuple(_grid_divisible(g, v, name="block_size", verbose=False) for v, g in zip(block_size, grid))
    min_overlap = tuple(_grid_divisible(g, v, name="min_overlap", verbose=False) for v, g in zip(min_overlap, grid))
    context = tuple(_grid_divisible(g, v, name="context", verbose=False) for v, g in zip(context, grid))
    return BlockND.cover(tuple(shape), axes, block_size, min_overlap, context, grid), axes, (block_size, min_overlap, context)


class ShardedInput(object):
    """One rank's view of a large input: the shape of the WHOLE array, but only the read regions (block + context) of the blocks this rank
    owns are held -- read from a numpy memmap / zarr-like / array `source` (anything with .shape and slicing) and, with `device`, kept
    resident there.  predict_instances_sharded takes it in place of the array: `block.read` (big.py:164 of the reference: x[slices]) finds
    the region by its slices.  A rank of an N-rank job so holds ~1/N of the input (plus the blocks' context) instead of all of it
    (1024^3 float32: 4 GiB on every rank before); a region that was not prefetched is read from the source on demand.

        src = np.load("slide.npy", mmap_mode="r")
        x = ShardedInput.for_rank(model, src, "YX", block_size, min_overlap, context, rank, world, device=model.device)
        labels, res = predict_instances_sharded(model, x, "YX", block_size, min_overlap, context)
    """

    def __init__(self, source, device=None):
        self.source, self.device = source, device
        self.shape = tuple(int(v) for v in source.shape)
        self.ndim = len(self.shape)
        self.dtype = getattr(source, "dtype", None)
        self._held = {}
        self.bytes_held = 0

    @staticmethod
    def _key(slices):
        return tuple((int(s.start), int(s.stop)) for s in slices)

    def _load(self, slices):
        a = self.source[tuple(slices)]
        if self.device is not None:
            import torch
            t = a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
            a = t.to(self.device)
        elif not isinstance(a, np.ndarray) and not type(a).__module__.startswith("torch"):
            a = np.asarray(a)
        return a

    def prefetch(self, slices):
        k = self._key(slices)
        if k not in self._held:
            a = self._load(slices)
            self._held[k] = a
            self.bytes_held += int(np.prod(a.shape)) * (a.element_size() if hasattr(a, "element_size") else a.dtype.itemsize)
        return self._held[k]

    def __getitem__(self, slices):
        if not isinstance(slices, tuple):
            slices = (slices,)
        if len(slices) == self.ndim and all(isinstance(s, slice) and s.step in (None, 1) and s.start is not None and s.stop is not None for s in slices):
            a = self._held.get(self._key(slices))
            if a is not None:
                return a
        return self._load(slices)

    @classmethod
    def for_rank(cls, model, source, axes, block_size, min_overlap, context=None, rank=0, world=1, device=None):
        """the view of rank `rank` of `world`: the read regions of blocks rank, rank + world, ... (the deal of predict_instances_sharded) prefetched"""
        x = cls(source, device)
        blocks, axes_n, _ = sharded_cover(model, x.shape, axes, block_size, min_overlap, context)
        for bi, b in enumerate(blocks):
            if bi % world == rank:
                x.prefetch(b.slice_read(axes_n))
        return x


def predict_instances_sharded(model, img, axes, block_size, min_overlap, context=None, prob_thresh=None, nms_thresh=None,
                              return_labels=True, labels_out=None, show_progress=False, distributed=None, predict_kwargs=None,
                              nms_kwargs=None, broadcast_result=True, keep_debug=False):
    """Block-sharded prediction with a final cross-tile NMS (SURVEY.md 8e design A, the north-star's multi-GPU path).

    The blocks of `BlockND.cover` (big.py:426-450) are dealt round-robin to the ranks of the default torch.distributed group
    (one process per GPU; RCCL when the backend is "nccl").  Per block, on the owning rank and on the device: network + candidate
    selection (`predict_sparse`) + LOCAL NMS on the block incl. its context; a block keeps the local survivors whose centre lies in its
    write region (= block minus context; neighbouring write regions overlap by >= min_overlap, so an object in an overlap band is
    seen with full context by both blocks).  Two survivors of one block never suppress each other (the local NMS kept both), so a
    survivor whose bounding box, grown by the largest bounding radius of all survivors (one scalar all_reduce(MAX)), stays inside the
    part of its block's write region that no other block covers is final as it is ("interior"); the OWNER decides that.  Exchange: the
    kept survivors -- one packed float32 record [dist(R) | prob | centre(nd) | block id | class probabilities] each, 141 B (2D) /
    401 B (3D) + 4, interior records first -- go to rank 0 point to point in their exact sizes (batch_isend_irecv after an all_gather
    of the two counts; nothing is padded to the largest rank).  Rank 0 drops exact duplicates among the "band" records (same pixel
    reported by two blocks) and runs the SAME NMS once more over the band only.  The result equals the NMS over the whole union.  The final instances, in global score
    order (= label ids, as predict_instances numbers them), are broadcast, and every rank renders the write regions of ITS blocks
    from that list (windowed rasteriser; pixels in overlapping write regions come out identical on both owners).

    big == whole exactly: `context` >= the network's receptive field (the default, model._axes_tile_overlap, is).  A band candidate that
    two blocks report is taken from the block it lies deepest in (round 6; design B's responsibility rule -- every object from the block
    it is most central in -- applied to a point), so a smaller context degrades gracefully instead of depending on the block order.
    (tests/test_cpu_reference_end_to_end.py runs the reference's own big == whole acceptance test on both designs.)

    `img` may be a ShardedInput (this rank's read regions only, e.g. from a memmap) instead of the whole array.

    Label image: `labels_out=None` -- the tiles are sent to rank 0, which returns the whole image (ranks != 0 return None);
    a shared `np.memmap` / zarr-like array -- every rank writes its tiles in place (big.py:319-326 block.write), returned on every rank;
    `labels_out="local"` -- nothing is moved: every rank returns [(block index, slices, tile), ...] for its blocks.
    Unlike `predict_instances_big` (design B, the reference's own semantics: per-block NMS + bbox responsibility rule, no
    cross-tile NMS) label ids follow the global score order.  Per-stage wall times and counters: model._last_sharded_stats.

    keep_debug=True: rank 0 leaves the unique gathered records, the interior / band split and the keep mask in model._last_sharded_debug
    (the full-size parity tests compare them with the reference NMS over the same records).

    Returns (labels, dict) on rank 0; (labels-or-None, dict) on the other ranks (dict None there with broadcast_result=False)."""
    import time
    import torch
    from .models.base import axes_check_and_normalize, axes_dict
    predict_kwargs = dict(predict_kwargs or {})
    nms_kwargs = dict(nms_kwargs or {})
    n = img.ndim
    blocks, axes, (block_size, min_overlap, context) = sharded_cover(model, img.shape, axes, block_size, min_overlap, context)
    axes_out = model.config.axes.replace("C", "")
    shape_dict = dict(zip(axes, img.shape))
    shape_out = tuple(shape_dict[a] for a in axes_out)
    if show_progress:
        print("sharded: %d blocks, block_size=%s, min_overlap=%s, context=%s" % (len(blocks), block_size, min_overlap, context), flush=True)

    dist_, rank, world = None, 0, 1
    try:
        import torch.distributed as td
        if (distributed is None and td.is_available() and td.is_initialized()) or distributed:
            dist_, rank, world = td, td.get_rank(), td.get_world_size()
    except ImportError:
        pass
    multi = dist_ is not None and world > 1

    # the product path keeps every candidate on the GPU (device tensors through selection, local NMS, exchange, final NMS, raster); a model
    # without predict_sparse_device (the CPU tests' stand-in, whose kernels are the oracle) is driven through numpy at the same places
    on_dev = hasattr(model, "predict_sparse_device") and str(getattr(model, "device", "cpu")).startswith("cuda")
    dev = model.device if on_dev else torch.device("cpu")
    nd = len(axes_out)
    R = model.config.n_rays
    n_cls = (model.config.n_classes + 1) if getattr(model.config, "n_classes", None) is not None else 0
    W = R + 1 + nd + 1 + n_cls                                      # record width
    c_prob, c_pts, c_blk, c_cls = R, R + 1, R + 1 + nd, R + 2 + nd
    st = dict(blocks=0, candidates=0, local_survivors=0, t_phase1=0.0, t_predict=0.0, t_local_nms=0.0, t_exchange=0.0, t_final=0.0, t_final_nms=0.0,
              t_raster=0.0, gathered=0, gathered_bytes=0, unique=0, band=0, interior=0, instances=0, per_block=[])

    def tick():
        if on_dev:
            torch.cuda.synchronize(dev)
        return time.perf_counter()

    def as_t(a, dtype=None):
        t = a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
        return t.to(dev) if dtype is None else t.to(dev, dtype)

    def local_nms(dist, prob, points):
        """indices of the survivors, best score first"""
        if on_dev:
            return model._nms_sparse_device(dist, prob, points, nms_thresh=nms_thresh, **nms_kwargs)[3]
        keep = model._nms_sparse(dist.numpy(), prob.numpy(), points.numpy(), nms_thresh=nms_thresh, **nms_kwargs)
        return torch.from_numpy(np.asarray(keep).astype(np.int64))

    # ---- phase 1: my blocks -> local survivors with their centre in the block's write region, global coordinates
    recs = []

    def block_record(bi, block, prob, dist, points, pcls):
        keep = local_nms(dist, prob, points)
        bl = block.blocks_for_axes(axes_out)
        start = torch.tensor([t.start for t in bl], device=dev, dtype=torch.int64).reshape(1, nd)
        lo = torch.tensor([t.start + t.context_start for t in bl], device=dev, dtype=torch.int64).reshape(1, nd)
        hi = torch.tensor([t.end - t.context_end for t in bl], device=dev, dtype=torch.int64).reshape(1, nd)
        gp = points[keep].to(torch.int64) + start
        inside = torch.all((gp >= lo) & (gp < hi), dim=1)
        keep, gp = keep[inside], gp[inside]
        rec = torch.empty((int(keep.numel()), W), dtype=torch.float32, device=dev)
        rec[:, :R] = dist[keep].float(); rec[:, c_prob] = prob[keep].float()
        rec[:, c_pts:c_pts + nd] = gp.float(); rec[:, c_blk] = float(bi)     # coordinates and block ids are < 2^24: exact in float32
        if n_cls:
            rec[:, c_cls:] = pcls[keep].float()
        return rec

    mine = [bi for bi in range(len(blocks)) if bi % world == rank]
    for bi in mine:
        block = blocks[bi]
        t0 = tick()
        x = block.read(img, axes=axes)
        res = (model.predict_sparse_device if on_dev else model.predict_sparse)(x, axes=axes, prob_thresh=prob_thresh, **predict_kwargs)
        prob, dist, points = as_t(res[0]), as_t(res[1]), as_t(res[-1])
        pcls = as_t(res[2]) if n_cls else None
        t1 = tick()
        st["blocks"] += 1; st["candidates"] += int(prob.numel()); st["t_predict"] += t1 - t0
        if prob.numel() == 0:
            st["per_block"].append((bi, round(t1 - t0, 5), 0.0))
            continue
        rec = block_record(bi, block, prob, dist, points, pcls)
        recs.append(rec)
        t2 = tick()
        st["local_survivors"] += int(rec.shape[0]); st["t_local_nms"] += t2 - t1
        st["per_block"].append((bi, round(t1 - t0, 5), round(t2 - t1, 5)))
    st["t_phase1"] = st["t_predict"] + st["t_local_nms"]
    rec = torch.cat(recs) if recs else torch.zeros((0, W), dtype=torch.float32, device=dev)

    # ---- phase 2: interior / band split ON THE OWNERS, then one exact-size exchange of the records to rank 0.
    # Two survivors of one block never suppress each other (the local NMS kept both), so a survivor whose bounding box, grown by the
    # largest bounding radius of ALL survivors (+1), stays inside the part of its block's write region that no other block's write region
    # covers can neither suppress nor be suppressed, nor be reported twice: it is final ("interior").  The owner decides that from its
    # own records and ONE scalar all_reduce(MAX) of the bounding radius; only the remaining "band" records are de-duplicated and run
    # through the NMS again on rank 0.
    t0 = tick()
    cdev = dev
    if multi:
        cdev = dev if dist_.get_backend() == "nccl" else torch.device("cpu")
    vmax = 1.0
    if nd == 3:
        from .rays3d import rays_from_json
        vmax = float(np.abs(rays_from_json(model.config.rays_json).vertices).max())
    rad = rec[:, :R].amax(dim=1).double() * vmax + 1.0 if rec.shape[0] else torch.zeros((0,), dtype=torch.float64, device=dev)
    rmax = torch.tensor([float(rad.max()) if rec.shape[0] else 0.0], dtype=torch.float64, device=cdev)   # (+1 above: integer truncation / rounding of vertices)
    if multi:
        dist_.all_reduce(rmax, op=dist_.ReduceOp.MAX)
    if rec.shape[0]:
        margin = (rad + float(rmax.item()) + 1.0).reshape(-1, 1)
        ex = torch.from_numpy(_exclusive_intervals(blocks, axes_out)).to(dev)[rec[:, c_blk].to(torch.int64)]   # (n, nd, 2)
        c = rec[:, c_pts:c_pts + nd].double()
        is_int = torch.all((c - margin >= ex[:, :, 0]) & (c + margin < ex[:, :, 1]), dim=1)
        rec = torch.cat([rec[is_int], rec[~is_int]])                # interior records first
        n_int = int(is_int.sum())
    else:
        n_int = 0
    n_loc = int(rec.shape[0])
    ints, bands = [rec[:n_int]], [rec[n_int:]]
    if multi:
        cnt = torch.tensor([n_int, n_loc - n_int], dtype=torch.int64, device=cdev)
        cnts = [torch.zeros_like(cnt) for _ in range(world)]
        dist_.all_gather(cnts, cnt)
        cnts = [(int(c[0].item()), int(c[1].item())) for c in cnts]
        # exact sizes, point to point (RCCL over xGMI: the senders use their own links to rank 0 in parallel): no padding to the largest rank
        ops, bufs = [], {}
        if rank == 0:
            for r in range(1, world):
                if sum(cnts[r]):
                    bufs[r] = torch.empty((sum(cnts[r]), W), dtype=torch.float32, device=cdev)
                    ops.append(dist_.P2POp(dist_.irecv, bufs[r], r))
        elif n_loc:
            ops.append(dist_.P2POp(dist_.isend, rec.to(cdev).contiguous(), 0))
        if ops:
            for q in dist_.batch_isend_irecv(ops):
                q.wait()
        if rank == 0:
            for r in range(1, world):
                if r in bufs:
                    b = bufs[r].to(dev)
                    ints.append(b[:cnts[r][0]]); bands.append(b[cnts[r][0]:])
        st["rank_counts"] = cnts                                    # (interior, band) records of every rank
        st["sent_bytes"] = 0 if rank == 0 else n_loc * W * 4        # this rank's payload on its link to rank 0
        st["gathered"] = int(sum(a + b for a, b in cnts))
        st["gathered_bytes"] = int(sum((a + b) for r, (a, b) in enumerate(cnts) if r != 0)) * W * 4      # what actually crosses a link
        st["exact_record_bytes"] = st["gathered"] * W * 4
    else:
        st["gathered"] = n_loc
    st["t_exchange"] = tick() - t0

    # ---- phase 3 (rank 0): duplicates among the band records, cross-tile NMS over the band, global score order
    t0 = tick()
    final = None
    if rank == 0:
        rint, rband = torch.cat(ints), torch.cat(bands)
        order = torch.sort(rband[:, c_blk], stable=True)[1]         # canonical order (block index, then the block's score order):
        rband = rband[order]                                        # the result does not depend on the number of ranks

        def pixel_key(r):
            p = r[:, c_pts:c_pts + nd].to(torch.int64)
            key = p[:, 0]
            for d in range(1, nd):
                key = key * int(shape_out[d]) + p[:, d]
            return key
        raw_band = rband if keep_debug else None                    # (every report of the band, before the duplicates are dropped)
        if rband.shape[0]:
            # the same pixel reported by two (or more) overlapping blocks: keep the report of the block the pixel lies DEEPEST in (largest
            # distance to that block's border, context included; ties: the lower block index) -- the prediction with the most context, what
            # the reference's responsibility rule (big.py:89-122: every object from the block it is most central in) amounts to for a
            # point.  With context >= the receptive field all reports of a pixel are identical and the choice does not matter.
            ext = torch.tensor([[[t.start, t.end] for t in b.blocks_for_axes(axes_out)] for b in blocks], dtype=torch.float32, device=dev)   # (blocks, nd, 2)
            e = ext[rband[:, c_blk].to(torch.int64)]
            pc = rband[:, c_pts:c_pts + nd]
            depth = torch.minimum(pc - e[:, :, 0], e[:, :, 1] - 1.0 - pc).amin(dim=1)
            rband = rband[torch.sort(-depth, stable=True)[1]]       # (stable: equal depths stay in block order)
            ks, ki = torch.sort(pixel_key(rband), stable=True)
            first = torch.ones_like(ks, dtype=torch.bool)
            first[1:] = ks[1:] != ks[:-1]
            rband = rband[ki[first]]
            rband = rband[torch.sort(rband[:, c_blk], stable=True)[1]]   # back to the canonical (block, score) order
        # all unique records in row-major pixel order: candidates of EQUAL score are then taken by every later sort (stable ascending,
        # reversed: nms._argsort_desc) in the order predict_instances on the whole image takes them, whatever the blocks
        rec = torch.cat([rint, rband])
        interior = torch.cat([torch.ones(rint.shape[0], dtype=torch.bool, device=dev), torch.zeros(rband.shape[0], dtype=torch.bool, device=dev)])
        if rec.shape[0]:
            ki = torch.sort(pixel_key(rec), stable=True)[1]
            rec, interior = rec[ki], interior[ki]
        pts = rec[:, c_pts:c_pts + nd].to(torch.int64)
        nU = int(rec.shape[0])
        st["unique"] = nU
        band = torch.nonzero(~interior).reshape(-1)
        st["band"], st["interior"] = int(band.numel()), nU - int(band.numel())
        t1 = tick()
        keep_mask = interior.clone()
        if band.numel():
            kb = local_nms(rec[band, :R].contiguous(), rec[band, c_prob].contiguous(), pts[band].contiguous())
            keep_mask[band[kb]] = True
        st["t_final_nms"] = tick() - t1
        from .nms import _argsort_desc
        so = _argsort_desc(rec[:, c_prob])                          # the order predict_instances' NMS works in and returns
        final = rec[so[keep_mask[so]]].contiguous()
        st["instances"] = int(final.shape[0])
        if keep_debug:                                              # the parity tests' view of the exchange: the unique gathered records
            model._last_sharded_debug = dict(dist=rec[:, :R], prob=rec[:, c_prob], points=pts, block=rec[:, c_blk], interior=interior,
                                             keep=keep_mask, order=so, raw_band_points=raw_band[:, c_pts:c_pts + nd].to(torch.int64),
                                             raw_band_block=raw_band[:, c_blk].to(torch.int64))
    if multi and (return_labels or broadcast_result):               # the final instances to every rank (M x record)
        m = torch.tensor([final.shape[0] if rank == 0 else 0], dtype=torch.int64, device=cdev)
        dist_.broadcast(m, src=0)
        fb = final.to(cdev) if rank == 0 else torch.empty((int(m.item()), W), dtype=torch.float32, device=cdev)
        dist_.broadcast(fb, src=0)
        final = fb.to(dev)

    res_dict = None
    f_pts = f_prob = f_dist = f_cls = None
    if final is not None:
        f_pts, f_prob, f_dist = final[:, c_pts:c_pts + nd].to(torch.int64), final[:, c_prob].contiguous(), final[:, :R].contiguous()
        f_cls = final[:, c_cls:].contiguous() if n_cls else None
        cv = (lambda t: t) if on_dev else (lambda t: None if t is None else t.numpy())
        if rank == 0 or broadcast_result:
            kw = dict(prob_class=cv(f_cls)) if n_cls else {}
            res_dict = model._instances_from_survivors(shape_out, cv(f_pts), cv(f_prob), cv(f_dist), return_labels=False, **kw)[1]

    # ---- phase 4: labels -- every rank renders the write regions of its blocks from the final list
    t1 = tick()
    labels = None
    if return_labels and final is not None:
        cv = (lambda t: t) if on_dev else (lambda t: t.numpy())
        if not multi and not (isinstance(labels_out, str) and labels_out == "local"):
            labels = model._instances_from_survivors(shape_out, cv(f_pts), cv(f_prob), cv(f_dist), return_labels=True)[0]
            if labels_out is not None and not np.isscalar(labels_out):
                labels_out[...] = labels
                labels = labels_out
        else:
            tiles = []
            for bi, block in enumerate(blocks):
                if bi % world != rank:
                    continue
                bl = block.blocks_for_axes(axes_out)
                sl = tuple(slice(t.start + t.context_start, t.end - t.context_end) for t in bl)
                window = (tuple(s.start for s in sl), tuple(s.stop - s.start for s in sl))
                tile = model._instances_from_survivors(shape_out, cv(f_pts), cv(f_prob), cv(f_dist), return_labels=True, window=window)[0]
                tiles.append((bi, sl, as_t(tile)))
            if nd == 3:                                             # relabel_sequential over the whole volume (model3d.py:646): ids that are
                present = torch.zeros(int(final.shape[0]) + 1, dtype=torch.int32, device=dev)   # hidden everywhere are closed up
                for _, _, t in tiles:
                    present[torch.unique(t).to(torch.int64)] = 1
                if multi:
                    pc = present.to(cdev)
                    dist_.all_reduce(pc, op=dist_.ReduceOp.MAX)
                    present = pc.to(dev)
                present[0] = 0
                fwd = torch.cumsum(present, 0).to(torch.int32) * present
                tiles = [(bi, sl, fwd[t.to(torch.int64)]) for bi, sl, t in tiles]
            if isinstance(labels_out, str) and labels_out == "local":
                labels = tiles
            elif labels_out is not None and (isinstance(labels_out, np.memmap) or hasattr(labels_out, "store")):
                for bi, sl, t in tiles:                             # overlapping write regions hold identical pixels: no ordering needed
                    labels_out[sl] = t.cpu().numpy().astype(labels_out.dtype, copy=False)
                if hasattr(labels_out, "flush"):
                    labels_out.flush()
                dist_.barrier()
                labels = labels_out
            else:                                                   # the whole image on rank 0: tiles point-to-point, in block order
                if rank == 0:
                    labels = np.zeros(shape_out, np.int32) if labels_out is None else labels_out
                mine = {bi: (sl, t) for bi, sl, t in tiles}
                for bi, block in enumerate(blocks):
                    owner = bi % world
                    bl = block.blocks_for_axes(axes_out)
                    sl = tuple(slice(t.start + t.context_start, t.end - t.context_end) for t in bl)
                    if owner == 0:
                        if rank == 0:
                            labels[sl] = mine[bi][1].cpu().numpy()
                    elif rank == owner:
                        dist_.send(mine[bi][1].to(torch.int32).contiguous().to(cdev), dst=0)
                    elif rank == 0:
                        buf = torch.empty(tuple(s.stop - s.start for s in sl), dtype=torch.int32, device=cdev)
                        dist_.recv(buf, src=owner)
                        labels[sl] = buf.cpu().numpy()
    st["t_raster"] = tick() - t1
    st["t_final"] = tick() - t0
    model._last_sharded_stats = st
    return labels, res_dict
