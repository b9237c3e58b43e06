"""CPU: block tiling (stardist_amd/big.py) against golden covers from the reference, the reference's own
cover/filter/reassemble identity test (tests/test_big.py:50-76), and the multi-process (gloo, world_size 2) sharding."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_block_cover_matches_reference_golden():
    from stardist_amd.big import Block
    G = np.load(os.path.join(ROOT, "tests", "golden", "big_cover.npz"))
    k = 0
    while "case%d" % k in G:
        size, bs, mo, ctx, grid = (int(v) for v in G["args%d" % k])
        bl = Block.cover(size, bs, mo, ctx, grid, verbose=False)
        rows = np.array([[t.start, t.end, t.slice_write.start, t.slice_write.stop, t.context_start, t.context_end, t._r_start] for t in bl])
        assert np.array_equal(rows, G["case%d" % k]), k
        k += 1
    assert k == 10


def _gt_labels(shape, seed=0, n=60, r=(3, 6)):
    rng = np.random.RandomState(seed)
    lbl = np.zeros(shape, np.int32)
    grids = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    k = 0
    for _ in range(n):
        c = [rng.uniform(r[1], s - r[1]) for s in shape]; rad = rng.uniform(*r)
        m = sum((g - ci) ** 2 for g, ci in zip(grids, c)) <= rad * rad
        if (lbl[m] == 0).all():
            k += 1; lbl[m] = k
    return lbl


class _FakeModel(object):
    """predict_instances = ground-truth labels of the block (as the reference's test_cover does with relabelling)"""
    def __init__(self, ndim, grid):
        from stardist_amd.models.config import Config2D, Config3D
        self.config = Config2D() if ndim == 2 else Config3D()
        self._grid = grid

    def _axes_div_by(self, axes): return tuple(self._grid if a != "C" else 1 for a in axes)

    def _axes_tile_overlap(self, axes): return tuple(0 for a in axes)

    def predict_instances(self, x, **kwargs):
        from scipy import ndimage as ndi
        from stardist_amd.matching import relabel_sequential
        lab = relabel_sequential(np.asarray(x).astype(np.int32))[0]
        objs = ndi.find_objects(lab)
        pts = np.array([[0.5 * (s.start + s.stop) for s in o] for o in objs]).reshape(len(objs), x.ndim)
        return lab, dict(points=pts, prob=np.ones(len(objs)), coord=np.zeros((len(objs), x.ndim, 4)) + pts[:, :, None])


@pytest.mark.parametrize("shape,axes,block,overlap,ctx,grid", [((160, 200), "YX", 64, 16, 8, 1), ((150, 131), "YX", (48, 64), 16, (4, 8), 2),
                                                                 ((40, 80, 72), "ZYX", (32, 40, 40), 12, 2, 1)])
def test_cover_filter_reassemble_is_identity(shape, axes, block, overlap, ctx, grid):
    from stardist_amd.big import predict_instances_big
    from stardist_amd.matching import relabel_sequential
    gt = _gt_labels(shape, n=80 if len(shape) == 2 else 40)
    model = _FakeModel(len(shape), grid)
    labels, polys = predict_instances_big(model, gt, axes, block, overlap, context=ctx, show_progress=False)
    assert len(polys["prob"]) == gt.max() == labels.max()
    # same partition into objects (ids may be permuted by block order)
    assert np.array_equal(labels > 0, gt > 0)
    pairs = np.unique(np.stack([labels[gt > 0], gt[gt > 0]], 1), axis=0)
    assert len(pairs) == gt.max()
    # global coordinates: every reported centre lies inside its object
    for p in polys["points"]:
        assert gt[tuple(int(v) for v in p)] > 0


def test_object_larger_than_overlap_raises():
    from stardist_amd.big import predict_instances_big
    gt = np.zeros((128, 128), np.int32); gt[10:100, 10:100] = 1
    with pytest.raises(RuntimeError):
        predict_instances_big(_FakeModel(2, 1), gt, "YX", 64, 16, context=4, show_progress=False)


def _worker(rank, world, port, q, memmap_path):
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stardist_amd.big import predict_instances_big
    from test_cpu_big import _FakeModel, _gt_labels
    gt = _gt_labels((160, 200), n=80)
    out = None
    if memmap_path:      # the streaming form: every rank opens the same file and writes the write regions of its own blocks
        out = np.lib.format.open_memmap(memmap_path, mode="r+")
    labels, polys = predict_instances_big(_FakeModel(2, 1), gt, "YX", 64, 16, context=8, show_progress=False, labels_out=out)
    if not memmap_path:
        assert (labels is None) == (rank != 0), "without a shared labels_out the label image lives on rank 0 only (ADVICE r2)"
    q.put((rank, None if (memmap_path or labels is None) else np.asarray(labels), polys["points"], polys["prob"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("shared", [False, True])
def test_sharded_equals_sequential_gloo_world2(shared, tmp_path):
    """N>1 path: blocks dealt round-robin to 2 ranks.  Object dict identical to the single-process loop on every rank; the label
    image is complete on rank 0 (block labels sent point-to-point, no all_reduce of the image) or in the shared memmap."""
    import torch.multiprocessing as mp
    from stardist_amd.big import predict_instances_big
    gt = _gt_labels((160, 200), n=80)
    ref_labels, ref_polys = predict_instances_big(_FakeModel(2, 1), gt, "YX", 64, 16, context=8, show_progress=False)
    path = ""
    if shared:
        path = str(tmp_path / "labels.npy")
        mm = np.lib.format.open_memmap(path, mode="w+", dtype=np.int32, shape=gt.shape); mm[...] = 0; mm.flush(); del mm
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300 + (1 if shared else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, path)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs: p.join(60)
    for rank, labels, pts, prob in res:
        if rank == 0 and not shared:
            assert np.array_equal(labels, ref_labels)
        assert np.array_equal(pts, ref_polys["points"]) and len(prob) == len(ref_polys["prob"])
    if shared:
        assert np.array_equal(np.load(path), ref_labels)


# ---------------------------------------------------------------- design A: sharded prediction + final cross-tile NMS
class _FieldModel(object):
    """'Network' = identity on a (H, W, 1 + n_rays) field holding prob and dist; NMS / rasteriser = the oracle (compiled
    reference NMS + numpy port of the Python rasteriser), i.e. exactly what the product calls on the GPU."""
    n_rays = 32

    def __init__(self):
        from stardist_amd.models.config import Config2D
        self.config = Config2D(n_rays=self.n_rays, n_channel_in=1 + self.n_rays)

        class T: prob, nms = 0.5, 0.4
        self.thresholds = T()

    def _axes_div_by(self, axes): return tuple(1 for a in axes)

    def _axes_tile_overlap(self, axes): return tuple(0 for a in axes)

    def predict_sparse(self, x, axes=None, prob_thresh=None, **kw):
        from oracle import port
        prob, dist = x[..., 0], x[..., 1:]
        mask = port.ind_prob_thresh(prob, self.thresholds.prob if prob_thresh is None else prob_thresh, b=2)
        return prob[mask], dist[mask], np.stack(np.where(mask), 1)

    def _order(self, prob): return np.argsort(prob, kind="stable")[::-1]

    def _nms_sparse(self, dist, prob, points, nms_thresh=None, **kw):
        from oracle import ref
        ind = self._order(prob)
        keep = ref.stardist2d().c_non_max_suppression_inds(np.ascontiguousarray(dist[ind], np.float32), np.ascontiguousarray(points[ind], np.float32),
                                                           1, 1, 0, np.float32(self.thresholds.nms if nms_thresh is None else nms_thresh))
        return ind[keep]

    def _instances_from_prediction(self, shape, prob, dist, points=None, prob_thresh=None, nms_thresh=None, return_labels=True, **kw):
        s = self._nms_sparse(dist, prob, points, nms_thresh)
        return self._instances_from_survivors(shape, points[s], prob[s], dist[s], return_labels=return_labels)

    def _instances_from_survivors(self, shape, p, pr, d, return_labels=True, window=None, **kw):
        from oracle import port
        labels = port.polygons_to_label(d, p, shape, prob=pr) if return_labels else None
        if window is not None:
            (y0, x0), (h, w) = window
            return labels[y0:y0 + h, x0:x0 + w].copy(), None
        return labels, dict(coord=port.dist_to_coord(d, p), points=p, prob=pr)


def _field(shape=(192, 224), seed=3):
    """StarDist-style targets of a synthetic nuclei image: prob = normalised distance transform, dist = reference c_star_dist"""
    from scipy import ndimage as ndi
    from oracle import ref, synth
    lbl = synth.s2d_nuclei_labels(shape[0], shape[1], seed=seed)[0].astype(np.uint16)
    dist = ref.stardist2d().c_star_dist(lbl, _FieldModel.n_rays, 1, 1)
    edt = ndi.distance_transform_edt(lbl > 0)
    mx = ndi.maximum(edt, lbl, index=np.arange(0, lbl.max() + 1)); mx[0] = 1
    prob = (edt / np.maximum(mx[lbl], 1e-6)).astype(np.float32)
    prob += np.random.RandomState(seed).uniform(0, 1e-3, prob.shape).astype(np.float32)   # break exact ties
    return np.concatenate([prob[..., None], dist.astype(np.float32)], -1), lbl


def _sharded_worker(rank, world, port_, q, mm_path=None):
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port_)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stardist_amd.big import predict_instances_sharded
    from test_cpu_big import _FieldModel, _field
    x, _ = _field()
    m = _FieldModel()
    labels, res = predict_instances_sharded(m, x, "YXC", 96, 32, context=16)
    st = dict(m._last_sharded_stats)
    # rank-local tiles (nothing moved) and a shared memmap written by the owners
    tiles, _ = predict_instances_sharded(m, x, "YXC", 96, 32, context=16, labels_out="local")
    mm = np.load(mm_path, mmap_mode="r+")
    predict_instances_sharded(m, x, "YXC", 96, 32, context=16, labels_out=mm)
    # bench.py's combination at N > 1: tiles stay local, the result dict stays on rank 0
    tiles_b, res_b = predict_instances_sharded(m, x, "YXC", 96, 32, context=16, labels_out="local", broadcast_result=False)
    assert (res_b is None) == (rank != 0) and len(tiles_b) == len(tiles)
    assert all(np.array_equal(a[2].numpy(), b[2].numpy()) for a, b in zip(tiles, tiles_b))
    assert rank != 0 or np.array_equal(res_b["points"], res["points"])
    q.put((rank, labels, res["points"], res["prob"], [(bi, tuple((s.start, s.stop) for s in sl), t.numpy()) for bi, sl, t in tiles], st))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_cross_tile_nms_equals_whole_image(refmods, tmp_path):
    """design A: per-block local NMS + survivor exchange + final NMS on rank 0 == predict_instances on the whole image"""
    from stardist_amd.big import predict_instances_sharded
    m = _FieldModel()
    x, lbl = _field()
    p, d, pts = m.predict_sparse(x)
    ref_labels, ref_res = m._instances_from_prediction(x.shape[:2], p, d, points=pts)
    assert len(ref_res["prob"]) >= 0.9 * lbl.max() > 10
    labels, res = predict_instances_sharded(m, x, "YXC", 96, 32, context=16)              # single process, 20 blocks
    assert np.array_equal(res["points"], ref_res["points"]) and np.array_equal(labels, ref_labels)
    st = m._last_sharded_stats
    # the cross-tile NMS only sees the survivors near a write-region boundary; the others are final after their block's NMS
    assert st["band"] + st["interior"] == st["unique"] <= st["gathered"] and st["interior"] > 0 and st["band"] > 0 and st["instances"] == len(ref_res["prob"])
    tiles, _ = predict_instances_sharded(m, x, "YXC", 96, 32, context=16, labels_out="local")
    assert len(tiles) == 20 and all(np.array_equal(t.numpy(), ref_labels[sl]) for _, sl, t in tiles)
    import torch.multiprocessing as mp
    mm_path = str(tmp_path / "labels.npy")
    mm = np.lib.format.open_memmap(mm_path, mode="w+", dtype=np.int32, shape=x.shape[:2]); mm[...] = 0; mm.flush(); del mm
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port_ = 29950 + os.getpid() % 40
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port_, q, mm_path)) for r in range(2)]
    for pr in procs: pr.start()
    out = [q.get(timeout=240) for _ in range(2)]
    for pr in procs: pr.join(60)
    seen = set()
    for rank, lab, pts2, prob2, tiles2, st2 in out:
        assert np.array_equal(pts2, ref_res["points"]) and np.allclose(prob2, ref_res["prob"]), rank
        if rank == 0:
            assert np.array_equal(lab, ref_labels)
            # exact-size exchange: what crosses a link is the other rank's records, nothing is padded
            assert st2["gathered"] == st["gathered"] and st2["band"] == st["band"] and st2["interior"] == st["interior"]
            assert st2["exact_record_bytes"] == st["gathered"] * (32 + 1 + 2 + 1) * 4 and 0 < st2["gathered_bytes"] < st2["exact_record_bytes"]
        else:
            assert lab is None
        for bi, sl, t in tiles2:                                  # every rank rendered the write regions of ITS blocks, with global ids
            assert bi % 2 == rank and np.array_equal(t, ref_labels[tuple(slice(a, b) for a, b in sl)])
            seen.add(bi)
    assert seen == set(range(20))
    assert np.array_equal(np.load(mm_path), ref_labels)


def _sharded_worker4(rank, world, port_, q, block):
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port_)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stardist_amd.big import predict_instances_sharded
    from test_cpu_big import _FieldModel, _field
    x, _ = _field()
    m = _FieldModel()
    tiles, res = predict_instances_sharded(m, x, "YXC", block, 32, context=16, labels_out="local", broadcast_result=False)
    st = dict(m._last_sharded_stats)
    q.put((rank, None if res is None else (res["points"], res["prob"]), [(bi, tuple((s.start, s.stop) for s in sl), t.numpy()) for bi, sl, t in tiles], st))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,block,n_blocks", [(4, 120, 9), (4, 128, 6), (8, 96, 20)])
def test_sharded_gloo_uneven_block_counts(refmods, world, block, n_blocks):
    """four ranks with block counts that do not divide by four (9 -> 3/2/2/2, 6 -> 2/2/1/1) and EIGHT ranks with 20 blocks (3/3/3/3/2/2/2/2, the
    shape of the driver's 8-GPU run): the form bench.py runs at N > 1 (owner-side tiles, result dict on rank 0 only) equals
    predict_instances on the whole image, every block is rendered by exactly its owner, and the exchange moves exactly the records of
    the other ranks (no padding to the largest rank: bytes over the links <= the exact record bytes)"""
    import torch.multiprocessing as mp
    m = _FieldModel()
    x, lbl = _field()
    p, d, pts = m.predict_sparse(x)
    ref_labels, ref_res = m._instances_from_prediction(x.shape[:2], p, d, points=pts)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port_ = 29850 + (os.getpid() + block) % 60
    procs = [ctx.Process(target=_sharded_worker4, args=(r, world, port_, q, block)) for r in range(world)]
    for pr in procs: pr.start()
    out = [q.get(timeout=600) for _ in range(world)]
    for pr in procs: pr.join(60)
    seen = {}
    local = 0
    for rank, res, tiles, st in out:
        assert (res is None) == (rank != 0)
        local += st["local_survivors"]
        if rank == 0:
            assert np.array_equal(res[0], ref_res["points"]) and np.allclose(res[1], ref_res["prob"])
            assert st["instances"] == len(ref_res["prob"]) and st["band"] + st["interior"] == st["unique"]
            st0 = st
        assert st["blocks"] == len([b for b in range(n_blocks) if b % world == rank]) == len(tiles)
        for bi, sl, t in tiles:
            assert bi % world == rank and bi not in seen and np.array_equal(t, ref_labels[tuple(slice(a, b) for a, b in sl)])
            seen[bi] = rank
    assert sorted(seen) == list(range(n_blocks))
    W = 32 + 1 + 2 + 1
    assert st0["gathered"] == local and st0["exact_record_bytes"] == local * W * 4
    assert st0["gathered_bytes"] == (local - st0["local_survivors"]) * W * 4 <= 1.05 * st0["exact_record_bytes"]


@pytest.mark.parametrize("name,axes", [("2d", "YX"), ("2dg", "YX"), ("3d", "ZYX")])
def test_cover_crop_filter_objects_equal_reference_golden(name, axes):
    """BlockND.cover + read + crop_context + filter_objects (responsibility rule, coordinate translation) against outputs of the
    reference's own classes with the real skimage regionprops (tests/golden/make_big_filter_golden.py)"""
    from stardist_amd.big import BlockND
    g = np.load(os.path.join(ROOT, "tests", "golden", "big_filter.npz"))
    gt = g[name + "_gt"]
    bs, mo, ctx, grid = [tuple(int(v) for v in r) for r in g[name + "_args"]]
    blocks = BlockND.cover(gt.shape, axes, bs, mo, ctx, grid)
    assert len(blocks) == int(g[name + "_nblocks"])
    for bi, block in enumerate(blocks):
        sub = block.read(gt, axes=axes)
        ids = np.unique(sub); ids = ids[ids > 0]
        lab = np.zeros_like(sub)
        for j, v in enumerate(ids, 1):
            lab[sub == v] = j
        pts = np.array([np.mean(np.nonzero(lab == j), axis=1) for j in range(1, len(ids) + 1)]).reshape(len(ids), gt.ndim)
        lf, pf = block.filter_objects(block.crop_context(lab, axes=axes), dict(points=pts, prob=np.linspace(1, 0.5, len(ids))), axes=axes)
        assert np.array_equal(lf, g["%s_b%d_labels" % (name, bi)]), (name, bi)
        assert np.allclose(pf["points"], g["%s_b%d_points" % (name, bi)]) and np.allclose(pf["prob"], g["%s_b%d_prob" % (name, bi)])
        # the tensor form (bounding boxes by scatter-min/max, what runs on the device) gives the same block
        import torch
        lt, pt = block.filter_objects(torch.from_numpy(np.ascontiguousarray(block.crop_context(lab, axes=axes))), dict(points=pts, prob=np.linspace(1, 0.5, len(ids))), axes=axes)
        assert np.array_equal(lt.numpy(), lf) and np.array_equal(pt["points"], pf["points"])


def test_bench_dry_collectives_gloo_world2():
    """`bench.py --gpus 2 --dry-collectives` as the driver would launch it: the sharded path's exchange on gloo with the byte count of
    every rank asserted inside (exact sizes, nothing padded) and every object of the synthetic set reported once"""
    import json
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-collectives"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["ok"] and d["world"] == 2 and d["sent_bytes_per_rank"][0] == 0
    assert d["gathered_bytes"] == d["records_per_rank"][1] * d["record_bytes"] < d["padded_gather_would_move"]
    assert d["unique"] == d["objects"] and d["duplicates_dropped"] > 0


def test_bench_sharded_leg_plumbing_gloo_world2():
    """bench.run_sharded_leg -- the N > 1 headline of bench.py -- under gloo with two ranks and a stand-in model: W untimed passes and exactly K
    timed ones, the per-rank statistics arrive on rank 0, the blocks of the one input are dealt over both ranks and every object is reported once"""
    import json
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tests", "_bench_sharded_leg_worker.py")],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["passes"] == 3 and d["warm_passes"] == 2 and d["scaling"] == "strong"
    assert d["_calls_per_rank_min"] == 1 + 2 + 3                  # one-block warm-up (not distributed) + W + K passes on every rank
    assert len(d["per_rank"]) == 2 and all(p["blocks"] > 0 for p in d["per_rank"]) and d["blocks"] == sum(p["blocks"] for p in d["per_rank"]) == 16
    assert d["instances"] == d["_objects"] and d["band_survivors"] + d["interior_survivors"] == d["_objects"]
    assert d["gathered_bytes"] <= d["exact_record_bytes"] and d["value"] > 0 and "predicted_scaling" not in d


def _sharded_worker_edge(rank, world, port_, q, case):
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port_)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stardist_amd.big import predict_instances_sharded
    from test_cpu_big import _FieldModel, _field
    x, _ = _field()
    if case == "background":
        x = x.copy(); x[..., 0] = 0                                  # nothing above the threshold anywhere: zero candidates on every rank
    elif case == "one_corner":
        x = x.copy(); x[96:, :, 0] = 0; x[:, 112:, 0] = 0            # candidates in the first block only: the other ranks send nothing
    m = _FieldModel()
    block = 160 if case == "idle_rank" else 96                       # 160 -> 4 blocks for 6 ranks: two ranks own no block at all
    labels, res = predict_instances_sharded(m, x, "YXC", block, 32, context=16)
    tiles, _ = predict_instances_sharded(m, x, "YXC", block, 32, context=16, labels_out="local", broadcast_result=False)
    st = dict(m._last_sharded_stats)
    q.put((rank, labels, res["points"], [(bi, tuple((s.start, s.stop) for s in sl), t.numpy()) for bi, sl, t in tiles], st))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case,world", [("background", 2), ("one_corner", 3), ("idle_rank", 6)])
def test_sharded_gloo_degenerate_exchanges(refmods, case, world):
    """the exchange when there is little or nothing to exchange: no candidate at all (every rank sends zero records, the final list is
    empty, every tile is background), candidates in one block only (all but one rank send nothing; zero-size messages are never posted),
    more ranks than blocks (ranks that own no block take part in every collective and render nothing)"""
    import torch.multiprocessing as mp
    m = _FieldModel()
    x, lbl = _field()
    if case == "background":
        x = x.copy(); x[..., 0] = 0
    elif case == "one_corner":
        x = x.copy(); x[96:, :, 0] = 0; x[:, 112:, 0] = 0
    p, d, pts = m.predict_sparse(x)
    if len(p):
        ref_labels, ref_res = m._instances_from_prediction(x.shape[:2], p, d, points=pts)
    else:
        ref_labels, ref_res = np.zeros(x.shape[:2], np.int32), dict(points=np.zeros((0, 2), np.int64))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port_ = 29700 + (os.getpid() + 7 * world) % 80
    procs = [ctx.Process(target=_sharded_worker_edge, args=(r, world, port_, q, case)) for r in range(world)]
    for pr in procs: pr.start()
    out = [q.get(timeout=600) for _ in range(world)]
    for pr in procs: pr.join(60)
    seen = set()
    for rank, lab, pts2, tiles, st in out:
        assert np.array_equal(np.asarray(pts2).reshape(-1, 2), ref_res["points"]), (case, rank)
        if rank == 0:
            assert np.array_equal(lab, ref_labels)
            assert st["instances"] == len(ref_res["points"])
            if case == "background":
                assert st["gathered"] == 0 and st["gathered_bytes"] == 0
        else:
            assert lab is None
        if case == "idle_rank" and rank >= 4:
            assert st["blocks"] == 0 and tiles == []
        for bi, sl, t in tiles:
            assert bi % world == rank and np.array_equal(t, ref_labels[tuple(slice(a, b) for a, b in sl)])
            seen.add(bi)
    assert len(seen) == (4 if case == "idle_rank" else 20)


# ---------------------------------------------------------------- design A in 3D on the CPU (the relabel of the owners' tiles needs a collective)
class _FieldModel3D(object):
    """3D twin of _FieldModel: 'network' = identity on a (Z, Y, X, 1 + n_rays) field; NMS and polyhedron rasteriser = the compiled reference
    natives; the windowed raster hands back the polyhedra's running numbers (as the product's windowed native does) and the whole-volume
    form closes the label ids up (model3d.py:646 relabel_sequential)"""
    n_rays = 16

    def __init__(self):
        from stardist_amd.models.config import Config3D
        from stardist_amd.rays3d import rays_from_json
        self.config = Config3D(rays=self.n_rays, n_channel_in=1 + self.n_rays)
        self.rays = rays_from_json(self.config.rays_json)

        class T: prob, nms = 0.5, 0.3
        self.thresholds = T()

    def _axes_div_by(self, axes): return tuple(1 for a in axes)

    def _axes_tile_overlap(self, axes): return tuple(0 for a in axes)

    def predict_sparse(self, x, axes=None, prob_thresh=None, **kw):
        from oracle import port
        prob, dist = x[..., 0], x[..., 1:]
        mask = port.ind_prob_thresh(prob, self.thresholds.prob if prob_thresh is None else prob_thresh, b=2)
        return prob[mask], dist[mask], np.stack(np.where(mask), 1)

    def _nms_sparse(self, dist, prob, points, nms_thresh=None, **kw):
        from oracle import ref
        ind = np.argsort(prob, kind="stable")[::-1]
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        keep = ref.stardist3d().c_non_max_suppression_inds(f32(dist[ind]), f32(points[ind]), f32(self.rays.vertices), np.ascontiguousarray(self.rays.faces, np.int32),
                                                           f32(prob[ind]), 1, 1, 0, np.float32(self.thresholds.nms if nms_thresh is None else nms_thresh))
        return ind[keep.astype(bool)]

    def _raster(self, shape, p, d):
        from oracle import ref
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        if len(p) == 0:
            return np.zeros(shape, np.int32)
        return ref.stardist3d().c_polyhedron_to_label(f32(d), f32(p), f32(self.rays.vertices), np.ascontiguousarray(self.rays.faces, np.int32),
                                                      np.arange(1, len(p) + 1, dtype=np.int32), 0, 0, 0, 0, tuple(int(s) for s in shape))

    def _instances_from_prediction(self, shape, prob, dist, points=None, **kw):
        s = self._nms_sparse(dist, prob, points)
        return self._instances_from_survivors(shape, points[s], prob[s], dist[s])

    def _instances_from_survivors(self, shape, p, pr, d, return_labels=True, window=None, **kw):
        from stardist_amd.matching import relabel_sequential
        labels = self._raster(shape, p, d) if return_labels else None
        if window is not None:
            (z0, y0, x0), (nz, ny, nx) = window
            return labels[z0:z0 + nz, y0:y0 + ny, x0:x0 + nx].copy(), None
        if labels is not None:
            labels = relabel_sequential(labels)[0]
        return labels, dict(dist=d, points=p, prob=pr)


def _field3d(n=72, seed=2):
    from oracle import synth
    m = _FieldModel3D()
    d, p, s, _ = synth.s3d_nuclei(n, m.rays.vertices, spacing=24, R=(6, 9), rc=2, seed=seed)
    x = np.zeros((n, n, n, 1 + m.n_rays), np.float32)
    pi = p.astype(np.int64)
    x[pi[:, 0], pi[:, 1], pi[:, 2], 0] = s
    x[pi[:, 0], pi[:, 1], pi[:, 2], 1:] = d
    return x


def _sharded_worker3d(rank, world, port_, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port_)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stardist_amd.big import predict_instances_sharded
    from test_cpu_big import _FieldModel3D, _field3d
    x = _field3d()
    m = _FieldModel3D()
    labels, res = predict_instances_sharded(m, x, "ZYXC", 48, 16, context=8)
    tiles, _ = predict_instances_sharded(m, x, "ZYXC", 48, 16, context=8, labels_out="local", broadcast_result=False)
    q.put((rank, labels, res["points"], [(bi, tuple((s.start, s.stop) for s in sl), t.numpy()) for bi, sl, t in tiles], dict(m._last_sharded_stats)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_3d_equals_whole_volume_single_process_and_gloo_world2(refmods):
    """design A in 3D: blocks of a 72^3 field of spheres -> local NMS (compiled reference) -> exchange -> cross-tile NMS -> owner-side windowed
    rasters whose label ids are closed up over ALL ranks' tiles (one all_reduce(MAX) of the 'id is visible somewhere' table,
    model3d.py:646 relabel_sequential on the whole volume): == predict_instances on the whole volume, in one process and over two gloo ranks"""
    import torch.multiprocessing as mp
    from stardist_amd.big import predict_instances_sharded
    m = _FieldModel3D()
    x = _field3d()
    p, d, pts = m.predict_sparse(x)
    ref_labels, ref_res = m._instances_from_prediction(x.shape[:3], p, d, points=pts)
    assert len(ref_res["prob"]) >= 20 and ref_labels.max() <= len(ref_res["prob"])
    labels, res = predict_instances_sharded(m, x, "ZYXC", 48, 16, context=8)
    assert np.array_equal(res["points"], ref_res["points"]) and np.array_equal(labels, ref_labels)
    st = m._last_sharded_stats
    assert st["blocks"] == 27 and st["band"] > 0 and st["instances"] == len(ref_res["prob"])
    tiles, _ = predict_instances_sharded(m, x, "ZYXC", 48, 16, context=8, labels_out="local")
    assert len(tiles) == 27 and all(np.array_equal(t.numpy(), ref_labels[sl]) for _, sl, t in tiles)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port_ = 29800 + os.getpid() % 40
    procs = [ctx.Process(target=_sharded_worker3d, args=(r, 2, port_, q)) for r in range(2)]
    for pr in procs: pr.start()
    out = [q.get(timeout=600) for _ in range(2)]
    for pr in procs: pr.join(60)
    seen = set()
    for rank, lab, pts2, tiles2, st2 in out:
        assert np.array_equal(pts2, ref_res["points"]), rank
        assert (lab is None) == (rank != 0)
        if rank == 0:
            assert np.array_equal(lab, ref_labels)
            assert st2["gathered"] == st["gathered"] and st2["band"] == st["band"]
        for bi, sl, t in tiles2:
            assert bi % 2 == rank and np.array_equal(t, ref_labels[tuple(slice(a, b) for a, b in sl)]), (rank, bi)
            seen.add(bi)
    assert seen == set(range(27))


# ---------------------------------------------------------------- round 6: every rank holds only ITS blocks of the input (ShardedInput over a memmap)
def _sharded_worker_memmap(rank, world, port_, q, path, block):
    import torch.distributed as dist
    from stardist_amd.big import ShardedInput, predict_instances_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port_)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _FieldModel()
    src = np.load(path, mmap_mode="r")                                 # the shared source: nobody reads all of it
    x = ShardedInput.for_rank(m, src, "YXC", block, 32, 16, rank=rank, world=world)
    held = (len(x._held), x.bytes_held)
    reads = []
    orig = ShardedInput._load

    def spy(self, slices):                                             # a read region that was not prefetched would go to the source again
        reads.append(tuple((s.start, s.stop) for s in slices))
        return orig(self, slices)
    ShardedInput._load = spy
    tiles, res = predict_instances_sharded(m, x, "YXC", block, 32, context=16, labels_out="local")
    q.put((rank, res["points"], [(bi, tuple((s.start, s.stop) for s in sl), t.numpy()) for bi, sl, t in tiles], held, reads, src.nbytes))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_gloo_world8_every_rank_reads_only_its_blocks_from_a_memmap(refmods, tmp_path):
    """SURVEY 8e / VERDICT r5 #7: the input is a memmap on disk; rank r prefetches the read regions (block + context) of blocks r, r + 8, ...
    -- 20 blocks over 8 ranks -- and nothing else: the result equals the whole-image prediction, no rank touches the source during the pass,
    and what a rank holds is about its share of the blocks, not the whole input"""
    import torch.multiprocessing as mp
    m = _FieldModel()
    x, lbl = _field()
    p, d, pts = m.predict_sparse(x)
    ref_labels, ref_res = m._instances_from_prediction(x.shape[:2], p, d, points=pts)
    path = str(tmp_path / "slide.npy")
    np.save(path, x)
    world, block = 8, 96
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port_ = 29800 + os.getpid() % 90
    procs = [ctx.Process(target=_sharded_worker_memmap, args=(r, world, port_, q, path, block)) for r in range(world)]
    for pr in procs: pr.start()
    out = [q.get(timeout=600) for _ in range(world)]
    for pr in procs: pr.join(60)
    seen, total_held = set(), 0
    for rank, pts2, tiles, (n_held, bytes_held), reads, nbytes in out:
        assert np.array_equal(np.asarray(pts2).reshape(-1, 2), ref_res["points"]), rank
        assert n_held == len([b for b in range(20) if b % world == rank]) == len(tiles)
        assert reads == [], (rank, reads)                              # every block.read was served from what the rank holds
        # 2-3 of the 20 (heavily overlapping: 96^2 blocks of a 192 x 224 field) read regions, not the whole input
        assert bytes_held <= n_held * 96 * 96 * x.shape[2] * 4, (rank, bytes_held, nbytes)
        total_held += bytes_held
        for bi, sl, t in tiles:
            assert bi % world == rank and np.array_equal(t, ref_labels[tuple(slice(a, b) for a, b in sl)])
            seen.add(bi)
    assert len(seen) == 20


class _ContextHungryModel(_FieldModel):
    """a 'network' whose prediction degrades towards the border of what it is given (as a real one does with too little context): the
    probability of a pixel is lowered by up to 20 % within 24 pixels of the block's border"""

    def predict_sparse(self, x, axes=None, prob_thresh=None, **kw):
        from oracle import port
        H, W = x.shape[:2]
        yy, xx = np.mgrid[:H, :W]
        depth = np.minimum(np.minimum(yy, H - 1 - yy), np.minimum(xx, W - 1 - xx)).astype(np.float32)
        prob = x[..., 0] * (0.8 + 0.2 * np.minimum(depth, 24.0) / 24.0)
        mask = port.ind_prob_thresh(prob, self.thresholds.prob if prob_thresh is None else prob_thresh, b=2)
        return prob[mask], x[..., 1:][mask], np.stack(np.where(mask), 1)


def test_band_duplicates_come_from_the_block_they_lie_deepest_in(refmods):
    """with a context smaller than what the 'network' needs, a candidate in the overlap of two write regions is reported by both blocks
    with DIFFERENT probabilities; the report of the block it lies deepest in (the reference's responsibility rule applied to a point,
    big.py:89-122) is the one that enters the cross-tile NMS -- not the report of whichever block comes first"""
    from stardist_amd.big import predict_instances_sharded, sharded_cover
    m = _ContextHungryModel()
    x, _ = _field()
    predict_instances_sharded(m, x, "YXC", 96, 32, context=8, return_labels=False, keep_debug=True)
    dbg = m._last_sharded_debug
    blocks, axes_n, _ = sharded_cover(m, x.shape, "YXC", 96, 32, 8)
    ext = np.array([[[t.start, t.end] for t in b.blocks_for_axes("YX")] for b in blocks], np.float64)
    pts, blk = dbg["points"].numpy(), dbg["block"].numpy().astype(int)
    rp, rb = dbg["raw_band_points"].numpy(), dbg["raw_band_block"].numpy()
    reporters = {}
    for (y, xx), b in zip(rp, rb):
        reporters.setdefault((int(y), int(xx)), []).append(int(b))
    n_multi = n_differ = 0
    for i in np.flatnonzero(~dbg["interior"].numpy()):
        y, xx = (int(v) for v in pts[i])
        rep = sorted(reporters[(y, xx)])
        if len(rep) < 2:
            continue
        depth = [min(min(c - ext[b, a, 0], ext[b, a, 1] - 1 - c) for a, c in enumerate((y, xx))) for b in rep]
        best = rep[int(np.argmax(depth))]                            # (argmax: the first of equal depths = the lower block index)
        n_multi += 1
        n_differ += best != rep[0]
        assert blk[i] == best, (i, (y, xx), rep, depth, blk[i])
    assert n_multi > 20 and n_differ > 5                             # (the rule differs from "the first block that reports it" on this input)
