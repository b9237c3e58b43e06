"""GPU: the block-sharded prediction over 2 ranks.  With >= 2 devices (skipped on the one-GPU box): the RCCL ("nccl") backend --
candidates stay on the device from the selection kernel through the local NMS into the gather -- equals the one-rank result.
On any box: two processes sharing cuda:0 with gloo collectives run the same multi-rank device path (everything but the RCCL transport)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    import bench
    from oracle import synth
    from stardist_amd.models import Config2D, StarDist2D
    dev = torch.device("cuda", rank)
    img = torch.from_numpy(synth.s2d_nuclei_image(1024, 1024, seed=3)).to(dev)
    model = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
    bench.calibrate_heads(model, img, frac=0.03)
    labels, res = model.predict_instances_sharded(img, "YX", block_size=512, min_overlap=64, context=64)
    st = model._last_sharded_stats
    q.put((rank, None if labels is None else np.asarray(labels), res["points"], res["prob"], st["gathered"], st["gathered_bytes"],
           st["exact_record_bytes"], [tuple(c) for c in st["rank_counts"]]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_two_ranks_rccl_equals_one_rank():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    import torch.multiprocessing as mp
    import bench
    from oracle import synth
    from stardist_amd.models import Config2D, StarDist2D
    dev = torch.device("cuda:0")
    img = torch.from_numpy(synth.s2d_nuclei_image(1024, 1024, seed=3)).to(dev)
    model = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
    bench.calibrate_heads(model, img, frac=0.03)
    l1, r1 = model.predict_instances_sharded(img, "YX", block_size=512, min_overlap=64, context=64)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=600) for _ in range(2)]
    for p in procs: p.join(120)
    rec_bytes = (32 + 1 + 2 + 1) * 4
    for rank, labels, pts, prob, gathered, nbytes, exact, counts in res:
        assert np.array_equal(pts, r1["points"]) and np.array_equal(prob, r1["prob"])
        # exact-size exchange: what crosses a link are the records of the ranks other than 0, nothing padded
        assert gathered == model._last_sharded_stats["gathered"] == sum(a + b for a, b in counts) and exact == gathered * rec_bytes
        assert nbytes == sum(a + b for a, b in counts[1:]) * rec_bytes
        if rank == 0:
            assert np.array_equal(labels, l1)


def _worker_one_gpu(rank, world, port, q, dim):
    """two ranks SHARING cuda:0, collectives over gloo (RCCL wants one device per rank): everything of the multi-rank path except the
    RCCL transport itself runs on the device -- block dealing, device-side local NMS, record exchange, broadcast of the final
    instances, owner-side window rasters, the global relabel through all_reduce(MAX)"""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model, img, args = _one_gpu_case(dim)
    tiles, res = model.predict_instances_sharded(img, labels_out="local", broadcast_result=False, **args)
    st = dict(model._last_sharded_stats)
    q.put((rank, None if res is None else (np.asarray(res["points"]), np.asarray(res["prob"])),
           [(bi, tuple((s.start, s.stop) for s in sl), t.cpu().numpy()) for bi, sl, t in tiles], st["blocks"], st["gathered"]))
    dist.barrier()
    dist.destroy_process_group()


def _one_gpu_case(dim):
    import torch
    import bench
    from oracle import synth
    from stardist_amd.models import Config2D, Config3D, StarDist2D, StarDist3D
    dev = torch.device("cuda:0")
    if dim == "3d":
        img = torch.from_numpy(synth.s3d_nuclei_image(96, seed=5)).to(dev)
        model = StarDist3D(Config3D(rays=96), basedir=None, device=dev, seed=0)
        model.thresholds = dict(prob=0.5, nms=0.3)
        bench.calibrate_heads(model, img, frac=0.02, radius=8.5, noise=0.03)
        return model, img, dict(axes="ZYX", block_size=64, min_overlap=16, context=8)
    img = torch.from_numpy(synth.s2d_nuclei_image(768, 1024, seed=3)).to(dev)
    model = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
    bench.calibrate_heads(model, img, frac=0.03)
    return model, img, dict(axes="YX", block_size=384, min_overlap=64, context=64)


@pytest.mark.parametrize("dim", ["2d", "3d"])
def test_sharded_two_ranks_on_one_gpu_equal_one_rank(dim):
    """runs on the one-GPU box: the form bench.py uses at N > 1 (owner-side tiles, result dict on rank 0) with two processes sharing the
    device == the one-rank prediction (instances, and every tile == the corresponding part of the one-rank label image)"""
    import torch.multiprocessing as mp
    model, img, args = _one_gpu_case(dim)
    l1, r1 = model.predict_instances_sharded(img, **args)
    n_blocks = model._last_sharded_stats["blocks"]
    del model
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29400 + os.getpid() % 200 + (7 if dim == "3d" else 0)
    procs = [ctx.Process(target=_worker_one_gpu, args=(r, 2, port, q, dim)) for r in range(2)]
    for p in procs: p.start()
    out = [q.get(timeout=600) for _ in range(2)]
    for p in procs: p.join(120)
    seen = set()
    for rank, res, tiles, blocks, gathered in out:
        assert (res is None) == (rank != 0)
        if rank == 0:
            assert np.array_equal(res[0], r1["points"]) and np.array_equal(res[1], r1["prob"])
        assert blocks == len([b for b in range(n_blocks) if b % 2 == rank]) == len(tiles)
        for bi, sl, t in tiles:
            assert bi % 2 == rank and np.array_equal(t, np.asarray(l1)[tuple(slice(a, b) for a, b in sl)]), (rank, bi)
            seen.add(bi)
    assert seen == set(range(n_blocks))
