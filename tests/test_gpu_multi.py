"""GPU, >= 2 devices (skipped on the one-GPU box): the block-sharded prediction over 2 ranks with the RCCL ("nccl") backend --
candidates stay on the device from the selection kernel through the local NMS into the all_gather -- equals the one-rank result."""
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
    q.put((rank, None if labels is None else np.asarray(labels), res["points"], res["prob"], st["gathered"], st["gathered_bytes"]))
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
    for rank, labels, pts, prob, gathered, nbytes in res:
        assert np.array_equal(pts, r1["points"]) and np.array_equal(prob, r1["prob"])
        assert gathered == model._last_sharded_stats["gathered"] and nbytes == gathered * (32 + 1 + 2 + 1) * 4
        if rank == 0:
            assert np.array_equal(labels, l1)
