"""Worker of tests/test_cpu_big.py::test_bench_sharded_leg_plumbing_gloo_world2 (launched by torch.distributed.run, gloo, CPU).

Runs bench.run_sharded_leg -- the function behind the N > 1 headline of bench.py -- on a stand-in model whose kernels are trivial, so that
what is exercised is the plumbing of that leg: W untimed passes, exactly K timed ones, barriers, the max-over-ranks clock, the per-rank
statistics gathered with all_gather_object and the JSON fields the headline is built from.  No GPU: torch.cuda.synchronize is a no-op here."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist_
    import bench
    from stardist_amd.big import predict_instances_sharded
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.max_memory_allocated = lambda *a, **k: 0
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist_.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    class Model(bench._DryModel):
        def predict_sparse(self, x, axes=None, prob_thresh=None, **kw):
            return bench._DryModel.predict_sparse(self, np.asarray(x), axes=axes, prob_thresh=prob_thresh, **kw)

        def _instances_from_survivors(self, shape, p, pr, d, return_labels=True, window=None, **kw):
            lab = np.zeros(shape if window is None else window[1], np.int32) if return_labels else None
            return lab, dict(points=p, prob=pr)

        def predict_instances_sharded(self, *a, **k):
            calls.append(k.get("distributed", None))
            return predict_instances_sharded(self, *a, **k)

    size, block = 1152, 448
    img = np.zeros((size, size), np.float32)
    g = np.arange(12, size - 12, 24)
    img[np.ix_(g, g)] = np.random.RandomState(0).uniform(0.5, 1.0, (len(g), len(g))).astype(np.float32)
    K, Wp = 3, 2
    # the input exactly as bench.py builds it at N > 1: every rank holds the read regions of ITS blocks only (ShardedInput over a source)
    model = Model()
    big = bench.sharded_input(model, img, 1, "YX", block, 64, 32, rank, world, torch.device("cpu"))
    assert type(big).__name__ == "ShardedInput" and big.bytes_held > 0 and len(big._held) == 8       # 16 blocks over 2 ranks
    out = bench.run_sharded_leg(model, big, "YX", block, 64, 32, K, world, dist_, rank, warm_passes=Wp)
    n_calls = torch.tensor([len(calls)], dtype=torch.int64)
    dist_.all_reduce(n_calls, op=dist_.ReduceOp.MIN)
    if rank == 0:
        out["_calls_per_rank_min"] = int(n_calls.item())
        out["_objects"] = int(len(g) ** 2)
        print(json.dumps(out), flush=True)
    else:
        assert out is None
    dist_.barrier()
    dist_.destroy_process_group()


if __name__ == "__main__":
    main()
