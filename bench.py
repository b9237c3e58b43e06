#!/usr/bin/env python
"""bench.py -- end-to-end predict_instances() throughput of the MI355X-native StarDist path.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON
line on rank 0.  One "step" = one full `predict_instances()` on one synthetic input already
resident in HBM: network forward (fp32, channels_last) -> threshold/compaction -> score sort ->
NMS -> label rasteriser -> labels + survivor dict back on the host.

  headline (value, Mpix/s): BASELINE.json configs[1], StarDist2D 32-ray U-Net on a 2048x2048 tile;
  second leg (value_3d, Mvox/s): configs[2], StarDist3D Rays_GoldenSpiral(96) on a 256^3 volume
  (runs in the same invocation after the 2D leg; `--skip-3d` drops it).

Every run also reports BASELINE.json configs 4/5 -- `sharded_2d` (one 16384^2 slide) and `sharded_3d` (one 1024^3 volume) through
predict_instances_sharded: blocks dealt over the ranks, local NMS, one gather of the survivors, cross-tile NMS over the band on rank
0, write regions rendered by their owners -- with t_predict / t_local_nms / t_exchange / t_final, gathered count and bytes.

N = 1: `value` is the 2048^2 tile leg (configs[1]).  N > 1 (one process per GPU, torchrun): `value` is the sharded 16384^2 slide
(strong scaling of one input, the north star's scaling curve; `scaling: "strong"`), and `value_tiles` carries the weak-scaling figure
(every rank its own 2048^2 tile, no data-path collective; all ranks' pixels / max-over-ranks time).

Weights are seeded random (no checkpoints offline).  The two 1x1 heads are re-scaled once, before
timing, so that the network's own outputs have the candidate statistics of the reference's NMS
workloads (2D: tests/test_nms2D.py:9-15, ~10 % of pixels above threshold, radius 10 +- 10 %;
3D: SURVEY.md 8d S3D-nuclei, ~0.9 % of voxels, radius 8.5 +- 3 %); otherwise a random net yields either
no or millions of candidates and the NMS / raster stages would be meaningless.  Nothing is skipped
in the timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32-input MFMA peak
MFMA_BF16_PEAK_TFLOPS = 2500.0
# HBM bytes per pair-kernel launch are NOT a constant in this file: tools/profile_traffic.py (run under rocprofv3 --pmc FETCH_SIZE /
# WRITE_SIZE in separate passes) writes them, together with the pair count of the profiled workload, to this json; bench.py
# reports them as roofline.traffic only when its own pair count matches the profiled one.
PAIR_TRAFFIC_JSON = os.path.join(ROOT, "profiles", "pair_kernel_traffic.json")


def calibrate_heads(model, img, frac=0.10, radius=10.0, noise=0.1):
    """Re-scale the prob/dist 1x1 heads (weights stay seeded-random directions) so that `frac` of the
    pixels exceed prob 0.5 and dist ~ radius*(1 +- noise)."""
    import torch
    net = model.net
    feats = {}
    h = net.features.register_forward_hook(lambda m, i, o: feats.__setitem__("f", o))
    limit = net._INDEX_LIMIT
    net._INDEX_LIMIT = 2 ** 62          # whole-volume head for the statistics (the timed path runs it in slabs)
    net.fused_heads = False             # ... through the plain modules, so that the hook sees the features
    with torch.no_grad():
        model.predict(img)
    net._INDEX_LIMIT = limit
    del net.fused_heads
    model.__dict__.pop("_graphs", None)   # the captured HIP graph of this pass has the whole-volume head baked in
    h.remove()
    f = feats["f"].float()
    with torch.no_grad():
        bshape = (1, -1) + (1,) * (f.dim() - 2)
        z = net.prob(f) - net.prob.bias.reshape(bshape)
        zs = z.flatten()
        if zs.numel() > 4_000_000:
            zs = zs[:: zs.numel() // 4_000_000]
        q = torch.quantile(zs, 1.0 - frac)
        net.prob.bias.fill_(float(-q))
        d = net.dist(f) - net.dist.bias.reshape(bshape)
        sd = float(d[:, :, ::2].std())
        net.dist.weight.mul_(radius * noise * 0.58 / max(sd, 1e-12))
        net.dist.bias.fill_(radius)
    del feats, f


def cpu_baseline_2d(img_np, model, sample, threads):
    """Reference CPU path on a bounded sample: U-Net = the same PyTorch module on CPU (stand-in for
    TF-CPU, which is not installed -- flagged deviation); post-processing = the COMPILED REFERENCE
    natives (oracle/_ref: stardist2d.cpp + Clipper + nanoflann, OpenMP) + the numpy restatement of
    the reference's Python rasteriser loop."""
    import copy
    import torch
    from oracle import port, ref
    torch.set_num_threads(threads)
    ref.stardist2d(); ref.set_threads(threads)
    x = img_np[:sample, :sample]
    net_cpu = copy.deepcopy(model.net).to("cpu").float()
    t0 = time.time()
    with torch.no_grad():
        prob, dist = net_cpu(torch.from_numpy(x)[None, None])
    prob = prob[0, 0].numpy()
    dist = np.maximum(1e-3, np.moveaxis(dist[0].numpy(), 0, -1))
    t_net = time.time() - t0
    t0 = time.time()
    mask = port.ind_prob_thresh(prob, model.thresholds.prob, b=2)
    pts = np.stack(np.where(mask), 1)
    d, s = dist[mask], prob[mask]
    ind = np.argsort(s)[::-1]
    d, s, pts = d[ind], s[ind], pts[ind]
    keep = ref.stardist2d().c_non_max_suppression_inds(np.ascontiguousarray(d, np.float32), np.ascontiguousarray(pts.astype(np.float32)),
                                                       1, 1, 0, np.float32(model.thresholds.nms))
    t_nms = time.time() - t0
    t0 = time.time()
    port.polygons_to_label(d[keep], pts[keep], prob=s[keep], shape=x.shape)
    t_ras = time.time() - t0
    tot = t_net + t_nms + t_ras
    # BASELINE.md 3.1 asks for the native post-processing at cpu_count AND at one thread: the first quarter of the candidates
    # (bounded sample) through the compiled reference NMS with a single OpenMP thread
    n1 = max(1, len(d) // 4)
    ref.set_threads(1)
    t0 = time.time()
    k1 = ref.stardist2d().c_non_max_suppression_inds(np.ascontiguousarray(d[:n1], np.float32), np.ascontiguousarray(pts[:n1].astype(np.float32)),
                                                     1, 1, 0, np.float32(model.thresholds.nms))
    t_nms1 = time.time() - t0
    ref.set_threads(threads)
    return dict(value=round(x.size / tot / 1e6, 4), unit="Mpix/s", cores=threads, kind="reference",
                nms_only={"threads_%d" % threads: {"candidates": int(len(d)), "seconds": round(t_nms, 3), "cand_per_s": round(len(d) / t_nms)},
                          "threads_1": {"candidates": int(n1), "seconds": round(t_nms1, 3), "cand_per_s": round(n1 / t_nms1), "survivors": int(k1.sum())}},
                sample="%dx%d crop of the bench image: torch-CPU U-Net %.2fs (TF-CPU stand-in) + compiled reference NMS "
                       "(oracle/_ref, %d candidates -> %d) %.2fs + numpy restatement of the Python rasteriser loop %.2fs"
                       % (sample, sample, t_net, len(d), int(keep.sum()), t_nms, t_ras))


def cpu_baseline_3d(vol_np, model, sample, threads):
    import copy
    import torch
    from oracle import port, ref
    from stardist_amd.rays3d import rays_from_json
    torch.set_num_threads(threads)
    m3 = ref.stardist3d(); ref.set_threads(threads)
    rays = rays_from_json(model.config.rays_json)
    x = vol_np[:sample, :sample, :sample]
    net_cpu = copy.deepcopy(model.net).to("cpu").float()
    t0 = time.time()
    with torch.no_grad():
        prob, dist = net_cpu(torch.from_numpy(x)[None, None])
    prob = prob[0, 0].numpy()
    dist = np.maximum(1e-3, np.moveaxis(dist[0].numpy(), 0, -1))
    t_net = time.time() - t0
    t0 = time.time()
    mask = port.ind_prob_thresh(prob, model.thresholds.prob, b=2)
    pts = np.stack(np.where(mask), 1)
    d, s = dist[mask], prob[mask]
    ind = np.argsort(s)[::-1]
    d, s, pts = np.ascontiguousarray(d[ind], np.float32), np.ascontiguousarray(s[ind], np.float32), np.ascontiguousarray(pts[ind].astype(np.float32))
    V, F = rays.vertices, rays.faces.astype(np.int32)
    keep = m3.c_non_max_suppression_inds(d, pts, V, F, s, 1, 1, 0, np.float32(model.thresholds.nms))
    t_nms = time.time() - t0
    t0 = time.time()
    m3.c_polyhedron_to_label(d[keep], pts[keep], V, F, np.arange(1, keep.sum() + 1, dtype=np.int32), 0, 0, 0, 0, x.shape)
    t_ras = time.time() - t0
    tot = t_net + t_nms + t_ras
    full = sample >= vol_np.shape[0]
    return dict(value=round(x.size / tot / 1e6, 4), unit="Mvox/s", cores=threads, kind="reference", seconds=round(tot, 2), same_size=bool(full),
                note=("the whole bench volume: a same-size comparison with value_3d" if full else
                      "measured on a crop (the full-size reference run was predicted to exceed --cpu-budget3d): any GPU/CPU ratio formed with "
                      "value_3d is an extrapolation from this crop, not a same-size comparison"),
                sample="%s of the bench volume: torch-CPU U-Net %.2fs (TF-CPU stand-in) + compiled reference 3D NMS "
                       "(oracle/_ref incl. Qhull, %d candidates -> %d) %.2fs + compiled reference rasteriser %.2fs"
                       % ("all %d^3 voxels" % sample if full else "%d^3 crop" % sample, t_net, len(d), int(keep.sum()), t_nms, t_ras))


def net_macs(model, macs):
    """multiply-accumulates per input pixel the timed network region executes: with the sparse head (models/unet.py) the distance head
    is evaluated on the candidate pixels only, outside that region -- its dense MACs are not counted"""
    if getattr(model, "_head_mode", "dense") != "sparse":
        return macs
    d = model.net.dist
    return macs - d.in_channels * d.out_channels / float(np.prod(model.config.grid))


def run_leg(model, img, steps, warmup, world, dist_):
    """W untimed + K timed predict_instances, barrier + synchronize on both sides, max over ranks.
    Returns (elapsed_s, avg net ms, last result, summed native stats)."""
    import torch
    from stardist_amd.lib import _native
    pairs = []
    orig_forward = model._net_forward

    def timed_forward(x, **kw):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); r = orig_forward(x, **kw); b.record()
        pairs.append((a, b))
        return r
    model._net_forward = timed_forward
    for _ in range(warmup):
        res = model.predict_instances(img)
    pairs.clear()
    if world > 1:
        dist_.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = {}
    for _ in range(steps):
        res = model.predict_instances(img)
        for k, v in _native.last_stats.items():
            stats[k] = stats.get(k, 0) + v
    torch.cuda.synchronize()
    if world > 1:
        dist_.barrier()
    elapsed = time.perf_counter() - t0
    net_ms = sum(a.elapsed_time(b) for a, b in pairs) / max(1, len(pairs))
    model._net_forward = orig_forward
    if world > 1:
        tt = torch.tensor([elapsed], device=img.device, dtype=torch.float64)
        dist_.all_reduce(tt, op=dist_.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed, net_ms, res, stats


def run_exact_leg(model, img, steps, warmup, world, dist_):
    """The same steps with the exact-f32 convolution kernel (csrc/conv3x3.hip, STARDIST_AMD_CONV=hand: one f32 fma chain per output)
    instead of the default split-bf16 kernel (csrc/conv3x3_bf16.hip: six bf16 MFMA products per f32 product, f32 accumulation; both are
    held to the same 1e-5 bar against float64 by tests/test_gpu_unet_parity.py and tests/test_gpu_conv3x3.py).  Reported NEXT TO `value`.
    Returns a dict, or an error note (this leg must never take the bench down)."""
    old = os.environ.get("STARDIST_AMD_CONV")
    graphs = model.__dict__.pop("_graphs", None)
    os.environ["STARDIST_AMD_CONV"] = "hand"
    try:
        elapsed, net_ms, res, _ = run_leg(model, img, steps, warmup, world, dist_)
        n = int(np.prod(img.shape))
        return {"value": round(world * n * steps / elapsed / 1e6, 3), "ms_per_step": round(1e3 * elapsed / steps, 3), "unet_forward_ms": round(net_ms, 3),
                "instances": len(res[1]["prob"]), "steps": steps,
                "arithmetic": "exact f32 MFMA (v_mfma_f32_32x32x2_f32), one fma chain per output (STARDIST_AMD_CONV=hand)"}
    except Exception as e:                       # pragma: no cover
        return {"value": None, "error": repr(e)[:200]}
    finally:
        if old is None:
            os.environ.pop("STARDIST_AMD_CONV", None)
        else:
            os.environ["STARDIST_AMD_CONV"] = old
        model.__dict__.pop("_graphs", None)
        if graphs is not None:
            model._graphs = graphs


def run_sharded_leg(model, big, axes, block, overlap, context, passes, world, dist_, rank, pipeline=True):
    """BASELINE.json configs 4/5: ONE large input, its blocks dealt round-robin to the ranks; per block network + selection + local NMS on
    the device; one gather of the surviving records to rank 0; cross-tile NMS over the band survivors only; final instances broadcast and
    every rank renders the write regions of its blocks (stardist_amd/big.py, design A of SURVEY.md 8e).  Strong scaling: the input is
    the same for every N.  N = 1: the label image comes back as ONE host array (as predict_instances returns it); N > 1: the tiles stay on
    the ranks that rendered them (labels_out="local"), as the reference's block.write leaves them in the shared output.
    pipeline: the network of block k+1 runs on the main stream while the NMS of block k runs on a second one; the warm-up runs a
    two-block input through both loops and the pipelined one is used only if it returns bit-identical instances and labels.
    Returns a dict (rank 0) or None."""
    import torch
    # warm-up on TWO blocks' worth of the input (HIP graph of the block shape, arena growth): a full pass of the 1024^3 volume is ~10 s
    warm = big[tuple(slice(0, block) for _ in range(big.dim() - 1)) + (slice(0, min(big.shape[-1], 2 * block - overlap - 2 * context)),)]
    wkw = dict(block_size=block, min_overlap=overlap, context=context, distributed=False)
    l0, r0_ = model.predict_instances_sharded(warm, axes, pipeline=False, **wkw)
    check = "serial loop"
    if pipeline:
        l1, r1_ = model.predict_instances_sharded(warm, axes, pipeline=True, **wkw)
        same = (bool(model._last_sharded_stats["pipelined"]) and np.array_equal(np.asarray(l0), np.asarray(l1))
                and all(np.array_equal(np.asarray(r0_[k]), np.asarray(r1_[k])) for k in ("points", "prob")))
        check = "two-block warm-up: pipelined == serial (labels, points, prob bit-identical)" if same else "two-block warm-up DISAGREED: serial loop used"
        pipeline = same
        del l1, r1_
    if world > 1:                          # one decision for all ranks
        flag = torch.tensor([int(pipeline)], device=big.device)
        dist_.all_reduce(flag, op=dist_.ReduceOp.MIN)
        if pipeline and not flag.item():
            check = "another rank's two-block warm-up disagreed: serial loop used"
        pipeline = bool(flag.item())
    del warm, l0, r0_
    kw = dict(block_size=block, min_overlap=overlap, context=context, broadcast_result=False, pipeline=pipeline)
    if world > 1:
        kw["labels_out"] = "local"
    if world > 1:
        dist_.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc = {}
    for _ in range(passes):
        labels, res = model.predict_instances_sharded(big, axes, **kw)
        for k, v in model._last_sharded_stats.items():
            acc[k] = acc.get(k, 0) + v
    torch.cuda.synchronize()
    if world > 1:
        dist_.barrier()
    elapsed = time.perf_counter() - t0
    mean = {k: (round(v / passes, 4) if isinstance(v, float) else v // passes) for k, v in acc.items()}
    per_rank = [None] * world
    if world > 1:
        tt = torch.tensor([elapsed], device=big.device, dtype=torch.float64)
        dist_.all_reduce(tt, op=dist_.ReduceOp.MAX)
        elapsed = float(tt.item())
        dist_.all_gather_object(per_rank, mean)
    else:
        per_rank = [mean]
    if rank != 0:
        return None
    n = int(np.prod(big.shape))
    r0 = per_rank[0]
    s_pass = elapsed / passes
    return {"value": round(n * passes / elapsed / 1e6, 3), "s_per_pass": round(s_pass, 4), "passes": passes, "scaling": "strong",
            "input_shape": list(big.shape), "block_size": block, "min_overlap": overlap, "context": context, "blocks": sum(p["blocks"] for p in per_rank),
            "instances": r0["instances"], "candidates": sum(p["candidates"] for p in per_rank),
            "gathered_survivors": r0["gathered"], "gathered_bytes": r0["gathered_bytes"], "band_survivors": r0["band"], "interior_survivors": r0["interior"],
            "pipelined": bool(r0["pipelined"]), "pipeline_check": check, "t_phase1": max(p["t_phase1"] for p in per_rank),
            "t_predict": max(p["t_predict"] for p in per_rank), "t_local_nms": max(p["t_local_nms"] for p in per_rank), "t_exchange": r0["t_exchange"],
            "t_final": r0["t_final"], "t_final_nms": r0["t_final_nms"], "t_raster": max(p["t_raster"] for p in per_rank),
            "t_final_frac": round(r0["t_final"] / s_pass, 4),
            "labels": "one host array on rank 0" if world == 1 else "rank-local tiles of the owned write regions (not gathered)",
            "per_rank": per_rank}


def guarded(fn, what):
    """(result, None) or (None, message): an extra leg that fails is reported in the JSON line instead of losing the whole line"""
    try:
        return fn(), None
    except Exception as e:                                          # noqa: BLE001 -- reported, not swallowed
        import traceback
        traceback.print_exc()
        return None, "%s failed: %r" % (what, e)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--size3d", type=int, default=256)
    ap.add_argument("--dtype", default="float32", choices=["float32", "bfloat16", "float16"])
    ap.add_argument("--skip-3d", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=2048)
    ap.add_argument("--cpu-sample3d", type=int, default=128)
    ap.add_argument("--cpu-budget3d", type=float, default=75.0, help="run the 3D CPU baseline at full size if the crop predicts at most this many seconds")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-sharded", action="store_true", help="skip the block-sharded big-input legs (configs 4/5; run by default at every N)")
    ap.add_argument("--sharded-size", type=int, default=16384)
    ap.add_argument("--sharded-size3d", type=int, default=1024)
    # block sizes (read size incl. context, as in the reference's BlockND.cover): 4480 -> 16 blocks of the 16384^2 slide (1.20x the slide's
    # pixels, an equal number per rank at N = 1, 2, 4, 8); 416 -> 27 blocks of the 1024^3 volume (1.81x; a 32-channel level of one block is
    # 9.2 GB, the 128-channel features 37 GB)
    ap.add_argument("--sharded-block", type=int, default=4480)
    ap.add_argument("--sharded-block3d", type=int, default=416)
    ap.add_argument("--skip-sharded-3d", action="store_true")
    ap.add_argument("--no-split-leg", "--no-exact-leg", dest="no_split_leg", action="store_true",
                    help="skip the extra legs with the exact-f32 convolution kernel")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist_
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local_rank)
        dist_.init_process_group("nccl", rank=rank, world_size=world)   # "nccl" IS RCCL on ROCm
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)
    threads = args.cpu_threads or min(os.cpu_count() or 1, 32)

    from oracle import synth                       # input generators only (numpy), shared with the tests
    from stardist_amd.models import Config2D, Config3D, StarDist2D, StarDist3D
    from stardist_amd.models.unet import conv_macs_per_input_pixel
    peak = MFMA_F32_PEAK_TFLOPS if args.dtype == "float32" else MFMA_BF16_PEAK_TFLOPS
    dt = {"float32": "f32", "bfloat16": "bf16", "float16": "f16"}[args.dtype]

    # ------------------------------------------------------------------ 2D leg (headline)
    H = W = args.size
    img_np = synth.s2d_nuclei_image(H, W, seed=rank)
    img = torch.from_numpy(img_np).to(dev)
    model = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0, compute_dtype=args.dtype)
    # every rank calibrates on the SAME image (seed 0), so that all ranks run identical weights -- the sharded legs deal the blocks of one
    # input over the ranks and must see one model; the timed tile of a rank is its own (seed = rank)
    calibrate_heads(model, img if rank == 0 else torch.from_numpy(synth.s2d_nuclei_image(H, W, seed=0)).to(dev))
    macs = conv_macs_per_input_pixel(model.net, model.config)
    elapsed, net_ms, res, st = run_leg(model, img, args.steps, args.warmup, world, dist_)
    # second number (SURVEY.md 8d defines the metric host-array-in): the same steps with the image handed over as a host numpy array,
    # i.e. including the 16.8 MB H2D copy; `value` stays the HBM-resident figure the contract asks for
    torch.cuda.synchronize(); t0h = time.perf_counter()
    for _ in range(args.steps):
        model.predict_instances(img_np)
    torch.cuda.synchronize(); elapsed_host = time.perf_counter() - t0h
    out = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        p, d = model.predict(img)
        n_cand = int(((p > model.thresholds.prob)[2:-2, 2:-2]).sum())
        s2 = st.get("nms2d", np.zeros(16, np.int64)) / max(1, args.steps)
        flops = 2.0 * net_macs(model, macs) * H * W
        conv_tf = flops / (net_ms * 1e-3) / 1e12
        pair_ms, pair_launches, n_pairs = s2[4] / 1e6, max(1.0, s2[5]), s2[0]
        # algorithmic bytes of the pair kernel: SURVEY.md 8(d) pair-traffic model B_pair = 2*(4R+4D) = 272 B per pair (R=32, D=2)
        pair_bytes_per_launch = 272.0 * n_pairs / pair_launches
        pair_gbs = pair_bytes_per_launch / max(1e-9, (pair_ms / pair_launches) * 1e-3) / 1e9
        stages = {"unet_forward": round(net_ms, 3), "nms_pair_kernel": round(float(pair_ms), 3),
                  "nms_exact_join_kernel": round(float(s2[6] / 1e6), 3), "nms_build_bin_neighbours": round(float(s2[7] / 1e6), 3),
                  "other(select,sort,greedy-scan,raster,d2h)": round(ms_per_step - net_ms - float(pair_ms + s2[6] / 1e6 + s2[7] / 1e6), 3)}
        from stardist_amd.models.unet import conv_mode
        mode = conv_mode() if args.dtype == "float32" else "miopen"
        # roofline of the convolutions.  Split kernel (default): every f32 multiply-accumulate is six bf16 x bf16 MFMA products, so the
        # matrix cores execute 6x the algorithmic FLOPs, on the bf16 pipe (dense peak 2.5 PFLOP/s); exact kernel: the FLOPs themselves on
        # the f32 pipe (157.3 TFLOP/s).  `achieved` / `peak` / `frac` are those of the pipe the kernel runs on; `f32_equivalent_tflops` is
        # the algorithmic rate either way.
        mult, cpeak, pipe = (6.0, MFMA_BF16_PEAK_TFLOPS, "bf16") if mode == "bf16x6" else (1.0, peak, "f32" if args.dtype == "float32" else args.dtype)
        conv_kernel = {"bf16x6": "network forward = one HIP graph: k_conv3_bf16 (hand-written split-bf16 implicit GEMM, csrc/conv3x3_bf16.hip: every 3x3 layer incl. "
                                 "folded up-sampling / concatenation / bias / ReLU; v_mfma_f32_32x32x16_bf16, six products per f32 product, f32 accumulate) + "
                                 "k_conv3_c1x32 + k_maxpool_cl4 + probability-head pass",
                       "hand": "network forward = one HIP graph: k_conv3<1> (hand-written exact-f32 MFMA implicit GEMM, csrc/conv3x3.hip) + k_conv3_c1x32 + "
                               "k_maxpool_cl4 + probability-head pass"}.get(mode, "network forward (MIOpen / CK convolution kernels, NHWC) + epilogue passes")
        roof_conv = {"bound": "mfma", "pipe": pipe, "kernel": conv_kernel, "achieved": round(conv_tf * mult, 3), "peak": cpeak,
                     "unit": "TFLOP/s", "frac": round(conv_tf * mult / cpeak, 4), "traffic": None, "flops_per_launch": flops * mult,
                     "f32_equivalent_tflops": round(conv_tf, 3), "frac_of_f32_peak_equivalent": round(conv_tf / MFMA_F32_PEAK_TFLOPS, 4), "avg_ms": round(net_ms, 3),
                     "note": "algorithmic FLOPs of the convolutions (2 x MACs, recomputed from the instantiated module; x6 executed products for the split "
                             "kernel) / HIP-event time of the whole forward pass on the caller's stream; per-kernel durations: profiles/r03_bench_kernel_stats.md"}
        roof_pair = {"bound": "hbm", "kernel": "k_pairs_beam<32,8,6,4,64> (bound-slot scan-beam polygon intersection, one pair per lane, state in LDS)", "achieved": round(pair_gbs, 3),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(pair_gbs / HBM_PEAK_GBS, 6), "traffic": None,
                     "bytes_per_launch": round(pair_bytes_per_launch), "avg_launch_ms": round(float(pair_ms / pair_launches), 4),
                     "launches_per_step": float(pair_launches), "pairs_per_step": float(n_pairs),
                     "note": "integer scan-beam sweep, one pair per lane, per-pair state lane-interleaved in LDS: bound by instruction issue under lane "
                             "divergence and by one sweep's serial latency, not by HBM; algorithmic bytes = 272 B/pair (SURVEY.md 8d)"}
        try:
            with open(PAIR_TRAFFIC_JSON) as fh:
                tj = json.load(fh)
            if abs(float(n_pairs) - tj["pairs_per_step"]) <= 0.02 * tj["pairs_per_step"] and H == tj.get("size", 2048):
                roof_pair["traffic"] = tj["bytes_per_launch"]
                roof_pair["traffic_source"] = tj["source"]
        except (OSError, KeyError, ValueError):
            pass
        dominant = roof_pair if pair_ms + s2[6] / 1e6 > net_ms else roof_conv
        out = {
            "metric": "predict_instances() Mpix/s (2D) + Mvox/s (3D) end-to-end at 1/2/4/8 GPU",
            "value": round(world * H * W * args.steps / elapsed / 1e6, 3), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dt, "data": "synthetic",
            "arithmetic": ("float32 data, float32 accumulation; 3x3 convolution products " +
                           ("as six bf16 x bf16 terms of operands split into three bf16 parts (f32-accurate: layers and networks within 3e-6 of a float64 "
                            "evaluation, tests/test_gpu_conv3x3.py, test_gpu_unet_parity.py; `exact_f32` = the same step with the exact-f32 MFMA kernel)"
                            if args.dtype == "float32" and conv_mode() == "bf16x6" else "in the named dtype") +
                           "; NMS / rasteriser in the reference's own int64 / float32 / float64 arithmetic"),
            "config": {"workload": "StarDist2D 32-ray U-Net (depth 3, 32 base filters), %dx%d synthetic fluo tile per GPU, predict_instances "
                                   "(U-Net + select + 2D NMS + polygon raster), seeded random weights, heads calibrated to ~10%% candidates "
                                   "radius 10+-10%%" % (H, W),
                       "candidates": n_cand, "survivors": len(res[1]["prob"]), "prob_thresh": model.thresholds.prob,
                       "nms_thresh": model.thresholds.nms, "parallelism": "tiles-per-gpu x%d" % world},
            "stages_ms": stages, "roofline": dominant, "roofline_convs": roof_conv, "roofline_pair_kernel": roof_pair,
            "value_host_input": {"value": round(H * W * args.steps / elapsed_host / 1e6, 3), "unit": "Mpix/s", "ms_per_step": round(1e3 * elapsed_host / args.steps, 3),
                                 "note": "same steps with the image passed as a host numpy array (H2D of the input inside the timed region), this rank only"},
        }
        if not args.no_cpu_baseline and world == 1:      # reported baseline: rank 0 at N=1 only
            try:
                out["cpu_baseline"] = cpu_baseline_2d(img_np, model, min(args.cpu_sample, H), threads)
            except Exception as e:   # oracle/_ref must have travelled with the tree
                out["cpu_baseline"] = {"value": None, "unit": "Mpix/s", "cores": 0, "kind": "reference", "sample": "unavailable: %r" % (e,)}
    if args.dtype == "float32" and not args.no_split_leg:
        r = run_exact_leg(model, img, max(1, min(args.steps, 10)), 2, world, dist_)
        if rank == 0:
            r["unit"] = "Mpix/s"
            out["exact_f32"] = r
    # ---- config 4: one 16384^2 slide (the 2048^2 synthetic tile repeated), blocks 4480 / overlap 128 / context 128, sharded over the ranks
    if not args.no_sharded:
        rep = max(1, args.sharded_size // H)
        big = torch.from_numpy(synth.s2d_nuclei_image(H, W, seed=0)).to(dev).repeat(rep, rep)
        r, err = guarded(lambda: run_sharded_leg(model, big, "YX", min(args.sharded_block, big.shape[0]), 128, 128, 2, world, dist_, rank), "sharded_2d")
        if rank == 0 and err:
            out["sharded_2d"] = {"error": err}                      # the headline stays the tile leg
        elif rank == 0:
            r["unit"] = "Mpix/s"
            out["sharded_2d"] = r
            if world > 1:
                # N > 1: the headline is ONE 16384^2 slide whose blocks are dealt over the ranks (BASELINE.json config 4, the north star's
                # scaling curve), not N independent tiles; the tile figure stays next to it
                out["value_tiles"] = {"value": out["value"], "unit": "Mpix/s", "ms_per_step": out["ms_per_step"], "scaling": "weak",
                                      "note": "every rank its own %dx%d tile, no collective" % (H, W)}
                out["value"], out["scaling"] = r["value"], "strong"
                out["ms_per_step"] = round(1e3 * r["s_per_pass"], 3)
                out["steps"], out["warmup"] = r["passes"], 1
                out["config"]["workload"] = ("predict_instances_sharded on ONE %dx%d synthetic slide (BASELINE.json config 4): blocks %d / overlap 128 / "
                                             "context 128 dealt round-robin over the ranks, local NMS per block, gather of the survivors, cross-tile "
                                             "NMS over the band on rank 0, write regions rendered by their owners" % (big.shape[0], big.shape[1], r["block_size"]))
                out["config"]["parallelism"] = "blocks-of-one-slide x%d (strong scaling; `value_tiles` = independent tile per rank)" % world
        del big
    del model, img
    torch.cuda.empty_cache()

    # ------------------------------------------------------------------ 3D leg
    if not args.skip_3d:
        S = args.size3d
        vol_np = synth.s3d_nuclei_image(S, seed=rank)
        vol = torch.from_numpy(vol_np).to(dev)
        m3 = StarDist3D(Config3D(rays=96), basedir=None, device=dev, seed=0, compute_dtype=args.dtype)
        m3.thresholds = dict(prob=0.5, nms=0.3)
        calibrate_heads(m3, vol if rank == 0 else torch.from_numpy(synth.s3d_nuclei_image(S, seed=0)).to(dev),
                        frac=0.009, radius=8.5, noise=0.03)   # near-spherical objects; same image on every rank
        macs3 = conv_macs_per_input_pixel(m3.net, m3.config)
        steps3 = max(1, min(args.steps, 5))
        elapsed3, net3_ms, res3, st3 = run_leg(m3, vol, steps3, 2, world, dist_)
        if rank == 0:
            s3 = st3.get("nms3d", np.zeros(16, np.int64)) / steps3
            ms3 = 1e3 * elapsed3 / steps3
            flops3 = 2.0 * net_macs(m3, macs3) * S ** 3
            conv3_tf = flops3 / (net3_ms * 1e-3) / 1e12
            out["value_definition"] = ("`value` / `value_3d`: input resident in HBM when the timed region starts (the bench contract); `value_host_input`: "
                                       "the same step from a host numpy array in to (labels, dict) out, SURVEY.md 8d's definition")
            out["value_3d"] = round(world * S ** 3 * steps3 / elapsed3 / 1e6, 3)
            out["unit_3d"] = "Mvox/s"
            out["ms_per_step_3d"] = round(ms3, 3)
            out["config_3d"] = {"workload": "StarDist3D Rays_GoldenSpiral(96) U-Net (depth 2), %d^3 synthetic volume per GPU, predict_instances "
                                            "(U-Net + select + 3D NMS cascade + polyhedron raster + relabel)" % S,
                                "survivors": len(res3[1]["prob"]), "steps": steps3, "nms_thresh": 0.3,
                                "cascade_calls": {"upper": float(s3[0]), "lower": float(s3[1]), "kernel_volume": float(s3[2]),
                                                  "hull_volume": float(s3[11]), "render": float(s3[3])}}
            out["stages_ms_3d"] = {"unet_forward": round(net3_ms, 3), "nms_stage3_kernel_volume": round(float(s3[8] / 1e6), 3),
                                   "nms_stage4_hull_volume": round(float(s3[9] / 1e6), 3), "nms_stage5_render": round(float(s3[10] / 1e6), 3),
                                   "other": round(ms3 - net3_ms - float((s3[8] + s3[9] + s3[10]) / 1e6), 3)}
            out["roofline_convs_3d"] = {"bound": "mfma", "pipe": pipe, "kernel": "network forward (3x3x3 layers as three z-plane units per 32-channel chunk; kernel family as `roofline_convs`)",
                                        "achieved": round(conv3_tf * mult, 3), "peak": cpeak, "unit": "TFLOP/s",
                                        "frac": round(conv3_tf * mult / cpeak, 4), "flops_per_launch": flops3 * mult,
                                        "f32_equivalent_tflops": round(conv3_tf, 3), "frac_of_f32_peak_equivalent": round(conv3_tf / MFMA_F32_PEAK_TFLOPS, 4),
                                        "avg_ms": round(net3_ms, 3)}
            if not args.no_cpu_baseline and world == 1:
                try:
                    cb = cpu_baseline_3d(vol_np, m3, min(args.cpu_sample3d, S), threads)
                    # same-size baseline when the host can afford it: the crop's time scaled by the voxel ratio predicts the full run
                    if args.cpu_sample3d < S and cb.get("seconds", 1e9) * (S / float(min(args.cpu_sample3d, S))) ** 3 <= args.cpu_budget3d:
                        crop = cb
                        cb = cpu_baseline_3d(vol_np, m3, S, threads)
                        cb["crop_run"] = {"sample": crop["sample"], "value": crop["value"]}
                    out["cpu_baseline_3d"] = cb
                except Exception as e:
                    out["cpu_baseline_3d"] = {"value": None, "unit": "Mvox/s", "cores": 0, "kind": "reference", "sample": "unavailable: %r" % (e,)}
        if args.dtype == "float32" and not args.no_split_leg:
            r = run_exact_leg(m3, vol, steps3, 1, world, dist_)
            if rank == 0:
                r["unit"] = "Mvox/s"
                out["exact_f32_3d"] = r
        # ---- config 5: one 1024^3 volume (the 256^3 synthetic volume repeated), 416^3 blocks / overlap 32 / context 32, sharded
        if not args.no_sharded and not args.skip_sharded_3d:
            rep = max(1, args.sharded_size3d // S)
            bigv = torch.from_numpy(synth.s3d_nuclei_image(S, seed=0)).to(dev).repeat(rep, rep, rep)
            r, err = guarded(lambda: run_sharded_leg(m3, bigv, "ZYX", min(args.sharded_block3d, bigv.shape[0]), 32, 32, 1, world, dist_, rank), "sharded_3d")
            if rank == 0 and err:
                out["sharded_3d"] = {"error": err}
            elif rank == 0:
                r["unit"] = "Mvox/s"
                out["sharded_3d"] = r
            del bigv
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist_.destroy_process_group()


if __name__ == "__main__":
    main()
