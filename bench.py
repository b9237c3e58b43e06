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
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 / fp16 MFMA peak (same rate on gfx950)
# the matrix instructions a kernel family executes per algorithmic multiply-accumulate, and the pipe they run on
CONV_FORMS = {"f16x3": (3.0, MFMA_BF16_PEAK_TFLOPS, "f16"), "bf16x6": (6.0, MFMA_BF16_PEAK_TFLOPS, "bf16"), "hand": (1.0, MFMA_F32_PEAK_TFLOPS, "f32")}
# HBM bytes per pair-kernel launch are NOT a constant in this file: tools/profile_traffic.py (run under rocprofv3 --pmc FETCH_SIZE /
# WRITE_SIZE in separate passes) writes them, together with the pair count of the profiled workload, to this json; bench.py
# reports them as roofline.traffic only when its own pair count matches the profiled one.
PAIR_TRAFFIC_JSON = os.path.join(ROOT, "profiles", "pair_kernel_traffic.json")
# the same for the convolution kernel, per forward pass (tools/pmc_forward.py + tools/conv_traffic.py, profiles/r06_conv_forward.md)
CONV_TRAFFIC_JSON = os.path.join(ROOT, "profiles", "conv_kernel_traffic.json")


def conv_traffic(which, mode, size, full):
    """HBM bytes per forward pass of the convolution kernel as profiled (None when the profile is of another kernel form / size)"""
    try:
        with open(CONV_TRAFFIC_JSON) as fh:
            tj = json.load(fh)[which]
        if mode == "f16x3" and size == full:
            return tj
    except (OSError, KeyError, ValueError):
        pass
    return None


def calibrate_heads(model, img, frac=0.10, radius=10.0, noise=0.1):
    """Re-scale the prob/dist 1x1 heads (weights stay seeded-random directions) so that `frac` of the
    pixels exceed prob 0.5 and dist ~ radius*(1 +- noise)."""
    import torch
    net = model.net
    feats = {}
    h = net.features.register_forward_hook(lambda m, i, o: feats.__setitem__("f", o))
    net.fused_heads = False             # the plain graph, so that the hook sees the features
    with torch.no_grad():
        model.predict(img)
    del net.fused_heads
    model.__dict__.pop("_graphs", None)   # the captured HIP graph of this pass has the whole-volume head baked in
    h.remove()
    f = feats["f"].float()
    with torch.no_grad():
        bshape = (1, -1) + (1,) * (f.dim() - 2)
        if f.is_cuda:                       # the 1x1 heads through the library's own kernels here as well (no framework convolution on the device)
            from stardist_amd.models.unet import _conv_bias_act
            head = lambda conv: _conv_bias_act(conv, f, 0)
        else:
            head = lambda conv: conv(f)
        z = head(net.prob) - net.prob.bias.reshape(bshape)
        zs = z.flatten()
        if zs.numel() > 4_000_000:
            zs = zs[:: zs.numel() // 4_000_000]
        q = torch.quantile(zs, 1.0 - frac)
        net.prob.bias.fill_(float(-q))
        d = head(net.dist) - net.dist.bias.reshape(bshape)
        sd = float(d[:, :, ::2].std())
        net.dist.weight.mul_(radius * noise * 0.58 / max(sd, 1e-12))
        net.dist.bias.fill_(radius)
    del feats, f


def _gpu_candidates(model, x_np):
    """the candidates the GPU path's own network + selection produce for x (score-sorted, as the NMS natives take them) and the
    keep flags of the HIP NMS on them"""
    import torch
    from stardist_amd.nms import _argsort_desc
    with torch.no_grad():
        r = model.predict_sparse_device(torch.from_numpy(np.ascontiguousarray(x_np)).to(model.device))
    prob, dist, points = r[0], r[1], r[-1]
    ind = _argsort_desc(prob)
    prob, dist, points = prob[ind].contiguous(), dist[ind].contiguous(), points[ind].contiguous()
    from stardist_amd import nms as sd_nms
    if model.config.n_dim == 2:
        keep = sd_nms.non_maximum_suppression_inds(dist, points, prob, thresh=model.thresholds.nms, verbose=0)
    else:
        from stardist_amd.lib.stardist3d import c_non_max_suppression_inds       # (the *_3d_inds wrapper would sort again)
        from stardist_amd.rays3d import rays_from_json
        rays = rays_from_json(model.config.rays_json)
        verts = torch.as_tensor(np.ascontiguousarray(rays.vertices, np.float32), device=dist.device)
        faces = torch.as_tensor(np.ascontiguousarray(rays.faces, np.int32), device=dist.device)
        keep = c_non_max_suppression_inds(dist.float().contiguous(), points.float().contiguous(), verts, faces, prob.float().contiguous(), 1, 1, 0,
                                          np.float32(model.thresholds.nms))
    keep = (keep.cpu().numpy() if torch.is_tensor(keep) else np.asarray(keep)).astype(bool)
    return dist.cpu().numpy().astype(np.float32), prob.cpu().numpy().astype(np.float32), points.cpu().numpy().astype(np.float32), keep


def cpu_baseline_2d(img_np, model, sample, threads):
    """Reference CPU path on a bounded sample: U-Net = the same PyTorch module on CPU (stand-in for TF-CPU, which is not installed --
    flagged deviation) + threshold / sort on its output; post-processing = the COMPILED REFERENCE natives (oracle/_ref: stardist2d.cpp +
    Clipper + nanoflann, OpenMP) + the numpy restatement of the reference's Python rasteriser loop, run on the candidates the GPU
    path's network produced for the same sample -- so that the reference's keep flags can be compared ONE BY ONE with the HIP NMS's
    (`parity_checked`); the candidate statistics are those of the timed workload either way."""
    import copy
    import torch
    from oracle import port, ref
    torch.set_num_threads(threads)
    ref.stardist2d(); ref.set_threads(threads)
    x = np.ascontiguousarray(img_np[:sample, :sample])
    net_cpu = copy.deepcopy(model.net).to("cpu").float()
    t0 = time.time()
    with torch.no_grad():
        prob, dist = net_cpu(torch.from_numpy(x)[None, None])
    prob = prob[0, 0].numpy()
    dist = np.maximum(1e-3, np.moveaxis(dist[0].numpy(), 0, -1))
    t_net = time.time() - t0
    t0 = time.time()
    mask = port.ind_prob_thresh(prob, model.thresholds.prob, b=2)
    pts_c = np.stack(np.where(mask), 1)
    ind = np.argsort(prob[mask])[::-1]
    d_c = dist[mask][ind]
    t_sel = time.time() - t0
    d, s, pts, keep_gpu = _gpu_candidates(model, x)
    t0 = time.time()
    keep = ref.stardist2d().c_non_max_suppression_inds(d, pts, 1, 1, 0, np.float32(model.thresholds.nms))
    t_nms = time.time() - t0
    t0 = time.time()
    port.polygons_to_label(d[keep], pts[keep], prob=s[keep], shape=x.shape)
    t_ras = time.time() - t0
    tot = t_net + t_sel + t_nms + t_ras
    # BASELINE.md 3.1 asks for the native post-processing at cpu_count AND at one thread: the first quarter of the candidates
    # (bounded sample) through the compiled reference NMS with a single OpenMP thread
    n1 = max(1, len(d) // 4)
    ref.set_threads(1)
    t0 = time.time()
    k1 = ref.stardist2d().c_non_max_suppression_inds(np.ascontiguousarray(d[:n1]), np.ascontiguousarray(pts[:n1]), 1, 1, 0, np.float32(model.thresholds.nms))
    t_nms1 = time.time() - t0
    ref.set_threads(threads)
    same = bool(np.array_equal(keep.astype(bool), keep_gpu))
    return dict(value=round(x.size / tot / 1e6, 4), unit="Mpix/s", cores=threads, kind="reference",
                stages_s={"network(torch-CPU, TF-CPU stand-in)": round(t_net, 3), "select_sort": round(t_sel, 3), "nms(compiled reference)": round(t_nms, 3),
                          "raster(numpy restatement of the reference's Python loop)": round(t_ras, 3)}, sample_elements=int(x.size),
                parity_checked=same, parity="keep flags of the compiled reference NMS %s the HIP NMS on the %d candidates of the GPU path (%d survivors)"
                                            % ("==" if same else "DIFFER FROM (%d flags)" % int((keep.astype(bool) != keep_gpu).sum()), len(d), int(keep.sum())),
                nms_only={"threads_%d" % threads: {"candidates": int(len(d)), "seconds": round(t_nms, 3), "cand_per_s": round(len(d) / t_nms)},
                          "threads_1": {"candidates": int(n1), "seconds": round(t_nms1, 3), "cand_per_s": round(n1 / t_nms1), "survivors": int(k1.sum())}},
                sample="%dx%d crop of the bench image: torch-CPU U-Net %.2fs (TF-CPU stand-in; its own %d candidates thresholded + sorted in %.2fs) + "
                       "compiled reference NMS (oracle/_ref) on the GPU path's %d candidates -> %d: %.2fs + numpy restatement of the Python "
                       "rasteriser loop %.2fs" % (sample, sample, t_net, len(d_c), t_sel, len(d), int(keep.sum()), t_nms, t_ras))


def cpu_baseline_3d(vol_np, model, sample, threads):
    import copy
    import torch
    from oracle import port, ref
    from stardist_amd.rays3d import rays_from_json
    torch.set_num_threads(threads)
    m3 = ref.stardist3d(); ref.set_threads(threads)
    rays = rays_from_json(model.config.rays_json)
    x = np.ascontiguousarray(vol_np[:sample, :sample, :sample])
    net_cpu = copy.deepcopy(model.net).to("cpu").float()
    t0 = time.time()
    with torch.no_grad():
        prob, dist = net_cpu(torch.from_numpy(x)[None, None])
    prob = prob[0, 0].numpy()
    dist = np.maximum(1e-3, np.moveaxis(dist[0].numpy(), 0, -1))
    t_net = time.time() - t0
    t0 = time.time()
    mask = port.ind_prob_thresh(prob, model.thresholds.prob, b=2)
    n_c = int(mask.sum())
    ind = np.argsort(prob[mask])[::-1]
    d_c = dist[mask][ind]
    t_sel = time.time() - t0
    del d_c, dist
    # the reference natives on the candidates of the GPU path's network (see cpu_baseline_2d): keep flags compared one by one
    d, s, pts, keep_gpu = _gpu_candidates(model, x)
    V, F = rays.vertices, rays.faces.astype(np.int32)
    t0 = time.time()
    keep = m3.c_non_max_suppression_inds(d, pts, V, F, s, 1, 1, 0, np.float32(model.thresholds.nms))
    t_nms = time.time() - t0
    t0 = time.time()
    m3.c_polyhedron_to_label(d[keep], pts[keep], V, F, np.arange(1, keep.sum() + 1, dtype=np.int32), 0, 0, 0, 0, x.shape)
    t_ras = time.time() - t0
    tot = t_net + t_sel + t_nms + t_ras
    full = sample >= vol_np.shape[0]
    same = bool(np.array_equal(keep.astype(bool), keep_gpu))
    return dict(value=round(x.size / tot / 1e6, 4), unit="Mvox/s", cores=threads, kind="reference", seconds=round(tot, 2), same_size=bool(full),
                stages_s={"network(torch-CPU, TF-CPU stand-in)": round(t_net, 3), "select_sort": round(t_sel, 3), "nms(compiled reference incl. Qhull)": round(t_nms, 3),
                          "raster(compiled reference)": round(t_ras, 3)}, sample_elements=int(x.size),
                parity_checked=same, parity="keep flags of the compiled reference 3D NMS %s the HIP NMS on the %d candidates of the GPU path (%d survivors)"
                                            % ("==" if same else "DIFFER FROM (%d flags)" % int((keep.astype(bool) != keep_gpu).sum()), len(d), int(keep.sum())),
                note=("the whole bench volume: a same-size comparison with value_3d" if full else
                      "measured on a crop (the full-size reference run was predicted to exceed --cpu-budget3d): any GPU/CPU ratio formed with "
                      "value_3d is an extrapolation from this crop, not a same-size comparison"),
                sample="%s of the bench volume: torch-CPU U-Net %.2fs (TF-CPU stand-in; its own %d candidates thresholded + sorted in %.2fs) + compiled "
                       "reference 3D NMS (oracle/_ref incl. Qhull) on the GPU path's %d candidates -> %d: %.2fs + compiled reference rasteriser %.2fs"
                       % ("all %d^3 voxels" % sample if full else "%d^3 crop" % sample, t_net, n_c, t_sel, len(d), int(keep.sum()), t_nms, t_ras))


def net_macs(model, macs):
    """multiply-accumulates per input pixel the timed network region executes: with the sparse head (models/unet.py) the distance head
    is evaluated on the candidate pixels only, outside that region -- its dense MACs are not counted"""
    if getattr(model, "_head_mode", "dense") not in ("sparse", "sparse_lazy"):
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
    # the native post-processing calls, timed the same way (events on the stream the C-ABI entry points are handed)
    nat_pairs = {"nms": [], "raster": []}
    orig_dcall = _native.dcall
    NAT = {"sd_nms2d_device": "nms", "sd_nms3d_device": "nms", "sd_polygons_to_label_device": "raster", "sd_polygons_to_label_window_device": "raster",
           "sd_polyhedron_to_label_device": "raster", "sd_polyhedron_to_label_window_device": "raster"}

    def timed_dcall(t, name, *a):
        k = NAT.get(name)
        if k is None:
            return orig_dcall(t, name, *a)
        with torch.cuda.device(t.device):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r = orig_dcall(t, name, *a); e1.record()
        nat_pairs[k].append((e0, e1))
        return r
    _native.dcall = timed_dcall
    for _ in range(warmup):
        res = model.predict_instances(img)
    pairs.clear()
    for v in nat_pairs.values():
        v.clear()
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
    _native.dcall = orig_dcall
    for k, v in nat_pairs.items():                 # ms per step of the whole native call (every kernel, copy and host read-back inside it)
        stats["call_ms_" + k] = sum(a.elapsed_time(b) for a, b in v) / max(1, steps)
    if world > 1:
        tt = torch.tensor([elapsed], device=img.device if dist_.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist_.all_reduce(tt, op=dist_.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed, net_ms, res, stats


def run_host_legs(model, x_np, steps):
    """SURVEY.md 8d's definition of the metric -- host array in, (labels, dict) out: `steps` predict_instances on the host array, (a)
    through predict_instances_iter (models/base.py: the upload of input k + 1 overlaps step k -- helper thread, page-locked staging,
    copy stream), (b) as a plain loop of predict_instances(host array) calls (each step waits for its own upload first)."""
    import torch
    for _ in model.predict_instances_iter([x_np, x_np]):        # warm-up of the staging path
        pass
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in model.predict_instances_iter(x_np for _ in range(steps)):
        pass
    torch.cuda.synchronize(); piped = time.perf_counter() - t0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        model.predict_instances(x_np)
    torch.cuda.synchronize(); serial = time.perf_counter() - t0
    return piped, serial


def host_leg_dict(n, steps, piped, serial, unit, ms_device):
    return {"value": round(n * steps / piped / 1e6, 3), "unit": unit, "ms_per_step": round(1e3 * piped / steps, 3),
            "gap_to_device_resident_input": round(1e3 * piped / steps / ms_device - 1.0, 4),
            "serial_loop": {"value": round(n * steps / serial / 1e6, 3), "ms_per_step": round(1e3 * serial / steps, 3)},
            "note": "SURVEY.md 8d's definition: the same steps with the input handed over as a HOST numpy array, (labels, dict) back on the host; "
                    "predict_instances_iter overlaps the upload of input k + 1 with step k (`serial_loop`: plain predict_instances(host array) "
                    "calls, upload in front of every step); this rank only"}


def run_mode_leg(model, img, steps, warmup, world, dist_, mode):
    """The same steps with another form of the 3x3 convolutions (models/unet.py conv_mode): 'hand' = the exact-f32 MFMA kernel
    (csrc/conv3x3.hip, one f32 fma chain per output), 'bf16x6' = six bf16 products per f32 product (csrc/conv3x3_bf16.hip, the round-3
    default and the range fallback of the default).  All forms are held to the same 1e-5 bar against float64 by
    tests/test_gpu_unet_parity.py and tests/test_gpu_conv3x3.py.  Reported NEXT TO `value`.
    Returns a dict, or an error note (this leg must never take the bench down)."""
    from stardist_amd.models.unet import force_conv_mode
    graphs = model.__dict__.pop("_graphs", None)
    try:
        with force_conv_mode(mode):
            elapsed, net_ms, res, _ = run_leg(model, img, steps, warmup, world, dist_)
        n = int(np.prod(img.shape))
        return {"value": round(world * n * steps / elapsed / 1e6, 3), "ms_per_step": round(1e3 * elapsed / steps, 3), "unet_forward_ms": round(net_ms, 3),
                "instances": len(res[1]["prob"]), "steps": steps,
                "arithmetic": {"hand": "exact f32 MFMA (v_mfma_f32_32x32x2_f32), one fma chain per output (STARDIST_AMD_CONV=hand)",
                               "bf16x6": "six bf16 x bf16 MFMA products per f32 product, f32 accumulation (STARDIST_AMD_CONV=bf16x6)"}[mode]}
    except Exception as e:                       # pragma: no cover
        return {"value": None, "error": repr(e)[:200]}
    finally:
        model.__dict__.pop("_graphs", None)
        if graphs is not None:
            model._graphs = graphs


def stage_ratios(cpu, gpu_stages, n_elements):
    """per-stage time ratios reference-CPU / this path, each scaled to the same number of pixels (the CPU sample may be a crop): the whole-step
    ratio mixes a flagged network stand-in, the compiled reference natives and (2D) a numpy restatement of the reference's Python
    rasteriser loop -- quote the stages, not the quotient of the totals.  A reported baseline, not a target."""
    try:
        st, ne = cpu["stages_s"], float(cpu["sample_elements"])
        per = lambda sec: sec / ne * n_elements * 1e3                      # CPU ms for the GPU step's number of elements
        g = {"network": gpu_stages["unet_forward"], "nms": gpu_stages["nms"], "raster": gpu_stages["raster"]}
        c = {"network": per([v for k, v in st.items() if k.startswith("network")][0]), "nms": per([v for k, v in st.items() if k.startswith("nms")][0]),
             "raster": per([v for k, v in st.items() if k.startswith("raster")][0])}
        return {k: {"cpu_ms": round(c[k], 1), "gpu_ms": round(g[k], 3), "ratio": round(c[k] / g[k], 1) if g[k] > 0 else None} for k in g}
    except Exception as e:       # pragma: no cover
        return {"error": repr(e)[:120]}


def predicted_scaling(per_block, t_exchange, t_final_nms, t_raster_local, s_pass_n1):
    """From the N = 1 per-block times: the per-rank critical path max_r sum_{b % N == r} (t_predict + t_local_nms) for N = 2, 4, 8 with the
    round-robin deal of predict_instances_sharded, plus the serial tail on rank 0 (exchange + cross-tile NMS; measured at N = 1, the
    gather over xGMI is not modelled) and the owner-side rasters (the N = 1 total / N)."""
    out = {}
    for n in (2, 4, 8):
        loads = [sum(tp + tn for bi, tp, tn in per_block if bi % n == r) for r in range(n)]
        t = max(loads) + t_exchange + t_final_nms + t_raster_local / n
        out[str(n)] = {"critical_path_s": round(max(loads), 4), "pass_s": round(t, 4), "serial_tail_share": round((t_exchange + t_final_nms) / t, 4),
                       "strong_scaling_efficiency": round(s_pass_n1 / (n * t), 4)}
    return out


class TiledSource(object):
    """the large synthetic input of the sharded legs as a SOURCE (shape + slicing, like a memmap): the base tile repeated `rep` times along
    every axis, materialised only where it is sliced -- so that at N > 1 a rank uploads the read regions of ITS blocks (stardist_amd.big
    ShardedInput) and never holds the whole 16384^2 / 1024^3 array, neither on the host nor in HBM"""

    def __init__(self, base, rep):
        self.base, self.rep = np.ascontiguousarray(base), int(rep)
        self.shape = tuple(int(v) * self.rep for v in self.base.shape)
        self.ndim, self.dtype = self.base.ndim, self.base.dtype
        self.nbytes = int(np.prod(self.shape)) * self.base.dtype.itemsize

    def __getitem__(self, slices):
        idx = [np.arange(*s.indices(n)) % b for s, n, b in zip(slices, self.shape, self.base.shape)]
        return self.base[np.ix_(*idx)]


def sharded_input(model, base_np, rep, axes, block, overlap, context, rank, world, dev):
    """N = 1: the whole input resident in HBM (the bench contract: inputs resident when the timed region starts); N > 1: this rank's blocks
    only (ShardedInput over the lazily tiled source), resident as well"""
    import torch
    if world == 1:
        return torch.from_numpy(base_np).to(dev).repeat(*([rep] * base_np.ndim))
    from stardist_amd.big import ShardedInput
    src = TiledSource(base_np, rep)
    return ShardedInput.for_rank(model, src, axes, min(block, src.shape[0]), overlap, context, rank=rank, world=world, device=dev)


def run_sharded_leg(model, big, axes, block, overlap, context, passes, world, dist_, rank, warm_passes=1):
    """BASELINE.json configs 4/5: ONE large input, its blocks dealt round-robin to the ranks; per block network + selection + local NMS on
    the device; one gather of the surviving records to rank 0; cross-tile NMS over the band survivors only; final instances broadcast and
    every rank renders the write regions of its blocks (stardist_amd/big.py, design A of SURVEY.md 8e).  Strong scaling: the input is
    the same for every N.  N = 1: the label image comes back as ONE host array (as predict_instances returns it); N > 1: the tiles stay on
    the ranks that rendered them (labels_out="local": owner-side window rasters, no whole-volume raster and no label gather on rank 0),
    as the reference's block.write leaves them in the shared output.  At N = 1 one extra pass in that owner-side form is timed as well
    and, from the per-block times, the per-rank critical path and strong-scaling efficiency at N = 2, 4, 8 are predicted.
    Returns a dict (rank 0) or None."""
    import torch
    # warm-up: one block's worth of the input (HIP graph of the block shape, weight packing, arena growth), then ONE untimed pass over the
    # whole input -- the label image of a pass is returned in page-locked memory (stardist_amd/utils.py to_host), and allocating 1 - 4 GiB of it
    # costs ~0.1 s the first time; a result is dropped before the next pass starts (as a caller working through slides would), so the timed
    # passes recycle the block
    warm = big[tuple(slice(0, block) for _ in range(len(big.shape)))]
    model.predict_instances_sharded(warm, axes, block_size=block, min_overlap=overlap, context=context, distributed=False)
    del warm
    kw = dict(block_size=block, min_overlap=overlap, context=context, broadcast_result=False)
    if world > 1:
        kw["labels_out"] = "local"
    for _ in range(max(1, warm_passes)):
        labels, res = model.predict_instances_sharded(big, axes, **kw)
        del labels, res
    if world > 1:
        dist_.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc = {}
    for _ in range(passes):
        labels, res = model.predict_instances_sharded(big, axes, **kw)
        for k, v in model._last_sharded_stats.items():
            acc[k] = (acc.get(k, 0) + v) if not isinstance(v, list) else v
        if _ + 1 < passes:
            del labels, res
    torch.cuda.synchronize()
    if world > 1:
        dist_.barrier()
    elapsed = time.perf_counter() - t0
    del labels
    mean = {k: (v if isinstance(v, list) else (round(v / passes, 4) if isinstance(v, float) else v // passes)) for k, v in acc.items()}
    mean["input_bytes_resident"] = int(getattr(big, "bytes_held", 0) or int(np.prod(big.shape)) * 4)
    mean["peak_device_bytes"] = int(torch.cuda.max_memory_allocated())
    per_rank = [None] * world
    if world > 1:
        tt = torch.tensor([elapsed], device=big.device if dist_.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist_.all_reduce(tt, op=dist_.ReduceOp.MAX)
        elapsed = float(tt.item())
        dist_.all_gather_object(per_rank, mean)
    else:
        per_rank = [mean]
    local_form = None
    if world == 1:                         # the owner-side form of N > 1, once, on one GPU: its raster time feeds the prediction
        torch.cuda.synchronize(); t1 = time.perf_counter()
        model.predict_instances_sharded(big, axes, labels_out="local", **kw)
        torch.cuda.synchronize()
        local_form = {"s_per_pass": round(time.perf_counter() - t1, 4), "t_raster": round(model._last_sharded_stats["t_raster"], 4),
                      "t_final": round(model._last_sharded_stats["t_final"], 4)}
    if rank != 0:
        return None
    n = int(np.prod(big.shape))
    r0 = per_rank[0]
    s_pass = elapsed / passes
    out = {"value": round(n * passes / elapsed / 1e6, 3), "s_per_pass": round(s_pass, 4), "passes": passes, "warm_passes": max(1, warm_passes), "scaling": "strong",
           "input_shape": list(big.shape), "ranks": world, "backend": (dist_.get_backend() if world > 1 else None),
           "input_form": "whole array resident in HBM" if world == 1 else "every rank holds the read regions of its own blocks only (ShardedInput)",
           "block_size": block, "min_overlap": overlap, "context": context, "blocks": sum(p["blocks"] for p in per_rank),
           "redundancy": round(sum(p["blocks"] for p in per_rank) * float(block) ** len(big.shape) / n, 3),
           "instances": r0["instances"], "candidates": sum(p["candidates"] for p in per_rank),
           "gathered_survivors": r0["gathered"], "gathered_bytes": r0["gathered_bytes"], "exact_record_bytes": r0.get("exact_record_bytes", 0),
           "band_survivors": r0["band"], "interior_survivors": r0["interior"],
           "t_phase1": max(p["t_phase1"] for p in per_rank),
           "t_predict": max(p["t_predict"] for p in per_rank), "t_local_nms": max(p["t_local_nms"] for p in per_rank), "t_exchange": r0["t_exchange"],
           "t_final": r0["t_final"], "t_final_nms": r0["t_final_nms"], "t_raster": max(p["t_raster"] for p in per_rank),
           "t_final_frac": round(r0["t_final"] / s_pass, 4),
           "labels": "one host array on rank 0" if world == 1 else "rank-local tiles of the owned write regions (owner-side window rasters, not gathered)",
           "per_rank": [{k: v for k, v in p.items() if k != "per_block"} for p in per_rank]}
    if world == 1:
        out["owner_side_form"] = local_form
        # the N = 1 pass in the owner-side form is the numerator: the same work, the same output form as N > 1
        out["predicted_scaling"] = predicted_scaling(r0["per_block"], r0["t_exchange"], r0["t_final_nms"], local_form["t_raster"], local_form["s_per_pass"])
        out["predicted_scaling"]["basis"] = ("per-block (t_predict + t_local_nms) of this N = 1 run dealt round-robin, + exchange + cross-tile NMS "
                                             "on rank 0 (serial tail) + owner-side rasters / N; numerator = the N = 1 pass in the owner-side form")
    return out


class _DryModel(object):
    """Stand-in of `--dry-collectives` (no GPU, no kernels, no network): the 'image' holds the score of a disc of radius 6 at every
    point of a 24-pixel lattice, every disc is a candidate and a survivor (they are 12 pixels apart, nothing suppresses anything).
    What the mode checks is the EXCHANGE of predict_instances_sharded -- how many bytes each rank puts on its link to rank 0 and that
    every object comes out exactly once although the blocks' write regions overlap -- not any result of the product's kernels."""
    n_rays = 32

    def __init__(self):
        from stardist_amd.models.config import Config2D
        self.config = Config2D(n_rays=self.n_rays, n_channel_in=1)
        self.device = "cpu"

    def _axes_div_by(self, axes): return tuple(1 for a in axes)

    def _axes_tile_overlap(self, axes): return tuple(0 for a in axes)

    def predict_sparse(self, x, axes=None, prob_thresh=None, **kw):
        pts = np.argwhere(x > 0)
        return x[x > 0].astype(np.float32), np.full((len(pts), self.n_rays), 6.0, np.float32), pts

    def _nms_sparse(self, dist, prob, points, nms_thresh=None, **kw):
        return np.argsort(prob, kind="stable")[::-1].copy()

    def _instances_from_survivors(self, shape, p, pr, d, return_labels=True, window=None, **kw):
        return None, dict(points=p, prob=pr)


def dry_collectives(args):
    """`bench.py --gpus N --dry-collectives` (under torch.distributed.run, backend gloo, CPU only): the collectives of the block-sharded
    path on a synthetic survivor set, with the byte count of every rank asserted -- the N > 1 exchange rehearsed where no N-GPU node is
    at hand.  Prints one JSON line on rank 0 (not a bench line: no metric)."""
    import torch
    import torch.distributed as dist_
    from stardist_amd.big import predict_instances_sharded
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if world > 1:
        dist_.init_process_group("gloo", rank=rank, world_size=world)
    size, block, overlap, context = 2048, 576, 64, 32
    img = np.zeros((size, size), np.float32)
    g = np.arange(12, size - 12, 24)
    rs = np.random.RandomState(0)
    img[np.ix_(g, g)] = rs.uniform(0.5, 1.0, (len(g), len(g))).astype(np.float32)
    m = _DryModel()
    _, res = predict_instances_sharded(m, img, "YX", block, overlap, context=context, return_labels=False, broadcast_result=True)
    st = m._last_sharded_stats
    W = m.n_rays + 1 + 2 + 1
    mine = torch.tensor([st.get("sent_bytes", 0), st["local_survivors"], st["blocks"]], dtype=torch.int64)
    every = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist_.all_gather(every, mine)
    else:
        every = [mine]
    sent = [int(t[0]) for t in every]
    kept = [int(t[1]) for t in every]
    n_obj = len(g) ** 2
    assert len(res["points"]) == n_obj, (len(res["points"]), n_obj)                       # every object exactly once, on every rank
    for r in range(world):
        assert sent[r] == (0 if r == 0 else kept[r] * W * 4), (r, sent[r], kept[r])       # exact sizes: records x record bytes, nothing padded
    if rank == 0:
        out = dict(dry_collectives=True, backend="gloo", world=world, blocks_per_rank=[int(t[2]) for t in every],
                   record_bytes=W * 4, objects=n_obj, records_per_rank=kept, sent_bytes_per_rank=sent)
        if world > 1:
            counts = st["rank_counts"]
            assert [a + b for a, b in counts] == kept, (counts, kept)
            assert st["gathered_bytes"] == sum(sent) and st["exact_record_bytes"] == sum(kept) * W * 4
            out.update(gathered_bytes=st["gathered_bytes"], exact_record_bytes=st["exact_record_bytes"],
                       padded_gather_would_move=world * max(kept) * W * 4, interior=st["interior"], band=st["band"], unique=st["unique"],
                       duplicates_dropped=sum(kept) - st["unique"],
                       collectives=["all_reduce(MAX) of 1 float64", "all_gather of 2 int64 per rank", "batch_isend_irecv: one message per rank with records, exact size",
                                    "broadcast of the instance count", "broadcast of the final records"])
            assert st["unique"] == n_obj and sum(kept) > n_obj                                # the overlap bands did report objects twice
        out["ok"] = True
        print(json.dumps(out), flush=True)
    if world > 1:
        dist_.barrier()
        dist_.destroy_process_group()


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
    ap.add_argument("--skip-3d", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=2048)
    ap.add_argument("--cpu-sample3d", type=int, default=128)
    ap.add_argument("--cpu-budget3d", type=float, default=75.0, help="run the 3D CPU baseline at full size if the crop predicts at most this many seconds")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-sharded", action="store_true", help="skip the block-sharded big-input legs (configs 4/5; run by default at every N)")
    ap.add_argument("--sharded-size", type=int, default=16384)
    ap.add_argument("--sharded-size3d", type=int, default=1024)
    # block sizes (read size incl. context, as in the reference's BlockND.cover), chosen so that the block count divides by 8:
    # 4416 -> 16 blocks of the 16384^2 slide (1.16x the slide's pixels, the smallest 4 x 4 cover); 560 -> 8 blocks of the 1024^3 volume
    # (1.31x its voxels, the smallest 2 x 2 x 2 cover; the 128-channel features of one block are 90 GB of the 288 GB) -- round 3 used 27
    # blocks of 416^3 (1.81x, 4/3/3/.. blocks per rank at N = 8).  Should the 560^3 block not fit, the leg falls back to those and says so.
    ap.add_argument("--sharded-block", type=int, default=4416)
    ap.add_argument("--sharded-block3d", type=int, default=560)
    ap.add_argument("--sharded-block3d-fallback", type=int, default=416)
    ap.add_argument("--skip-sharded-3d", action="store_true")
    ap.add_argument("--no-split-leg", "--no-exact-leg", dest="no_split_leg", action="store_true",
                    help="skip the extra legs with the exact-f32 and the six-product bf16 convolution kernels")
    ap.add_argument("--dry-collectives", action="store_true",
                    help="CPU / gloo rehearsal of the sharded path's exchange with asserted byte counts per rank (no GPU, no bench line)")
    args = ap.parse_args()
    if args.dry_collectives:
        return dry_collectives(args)

    # dmabuf IPC (the pool's host driver supports no legacy IPC handles): RCCL's buffer registration between the ranks of one node fails
    # with `hipIpcGetMemHandle: invalid argument` without it; exported by the image, set here as well in case the launcher's env was trimmed
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist_
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # plumbing check on a one-GPU box: STARDIST_AMD_BENCH_BACKEND=gloo runs the N ranks on ONE device with gloo collectives (every
        # code path of the N > 1 legs except the RCCL transport; the numbers of such a run mean nothing)
        backend = os.environ.get("STARDIST_AMD_BENCH_BACKEND", "nccl")
        if backend != "nccl":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dist_.init_process_group(backend, rank=rank, world_size=world)   # "nccl" IS RCCL on ROCm
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)
    threads = args.cpu_threads or min(os.cpu_count() or 1, 32)

    from oracle import synth                       # input generators only (numpy), shared with the tests
    from stardist_amd.models import Config2D, Config3D, StarDist2D, StarDist3D
    from stardist_amd.models.unet import conv_macs_per_input_pixel
    from stardist_amd.models.unet import conv_mode
    mode = conv_mode()
    mult, cpeak, pipe = CONV_FORMS[mode]

    # ------------------------------------------------------------------ 2D leg (headline)
    H = W = args.size
    img_np = synth.s2d_nuclei_image(H, W, seed=rank)
    img = torch.from_numpy(img_np).to(dev)
    model = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
    # every rank calibrates on the SAME image (seed 0), so that all ranks run identical weights -- the sharded legs deal the blocks of one
    # input over the ranks and must see one model; the timed tile of a rank is its own (seed = rank)
    calibrate_heads(model, img if rank == 0 else torch.from_numpy(synth.s2d_nuclei_image(H, W, seed=0)).to(dev))
    macs = conv_macs_per_input_pixel(model.net, model.config)
    elapsed, net_ms, res, st = run_leg(model, img, args.steps, args.warmup, world, dist_)
    # second number (SURVEY.md 8d defines the metric host-array-in): the same steps with the image handed over as a host numpy array,
    # i.e. including the 16.8 MB H2D copy (staged through page-locked memory, stardist_amd/utils.py to_device); `value` stays the
    # HBM-resident figure the bench contract asks for
    host2d, host2d_err = guarded(lambda: run_host_legs(model, img_np, args.steps), "host-input leg (2D)")      # an extra leg must never lose the line
    # the bit-exact-by-construction mode (sd_set_option("nms2d_strict", 1): every pair through the Clipper-exact sweep, no decision from the
    # area enclosure): same steps, and the instances must be the very same
    from stardist_amd.lib import _native as _nat
    strict_steps = max(1, min(args.steps, 10))

    def strict_run():
        with _nat.option("nms2d_strict", 1):
            return run_leg(model, img, strict_steps, 1, world, dist_)
    strict, strict_err = guarded(strict_run, "nms2d_strict leg")
    if strict is None:
        strict_leg = {"value": None, "error": strict_err}
    else:
        el_s, _, res_s, _ = strict
        strict_leg = {"value": round(world * H * W * strict_steps / el_s / 1e6, 3), "unit": "Mpix/s", "ms_per_step": round(1e3 * el_s / strict_steps, 3),
                      "instances": len(res_s[1]["prob"]),
                      "same_result_as_default": bool(np.array_equal(res_s[0], res[0]) and np.array_equal(res_s[1]["points"], res[1]["points"])),
                      "note": "sd_set_option('nms2d_strict', 1): every 2D pair decided by the Clipper-exact sweep (bit-exact by construction); the default decides "
                              "pairs far from the threshold from an adversarially validated band around the exact area (DESIGN.md 3.4)"}
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
        nms_ms, ras_ms = float(st.get("call_ms_nms", 0.0)), float(st.get("call_ms_raster", 0.0))
        stages = {"unet_forward": round(net_ms, 3),
                  "nms": round(nms_ms, 3),
                  "nms_parts": {"build_grid_neighbour_lists": round(float(s2[7] / 1e6), 3), "pair_stage(decide+bucket+sweep)": round(float(pair_ms), 3),
                                "general_path(joins)": round(float(s2[6] / 1e6), 3),
                                "rounds_bookkeeping_and_readbacks": round(nms_ms - float(pair_ms + s2[6] / 1e6 + s2[7] / 1e6), 3)},
                  "raster": round(ras_ms, 3),
                  "select_sort_head_rows_results_to_host_glue": round(ms_per_step - net_ms - nms_ms - ras_ms, 3),
                  "note": "nms / raster = HIP events around the whole native call (sd_nms2d_device, sd_polygons_to_label_device) on the caller's stream; "
                          "nms_parts = the library's own events (the general path overlaps the tail batch, so the parts need not add up exactly)"}
        # Roofline of the convolutions, SURVEY.md 8(d): ALGORITHMIC flops (2 x MACs of the instantiated module) / HIP-event time of the forward
        # pass / dense peak of the pipe the kernel runs on.  A split form executes `mult` matrix products per algorithmic one (f16x3: 3 on
        # the fp16 pipe, bf16x6: 6 on the bf16 pipe, both 2.5 PFLOP/s dense; exact: 1 on the f32 pipe, 157.3 TFLOP/s): `executed_frac` is
        # the pipe occupancy, `frac` the roofline fraction.
        conv_kernel = {"f16x3": "network forward = one HIP graph: k_conv3_f16 (hand-written split-fp16 implicit GEMM, csrc/conv3x3_f16.hip: every 3x3 layer incl. "
                                "folded up-sampling / concatenation / bias / ReLU; v_mfma_f32_32x32x16_f16, three products per f32 product in two f32 "
                                "accumulators, two workgroups per CU) + k_conv3_c1x32 + k_maxpool_cl4 + probability-head pass",
                       "bf16x6": "network forward = one HIP graph: k_conv3_bf16 (csrc/conv3x3_bf16.hip, six bf16 products per f32 product) + k_conv3_c1x32 + "
                                 "k_maxpool_cl4 + probability-head pass",
                       "hand": "network forward = one HIP graph: k_conv3<1> (hand-written exact-f32 MFMA implicit GEMM, csrc/conv3x3.hip) + k_conv3_c1x32 + "
                               "k_maxpool_cl4 + probability-head pass"}[mode]
        roof_conv = {"bound": "mfma", "pipe": pipe, "kernel": conv_kernel, "achieved": round(conv_tf, 3), "peak": cpeak,
                     "unit": "TFLOP/s", "frac": round(conv_tf / cpeak, 4), "traffic": None, "flops_per_launch": flops,
                     "executed_products_per_mac": mult, "executed_tflops": round(conv_tf * mult, 3), "executed_frac": round(conv_tf * mult / cpeak, 4),
                     "avg_ms": round(net_ms, 3),
                     "note": "achieved = algorithmic FLOPs of the convolutions (2 x MACs, recomputed from the instantiated module) / HIP-event time of the "
                             "whole forward pass (14 conv launches + pools + head) on the caller's stream; executed_* = x%d matrix products per MAC; "
                             "for reference the f32-MFMA peak is %.1f TFLOP/s; per-kernel durations and the MFMA-busy counter: profiles/r06_*; the kernel is POWER-limited "
                             "(same instruction stream on zero data: 31-37 %% faster at a 45 %% higher clock; the matrix pipe alone sustains 1.4-1.6 PFLOP/s on "
                             "realistic operands: profiles/r06_conv_power_*.txt, r06_mfma_power_roof.txt, DESIGN.md 3f)"
                             % (int(mult), MFMA_F32_PEAK_TFLOPS)}
        ct = conv_traffic("2d", mode, H, 2048)
        if ct:
            roof_conv["traffic"] = ct["bytes_per_forward"]
            roof_conv["traffic_source"] = ct["source"]
            roof_conv["kernel_ms_per_forward_rocprof"] = ct["kernel_ms_per_forward"]
        n_decided = float(s2[9])
        roof_pair = {"bound": "hbm", "kernel": "pair stage of a greedy round: k_pairs_decide (area enclosure from a boundary integral, 32 lanes per pair, csrc/area_bounds.h) "
                                               "+ pair bucketing + k_pairs_beam<32,8,6,4,64,REL16> (Clipper-exact scan-beam sweep, one pair per lane, state in LDS) for the pairs "
                                               "the enclosure leaves undecided", "achieved": round(pair_gbs, 3),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(pair_gbs / HBM_PEAK_GBS, 6), "traffic": None,
                     "bytes_per_launch": round(pair_bytes_per_launch), "avg_launch_ms": round(float(pair_ms / pair_launches), 4),
                     "launches_per_step": float(pair_launches), "pairs_per_step": float(n_pairs),
                     "pairs_decided_by_area_enclosure": n_decided, "pairs_swept_exactly": float(n_pairs) - n_decided,
                     "note": "not HBM-bound.  The decision kernel is regular VALU work (~1000 instructions per pair, n^2 edge pairs); the sweep is an integer "
                             "state machine, one pair per lane, per-pair state (420 B) lane-interleaved in LDS: LATENCY-bound under lane divergence, a launch "
                             "costs at least one sweep's serial latency (~0.7 ms) however few pairs it holds; algorithmic bytes = 272 B/pair (SURVEY.md 8d)"}
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
            "dtype": "f32", "data": "synthetic",
            "arithmetic": ("float32 data, float32 accumulation; 3x3 convolution products " +
                           {"f16x3": "as three fp16 x fp16 terms of operands split into two fp16 parts (hi + lo * 2^-11)",
                            "bf16x6": "as six bf16 x bf16 terms of operands split into three bf16 parts", "hand": "exact in f32"}[mode] +
                           " (f32-accurate: layers and networks within 3e-6 of a float64 evaluation, tests/test_gpu_conv3x3.py, "
                           "test_gpu_unet_parity.py; the bar on distances is RELATIVE to max(1, |dist|): the absolute error is ~1.2e-5 px at 12 px, the same "
                           "as the float32 CPU evaluation of the module, profiles/r06_unet_parity_vs_float64.txt; activations travel between the layers as "
                           "the two fp16 terms the consumer multiplies with (split16, made by the producer: bit-identical); `exact_f32` / `split_bf16x6` = "
                           "the same step with the other kernel forms, `exact_f32` being the figure for a reader who takes 'f32' literally)"
                           "; NMS / rasteriser in the reference's own int64 / float32 / float64 arithmetic"),
            "config": {"workload": "StarDist2D 32-ray U-Net (depth 3, 32 base filters), %dx%d synthetic fluo tile per GPU, predict_instances "
                                   "(U-Net + select + 2D NMS + polygon raster), seeded random weights, heads calibrated to ~10%% candidates "
                                   "radius 10+-10%%" % (H, W),
                       "candidates": n_cand, "survivors": len(res[1]["prob"]), "prob_thresh": model.thresholds.prob,
                       "nms_thresh": model.thresholds.nms, "parallelism": "tiles-per-gpu x%d" % world,
                       "nms2d_mode": "default: pairs far from the threshold decided from the area enclosure (exact intersection area +- a band for Clipper's "
                                     "lattice rounding: 0.5 per crossing, max(0.15 per near edge pair, 0.45 per strip), in units of lmax_P + lmax_Q; validated "
                                     "adversarially against the vendored Clipper, DESIGN.md 3.4), the rest by the Clipper-exact sweep; `nms2d_strict` = every "
                                     "pair swept (bit-exact by construction), same instances"},
            "stages_ms": stages, "roofline": dominant, "roofline_convs": roof_conv, "roofline_pair_kernel": roof_pair,
            "nms2d_strict": strict_leg,
            "value_host_input": (host_leg_dict(H * W, args.steps, host2d[0], host2d[1], "Mpix/s", ms_per_step) if host2d is not None
                                 else {"value": None, "error": host2d_err}),
        }
        if not args.no_cpu_baseline and world == 1:      # reported baseline: rank 0 at N=1 only
            try:
                out["cpu_baseline"] = cpu_baseline_2d(img_np, model, min(args.cpu_sample, H), threads)
                out["cpu_baseline"]["stage_ratios"] = stage_ratios(out["cpu_baseline"], stages, H * W)
            except Exception as e:   # oracle/_ref must have travelled with the tree
                out["cpu_baseline"] = {"value": None, "unit": "Mpix/s", "cores": 0, "kind": "reference", "sample": "unavailable: %r" % (e,)}
    if not args.no_split_leg:
        for key, md in (("exact_f32", "hand"), ("split_bf16x6", "bf16x6")):
            if md == mode:
                continue
            r = run_mode_leg(model, img, max(1, min(args.steps, 10)), 2, world, dist_, md)
            if rank == 0:
                r["unit"] = "Mpix/s"
                out[key] = r
    # ---- config 4: one 16384^2 slide (the 2048^2 synthetic tile repeated), blocks 4480 / overlap 128 / context 128, sharded over the ranks
    if not args.no_sharded:
        rep = max(1, args.sharded_size // H)
        big, big_err = guarded(lambda: sharded_input(model, synth.s2d_nuclei_image(H, W, seed=0), rep, "YX", args.sharded_block, 128, 128, rank, world, dev),
                               "sharded_2d input")
        # N > 1: this leg IS the headline, so it is timed as the contract prescribes -- W untimed passes, then exactly K timed ones
        # (a pass over the slide is one "step"); N = 1: two timed passes next to the tile leg
        sh_passes, sh_warm = (args.steps, max(1, args.warmup)) if world > 1 else (2, 1)
        r, err = (None, big_err) if big is None else guarded(
            lambda: run_sharded_leg(model, big, "YX", min(args.sharded_block, big.shape[0]), 128, 128, sh_passes, world, dist_, rank, warm_passes=sh_warm), "sharded_2d")
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
                out["steps"], out["warmup"] = r["passes"], r["warm_passes"]
                out["config"]["workload"] = ("predict_instances_sharded on ONE %dx%d synthetic slide (BASELINE.json config 4): blocks %d / overlap 128 / "
                                             "context 128 dealt round-robin over the ranks, local NMS per block, gather of the survivors, cross-tile "
                                             "NMS over the band on rank 0, write regions rendered by their owners" % (big.shape[0], big.shape[1], r["block_size"]))
                out["config"]["parallelism"] = "blocks-of-one-slide x%d (strong scaling; `value_tiles` = independent tile per rank)" % world
                out["scaling_base"] = ("the same workload at N = 1 is `sharded_2d.value` of the N = 1 line (one slide, all blocks on one GPU), NOT that "
                                       "line's `value` (the 2048x2048 tile, BASELINE.json configs[1]): strong-scaling efficiency = value / (N x sharded_2d.value at N = 1)")
        del big
    del model, img
    torch.cuda.empty_cache()

    # ------------------------------------------------------------------ 3D leg (a closure run through guarded(): whatever fails in it, the
    # line with the 2D figures -- at N > 1 the headline -- is still printed)
    def legs_3d():
        S = args.size3d
        vol_np = synth.s3d_nuclei_image(S, seed=rank)
        vol = torch.from_numpy(vol_np).to(dev)
        m3 = StarDist3D(Config3D(rays=96), basedir=None, device=dev, seed=0)
        m3.thresholds = dict(prob=0.5, nms=0.3)
        calibrate_heads(m3, vol if rank == 0 else torch.from_numpy(synth.s3d_nuclei_image(S, seed=0)).to(dev),
                        frac=0.009, radius=8.5, noise=0.03)   # near-spherical objects; same image on every rank
        macs3 = conv_macs_per_input_pixel(m3.net, m3.config)
        steps3 = max(1, min(args.steps, 5))
        elapsed3, net3_ms, res3, st3 = run_leg(m3, vol, steps3, 2, world, dist_)
        host3d, host3d_err = guarded(lambda: run_host_legs(m3, vol_np, steps3), "host-input leg (3D)")
        if rank == 0:
            s3 = st3.get("nms3d", np.zeros(16, np.int64)) / steps3
            ms3 = 1e3 * elapsed3 / steps3
            flops3 = 2.0 * net_macs(m3, macs3) * S ** 3
            conv3_tf = flops3 / (net3_ms * 1e-3) / 1e12
            out["value_definition"] = ("`value` / `value_3d`: input resident in HBM when the timed region starts (the bench contract); `value_host_input`: "
                                       "the same step from a host numpy array in to (labels, dict) out, SURVEY.md 8d's definition")
            out["value_3d"] = round(world * S ** 3 * steps3 / elapsed3 / 1e6, 3)
            out["unit_3d"] = "Mvox/s"
            out["value_host_input_3d"] = (host_leg_dict(S ** 3, steps3, host3d[0], host3d[1], "Mvox/s", ms3) if host3d is not None
                                          else {"value": None, "error": host3d_err})
            out["ms_per_step_3d"] = round(ms3, 3)
            out["config_3d"] = {"workload": "StarDist3D Rays_GoldenSpiral(96) U-Net (depth 2), %d^3 synthetic volume per GPU, predict_instances "
                                            "(U-Net + select + 3D NMS cascade + polyhedron raster + relabel)" % S,
                                "survivors": len(res3[1]["prob"]), "steps": steps3, "nms_thresh": 0.3,
                                "cascade_calls": {"upper": float(s3[0]), "lower": float(s3[1]), "kernel_volume": float(s3[2]),
                                                  "hull_volume": float(s3[11]), "render": float(s3[3])},
                                "near_threshold_volume_decisions": float(s3[13])}
            nms3_ms, ras3_ms = float(st3.get("call_ms_nms", 0.0)), float(st3.get("call_ms_raster", 0.0))
            out["stages_ms_3d"] = {"unet_forward": round(net3_ms, 3), "nms": round(nms3_ms, 3),
                                   "nms_parts": {"stage3_kernel_volume": round(float(s3[8] / 1e6), 3), "stage4_hull_volume": round(float(s3[9] / 1e6), 3),
                                                 "stage5_render": round(float(s3[10] / 1e6), 3),
                                                 "broad_phase_rounds_readbacks": round(nms3_ms - float((s3[8] + s3[9] + s3[10]) / 1e6), 3)},
                                   "raster": round(ras3_ms, 3),
                                   "select_sort_head_rows_results_to_host_glue": round(ms3 - net3_ms - nms3_ms - ras3_ms, 3)}
            out["roofline_convs_3d"] = {"bound": "mfma", "pipe": pipe, "kernel": "network forward (3x3x3 layers as three z-plane units per 32-channel chunk; kernel family as `roofline_convs`)",
                                        "achieved": round(conv3_tf, 3), "peak": cpeak, "unit": "TFLOP/s", "frac": round(conv3_tf / cpeak, 4),
                                        "traffic": None, "flops_per_launch": flops3, "executed_products_per_mac": mult,
                                        "executed_tflops": round(conv3_tf * mult, 3), "executed_frac": round(conv3_tf * mult / cpeak, 4), "avg_ms": round(net3_ms, 3)}
            ct3 = conv_traffic("3d", mode, S, 256)
            if ct3:
                out["roofline_convs_3d"]["traffic"] = ct3["bytes_per_forward"]
                out["roofline_convs_3d"]["kernel_ms_per_forward_rocprof"] = ct3["kernel_ms_per_forward"]
            if not args.no_cpu_baseline and world == 1:
                try:
                    cb = cpu_baseline_3d(vol_np, m3, min(args.cpu_sample3d, S), threads)
                    # same-size baseline when the host can afford it: the crop's time scaled by the voxel ratio predicts the full run
                    if args.cpu_sample3d < S and cb.get("seconds", 1e9) * (S / float(min(args.cpu_sample3d, S))) ** 3 <= args.cpu_budget3d:
                        crop = cb
                        cb = cpu_baseline_3d(vol_np, m3, S, threads)
                        cb["crop_run"] = {"sample": crop["sample"], "value": crop["value"]}
                    cb["stage_ratios"] = stage_ratios(cb, out["stages_ms_3d"], S ** 3)
                    out["cpu_baseline_3d"] = cb
                except Exception as e:
                    out["cpu_baseline_3d"] = {"value": None, "unit": "Mvox/s", "cores": 0, "kind": "reference", "sample": "unavailable: %r" % (e,)}
        if not args.no_split_leg:
            for key, md in (("exact_f32_3d", "hand"), ("split_bf16x6_3d", "bf16x6")):
                if md == mode:
                    continue
                r = run_mode_leg(m3, vol, steps3, 1, world, dist_, md)
                if rank == 0:
                    r["unit"] = "Mvox/s"
                    out[key] = r
        # ---- config 5: one 1024^3 volume (the 256^3 synthetic volume repeated), 416^3 blocks / overlap 32 / context 32, sharded
        if not args.no_sharded and not args.skip_sharded_3d:
            rep = max(1, args.sharded_size3d // S)
            base3 = synth.s3d_nuclei_image(S, seed=0)
            def sharded_3d(block):                                 # (the input is built inside the guarded call too)
                bigv = sharded_input(m3, base3, rep, "ZYX", block, 32, 32, rank, world, dev)
                try:
                    return run_sharded_leg(m3, bigv, "ZYX", min(block, bigv.shape[0]), 32, 32, 1, world, dist_, rank)
                finally:
                    del bigv
            r, err = guarded(lambda: sharded_3d(args.sharded_block3d), "sharded_3d")
            if err and args.sharded_block3d_fallback and args.sharded_block3d_fallback < args.sharded_block3d:      # (every rank fails alike: same shapes)
                m3.__dict__.pop("_graphs", None)
                torch.cuda.empty_cache()
                first_err = err
                r, err = guarded(lambda: sharded_3d(args.sharded_block3d_fallback), "sharded_3d (fallback block)")
                if rank == 0 and r is not None:
                    r["fallback_from"] = {"block_size": args.sharded_block3d, "error": first_err[:300]}
            if rank == 0 and err:
                out["sharded_3d"] = {"error": err}
            elif rank == 0:
                r["unit"] = "Mvox/s"
                out["sharded_3d"] = r
    if not args.skip_3d:
        _, err3 = guarded(legs_3d, "3D legs")
        if err3 and rank == 0:
            out["error_3d"] = err3
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist_.destroy_process_group()


if __name__ == "__main__":
    main()
