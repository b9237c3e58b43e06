"""Goldens of the full-size sharded workloads, made ON THE GPU BOX (the candidates are the network's own: they need the device)
with the COMPILED REFERENCE natives (oracle/_ref) as the judge:

  2D  one 16384^2 slide (bench.py's `sharded_2d` input, 32.3 M candidates): c_non_max_suppression_inds of the reference over ALL
      candidates of the whole slide in predict_instances' order -> number of survivors, SHA-256 of their centres, SHA-256 of the label
      image the (separately pinned) polygon rasteriser paints from them.
  3D  one 512^3 volume (8 blocks of 304^3; 1.3 M candidates): the reference's 3D NMS incl. Qhull over all candidates -> the same.

usage (GPU box, from the repo root):  python tests/golden/make_sharded_golden.py [2d] [3d]   -> tests/golden/sharded_fullsize.json
(copy it from gpurun_out/ into tests/golden/ and commit it)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import _bigparity as B  # noqa: E402
from oracle import ref  # noqa: E402


def mem_available_gb():
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable"):
            return int(line.split()[1]) / 1e6
    return 0.0


def main():
    which = sys.argv[1:] or ["2d", "3d"]
    dev = torch.device("cuda:0")
    threads = min(os.cpu_count() or 1, 64)
    out_path = os.path.join(ROOT, "gpurun_out", "sharded_fullsize.json")
    out = json.load(open(out_path)) if os.path.exists(out_path) else {}
    print("host: %d cpus, %.0f GB available" % (os.cpu_count(), mem_available_gb()), flush=True)
    if "2d" in which:
        cfg = B.CFG2D
        model, big, axes = B.model_and_input(2, cfg, dev)
        t0 = time.time()
        dist, prob, pts, nb = B.whole_input_candidates(model, big, axes, cfg)
        n = int(prob.numel())
        print("2D: %d candidates from %d blocks in %.1f s" % (n, nb, time.time() - t0), flush=True)
        need = n * (512 + 200) / 1e9
        if mem_available_gb() < 2.5 * need + 20:
            print("2D: not enough host memory for the reference run (%.0f GB needed)" % (2.5 * need + 20)); return 1
        d = dist.cpu().numpy().astype(np.float32); p = pts.cpu().numpy().astype(np.float32)
        ref.stardist2d(); ref.set_threads(threads)
        t0 = time.time()
        keep = ref.stardist2d().c_non_max_suppression_inds(d, p, 1, 1, 0, np.float32(model.thresholds.nms)).astype(bool)
        t_ref = time.time() - t0
        print("2D: reference NMS over %d candidates: %d survivors, %.1f s on %d threads" % (n, int(keep.sum()), t_ref, threads), flush=True)
        kd = torch.from_numpy(keep).to(dev)
        from stardist_amd.geometry.geom2d import polygons_to_label
        lab = polygons_to_label(dist[kd], pts[kd], tuple(big.shape), prob=prob[kd])
        lab = lab.cpu().numpy() if torch.is_tensor(lab) else np.asarray(lab)
        out["2d"] = dict(cfg, candidates=n, survivors=int(keep.sum()), points_sha256=B.points_digest(p[keep]), labels_sha256=B.array_digest(lab.astype(np.int32)),
                         keep_sha256=B.array_digest(np.packbits(keep)), reference_seconds=round(t_ref, 1), threads=threads, nms_thresh=float(model.thresholds.nms))
        json.dump(out, open(out_path, "w"), indent=1)
        del model, big, dist, prob, pts, d, p, lab
        torch.cuda.empty_cache()
    if "3d" in which:
        cfg = B.CFG3D_REF
        model, big, axes = B.model_and_input(3, cfg, dev)
        from stardist_amd.rays3d import rays_from_json
        rays = rays_from_json(model.config.rays_json)
        V, F = np.ascontiguousarray(rays.vertices, np.float32), np.ascontiguousarray(rays.faces, np.int32)
        t0 = time.time()
        dist, prob, pts, nb = B.whole_input_candidates(model, big, axes, cfg)
        n = int(prob.numel())
        print("3D: %d candidates from %d blocks in %.1f s" % (n, nb, time.time() - t0), flush=True)
        d = dist.cpu().numpy().astype(np.float32); p = pts.cpu().numpy().astype(np.float32); s = prob.cpu().numpy().astype(np.float32)
        m3 = ref.stardist3d(); ref.set_threads(threads)
        t0 = time.time()
        keep = m3.c_non_max_suppression_inds(d, p, V, F, s, 1, 1, 0, np.float32(model.thresholds.nms)).astype(bool)
        t_ref = time.time() - t0
        print("3D: reference NMS over %d candidates: %d survivors, %.1f s on %d threads" % (n, int(keep.sum()), t_ref, threads), flush=True)
        out["3d"] = dict(cfg, candidates=n, survivors=int(keep.sum()), points_sha256=B.points_digest(p[keep]), keep_sha256=B.array_digest(np.packbits(keep)),
                         reference_seconds=round(t_ref, 1), threads=threads, nms_thresh=float(model.thresholds.nms))
        json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
