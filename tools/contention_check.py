"""Are the results of the path independent of what else runs on the device?  P processes share ONE GPU; each predicts the bench's 2048^2
tile REPS times (network -> selection -> NMS through predict_sparse_device + the 2D NMS native) and reports the CRCs of prob / dist / points
and of the NMS keep flags per repetition.  Every CRC must be the same in every repetition and every process (same seed, same weights).
usage: python tools/contention_check.py P REPS [2d|3d]      (the parent spawns P children and compares their lines)"""
import os, sys, subprocess, zlib, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(reps, which):
    import numpy as np, torch
    import bench
    from oracle import synth
    from stardist_amd import nms
    from stardist_amd.lib import stardist2d as sd2, stardist3d as sd3
    dev = torch.device("cuda:0")
    crc = lambda t: "%08x" % zlib.crc32(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes())
    if which == "2d":
        from stardist_amd.models import Config2D, StarDist2D
        img = torch.from_numpy(synth.s2d_nuclei_image(2048, 2048, seed=0)).to(dev)
        m = StarDist2D(Config2D(n_rays=32), basedir=None, device=dev, seed=0)
        bench.calibrate_heads(m, img)
    else:
        from stardist_amd.models import Config3D, StarDist3D
        img = torch.from_numpy(synth.s3d_nuclei_image(256, seed=0)).to(dev)
        m = StarDist3D(Config3D(rays=96), basedir=None, device=dev, seed=0)
        m.thresholds = dict(prob=0.5, nms=0.3)
        bench.calibrate_heads(m, img, frac=0.009, radius=8.5, noise=0.03)
    from stardist_amd.lib import _native
    for kv in filter(None, os.environ.get("SD_OPTS", "").split(",")):          # SD_OPTS="name=value,...": any sd_set_option switch
        k_, v_ = kv.split("=")
        _native.check(_native.lib().sd_set_option(k_.encode(), int(v_)))
    keep0 = None
    for r in range(reps):
        res = m.predict_sparse_device(img, prob_thresh=0.5)
        prob, dist, points = res[0], res[1], res[-1]
        o = nms._argsort_desc(prob)
        td = dist[o].float().contiguous(); tp = points[o].float().contiguous()
        if which == "2d":
            keep = sd2.c_non_max_suppression_inds(td, tp, 1, 1, 0, np.float32(0.4))
        else:
            rays = m._rays if hasattr(m, "_rays") else None
            from stardist_amd.rays3d import rays_from_json
            rays = rays_from_json(m.config.rays_json)
            keep = nms.non_maximum_suppression_3d_inds(td, tp, rays=rays, scores=prob[o].float().contiguous(), thresh=0.3)
        torch.cuda.synchronize()
        k = keep if torch.is_tensor(keep) else torch.from_numpy(np.asarray(keep))
        if keep0 is None:
            keep0 = k.clone()
        elif not torch.equal(keep0.to(torch.uint8), k.to(torch.uint8)):
            ix = torch.nonzero(keep0.to(torch.uint8) != k.to(torch.uint8)).flatten().tolist()
            print("DIFF pid %d rep %d: positions (score order) %s, flags there now %s, before %s" % (os.getpid(), r, ix[:12], [int(k[i]) for i in ix[:12]], [int(keep0[i]) for i in ix[:12]]), flush=True)
        st = _native.last_stats.get("nms2d" if which == "2d" else "nms3d")
        stl = [int(v) for v in st] if st is not None else []
        if which == "2d" and len(stl) > 10:
            stl = [stl[0], stl[1], stl[2], stl[3], stl[8], stl[9], stl[10]]      # pairs, general-path pairs, rounds, neighbour entries, spilled, decided by the band, deferred undecided
        print("pid %d rep %d: n=%d prob %s dist %s points %s keep %s survivors %d stats %s" % (os.getpid(), r, prob.numel(), crc(prob), crc(dist), crc(points), crc(k.to(torch.uint8)), int(k.sum()), stl), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "child":
        child(int(sys.argv[2]), sys.argv[3])
    else:
        P, reps = int(sys.argv[1]), int(sys.argv[2]); which = sys.argv[3] if len(sys.argv) > 3 else "2d"
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "child", str(reps), which], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(P)]
        lines = []
        for p in procs:
            out, _ = p.communicate(timeout=900)
            lines += [l for l in out.splitlines() if l.startswith("pid")]
            for l in out.splitlines():
                if l.startswith("DIFF"):
                    print(l)
        sigs = {}
        for l in lines:
            sig = l.split(": ", 1)[1]
            sigs.setdefault(sig, []).append(l.split(":")[0])
        print("SD_OPTS=%r" % os.environ.get("SD_OPTS", ""))
        print("%s, %d processes x %d repetitions on one device: %d result lines, %d distinct signatures" % (which, P, reps, len(lines), len(sigs)))
        for sig, who in sorted(sigs.items(), key=lambda kv: -len(kv[1])):
            print("  %4d x  %s%s" % (len(who), sig, "" if len(who) > 3 else "   <- " + ", ".join(who)))
