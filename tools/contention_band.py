"""Is the decision shortcut's probe (k_poly_props + pair_enclosure: sd_area_bounds_pairs_device) a function of its input alone?  P processes share
ONE GPU; each evaluates the same 400 000 seeded pairs REPS times and reports the CRC of (area, band, info).  usage: python tools/contention_band.py P REPS"""
import os, sys, subprocess, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(reps):
    import numpy as np
    from test_gpu_parity2d import _star_polys
    from stardist_amd.lib import stardist2d as sd2
    rng = np.random.RandomState(7)
    n = 400000
    xa, ya = _star_polys(rng, n, 32, 10, 0.1, 12)
    xb, yb = _star_polys(rng, n, 32, 8, 0.1, 12)
    first = None
    for r in range(reps):
        area, band, usable, K, T = sd2.area_bounds_pairs(xa, ya, xb, yb)
        sig = "%08x %08x %08x" % (zlib.crc32(area.tobytes()), zlib.crc32(band.tobytes()), zlib.crc32(np.ascontiguousarray(usable).tobytes() + np.ascontiguousarray(K).tobytes() + np.ascontiguousarray(T).tobytes()))
        if first is None:
            first = (area.copy(), band.copy(), usable.copy(), K.copy(), T.copy())
        else:
            bad = np.nonzero((area != first[0]) | (band != first[1]) | (usable != first[2]) | (K != first[3]) | (T != first[4]))[0]
            if len(bad):
                b = bad[:6]
                print("DIFF pid %d rep %d: %d pairs differ, e.g. %s: area %s vs %s, band %s vs %s, usable %s vs %s, K %s vs %s, T %s vs %s" % (
                    os.getpid(), r, len(bad), b.tolist(), area[b], first[0][b], band[b], first[1][b], usable[b], first[2][b], K[b], first[3][b], T[b], first[4][b]), flush=True)
        print("pid %d rep %d: %s" % (os.getpid(), r, sig), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "child":
        child(int(sys.argv[2]))
    else:
        P, reps = int(sys.argv[1]), int(sys.argv[2])
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "child", str(reps)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(P)]
        sigs = {}; nl = 0
        for p in procs:
            out, _ = p.communicate(timeout=900)
            for l in out.splitlines():
                if l.startswith("DIFF"):
                    print(l[:600])
                elif l.startswith("pid"):
                    nl += 1; sigs.setdefault(l.split(": ", 1)[1], []).append(l.split(":")[0])
        print("band probe, %d processes x %d repetitions on one device: %d result lines, %d distinct signatures" % (P, reps, nl, len(sigs)))
        for sig, who in sorted(sigs.items(), key=lambda kv: -len(kv[1])):
            print("  %4d x  %s%s" % (len(who), sig, "" if len(who) > 3 else "   <- " + ", ".join(who)))
