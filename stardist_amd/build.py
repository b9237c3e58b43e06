"""Build libstardist_hip.so (the C-ABI library of include/stardist_hip.h) for gfx950.

`python -m stardist_amd.build` or `stardist_amd.build.build_lib()`; hipcc cross-compiles
without a GPU.  The .so is written next to the sources (stardist_amd/csrc/) so it travels
with the tree to the GPU box; it is git-ignored.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libstardist_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the reference natives are built for baseline x86-64 (no FMA, setup.py:109,115);
# every float expression restated from them must round after each operation.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wno-unused-result", "-I" + os.path.join(HERE, "..", "include")]
# probe builds only (tools/time_nms2d_bench.py and friends): SD_BUILD_DEBUG_SWITCHES=1 compiles the A/B tuning knobs in, which are then
# read from the environment (SD_NMS_PAIR_SORT, SD_NMS_PAIR_KEY, SD_NMS_TAIL_DIV, SD_NMS_TAIL_MAX, SD_NMS_DEFER); release builds ignore them
if os.environ.get("SD_BUILD_DEBUG_SWITCHES") == "1":
    FLAGS.append("-DSD_DEBUG_SWITCHES")


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest():
    h = hashlib.sha1()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(fh.read())
    with open(os.path.join(HERE, "..", "include", "stardist_hip.h"), "rb") as fh:
        h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build_lib(force=False, verbose=True):
    stamp = os.path.join(CSRC, ".build_stamp")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)

    def cc(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(cc, _sources()))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    with open(stamp, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
