"""Probe builds of the C-ABI library with experiment macros, next to the product library:
    python tools/build_variant.py NAME file.hip=-DMACRO=1 [file2.hip=-DX=2 ...]
compiles the named sources with the extra flags, links them with the product build's other objects into
stardist_amd/csrc/libstardist_hip_NAME.so.  A probe TOOL selects it with STARDIST_AMD_PROBE_LIB=NAME (read by tools/_probe_lib.py before
stardist_amd is imported; the package itself never reads it): A/B timings of kernel variants in one GPU call."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stardist_amd import build  # noqa: E402


def main():
    name = sys.argv[1]
    extra = dict(a.split("=", 1) for a in sys.argv[2:])
    build.build_lib(verbose=False)
    objdir = os.path.join(build.CSRC, "build")
    objs = []
    for src in build._sources():
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        if src in extra:
            obj = os.path.join(objdir, src.replace(".hip", "_%s.o" % name))
            cmd = [build.HIPCC] + build.FLAGS + extra[src].split() + ["-c", os.path.join(build.CSRC, src), "-o", obj]
            print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        objs.append(obj)
    lib = os.path.join(build.CSRC, "libstardist_hip_%s.so" % name)
    subprocess.run([build.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib], check=True)
    print(lib)


if __name__ == "__main__":
    main()
