"""import this FIRST in a probe tool: STARDIST_AMD_PROBE_LIB=NAME points the ctypes binding at stardist_amd/csrc/libstardist_hip_NAME.so
(tools/build_variant.py) for this process; unset: the product library"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stardist_amd.lib import _native as _N  # noqa: E402

_name = os.environ.get("STARDIST_AMD_PROBE_LIB")
if _name:
    _N.LIB_PATH = os.path.join(os.path.dirname(_N.LIB_PATH), "libstardist_hip_%s.so" % _name)
    print("probe library:", os.path.abspath(_N.LIB_PATH), flush=True)
