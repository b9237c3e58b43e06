import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stardist_amd.lib import stardist2d as sd2
a = np.loadtxt(os.path.join(os.path.dirname(__file__), "_pairs_dbg.txt"), dtype=np.int64).astype(np.int32)
R = 32
tw, fl = sd2.clip_pairs(a[:, :R], a[:, R:2*R], a[:, 2*R:3*R], a[:, 3*R:])
print("RESULT", tw[987], fl[987])
