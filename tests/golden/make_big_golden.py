"""Golden block covers from the reference's stardist/big.py (Block.cover, :170-279; is_responsible :89-122).
big.py cannot be imported (scikit-image/csbdeep missing): the Block class + helpers are exec'd from the source text."""
import math, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open("/root/reference/stardist/big.py").read()
a = src.index("class Block:"); b = src.index("class BlockND:")
c = src.index("class NotFullyVisible"); d = src.index("# def render_polygons")
ns = {"np": np, "math": math}
exec(src[c:d] + "\n" + src[a:b], ns)
Block = ns["Block"]
cases = [(100, 30, 4, 2, 1), (1000, 256, 32, 16, 8), (16384, 4096, 128, 128, 16), (1024, 256, 32, 32, 4), (333, 64, 8, 4, 2),
         (77, 33, 5, 3, 1), (2048, 512, 96, 48, 16), (999, 200, 20, 10, 1), (130, 128, 16, 8, 8), (64, 64, 0, 0, 1)]
out = {}
for k, (size, bs, mo, ctx, grid) in enumerate(cases):
    blocks = Block.cover(size, bs, mo, ctx, grid, verbose=False)
    rows = []
    for t in blocks:
        r_start = 0 if t.at_begin else (t.pred.overlap - t.pred.context_end - t.context_start)
        rows.append([t.start, t.end, t.slice_write.start, t.slice_write.stop, t.context_start, t.context_end, r_start])
    out["case%d" % k] = np.array(rows)
    out["args%d" % k] = np.array([size, bs, mo, ctx, grid])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "big_cover.npz"), **out)
print({k: v.shape for k, v in out.items() if k.startswith("case")})
