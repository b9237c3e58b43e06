"""Copies the config.json / thresholds.json of the reference's example model folders (models/examples/2D_demo, 3D_demo;
their weight files are not in the offline tree, .MISSING_LARGE_BLOBS) into tests/golden/pretrained/ -- the folder layout
`from_pretrained` unpacks (TEST INFRASTRUCTURE).  usage: python tests/golden/make_pretrained_fixture.py"""
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for cls, key in (("StarDist2D", "2D_demo"), ("StarDist3D", "3D_demo")):
    dst = os.path.join(ROOT, "tests", "golden", "pretrained", cls, key)
    os.makedirs(dst, exist_ok=True)
    for f in ("config.json", "thresholds.json"):
        shutil.copy(os.path.join("/root/reference/models/examples", key, f), os.path.join(dst, f))
print("ok")
