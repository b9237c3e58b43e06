"""The reference's own test fixture images (tests/data/img2d.tif 256x256 uint8, img3d.tif 64x128x128 uint16 multi-page; read through
Pillow exactly as a user without tifffile would) stored as one compressed .npz so that the GPU box -- which has no /root/reference --
can run the config-1 substitute of SURVEY.md 8d (2D_demo topology on the reference's fixture image).  TEST INFRASTRUCTURE.
usage: python tests/golden/make_fixture_images.py"""
import os

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = "/root/reference/tests/data"


def read(path):
    im = Image.open(path)
    pages = []
    for k in range(getattr(im, "n_frames", 1)):
        im.seek(k)
        pages.append(np.array(im))
    return pages[0] if len(pages) == 1 else np.stack(pages)


out = {name: read(os.path.join(SRC, name + ".tif")) for name in ("img2d", "mask2d", "img3d", "mask3d")}
for k, v in out.items():
    print(k, v.shape, v.dtype, int(v.min()), int(v.max()))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "fixture_images.npz"), **out)
