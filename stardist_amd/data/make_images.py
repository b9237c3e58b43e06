"""The reference's sample images (stardist/data/images: img2d / mask2d, img3d / mask3d, histo.jpg -- data, the files its own tests and demos
read through `stardist.data`) stored as one compressed .npz next to this file, so that `stardist_amd.data` serves them without tifffile /
imageio.  Run in the build container only (needs /root/reference and Pillow):  python -m stardist_amd.data.make_images"""
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/stardist/data/images"


def read(path):
    im = Image.open(path)
    pages = []
    for k in range(getattr(im, "n_frames", 1)):
        im.seek(k)
        pages.append(np.array(im))
    return pages[0] if len(pages) == 1 else np.stack(pages)


if __name__ == "__main__":
    out = {n: read(os.path.join(SRC, n + ".tif")) for n in ("img2d", "mask2d", "img3d", "mask3d")}
    out["histo"] = np.array(Image.open(os.path.join(SRC, "histo.jpg")).convert("RGB"))
    for k, v in out.items():
        print(k, v.shape, v.dtype, int(v.min()), int(v.max()))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "images.npz"), **out)
