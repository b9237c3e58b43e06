"""TIFF read/write for the command line scripts: tifffile / imageio when they are installed (as in the reference scripts),
otherwise Pillow (multi-page TIFF = Z stack)."""
import numpy as np


def imread(fname):
    try:
        from tifffile import imread as _r
        return _r(str(fname))
    except ImportError:
        pass
    from PIL import Image
    im = Image.open(str(fname))
    frames = []
    for i in range(getattr(im, "n_frames", 1)):
        im.seek(i)
        frames.append(np.array(im))
    return frames[0] if len(frames) == 1 else np.stack(frames)


def imwrite(fname, arr):
    arr = np.asarray(arr)
    try:
        from tifffile import imwrite as _w
        _w(str(fname), arr, compression="zlib")
        return
    except ImportError:
        pass
    from PIL import Image
    if arr.dtype not in (np.uint8, np.uint16, np.int32, np.float32):
        arr = arr.astype(np.int32)
    if arr.ndim == 2:
        Image.fromarray(arr).save(str(fname), compression="tiff_deflate")
    else:
        pages = [Image.fromarray(a) for a in arr]
        pages[0].save(str(fname), save_all=True, append_images=pages[1:], compression="tiff_deflate")
