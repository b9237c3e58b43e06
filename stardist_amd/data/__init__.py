"""Sample images (mirror of stardist/data/__init__.py:7-39): the same arrays the reference's accessors return, served from `images.npz`
(made from the reference's image files by make_images.py) instead of through tifffile / imageio."""
import os

import numpy as np

_CACHE = {}


def abspath(path):
    return os.path.join(os.path.abspath(os.path.dirname(__file__)), path)


def _img(name):
    if not _CACHE:
        p = abspath("images.npz")
        if not os.path.exists(p):
            raise FileNotFoundError("%s is missing (python -m stardist_amd.data.make_images in a checkout that has the reference's images)" % p)
        with np.load(p) as d:
            _CACHE.update({k: d[k] for k in d.files})
    return _CACHE[name].copy()


def test_image_nuclei_2d(return_mask=False):
    """Fluorescence microscopy image and mask from the 2018 kaggle DSB challenge (Caicedo et al., Nature Methods 16.12)"""
    img, mask = _img("img2d"), _img("mask2d")
    return (img, mask) if return_mask else img


def test_image_he_2d():
    """H&E stained RGB example image from the Cancer Imaging Archive (https://www.cancerimagingarchive.net)"""
    return _img("histo")


def test_image_nuclei_3d(return_mask=False):
    """synthetic nuclei"""
    img, mask = _img("img3d"), _img("mask3d")
    return (img, mask) if return_mask else img


test_image_nuclei_2d.__test__ = test_image_he_2d.__test__ = test_image_nuclei_3d.__test__ = False     # (accessors, not tests)
