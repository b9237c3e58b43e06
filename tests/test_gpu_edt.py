"""GPU: edt_prob (csrc/edt.hip, the training target of stardist/utils.py:71-125) against the goldens made by the reference's own
function and against the exhaustive oracle (oracle.port.edt_prob, itself pinned to those goldens) on larger label images."""
import os
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_edt_prob_equals_reference_goldens():
    from stardist_amd import utils
    g = np.load(os.path.join(ROOT, "tests", "golden", "utils_reference.npz"))
    assert np.array_equal(utils.edt_prob(g["edt_lab2"]), g["edt_prob2"])
    assert np.array_equal(utils.edt_prob(g["edt_lab3"], anisotropy=(2.0, 1.0, 1.0)), g["edt_prob3"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert np.array_equal(utils.edt_prob(g["edt_const"]), g["edt_prob_const"])


def _blobs(shape, n, seed):
    rs = np.random.RandomState(seed)
    lab = np.zeros(shape, np.int32)
    grid = np.stack(np.meshgrid(*[np.arange(s) for s in shape], indexing="ij"), -1)
    for k in range(1, n + 1):
        c = np.array([rs.uniform(0, s) for s in shape])
        r = rs.uniform(2.5, 9.0, len(shape))
        lab[(((grid - c) / r) ** 2).sum(-1) <= 1.0] = k if k != 5 else 40            # touching objects, a gap in the ids
    return lab


@pytest.mark.parametrize("shape,n,aniso", [((96, 120), 30, None), ((70, 64), 12, (1.5, 1.0)), ((14, 40, 48), 14, (2.0, 1.0, 1.0)),
                                           ((1, 33, 40), 6, None), ((30, 1, 25), 4, (1.0, 3.0, 0.5))])
def test_edt_prob_equals_exhaustive_oracle(shape, n, aniso):
    import torch
    from oracle import port
    from stardist_amd import utils
    lab = _blobs(shape, n, seed=len(shape) * 100 + n)
    want = port.edt_prob(lab, anisotropy=aniso)
    got = utils.edt_prob(lab, anisotropy=aniso)
    assert got.dtype == np.float32 and got.shape == lab.shape
    assert np.array_equal(got, want), float(np.abs(got - want).max())
    t = utils.edt_prob(torch.from_numpy(lab).cuda(), anisotropy=aniso)                   # device tensor in -> device tensor out
    assert t.is_cuda and np.array_equal(t.cpu().numpy(), want)
    assert (got[lab == 0] == 0).all() and got.max() <= 1.0


def test_training_targets_equal_reference_composition(refmods):
    """stardist_targets (edt_prob + star_dist on the GPU) against the composition the reference's data generator does per sample
    (model2d.py:63-104): oracle edt_prob + the compiled reference c_star_dist, incl. grid subsampling, negative labels and the mask channel"""
    from oracle import port
    from stardist_amd.rays3d import Rays_GoldenSpiral
    from stardist_amd.targets import stardist_targets
    labs = [_blobs((64, 80), 10, seed=s) for s in (1, 2)]
    labs[1][5:9, 5:9] = -1
    for grid in ((1, 1), (2, 2)):
        prob, dm = stardist_targets(labs, n_rays=32, grid=grid)
        for k, y in enumerate(labs):
            y0 = np.maximum(y, 0)
            want_p = port.edt_prob(y0[::grid[0], ::grid[1]])
            want_d = port.star_dist(y0, 32, grid=grid)
            assert np.array_equal(dm[k, ..., :-1], want_d) and np.array_equal(dm[k, ..., -1], want_p)
            negm = y[::grid[0], ::grid[1]] < 0
            assert np.array_equal(prob[k, ..., 0][~negm], want_p[~negm]) and (prob[k, ..., 0][negm] == -1).all()
    rays = Rays_GoldenSpiral(24)
    lab3 = [_blobs((12, 32, 36), 6, seed=7)]
    prob, dm = stardist_targets(lab3, rays=rays, grid=(1, 2, 2), anisotropy=(2.0, 1.0, 1.0))
    assert np.array_equal(dm[0, ..., :-1], port.star_dist3D(lab3[0], rays.vertices, grid=(1, 2, 2)))
    # StarDistData3D computes the distance transform at FULL resolution and subsamples afterwards (model3d.py:88) -- unlike the 2D
    # generator (model2d.py:86); the two orders differ (this used to subsample first in 3D as well)
    want3 = port.edt_prob(lab3[0], anisotropy=(2.0, 1.0, 1.0))[:, ::2, ::2]
    assert np.array_equal(prob[0, ..., 0], want3) and np.array_equal(dm[0, ..., -1], want3)
    assert not np.array_equal(want3, port.edt_prob(lab3[0][:, ::2, ::2], anisotropy=(2.0, 1.0, 1.0)))
    prob2, dm2 = stardist_targets(lab3, rays=rays, grid=(2, 2, 2))
    assert np.array_equal(prob2[0, ..., 0], port.edt_prob(lab3[0])[::2, ::2, ::2])


def test_edt_prob_huge_and_sparse_label_ids():
    """ids beyond int32 / hash-like sparse ids are compacted first: same result as with small consecutive ids"""
    from stardist_amd import utils
    lab = _blobs((40, 56), 7, seed=4).astype(np.int64)
    want = utils.edt_prob(lab)
    big = np.where(lab > 0, lab * 7919 + 2 ** 33, 0)
    assert np.array_equal(utils.edt_prob(big), want)
    sparse = np.where(lab > 0, lab * 1000003, 0)
    assert np.array_equal(utils.edt_prob(sparse), want)
