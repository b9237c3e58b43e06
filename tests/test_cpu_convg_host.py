"""CPU: the general convolution kernel's layouts (csrc/conv_general.hip) without a GPU.  The host packer of the C ABI writes the
device layout; this test reads the packed buffer back with the index expressions THE KERNEL uses (k order, lane halves, tap table,
padding-before / stride coordinate algebra) and checks that the sum the kernel forms equals the convolution (float64 torch) for the
layer shapes of the reference's networks it serves: ResNet stem 7x7x7, strided 3x3x3 with TensorFlow 'same' padding, strided 1x1x1
projection (stardist/models/model3d.py:400-447), 3-channel first layer (model2d.py:310-316), narrow 1x1 head."""
import numpy as np
import pytest


def _emulate(x, w, bias, k3, s3, p3, O3):
    """x (D,H,W,C) float64, w (co,ci,kz,ky,kx) float32 -> (Do,Ho,Wo,co) float64, following k_convg_vec / k_convg_small"""
    from stardist_amd.lib import _native as N
    L = N.lib()
    co, ci = w.shape[:2]
    kz, ky, kx = k3
    n = L.sd_convg_packed_floats(ci, co, kz, ky, kx)
    assert n > 0
    packed = np.zeros(n, np.float32)
    N.check(L.sd_convg_pack_weights_host(N.ptr(np.ascontiguousarray(w)), ci, co, kz, ky, kx, N.ptr(packed)))
    groups, T = (co + 31) // 32, kz * ky * kx
    D, H, W, _ = x.shape
    Do, Ho, Wo = O3
    out = np.zeros((Do, Ho, Wo, groups * 32))
    zo, yo, xo = np.meshgrid(np.arange(Do), np.arange(Ho), np.arange(Wo), indexing="ij")

    def gather(dz, dy, dx, ch):
        iz, iy, ix = zo * s3[0] - p3[0] + dz, yo * s3[1] - p3[1] + dy, xo * s3[2] - p3[2] + dx
        ok = (iz >= 0) & (iz < D) & (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W)
        v = x[np.clip(iz, 0, D - 1), np.clip(iy, 0, H - 1), np.clip(ix, 0, W - 1), ch]
        return np.where(ok, v, 0.0)
    if ci % 32 == 0:
        nch = ci // 32
        P = packed.reshape(groups, T, nch, 4, 2, 32, 4)                 # g, tap, c, j, h, i, e   (v4f index * 4 + e)
        for g in range(groups):
            acc = np.zeros((Do, Ho, Wo, 32))
            for tap in range(T):
                dz, dy, dx = tap // (ky * kx), (tap // kx) % ky, tap % kx
                for c in range(nch):
                    for j in range(4):
                        for e in range(4):
                            for h in range(2):
                                a = gather(dz, dy, dx, c * 32 + h * 16 + j * 4 + e)
                                acc += a[..., None] * P[g, tap, c, j, h, :, e].astype(np.float64)
            out[..., g * 32:(g + 1) * 32] = acc
    else:
        K = T * ci
        kp = (K + 7) // 8 * 8
        n_k4 = kp // 8
        P = packed[:groups * n_k4 * 256].reshape(groups, n_k4, 2, 32, 4)  # g, m, h, i, e
        tab = packed[groups * n_k4 * 256:].view(np.int32)
        assert len(tab) == kp
        for g in range(groups):
            acc = np.zeros((Do, Ho, Wo, 32))
            for m in range(n_k4):
                for e in range(4):
                    for h in range(2):
                        t = int(tab[m * 8 + e * 2 + h])
                        if t < 0:
                            assert not P[g, m, h, :, e].any()
                            continue
                        dz, dy, dx, c = t & 63, (t >> 6) & 63, (t >> 12) & 63, (t >> 18) & 0x1fff
                        acc += gather(dz, dy, dx, c)[..., None] * P[g, m, h, :, e].astype(np.float64)
            out[..., g * 32:(g + 1) * 32] = acc
    assert not out[..., co:].any()                                       # padded output channels carry zero weights
    return out[..., :co] + bias.astype(np.float64)


CASES = [  # nd, c_in, c_out, kernel, stride, spatial, tf_same
    (3, 1, 32, (7, 7, 7), (1, 1, 1), (5, 9, 11), False),
    (3, 32, 64, (3, 3, 3), (1, 2, 2), (4, 9, 10), True),
    (3, 32, 64, (1, 1, 1), (1, 2, 2), (4, 9, 10), True),
    (3, 64, 32, (3, 3, 3), (2, 2, 2), (5, 7, 8), True),
    (2, 3, 32, (3, 3), (1, 1), (9, 13), False),
    (2, 64, 5, (1, 1), (1, 1), (6, 7), False),
    (2, 5, 40, (5, 5), (1, 1), (8, 9), False),
]


@pytest.mark.parametrize("nd,ci,co,k,s,S,tf_same", CASES)
def test_packed_layout_and_coordinates_reproduce_the_convolution(nd, ci, co, k, s, S, tf_same):
    import torch
    import torch.nn.functional as F
    from stardist_amd.models.unet import tf_same_pad_before
    rs = np.random.RandomState(0)
    w = rs.randn(co, ci, *k).astype(np.float32)
    b = rs.randn(co).astype(np.float32)
    x = rs.randn(*S, ci)
    k3, s3, S3 = (1,) * (3 - nd) + tuple(k), (1,) * (3 - nd) + tuple(s), (1,) * (3 - nd) + tuple(S)
    if tf_same:
        p3 = tuple(tf_same_pad_before(n, kk, st) for n, kk, st in zip(S3, k3, s3))
        O3 = tuple(-(-n // st) for n, st in zip(S3, s3))
        tot = [max(kk - st, 0) if n % st == 0 else max(kk - n % st, 0) for n, kk, st in zip(S3, k3, s3)]
        pads = []
        for d in reversed(range(3)):
            pads += [tot[d] // 2, tot[d] - tot[d] // 2]
        xt = F.pad(torch.from_numpy(x.reshape(S3 + (ci,))).permute(3, 0, 1, 2)[None], pads)
        want = F.conv3d(xt, torch.from_numpy(w.reshape((co, ci) + k3)).double(), torch.from_numpy(b).double(), stride=s3)
    else:
        p3 = tuple(kk // 2 for kk in k3)
        O3 = tuple((n + 2 * p - kk) // st + 1 for n, p, kk, st in zip(S3, p3, k3, s3))
        xt = torch.from_numpy(x.reshape(S3 + (ci,))).permute(3, 0, 1, 2)[None]
        want = F.conv3d(xt, torch.from_numpy(w.reshape((co, ci) + k3)).double(), torch.from_numpy(b).double(), stride=s3, padding=p3)
    want = want[0].permute(1, 2, 3, 0).numpy()
    got = _emulate(x.reshape(S3 + (ci,)), w.reshape((co, ci) + k3), b, k3, s3, p3, O3)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-6 * max(1.0, np.abs(want).max())


def test_unsupported_shapes_are_refused():
    from stardist_amd.lib import _native as N
    L = N.lib()
    assert L.sd_convg_packed_floats(1000, 32, 1, 3, 3) == -1            # neither 32-channel chunks nor the small form
    assert L.sd_convg_packed_floats(32, 32, 1, 64, 1) == -1
    assert L.sd_convg_packed_floats(1, 32, 7, 7, 7) == 1 * (344 // 8) * 256 + 344
    assert L.sd_convg_packed_floats(64, 5, 1, 1, 1) == 1 * 1 * 2 * 1024
