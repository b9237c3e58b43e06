"""GPU: the prepared-polygon records (Clipper::AddPath once per candidate, stardist_amd/csrc/clip_beam.h) written by the
device kernel must equal, byte for byte, the records the same header produces when compiled for the host -- the host
build is the one pinned against the reference's Clipper by tests/host/beam_check.cpp."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hostlib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("beamprep") / "libbeam_prep_host.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", os.path.join(ROOT, "tests", "host", "beam_prep_lib.cpp"), "-o", so], check=True)
    lib = ctypes.CDLL(so)
    lib.beam_prep_record_bytes.restype = ctypes.c_long
    return lib


def _star_polys(rng, n, R, radius, noise, spread):
    ang = np.float32(2 * np.pi / R)
    k = np.arange(R, dtype=np.int32)
    s = np.sin((ang * k).astype(np.float32)).astype(np.float32)
    c = np.cos((ang * k).astype(np.float32)).astype(np.float32)
    d = np.maximum((radius * (1 + noise * rng.uniform(-1, 1, (n, R)))).astype(np.float32), np.float32(1e-3))
    p = np.floor(rng.uniform(50, 50 + spread, (n, 2))).astype(np.float32)
    y = (p[:, :1] + d * s).astype(np.float32)
    x = (p[:, 1:] + d * c).astype(np.float32)
    return np.ascontiguousarray(x.astype(np.int64).astype(np.int32)), np.ascontiguousarray(y.astype(np.int64).astype(np.int32))


@pytest.mark.parametrize("R,radius,noise", [(32, 10, 0.1), (32, 3, 0.5), (32, 2, 0.9), (11, 10, 0.3), (64, 20, 0.3), (100, 30, 0.6), (200, 40, 0.5), (5, 1, 0.5)])
def test_device_prepare_equals_host(hostlib, R, radius, noise):
    import torch
    from stardist_amd.lib import _native as N
    rng = np.random.RandomState(R * 7 + int(radius))
    n = 20000
    x, y = _star_polys(rng, n, R, radius, noise, 12)
    rec = hostlib.beam_prep_record_bytes(R)
    host = np.full(n * rec, 0xAB, np.uint8)
    hostlib.beam_prepare_host(x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p), n, R, host.ctypes.data_as(ctypes.c_void_p))
    dev = torch.device("cuda")
    tx, ty = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    out = torch.full((n * rec,), 0xAB, dtype=torch.uint8, device=dev)
    N.check(N.lib().sd_prepare_polys_device(N.tptr(tx), N.tptr(ty), n, R, N.tptr(out), n * rec, N.current_stream()))
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(n, rec)
    host = host.reshape(n, rec)
    bad = np.flatnonzero((got != host).any(1))
    assert len(bad) == 0, "%d of %d records differ, first: polygon %d x=%s y=%s bytes %s" % (
        len(bad), n, bad[0], x[bad[0]].tolist(), y[bad[0]].tolist(), np.flatnonzero(got[bad[0]] != host[bad[0]])[:16].tolist())
