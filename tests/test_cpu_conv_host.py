"""CPU: host emulation of the hand-written convolution's data flow (tests/host/conv_check.cpp): halo staging, packed weights,
per-lane operand fetch, f32 MFMA semantics and the accumulator -> pixel map, all through the index functions the device kernel
uses (stardist_amd/csrc/conv3x3_layout.h), against a direct float64 convolution; and the C ABI's weight packer against the layout."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_conv_layout_emulation(tmp_path):
    exe = str(tmp_path / "conv_check")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "host", "conv_check.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


def test_conv_bf16_layout_emulation(tmp_path):
    """the split-bf16 kernel's element assignment (one owner per halo element, conflict-free LDS stores), its address rule on interior
    and border tiles, the f32 -> 3 x bf16 split, the weight packer and a whole layer through the LDS layouts vs float64"""
    exe = str(tmp_path / "conv_bf16_check")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "host", "conv_bf16_check.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


def test_conv_f16_layout_emulation(tmp_path):
    """the split-fp16 kernel (the default): conflict-free LDS stores and A-operand reads with the two-plane tile, the buffer-load offset
    rule on interior and border tiles, the f32 -> 2 x fp16 split, the weight packer, and a whole layer through the LDS layouts with the
    three products in two accumulators vs float64"""
    exe = str(tmp_path / "conv_f16_check")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "host", "conv_f16_check.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


def test_pack_weights_abi():
    from stardist_amd.lib import _native as N
    l = N.lib()
    assert l.sd_conv3_packed_floats(544, 32, 1) == -1 and l.sd_conv3_packed_floats(300, 32, 1) == -1 and l.sd_conv3_packed_floats(32, 48, 1) == -1 and l.sd_conv3_packed_floats(1, 32, 1) == 288
    assert l.sd_conv3_packed_floats(32, 32, 2) == -1 and l.sd_conv3_packed_floats(1, 32, 3) == 864
    rs = np.random.RandomState(0)
    for ci, co, kz in ((32, 32, 1), (32, 64, 1), (64, 128, 1), (256, 32, 1), (32, 32, 3), (64, 64, 3)):
        w = rs.randn(*((co, ci) + ((3, 3, 3) if kz == 3 else (3, 3)))).astype(np.float32)
        n = l.sd_conv3_packed_floats(ci, co, kz)
        assert n == co * ci * 9 * kz + 4                       # + 16 bytes of zeros: the kernel's zero-padding source
        out = np.empty(n, np.float32)
        N.check(l.sd_conv3_pack_weights_host(N.ptr(w), ci, co, kz, N.ptr(out)))
        assert not out[-4:].any()
        o = out[:-4].reshape(co // 32, (ci // 32) * kz, 9, 4, 2, 32, 4)        # [group][unit = chunk*kz + z][tap][j][h][n][e]
        w5 = w.reshape(co, ci, kz, 9)
        for g in range(o.shape[0]):
            for u in range(o.shape[1]):
                c, z = divmod(u, kz)
                for j in range(4):
                    for h in range(2):
                        for e in range(4):
                            cin = c * 32 + h * 16 + j * 4 + e
                            want = w5[g * 32:(g + 1) * 32, cin, z].T       # [tap][n]
                            assert np.array_equal(o[g, u, :, j, h, :, e], want)
    w1 = rs.randn(32, 1, 3, 3, 3).astype(np.float32)
    out = np.empty(864, np.float32)
    N.check(l.sd_conv3_pack_weights_host(N.ptr(w1), 1, 32, 3, N.ptr(out)))
    assert np.array_equal(out.reshape(27, 32), w1.reshape(32, 27).T)
