"""CPU: the product's scan-beam sweep (stardist_amd/csrc/clip_sweep*.h compiled for the host) against the
reference's vendored Clipper (oracle/_ref/libclipper_ref.so) on seeded random star-polygon pairs."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(refmods, tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("clip") / "clip_check")
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    cmd = ["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "host", "clip_check.cpp"), "-o", exe,
           "-L" + ref_dir, "-lclipper_ref", "-Wl,-rpath," + ref_dir]
    subprocess.run(cmd, check=True)
    return exe


@pytest.mark.parametrize("args", ["60000 32 10 0.1 2", "40000 32 10 0.9 4", "40000 32 3 0.5 6", "40000 32 2 0.9 7",
                                  "20000 11 10 0.3 8", "20000 64 20 0.3 9", "20000 32 40 0.2 11 16000", "10000 32 10 0.3 21 0 0 1"])
def test_sweep_equals_clipper(harness, args):
    r = subprocess.run([harness] + args.split(), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches=0" in r.stdout and "flagged=0" in r.stdout, r.stdout


# ---- bound-slot sweep (clip_beam.h, the layout the GPU pair kernel runs): host build vs clip_sweep.h AND vs the reference Clipper
@pytest.fixture(scope="module", params=["tier1", "tier2"])
def beam_harness(request, refmods, tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("beam") / ("beam_check_" + request.param))
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    defs = ["-DBEAM_K=8", "-DBEAM_MAXIL=6", "-DBEAM_MAXREC=4"] if request.param == "tier1" else ["-DBEAM_K=15", "-DBEAM_MAXIL=16", "-DBEAM_MAXREC=8"]
    cmd = ["g++", "-O2", "-std=c++17"] + defs + [os.path.join(ROOT, "tests", "host", "beam_check.cpp"), "-o", exe,
                                                  "-L" + ref_dir, "-lclipper_ref", "-Wl,-rpath," + ref_dir]
    subprocess.run(cmd, check=True)
    return exe


# args: n_pairs n_rays radius noise seed [offset verbose lds_mode]
@pytest.mark.parametrize("args", ["60000 32 10 0.1 2", "30000 32 10 0.5 4", "30000 32 3 0.5 6 0 0 1", "30000 32 2 0.9 7", "20000 11 10 0.3 8",
                                  "20000 5 1 0.5 3", "20000 32 40 0.2 11 16000", "20000 32 10 0.05 12 0 0 1"])
def test_beam_equals_sweep_and_clipper(beam_harness, args):
    """0 mismatches among the pairs the tier can hold; pairs that exceed its capacities are only flagged (the GPU sends them on)"""
    r = subprocess.run([beam_harness] + args.split(), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mism_vs_sweep=0" in r.stdout and "mism_vs_clipper=0" in r.stdout and "join_flag_mism=0" in r.stdout and "other_flagged=0" in r.stdout, r.stdout
