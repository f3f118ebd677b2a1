"""include/mazu_amd.hpp -- the C++ mirror of the reference's kphf surface -- compiled and run as a
plain C++ program against libmazu_amd.so (no Python in the loop)."""
import os
import subprocess

import pytest

import helpers as H

EXE = os.path.join(H.ROOT, "tests", "cpp", "cxx_mirror_check")


@pytest.fixture(scope="module")
def exe():
    lib = os.path.join(H.ROOT, "mazu_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", os.path.join(H.ROOT, "tests", "cpp", "cxx_mirror_check.cpp"),
                           "-L" + lib, "-lmazu_amd", "-Wl,-rpath," + lib, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", EXE])
    return EXE


def run(exe):
    return subprocess.run([exe, os.path.join(H.GOLDEN, "cf", "tiny", "tiny")], capture_output=True, text=True, cwd=H.ROOT)


def test_cxx_mirror_host_side(exe):
    r = run(exe)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip() in ("nogpu", "gpu-ok")


@pytest.mark.gpu
def test_cxx_mirror_known_answers_on_gpu(exe):
    r = run(exe)
    assert r.returncode == 0 and r.stdout.strip() == "gpu-ok", r.stdout + r.stderr
