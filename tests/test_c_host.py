"""The boundary from COMPILED code: include/openintel_hip.h is valid C99 (the reference's FFI binds a C ABI, not C++), a plain-C
host links against the library, and -- without a GPU -- gets the loud no-device error; on the GPU box the same program scores
the reference's lexicon sentences and runs a hybrid query through the C ABI with no Python in the process."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "integration", "c", "abi_check.c")


def _build(tmp_path):
    from openintel_amd import build
    build.build()
    exe = str(tmp_path / "abi_check")
    libdir = os.path.join(ROOT, "openintel_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), SRC, "-L", libdir,
                    "-lopenintel_hip", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True, capture_output=True, text=True)
    return exe


def test_header_is_valid_c99_and_cxx():
    hdr = os.path.join(ROOT, "include", "openintel_hip.h")
    for cc, std in (("gcc", "-std=c99"), ("g++", "-std=c++11")):
        r = subprocess.run([cc, std, "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-x", "c" if cc == "gcc" else "c++", hdr],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_plain_c_host_links_and_reports_no_device_loudly(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the gpu-marked test runs the same program for real")
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "no gfx950 device" in r.stdout and "link and header ok" in r.stdout


@pytest.mark.gpu
def test_plain_c_host_runs_the_paths_on_the_gpu(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "abi_check: ok" in r.stdout, (r.stdout, r.stderr)
