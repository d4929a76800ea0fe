"""A compiled C++ caller of the C-ABI (tests/capi_client.cpp), linked against libstark_mlwe_hip.so — the boundary exercised the way
the reference's FFI would use it, without Python in between.  CPU: it must build, link (every symbol it uses resolves) and report
"no device" loudly (exit code 3: no CPU fallback).  GPU: every check in it must pass."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "_build", "capi_client")


def _build():
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    lib = os.path.join(ROOT, "stark_mlwe_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "capi_client.cpp"),
                           "-L", lib, "-lstark_mlwe_hip", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-o", BIN])
    return BIN


def test_capi_client_builds_links_and_fails_loudly_without_a_device():
    import torch
    exe = _build()
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the gpu-marked test runs the client")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 3, out.stdout + out.stderr
    assert "STARK_ERR_HIP" in out.stdout


@pytest.mark.gpu
def test_capi_client_on_gpu():
    exe = _build()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "all checks passed" in out.stdout
