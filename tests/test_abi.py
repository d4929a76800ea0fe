"""The C-ABI library: loads without a GPU, exports every symbol include/stark_mlwe.h declares, and
refuses to run without a device (no CPU fallback).  CPU only."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "stark_mlwe.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(stark_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from stark_mlwe_amd._abi import SIGNATURES, load_library
    lib = load_library()
    syms = header_symbols()
    assert len(syms) >= 60
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/stark_mlwe.h but not exported"
    assert sorted(SIGNATURES) == syms, "ctypes table and header disagree"


def test_no_device_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from stark_mlwe_amd._abi import load_library
    from stark_mlwe_amd.api import Context, StarkError
    lib = load_library()
    h = C.c_void_p()
    assert lib.stark_ctx_create(0, None, C.byref(h)) == -2        # STARK_ERR_HIP
    with pytest.raises(StarkError):
        Context(0)


def test_product_does_not_link_or_import_the_oracle():
    import subprocess
    so = os.path.join(ROOT, "stark_mlwe_amd", "libstark_mlwe_hip.so")
    needed = subprocess.check_output(["readelf", "-d", so], text=True)
    assert "oracle" not in needed
    for dirpath, _, files in os.walk(os.path.join(ROOT, "stark_mlwe_amd")):
        for fn in files:
            if fn.endswith((".py", ".hpp", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "liboracle" not in src and "oracle_lib" not in src and '#include "../../oracle' not in src, fn
