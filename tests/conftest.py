import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")
    # Build the oracle and (if missing) the product libraries once per session.  Both are plain
    # compiles: g++ for the oracle, hipcc cross-compiling gfx950 for the product (no GPU needed).
    if not os.path.exists(os.path.join(ROOT, "oracle", "_build", "liboracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-j4"])
    pkg = os.path.join(ROOT, "stark_mlwe_amd")
    if not (os.path.exists(os.path.join(pkg, "libstark_mlwe_hip.so")) and os.path.exists(os.path.join(pkg, "libstark_mlwe_hostcheck.so"))):
        subprocess.check_call(["make", "-C", os.path.join(pkg, "csrc"), "-j4"])


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.Oracle()


@pytest.fixture(scope="session")
def hostcheck():
    import hostcheck_lib
    return hostcheck_lib.HostCheck()


@pytest.fixture(scope="session")
def gpu_ctx():
    from stark_mlwe_amd.api import Context
    ctx = Context(0)
    yield ctx
    ctx.close()
