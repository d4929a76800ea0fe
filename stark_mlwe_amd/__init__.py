"""stark_mlwe_amd — MI355X-native proving hot path of saholmes/stark-mlwe.

The product is `libstark_mlwe_hip.so` (hand-written HIP kernels + C++ host orchestration behind the
C-ABI of include/stark_mlwe.h).  This package is the thin Python host side used by the tests and by
bench.py: ctypes bindings (`_abi`) and a mirror of the reference's operator interface (`api`).
There is no CPU compute path here: every operation goes through the C-ABI and fails loudly when the
library or a HIP device is missing.
"""
from ._abi import StarkError, lib_path, load_library  # noqa: F401
