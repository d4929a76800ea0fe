#!/bin/bash
# tools/host_sanitize.sh — the host-side unit tests against an AddressSanitizer + UBSan build of the product's shared inline code (CPU only:
# GPU sanitizers are not available on this pool).  Builds stark_mlwe_amd/csrc/hostcheck.cpp into tests/_build/ and points the tests at it.
set -e
cd "$(dirname "$0")/.."
mkdir -p tests/_build
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-sanitize-recover=undefined -Wno-unused-function -Wno-misleading-indentation \
    -o tests/_build/libstark_mlwe_hostcheck_san.so stark_mlwe_amd/csrc/hostcheck.cpp
STARK_HOSTCHECK_LIB=$PWD/tests/_build/libstark_mlwe_hostcheck_san.so ASAN_OPTIONS=detect_leaks=0 \
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) python -m pytest tests/test_hostcheck.py -q "$@"
