#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_r3_step.py tests/test_gpu_r2_verify.py -m gpu -q --durations=8 -k "wide or 128 or preset or verif or merkle" > gpurun_out/r3e_tests.txt 2>&1; rc=$?; echo "tests rc=$rc"; tail -22 gpurun_out/r3e_tests.txt
