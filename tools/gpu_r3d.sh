#!/bin/bash
# round-3 GPU session D: the whole gpu suite, then the bench line
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r3d_tests.txt 2>&1; rc=$?; echo "tests rc=$rc"; tail -30 gpurun_out/r3d_tests.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --csv gpurun_out/r3d_benchmarkdata.csv > gpurun_out/r3d_bench.json 2> gpurun_out/r3d_bench.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/r3d_bench.json; tail -3 gpurun_out/r3d_bench.err
