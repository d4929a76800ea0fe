#!/bin/bash
# round-3 GPU session A: row-form chain microbenchmark + the new tests
set -o pipefail
mkdir -p gpurun_out
cd tools && ./bin/chain_row 3 > ../gpurun_out/chain_row_3.txt && python3 chain_row_check.py ../gpurun_out/chain_row_3.txt > ../gpurun_out/chain_row_check.txt 2>&1
./bin/chain_row 2000 > ../gpurun_out/chain_row_2000.txt && python3 chain_row_check.py ../gpurun_out/chain_row_2000.txt >> ../gpurun_out/chain_row_check.txt 2>&1
./bin/chain_bench > ../gpurun_out/chain_bench.txt 2>&1
cd ..
cat gpurun_out/chain_row_check.txt; tail -1 gpurun_out/chain_row_2000.txt; cat gpurun_out/chain_bench.txt
timeout -k 10 900 python -m pytest tests/test_gpu_r3_step.py tests/test_gpu_zz_dist.py -m gpu -x -q -k "not (bench_step and 20) and not mod16" > gpurun_out/r3a_tests.txt 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r3a_tests.txt
