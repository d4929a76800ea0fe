#!/bin/bash
# the whole GPU suite, then the small-launch latencies (five-wave kernels against the one-wave / wave-pair ones)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q 2>&1 | tail -6 && timeout -k 10 300 python tools/latency_timing.py 2>&1 | grep one_wave > gpurun_out/latency_timing.jsonl; cat gpurun_out/latency_timing.jsonl
