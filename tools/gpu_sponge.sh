#!/bin/bash
# quick sponge check: digests equal to the one-wave kernel's, timings, breakdown
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/sponge_timing.py 2>&1 | grep -v amdgpu.ids > gpurun_out/sponge_timing.jsonl; echo "timing rc=$?"; cat gpurun_out/sponge_timing.jsonl
timeout -k 10 300 python tools/sponge_debug_timing.py 2>&1 | grep -v amdgpu.ids > gpurun_out/sponge_debug.jsonl; cat gpurun_out/sponge_debug.jsonl
