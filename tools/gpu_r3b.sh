#!/bin/bash
# round-3 GPU session B: the three-wave sponge — parity first, then timing
set -o pipefail
mkdir -p gpurun_out
tools/bin/swap_probe > gpurun_out/swap_probe.txt 2>&1; head -c 600 gpurun_out/swap_probe.txt; echo
timeout -k 10 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3b_smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r3b_smoke.txt
timeout -k 10 300 python tools/sponge_timing.py > gpurun_out/r3b_sponge_timing.jsonl 2>&1; echo "timing rc=$?"; tail -4 gpurun_out/r3b_sponge_timing.jsonl
