set -o pipefail; mkdir -p gpurun_out; export TMPDIR=/tmp; OUT=$PWD/gpurun_out
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ptrace -- python3 $OUT/../tools/prove_trace.py 16 3 > $OUT/ptrace.log 2>&1; echo "exit $?")
f=$(find /tmp/ptrace -name "*kernel_trace.csv" | head -1); cp "$f" $OUT/prove_trace_k16.csv
python tools/prove_trace.py summarize $OUT/prove_trace_k16.csv > $OUT/prove_trace_k16.txt; tail -80 $OUT/prove_trace_k16.txt
