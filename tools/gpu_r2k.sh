#!/bin/bash
# tools/gpu_r2k.sh — NTT tile/occupancy sweep of the nine-limb engine (timing only; digests compared by eye)
set -o pipefail
mkdir -p gpurun_out; cd "${GRAFT_REPO_ROOT:-.}"; OUT=$PWD/gpurun_out; R=${1:-r2k}
run() { timeout -k 10 120 python tools/variant_bench.py "$1" "$2" "$3" ntt >> $OUT/variants_$R.jsonl || exit 1; }
run stark_mlwe_amd/libstark_mlwe_hip.so base ""
run tools/bin/libvar_s128.so s128_tile10 "ntt_log_tile=10"
run stark_mlwe_amd/libstark_mlwe_hip.so t256_tile10 "ntt_log_tile=10"
run stark_mlwe_amd/libstark_mlwe_hip.so minw4_tile10 "ntt_log_tile=10,ntt_min_waves=4"
run stark_mlwe_amd/libstark_mlwe_hip.so tile12 "ntt_log_tile=12"
run stark_mlwe_amd/libstark_mlwe_hip.so nodirect "ntt_direct_max_log=0"
run tools/bin/libvar_s128.so s128_tile9 "ntt_log_tile=9"
python - "$R" <<'PY'
import json
for l in open("gpurun_out/variants_%s.jsonl" % "R".replace("R", __import__("sys").argv[1] if len(__import__("sys").argv) > 1 else "r2k")):
    d = json.loads(l); print(d["variant"], d["options"], round(d["ntt_2^23_coset_ms"], 4), d["ntt_digest"])
PY
