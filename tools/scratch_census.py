#!/usr/bin/env python3
"""tools/scratch_census.py — where the spilled registers of the two Poseidon throughput kernels are touched.
Compiles capi_core.hip to gfx950 assembly (same flags as the library) and counts scratch_load / scratch_store instructions per loop
nesting depth (LLVM's "Loop Header: Depth=" / "in Loop ... Depth=" / "Parent Loop ... Depth=" block comments) next to the kernel's total
instruction count.  A spill that sits outside the round loops is executed a handful of times per permutation."""
import os, re, subprocess, sys, tempfile, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "stark_mlwe_amd", "csrc", "capi_core.hip")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "core.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-w", "-o", out, src])
    text = open(out).read()
res = []
for sym, name in [("_ZN5stark12k_leaf_pair2", "k_leaf_pair2"), ("_ZN5stark10k_hash_ds2ILi17EE", "k_hash_ds2<17>")]:
    m = re.search(r"^%s\w*:[^\n]*\n(.*?)s_endpgm" % re.escape(sym), text, re.S | re.M)
    if not m:
        continue
    depth, per_depth, total = 0, {}, 0
    for line in m.group(1).split("\n"):
        lm = re.match(r"^\.LBB\d+_\d+:\s*;(.*)$", line) or re.match(r"^; %bb\.\d+:\s*;(.*)$", line)
        if re.match(r"^\.LBB\d+_\d+:\s*$", line):
            depth = 0
        if lm:
            d = re.search(r"Depth=(\d+)", lm.group(1)); depth = int(d.group(1)) if d else 0
            if "Parent Loop" in lm.group(1):      # the innermost depth follows on a continuation line; take the deepest seen
                depth = depth + 1
        if re.match(r"^\s+(v_|s_|ds_|global_|buffer_|flat_|scratch_)", line):
            total += 1
            if "scratch_" in line:
                per_depth[depth] = per_depth.get(depth, 0) + 1
    res.append({"kernel": name, "instructions_in_code_object": total, "scratch_instructions": sum(per_depth.values()), "scratch_by_loop_depth": {str(k): v for k, v in sorted(per_depth.items())}})
print(json.dumps(res, indent=1))
