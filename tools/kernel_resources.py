#!/usr/bin/env python3
"""tools/kernel_resources.py — per-kernel resource table of the SHIPPED library, read from the code objects inside
stark_mlwe_amd/libstark_mlwe_hip.so (not from a separate compile): VGPRs, AGPRs, SGPRs, spilled VGPRs / SGPRs, scratch (private
segment) bytes, static LDS bytes, max workgroup size.

How: `llvm-objcopy --dump-section .hip_fatbin`, split the section into its clang offload bundles, `clang-offload-bundler --unbundle`
the gfx950 code object of each, `llvm-readelf --notes` -> the AMDGPU metadata (amdhsa.kernels).

    python tools/kernel_resources.py [--so PATH] [--csv profiles/r03_kernel_resources.csv]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FIELDS = [("vgpr_count", "vgpr"), ("agpr_count", "agpr"), ("sgpr_count", "sgpr"), ("vgpr_spill_count", "vgpr_spill"), ("sgpr_spill_count", "sgpr_spill"),
          ("private_segment_fixed_size", "scratch_bytes"), ("group_segment_fixed_size", "static_lds_bytes"), ("max_flat_workgroup_size", "max_wg")]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def kernels_of(so):
    rows = []
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", so, os.path.join(td, "discard.so")])
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)] + [len(blob)]
        for bi in range(len(starts) - 1):
            part = os.path.join(td, f"bundle{bi}.bin")
            open(part, "wb").write(blob[starts[bi]:starts[bi + 1]])
            co = os.path.join(td, f"bundle{bi}.co")
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={part}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"],
                               capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            cur = None
            for line in notes.split("\n"):
                m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip().strip("'")
                if k == "agpr_count":            # first per-kernel key in the metadata block of a kernel (alphabetical order)
                    cur = {"agpr_count": v}; rows.append(cur)
                elif cur is not None and k in ("name", "symbol") or (cur is not None and any(k == f for f, _ in FIELDS)):
                    cur[k] = v
    names = [r.get("name", "?") for r in rows]
    dm = demangle(names)
    for r in rows:
        r["kernel"] = re.sub(r"^void ", "", dm.get(r.get("name", "?"), r.get("name", "?")))
        r["kernel"] = re.sub(r"\(.*$", "", r["kernel"])
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--so", default=os.path.join(ROOT, "stark_mlwe_amd", "libstark_mlwe_hip.so"))
    ap.add_argument("--csv", default=None)
    ap.add_argument("--only-spilling", action="store_true")
    a = ap.parse_args()
    rows = kernels_of(a.so)
    rows.sort(key=lambda r: r["kernel"])
    hdr = ["kernel"] + [h for _, h in FIELDS]
    lines = [",".join(hdr)]
    for r in rows:
        if a.only_spilling and int(r.get("vgpr_spill_count", 0)) == 0 and int(r.get("sgpr_spill_count", 0)) == 0:
            continue
        lines.append(",".join(['"%s"' % r["kernel"]] + [str(r.get(f, "")) for f, _ in FIELDS]))
    text = "\n".join(lines) + "\n"
    if a.csv:
        open(a.csv, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
