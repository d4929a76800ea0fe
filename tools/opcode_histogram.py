"""tools/opcode_histogram.py — static opcode histogram of one kernel of the shipped library (gfx950 code object inside .hip_fatbin).
Usage: python tools/opcode_histogram.py <kernel-name-substring> [lib.so]   (a static count: loops are counted once)"""
import collections, os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"; MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
want = sys.argv[1]; so = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "stark_mlwe_amd", "libstark_mlwe_hip.so")
with tempfile.TemporaryDirectory() as td:
    fat = os.path.join(td, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", so, os.path.join(td, "discard.so")])
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)] + [len(blob)]
    for bi in range(len(starts) - 1):
        part = os.path.join(td, f"b{bi}.bin"); open(part, "wb").write(blob[starts[bi]:starts[bi + 1]]); co = os.path.join(td, f"b{bi}.co")
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={part}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0: continue
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
        cur = None; hist = {}
        for line in dis.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                cur = m.group(1); continue
            m = re.match(r"^\s+([a-z_0-9]+)\s", line + " ")
            if m and cur: hist.setdefault(cur, collections.Counter())[m.group(1)] += 1
        for name, h in hist.items():
            dm = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            if want in dm and not name.endswith(".kd"):
                tot = sum(h.values()); print(f"== {dm[:100]}  ({tot} instructions)")
                for op, c in h.most_common(28): print(f"  {op:28s} {c:7d}  {100.0 * c / tot:5.1f} %")
