"""tools/prove_trace.py — runs deep_fri_prove given f0 at 2^K a few times (for `rocprofv3 --kernel-trace`: which launches make up the tail of a prove).
Usage: prove_trace.py [K | mfK] [reps];  tools/prove_trace.py summarize <kernel_trace.csv> prints the launches of the last prove with durations and gaps."""
import sys, os
if len(sys.argv) > 2 and sys.argv[1] == "summarize":
    import csv
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last prove = the launches after the last k_synth (a marker launch between proves)
    last = max(i for i, r in enumerate(rows) if "k_synth" in r["Kernel_Name"])
    seq = rows[last + 1:]
    t0 = int(seq[0]["Start_Timestamp"]); prev_end = t0; busy = 0
    for r in seq:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0) / 1e3:9.1f} us  +gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f}  grid {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>8}  {r['Kernel_Name'][:70]}")
        busy += e - s; prev_end = max(prev_end, e)
    print(f"total {(prev_end - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, launches {len(seq)}")
    sys.exit(0)
import ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stark_mlwe_amd.api import Context
import bench
MF = len(sys.argv) > 1 and sys.argv[1].startswith("mf")      # mfK: the Merkle-folded sum-check prove_mf at 2^K entries instead
K = int(sys.argv[1][2:] if MF else sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
ctx = Context(0, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)); lib = ctx.lib
n0 = 1 << K
f0 = torch.empty((n0, 4), dtype=torch.int64, device=dev)
sched = (C.c_size_t * len(bench.SCHEDULE))(*bench.SCHEDULE)
for i in range(reps):
    ctx._chk(lib.stark_synth_column_dev(ctx.h, 5, 0, 0, n0, C.c_void_p(f0.data_ptr())))      # the marker between proves
    if MF:
        import numpy as np
        w = f0.cpu().numpy().view(np.uint64)
        ctx._chk(lib.stark_synth_column_dev(ctx.h, 5, 0, 0, n0, C.c_void_p(f0.data_ptr())))      # marker again: the upload above is not part of the prove
        ctx.prove_mf(K, 2025, 2, w); continue
    ph = C.c_void_p()
    ctx._chk(lib.stark_deep_fri_prove_dev(ctx.h, None, None, None, None, C.c_void_p(f0.data_ptr()), n0, sched, len(bench.SCHEDULE), 32, bench.SEED_Z, C.byref(ph)))
    lib.stark_proof_free(ph)
ctx.sync(); ctx.close()
