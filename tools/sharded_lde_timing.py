"""tools/sharded_lde_timing.py — compute side of the sharded LDE: W virtual ranks emulated on ONE GPU (stark_diag_lde_sharded_emulated_dev: the five local
phases of every rank and the exchanges as device copies) against the single-GPU LDE of the same column.  (time of all W ranks' work) / (single-GPU time) is
the work the sharding ADDS (pack kernels, the extra passes of the six-step form); the xGMI exchanges themselves are not in it.  Not product code."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stark_mlwe_amd.api import Context, _ptr
import bench
dev = torch.device("cuda", 0)
ctx = Context(0, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)); lib = ctx.lib
P = lambda t: C.c_void_p(t.data_ptr())
def timed(fn, reps=3):
    fn(); ms = C.c_float(); ctx._chk(lib.stark_timer_start(ctx.h))
    for _ in range(reps): fn()
    ctx._chk(lib.stark_timer_stop_ms(ctx.h, C.byref(ms))); return ms.value / reps
shift = bench._mont_small(5)
for log_n, lb in ((20, 3), (23, 3)):
    x = torch.empty((1 << log_n, 4), dtype=torch.int64, device=dev); y = torch.empty((1 << (log_n + lb), 4), dtype=torch.int64, device=dev)
    ctx._chk(lib.stark_synth_column_dev(ctx.h, 9, 0, 0, 1 << log_n, P(x)))
    one = timed(lambda: ctx._chk(lib.stark_lde_dev(ctx.h, 0, P(x), log_n, lb, _ptr(shift), P(y))))
    row = {"log_n": log_n, "log_blowup": lb, "single_gpu_lde_ms": round(one, 3)}
    for W in (2, 4, 8):
        t = timed(lambda: ctx._chk(lib.stark_diag_lde_sharded_emulated_dev(ctx.h, 0, W, P(x), log_n, lb, _ptr(shift), P(y))))
        row[f"emulated_W{W}_all_ranks_ms"] = round(t, 3); row[f"W{W}_work_ratio"] = round(t / one, 2)
    print(json.dumps(row), flush=True)
    del x, y; ctx.trim()
ctx.close()
