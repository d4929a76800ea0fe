"""NTT timings for the current STARK_NTT_LOG_E / STARK_NTT_MINW setting (run once per setting)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stark_mlwe_amd.api import Context, PALLAS_FR
dev = torch.device("cuda", 0)
ctx = Context(0); lib = ctx.lib
for lg in (16, 20, 23, 24):
    x = torch.empty((1 << lg, 4), dtype=torch.int64, device=dev)
    ctx._chk(lib.stark_synth_column_dev(ctx.h, 1, 7, 0, 1 << lg, C.c_void_p(x.data_ptr())))
    f = lambda: ctx._chk(lib.stark_ntt_dev(ctx.h, PALLAS_FR, C.c_void_p(x.data_ptr()), lg, 0, None))
    f(); ms = C.c_float(); ctx._chk(lib.stark_timer_start(ctx.h))
    for _ in range(5): f()
    ctx._chk(lib.stark_timer_stop_ms(ctx.h, C.byref(ms)))
    print(json.dumps({"log_e": os.environ.get("STARK_NTT_LOG_E", "11"), "minw": os.environ.get("STARK_NTT_MINW", "2"), "log_n": lg, "ms": ms.value / 5, "GBps": 64.0 * (1 << lg) / (ms.value / 5) / 1e6}), flush=True)
ctx.close()
