"""NTT timings over tile sizes / occupancy hints (stark_ctx_set_option "ntt_log_tile", "ntt_min_waves"): one JSON line per setting and size."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stark_mlwe_amd.api import Context, PALLAS_FR
ctx = Context(0); lib = ctx.lib
for log_e in (9, 10, 11, 12):
    for minw in (2, 4):
        ctx.set_option("ntt_log_tile", log_e); ctx.set_option("ntt_min_waves", minw)
        for lg in (20, 23, 24):
            x = torch.empty((1 << lg, 4), dtype=torch.int64, device="cuda")
            ctx._chk(lib.stark_synth_column_dev(ctx.h, 1, 7, 0, 1 << lg, C.c_void_p(x.data_ptr())))
            ctx._chk(lib.stark_ntt_dev(ctx.h, PALLAS_FR, C.c_void_p(x.data_ptr()), lg, 0, None))
            ctx._chk(lib.stark_timer_start(ctx.h))
            for _ in range(5):
                ctx._chk(lib.stark_ntt_dev(ctx.h, PALLAS_FR, C.c_void_p(x.data_ptr()), lg, 0, None))
            ms = C.c_float(); ctx._chk(lib.stark_timer_stop_ms(ctx.h, C.byref(ms)))
            print(json.dumps({"log_tile": log_e, "min_waves": minw, "log_n": lg, "ms": ms.value / 5, "GBps": 64.0 * (1 << lg) / (ms.value / 5 * 1e-3) / 1e9}), flush=True)
ctx.close()
