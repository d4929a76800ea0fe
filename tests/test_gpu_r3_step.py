"""Round-3 parity tests: the bench step itself (BASELINE.json configs[1]) at its own size against roots the CPU oracle produced
ONCE (tools/gen_golden.py step:K -> tests/golden/step_roots_kK.json).  Needs an MI355X: `pytest -m gpu`."""
import ctypes as C
import json
import os

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.mark.parametrize("k", [10, 20])
def test_bench_step_roots_match_oracle_golden(gpu_ctx, k):
    """Exactly `bench.py`'s N = 1 step (`bench.make_single_gpu_step`: LDE of the four synthetic columns 2^k -> 2^(k+3) on the coset
    5*<w>, DEEP-ALI merge at z = 0xC0FFEE, fri_build_transcript [16,16,8] — crates/deep_ali/src/lib.rs:48-105,
    crates/deep_ali/src/fri.rs:231-312): the four layer roots equal the oracle's, at 2^10 rows and at the bench size 2^20 rows x 8."""
    import torch
    import bench
    path = os.path.join(GOLD, f"step_roots_k{k}.json")
    assert os.path.exists(path), f"{path} missing (tools/gen_golden.py step:{k})"
    gold = json.load(open(path))
    seed = 0x5EED0000 + k
    assert bench.golden_step_roots(k, seed) == gold["roots"], "bench.py's step parameters differ from the golden's"
    dev = torch.device("cuda", 0)
    cols = [torch.empty((1 << k, 4), dtype=torch.int64, device=dev) for _ in range(4)]
    for c in range(4):
        gpu_ctx._chk(gpu_ctx.lib.stark_synth_column_dev(gpu_ctx.h, seed, c, 0, 1 << k, C.c_void_p(cols[c].data_ptr())))
    step = bench.make_single_gpu_step(gpu_ctx, cols, k, dev)
    assert bench.roots_hex(step()) == gold["roots"]
    assert bench.roots_hex(step()) == gold["roots"]          # a second pass over the same buffers (what the timed loop does)
