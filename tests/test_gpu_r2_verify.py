"""N3 on the GPU: the library's verifier entry points (capi_verify.hip: fri_verify.hpp with every hash batched onto the prover's
kernels) against the oracle's restatement of deep_fri_verify / verify_many_ds / verify_pairs_ds, on proofs the GPU prover made.
Needs an MI355X: `pytest -m gpu`."""
import ctypes as C
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from stark_mlwe_amd.api import DeepFriParams, StarkError


@pytest.mark.parametrize("n0,sched,r", [(1 << 10, [16, 8], 8), (1 << 11, [16, 16, 8], 32), (1 << 9, [8, 4, 2], 5), (1 << 10, [32, 32], 40), (1 << 12, [64, 64], 6), (2, [2], 1)])
def test_gpu_verifier_accepts_gpu_proofs_and_agrees_with_oracle_under_tampering(gpu_ctx, oracle, n0, sched, r):
    cols = oracle.rand_fr_columns(2025 + n0, n0, 4)
    prm = DeepFriParams(sched, r, 0xDEEFBAAD)
    proof, _, _ = gpu_ctx.deep_fri_prove(cols[0], cols[1], cols[2], cols[3], n0, prm)
    assert gpu_ctx.deep_fri_verify(prm, proof) is True
    assert oracle.deep_fri_verify(proof, sched, r, 0xDEEFBAAD) == 1
    assert gpu_ctx.deep_fri_verify(DeepFriParams(sched, r + 1, 0xDEEFBAAD), proof) is False
    assert gpu_ctx.deep_fri_verify(DeepFriParams(sched[:-1], r, 0xDEEFBAAD), proof) is False
    for bad in (b"", proof[:-1], proof + b"\0"):
        assert gpu_ctx.deep_fri_verify(prm, bad) is False
    rng = random.Random(n0 * 31 + r)
    positions = sorted(set([8, 8 + 31, 41, len(proof) - 1, len(proof) - 41] + [rng.randrange(len(proof)) for _ in range(40)]))
    rejected = 0
    for pos in positions:
        bad = bytearray(proof); bad[pos] ^= 1 << rng.randrange(8)
        want = oracle.deep_fri_verify(bytes(bad), sched, r, 0xDEEFBAAD) == 1
        got = gpu_ctx.deep_fri_verify(prm, bytes(bad))
        assert got == want, f"byte {pos}: library {got}, oracle {want}"
        rejected += not got
    assert rejected >= len(positions) // 2


def test_gpu_verifier_at_bench_size(gpu_ctx, oracle):
    """A 2^20-row proof (r = 32, [16,16,8]) made on the GPU from a synthetic f0: accepted by the library's verifier and by the
    oracle's; one flipped bit in the middle is rejected by both."""
    import torch
    n0, sched, r = 1 << 20, [16, 16, 8], 32
    f0 = torch.empty((n0, 4), dtype=torch.int64, device="cuda")
    gpu_ctx._chk(gpu_ctx.lib.stark_synth_column_dev(gpu_ctx.h, 0x5EED0014, 5, 0, n0, C.c_void_p(f0.data_ptr())))
    sch = np.ascontiguousarray(sched, dtype=np.uint64); h = C.c_void_p()
    gpu_ctx._chk(gpu_ctx.lib.stark_deep_fri_prove_dev(gpu_ctx.h, None, None, None, None, C.c_void_p(f0.data_ptr()), n0, sch.ctypes.data_as(C.c_void_p), 3, r, 0xDEEFBAAD, C.byref(h)))
    proof, _ = gpu_ctx._proof_out(h)
    prm = DeepFriParams(sched, r, 0xDEEFBAAD)
    assert gpu_ctx.deep_fri_verify(prm, proof) is True and oracle.deep_fri_verify(proof, sched, r, 0xDEEFBAAD) == 1
    bad = bytearray(proof); bad[len(bad) // 2] ^= 0x10
    assert gpu_ctx.deep_fri_verify(prm, bytes(bad)) is False and oracle.deep_fri_verify(bytes(bad), sched, r, 0xDEEFBAAD) == 0


@pytest.mark.parametrize("arity,n,label", [(16, 4096, 0), (16, 55, 9), (8, 19, 3), (2, 8, 1), (4, 64, 7), (32, 1024, 4), (128, 300, 2)])
def test_merkle_commit_open_verify_roundtrip_on_gpu(gpu_ctx, oracle, arity, n, label):
    """merkle/src/lib.rs:1053-1136: commit -> open_many -> verify_single is true; tampering a leaf / the root / the label / the
    proof bytes / the index set makes it false (MerkleProver facade through the C-ABI)."""
    leaves = oracle.synth_column(61, arity, 0, n)
    cfg = gpu_ctx.merkle_cfg(arity, label)
    t = gpu_ctx.merkle_new(leaves, cfg)
    rng = random.Random(n + arity); idx = sorted(set(rng.randrange(n) for _ in range(9)))
    pr = t.open_many(idx); root = t.root(); t.free()
    vals = leaves[idx]
    assert gpu_ctx.merkle_verify_single(cfg, root, idx, vals, pr) is True
    bad = vals.copy(); bad[-1, 3] ^= np.uint64(1)
    assert gpu_ctx.merkle_verify_single(cfg, root, idx, bad, pr) is False
    assert gpu_ctx.merkle_verify_single(cfg.with_tree_label(label + 1), root, idx, vals, pr) is False
    r2 = root.copy(); r2[0] ^= np.uint64(1)
    assert gpu_ctx.merkle_verify_single(cfg, r2, idx, vals, pr) is False
    assert gpu_ctx.merkle_verify_single(cfg, root, idx[1:], vals[1:], pr) is False
    b = bytearray(pr); b[len(pr) - 9] ^= 1                          # last group-size byte / arity word region
    assert gpu_ctx.merkle_verify_single(cfg, root, idx, vals, bytes(b)) is False
    with pytest.raises(StarkError):
        gpu_ctx.merkle_verify_single(gpu_ctx.merkle_cfg(arity, label).__class__(129, None, 0), root, idx, vals, pr)     # MerkleChannelCfg::new(129) panics


@pytest.mark.parametrize("arity,n", [(2, 8), (2, 2), (16, 64), (8, 32)])
def test_merkle_pairs_commit_open_verify_roundtrip_on_gpu(gpu_ctx, oracle, arity, n):
    """merkle/src/lib.rs:1138-1168 (commit_pairs / open_pairs / verify_pairs)."""
    f = oracle.synth_column(62, 0, 0, n); cp = oracle.synth_column(62, 1, 0, n)
    cfg = gpu_ctx.merkle_cfg(arity, 5)
    t = gpu_ctx.merkle_new_pairs(f, cp, cfg)
    idx = sorted({0, n - 1, n // 2})
    pr = t.open_many(idx); root = t.root(); t.free()
    assert gpu_ctx.merkle_verify_pairs(cfg, root, idx, f[idx], cp[idx], pr) is True
    bad = f[idx].copy(); bad[0, 0] ^= np.uint64(4)
    assert gpu_ctx.merkle_verify_pairs(cfg, root, idx, bad, cp[idx], pr) is False
    assert gpu_ctx.merkle_verify_pairs(cfg.with_tree_label(6), root, idx, f[idx], cp[idx], pr) is False
    assert gpu_ctx.merkle_verify_single(cfg, root, idx, f[idx], pr) is False
