"""Round-2 parity tests: the BASELINE configs at their stated sizes, pinned to data the reference holds or to
goldens generated ONCE by the CPU oracle (tools/gen_golden.py), plus the boundary rules added this round
(stream ordering, parameter-handle hygiene).  Needs an MI355X: `pytest -m gpu`."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from stark_mlwe_amd.api import PALLAS_FR, DeepFriParams, StarkError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
SCHED, SEED_Z = [16, 16, 8], 0xDEEFBAAD


def _dev_cols(ctx, seed, n0, ncols=4):
    import torch
    cols = [torch.empty((n0, 4), dtype=torch.int64, device="cuda") for _ in range(ncols)]
    for c in range(ncols):
        ctx._chk(ctx.lib.stark_synth_column_dev(ctx.h, seed, c, 0, n0, C.c_void_p(cols[c].data_ptr())))
    return cols


# ---- configs[2] and the reference's own bench shape, END TO END from (a, s, e, t) -----------------------------------
@pytest.mark.parametrize("k,r", [(16, 32), (20, 32), (22, 40)])
def test_end_to_end_prove_matches_oracle_golden(gpu_ctx, k, r):
    """`stark_deep_fri_prove_dev` from the four trace columns (DeepAliRealBuilder: the serial column sponges, the
    FS challenges, the merge, folds, leaf hashes, trees, queries, encoding) at 2^16 / 2^20 and at BASELINE
    configs[2] (2^22 rows, 40 queries): sha256 of the proof bytes equals the golden the CPU oracle produced for the same
    synthetic trace (tests/golden/proof_k*_r*.json, written once by tools/gen_golden.py)."""
    path = os.path.join(GOLD, f"proof_k{k}_r{r}.json")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated yet (tools/gen_golden.py {k}:{r})")
    gold = json.load(open(path))
    assert gold["log_n0"] == k and gold["r"] == r and gold["schedule"] == SCHED and gold["seed_z"] == SEED_Z
    n0 = 1 << k
    cols = _dev_cols(gpu_ctx, gold["synth_seed"], n0)
    sch = np.ascontiguousarray(SCHED, dtype=np.uint64); h = C.c_void_p()
    gpu_ctx._chk(gpu_ctx.lib.stark_deep_fri_prove_dev(gpu_ctx.h, *[C.c_void_p(c.data_ptr()) for c in cols], None, n0,
                                                      sch.ctypes.data_as(C.c_void_p), 3, r, SEED_Z, C.byref(h)))
    proof, est = gpu_ctx._proof_out(h)
    assert len(proof) == gold["proof_len"] and est == gold["size_estimate"]
    assert hashlib.sha256(proof).hexdigest() == gold["sha256"]


def test_end_to_end_prove_2pow24_matches_oracle_golden(gpu_ctx):
    """(In the default `-m gpu` selection since round 3: about 2.5 GPU-minutes, almost all of it the serial column sponges.)
    north_star's target size on ONE GPU: `stark_deep_fri_prove_dev` of a 2^24-row trace (r = 40, [16,16,8]) from (a, s, e, t), proof
    bytes against the golden the CPU oracle produced (tests/golden/proof_k24_r40.json, tools/gen_golden.py 24:40 — about two hours of 8 cores)."""
    path = os.path.join(GOLD, "proof_k24_r40.json")
    if not os.path.exists(path):
        pytest.skip("golden for 2^24 not generated")
    gold = json.load(open(path))
    n0 = 1 << 24
    cols = _dev_cols(gpu_ctx, gold["synth_seed"], n0)
    sch = np.ascontiguousarray(SCHED, dtype=np.uint64); h = C.c_void_p()
    gpu_ctx._chk(gpu_ctx.lib.stark_deep_fri_prove_dev(gpu_ctx.h, *[C.c_void_p(c.data_ptr()) for c in cols], None, n0, sch.ctypes.data_as(C.c_void_p), 3, 40, SEED_Z, C.byref(h)))
    ms = [gpu_ctx.lib.stark_proof_stage_ms(h, i) for i in range(3)]
    proof, est = gpu_ctx._proof_out(h)
    rec = {"log_n0": 24, "r": 40, "proof_len": len(proof), "size_estimate": est, "sha256": hashlib.sha256(proof).hexdigest(), "golden_sha256": gold["sha256"],
           "build_f0_ms": ms[0], "fri_build_ms": ms[1], "queries_encode_ms": ms[2], "matches_golden": hashlib.sha256(proof).hexdigest() == gold["sha256"]}
    out = os.path.join(ROOT, "gpurun_out"); os.makedirs(out, exist_ok=True)
    json.dump(rec, open(os.path.join(out, "prove_2pow24_end_to_end.json"), "w"), indent=1)
    assert len(proof) == gold["proof_len"] and est == gold["size_estimate"] and rec["matches_golden"]


@pytest.mark.parametrize("k,want_est,want_len", [(12, 52000, 55633), (13, 60968, 64844), (14, 72936, 76973), (15, 87736, 91998), (16, 101976, 106420)])
def test_published_fingerprints_k12_to_k16_on_gpu(gpu_ctx, oracle, k, want_est, want_len):
    """The reference's own bench inputs (end_to_end.rs:214, 229-253: seed chain from 1337, one LCG step per (preset, k),
    "paper" first, k from 11; a, s, e, t drawn from one StdRng) through the GPU prover: deep_fri_proof_size_bytes must be the
    value the reference PUBLISHED (crates/channel/benchmarkdata.csv:3-7); the encoded length equals the oracle's
    (tests/golden/oracle_fingerprint_k11_k18.txt)."""
    seed = 1337
    for _ in range(k - 10):
        seed = (seed * 1103515245 + 12345) % 2**64
    cols = oracle.rand_fr_columns(seed, 1 << k, 4)
    got, est, _ = gpu_ctx.deep_fri_prove(cols[0], cols[1], cols[2], cols[3], 1 << k, DeepFriParams(SCHED, 32, SEED_Z))
    assert est == want_est
    assert len(got) == want_len
    if k <= 13:
        assert oracle.deep_fri_verify(got, SCHED, 32, SEED_Z) == 1


# ---- configs[1]: LDE 2^20 -> 2^23 at the bench size --------------------------------------------------------------------
def test_lde_2pow20_to_2pow23_against_horner(gpu_ctx, oracle):
    """`stark_lde` at the bench shape.  (i) shift 1: the extension restricted to the original domain is the input
    (out[::8] == evals).  (ii) shift 5: 1024 sampled outputs equal the Horner evaluation (the O(n) definition,
    oracle_poly_eval_many) of the interpolating polynomial — whose 2^20 coefficients come from the ORACLE's inverse NTT —
    at 5*w_N^i.  (iii) the 2^23 coset NTT kernel the bench rooflines, on its own: sampled outputs of the coset transform of
    a full 2^23 coefficient vector against Horner."""
    lg, lb = 20, 3
    n, N = 1 << lg, 1 << (lg + lb)
    ev = oracle.synth_column(0x5EED0000 + lg, 0, 0, n)
    out1 = gpu_ctx.lde(ev, lb, field=PALLAS_FR)
    assert (out1[::8] == ev).all()
    g = oracle.from_u64(5)
    out5 = gpu_ctx.lde(ev, lb, field=PALLAS_FR, coset=g)
    coeffs = oracle.ntt(0, ev, inverse=True)
    rng = np.random.default_rng(20)
    idx = np.unique(np.concatenate([[0, 1, N - 1, N // 2, N // 8], rng.integers(0, N, 1024)]))
    wN = oracle.root_of_unity(lg + lb)
    pts = np.stack([oracle.mul(g, oracle.pow(wN, int(i))) for i in idx])
    assert (out5[idx] == oracle.poly_eval_many(0, coeffs, pts)).all()
    # unit-coset extension at the same sampled points (no shift)
    pts1 = np.stack([oracle.pow(wN, int(i)) for i in idx])
    assert (out1[idx] == oracle.poly_eval_many(0, coeffs, pts1)).all()
    # (iii) full-length 2^23 coset NTT
    x = oracle.synth_column(23, 7, 0, N)
    y = gpu_ctx.fft(x, field=PALLAS_FR, coset=g)
    sub = idx[:: max(1, len(idx) // 128)]
    pts_s = np.stack([oracle.mul(g, oracle.pow(wN, int(i))) for i in sub])
    assert (y[sub] == oracle.poly_eval_many(0, x, pts_s)).all()
    assert (gpu_ctx.ifft(y, field=PALLAS_FR, coset=g) == x).all()


# ---- configs[3] on one rank: the six-step decomposition at 2^24 ---------------------------------------------------------
def test_six_step_2pow24_one_rank_sampled_dft_rows(gpu_ctx, oracle):
    """The six-step path (column NTTs + twiddle, transpose, row NTTs) at 2^24 on ONE rank: equal to the plain three-pass NTT
    of the same vector (two independent decompositions) and, at sampled indices, to the DFT row sum_j x_j w^(jk) computed by
    Horner on the CPU (the definition)."""
    import torch
    from stark_mlwe_amd import dist as sd
    lg = 24; n = 1 << lg
    x = oracle.synth_column(0x5EED0000 + lg, 7, 0, n)
    plan = sd.DistNtt(sd.HipProvider(gpu_ctx), lg, 10)
    slab = torch.from_numpy(x[plan.local_input_indices().reshape(-1).numpy()].view(np.int64).copy()).cuda()
    rows = plan.forward(slab)
    nat = plan.to_natural_blocks(rows)
    gpu_ctx.sync(); torch.cuda.synchronize()
    nat = nat.cpu().numpy().view(np.uint64)
    y = gpu_ctx.fft(x, field=PALLAS_FR)
    assert (nat == y).all()
    rng = np.random.default_rng(24)
    ks = np.unique(np.concatenate([[0, 1, n - 1, n // 2 + 1], rng.integers(0, n, 60)]))
    w = oracle.root_of_unity(lg)
    pts = np.stack([oracle.pow(w, int(k)) for k in ks])
    assert (nat[ks] == oracle.poly_eval_many(0, x, pts)).all()


# ---- configs[4]'s size (2^26) on one GPU: the transform above the direct-table limit, and the commit + query phases ----------------
def test_ntt_2pow26_two_level_tables_round_trip_and_sampled_dft_rows(gpu_ctx, oracle):
    """2^26 points (BASELINE configs[4]; 2 GiB per vector): above 2^24 the plan has no direct twiddle tables, every inter-pass twiddle is a
    two-level lookup.  Checked on the device: inverse(forward(x)) == x bit for bit; and against the definition: sampled outputs equal the
    DFT row sum_j x_j w^(jk) computed by Horner on the CPU (2^26 terms per row)."""
    import torch
    lg = 26; n = 1 << lg
    x = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    gpu_ctx._chk(gpu_ctx.lib.stark_synth_column_dev(gpu_ctx.h, 0x5EED0000 + lg, 7, 0, n, C.c_void_p(x.data_ptr())))
    y = x.clone()
    gpu_ctx._chk(gpu_ctx.lib.stark_ntt_dev(gpu_ctx.h, PALLAS_FR, C.c_void_p(y.data_ptr()), lg, 0, None))
    z = y.clone()
    gpu_ctx._chk(gpu_ctx.lib.stark_ntt_dev(gpu_ctx.h, PALLAS_FR, C.c_void_p(z.data_ptr()), lg, 1, None))
    gpu_ctx.sync(); torch.cuda.synchronize()
    assert torch.equal(z, x) and not torch.equal(y, x)
    ks = [0, 1, n - 1, 0x2A5F3C1]
    got = y[torch.tensor(ks, device="cuda")].cpu().numpy().view(np.uint64)
    xh = x.cpu().numpy().view(np.uint64); del x, y, z
    w = oracle.root_of_unity(lg)
    pts = np.stack([oracle.pow(w, int(k)) for k in ks])
    assert (got == oracle.poly_eval_many(0, xh, pts)).all()
    gpu_ctx.trim()


def test_commit_and_query_phases_at_2pow26_accepted_by_reference_verifier(gpu_ctx, oracle):
    """FRI commit (folds, 2^26 leaf hashes, four trees) + query phase + canonical encoding on n0 = 2^26 evaluations (configs[4]'s trace
    size; f0 synthetic — the serial column sponges in front of it would take ten minutes): the reference's verifier restatement accepts the
    bytes, the size estimator agrees, a flipped bit is rejected, and the product's own verifier agrees with both decisions."""
    import torch
    lg, r, sched = 26, 40, [16, 16, 8]
    n0 = 1 << lg
    f0 = torch.empty((n0, 4), dtype=torch.int64, device="cuda")
    gpu_ctx._chk(gpu_ctx.lib.stark_synth_column_dev(gpu_ctx.h, 0x5EED0000 + lg, 5, 0, n0, C.c_void_p(f0.data_ptr())))
    sch = np.ascontiguousarray(sched, dtype=np.uint64); h = C.c_void_p()
    gpu_ctx._chk(gpu_ctx.lib.stark_deep_fri_prove_dev(gpu_ctx.h, None, None, None, None, C.c_void_p(f0.data_ptr()), n0, sch.ctypes.data_as(C.c_void_p), 3, r, 0xDEEFBAAD, C.byref(h)))
    proof, est = gpu_ctx._proof_out(h)
    del f0; gpu_ctx.trim()
    assert oracle.deep_fri_verify(proof, sched, r, 0xDEEFBAAD) == 1
    assert oracle.proof_size_estimate_from_bytes(proof) == est
    prm = DeepFriParams(sched, r, 0xDEEFBAAD)
    assert gpu_ctx.deep_fri_verify(prm, proof) is True
    bad = bytearray(proof); bad[len(bad) // 2] ^= 0x10
    assert oracle.deep_fri_verify(bytes(bad), sched, r, 0xDEEFBAAD) == 0
    assert gpu_ctx.deep_fri_verify(prm, bytes(bad)) is False


# Opt-in (about 5 GPU-minutes: 2^22 dependent permutations per column): defined only under STARK_LONG_TESTS=1, so that the default selection has nothing
# skipped; the recorded run is profiles/r03_prove_2pow26_end_to_end.json.
if os.environ.get("STARK_LONG_TESTS") == "1":
    def test_end_to_end_prove_2pow26_from_trace_columns(gpu_ctx, oracle):
        """configs[4]'s trace size END TO END on one GPU: `stark_deep_fri_prove_dev` from four 2^26-row columns (8 GiB of trace; DeepAliRealBuilder's
        serial sponges, merge, commit, 40 queries).  No oracle golden exists at this size (the oracle would need about seven hours): the proof must be
        accepted by the oracle's verifier restatement and by the product's, with the size estimate agreeing and a flipped bit rejected by both.
        (Exact bytes at scale are pinned by the 2^24 golden; this run shows the path at configs[4]'s size: 305 s, 304 s of it the column sponges.)"""
        import torch
        lg, r, sched = 26, 40, [16, 16, 8]
        n0 = 1 << lg
        cols = _dev_cols(gpu_ctx, 0x5EED0000 + lg, n0)
        sch = np.ascontiguousarray(sched, dtype=np.uint64); h = C.c_void_p()
        gpu_ctx._chk(gpu_ctx.lib.stark_deep_fri_prove_dev(gpu_ctx.h, *[C.c_void_p(c.data_ptr()) for c in cols], None, n0, sch.ctypes.data_as(C.c_void_p), 3, r, SEED_Z, C.byref(h)))
        ms = [gpu_ctx.lib.stark_proof_stage_ms(h, i) for i in range(3)]
        proof, est = gpu_ctx._proof_out(h)
        ok_oracle = oracle.deep_fri_verify(proof, sched, r, SEED_Z) == 1
        ok_est = oracle.proof_size_estimate_from_bytes(proof) == est
        ok_gpu = gpu_ctx.deep_fri_verify(DeepFriParams(sched, r, SEED_Z), proof) is True
        bad = bytearray(proof); bad[len(bad) // 3] ^= 0x04
        rej = oracle.deep_fri_verify(bytes(bad), sched, r, SEED_Z) == 0 and gpu_ctx.deep_fri_verify(DeepFriParams(sched, r, SEED_Z), bytes(bad)) is False
        rec = {"log_n0": lg, "r": r, "proof_len": len(proof), "size_estimate": est, "sha256": hashlib.sha256(proof).hexdigest(), "build_f0_ms": ms[0], "fri_build_ms": ms[1],
               "queries_encode_ms": ms[2], "us_per_dependent_permutation": ms[0] * 1e3 / (n0 / 16 + 2), "accepted_by_oracle_verifier": ok_oracle, "size_estimate_agrees": ok_est,
               "accepted_by_product_verifier": ok_gpu, "tampered_rejected_by_both": rej}
        out = os.path.join(ROOT, "gpurun_out"); os.makedirs(out, exist_ok=True)
        json.dump(rec, open(os.path.join(out, "prove_2pow26_end_to_end.json"), "w"), indent=1)
        del cols; gpu_ctx.trim()
        assert ok_oracle and ok_est and ok_gpu and rej, rec


# ---- boundary rules --------------------------------------------------------------------------------------------------
def test_default_stream_context_orders_against_torch_without_manual_sync(oracle):
    """include/stark_mlwe.h "Stream rule": Context(stream=None) runs on the legacy default stream, so a torch default-stream
    producer (zero fill + index gather into the input) followed by a *_dev call and a torch consumer needs NO manual
    synchronisation.  (Round 1: a NULL context silently ran on a private non-blocking stream and raced exactly this pattern.)"""
    import torch
    from stark_mlwe_amd.api import Context
    ctx = Context(0)
    try:
        assert ctx.stream_handle == 0 and not ctx.private_stream
        n, m = 1 << 21, 16
        x = oracle.synth_column(5, 3, 0, n)
        z = oracle.from_u64(0xABCDEF)
        want = oracle.fri_fold_layer(x, z, m)
        src = torch.from_numpy(x.view(np.int64).copy()).cuda()
        perm = torch.randperm(n, device="cuda"); inv = torch.argsort(perm)
        shuffled = src[perm].contiguous()
        torch.cuda.synchronize()
        zz = np.ascontiguousarray(z)
        for rep in range(4):
            buf = torch.zeros((n, 4), dtype=torch.int64, device="cuda")         # producer 1: zero fill (default stream)
            buf.copy_(shuffled[inv])                                             # producer 2: gather + copy (default stream)
            out = torch.zeros((n // m, 4), dtype=torch.int64, device="cuda")
            ctx._chk(ctx.lib.stark_fri_fold_dev(ctx.h, C.c_void_p(buf.data_ptr()), n, zz.ctypes.data_as(C.c_void_p), m, C.c_void_p(out.data_ptr())))
            got = out.cpu().numpy().view(np.uint64)                             # consumer: torch D2H on the default stream
            assert (got == want).all(), f"rep {rep}: library work was not ordered against torch's default stream"
    finally:
        ctx.close()


def test_private_stream_context_with_explicit_sync(oracle):
    import torch
    from stark_mlwe_amd.api import Context, STREAM_PRIVATE
    ctx = Context(0, STREAM_PRIVATE)
    try:
        assert ctx.private_stream
        x = oracle.synth_column(5, 4, 0, 1 << 12)
        d = torch.from_numpy(x.view(np.int64).copy()).cuda(); torch.cuda.synchronize()
        ctx._chk(ctx.lib.stark_ntt_dev(ctx.h, PALLAS_FR, C.c_void_p(d.data_ptr()), 12, 0, None)); ctx.sync()
        assert (d.cpu().numpy().view(np.uint64) == oracle.ntt(0, x)).all()
    finally:
        ctx.close()


def test_leaf_pair_hash_rejects_foreign_t17_params(gpu_ctx, oracle):
    """hash_leaf_pair is defined over transcript::default_params() (fri.rs:39).  A t = 17 handle with other constants used to
    mix two parameter sets silently; it is refused now, while NULL or a handle holding the transcript constants work."""
    f = oracle.synth_column(9, 1, 0, 70); want = oracle.leaf_pair_hash(f, None, 1)
    out = np.zeros((70, 4), np.uint64)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    other = gpu_ctx.generate_params_t17_x5(b"POSEIDON-T17-X5-SEED")
    assert gpu_ctx.lib.stark_leaf_pair_hash(gpu_ctx.h, other.h, P(f), None, 70, 1, P(out)) == -1       # STARK_ERR_INVALID_ARG
    assert b"default_params" in gpu_ctx.lib.stark_last_error(gpu_ctx.h)
    other.free()
    same = gpu_ctx.generate_params_t17_x5(b"POSEIDON-T17-X5-TRANSCRIPT")
    assert gpu_ctx.lib.stark_leaf_pair_hash(gpu_ctx.h, same.h, P(f), None, 70, 1, P(out)) == 0 and (out == want).all()
    same.free()
    out[:] = 0
    assert gpu_ctx.lib.stark_leaf_pair_hash(gpu_ctx.h, None, P(f), None, 70, 1, P(out)) == 0 and (out == want).all()
    p9 = gpu_ctx.poseidon_params_for_width(9)
    assert gpu_ctx.lib.stark_leaf_pair_hash(gpu_ctx.h, p9.h, P(f), None, 70, 1, P(out)) == -1


def test_pooled_allocator_reuses_blocks_and_trims(gpu_ctx, oracle):
    """The library's temporaries / layers / levels come from a per-context cache: a second identical call allocates nothing
    new, results stay bit-exact, and stark_ctx_trim returns the cache."""
    n0 = 1 << 12
    f0 = oracle.synth_column(12, 5, 0, n0)
    st = gpu_ctx.fri_build_transcript(f0, SCHED, SEED_Z); r1 = [st.root(l).copy() for l in range(4)]; st.free()
    cached = gpu_ctx.lib.stark_ctx_cached_bytes(gpu_ctx.h)
    assert cached > 0
    st = gpu_ctx.fri_build_transcript(f0, SCHED, SEED_Z); r2 = [st.root(l).copy() for l in range(4)]; st.free()
    assert all((a == b).all() for a, b in zip(r1, r2))
    assert gpu_ctx.lib.stark_ctx_cached_bytes(gpu_ctx.h) == cached           # steady state: nothing new was allocated
    ref = oracle.deep_fri_prove(None, None, None, None, n0, SCHED, 1, SEED_Z, f0=f0)
    assert all((ref.root(l) == r1[l]).all() for l in range(4)); ref.free()
    gpu_ctx.trim()
    assert gpu_ctx.lib.stark_ctx_cached_bytes(gpu_ctx.h) == 0


# ---- multi-GPU split, rehearsed on the one GPU ---------------------------------------------------------------------------
def test_library_rccl_communicator_one_rank(gpu_ctx):
    """stark_comm_* (RCCL bound at run time) on a one-rank communicator: the exact calls the N > 1 bench makes (all-to-all of
    the six-step transpose, all-gather of tree tops, u64 all-reduce of the query table, gather of a column), on the context's
    stream, with int64 limb tensors."""
    import torch
    from stark_mlwe_amd import dist as sd
    comm = sd.LibComm(gpu_ctx, 0, 1)
    try:
        assert gpu_ctx.lib.stark_comm_size(gpu_ctx.h) == 1 and gpu_ctx.lib.stark_comm_rank(gpu_ctx.h) == 0
        x = (torch.arange(4 * 64 * 4, dtype=torch.int64, device="cuda") * 0x9E3779B97F4A7C15 % (1 << 62)).view(4 * 64, 4)
        assert bool((comm.all_to_all(x.view(1, -1, 4)).view(-1, 4) == x).all())
        assert bool((comm.all_gather(x) == x).all())
        assert bool((comm.all_reduce_sum(x) == x).all())
        assert bool((comm.all_reduce_sum(x.cpu()) == x.cpu()).all())
        assert bool((comm.gather_to(x, 0) == x).all())
        gpu_ctx.sync()
    finally:
        comm.close()
    # without a communicator the data-path calls fail loudly
    import torch
    y = torch.zeros((4, 4), dtype=torch.int64, device="cuda"); z2 = torch.zeros_like(y)
    assert gpu_ctx.lib.stark_comm_all_to_all_dev(gpu_ctx.h, C.c_void_p(y.data_ptr()), C.c_void_p(z2.data_ptr()), 128) == -3      # STARK_ERR_RCCL


@pytest.mark.parametrize("log_n", [12, 16, 22])
def test_sharded_trace_world1_equals_single_gpu_step(gpu_ctx, oracle, log_n):
    """The N > 1 bench step (ShardedLde = 2^b coset transforms through the six-step building blocks + pack kernels, shard merge,
    sharded commit) on ONE rank through the library's communicator must give the roots of the single-GPU step (stark_lde_dev,
    stark_ali_merge_dev, stark_fri_build_dev) — and, at 2^12, the oracle's.  2^22 rows (LDE to 2^25, six-step transforms of
    2^22 points) is the largest trace whose two code paths fit a test's time budget side by side."""
    import torch
    import bench
    from stark_mlwe_amd import dist as sd
    from stark_mlwe_amd.api import _ptr
    lb, sched = 3, [16, 16, 8]
    n, N = 1 << log_n, 1 << (log_n + lb)
    cols = _dev_cols(gpu_ctx, 0x5EED0000 + log_n, n)
    coset, z, omega = bench._mont_small(5), bench._mont_small(0xC0FFEE), bench._root_of_unity_pallas(log_n + lb)
    comm = sd.set_comm(sd.LibComm(gpu_ctx, 0, 1))
    try:
        job = sd.ShardedTrace(sd.HipProvider(gpu_ctx), log_n, lb, sched, SEED_Z, coset, z)
        roots = job.step(cols)
        # sharded LDE alone against the library's single-GPU LDE
        ext_sh = job.lde(cols[1]); ext = torch.empty((N, 4), dtype=torch.int64, device="cuda")
        gpu_ctx._chk(gpu_ctx.lib.stark_lde_dev(gpu_ctx.h, PALLAS_FR, C.c_void_p(cols[1].data_ptr()), log_n, lb, _ptr(coset), C.c_void_p(ext.data_ptr())))
        assert bool((ext_sh == ext).all())
    finally:
        sd.set_comm(None); comm.close()
    exts = [torch.empty((N, 4), dtype=torch.int64, device="cuda") for _ in range(4)]
    for c in range(4):
        gpu_ctx._chk(gpu_ctx.lib.stark_lde_dev(gpu_ctx.h, PALLAS_FR, C.c_void_p(cols[c].data_ptr()), log_n, lb, _ptr(coset), C.c_void_p(exts[c].data_ptr())))
    f0 = torch.empty((N, 4), dtype=torch.int64, device="cuda")
    gpu_ctx._chk(gpu_ctx.lib.stark_ali_merge_dev(gpu_ctx.h, *[C.c_void_p(e.data_ptr()) for e in exts], None, None, _ptr(omega), _ptr(z), N, C.c_void_p(f0.data_ptr()), None))
    st = C.c_void_p(); sch = np.ascontiguousarray(sched, dtype=np.uint64)
    gpu_ctx._chk(gpu_ctx.lib.stark_fri_build_dev(gpu_ctx.h, C.c_void_p(f0.data_ptr()), N, _ptr(sch), 3, SEED_Z, C.byref(st)))
    for l in range(4):
        r = np.zeros(4, np.uint64); gpu_ctx._chk(gpu_ctx.lib.stark_fri_layer_root(st, l, _ptr(r)))
        assert (np.asarray(roots[l]).view(np.uint64).reshape(4) == r).all(), f"layer {l}"
    gpu_ctx._chk(gpu_ctx.lib.stark_fri_state_free(st))
    if log_n <= 12:
        ref = oracle.deep_fri_prove(None, None, None, None, N, sched, 1, SEED_Z, f0=f0.cpu().numpy().view(np.uint64))
        assert all((ref.root(l) == np.asarray(roots[l]).view(np.uint64).reshape(4)).all() for l in range(4)); ref.free()
