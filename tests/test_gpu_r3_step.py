"""Round-3 parity tests: the bench step itself (BASELINE.json configs[1]) at its own size against roots the CPU oracle produced
ONCE (tools/gen_golden.py step:K -> tests/golden/step_roots_kK.json).  Needs an MI355X: `pytest -m gpu`."""
import ctypes as C
import json
import os

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.mark.parametrize("k", [10, 20, 21, 22])
def test_bench_step_roots_match_oracle_golden(gpu_ctx, k):
    """Exactly `bench.py`'s N = 1 step (`bench.make_single_gpu_step`: LDE of the four synthetic columns 2^k -> 2^(k+3) on the coset
    5*<w>, DEEP-ALI merge at z = 0xC0FFEE, fri_build_transcript [16,16,8] — crates/deep_ali/src/lib.rs:48-105,
    crates/deep_ali/src/fri.rs:231-312): the four layer roots equal the oracle's, at 2^10 rows, at the bench size 2^20 rows x 8, and at 2^21 / 2^22 rows (the traces of the N = 2 and N = 4 bench)."""
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


def test_library_communicator_on_private_stream_context_is_bracketed():
    """LibComm enqueues on the CONTEXT's stream while torch produces / consumes its tensors on torch's current
    stream.  With a STARK_STREAM_PRIVATE context nothing orders the two, so LibComm must bracket every call like HipProvider._run
    does.  A long torch-side producer (a chain of elementwise kernels) right before each collective makes a missing bracket visible:
    the collective would read the buffer before the producer has finished."""
    import torch
    from stark_mlwe_amd import dist as sd
    from stark_mlwe_amd.api import Context, STREAM_PRIVATE
    ctx = Context(0, STREAM_PRIVATE)
    try:
        assert ctx.private_stream
        assert ctx.lib.stark_comm_available(None) == 0
        comm = sd.LibComm(ctx, 0, 1)
        try:
            assert not comm._shared(torch.device("cuda", 0))
            for it in range(3):
                x = torch.zeros((1 << 20, 4), dtype=torch.int64, device="cuda")
                for k in range(40):                      # producer still running on torch's stream when the collective is issued
                    x += 1
                want = torch.full_like(x, 40)
                assert bool((comm.all_to_all(x.view(1, -1, 4)).view(-1, 4) == want).all())
                assert bool((comm.all_gather(x) == want).all())
                assert bool((comm.all_reduce_sum(x) == want).all())
                assert bool((comm.all_reduce_sum(x).cpu() == want.cpu()).all())          # consumer-side: the .cpu() copy runs on torch's stream
                assert bool((comm.gather_to(x, 0) == want).all())
        finally:
            comm.close()
    finally:
        ctx.close()


@pytest.mark.parametrize("k,B", [(10, 5), (16, 16), (11, 70)])
def test_batch_prove_equals_single_proves(gpu_ctx, oracle, k, B):
    """stark_deep_fri_prove_batch_dev: B independent traces in one call (the reference's bench proves one after another,
    channel/benches/end_to_end.rs:229-309).  Each proof is byte-equal to the single prove of its trace (and, at 2^10, to the
    oracle's).  The 4 * B serial column sponges of crates/deep_ali/src/fri.rs:548-557 run concurrently, so the sponge stage of the batch costs what it
    costs for one trace; the B tails (merge, commit, queries: a few ms each, latency-bound) run up to four at a time on worker contexts.  16 proves of
    2^16 rows: 0.39 s against 0.31 s for one."""
    import time
    import torch
    from stark_mlwe_amd.api import DeepFriParams
    n0, sched, r, seed_z = 1 << k, [16, 16, 8] if k >= 12 else [16, 8], 32 if k >= 12 else 8, 0xDEEFBAAD
    prm = DeepFriParams(sched, r, seed_z)
    traces, keep = [], []
    for p in range(B):
        cols = [torch.empty((n0, 4), dtype=torch.int64, device="cuda") for _ in range(4)]
        for c in range(4):
            gpu_ctx._chk(gpu_ctx.lib.stark_synth_column_dev(gpu_ctx.h, 0xBA7C4000 + 16 * k + p, c, 0, n0, C.c_void_p(cols[c].data_ptr())))
        keep.append(cols); traces.append([c.data_ptr() for c in cols])
    torch.cuda.synchronize()
    sch = __import__("numpy").ascontiguousarray(sched, dtype="uint64")

    def single(p):
        h = C.c_void_p()
        gpu_ctx._chk(gpu_ctx.lib.stark_deep_fri_prove_dev(gpu_ctx.h, *[C.c_void_p(x) for x in traces[p]], None, n0, sch.ctypes.data_as(C.c_void_p), len(sched), r, seed_z, C.byref(h)))
        return gpu_ctx._proof_out(h)[0]
    single(0)                                              # warm constants, plans and the pool
    t0 = time.perf_counter(); one = single(0); t_one = time.perf_counter() - t0
    gpu_ctx.deep_fri_prove_batch_dev(traces[:1], n0, prm)   # warm the batch path
    t0 = time.perf_counter(); got = gpu_ctx.deep_fri_prove_batch_dev(traces, n0, prm); t_batch = time.perf_counter() - t0
    assert len(got) == B and got[0][0] == one
    for p in (range(B) if B <= 8 else (1, B // 2, B - 1)):          # B = 70: 280 chains, more workgroups than the chip holds at once
        assert got[p][0] == single(p), f"trace {p}"
    assert len({g[0] for g in got}) == B                    # different traces, different proofs
    if k == 10 and B <= 8:
        host_cols = [[c.cpu().numpy().view("uint64") for c in cols] for cols in keep]
        for p in range(B):
            ref = oracle.deep_fri_prove(*host_cols[p], n0, sched, r, seed_z)
            assert got[p][0] == ref.bytes(); ref.free()
    if k == 16:
        rec = {"log_n0": k, "batch": B, "one_prove_s": t_one, "batch_s": t_batch, "ratio": t_batch / t_one}
        out = os.path.join(ROOT, "gpurun_out"); os.makedirs(out, exist_ok=True)
        json.dump(rec, open(os.path.join(out, "batch_prove.json"), "w"), indent=1)
        assert t_batch <= 1.4 * t_one, rec


PRESET_CASES = [("mod16", [16, 16, 16, 16], 16), ("uni32x3", [32, 32, 32], 15), ("uni64x2x8", [64, 64, 8], 15), ("hi64_32_8", [64, 32, 8], 14), ("hi32_32_16", [32, 32, 16], 14)]


@pytest.mark.parametrize("label,sched,k", PRESET_CASES)
def test_reference_bench_presets_match_oracle_golden(gpu_ctx, label, sched, k):
    """The reference bench's schedules other than "paper" (channel/benches/end_to_end.rs:195-201: layers of arity 32 / 64, Poseidon widths
    33 / 65) at the smallest size the reference ran them (k_min = sum log2 m): proof bytes from (a, s, e, t) equal the oracle's
    (tests/golden/proof_k{k}_r32_{label}.json, tools/gen_golden.py preset:{label}:{k}), and the product's verifier accepts them."""
    import hashlib
    import numpy as np
    import torch
    from stark_mlwe_amd.api import DeepFriParams
    path = os.path.join(GOLD, f"proof_k{k}_r32_{label}.json")
    assert os.path.exists(path), f"{path} missing (tools/gen_golden.py preset:{label}:{k})"
    gold = json.load(open(path))
    assert gold["schedule"] == sched and gold["r"] == 32 and gold["log_n0"] == k
    n0 = 1 << k
    cols = [torch.empty((n0, 4), dtype=torch.int64, device="cuda") for _ in range(4)]
    for c in range(4):
        gpu_ctx._chk(gpu_ctx.lib.stark_synth_column_dev(gpu_ctx.h, gold["synth_seed"], c, 0, n0, C.c_void_p(cols[c].data_ptr())))
    sch = np.ascontiguousarray(sched, dtype=np.uint64); h = C.c_void_p()
    gpu_ctx._chk(gpu_ctx.lib.stark_deep_fri_prove_dev(gpu_ctx.h, *[C.c_void_p(c.data_ptr()) for c in cols], None, n0, sch.ctypes.data_as(C.c_void_p), len(sched), 32, gold["seed_z"], C.byref(h)))
    proof, est = gpu_ctx._proof_out(h)
    assert len(proof) == gold["proof_len"] and est == gold["size_estimate"]
    assert hashlib.sha256(proof).hexdigest() == gold["sha256"]
    assert gpu_ctx.deep_fri_verify(DeepFriParams(sched, 32, gold["seed_z"]), proof)


def test_handles_outlive_their_context(oracle):
    """a tree / FRI state / transcript freed AFTER stark_ctx_destroy used to touch a freed context (use-after-free in the pool).
    Handles now keep the context alive: destroy with handles outstanding only marks it, the handles stay usable, the last free tears it down."""
    import numpy as np
    from stark_mlwe_amd.api import Context, _ptr
    ctx = Context(0)
    lib = ctx.lib
    leaves = oracle.synth_column(9, 1, 0, 512)
    want = oracle.merkle_build(16, 7, leaves)
    p = C.c_void_p(); ctx._chk(lib.stark_poseidon_params_for_width(ctx.h, 17, C.byref(p)))
    t = C.c_void_p(); ctx._chk(lib.stark_merkle_build(ctx.h, p, 16, 7, _ptr(leaves), 512, 0, None, C.byref(t)))
    f0 = oracle.synth_column(10, 2, 0, 1 << 10); sch = np.ascontiguousarray([16, 8], dtype=np.uint64)
    st = C.c_void_p(); ctx._chk(lib.stark_fri_build(ctx.h, _ptr(f0), 1 << 10, _ptr(sch), 2, 0xDEEFBAAD, C.byref(st)))
    h = ctx.h; ctx.h = None                                  # the Python wrapper must not destroy it a second time
    assert lib.stark_ctx_destroy(h) == 0                     # three handles outstanding: marked, not torn down
    assert lib.stark_ctx_destroy(h) == -1                    # a second destroy is an error, not a double free
    root = np.zeros(4, np.uint64); assert lib.stark_merkle_root(t, _ptr(root)) == 0
    assert (root == want.root()).all()
    r0 = np.zeros(4, np.uint64); assert lib.stark_fri_layer_root(st, 0, _ptr(r0)) == 0
    ref = oracle.deep_fri_prove(None, None, None, None, 1 << 10, [16, 8], 1, 0xDEEFBAAD, f0=f0)
    assert (r0 == ref.root(0)).all(); ref.free(); want.free()
    assert lib.stark_fri_state_free(st) == 0
    assert lib.stark_merkle_free(t) == 0
    assert lib.stark_poseidon_params_free(p) == 0            # the last handle: the context is released here


@pytest.mark.parametrize("k,n", [(0, 3), (1, 2), (15, 2), (16, 2), (17, 3), (127, 2), (128, 3), (129, 1), (1000, 5), (4096 + 7, 2), (3, 512), (40, 513)])
def test_five_wave_sponge_equals_oracle_and_one_wave_kernel(gpu_ctx, oracle, k, n):
    """tr_hash_fields_tagged (crates/deep_ali/src/fri.rs:28-35) over n streams of k fields: up to 512 streams run the five-wave
    kernel (poseidon_chain.hpp: partial rounds unrolled, dependent chain in row form, helper waves), more than that and the option
    "sponge_one_wave" the round-2 one-wave kernel.  Both equal the oracle at every length around the rate-16 block boundaries."""
    import numpy as np
    fields = oracle.synth_column(4242 + k, 3, 0, max(1, k * n))[:k * n]
    want = np.stack([oracle.tr_hash_fields_tagged(b"ALI/S", fields[i * k:(i + 1) * k]) for i in range(n)])
    lib = gpu_ctx.lib
    got = {}
    for one_wave in (0, 1):
        gpu_ctx._chk(lib.stark_ctx_set_option(gpu_ctx.h, b"sponge_one_wave", one_wave))
        out = np.zeros((n, 4), np.uint64)
        buf = np.ascontiguousarray(fields if k * n else np.zeros((1, 4), np.uint64))
        gpu_ctx._chk(lib.stark_tr_hash_fields_tagged(gpu_ctx.h, None, b"ALI/S", buf.ctypes.data_as(C.c_void_p), k, n, out.ctypes.data_as(C.c_void_p)))
        got[one_wave] = out
    gpu_ctx._chk(lib.stark_ctx_set_option(gpu_ctx.h, b"sponge_one_wave", 0))
    assert (got[0] == want).all() and (got[1] == want).all()


@pytest.mark.parametrize("field,log_n,lb", [(0, 4, 1), (0, 12, 3), (0, 16, 3), (0, 21, 2), (1, 13, 3)])
def test_library_sharded_lde_one_rank_equals_single_gpu_lde(gpu_ctx, oracle, field, log_n, lb):
    """stark_lde_sharded_dev (the multi-GPU LDE as ONE C-ABI call: four exchanges inside the library) on a single rank without a communicator equals
    stark_lde_dev, and at small sizes the oracle's LDE; both fields.  (N > 1 runs the same code with RCCL exchanges — unmeasured, see INTEGRATION.md.)"""
    import numpy as np
    import torch
    from stark_mlwe_amd.api import _ptr
    n = 1 << log_n
    x = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    if field == 0:
        gpu_ctx._chk(gpu_ctx.lib.stark_synth_column_dev(gpu_ctx.h, 77 + log_n, 1, 0, n, C.c_void_p(x.data_ptr())))
        host = x.cpu().numpy().view(np.uint64)
    else:
        host = np.stack([oracle.from_int(pow(3, i + 1, 2**61 - 1) * 0x9E3779B97F4A7C15, field=1) for i in range(n)]) if log_n <= 13 else None
        x.copy_(torch.from_numpy(host.view(np.int64)))
    shift = oracle.from_u64(5 if field == 0 else 7, field)
    a = torch.empty((n << lb, 4), dtype=torch.int64, device="cuda"); b = torch.empty_like(a)
    gpu_ctx._chk(gpu_ctx.lib.stark_lde_sharded_dev(gpu_ctx.h, field, C.c_void_p(x.data_ptr()), log_n, lb, _ptr(shift), C.c_void_p(a.data_ptr())))
    gpu_ctx._chk(gpu_ctx.lib.stark_lde_dev(gpu_ctx.h, field, C.c_void_p(x.data_ptr()), log_n, lb, _ptr(shift), C.c_void_p(b.data_ptr())))
    assert bool((a == b).all())
    got = a.cpu().numpy().view(np.uint64) if log_n <= 13 else None
    del a, b, x; gpu_ctx.trim()
    if log_n <= 13:
        assert (got == oracle.lde(field, host, lb, shift)).all()


@pytest.mark.parametrize("W,log_n,lb", [(2, 6, 2), (2, 12, 3), (4, 12, 3), (8, 14, 3), (4, 21, 2), (8, 20, 3), (2, 21, 3), (8, 23, 3)])
def test_library_sharded_lde_emulated_ranks_equal_single_gpu_lde(gpu_ctx, W, log_n, lb):
    """The index arithmetic of stark_lde_sharded_dev for W > 1 (packs, per-peer chunks, global row / column offsets, the coset interleave) checked on one
    GPU: stark_diag_lde_sharded_emulated_dev runs the SAME phase code for W virtual ranks and does every exchange as device copies; the concatenated
    rank outputs must equal stark_lde_dev of the whole column.  (RCCL itself moves opaque bytes; what it cannot be tested for here is hardware.)"""
    import torch
    from stark_mlwe_amd.api import _ptr
    import bench
    n = 1 << log_n
    x = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    gpu_ctx._chk(gpu_ctx.lib.stark_synth_column_dev(gpu_ctx.h, 901 + log_n + W, 2, 0, n, C.c_void_p(x.data_ptr())))
    shift = bench._mont_small(5)
    a = torch.empty((n << lb, 4), dtype=torch.int64, device="cuda"); b = torch.empty_like(a)
    gpu_ctx._chk(gpu_ctx.lib.stark_diag_lde_sharded_emulated_dev(gpu_ctx.h, 0, W, C.c_void_p(x.data_ptr()), log_n, lb, _ptr(shift), C.c_void_p(a.data_ptr())))
    gpu_ctx._chk(gpu_ctx.lib.stark_lde_dev(gpu_ctx.h, 0, C.c_void_p(x.data_ptr()), log_n, lb, _ptr(shift), C.c_void_p(b.data_ptr())))
    assert bool((a == b).all())
    del a, b, x; gpu_ctx.trim()


@pytest.mark.parametrize("field,log_n", [(0, 12), (0, 14), (0, 17), (0, 22), (1, 14), (1, 22)])
def test_coset_ntt_with_merged_tables_equals_separate_tables_and_oracle(gpu_ctx, oracle, field, log_n):
    """A coset transform of two or more passes reads ONE table in its first pass (pre-scale folded into the twiddles, ntt_dev.hpp k_fill_coset_merged);
    option "ntt_merged_coset" = 0 keeps the separate pre-scale and twiddle tables: same values both ways, and the oracle's coset NTT at 2^12 / 2^14."""
    import numpy as np
    import torch
    from stark_mlwe_amd.api import _ptr
    n = 1 << log_n; lib = gpu_ctx.lib
    x = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    gpu_ctx._chk(lib.stark_synth_column_dev(gpu_ctx.h, 31 + log_n, 2, 0, n, C.c_void_p(x.data_ptr())))      # below 2^254: valid in both fields
    host = x.cpu().numpy().view(np.uint64) if log_n <= 14 else None
    shift = oracle.from_u64(5 if field == 0 else 7, field)
    outs = []
    try:
        for merged in (1, 0):
            gpu_ctx._chk(lib.stark_ctx_set_option(gpu_ctx.h, b"ntt_merged_coset", merged))
            y = x.clone()
            for _ in range(2):      # second call: the cached plan
                y.copy_(x); gpu_ctx._chk(lib.stark_ntt_dev(gpu_ctx.h, field, C.c_void_p(y.data_ptr()), log_n, 0, _ptr(shift)))
            outs.append(y)
    finally:
        gpu_ctx._chk(lib.stark_ctx_set_option(gpu_ctx.h, b"ntt_merged_coset", 1))
    assert bool((outs[0] == outs[1]).all())
    if host is not None:
        assert (outs[0].cpu().numpy().view(np.uint64) == oracle.ntt(field, host, coset=shift)).all()
    del outs, x, y; gpu_ctx.trim()


@pytest.mark.parametrize("nodes,arity,last", [(1, 16, 16), (1, 16, 3), (37, 16, 16), (512, 16, 11), (256, 16, 7), (257, 16, 16), (513, 16, 16), (200, 9, 2), (64, 12, 12), (4096, 16, 5), (4097, 16, 16)])
def test_small_merkle_levels_five_wave_kernel_equals_oracle_and_one_wave_kernel(gpu_ctx, oracle, nodes, arity, last):
    """Merkle levels of up to 256 nodes run one node per five-wave workgroup (poseidon_chain.hpp k_hash_ds_chain), up to 4096 one wave per node, above that
    the wave-pair kernel:
    the oracle's hash_with_ds_dynamic on every node (ragged last node, one- and two-permutation inputs), and the one-wave kernel under option
    "sponge_one_wave" on all of them."""
    import numpy as np
    p17 = gpu_ctx.poseidon_params_for_width(17)
    ch = oracle.synth_column(900 + nodes, arity, 0, (nodes - 1) * arity + last)
    got = gpu_ctx.hash_ds_level(p17, arity, 3, 1000, 42, ch)
    assert got.shape[0] == nodes
    fe = lambda x: oracle.from_u64(x)
    for k in sorted({0, 1, nodes // 2, nodes - 2, nodes - 1} & set(range(nodes))):
        kids = ch[arity * k: arity * k + arity]
        ds = np.array([fe(arity), fe(3), fe(1000 + k), fe(42)])
        assert (got[k] == oracle.hash_with_ds_dynamic(0, 17, ds, kids, kids.shape[0])).all(), k
    try:
        gpu_ctx._chk(gpu_ctx.lib.stark_ctx_set_option(gpu_ctx.h, b"sponge_one_wave", 1))
        assert (gpu_ctx.hash_ds_level(p17, arity, 3, 1000, 42, ch) == got).all()
    finally:
        gpu_ctx._chk(gpu_ctx.lib.stark_ctx_set_option(gpu_ctx.h, b"sponge_one_wave", 0))


@pytest.mark.parametrize("n,m", [(1, 1), (300, 4), (2048, 16), (2049, 16), (4096, 16), (4097, 2), (8192, 16), (8193, 8)])
def test_small_leaf_layers_five_wave_kernel_equals_oracle(gpu_ctx, oracle, n, m):
    """hash_leaf_pair over layers of up to 2048 leaves: one leaf per five-wave workgroup (k_leaf_pair_chain), up to 4096 one wave per leaf (k_leaf_pair_coop),
    above that the wave-pair kernel; oracle on every leaf, with and without f_next."""
    f = oracle.synth_column(700 + n, 0, 0, n); fn = oracle.synth_column(700 + n, 1, 0, (n + m - 1) // m)
    assert (gpu_ctx.leaf_pair_hash(f, fn, m) == oracle.leaf_pair_hash(f, fn, m)).all()
    assert (gpu_ctx.leaf_pair_hash(f, None, 1) == oracle.leaf_pair_hash(f, None, 1)).all()
