"""Two ranks sharing ONE MI355X (gloo for the exchange, staged through host memory): the product kernels in
a real 2-way decomposition — non-zero global column offsets in the six-step NTT, global DS positions in
the sharded Merkle tree — against the oracle.  The 8-GPU RCCL run itself is the driver's.
(File name: sorts after test_gpu_parity.py so the single-process parity suite reports first.)"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rendezvous_env(port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["STARK_DIST_CHECK"] = "1"                    # DistProver compares the roots across ranks before the query phase
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # loopback only: never resolve the box's hostname


def _collect(procs, q, n, what, limit_s=420):
    """Results of n workers; a worker that died or hangs fails the test with a message instead of stalling the run."""
    import queue, time
    res, t0 = [], time.time()
    while len(res) < n:
        try:
            res.append(q.get(timeout=20))
        except queue.Empty:
            sys.__stderr__.write(f"[{what}: waiting {int(time.time() - t0)} s]\n"); sys.__stderr__.flush()
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > limit_s:
                for p in procs:
                    if p.is_alive(): p.terminate()
                pytest.fail(f"{what}: workers did not report (exit codes {[p.exitcode for p in procs]}, {int(time.time() - t0)} s)")
    for p in procs: p.join(60)
    return res


def _worker(rank, world, port, log_n, log_rows, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    _rendezvous_env(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_lib
        from stark_mlwe_amd import dist as sd
        from stark_mlwe_amd.api import Context
        torch.cuda.set_device(0)
        ctx = Context(0); o = oracle_lib.Oracle(); prov = sd.HipProvider(ctx)
        n = 1 << log_n
        x = o.synth_column(77, 7, 0, n)
        want = o.ntt(0, x)
        plan = sd.DistNtt(prov, log_n, log_rows)
        slab = torch.from_numpy(x[plan.local_input_indices().reshape(-1).numpy()].view(np.int64).copy()).cuda()
        rows = plan.forward(slab); ctx.sync()
        ok_t = bool((rows.cpu().numpy().view(np.uint64) == want[plan.local_output_indices().reshape(-1).numpy()]).all())
        nat = plan.to_natural_blocks(rows).cpu().numpy().view(np.uint64)
        ok_n = bool((nat == want[rank * n // world:(rank + 1) * n // world]).all())
        leaves = o.synth_column(5, 1, 0, 1 << 13)
        half = (1 << 13) // world
        mine = torch.from_numpy(leaves[rank * half:(rank + 1) * half].view(np.int64).copy()).cuda()
        root = sd.merkle_sharded_root(prov, ctx.poseidon_params_for_width(17), 16, 9, mine, half)
        ok_m = bool((root.cpu().numpy().view(np.uint64) == o.merkle_build(16, 9, leaves).root()).all())
        ctx.close()
        q.put((rank, ok_t, ok_n, ok_m))
    except Exception as ex:      # noqa: BLE001 — report instead of leaving the parent waiting
        import traceback
        q.put(("error", repr(ex), traceback.format_exc()[-1500:]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("log_n,log_rows", [(14, 7), (20, 10)])
def test_two_ranks_share_one_gpu(log_n, log_rows):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 1000) + log_n
    procs = [ctx.Process(target=_worker, args=(r, 2, port, log_n, log_rows, q)) for r in range(2)]
    for p in procs: p.start()
    res = _collect(procs, q, 2, "six-step + sharded merkle")
    assert sorted(res) == [(0, True, True, True), (1, True, True, True)], res


def _prove_worker(rank, world, port, log_n0, schedule, r, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    _rendezvous_env(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_lib
        from stark_mlwe_amd import dist as sd
        from stark_mlwe_amd.api import Context, DeepFriParams
        torch.cuda.set_device(0)
        if log_n0 >= 16:      # one stream for torch and the library (what bench.py does) ...
            import ctypes as C
            ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
            ctx = Context(0, C.c_void_p(ts.cuda_stream))
        else:                 # ... or the library on a private stream: the provider then brackets every call with explicit syncs
            from stark_mlwe_amd.api import STREAM_PRIVATE
            ctx = Context(0, STREAM_PRIVATE)
        o = oracle_lib.Oracle(); prov = sd.HipProvider(ctx)
        assert prov._shared() == (log_n0 >= 16)
        n0 = 1 << log_n0; nl = n0 // world
        cols = [o.synth_column(0x5EED0000 + log_n0, c, 0, n0) for c in range(4)]
        mine = [torch.from_numpy(c[rank * nl:(rank + 1) * nl].view(np.int64).copy()).cuda() for c in cols]
        dp = sd.DistProver(prov, n0, schedule, r, 0xDEEFBAAD)
        proof, est = dp.prove(*mine)
        single, est1, _ = ctx.deep_fri_prove(*cols, n0, DeepFriParams(schedule, r, 0xDEEFBAAD))      # the one-GPU product path on the whole trace
        ok_oracle = None
        if log_n0 <= 12:
            ref = o.deep_fri_prove(*cols, n0, schedule, r, 0xDEEFBAAD); ok_oracle = ref.bytes() == proof; ref.free()
        ctx.close()
        q.put((rank, proof == single, est == est1, ok_oracle, o.deep_fri_verify(proof, schedule, r, 0xDEEFBAAD)))
    except Exception as ex:      # noqa: BLE001 — report instead of leaving the parent waiting
        import traceback
        q.put(("error", repr(ex), traceback.format_exc()[-1500:]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("log_n0,schedule,r", [(12, [16, 16, 8], 32), (16, [16, 16, 8], 32)])
def test_sharded_prove_two_ranks_one_gpu(log_n0, schedule, r):
    """One trace block-sharded over 2 ranks (sharing the GPU): the same proof bytes as the one-GPU prove and the oracle."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 1000) + log_n0
    procs = [ctx.Process(target=_prove_worker, args=(r_, 2, port, log_n0, schedule, r, q)) for r_ in range(2)]
    for p in procs: p.start()
    res = _collect(procs, q, 2, "sharded prove")
    want_oracle = True if log_n0 <= 12 else None
    assert sorted(res) == [(0, True, True, want_oracle, 1), (1, True, True, want_oracle, 1)], res


def _trace_worker(rank, world, port, log_n, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    _rendezvous_env(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes as C
        import bench, oracle_lib
        from stark_mlwe_amd import dist as sd
        from stark_mlwe_amd.api import Context, PALLAS_FR, _ptr
        torch.cuda.set_device(0)
        ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
        ctx = Context(0, C.c_void_p(ts.cuda_stream)); lib = ctx.lib
        lb, sched = 3, [16, 16, 8]
        n_tot = 1 << log_n; nl = n_tot // world; N = n_tot << lb
        coset, z, omega = bench._mont_small(5), bench._mont_small(0xC0FFEE), bench._root_of_unity_pallas(log_n + lb)
        mine = [torch.empty((nl, 4), dtype=torch.int64, device="cuda") for _ in range(4)]
        whole = [torch.empty((n_tot, 4), dtype=torch.int64, device="cuda") for _ in range(4)]
        for c in range(4):
            ctx._chk(lib.stark_synth_column_dev(ctx.h, 0x5EED0000 + log_n, c, rank * nl, nl, C.c_void_p(mine[c].data_ptr())))
            ctx._chk(lib.stark_synth_column_dev(ctx.h, 0x5EED0000 + log_n, c, 0, n_tot, C.c_void_p(whole[c].data_ptr())))
        job = sd.ShardedTrace(sd.HipProvider(ctx, device=torch.device("cuda", 0)), log_n, lb, sched, 0xDEEFBAAD, coset, z)
        roots = job.step(mine)
        # the one-GPU path over the whole trace
        exts = [torch.empty((N, 4), dtype=torch.int64, device="cuda") for _ in range(4)]
        for c in range(4):
            ctx._chk(lib.stark_lde_dev(ctx.h, PALLAS_FR, C.c_void_p(whole[c].data_ptr()), log_n, lb, _ptr(coset), C.c_void_p(exts[c].data_ptr())))
        f0 = torch.empty((N, 4), dtype=torch.int64, device="cuda")
        ctx._chk(lib.stark_ali_merge_dev(ctx.h, *[C.c_void_p(e.data_ptr()) for e in exts], None, None, _ptr(omega), _ptr(z), N, C.c_void_p(f0.data_ptr()), None))
        st = C.c_void_p(); sch = np.ascontiguousarray(sched, dtype=np.uint64)
        ctx._chk(lib.stark_fri_build_dev(ctx.h, C.c_void_p(f0.data_ptr()), N, _ptr(sch), 3, 0xDEEFBAAD, C.byref(st)))
        ok = True
        for l in range(4):
            r = np.zeros(4, np.uint64); ctx._chk(lib.stark_fri_layer_root(st, l, _ptr(r)))
            ok = ok and bool((np.asarray(roots[l]).view(np.uint64).reshape(4) == r).all())
        gold = bench.golden_step_roots(log_n, 0x5EED0000 + log_n)      # the CPU oracle's roots for this very trace, where a golden exists (2^20 rows)
        if gold is not None:
            ok = ok and bench.roots_hex(roots) == gold
        ctx._chk(lib.stark_fri_state_free(st)); ctx.close()
        q.put((rank, ok))
    except Exception as ex:      # noqa: BLE001 — report instead of leaving the parent waiting
        import traceback
        q.put(("error", repr(ex), traceback.format_exc()[-1500:]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("log_n", [14, 18, 20])
def test_sharded_trace_two_ranks_one_gpu(log_n):
    """The N > 1 bench step on 2 ranks sharing the GPU (gloo exchange): sharded LDE (six-step NTTs with real rank offsets, pack
    kernels), shard merge and sharded commit give the roots of the one-GPU step over the whole trace — and at 2^20 rows (the bench's
    trace) the roots of the CPU oracle (tests/golden/step_roots_k20.json)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30300 + (os.getpid() % 1000) + log_n
    procs = [ctx.Process(target=_trace_worker, args=(r_, 2, port, log_n, q)) for r_ in range(2)]
    for p in procs: p.start()
    res = _collect(procs, q, 2, "sharded trace step")
    assert sorted(res) == [(0, True), (1, True)], res


def _rccl_worker(port, q):
    _rendezvous_env(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        x = (torch.arange(4 * 64 * 4, dtype=torch.int64, device="cuda") * 0x9E3779B97F4A7C15 % (1 << 62)).view(4 * 64, 4)
        out = torch.empty_like(x)
        dist.all_to_all_single(out.view(-1), x.contiguous().view(-1))                 # the six-step transpose call (int64 limbs)
        ok_a2a = bool((out == x).all())
        lst = [torch.empty_like(x)]
        dist.all_gather(lst, x.contiguous()); ok_ag = bool((lst[0] == x).all())       # tree tops / small layers
        y = x.clone(); dist.all_reduce(y); ok_ar = bool((y == x).all())               # the query value table (int64 SUM)
        parts = [torch.empty_like(x)]
        dist.gather(x.contiguous(), parts, dst=0); ok_g = bool((parts[0] == x).all()) # columns to their sponge rank
        t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier(); torch.cuda.synchronize()
        q.put((ok_a2a, ok_ag, ok_ar, ok_g, float(t.item())))
    except Exception as ex:      # noqa: BLE001 — report instead of leaving the parent waiting
        import traceback
        q.put(("error", repr(ex), traceback.format_exc()[-1500:]))
    finally:
        dist.destroy_process_group()


def test_rccl_collectives_accept_our_tensors():
    """The exact RCCL calls of stark_mlwe_amd/dist.py and bench.py (dtype int64 limbs on the device) on a
    one-rank communicator: catches API / dtype refusals that only the nccl backend would raise."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(30900 + os.getpid() % 1000, q))
    p.start()
    res = _collect([p], q, 1, "rccl one-rank", limit_s=240)[0]
    assert res == (True, True, True, True, 1.5), res


def _checked_comm_worker(port, q):
    _rendezvous_env(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        import ctypes as C
        from stark_mlwe_amd.api import Context
        from stark_mlwe_amd import dist as sd
        dev = torch.device("cuda", 0)
        ctx = Context(0, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        lc, msg = sd.checked_lib_comm(ctx, 0, 1, dev, True)
        none, msg2 = sd.checked_lib_comm(ctx, 0, 1, dev, False)
        ok = lc is not None and msg.startswith("library RCCL") and none is None and "unavailable" in msg2
        if lc is not None:
            x = (torch.arange(64 * 4, dtype=torch.int64, device=dev) * 7 + 3).view(64, 4)
            ok = ok and bool((lc.all_to_all(x) == x).all()) and bool((lc.all_reduce_sum(x) == x).all())
            lc.close()
        ctx.close()
        q.put((ok, msg))
    except Exception as ex:      # noqa: BLE001
        import traceback
        q.put(("error", repr(ex), traceback.format_exc()[-1500:]))
    finally:
        dist.destroy_process_group()


def test_library_communicator_start_up_check_against_torch_distributed():
    """bench.py's N > 1 start-up: the library's RCCL communicator is adopted only after one all-to-all / all-gather agreed with
    torch.distributed's on the same bytes, the ranks deciding together (here: the one rank a one-GPU box allows)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_checked_comm_worker, args=(31900 + os.getpid() % 1000, q))
    p.start()
    res = _collect([p], q, 1, "checked lib comm", limit_s=240)[0]
    assert res[0] is True, res
