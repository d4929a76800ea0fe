"""world_size-2 `gloo` tests of the multi-GPU orchestration (stark_mlwe_amd/dist.py) on CPU.
The local compute is a stand-in built from the oracle / the host-check library (tests only); what is
under test is the sharding: six-step indexing, the all-to-all exchange layout, global DS positions."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class CpuStandIn:
    """Provider with the same methods as dist.HipProvider, on CPU tensors, computing with the oracle
    (NTT) and the host-check build of the product's kernel body (Merkle level)."""

    def __init__(self, field=0):
        import hostcheck_lib, oracle_lib
        self.o, self.hc, self.field = oracle_lib.Oracle(), hostcheck_lib.HostCheck(), field
        self.hparams = {}

    def _np(self, t): return t.numpy().view(np.uint64)

    def ntt_columns(self, slab, log_rows, ncols, col0, log_n, inverse):
        a = self._np(slab).reshape(1 << log_rows, ncols, 4)
        w = self.o.root_of_unity(log_n, self.field)
        if inverse: w = self.o.inv(w, self.field)
        for c in range(ncols):
            col = self.o.ntt(self.field, np.ascontiguousarray(a[:, c, :]), inverse=inverse)
            if inverse:   # undo the per-transform n^-1: the building block leaves scaling to phase B
                col = np.array([self.o.mul(x, self.o.from_u64(1 << log_rows, self.field), self.field) for x in col])
            wc = self.o.pow(w, col0 + c, self.field)
            tw = self.o.from_u64(1, self.field)
            for k in range(1 << log_rows):
                a[k, c, :] = self.o.mul(col[k], tw, self.field); tw = self.o.mul(tw, wc, self.field)

    def ntt_columns_coset(self, slab, log_rows, ncols, col0, log_n, shift4):
        # x[j] *= shift^j with j = row * C + global column, then the plain forward column phase
        a = self._np(slab).reshape(1 << log_rows, ncols, 4); C = 1 << (log_n - log_rows)
        for c in range(ncols):
            base = self.o.pow(shift4, col0 + c, self.field); step = self.o.pow(shift4, C, self.field); cur = base
            for r in range(1 << log_rows):
                a[r, c, :] = self.o.mul(a[r, c, :], cur, self.field); cur = self.o.mul(cur, step, self.field)
        self.ntt_columns(slab, log_rows, ncols, col0, log_n, False)

    def ntt_rows_coset(self, src, out, nrows, log_cols, row0, log_n, shift4):
        # the definition: dst[i][m] = w_n^(k1 m) * NTT_C(src[i][k'] * shift^(k' R + k1))[m], k1 = row0 + i, R = 2^(log_n - log_cols)
        o, f = self.o, self.field
        a = self._np(src).reshape(nrows, 1 << log_cols, 4); d = self._np(out).reshape(nrows, 1 << log_cols, 4)
        R = 1 << (log_n - log_cols); w = o.root_of_unity(log_n, f); step = o.pow(shift4, R, f)
        for i in range(nrows):
            k1 = row0 + i; cur = o.pow(shift4, k1, f); x = np.zeros_like(a[i])
            for kp in range(1 << log_cols):
                x[kp] = o.mul(a[i, kp], cur, f); cur = o.mul(cur, step, f)
            y = o.ntt(f, x)
            wk = o.pow(w, k1, f); tw = o.from_u64(1, f)
            for m in range(1 << log_cols):
                d[i, m] = o.mul(y[m], tw, f); tw = o.mul(tw, wk, f)

    def interleave(self, dst, src, stride, offset):
        dst.view(-1, stride, 4)[:, offset, :] = src

    def ntt_rows(self, slab, nrows, log_cols, inverse, scale4=None):
        a = self._np(slab).reshape(nrows, 1 << log_cols, 4)
        for r in range(nrows):
            row = self.o.ntt(self.field, np.ascontiguousarray(a[r]), inverse=inverse)
            if inverse: row = np.array([self.o.mul(x, self.o.from_u64(1 << log_cols, self.field), self.field) for x in row])
            if scale4 is not None: row = np.array([self.o.mul(x, scale4, self.field) for x in row])
            a[r] = row

    def merkle_build(self, params, arity, tree_label, leaves, n, first_pos, level0, stop_at_len):
        t = 9 if arity <= 8 else 17
        if t not in self.hparams: self.hparams[t] = self.hc.params(0, t)
        cur = self._np(leaves).reshape(-1, 4).copy(); levels = [cur]; pos, level = first_pos, level0
        while cur.shape[0] > max(stop_at_len, 1):
            assert pos % arity == 0
            pos //= arity
            cur = self.hc.hash_ds_level(self.hparams[t], 0, arity, level, pos, tree_label, cur); levels.append(cur); level += 1
        return levels

    def merkle_last_level(self, h): return torch.from_numpy(h[-1].view(np.int64).copy()), len(h)
    def merkle_free(self, h): pass
    def sync(self): pass

    # ---- block-local steps of the sharded prover (same method names as dist.HipProvider) ----------------
    def _t(self, a): return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint64).view(np.int64).copy())
    def params_for_arity(self, arity): return None
    def new(self, n): return torch.empty((n, 4), dtype=torch.int64)
    def zeros(self, n): return torch.zeros((n, 4), dtype=torch.int64)
    def fri_sample_z(self, seed_z, level, n): return self.o.fri_sample_z_ell(seed_z, level, n)
    def fold(self, f, z, m): return self._t(self.o.fri_fold_layer(self._np(f), z, m))
    def leaf_pair_hash(self, f, f_next, m): return self._t(self.o.leaf_pair_hash(self._np(f), None if f_next is None else self._np(f_next.contiguous()), m))
    def merkle_build_pairs(self, params, arity, tree_label, f, cp, n):
        t = self.o.merkle_build(arity, tree_label, self._np(f), self._np(cp)); lv = [t.level(v) for v in range(t.num_levels())]; t.free(); return lv
    def merkle_num_levels(self, h): return len(h)
    def merkle_level_len(self, h, lvl): return h[lvl].shape[0]
    def merkle_gather(self, h, lvl, idx): return h[lvl][np.asarray(idx, dtype=np.int64)]
    def column_digest(self, tag, col): return self.o.tr_hash_fields_tagged(tag, self._np(col))
    def ali_challenges(self, digests, n0):
        # ali_sample_z_beta_fs (fri.rs:511-533) from primitives: seed = H("ALI/seed", digests || n0); ChaCha12 seeded by H("ALI/DEEP", seed || n0)
        o = self.o
        seed = o.tr_hash_fields_tagged(b"ALI/seed", np.vstack([digests, o.from_u64(n0)[None, :]]))
        fused = o.tr_hash_fields_tagged(b"ALI/DEEP", np.vstack([seed[None, :], o.from_u64(n0)[None, :]]))
        u = self.hc.chacha12_u64s(o.to_bytes_le(fused), 64)
        beta = o.from_u64(int(u[0])); one = o.from_u64(1); z = None
        for x in u[1:]:
            c = o.from_u64(int(x))
            if int(x) != 0 and not (o.pow(c, n0) == one).all(): z = c; break
        return np.stack([seed, z, beta])
    def ali_merge_shard(self, a, s, e, t, z, j0, n_global):
        o = self.o; a, s, e, t = (self._np(x) for x in (a, s, e, t)); w = o.domain_omega(n_global); out = np.zeros_like(a)
        wj = o.pow(w, j0)
        for i in range(a.shape[0]):
            phi = o.sub(o.add(o.mul(a[i], s[i]), e[i]), t[i])
            out[i] = o.mul(phi, o.inv(o.sub(wj, z))); wj = o.mul(wj, w)
        return self._t(out)
    def query_plan(self, roots, n0, schedule, r):
        if not hasattr(self, "tparams"): self.tparams = self.hc.params(1)
        return self.hc.fri_plan(self.tparams, roots, n0, schedule, r)


def _worker(rank, world, port, log_n, log_rows, inverse, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port); os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from stark_mlwe_amd import dist as sd
        prov = CpuStandIn()
        o = prov.o
        n = 1 << log_n
        x = o.synth_column(321, 7, 0, n)
        plan = sd.DistNtt(prov, log_n, log_rows, inverse=inverse)
        idx = plan.local_input_indices().reshape(-1).numpy()
        slab = torch.from_numpy(x[idx].view(np.int64).copy())
        scale = o.inv(o.from_u64(n)) if inverse else None
        rows = plan.forward(slab, scale)
        want = o.ntt(0, x, inverse=inverse)
        got = rows.numpy().view(np.uint64)
        ok_t = bool((got == want[plan.local_output_indices().reshape(-1).numpy()]).all())
        nat = plan.to_natural_blocks(rows).numpy().view(np.uint64)
        ok_n = bool((nat == want[rank * n // world:(rank + 1) * n // world]).all())
        # sharded Merkle: 2 x 256 leaves, arity 16 -> each rank climbs to 1 digest, top level of 2 is gathered
        leaves = o.synth_column(5, 1, 0, 512)
        mine = torch.from_numpy(leaves[rank * 256:(rank + 1) * 256].view(np.int64).copy())
        root = sd.merkle_sharded_root(prov, None, 16, 9, mine, 256).numpy().view(np.uint64)
        tree = o.merkle_build(16, 9, leaves)
        ok_m = bool((root == tree.root()).all())
        # ragged climb: 2 x 48 leaves, arity 16: shards stop at 3 digests each, 6 are gathered
        leaves2 = o.synth_column(6, 1, 0, 96)
        mine2 = torch.from_numpy(leaves2[rank * 48:(rank + 1) * 48].view(np.int64).copy())
        root2 = sd.merkle_sharded_root(prov, None, 16, 3, mine2, 48).numpy().view(np.uint64)
        ok_m2 = bool((root2 == o.merkle_build(16, 3, leaves2).root()).all())
        q.put((rank, ok_t, ok_n, ok_m, ok_m2))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("log_n,log_rows,inverse", [(8, 4, False), (9, 3, False), (8, 4, True)])
def test_six_step_ntt_and_sharded_merkle_world2(log_n, log_rows, inverse):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + log_n * 3 + log_rows + (7 if inverse else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, log_n, log_rows, inverse, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs: p.join(60)
    assert sorted(res) == [(0, True, True, True, True), (1, True, True, True, True)], res


def _prove_worker(rank, world, port, log_n0, schedule, r, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port); os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from stark_mlwe_amd import dist as sd
        prov = CpuStandIn(); o = prov.o
        n0 = 1 << log_n0; nl = n0 // world
        cols = [o.synth_column(0x5EED0000 + log_n0, c, 0, n0) for c in range(4)]
        mine = [prov._t(c[rank * nl:(rank + 1) * nl]) for c in cols]
        dp = sd.DistProver(prov, n0, schedule, r, 0xDEEFBAAD)
        proof, est = dp.prove(*mine)
        ref = o.deep_fri_prove(*cols, n0, schedule, r, 0xDEEFBAAD)
        want = ref.bytes(); west = ref.size_estimate(); ref.free()
        sharded = [bool(l.sharded) for l in dp.layers] if dp.layers else None
        q.put((rank, proof == want, est == west, o.deep_fri_verify(proof, schedule, r, 0xDEEFBAAD)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("log_n0,schedule,r,world", [(9, [16, 8], 6, 2), (10, [8, 8, 8], 5, 2), (7, [16, 8], 3, 2), (10, [8, 8, 8], 4, 4)])
def test_sharded_prove_world2_matches_reference_bytes(log_n0, schedule, r, world):
    """One trace block-sharded over 2 (and 4) ranks -> the SAME canonical proof bytes as the oracle's prove of the whole trace."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + log_n0 + 10 * world
    procs = [ctx.Process(target=_prove_worker, args=(r_, world, port, log_n0, schedule, r, q)) for r_ in range(world)]
    for p in procs: p.start()
    res = [q.get(timeout=400) for _ in procs]
    for p in procs: p.join(60)
    assert sorted(res) == [(r_, True, True, 1) for r_ in range(world)], res


def _lde_worker(rank, world, port, log_n, log_rows, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port); os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from stark_mlwe_amd import dist as sd
        prov = CpuStandIn(); o = prov.o
        n, lb = 1 << log_n, 2
        nl = n // world
        ev = o.synth_column(77, 0, 0, n)
        lde = sd.ShardedLde(prov, log_n, lb, 5, log_rows)
        sd.STATS["all_to_all"] = 0
        out = lde(prov._t(ev[rank * nl:(rank + 1) * nl])).numpy().view(np.uint64)
        want = o.lde(0, ev, lb, o.from_u64(5))
        ok_lde = bool((out == want[rank * nl * 4:(rank + 1) * nl * 4]).all())
        ok_lde = ok_lde and sd.STATS["all_to_all"] == lde.n_all_to_all == 4      # counted exchanges per column (at most 6)
        # the chained bench step: LDE of four columns -> merge -> sharded commit; roots must equal the oracle's for the whole trace
        cols = [o.synth_column(78, c, 0, n) for c in range(4)]
        z = o.from_u64(0xC0FFEE); coset = o.from_u64(5); sched = [8, 4]
        job = sd.ShardedTrace(prov, log_n, lb, sched, 0xDEEFBAAD, coset, z)
        roots = job.step([prov._t(c[rank * nl:(rank + 1) * nl]) for c in cols])
        ext = [o.lde(0, c, lb, coset) for c in cols]
        f0, _ = o.ali_merge(ext[0], ext[1], ext[2], ext[3], o.domain_omega(n << lb), z, want_c_star=False)
        ref = o.deep_fri_prove(None, None, None, None, n << lb, sched, 1, 0xDEEFBAAD, f0=f0)
        ok_roots = all(bool((np.asarray(roots[l]).view(np.uint64).reshape(4) == ref.root(l)).all()) for l in range(len(sched) + 1)); ref.free()
        q.put((rank, ok_lde, ok_roots, "all-to-all" in job.describe()))
    except Exception as ex:      # noqa: BLE001
        import traceback
        q.put(("error", repr(ex), traceback.format_exc()[-1500:]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("log_n,log_rows,world", [(7, 3, 2), (8, 4, 2), (8, 4, 4)])
def test_sharded_lde_and_chained_step_world2(log_n, log_rows, world):
    """north_star's multi-GPU split on 2 and 4 ranks (gloo): the LDE of a block-sharded column through the six-step NTTs equals the
    oracle's LDE of the whole column, and LDE -> merge -> sharded commit gives the oracle's roots for the whole trace."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + log_n + 10 * world
    procs = [ctx.Process(target=_lde_worker, args=(r_, world, port, log_n, log_rows, q)) for r_ in range(world)]
    for p in procs: p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs: p.join(60)
    assert sorted(res) == [(r_, True, True, True) for r_ in range(world)], res


def test_sharded_stop_len():
    sys.path.insert(0, ROOT)
    from stark_mlwe_amd.dist import sharded_stop_len
    assert sharded_stop_len(1 << 20, 16) == 1          # 2^20 leaves per rank, arity 16: climbs to a single digest
    assert sharded_stop_len(1 << 21, 16) == 2
    assert sharded_stop_len(48, 16) == 3
    assert sharded_stop_len(1 << 10, 8) == 2
