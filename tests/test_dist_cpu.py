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


def _worker(rank, world, port, log_n, log_rows, inverse, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
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


def test_sharded_stop_len():
    sys.path.insert(0, ROOT)
    from stark_mlwe_amd.dist import sharded_stop_len
    assert sharded_stop_len(1 << 20, 16) == 1          # 2^20 leaves per rank, arity 16: climbs to a single digest
    assert sharded_stop_len(1 << 21, 16) == 2
    assert sharded_stop_len(48, 16) == 3
    assert sharded_stop_len(1 << 10, 8) == 2
