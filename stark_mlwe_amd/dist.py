"""Multi-GPU orchestration of the path: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests) for the exchanges, the C-ABI for all local compute.

Only these steps of the path exchange data:

* the six-step NTT — ONE all-to-all (the row/column transpose between its two local phases);
* the top of a sharded Merkle tree — an all-gather of a few digests;
* `DistProver` (one trace block-sharded over the ranks, one proof): the column sponges of build_f0 need
  whole columns (gather of column c to rank c mod W — the four chains then run task-parallel), a layer that
  has become too small to shard is all-gathered once, and the query phase collects the few thousand opened
  values from their owners with one all-reduce of a zero-filled table.

Everything else (folds, leaf hashes, lower Merkle levels, the DEEP-ALI merge) is block-local with global
indices.  The local compute is behind a small provider interface so that the same orchestration runs in
the world_size-2 `gloo` tests on CPU (provider = test stand-in, tests/ only) and on GPUs
(`HipProvider`, the product).  Nothing here computes field arithmetic.

Distributions (n = R * C elements, W ranks, R = 2^log_rows <= 1024 rows, C = n / R columns):
  input  of `DistNtt.forward`: rank r holds the column block  x[j1*C + c],  c in [r*C/W, (r+1)*C/W),
          all j1, as a row-major [R][C/W] slab  (a block-cyclic view of the natural-order vector);
  output: rank q holds the rows k1 in [q*R/W, (q+1)*R/W) of Z[k1][k'] = X[k1 + R*k'] as [R/W][C].
A consumer that needs natural block order applies one more all-to-all (`DistNtt.to_natural_blocks`).
"""
import ctypes as C
import os

import torch
import torch.distributed as dist


class TorchComm:
    """Collectives through torch.distributed: "nccl" (= RCCL) on GPUs, "gloo" in the CPU tests / one-GPU rehearsals
    (gloo stages device tensors through host memory)."""

    def world(self):
        return (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)

    def all_to_all(self, send: torch.Tensor) -> torch.Tensor:
        rank, W = self.world()
        if W == 1:
            return send.clone()
        recv = torch.empty_like(send)
        if dist.get_backend() == "nccl":
            dist.all_to_all_single(recv.view(-1), send.contiguous().view(-1))     # RCCL: one message per peer link
            return recv
        # gloo has no all_to_all: W-1 point-to-point pairs, staged through host memory (tests / one-GPU rehearsal only)
        hs = send.cpu() if send.is_cuda else send
        hr = torch.empty_like(hs)
        hr[rank].copy_(hs[rank])
        ops = []
        for p in range(W):
            if p != rank:
                ops.append(dist.P2POp(dist.isend, hs[p].contiguous(), p))
                ops.append(dist.P2POp(dist.irecv, hr[p], p))
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        recv.copy_(hr)
        return recv

    def all_gather(self, x: torch.Tensor) -> torch.Tensor:
        rank, W = self.world()
        if W == 1:
            return x.clone()
        if dist.get_backend() == "nccl" and not x.is_cuda:      # RCCL moves device memory only
            return self.all_gather(x.cuda()).cpu()
        if dist.get_backend() != "nccl" and x.is_cuda:     # gloo: stage through host memory
            hx = x.cpu(); out = [torch.empty_like(hx) for _ in range(W)]
            dist.all_gather(out, hx.contiguous())
            return torch.cat(out, dim=0).to(x.device)
        out = [torch.empty_like(x) for _ in range(W)]
        dist.all_gather(out, x.contiguous())
        return torch.cat(out, dim=0)

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        rank, W = self.world()
        if W == 1:
            return t
        if dist.get_backend() == "nccl":
            d = t.cuda() if not t.is_cuda else t.clone()
            dist.all_reduce(d)
            return d if t.is_cuda else d.cpu()
        h = t.cpu().contiguous(); dist.all_reduce(h)
        return h.to(t.device) if t.is_cuda else h

    def gather_to(self, x: torch.Tensor, dst: int):
        rank, W = self.world()
        if W == 1:
            return x
        if dist.get_backend() != "nccl" and x.is_cuda:
            hx = x.cpu(); parts = [torch.empty_like(hx) for _ in range(W)] if rank == dst else None
            dist.gather(hx, parts, dst=dst)
            return torch.cat(parts, dim=0).to(x.device) if rank == dst else None
        parts = [torch.empty_like(x) for _ in range(W)] if rank == dst else None
        dist.gather(x.contiguous(), parts, dst=dst)
        return torch.cat(parts, dim=0) if rank == dst else None


class LibComm:
    """Collectives through the library's own RCCL communicator (stark_comm_*, include/stark_mlwe.h): enqueued on the
    context's stream, no host synchronisation, usable by a host that has no torch.  torch.distributed is needed only to hand
    the 128-byte unique id from rank 0 to its peers (any transport would do)."""

    def __init__(self, ctx, rank=None, nranks=None):
        import numpy as np
        self.ctx, self.lib = ctx, ctx.lib
        if rank is None:
            rank, nranks = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
        self.rank, self.W = rank, nranks
        idb = np.zeros(128, np.uint8)
        if rank == 0:
            ctx._chk(self.lib.stark_comm_unique_id(idb.ctypes.data_as(C.c_void_p)))
        if nranks > 1:
            obj = [idb.tobytes()]
            dist.broadcast_object_list(obj, src=0)
            idb = np.frombuffer(obj[0], np.uint8).copy()
        ctx._chk(self.lib.stark_comm_init(ctx.h, nranks, rank, idb.ctypes.data_as(C.c_void_p)))

    def world(self):
        return self.rank, self.W

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr())

    # Same stream discipline as HipProvider._run (include/stark_mlwe.h "Stream rule"): the collectives run on the CONTEXT's
    # stream while their tensors are produced and consumed by torch on ITS current stream.  One stream orders both only when the
    # two are the same (bench.py's shared stream; Context(stream=None) with torch on the default stream); a STREAM_PRIVATE context,
    # or torch moved to another stream, needs the explicit bracket — otherwise RCCL reads inputs that are not written yet and the
    # consumer reads results that are not reduced yet.
    def _shared(self, dev):
        if self.ctx.private_stream:
            return False
        return self.ctx.stream_handle == torch.cuda.current_stream(dev).cuda_stream

    def _run(self, dev, fn, *args):
        shared = self._shared(dev)
        if not shared:
            torch.cuda.current_stream(dev).synchronize()
        self.ctx._chk(fn(*args))
        if not shared:
            self.ctx.sync()

    def all_to_all(self, send):
        send = send.contiguous(); recv = torch.empty_like(send)
        self._run(send.device, self.lib.stark_comm_all_to_all_dev, self.ctx.h, self._p(send), self._p(recv), send.numel() * send.element_size() // self.W)
        return recv

    def all_gather(self, x):
        dev = x if x.is_cuda else x.cuda()
        dev = dev.contiguous(); out = torch.empty((self.W * dev.shape[0],) + tuple(dev.shape[1:]), dtype=dev.dtype, device=dev.device)
        self._run(dev.device, self.lib.stark_comm_all_gather_dev, self.ctx.h, self._p(dev), self._p(out), dev.numel() * dev.element_size())
        return out if x.is_cuda else out.cpu()

    def all_reduce_sum(self, t):
        d = (t if t.is_cuda else t.cuda()).contiguous().clone()
        self._run(d.device, self.lib.stark_comm_all_reduce_u64_dev, self.ctx.h, self._p(d), self._p(d), d.numel() * d.element_size() // 8)
        return d if t.is_cuda else d.cpu()

    def gather_to(self, x, dst):
        x = x.contiguous()
        out = torch.empty((self.W * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device) if self.rank == dst else None
        self._run(x.device, self.lib.stark_comm_gather_dev, self.ctx.h, self._p(x), None if out is None else self._p(out), x.numel() * x.element_size(), dst)
        return out

    def close(self):
        self.lib.stark_comm_destroy(self.ctx.h)


def checked_lib_comm(ctx, rank, nranks, dev, all_ranks_can_bind=True):
    """A LibComm that has passed one all-to-all and one all-gather against torch.distributed on the same bytes, or None — decided by ALL ranks
    together (all_reduce MIN), so that every rank ends up on the same transport.  The multi-rank branch of stark_comm_* cannot run on a
    one-GPU box; a job checks it once at start-up before its data path depends on it.  Returns (comm or None, a line for the log)."""
    if not all_ranks_can_bind:
        return None, "torch.distributed RCCL (library communicator unavailable on some rank)"
    lc, ok = None, 0
    try:
        lc = LibComm(ctx, rank, nranks)
        probe = (torch.arange(nranks * 256 * 4, dtype=torch.int64, device=dev) * 0x9E3779B97F4A7C15 + rank * 0x1234567).view(nranks * 256, 4)
        want = torch.empty_like(probe)
        dist.all_to_all_single(want.view(-1), probe.contiguous().view(-1))
        got = lc.all_to_all(probe)
        gath = lc.all_gather(probe[:4])
        ok = int(bool((got == want).all()) and bool((gath[4 * rank:4 * rank + 4] == probe[:4]).all()))
    except Exception as ex:      # noqa: BLE001 — reported, the job continues on torch.distributed
        import sys
        sys.stderr.write(f"[stark_mlwe_amd.dist] rank {rank}: library communicator failed its start-up check: {ex!r}\n")
    flag = torch.tensor([ok], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        return lc, "library RCCL communicator (stark_comm_*), checked against torch.distributed at start-up"
    if lc is not None:
        lc.close()
    return None, "torch.distributed RCCL (library communicator failed its start-up check on some rank)"


_COMM = TorchComm()


def set_comm(comm):
    """Route every collective of this module through `comm` (TorchComm by default; LibComm for the library's RCCL)."""
    global _COMM
    _COMM = comm if comm is not None else TorchComm()
    return _COMM


def world():
    return _COMM.world()


STATS = {"all_to_all": 0}          # exchanges issued through this module (tests and bench.py report the counted number, not a formula)


def exchange_all_to_all(send: torch.Tensor) -> torch.Tensor:
    """send: [W, chunk...] — chunk q goes to rank q.  Returns recv with chunk p = what rank p sent here."""
    STATS["all_to_all"] += 1
    return _COMM.all_to_all(send)


def all_gather_rows(x: torch.Tensor) -> torch.Tensor:
    return _COMM.all_gather(x)


class HipProvider:
    """Local compute through libstark_mlwe_hip.so on torch CUDA tensors of shape [rows, 4] (int64 view of the limbs)."""

    def __init__(self, ctx, field=0, device="cuda"):
        self.ctx, self.lib, self.field, self.device = ctx, ctx.lib, field, device

    # Stream discipline (include/stark_mlwe.h "Stream rule").  The orchestration mixes torch device ops (slices, zeros, index
    # gathers, staging copies, RCCL) with library kernels.  If the context runs ON torch's current stream (bench.py hands it a
    # dedicated torch stream; Context(stream=None) is the legacy default stream, which is also torch's default), one stream
    # orders everything.  Otherwise (a STREAM_PRIVATE context, or torch moved to another stream) every library call is
    # bracketed by explicit synchronisation.
    def _shared(self):
        cur = torch.cuda.current_stream(self.device).cuda_stream
        if self.ctx.private_stream:
            return False
        return self.ctx.stream_handle == cur

    def _run(self, fn, *args):
        shared = self._shared()
        if not shared:
            torch.cuda.current_stream(self.device).synchronize()       # torch-side producers of the arguments are done
        self.ctx._chk(fn(*args))
        if not shared:
            self.ctx.sync()                                            # results are visible to torch-side consumers

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr())

    def ntt_columns(self, slab, log_rows, ncols, col0, log_n, inverse):
        self._run(self.lib.stark_ntt_columns_dev, self.ctx.h, self.field, self._p(slab), log_rows, ncols, col0, log_n, int(inverse))

    def ntt_rows(self, slab, nrows, log_cols, inverse, scale4=None):
        from .api import _ptr
        self._run(self.lib.stark_ntt_rows_dev, self.ctx.h, self.field, self._p(slab), nrows, log_cols, int(inverse), _ptr(scale4))

    def ntt_columns_coset(self, slab, log_rows, ncols, col0, log_n, shift4):
        self._run(self.lib.stark_ntt_columns_coset_dev, self.ctx.h, self.field, self._p(slab), log_rows, ncols, col0, log_n, _npp(shift4))

    def ntt_rows_coset(self, src, out, nrows, log_cols, row0, log_n, shift4):
        """First local phase of a forward coset transform on the inverse's transposed output (stark_ntt_rows_coset_dev): src -> out."""
        self._run(self.lib.stark_ntt_rows_coset_dev, self.ctx.h, self.field, self._p(src), self._p(out), nrows, log_cols, row0, log_n, _npp(shift4))

    def lde_sharded(self, block, log_n, log_blowup, shift4):
        """The whole sharded LDE of one column inside the library (stark_lde_sharded_dev): its exchanges run on the library's communicator."""
        out = torch.empty((block.shape[0] << log_blowup, 4), dtype=torch.int64, device=block.device)
        self._run(self.lib.stark_lde_sharded_dev, self.ctx.h, self.field, self._p(block), log_n, log_blowup, _npp(shift4), self._p(out))
        return out

    def permute3(self, src, dims, perm):
        """Contiguous [dims[perm[0]], dims[perm[1]], dims[perm[2]], 4] copy of the [d0][d1][d2] array `src` (hand-written pack kernel)."""
        out = torch.empty((dims[perm[0]] * dims[perm[1]] * dims[perm[2]], 4), dtype=torch.int64, device=src.device)
        self._run(self.lib.stark_permute3_dev, self.ctx.h, self._p(src), self._p(out), dims[0], dims[1], dims[2], perm[0], perm[1], perm[2])
        return out

    def interleave(self, dst, src, stride, offset):
        self._run(self.lib.stark_interleave_dev, self.ctx.h, self._p(src), self._p(dst), src.shape[0], stride, offset)

    def pow_small(self, base_int, e):
        """Montgomery limbs of base^e in the provider's field (host scalar)."""
        return _mont(pow(base_int, e, _P[self.field]), self.field)

    def root_of_unity(self, log_n):
        return _mont(pow(_GEN[self.field], (_P[self.field] - 1) >> log_n, _P[self.field]), self.field)

    def merkle_build(self, params, arity, tree_label, leaves, n, first_pos, level0, stop_at_len):
        h = C.c_void_p()
        self._run(self.lib.stark_merkle_build_dev, self.ctx.h, params.h, arity, tree_label, self._p(leaves), n, 0, None, first_pos, level0, stop_at_len, C.byref(h))
        return h

    def merkle_last_level(self, h):
        """(device tensor [k, 4] copy of the last level, number of levels)."""
        nl = self.lib.stark_merkle_num_levels(h)
        k = self.lib.stark_merkle_level_len(h, nl - 1)
        out = torch.empty((k, 4), dtype=torch.int64, device=self.device)
        src = self.lib.stark_merkle_level_dev(h, nl - 1)
        # device-to-device on the context's stream (the pack kernel with stride 1 is a plain copy): the tree top never visits the host
        self._run(self.lib.stark_interleave_dev, self.ctx.h, C.c_void_p(src), self._p(out), k, 1, 0)
        return out, nl

    def merkle_free(self, h):
        self.lib.stark_merkle_free(h)

    def sync(self):
        self.ctx.sync()

    # ---- the block-local steps of one sharded proof (DistProver) ------------------------------------------
    def params_for_arity(self, arity):
        return self.ctx.poseidon_params_for_arity(arity)

    def new(self, n):
        return torch.empty((n, 4), dtype=torch.int64, device=self.device)

    def zeros(self, n):
        return torch.zeros((n, 4), dtype=torch.int64, device=self.device)

    def fri_sample_z(self, seed_z, level, n):
        return self.ctx.fri_sample_z_ell(seed_z, level, n)

    def fold(self, f, z, m):
        n = f.shape[0]
        out = self.new(n // m)
        self._run(self.lib.stark_fri_fold_dev, self.ctx.h, self._p(f), n, _npp(z), m, self._p(out))
        return out

    def leaf_pair_hash(self, f, f_next, m):
        out = self.new(f.shape[0])
        self._run(self.lib.stark_leaf_pair_hash_dev, self.ctx.h, self.ctx.transcript_params().h, self._p(f), None if f_next is None else self._p(f_next), f.shape[0], m, self._p(out))
        return out

    def merkle_build_pairs(self, params, arity, tree_label, f, cp, n):
        h = C.c_void_p()
        self._run(self.lib.stark_merkle_build_dev, self.ctx.h, params.h, arity, tree_label, self._p(f), n, 1, self._p(cp), 0, 0, 0, C.byref(h))
        return h

    def merkle_num_levels(self, h):
        return self.lib.stark_merkle_num_levels(h)

    def merkle_level_len(self, h, lvl):
        return self.lib.stark_merkle_level_len(h, lvl)

    def merkle_gather(self, h, lvl, idx):
        """numpy (k, 4) uint64: nodes idx[] of level lvl (a few digests: host hop)."""
        import numpy as np
        ix = np.ascontiguousarray(idx, dtype=np.uint64)
        out = np.zeros((len(ix), 4), np.uint64)
        self._run(self.lib.stark_merkle_gather, h, lvl, ix.ctypes.data_as(C.c_void_p), len(ix), out.ctypes.data_as(C.c_void_p))
        return out

    def column_digest(self, tag: bytes, col):
        """tr_hash_fields_tagged(tag, column) — the serial sponge of build_f0 (fri.rs:551-554); numpy (4,)."""
        out = self.new(1)
        self._run(self.lib.stark_tr_hash_fields_tagged_dev, self.ctx.h, None, tag, self._p(col), col.shape[0], 1, self._p(out))
        self.ctx.sync()
        return out.cpu().numpy().view("uint64")[0]

    def ali_challenges(self, digests, n0):
        return self.ctx.ali_challenges(digests, n0)

    def ali_merge_shard(self, a, s, e, t, z, j0, n_global):
        f0 = self.new(a.shape[0])
        self._run(self.lib.stark_ali_merge_shard_dev, self.ctx.h, self._p(a), self._p(s), self._p(e), self._p(t), None, None, None, _npp(z), a.shape[0], j0, n_global, self._p(f0), None)
        return f0

    def query_plan(self, roots, n0, schedule, r):
        return self.ctx.fri_query_plan(roots, n0, schedule, r)


_P = {0: 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001, 1: 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001}
_GEN = {0: 5, 1: 7}          # multiplicative generators


def _mont(x, field=0):
    import numpy as np
    m = (x << 256) % _P[field]
    return np.array([(m >> (64 * i)) & (2**64 - 1) for i in range(4)], np.uint64)


def _unmont(a, field=0):
    v = sum(int(a[i]) << (64 * i) for i in range(4))
    return v * pow(1 << 256, -1, _P[field]) % _P[field]


def _npp(a):
    import numpy as np
    a = np.ascontiguousarray(a, dtype=np.uint64)
    _npp.keep = a                      # keep the buffer alive across the call that follows
    return a.ctypes.data_as(C.c_void_p)


class DistNtt:
    """Six-step NTT of size 2^log_n over the ranks of the default process group."""

    def __init__(self, provider, log_n, log_rows=None, inverse=False):
        self.p, self.log_n, self.inverse = provider, log_n, inverse
        self.rank, self.W = world()
        if log_rows is None:
            log_rows = min(10, log_n // 2)
        self.log_rows, self.R = log_rows, 1 << log_rows
        self.C = 1 << (log_n - log_rows)
        if self.R % self.W or self.C % self.W:
            raise ValueError("rows and columns must divide over the ranks")
        self.ncl = self.C // self.W          # local columns in phase A
        self.nrl = self.R // self.W          # local rows in phase B

    def local_input_indices(self):
        """Natural-order indices j of the elements this rank holds on input, as a [R, C/W] index grid."""
        j1 = torch.arange(self.R).view(-1, 1)
        c = torch.arange(self.ncl).view(1, -1) + self.rank * self.ncl
        return j1 * self.C + c

    def local_output_indices(self):
        """Natural-order indices k of the elements this rank holds on output, as a [R/W, C] grid."""
        k1 = torch.arange(self.nrl).view(-1, 1) + self.rank * self.nrl
        kp = torch.arange(self.C).view(1, -1)
        return k1 + self.R * kp

    def _perm(self, t, dims, perm):
        if hasattr(self.p, "permute3"):
            return self.p.permute3(t, dims, perm)
        return t.view(*dims, 4).permute(*perm, 3).contiguous().view(-1, 4)

    def forward(self, slab: torch.Tensor, scale4=None, shift4=None) -> torch.Tensor:
        """slab: [R * C/W, 4] (row-major [R][C/W]).  Returns [R/W * C, 4] (row-major [R/W][C]).
        shift4 (forward only): evaluate on the coset shift * <w> (x[j] *= shift^j on load, j the natural index)."""
        R, ncl, nrl, W = self.R, self.ncl, self.nrl, self.W
        # phase A: column NTTs of size R on the local column block + twiddle w_N^(col_global * k1)
        if shift4 is not None:
            self.p.ntt_columns_coset(slab, self.log_rows, ncl, self.rank * ncl, self.log_n, shift4)
        else:
            self.p.ntt_columns(slab, self.log_rows, ncl, self.rank * ncl, self.log_n, self.inverse)
        # the one exchange: rows k1 of block q go to rank q (contiguous [R/W][C/W] chunks of the slab); stream-ordered, no host sync
        recv = exchange_all_to_all(slab.view(W, nrl * ncl, 4))                   # [W(src p), R/W, C/W]
        rows = self._perm(recv.view(-1, 4), (W, nrl, ncl), (1, 0, 2))             # [R/W][W][C/W] = [R/W][C]
        # phase B: R/W contiguous NTTs of size C
        self.p.ntt_rows(rows, nrl, self.log_n - self.log_rows, self.inverse, scale4)
        return rows

    def to_natural_blocks(self, rows: torch.Tensor) -> torch.Tensor:
        """Second all-to-all: from the transposed output to natural order, block-sharded (n/W contiguous)."""
        W, nrl, Cc = self.W, self.nrl, self.C
        # rank q holds X[k1 + R*k'] for its k1 block; natural block b holds k in [b*n/W, (b+1)*n/W) <=> k' in [b*C/W, (b+1)*C/W)
        send = self._perm(rows, (nrl, W, Cc // W), (1, 2, 0))                     # [dst b][k' local][k1 local]
        recv = exchange_all_to_all(send.view(W, -1, 4))                           # [src q][k' local][k1 local]
        return self._perm(recv.view(-1, 4), (W, Cc // W, nrl), (1, 0, 2))         # [k' local][k1 global]

    def from_natural_blocks(self, block: torch.Tensor) -> torch.Tensor:
        """Natural-order block [n/W] (= rows j1 in [rank*R/W, (rank+1)*R/W) of the [R][C] view) -> the input slab [R][C/W]."""
        W, nrl, ncl = self.W, self.nrl, self.ncl
        send = self._perm(block, (nrl, W, ncl), (1, 0, 2))                        # [dst r][j1 local][c]
        recv = exchange_all_to_all(send.view(W, -1, 4))                           # [src p][j1 local][c] = [R][C/W]
        return recv.view(-1, 4)


class ShardedLde:
    """LDE of ONE column of n = 2^log_n evaluations (natural order, block-sharded: rank q holds [q*n/W, (q+1)*n/W)) to
    N = n * 2^log_blowup evaluations on shift * <w_N>, natural order, block-sharded again.
        coefficients  c = iNTT_n(evals)
        out[b*k + s]  = NTT_n(c[j] * (shift * w_N^s)^j)[k],  s < b = 2^log_blowup     (b coset transforms of size n: the zero
                                                                            padding of the definition is never materialised)
    FOUR all-to-alls per column, whatever the blow-up (view the vector as [R][C], R <= 1024):
      1. natural row blocks -> column blocks [R][C/W]                      (the inverse's column phase needs whole columns)
      2. the inverse transform's own transpose -> rows k1 of c[k1 + R k'], [R/W][C], coefficients
         — this IS the input layout of a forward transform split the other way round (n = C x R): its first phase runs along the
         contiguous k' axis of the same slab (`ntt_rows_coset`: coset pre-scale, size-C transforms, inter-step twiddle), so nothing
         is exchanged between the inverse and the forward transforms;
      3. ONE exchange for the first-phase outputs of all b cosets -> [b][C/W][R], then plain size-R transforms over k1;
      4. ONE exchange of all b results into natural block order (the cosets are interleaved by the pack kernel on the way)."""

    def __init__(self, provider, log_n, log_blowup, shift_int=5, log_rows=None):
        self.p, self.log_n, self.lb = provider, log_n, log_blowup
        self.rank, self.W = world()
        self.inv = DistNtt(provider, log_n, log_rows, inverse=True)
        fld = getattr(provider, "field", 0)
        P = _P[fld]
        wN = pow(_GEN[fld], (P - 1) >> (log_n + log_blowup), P)
        self.shifts = [_mont(shift_int * pow(wN, s_, P) % P, fld) for s_ in range(1 << log_blowup)]
        self.ninv = _mont(pow(1 << log_n, -1, P), fld)
        self.n_all_to_all = 4

    def _perm(self, t, dims, perm):
        return self.inv._perm(t, dims, perm)

    def __call__(self, block: torch.Tensor) -> torch.Tensor:
        if hasattr(self.p, "lde_sharded") and (isinstance(_COMM, LibComm) or self.W == 1) and self.inv.log_rows == min(10, self.log_n // 2):
            # the same composition inside the library (one C-ABI call per column, what a host without Python uses); its four exchanges go through
            # the library's communicator, so this route needs LibComm (or a single rank)
            STATS["all_to_all"] += self.n_all_to_all
            return self.p.lde_sharded(block.contiguous(), self.log_n, self.lb, self.shifts[0])
        b, W, inv = 1 << self.lb, self.W, self.inv
        R, Cc, nrl, ncl = inv.R, inv.C, inv.nrl, inv.ncl
        coeff = inv.forward(inv.from_natural_blocks(block), self.ninv)                     # exchanges 1, 2: [R/W][C] = c[k1 + R k'], k1 local
        # first phase of every coset transform, local: P_s[k1][K1]
        first = self.p.new(b * nrl * Cc)
        for s_ in range(b):
            self.p.ntt_rows_coset(coeff, first[s_ * nrl * Cc:(s_ + 1) * nrl * Cc], nrl, self.log_n - inv.log_rows, self.rank * nrl, self.log_n, self.shifts[s_])
        # exchange 3: K1 block q of every coset and local row goes to rank q
        send = self._perm(first, (b * nrl, W, ncl), (1, 0, 2))                              # [dst q][s, k1 local][K1 local]
        recv = exchange_all_to_all(send.view(W, -1, 4))                                    # [src p][s][k1 local][K1 local]
        cols = self._perm(recv.view(-1, 4), (W, b, nrl * ncl), (1, 0, 2))                   # [s][p][k1 local][K1 local] = [s][k1][K1 local]
        rows = self._perm(cols, (b, R, ncl), (0, 2, 1))                                     # [s][K1 local][k1]
        self.p.ntt_rows(rows, b * ncl, inv.log_rows, False)                                 # second phase: Y_s[K1 + C K2], [s][K1 local][K2]
        # exchange 4: to natural blocks of the interleaved result, out[(K2 C + K1) b + s]; destination d owns K2 in [d R/W, (d+1) R/W)
        send = self._perm(rows, (b, ncl, R), (2, 1, 0))                                     # [K2][K1 local][s] — dst-major since K2 = d R/W + K2 local
        recv = exchange_all_to_all(send.view(W, -1, 4))                                    # [src p][K2 local][K1 local][s]
        return self._perm(recv.view(-1, 4), (W, nrl, ncl * b), (1, 0, 2))                   # [K2 local][p][K1 local][s] = natural order


def sharded_stop_len(n_local: int, arity: int) -> int:
    """Length of the shard's level at which climbing must stop: the first level whose local length is
    1 or no longer a multiple of the arity (its parents would need children from another rank)."""
    ln = n_local
    while ln > 1 and ln % arity == 0:
        ln //= arity
    return ln


def merkle_sharded_root(provider, params, arity, tree_label, leaves, n_local: int):
    """MerkleTree::new over W*n_local leaves, block-sharded: every rank builds the levels below its own
    n_local leaves with GLOBAL DS positions, the ranks all-gather the one level that crosses rank
    boundaries (W * stop_len digests — tens of bytes), and every rank finishes the top identically.
    Requires n_local to be a multiple of arity^levels_climbed (block alignment)."""
    rank, W = world()
    stop = sharded_stop_len(n_local, arity)
    h = provider.merkle_build(params, arity, tree_label, leaves, n_local, rank * n_local, 0, stop)
    top, nlev = provider.merkle_last_level(h)
    allv = all_gather_rows(top)
    if allv.shape[0] == 1:
        provider.merkle_free(h)
        return allv[0]
    ht = provider.merkle_build(params, arity, tree_label, allv, allv.shape[0], 0, nlev - 1, 1)
    root, _ = provider.merkle_last_level(ht)
    provider.merkle_free(ht)
    provider.merkle_free(h)
    return root[0]


# ---------------------------------------------------------------------------------------------------------
# One trace, block-sharded over the ranks, one proof (deep_fri_prove, fri.rs:601-641).
# ---------------------------------------------------------------------------------------------------------
def pick_arity_for_layer(n, m):
    """fri.rs:220-229 (index logic only)."""
    for a in (128, 64, 32, 16, 8, 4):
        if m >= a and n % a == 0:
            return a
    return 2 if n % 2 == 0 else 1


def hashed_arity(a):
    """fri.rs:275."""
    return a in (128, 64, 32, 16, 8)


def _allreduce_sum(t: torch.Tensor) -> torch.Tensor:
    """Sum of int64 tables in which every row is non-zero on exactly one rank (so the sum is a selection)."""
    return _COMM.all_reduce_sum(t)


def _gather_to(x: torch.Tensor, dst: int):
    """Concatenation of every rank's block on rank dst (None elsewhere)."""
    return _COMM.gather_to(x, dst)


class _Layer:
    """f_l and its commitment: either a contiguous block per rank (sharded) or a full copy on every rank."""
    __slots__ = ("n", "sharded", "f", "arity", "hashed", "tree", "top", "nlev_local", "local_lens")


class DistProver:
    """deep_fri_prove over W ranks, rank q holding rows [q*n0/W, (q+1)*n0/W) of a, s, e, t.  Every rank returns
    the same canonical proof bytes as a single-GPU (or the reference's) prove of the whole trace."""

    COLUMN_TAGS = (b"ALI/A", b"ALI/S", b"ALI/E", b"ALI/T")          # fri.rs:551-554

    def __init__(self, provider, n0, schedule, r, seed_z):
        self.p, self.n0, self.schedule, self.r, self.seed_z = provider, n0, list(schedule), r, seed_z
        self.rank, self.W = world()
        if n0 % self.W:
            raise ValueError("n0 must divide over the ranks")
        n = n0
        for m in self.schedule:                                       # fri.rs:150
            if m < 2 or n % m:
                raise ValueError("schedule not dividing domain size")
            n //= m
        self.layers = []
        self.timings = {}

    # -- build_f0 (DeepAliRealBuilder, fri.rs:535-569) -----------------------------------------------------
    def build_f0(self, a, s, e, t):
        import numpy as np
        rank, W, n0 = self.rank, self.W, self.n0
        nl = n0 // W
        digests = torch.zeros((4, 4), dtype=torch.int64)
        for c, col in enumerate((a, s, e, t)):                        # the four chains are independent: column c on rank c mod W
            whole = _gather_to(col, c % W)
            if rank == c % W:
                digests[c] = torch.from_numpy(self.p.column_digest(self.COLUMN_TAGS[c], whole).view(np.int64).copy())
            del whole
        digests = _allreduce_sum(digests)
        aux = self.p.ali_challenges(digests.numpy().view(np.uint64), n0)          # seed, z, beta — identical on every rank
        self.ali_aux = aux
        return self.p.ali_merge_shard(a, s, e, t, aux[1], rank * nl, n0)

    # -- fri_build_transcript (fri.rs:231-312) ------------------------------------------------------------
    def _shardable(self, l, n, prev_sharded):
        W, L = self.W, len(self.schedule)
        m = self.schedule[l] if l < L else 1
        ar = pick_arity_for_layer(n, m)
        if not prev_sharded or not hashed_arity(ar) or n % W:
            return False
        nl = n // W
        return nl % ar == 0 and (l == L or nl % m == 0)

    def commit(self, f0_local):
        import numpy as np
        p, rank, W, L = self.p, self.rank, self.W, len(self.schedule)
        self.layers = []
        sizes = [self.n0]
        for m in self.schedule:
            sizes.append(sizes[-1] // m)
        self.z = [p.fri_sample_z(self.seed_z, l, sizes[l]) for l in range(L)]     # challenges do not depend on commitments (fri.rs:250)
        # layers: fold while block-local, then all-gather once and continue replicated
        cur, sharded = f0_local, True
        for l in range(L + 1):
            lay = _Layer(); lay.n = sizes[l]
            m = self.schedule[l] if l < L else 1
            lay.arity = pick_arity_for_layer(lay.n, m); lay.hashed = hashed_arity(lay.arity)
            now = self._shardable(l, lay.n, sharded)
            if sharded and not now:                                    # transition: every rank gets the whole (small) layer
                cur = all_gather_rows(cur)
            sharded = now
            lay.sharded, lay.f = sharded, cur
            self.layers.append(lay)
            if l < L:
                cur = p.fold(cur, self.z[l], m)                        # block-local: m divides the local length (or the layer is whole)
        # commitments
        roots = []
        for l, lay in enumerate(self.layers):
            m = self.schedule[l] if l < L else 1
            params = p.params_for_arity(lay.arity)
            nxt = self.layers[l + 1] if l < L else None
            if lay.sharded:
                nl = lay.n // W
                if nxt is None:
                    f_next = None
                elif nxt.sharded:
                    f_next = nxt.f
                else:
                    f_next = nxt.f[rank * (nl // m):(rank + 1) * (nl // m)]        # this block's parents inside the whole next layer
                h = p.leaf_pair_hash(lay.f, f_next, m)                             # fri.rs:283, s = f_{l+1}[i/m]
                stop = sharded_stop_len(nl, lay.arity)
                lay.tree = p.merkle_build(params, lay.arity, l, h, nl, rank * nl, 0, stop)
                top, nlev = p.merkle_last_level(lay.tree)
                lay.nlev_local = nlev
                lay.local_lens = [p.merkle_level_len(lay.tree, v) for v in range(nlev)]
                allv = all_gather_rows(top)
                lay.top = p.merkle_build(params, lay.arity, l, allv, allv.shape[0], 0, nlev - 1, 1)
                rt, _ = p.merkle_last_level(lay.top)
                roots.append(rt[:1])
            else:
                n = lay.n
                if lay.hashed:
                    h = p.leaf_pair_hash(lay.f, nxt.f if nxt is not None else None, m)
                    lay.tree = p.merkle_build(params, lay.arity, l, h, n, 0, 0, 1)
                else:                                                               # commit_pairs(f_l, s_l), fri.rs:289
                    if nxt is not None:
                        idx = torch.arange(n, device=lay.f.device) // m
                        s_l = nxt.f[idx].contiguous()
                    else:
                        s_l = p.zeros(n)                                            # fri.rs:266
                    lay.tree = p.merkle_build_pairs(params, lay.arity, l, lay.f, s_l, n)
                lay.top, lay.nlev_local, lay.local_lens = None, 0, []
                rt, _ = p.merkle_last_level(lay.tree)
                roots.append(rt[:1])
        self.roots = torch.cat(roots, dim=0).cpu().numpy().view(np.uint64)           # ONE download for the L + 1 roots
        if os.environ.get("STARK_DIST_CHECK"):        # diagnostic: every rank must hold the same roots
            mine = torch.from_numpy(self.roots.view(np.int64).copy())
            allr = all_gather_rows(mine.view(1, -1)).view(W, -1)
            if not bool((allr == allr[0]).all()):
                bad = [l for l in range(L + 1) if not bool((allr.view(W, L + 1, 4)[:, l] == allr.view(W, L + 1, 4)[0, l]).all())]
                raise RuntimeError(f"rank {rank}: roots differ between ranks at layers {bad} (sharded flags {[bool(x.sharded) for x in self.layers]})")
        return self.roots

    # -- fri_prove_queries + encoding (fri.rs:355-466, 613-640) -------------------------------------------
    def queries(self):
        import numpy as np
        p, rank, W = self.p, self.rank, self.W
        plan = p.query_plan(self.roots, self.n0, self.schedule, self.r)
        kind, which, level, index = plan.requests()
        vals = np.zeros((len(kind), 4), np.uint64)
        # group the requests this rank owns by source, one gather per group
        groups = {}
        for i in range(len(kind)):
            lay = self.layers[int(which[i])]
            idx = int(index[i])
            if kind[i] == 0:
                if lay.sharded:
                    nl = lay.n // W
                    owner, key, loc = idx // nl, ("f", int(which[i])), idx % nl
                else:
                    owner, key, loc = 0, ("f", int(which[i])), idx
            else:
                v = int(level[i])
                if lay.sharded and v < lay.nlev_local - 1:
                    ln = lay.local_lens[v]
                    owner, key, loc = idx // ln, ("t", int(which[i]), v), idx % ln
                elif lay.sharded:
                    owner, key, loc = 0, ("top", int(which[i]), v - (lay.nlev_local - 1)), idx
                else:
                    owner, key, loc = 0, ("t", int(which[i]), v), idx
            if owner == rank:
                groups.setdefault(key, []).append((i, loc))
        for key, items in groups.items():
            pos = np.array([i for i, _ in items]); loc = np.array([j for _, j in items], dtype=np.int64)
            lay = self.layers[key[1]]
            if key[0] == "f":
                got = lay.f[torch.from_numpy(loc).to(lay.f.device)].cpu().numpy().view(np.uint64)
            elif key[0] == "t":
                got = p.merkle_gather(lay.tree, key[2], loc)
            else:
                got = p.merkle_gather(lay.top, key[2], loc)
            vals[pos] = got
        vals = _allreduce_sum(torch.from_numpy(vals.view(np.int64))).numpy().view(np.uint64)
        proof, est = plan.assemble(vals)
        plan.free()
        return proof, est

    def free(self):
        for lay in self.layers:
            if lay.tree is not None:
                self.p.merkle_free(lay.tree)
            if lay.top is not None:
                self.p.merkle_free(lay.top)
        self.layers = []

    def prove(self, a, s, e, t, f0_local=None):
        """(proof bytes, size estimate); a, s, e, t: this rank's [n0/W, 4] blocks."""
        import time
        t0 = time.perf_counter()
        if f0_local is None:
            f0_local = self.build_f0(a, s, e, t)
        self.p.sync(); t1 = time.perf_counter()
        self.commit(f0_local)
        self.p.sync(); t2 = time.perf_counter()
        out = self.queries()
        t3 = time.perf_counter()
        self.timings = {"build_f0_ms": (t1 - t0) * 1e3, "fri_build_ms": (t2 - t1) * 1e3, "queries_encode_ms": (t3 - t2) * 1e3}
        self.free()
        return out


class ShardedTrace:
    """The bench workload for N > 1 (BASELINE configs[3]/[4] shape): ONE trace of 2^log_n rows x 4 columns, natural-order
    blocks over the ranks; one step = sharded LDE of the four columns (six-step NTTs, all-to-all transposes), block-local
    DEEP-ALI merge with global indices, and the sharded commit phase of `DistProver` (folds, leaf hashes, lower Merkle levels
    block-local; tree tops all-gathered).  Returns the L+1 roots — the same values a single GPU computes for the whole trace."""

    def __init__(self, provider, log_n, log_blowup, schedule, seed_z, coset4, z4):
        self.p, self.log_n, self.lb = provider, log_n, log_blowup
        self.rank, self.W = world()
        self.lde = ShardedLde(provider, log_n, log_blowup, _unmont(coset4, getattr(provider, "field", 0)))
        self.z4 = z4
        self.N = 1 << (log_n + log_blowup)
        self.prover = DistProver(provider, self.N, schedule, 1, seed_z)

    def describe(self):
        return (f"one trace block-sharded over {self.W} ranks: LDE = {self.lde.n_all_to_all} all-to-all exchanges per column "
                f"(six-step NTT transposes, {type(_COMM).__name__}), merge / folds / leaf hashes / lower Merkle levels block-local, tree tops all-gathered")

    def step(self, cols):
        ext = [self.lde(c) for c in cols]
        nl = self.N // self.W
        f0 = self.p.ali_merge_shard(ext[0], ext[1], ext[2], ext[3], self.z4, self.rank * nl, self.N)
        del ext
        roots = self.prover.commit(f0)
        self.prover.free()
        return [roots[l] for l in range(roots.shape[0])]
