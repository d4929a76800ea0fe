"""Multi-GPU orchestration of the path: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests) for the exchanges, the C-ABI for all local compute.

Only two steps of the path exchange data:

* the six-step NTT — ONE all-to-all (the row/column transpose between its two local phases);
* the top of a sharded Merkle tree — an all-gather of a few digests.

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

import torch
import torch.distributed as dist


def world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)


def exchange_all_to_all(send: torch.Tensor) -> torch.Tensor:
    """send: [W, chunk...] — chunk q goes to rank q.  Returns recv with chunk p = what rank p sent here."""
    rank, W = world()
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


def all_gather_rows(x: torch.Tensor) -> torch.Tensor:
    rank, W = world()
    if W == 1:
        return x.clone()
    if dist.get_backend() != "nccl" and x.is_cuda:     # gloo: stage through host memory
        hx = x.cpu(); out = [torch.empty_like(hx) for _ in range(W)]
        dist.all_gather(out, hx.contiguous())
        return torch.cat(out, dim=0).to(x.device)
    out = [torch.empty_like(x) for _ in range(W)]
    dist.all_gather(out, x.contiguous())
    return torch.cat(out, dim=0)


class HipProvider:
    """Local compute through libstark_mlwe_hip.so on torch CUDA tensors of shape [rows, 4] (int64 view of the limbs)."""

    def __init__(self, ctx, field=0, device="cuda"):
        self.ctx, self.lib, self.field, self.device = ctx, ctx.lib, field, device

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr())

    def ntt_columns(self, slab, log_rows, ncols, col0, log_n, inverse):
        self.ctx._chk(self.lib.stark_ntt_columns_dev(self.ctx.h, self.field, self._p(slab), log_rows, ncols, col0, log_n, int(inverse)))

    def ntt_rows(self, slab, nrows, log_cols, inverse, scale4=None):
        from .api import _ptr
        self.ctx._chk(self.lib.stark_ntt_rows_dev(self.ctx.h, self.field, self._p(slab), nrows, log_cols, int(inverse), _ptr(scale4)))

    def merkle_build(self, params, arity, tree_label, leaves, n, first_pos, level0, stop_at_len):
        h = C.c_void_p()
        self.ctx._chk(self.lib.stark_merkle_build_dev(self.ctx.h, params.h, arity, tree_label, self._p(leaves), n, 0, None, first_pos, level0, stop_at_len, C.byref(h)))
        return h

    def merkle_last_level(self, h):
        """(device tensor [k, 4] copy of the last level, number of levels)."""
        nl = self.lib.stark_merkle_num_levels(h)
        k = self.lib.stark_merkle_level_len(h, nl - 1)
        tmp = torch.empty((k, 4), dtype=torch.int64)
        self.ctx._chk(self.lib.stark_merkle_level(h, nl - 1, C.c_void_p(tmp.data_ptr())))   # a few digests: host hop is fine
        return tmp.to(self.device), nl

    def merkle_free(self, h):
        self.lib.stark_merkle_free(h)

    def sync(self):
        self.ctx.sync()


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

    def forward(self, slab: torch.Tensor, scale4=None) -> torch.Tensor:
        """slab: [R * C/W, 4] (row-major [R][C/W]).  Returns [R/W * C, 4] (row-major [R/W][C])."""
        R, ncl, nrl, W = self.R, self.ncl, self.nrl, self.W
        # phase A: column NTTs of size R on the local column block + twiddle w_N^(col_global * k1)
        self.p.ntt_columns(slab, self.log_rows, ncl, self.rank * ncl, self.log_n, self.inverse)
        # the one exchange: rows k1 of block q go to rank q (contiguous [R/W][C/W] chunks of the slab)
        send = slab.view(W, nrl * ncl, 4)
        self.p.sync()
        recv = exchange_all_to_all(send)                                          # [W(src p), R/W, C/W]
        rows = recv.view(W, nrl, ncl, 4).permute(1, 0, 2, 3).contiguous().view(nrl * self.C, 4)
        # phase B: R/W contiguous NTTs of size C
        self.p.ntt_rows(rows, nrl, self.log_n - self.log_rows, self.inverse, scale4)
        return rows

    def to_natural_blocks(self, rows: torch.Tensor) -> torch.Tensor:
        """Second all-to-all: from the transposed output to natural order, block-sharded (n/W contiguous)."""
        W, nrl, Cc = self.W, self.nrl, self.C
        # rank q holds X[k1 + R*k'] for its k1 block; natural block b holds k in [b*n/W, (b+1)*n/W) <=> k' in [b*C/W, (b+1)*C/W)
        send = rows.view(nrl, W, Cc // W, 4).permute(1, 2, 0, 3).contiguous()   # [dst b][k' local][k1 local]
        recv = exchange_all_to_all(send.view(W, -1, 4))                          # [src q][k' local][k1 local]
        return recv.view(W, Cc // W, nrl, 4).permute(1, 0, 2, 3).contiguous().view(-1, 4)   # [k' local][k1 global]


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
