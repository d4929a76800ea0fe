"""Host-side mirror of the reference's operator interface for the hot path, over the C-ABI.

Same names, argument meaning and error behaviour as the Rust functions in SURVEY.md §8(b)
(`permute`, `hash_with_ds_dynamic`, `MerkleTree::new/new_pairs`, `fri_fold_layer`, `compute_s_layer`,
`fri_build_transcript`, `deep_fri_prove`, `fft`/`ifft` …); a failed precondition raises StarkError where
the reference panics.  Field vectors are numpy uint64 arrays of shape (n, 4): the 4 little-endian
Montgomery limbs of ark-ff — the same bytes a Rust `&[F]` holds.

Everything here calls libstark_mlwe_hip.so; nothing is computed in Python.
"""
import ctypes as C

import numpy as np

from ._abi import StarkError, load_library

PALLAS_FR = 0
BLS12_381_FR = 1


def _arr(x, cols=4):
    a = np.ascontiguousarray(x, dtype=np.uint64)
    if a.ndim == 1 and cols == 4 and a.size == 4:
        a = a.reshape(1, 4)
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Params:
    """PoseidonParams / PoseidonParamsDynamic handle (poseidon/src/lib.rs:16-21, 104-114)."""

    def __init__(self, ctx, handle, owned=True):
        self.ctx, self.h, self.owned = ctx, handle, owned
        t, rf, rp = C.c_int32(), C.c_int32(), C.c_int32()
        ctx._chk(ctx.lib.stark_poseidon_params_export(handle, C.byref(t), C.byref(rf), C.byref(rp), None, None, None))
        self.t, self.rounds_full, self.rounds_partial = t.value, rf.value, rp.value
        self.rate = self.t - 1

    def export(self):
        mds = np.zeros((self.t * self.t, 4), np.uint64)
        rcf = np.zeros((self.rounds_full * self.t, 4), np.uint64)
        rcp = np.zeros((self.rounds_partial, 4), np.uint64)
        self.ctx._chk(self.ctx.lib.stark_poseidon_params_export(self.h, None, None, None, _ptr(mds), _ptr(rcf), _ptr(rcp)))
        return mds, rcf, rcp

    def free(self):
        if self.owned and self.h:
            self.ctx.lib.stark_poseidon_params_free(self.h)
            self.h = None


class MerkleChannelCfg:
    """merkle/src/lib.rs:84-112."""

    def __init__(self, arity, params=None, tree_label=0):
        self.arity, self.params, self.tree_label = arity, params, tree_label

    def with_tree_label(self, label):
        return MerkleChannelCfg(self.arity, self.params, label)


class MerkleTree:
    """merkle/src/lib.rs:114-128: device-resident levels; `levels` materialises them lazily."""

    def __init__(self, ctx, handle, cfg, owned=True):
        self.ctx, self.h, self.cfg, self.owned = ctx, handle, cfg, owned

    @property
    def num_levels(self):
        return self.ctx.lib.stark_merkle_num_levels(self.h)

    def height(self):
        return self.num_levels - 1

    def level(self, lvl):
        n = self.ctx.lib.stark_merkle_level_len(self.h, lvl)
        out = np.zeros((n, 4), np.uint64)
        self.ctx._chk(self.ctx.lib.stark_merkle_level(self.h, lvl, _ptr(out)))
        return out

    @property
    def levels(self):
        return [self.level(i) for i in range(self.num_levels)]

    def root(self):
        out = np.zeros(4, np.uint64)
        self.ctx._chk(self.ctx.lib.stark_merkle_root(self.h, _ptr(out)))
        return out

    def gather(self, lvl, idx):
        ix = np.ascontiguousarray(idx, dtype=np.uint64)
        out = np.zeros((len(ix), 4), np.uint64)
        self.ctx._chk(self.ctx.lib.stark_merkle_gather(self.h, lvl, _ptr(ix), len(ix), _ptr(out)))
        return out

    def open_many(self, indices):
        """open_union_of_paths (merkle/src/lib.rs:246-315) -> canonical MerkleProof bytes."""
        ix = np.ascontiguousarray(indices, dtype=np.uint64)
        ln = C.c_size_t()
        self.ctx._chk(self.ctx.lib.stark_merkle_open(self.h, _ptr(ix), len(ix), None, 0, C.byref(ln)))
        buf = (C.c_uint8 * ln.value)()
        self.ctx._chk(self.ctx.lib.stark_merkle_open(self.h, _ptr(ix), len(ix), buf, ln.value, C.byref(ln)))
        return bytes(buf)

    open_many_single = open_many

    def free(self):
        if self.owned and self.h:
            self.ctx.lib.stark_merkle_free(self.h)
            self.h = None


class FriProverState:
    """fri.rs:210-216 (layers stay on the device)."""

    def __init__(self, ctx, handle, schedule):
        self.ctx, self.h, self.schedule = ctx, handle, list(schedule)

    @property
    def num_layers(self):
        return self.ctx.lib.stark_fri_num_layers(self.h)

    def f_layer(self, l):
        n = self.ctx.lib.stark_fri_layer_len(self.h, l)
        out = np.zeros((n, 4), np.uint64)
        self.ctx._chk(self.ctx.lib.stark_fri_layer_f(self.h, l, _ptr(out)))
        return out

    def root(self, l):
        out = np.zeros(4, np.uint64)
        self.ctx._chk(self.ctx.lib.stark_fri_layer_root(self.h, l, _ptr(out)))
        return out

    def z(self, l):
        out = np.zeros(4, np.uint64)
        self.ctx._chk(self.ctx.lib.stark_fri_layer_z(self.h, l, _ptr(out)))
        return out

    def tree(self, l):
        return MerkleTree(self.ctx, self.ctx.lib.stark_fri_layer_tree(self.h, l), None, owned=False)

    def free(self):
        if self.h:
            self.ctx.lib.stark_fri_state_free(self.h)
            self.h = None


class FriQueryPlan:
    """stark_fri_plan_t: the values the query phase needs, by (kind, which, level, index), and the assembler."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle

    def requests(self):
        n = self.ctx.lib.stark_fri_plan_num_requests(self.h)
        kind = np.zeros(n, np.uint32); which = np.zeros(n, np.uint32); level = np.zeros(n, np.uint32); index = np.zeros(n, np.uint64)
        if n:
            self.ctx._chk(self.ctx.lib.stark_fri_plan_requests(self.h, _ptr(kind), _ptr(which), _ptr(level), _ptr(index)))
        return kind, which, level, index

    def assemble(self, values):
        v = _arr(values).reshape(-1, 4)
        h = C.c_void_p()
        self.ctx._chk(self.ctx.lib.stark_fri_plan_assemble(self.h, _ptr(v), v.shape[0], C.byref(h)))
        return self.ctx._proof_out(h)

    def free(self):
        if self.h:
            self.ctx.lib.stark_fri_plan_free(self.h)
            self.h = None


class Transcript:
    """transcript/src/lib.rs:48-117 (device-resident state; absorbs run with the next challenge)."""

    def __init__(self, ctx, label: bytes):
        self.ctx = ctx; self.h = C.c_void_p()
        ctx._chk(ctx.lib.stark_transcript_new(ctx.h, label, len(label), C.byref(self.h)))

    def absorb_bytes(self, b: bytes):
        self.ctx._chk(self.ctx.lib.stark_transcript_absorb_bytes(self.h, b, len(b)))

    def absorb_fields(self, xs):
        a = _arr(xs).reshape(-1, 4)
        self.ctx._chk(self.ctx.lib.stark_transcript_absorb_fields(self.h, _ptr(a), a.shape[0]))

    absorb_field = absorb_fields

    def challenge(self, label: bytes):
        out = np.zeros(4, np.uint64)
        self.ctx._chk(self.ctx.lib.stark_transcript_challenge(self.h, label, len(label), _ptr(out)))
        return out

    def challenges(self, label: bytes, n):
        out = np.zeros((n, 4), np.uint64)
        self.ctx._chk(self.ctx.lib.stark_transcript_challenges(self.h, label, len(label), n, _ptr(out)))
        return out

    def free(self):
        if self.h:
            self.ctx.lib.stark_transcript_free(self.h); self.h = None


def ref_bench_inputs(seed, n, ncols=4):
    """The reference's bench inputs (end_to_end.rs:249-253): (ncols, n, 4) uint64 Montgomery limbs.  Host-only."""
    lib = load_library(); out = np.zeros((ncols * n, 4), np.uint64)
    rc = lib.stark_ref_bench_inputs(seed, n, ncols, _ptr(out))
    if rc != 0:
        raise StarkError(rc, "stark_ref_bench_inputs")
    return out.reshape(ncols, n, 4)


def root_of_unity(log_n, field=PALLAS_FR):
    """F::get_root_of_unity(2^log_n) (host-only)."""
    lib = load_library(); out = np.zeros(4, np.uint64)
    rc = lib.stark_root_of_unity(field, log_n, _ptr(out))
    if rc != 0:
        raise StarkError(rc, "no root of unity of that order (two-adicity 32)")
    return out


class DeepFriParams:
    """fri.rs:589."""

    def __init__(self, schedule, r, seed_z):
        self.schedule, self.r, self.seed_z = list(schedule), r, seed_z


STREAM_PRIVATE = C.c_void_p(-1)      # STARK_STREAM_PRIVATE: a private non-blocking stream (the caller synchronises around its own device work)


class Context:
    """One context = one GPU = one host thread (include/stark_mlwe.h conventions).

    `stream`: a hipStream_t handle (e.g. `torch.cuda.current_stream().cuda_stream`), None for the device's legacy default
    stream (ordered against torch's default stream: safe without manual synchronisation), or STREAM_PRIVATE."""

    def __init__(self, device=0, stream=None):
        self.lib = load_library()
        h = C.c_void_p()
        if isinstance(stream, int):
            stream = C.c_void_p(stream) if stream else None
        rc = self.lib.stark_ctx_create(device, stream, C.byref(h))
        if rc != 0:
            raise StarkError(rc, "stark_ctx_create failed: no usable HIP device (the product path has no CPU fallback)")
        self.h = h
        v = stream.value if isinstance(stream, C.c_void_p) else None
        self.private_stream = v is not None and v == STREAM_PRIVATE.value
        self.stream_handle = -1 if self.private_stream else (v or 0)       # 0: the legacy default stream
        self._tparams = None
        self._mparams = {}

    def _chk(self, rc):
        if rc != 0:
            raise StarkError(rc, (self.lib.stark_last_error(self.h) or b"").decode())

    def close(self):
        if self.h:
            for p in list(self._mparams.values()) + ([self._tparams] if self._tparams else []):
                p.free()
            self.lib.stark_ctx_destroy(self.h)
            self.h = None

    def sync(self):
        self._chk(self.lib.stark_ctx_sync(self.h))

    def set_option(self, key: str, value: int):
        """stark_ctx_set_option: "ntt_direct_max_log", "ntt_merged_coset", "ntt_log_tile", "ntt_min_waves", "poseidon_lane_only"."""
        self._chk(self.lib.stark_ctx_set_option(self.h, key.encode(), value))

    def trim(self):
        """Return the context's cached device blocks to the driver."""
        self._chk(self.lib.stark_ctx_trim(self.h))

    # ---- constants ------------------------------------------------------------------------------
    def poseidon_params_for_width(self, t):
        if t not in self._mparams:
            h = C.c_void_p()
            self._chk(self.lib.stark_poseidon_params_for_width(self.h, t, C.byref(h)))
            self._mparams[t] = Params(self, h)
        return self._mparams[t]

    def poseidon_params_for_arity(self, arity):
        t = 9 if arity <= 8 else 17 if arity <= 16 else 33 if arity <= 32 else 65 if arity <= 64 else 129
        if arity > 128:
            raise StarkError(-5, "unsupported Merkle arity; max supported = 128")
        return self.poseidon_params_for_width(t)

    def generate_params_t17_x5(self, seed: bytes):
        h = C.c_void_p()
        self._chk(self.lib.stark_poseidon_params_t17_seed(self.h, seed, len(seed), C.byref(h)))
        return Params(self, h)

    def transcript_params(self):
        if self._tparams is None:
            self._tparams = self.generate_params_t17_x5(b"POSEIDON-T17-X5-TRANSCRIPT")
        return self._tparams

    def params_upload(self, t, rf, rp, mds, rc_full, rc_partial):
        h = C.c_void_p()
        self._chk(self.lib.stark_poseidon_params_upload(self.h, t, rf, rp, _ptr(_arr(mds)), _ptr(_arr(rc_full)), _ptr(_arr(rc_partial)), C.byref(h)))
        return Params(self, h)

    # ---- poseidon -----------------------------------------------------------------------------------
    def permute(self, states, params):
        """permute / permute_dynamic over a batch: states (nstates, t, 4) -> same shape."""
        s = np.ascontiguousarray(states, dtype=np.uint64).copy()
        n = s.size // (params.t * 4)
        self._chk(self.lib.stark_poseidon_permute_batch(self.h, params.h, _ptr(s), n))
        return s

    permute_dynamic = permute

    def hash_with_ds_dynamic(self, ds_fields, inputs, params, n=1):
        ds, inp = _arr(ds_fields), _arr(inputs)
        nds, cnt = ds.shape[0] // n if ds.size else 0, inp.shape[0] // n if inp.size else 0
        out = np.zeros((n, 4), np.uint64)
        self._chk(self.lib.stark_poseidon_hash_with_ds_dynamic(self.h, params.h, _ptr(ds), nds, _ptr(inp), cnt, n, _ptr(out)))
        return out[0] if n == 1 else out

    def hash_with_ds(self, inputs, ds_tag, params):
        inp = _arr(inputs)
        out = np.zeros(4, np.uint64)
        self._chk(self.lib.stark_poseidon_hash_with_ds(self.h, params.h, _ptr(inp), inp.shape[0] if inp.size else 0, _ptr(_arr(ds_tag)), _ptr(out)))
        return out

    def hash_ds_level(self, params, arity, level, pos0, tree_label, children):
        ch = _arr(children)
        out = np.zeros(((ch.shape[0] + arity - 1) // arity, 4), np.uint64)
        self._chk(self.lib.stark_poseidon_hash_ds_batch(self.h, params.h, arity, level, pos0, tree_label, _ptr(ch), ch.shape[0], _ptr(out)))
        return out

    def leaf_pair_hash(self, f, f_next, m):
        """h[i] = hash_leaf_pair(f[i], f_next[i // m]) (f_next None => s = 0), fri.rs:283."""
        f = _arr(f)
        fn = None if f_next is None else _arr(f_next)
        out = np.zeros((f.shape[0], 4), np.uint64)
        self._chk(self.lib.stark_leaf_pair_hash(self.h, self.transcript_params().h, _ptr(f), _ptr(fn), f.shape[0], m, _ptr(out)))
        return out

    def hash_leaf_pair(self, f, s):
        return self.leaf_pair_hash(_arr(f), _arr(s), 1)[0]

    def tr_hash_fields_tagged(self, tag: bytes, fields, n=1):
        fl = _arr(fields)
        k = fl.shape[0] // n if fl.size else 0
        out = np.zeros((n, 4), np.uint64)
        self._chk(self.lib.stark_tr_hash_fields_tagged(self.h, None, tag, _ptr(fl), k, n, _ptr(out)))
        return out[0] if n == 1 else out

    # ---- merkle -------------------------------------------------------------------------------------
    def merkle_cfg(self, arity, tree_label=0):
        """MerkleChannelCfg::new(arity).with_tree_label(label)."""
        return MerkleChannelCfg(arity, self.poseidon_params_for_arity(arity), tree_label)

    def merkle_new(self, leaves, cfg):
        lv = _arr(leaves)
        h = C.c_void_p()
        self._chk(self.lib.stark_merkle_build(self.h, cfg.params.h, cfg.arity, cfg.tree_label, _ptr(lv), lv.shape[0] if lv.size else 0, 0, None, C.byref(h)))
        return MerkleTree(self, h, cfg)

    def merkle_new_pairs(self, f_vals, cp_vals, cfg):
        f, cp = _arr(f_vals), _arr(cp_vals)
        if f.shape != cp.shape:
            raise StarkError(-1, "f and cp length mismatch")
        h = C.c_void_p()
        self._chk(self.lib.stark_merkle_build(self.h, cfg.params.h, cfg.arity, cfg.tree_label, _ptr(f), f.shape[0] if f.size else 0, 1, _ptr(cp), C.byref(h)))
        return MerkleTree(self, h, cfg)

    # ---- fri ---------------------------------------------------------------------------------------
    def fri_sample_z_ell(self, seed_z, level, domain_size):
        out = np.zeros(4, np.uint64)
        self._chk(self.lib.stark_fri_sample_z(self.h, None, seed_z, level, domain_size, _ptr(out)))
        return out

    def fri_fold_layer(self, f_l, z_l, m):
        f = _arr(f_l)
        n = f.shape[0] if f.size else 0
        out = np.zeros((n // m if m else 0, 4), np.uint64)
        self._chk(self.lib.stark_fri_fold(self.h, _ptr(f), n, _ptr(_arr(z_l)), m, _ptr(out)))
        return out

    def compute_s_layer(self, f_l, z_l, m):
        """fri.rs:123-143: s[i] = fold(f)[i // m] — a replicated view of the folded layer."""
        return np.repeat(self.fri_fold_layer(f_l, z_l, m), m, axis=0)

    def fri_build_transcript(self, f0, schedule, seed_z):
        f = _arr(f0)
        sch = np.ascontiguousarray(schedule, dtype=np.uint64)
        h = C.c_void_p()
        self._chk(self.lib.stark_fri_build(self.h, _ptr(f), f.shape[0], _ptr(sch), len(sch), seed_z, C.byref(h)))
        return FriProverState(self, h, schedule)

    # ---- deep-ali / prove -----------------------------------------------------------------------------
    def deep_ali_merge_evals(self, a, s, e, t, omega, z, r_eval=None, beta=None, want_c_star=True):
        a, s, e, t = _arr(a), _arr(s), _arr(e), _arr(t)
        n = a.shape[0]
        f0 = np.zeros((n, 4), np.uint64)
        cs = np.zeros(4, np.uint64) if want_c_star else None
        r = None if r_eval is None else _arr(r_eval)
        b = None if beta is None else _arr(beta)
        self._chk(self.lib.stark_ali_merge(self.h, _ptr(a), _ptr(s), _ptr(e), _ptr(t), _ptr(r), _ptr(b), _ptr(_arr(omega)), _ptr(_arr(z)), n, _ptr(f0), _ptr(cs)))
        return f0, _arr(z).reshape(4), cs

    def build_f0(self, a, s, e, t, n0):
        """DeepAliRealBuilder::default().build_f0 (fri.rs:535-569); returns (f0, aux[7])."""
        a, s, e, t = _arr(a), _arr(s), _arr(e), _arr(t)
        f0 = np.zeros((n0, 4), np.uint64)
        aux = np.zeros((7, 4), np.uint64)
        self._chk(self.lib.stark_build_f0(self.h, _ptr(a), _ptr(s), _ptr(e), _ptr(t), n0, _ptr(f0), _ptr(aux)))
        return f0, aux

    def deep_fri_prove(self, a, s, e, t, n0, params: DeepFriParams, f0=None):
        """deep_fri_prove (fri.rs:601-641) -> (canonical proof bytes, size estimate, stage ms)."""
        sch = np.ascontiguousarray(params.schedule, dtype=np.uint64)
        h = C.c_void_p()
        if f0 is None:
            a, s, e, t = _arr(a), _arr(s), _arr(e), _arr(t)
            self._chk(self.lib.stark_deep_fri_prove(self.h, _ptr(a), _ptr(s), _ptr(e), _ptr(t), None, n0, _ptr(sch), len(sch), params.r, params.seed_z, C.byref(h)))
        else:
            f = _arr(f0)
            self._chk(self.lib.stark_deep_fri_prove(self.h, None, None, None, None, _ptr(f), n0, _ptr(sch), len(sch), params.r, params.seed_z, C.byref(h)))
        try:
            ln = self.lib.stark_proof_len(h)
            buf = (C.c_uint8 * ln)()
            self._chk(self.lib.stark_proof_bytes(h, buf))
            est = self.lib.stark_proof_size_estimate(h)
            ms = [self.lib.stark_proof_stage_ms(h, i) for i in range(3)]
        finally:
            self.lib.stark_proof_free(h)
        return bytes(buf), est, ms

    def deep_fri_prove_batch_dev(self, traces, n0, params: DeepFriParams):
        """`traces`: list of (a, s, e, t) DEVICE pointers (ints) of independent n0-row traces -> list of (proof bytes, size estimate, stage ms),
        each equal to deep_fri_prove of that trace alone; the 4 * len(traces) serial column sponges run concurrently (stark_deep_fri_prove_batch_dev)."""
        B = len(traces)
        sch = np.ascontiguousarray(params.schedule, dtype=np.uint64)
        cols = [(C.c_void_p * B)(*[int(tr[c]) for tr in traces]) for c in range(4)]
        out = (C.c_void_p * B)()
        self._chk(self.lib.stark_deep_fri_prove_batch_dev(self.h, B, cols[0], cols[1], cols[2], cols[3], n0, _ptr(sch), len(sch), params.r, params.seed_z, out))
        res = []
        for p in range(B):
            h = C.c_void_p(out[p])
            try:
                ln = self.lib.stark_proof_len(h); buf = (C.c_uint8 * ln)()
                self._chk(self.lib.stark_proof_bytes(h, buf))
                res.append((bytes(buf), self.lib.stark_proof_size_estimate(h), [self.lib.stark_proof_stage_ms(h, i) for i in range(3)]))
            finally:
                self.lib.stark_proof_free(h)
        return res

    def deep_fri_verify(self, params: DeepFriParams, proof: bytes) -> bool:
        """deep_fri_verify (fri.rs:643-762) on canonical proof bytes."""
        sch = np.ascontiguousarray(params.schedule, dtype=np.uint64)
        buf = (C.c_uint8 * max(1, len(proof))).from_buffer_copy(proof or b"\0")
        ok = C.c_int32(0)
        self._chk(self.lib.stark_deep_fri_verify(self.h, buf, len(proof), _ptr(sch), len(sch), params.r, params.seed_z, C.byref(ok)))
        return bool(ok.value)

    def merkle_verify_single(self, cfg, root, indices, leaves, proof: bytes) -> bool:
        """MerkleProver::verify_single (merkle/src/lib.rs:800-812)."""
        ix = np.ascontiguousarray(indices, dtype=np.uint64); lv = _arr(leaves)
        buf = (C.c_uint8 * max(1, len(proof))).from_buffer_copy(proof or b"\0"); ok = C.c_int32(0)
        self._chk(self.lib.stark_merkle_verify_many_ds(self.h, cfg.arity, cfg.tree_label, _ptr(_arr(root)), _ptr(ix), len(ix), _ptr(lv), buf, len(proof), C.byref(ok)))
        return bool(ok.value)

    def merkle_verify_pairs(self, cfg, root, indices, f_vals, cp_vals, proof: bytes) -> bool:
        """MerkleProver::verify_pairs (merkle/src/lib.rs:841-855)."""
        ix = np.ascontiguousarray(indices, dtype=np.uint64); f, cp = _arr(f_vals), _arr(cp_vals)
        buf = (C.c_uint8 * max(1, len(proof))).from_buffer_copy(proof or b"\0"); ok = C.c_int32(0)
        self._chk(self.lib.stark_merkle_verify_pairs_ds(self.h, cfg.arity, cfg.tree_label, _ptr(_arr(root)), _ptr(ix), len(ix), _ptr(f), _ptr(cp), buf, len(proof), C.byref(ok)))
        return bool(ok.value)

    # ---- sum-check consumer (channel/src/lib.rs:1045-1240) ------------------------------------------------
    def commitment_commit(self, ds_tag, leaves):
        """MerkleCommitment::new(MerkleConfig::with_default_params(ds_tag)).commit(leaves) -> (root, tree) (commitment/src/lib.rs:85-90)."""
        lv = _arr(leaves); h = C.c_void_p()
        self._chk(self.lib.stark_commitment_commit(self.h, ds_tag, _ptr(lv), lv.shape[0] if lv.size else 0, C.byref(h)))
        t = MerkleTree(self, h, MerkleChannelCfg(16, None, ds_tag))
        return t.root(), t

    def commitment_verify(self, ds_tag, root, indices, values, proof: bytes) -> bool:
        """CommitmentScheme::verify (commitment/src/lib.rs:96-113)."""
        ix = np.ascontiguousarray(indices, dtype=np.uint64); v = _arr(values)
        buf = (C.c_uint8 * max(1, len(proof))).from_buffer_copy(proof or b"\0"); ok = C.c_int32(0)
        self._chk(self.lib.stark_commitment_verify(self.h, ds_tag, _ptr(_arr(root)), _ptr(ix), len(ix), _ptr(v), buf, len(proof), C.byref(ok)))
        return bool(ok.value)

    def mle_evaluate(self, table, r):
        """Mle::new(table).evaluate(r) (channel/src/lib.rs:279-295)."""
        t, rr = _arr(table), _arr(r).reshape(-1, 4); out = np.zeros(4, np.uint64)
        if t.shape[0] != 1 << rr.shape[0]:
            raise StarkError(-1, "dimension mismatch")
        self._chk(self.lib.stark_mle_evaluate(self.h, _ptr(t), rr.shape[0], _ptr(rr), _ptr(out)))
        return out

    def prove_plain(self, k, tree_label, witness):
        """prove_plain(&build_vk_plain(k, F::from(tree_label)), witness) -> bincode-layout ProofPlain bytes."""
        w = _arr(witness); h = C.c_void_p()
        if w.shape[0] != 1 << k:
            raise StarkError(-1, "MLE length must be 2^k")
        self._chk(self.lib.stark_sumcheck_prove_plain(self.h, _ptr(w), k, tree_label, C.byref(h)))
        return self._proof_out(h)[0]

    def verify_plain(self, k, tree_label, proof: bytes) -> bool:
        buf = (C.c_uint8 * max(1, len(proof))).from_buffer_copy(proof or b"\0"); ok = C.c_int32(0)
        self._chk(self.lib.stark_sumcheck_verify_plain(self.h, k, tree_label, buf, len(proof), C.byref(ok)))
        return bool(ok.value)

    def prove_mf(self, k, tree_label, queries_per_round, witness):
        """prove_mf(&build_vk_mf(k, F::from(tree_label), q), witness) -> bincode-layout ProofMF bytes."""
        w = _arr(witness); h = C.c_void_p()
        if w.shape[0] != 1 << k:
            raise StarkError(-1, "MLE length must be 2^k")
        self._chk(self.lib.stark_sumcheck_prove_mf(self.h, _ptr(w), k, tree_label, queries_per_round, C.byref(h)))
        return self._proof_out(h)[0]

    def verify_mf(self, k, tree_label, queries_per_round, proof: bytes) -> bool:
        buf = (C.c_uint8 * max(1, len(proof))).from_buffer_copy(proof or b"\0"); ok = C.c_int32(0)
        self._chk(self.lib.stark_sumcheck_verify_mf(self.h, k, tree_label, queries_per_round, buf, len(proof), C.byref(ok)))
        return bool(ok.value)

    def _proof_out(self, h):
        try:
            ln = self.lib.stark_proof_len(h)
            buf = (C.c_uint8 * ln)()
            self._chk(self.lib.stark_proof_bytes(h, buf))
            est = self.lib.stark_proof_size_estimate(h)
        finally:
            self.lib.stark_proof_free(h)
        return bytes(buf), est

    # ---- one trace sharded over several GPUs: the pieces stark_mlwe_amd.dist composes ----------------------
    def ali_challenges(self, digests, n0):
        """(seed, z, beta) of DeepAliRealBuilder::build_f0 from the four column digests (fri.rs:551-560)."""
        d = _arr(digests)
        aux = np.zeros((3, 4), np.uint64)
        self._chk(self.lib.stark_ali_challenges(self.h, _ptr(d), n0, _ptr(aux)))
        return aux

    def fri_query_plan(self, roots, n0, schedule, r):
        """Query plan of deep_fri_prove over roots only (fri.rs:355-466): FriQueryPlan with .requests()."""
        rt = _arr(roots)
        sch = np.ascontiguousarray(schedule, dtype=np.uint64)
        h = C.c_void_p()
        self._chk(self.lib.stark_fri_plan_create(self.h, _ptr(rt), n0, _ptr(sch), len(sch), r, C.byref(h)))
        return FriQueryPlan(self, h)

    # ---- field helpers (crates/field/src/lib.rs) ----------------------------------------------------------
    def compute_powers(self, base, n, field=PALLAS_FR):
        """compute_powers(base, n) (field/src/lib.rs:125-133)."""
        out = np.zeros((n, 4), np.uint64)
        self._chk(self.lib.stark_compute_powers(self.h, field, _ptr(_arr(base)), n, _ptr(out)))
        return out

    def domain(self, log_n, field=PALLAS_FR, precompute=False):
        """Domain::new(log_n) (field/src/lib.rs:43-53): (size, log_n, omega[, elements])."""
        w = root_of_unity(log_n, field)
        return (1 << log_n, log_n, w, self.compute_powers(w, 1 << log_n, field) if precompute else None)

    # ---- fft (crates/fft/src/lib.rs:6-32) ------------------------------------------------------------
    def fft(self, coeffs, field=BLS12_381_FR, coset=None):
        v = _arr(coeffs).copy()
        log_n = int(v.shape[0]).bit_length() - 1
        if (1 << log_n) != v.shape[0]:
            raise StarkError(-1, "radix-2 domain size must be a power of two")
        self._chk(self.lib.stark_ntt(self.h, field, _ptr(v), log_n, 0, _ptr(None if coset is None else _arr(coset))))
        return v

    def ifft(self, evals, field=BLS12_381_FR, coset=None):
        v = _arr(evals).copy()
        log_n = int(v.shape[0]).bit_length() - 1
        if (1 << log_n) != v.shape[0]:
            raise StarkError(-1, "radix-2 domain size must be a power of two")
        self._chk(self.lib.stark_ntt(self.h, field, _ptr(v), log_n, 1, _ptr(None if coset is None else _arr(coset))))
        return v

    fft_in_place, ifft_in_place = fft, ifft

    def lde(self, evals, log_blowup, field=PALLAS_FR, coset=None):
        v = _arr(evals)
        log_n = int(v.shape[0]).bit_length() - 1
        out = np.zeros((v.shape[0] << log_blowup, 4), np.uint64)
        self._chk(self.lib.stark_lde(self.h, field, _ptr(v), log_n, log_blowup, _ptr(None if coset is None else _arr(coset)), _ptr(out)))
        return out
