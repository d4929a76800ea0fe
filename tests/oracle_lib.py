"""ctypes wrapper of oracle/_build/liboracle.so — TEST INFRASTRUCTURE (the checker, never the product)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
vp = C.c_void_p


def P(a):
    return None if a is None else a.ctypes.data_as(vp)


def A(x):
    return np.ascontiguousarray(x, dtype=np.uint64)


class Oracle:
    def __init__(self):
        self.l = C.CDLL(PATH)
        l = self.l
        l.oracle_merkle_level_len.restype = C.c_size_t
        l.oracle_proof_len.restype = C.c_size_t
        l.oracle_proof_size_estimate.restype = C.c_size_t
        l.oracle_proof_layer_len.restype = C.c_size_t
        l.oracle_pick_arity_for_layer.restype = C.c_size_t
        l.oracle_proof_size_estimate_from_bytes.restype = C.c_long
        l.oracle_proof_secs.restype = C.c_double
        for name in dir(l):
            pass

    # ---- field -----------------------------------------------------------------------------------
    def fr_op(self, field, op, a, b=None):
        a = A(a); out = np.zeros(4, np.uint64)
        b2 = None if b is None else A(b)
        assert self.l.oracle_fr_op(field, op, P(a), P(b2), P(out)) == 0
        return out

    def add(self, a, b, field=0): return self.fr_op(field, 0, a, b)
    def sub(self, a, b, field=0): return self.fr_op(field, 1, a, b)
    def mul(self, a, b, field=0): return self.fr_op(field, 2, a, b)
    def inv(self, a, field=0): return self.fr_op(field, 3, a)
    def from_u64(self, x, field=0): return self.fr_op(field, 4, np.array([x, 0, 0, 0], np.uint64))
    def to_canonical(self, a, field=0): return self.fr_op(field, 5, a)
    def from_canonical(self, a, field=0): return self.fr_op(field, 6, a)
    def root_of_unity(self, log_n, field=0): return self.fr_op(field, 7, np.array([log_n, 0, 0, 0], np.uint64))
    def pow(self, a, e, field=0): return self.fr_op(field, 8, a, np.array([e, 0, 0, 0], np.uint64))

    def from_int(self, x, field=0):
        return self.from_canonical(np.array([(x >> (64 * i)) & (2**64 - 1) for i in range(4)], np.uint64), field)

    def to_int(self, a, field=0):
        c = self.to_canonical(a, field)
        return sum(int(c[i]) << (64 * i) for i in range(4))

    def from_le_bytes_mod_order(self, b: bytes):
        out = np.zeros(4, np.uint64); buf = (C.c_uint8 * len(b)).from_buffer_copy(b)
        self.l.oracle_fr_from_le_bytes_mod_order(buf, C.c_size_t(len(b)), P(out)); return out

    def to_bytes_le(self, a):
        out = (C.c_uint8 * 32)(); self.l.oracle_fr_to_bytes_le(P(A(a)), out); return bytes(out)

    # ---- primitives --------------------------------------------------------------------------------
    def blake3(self, data: bytes):
        out = (C.c_uint8 * 32)(); buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data or b"\0")
        self.l.oracle_blake3(buf, C.c_size_t(len(data)), out); return bytes(out)

    def stdrng_from_seed_u64s(self, seed: bytes, n):
        out = np.zeros(n, np.uint64); buf = (C.c_uint8 * 32).from_buffer_copy(seed)
        self.l.oracle_stdrng_from_seed_u64s(buf, C.c_size_t(n), P(out)); return out

    def stdrng_seed_from_u64_u64s(self, seed, n):
        out = np.zeros(n, np.uint64); self.l.oracle_stdrng_seed_from_u64_u64s(C.c_uint64(seed), C.c_size_t(n), P(out)); return out

    def rand_fr_columns(self, seed, n, ncols=1):
        out = np.zeros((ncols * n, 4), np.uint64)
        self.l.oracle_rand_fr_columns(C.c_uint64(seed), C.c_size_t(n), C.c_size_t(ncols), P(out))
        return out.reshape(ncols, n, 4)

    def synth_column(self, seed, col, i0, n):
        out = np.zeros((n, 4), np.uint64)
        self.l.oracle_synth_column(C.c_uint64(seed), C.c_uint64(col), C.c_size_t(i0), C.c_size_t(n), P(out)); return out

    # ---- poseidon ------------------------------------------------------------------------------------
    def poseidon_params(self, kind, t):
        rf, rp = C.c_int(), C.c_int()
        assert self.l.oracle_poseidon_params(kind, t, None, None, None, C.byref(rf), C.byref(rp)) == 0
        mds = np.zeros((t * t, 4), np.uint64); rcf = np.zeros((rf.value * t, 4), np.uint64); rcp = np.zeros((rp.value, 4), np.uint64)
        assert self.l.oracle_poseidon_params(kind, t, P(mds), P(rcf), P(rcp), None, None) == 0
        return rf.value, rp.value, mds, rcf, rcp

    def permute(self, kind, t, states):
        s = A(states).copy(); n = s.size // (4 * t)
        assert self.l.oracle_permute(kind, t, P(s), C.c_size_t(n)) == 0; return s

    def hash_with_ds_dynamic(self, kind, t, ds4, inputs, cnt, n=1):
        ds4, inputs = A(ds4), A(inputs); out = np.zeros((n, 4), np.uint64)
        assert self.l.oracle_hash_with_ds_dynamic(kind, t, P(ds4), P(inputs), C.c_size_t(cnt), C.c_size_t(n), P(out)) == 0
        return out[0] if n == 1 else out

    def hash_with_ds(self, kind, inputs, ds_tag):
        inputs = A(inputs); out = np.zeros(4, np.uint64)
        cnt = inputs.shape[0] if inputs.size else 0
        assert self.l.oracle_hash_with_ds(kind, P(inputs), C.c_size_t(cnt), P(A(ds_tag)), P(out)) == 0; return out

    def tr_hash_fields_tagged(self, tag: bytes, fields):
        f = A(fields); n = f.shape[0] if f.size else 0; out = np.zeros(4, np.uint64)
        assert self.l.oracle_tr_hash_fields_tagged(tag, P(f), C.c_size_t(n), P(out)) == 0; return out

    def leaf_pair_hash(self, f, f_next, m):
        f = A(f); n = f.shape[0]; fn = None if f_next is None else A(f_next); out = np.zeros((n, 4), np.uint64)
        assert self.l.oracle_leaf_pair_hash(P(f), P(fn), C.c_size_t(n), C.c_size_t(m), P(out)) == 0; return out

    def transcript_vec(self, label: bytes, msg: bytes, ch: bytes):
        out = np.zeros(4, np.uint64); assert self.l.oracle_transcript_vec(label, msg, ch, P(out)) == 0; return out

    # ---- merkle ---------------------------------------------------------------------------------------
    def merkle_build(self, arity, label, leaves, cp=None, params_kind=0):
        lv = A(leaves); n = lv.shape[0] if lv.size else 0; h = vp()
        cpa = None if cp is None else A(cp)
        rc = self.l.oracle_merkle_build(params_kind, C.c_size_t(arity), C.c_uint64(label), P(lv), C.c_size_t(n), 0 if cp is None else 1, P(cpa), C.byref(h))
        if rc != 0:
            raise RuntimeError("oracle merkle build failed")
        return OTree(self, h)

    # ---- fri / ali ---------------------------------------------------------------------------------------
    def fri_sample_z_ell(self, seed_z, level, size):
        out = np.zeros(4, np.uint64); assert self.l.oracle_fri_sample_z_ell(C.c_uint64(seed_z), C.c_size_t(level), C.c_size_t(size), P(out)) == 0; return out

    def fri_fold_layer(self, f, z, m):
        f = A(f); n = f.shape[0] if f.size else 0; out = np.zeros((n // m if m else 0, 4), np.uint64)
        rc = self.l.oracle_fri_fold_layer(P(f), C.c_size_t(n), P(A(z)), C.c_size_t(m), P(out))
        if rc != 0:
            raise RuntimeError("oracle fold failed")
        return out

    def compute_s_layer(self, f, z, m):
        f = A(f); n = f.shape[0]; out = np.zeros((n, 4), np.uint64)
        assert self.l.oracle_compute_s_layer(P(f), C.c_size_t(n), P(A(z)), C.c_size_t(m), P(out)) == 0; return out

    def pick_arity_for_layer(self, n, m): return self.l.oracle_pick_arity_for_layer(C.c_size_t(n), C.c_size_t(m))

    def domain_omega(self, n):
        out = np.zeros(4, np.uint64); self.l.oracle_domain_omega(C.c_size_t(n), P(out)); return out

    def ali_merge(self, a, s, e, t, omega, z, r=None, beta=None, want_c_star=True):
        a, s, e, t = A(a), A(s), A(e), A(t); n = a.shape[0]; f0 = np.zeros((n, 4), np.uint64); cs = np.zeros(4, np.uint64)
        rc = self.l.oracle_ali_merge(P(a), P(s), P(e), P(t), P(None if r is None else A(r)), P(None if beta is None else A(beta)), P(A(omega)), P(A(z)), C.c_size_t(n), P(f0), P(cs) if want_c_star else None)
        if rc != 0:
            raise RuntimeError("oracle ali merge failed")
        return f0, cs

    def build_f0(self, a, s, e, t, n0, mock=False):
        a, s, e, t = A(a), A(s), A(e), A(t); f0 = np.zeros((n0, 4), np.uint64); aux = np.zeros((7, 4), np.uint64)
        assert self.l.oracle_build_f0(P(a), P(s), P(e), P(t), C.c_size_t(n0), 1 if mock else 0, P(f0), P(aux)) == 0
        return f0, aux

    def deep_fri_prove(self, a, s, e, t, n0, schedule, r, seed_z, f0=None):
        sch = np.ascontiguousarray(schedule, dtype=np.uint64); h = vp()
        args = [P(A(x)) if x is not None else None for x in (a, s, e, t, f0)]
        rc = self.l.oracle_deep_fri_prove(*args, C.c_size_t(n0), P(sch), C.c_size_t(len(sch)), C.c_size_t(r), C.c_uint64(seed_z), C.byref(h))
        if rc != 0:
            raise RuntimeError("oracle prove failed")
        return OProof(self, h)

    def deep_fri_verify(self, proof_bytes: bytes, schedule, r, seed_z):
        sch = np.ascontiguousarray(schedule, dtype=np.uint64); buf = (C.c_uint8 * len(proof_bytes)).from_buffer_copy(proof_bytes)
        return self.l.oracle_deep_fri_verify(buf, C.c_size_t(len(proof_bytes)), P(sch), C.c_size_t(len(sch)), C.c_size_t(r), C.c_uint64(seed_z))

    def proof_size_estimate_from_bytes(self, proof_bytes: bytes):
        buf = (C.c_uint8 * len(proof_bytes)).from_buffer_copy(proof_bytes)
        return self.l.oracle_proof_size_estimate_from_bytes(buf, C.c_size_t(len(proof_bytes)))

    # ---- sum-check (N4) -------------------------------------------------------------------------------
    def sumcheck_prove(self, variant, k, tree_label, witness, q=2):
        """variant 0: prove_plain, 1: prove_mf (channel/src/lib.rs:1045, :1130) -> bincode-layout proof bytes."""
        h = vp(); self.l.oracle_bytes_len.restype = C.c_size_t
        rc = self.l.oracle_sumcheck_prove(variant, C.c_size_t(k), C.c_uint64(tree_label), C.c_size_t(q), P(A(witness)), C.byref(h))
        if rc != 0:
            raise RuntimeError("oracle sumcheck prove failed")
        n = self.l.oracle_bytes_len(h); buf = (C.c_uint8 * n)(); self.l.oracle_bytes_copy(h, buf); self.l.oracle_bytes_free(h)
        return bytes(buf)

    def sumcheck_verify(self, variant, k, tree_label, proof: bytes, q=2):
        buf = (C.c_uint8 * max(1, len(proof))).from_buffer_copy(proof or b"\0")
        return self.l.oracle_sumcheck_verify(variant, C.c_size_t(k), C.c_uint64(tree_label), C.c_size_t(q), buf, C.c_size_t(len(proof)))

    def commitment_root(self, tree_label, leaves):
        lv = A(leaves); out = np.zeros(4, np.uint64)
        assert self.l.oracle_commitment_root(C.c_uint64(tree_label), P(lv), C.c_size_t(lv.shape[0]), P(out)) == 0; return out

    def mle_evaluate(self, table, r):
        t, rr = A(table), A(r); k = rr.shape[0]; out = np.zeros(4, np.uint64)
        assert self.l.oracle_mle_evaluate(P(t), C.c_size_t(k), P(rr), P(out)) == 0; return out

    # ---- ntt -----------------------------------------------------------------------------------------
    def ntt(self, field, data, inverse=False, coset=None):
        d = A(data).copy(); log_n = int(d.shape[0]).bit_length() - 1
        self.l.oracle_ntt(field, P(d), C.c_uint(log_n), 1 if inverse else 0, P(None if coset is None else A(coset))); return d

    def dft_naive(self, field, data, inverse=False):
        d = A(data); log_n = int(d.shape[0]).bit_length() - 1; out = np.zeros_like(d)
        self.l.oracle_dft_naive(field, P(d), C.c_uint(log_n), 1 if inverse else 0, P(out)); return out

    def poly_eval_many(self, field, coeffs, points):
        """Horner: sum_j coeffs[j] * x^j at every x in points (the definition; O(n) per point, all host cores)."""
        c, p = A(coeffs), A(points).reshape(-1, 4); out = np.zeros_like(p)
        assert self.l.oracle_poly_eval_many(field, P(c), C.c_size_t(c.shape[0]), P(p), C.c_size_t(p.shape[0]), P(out)) == 0
        return out

    def lde(self, field, evals, log_blowup, coset=None):
        d = A(evals); log_n = int(d.shape[0]).bit_length() - 1; out = np.zeros((d.shape[0] << log_blowup, 4), np.uint64)
        self.l.oracle_lde(field, P(d), C.c_uint(log_n), C.c_uint(log_blowup), P(None if coset is None else A(coset)), P(out)); return out


class OTree:
    def __init__(self, o, h): self.o, self.h = o, h
    def num_levels(self): return self.o.l.oracle_merkle_num_levels(self.h)
    def level(self, lvl):
        n = self.o.l.oracle_merkle_level_len(self.h, lvl); out = np.zeros((n, 4), np.uint64); self.o.l.oracle_merkle_level(self.h, lvl, P(out)); return out
    def root(self):
        out = np.zeros(4, np.uint64); self.o.l.oracle_merkle_root(self.h, P(out)); return out
    def open_verify(self, idx, values, cp_values=None):
        ix = np.ascontiguousarray(idx, dtype=np.uint64); ns = C.c_size_t()
        rc = self.o.l.oracle_merkle_open_verify(self.h, P(ix), C.c_size_t(len(ix)), P(A(values)), P(None if cp_values is None else A(cp_values)), C.byref(ns))
        return rc, ns.value
    def open_bytes(self, idx):
        ix = np.ascontiguousarray(idx, dtype=np.uint64); self.o.l.oracle_merkle_open_bytes.restype = C.c_size_t
        n = self.o.l.oracle_merkle_open_bytes(self.h, P(ix), C.c_size_t(len(ix)), None, C.c_size_t(0)); buf = (C.c_uint8 * n)()
        self.o.l.oracle_merkle_open_bytes(self.h, P(ix), C.c_size_t(len(ix)), buf, C.c_size_t(n)); return bytes(buf)
    def free(self): self.o.l.oracle_merkle_free(self.h)


class OProof:
    def __init__(self, o, h): self.o, self.h = o, h
    def bytes(self):
        n = self.o.l.oracle_proof_len(self.h); buf = (C.c_uint8 * n)(); self.o.l.oracle_proof_bytes(self.h, buf); return bytes(buf)
    def size_estimate(self): return self.o.l.oracle_proof_size_estimate(self.h)
    def num_layers(self): return self.o.l.oracle_proof_num_layers(self.h)
    def root(self, l):
        out = np.zeros(4, np.uint64); self.o.l.oracle_proof_root(self.h, l, P(out)); return out
    def layer_f(self, l):
        n = self.o.l.oracle_proof_layer_len(self.h, l); out = np.zeros((n, 4), np.uint64); self.o.l.oracle_proof_layer_f(self.h, l, P(out)); return out
    def z(self, l):
        out = np.zeros(4, np.uint64); self.o.l.oracle_proof_z(self.h, l, P(out)); return out
    def secs(self, which): return self.o.l.oracle_proof_secs(self.h, which)
    def free(self): self.o.l.oracle_proof_free(self.h)
