"""ctypes wrapper of stark_mlwe_amd/libstark_mlwe_hostcheck.so (diagnostic host build of the product's inline code)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.environ.get("STARK_HOSTCHECK_LIB") or os.path.join(ROOT, "stark_mlwe_amd", "libstark_mlwe_hostcheck.so")    # the override: a sanitizer build (tools/host_sanitize.sh)
vp = C.c_void_p


def P(a):
    return None if a is None else a.ctypes.data_as(vp)


def A(x):
    return np.ascontiguousarray(x, dtype=np.uint64)


class HostCheck:
    def __init__(self):
        self.l = C.CDLL(PATH)
        self.l.hc_params_new.restype = vp

    def fr_op(self, field, op, a, b=None):
        out = np.zeros(4, np.uint64); assert self.l.hc_fr_op(field, op, P(A(a)), P(None if b is None else A(b)), P(out)) == 0; return out

    def wide_dot(self, a, b):
        a, b = A(a), A(b); out = np.zeros(4, np.uint64); self.l.hc_wide_dot(P(a), P(b), C.c_size_t(a.shape[0]), P(out)); return out

    def ntt29(self, field, data, inverse=False):
        """one sub-NTT through the lazy nine-limb butterflies of ntt_dev.hpp (host instantiation); returns (result, max limb, max top limb)"""
        d = A(data).copy(); log_b = int(d.shape[0]).bit_length() - 1; ml, mt = C.c_uint64(0), C.c_uint64(0)
        assert self.l.hc_ntt29(field, P(d), log_b, 1 if inverse else 0, C.byref(ml), C.byref(mt)) == 0
        return d, ml.value, mt.value

    def partial_reduce(self, field, limbs):
        """fr29_partial_reduce on an (n, 9) uint32 array of lazy nine-limb values"""
        l = np.ascontiguousarray(limbs, dtype=np.uint32).copy()
        assert self.l.hc_partial_reduce(field, l.ctypes.data_as(vp), C.c_size_t(l.shape[0])) == 0; return l

    def wide_dot32(self, a, b):
        a, b = A(a), A(b); out = np.zeros(4, np.uint64); assert self.l.hc_wide_dot32(P(a), P(b), C.c_size_t(a.shape[0]), P(out)) == 0; return out

    def blake3(self, data: bytes):
        out = (C.c_uint8 * 32)(); buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data or b"\0")
        self.l.hc_blake3(buf, C.c_size_t(len(data)), out); return bytes(out)

    def chacha12_u64s(self, seed: bytes, n):
        out = np.zeros(n, np.uint64); buf = (C.c_uint8 * 32).from_buffer_copy(seed); self.l.hc_chacha12_u64s(buf, C.c_size_t(n), P(out)); return out

    def from_le_bytes_mod_order(self, b: bytes):
        out = np.zeros(4, np.uint64); buf = (C.c_uint8 * len(b)).from_buffer_copy(b); self.l.hc_from_le_bytes_mod_order(buf, C.c_size_t(len(b)), P(out)); return out

    def to_bytes_le(self, a):
        out = (C.c_uint8 * 32)(); self.l.hc_to_bytes_le(P(A(a)), out); return bytes(out)

    def params(self, kind, t=17, seed=b""):
        return vp(self.l.hc_params_new(kind, t, seed))

    def params_ok(self, h): return self.l.hc_params_ok(h)
    def params_free(self, h): self.l.hc_params_free(h)

    def params_export(self, h, t, rf, rp):
        mds = np.zeros((t * t, 4), np.uint64); rcf = np.zeros((rf * t, 4), np.uint64); rcp = np.zeros((rp, 4), np.uint64)
        self.l.hc_params_export(h, P(mds), P(rcf), P(rcp)); return mds, rcf, rcp

    def permute_kernel_form(self, h, states, t):
        s = A(states).copy(); self.l.hc_permute_kernel_form(h, P(s), C.c_size_t(s.size // (4 * t))); return s

    def full_round_linear(self, h, which, pre, states):
        """y = M x of t = 17 states of S-box outputs: which = 0 the L*U rows, 1 the emulated matrix-core path (fragment tables, fold)"""
        s = A(states).copy(); rc = self.l.hc_full_round_linear(h, which, 1 if pre else 0, P(s), C.c_size_t(s.size // (17 * 4)))
        assert rc == 0, rc; return s

    def row_consts(self):
        out = np.zeros(14, np.uint32); self.l.hc_row_consts(out.ctypes.data_as(C.c_void_p)); return out

    def chain_table(self, h, which):
        self.l.hc_chain_table.restype = C.c_size_t
        n = self.l.hc_chain_table(h, which, None, C.c_size_t(0)); out = np.zeros(n, np.uint32)
        self.l.hc_chain_table(h, which, out.ctypes.data_as(C.c_void_p), C.c_size_t(n)); return out

    def permute_chain_model(self, h, states, t=17):
        s = A(states).copy(); rc = self.l.hc_permute_chain_model(h, P(s), C.c_size_t(s.size // (4 * t))); assert rc == 0, rc; return s

    def permute_dense(self, h, states, t):
        s = A(states).copy(); self.l.hc_permute_dense(h, P(s), C.c_size_t(s.size // (4 * t))); return s

    def leaf_pair(self, h, f, f_next, m):
        f = A(f); out = np.zeros_like(f); self.l.hc_leaf_pair(h, P(f), P(None if f_next is None else A(f_next)), C.c_size_t(f.shape[0]), C.c_size_t(m), P(out)); return out

    def hash_ds_level(self, h, mode, arity, level, pos0, label, in0, in1=None):
        in0 = A(in0); n_in = in0.shape[0]; n_out = n_in if mode == 1 else (n_in + arity - 1) // arity; out = np.zeros((n_out, 4), np.uint64)
        self.l.hc_hash_ds_level(h, mode, C.c_size_t(arity), C.c_uint32(level), C.c_uint64(pos0), C.c_uint64(label), P(in0), P(None if in1 is None else A(in1)), C.c_size_t(n_in), P(out)); return out

    def tr_hash(self, h, tag: bytes, fields, n=1):
        f = A(fields); k = (f.shape[0] // n) if f.size else 0; out = np.zeros((n, 4), np.uint64)
        self.l.hc_tr_hash(h, tag, P(f), C.c_size_t(k), C.c_size_t(n), P(out)); return out[0] if n == 1 else out

    def hash_stream(self, h, mode, a, na, b, nb, tag=None, n=1):
        out = np.zeros((n, 4), np.uint64)
        self.l.hc_hash_stream(h, mode, P(A(a)) if na else None, C.c_size_t(na), P(A(b)) if nb else None, C.c_size_t(nb), P(None if tag is None else A(tag)), C.c_size_t(n), P(out))
        return out[0] if n == 1 else out

    # ---- query plan / assemble (fri_plan.hpp, host hasher) ----------------------------------------------
    def fri_plan(self, tparams, roots, n0, schedule, r):
        self.l.hc_fri_plan_create.restype = vp; self.l.hc_fri_plan_num_requests.restype = C.c_size_t; self.l.hc_fri_plan_assemble.restype = C.c_size_t
        sch = np.ascontiguousarray(schedule, dtype=np.uint64)
        h = self.l.hc_fri_plan_create(tparams, P(A(roots)), C.c_size_t(n0), P(sch), C.c_size_t(len(sch)), C.c_size_t(r))
        if not h:
            raise RuntimeError("plan failed")
        return HcPlan(self, vp(h))


def _hc_verify(self, tparams, proof: bytes, schedule, r):
    sch = np.ascontiguousarray(schedule, dtype=np.uint64); buf = (C.c_uint8 * max(1, len(proof))).from_buffer_copy(proof or b"\0")
    return self.l.hc_deep_fri_verify(tparams, buf, C.c_size_t(len(proof)), P(sch), C.c_size_t(len(sch)), C.c_size_t(r))


def _hc_merkle_verify(self, tparams, pairs, cfg_arity, label, root, idx, vals, cp, proof: bytes):
    ix = np.ascontiguousarray(idx, dtype=np.uint64); buf = (C.c_uint8 * max(1, len(proof))).from_buffer_copy(proof or b"\0")
    return self.l.hc_merkle_verify(tparams, 1 if pairs else 0, C.c_size_t(cfg_arity), C.c_uint64(label), P(A(root)), P(ix), C.c_size_t(len(ix)), P(A(vals)), P(None if cp is None else A(cp)), buf, C.c_size_t(len(proof)))


HostCheck.deep_fri_verify = _hc_verify
HostCheck.merkle_verify = _hc_merkle_verify


class HcPlan:
    def __init__(self, hc, h): self.hc, self.h = hc, h

    def requests(self):
        n = self.hc.l.hc_fri_plan_num_requests(self.h)
        kind = np.zeros(n, np.uint32); which = np.zeros(n, np.uint32); level = np.zeros(n, np.uint32); index = np.zeros(n, np.uint64)
        self.hc.l.hc_fri_plan_requests(self.h, P(kind), P(which), P(level), P(index)); return kind, which, level, index

    def assemble(self, values):
        v = A(values).reshape(-1, 4); est = C.c_size_t()
        ln = self.hc.l.hc_fri_plan_assemble(self.h, P(v), C.c_size_t(v.shape[0]), None, C.c_size_t(0), C.byref(est))
        if not ln:
            raise RuntimeError("assemble failed")
        buf = (C.c_uint8 * ln)()
        self.hc.l.hc_fri_plan_assemble(self.h, P(v), C.c_size_t(v.shape[0]), buf, C.c_size_t(ln), C.byref(est))
        return bytes(buf), est.value

    def free(self):
        if self.h: self.hc.l.hc_fri_plan_free(self.h); self.h = None
