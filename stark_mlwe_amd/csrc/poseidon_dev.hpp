// stark_mlwe_amd/csrc/poseidon_dev.hpp — Poseidon x^5 permutation and sponges on CDNA4 (gfx950).
//
// Device replacement for the reference's CPU inner loops
//   permute / permute_dynamic            crates/poseidon/src/lib.rs:31-68, 219-258
//   hash_with_ds_dynamic (eager sponge)  crates/poseidon/src/lib.rs:276-312
//   hash_with_ds (legacy sponge)         crates/poseidon/src/lib.rs:85-100
//   Transcript duplex (lazy permute)     crates/transcript/src/lib.rs:79-101
//   hash_leaf_pair                       crates/deep_ali/src/fri.rs:38-44
//
// Mapping: one lane owns one sponge (all 64 lanes busy in the partial rounds; round constants are
// wave-uniform, so they arrive through the scalar cache).  The t-element state lives in LDS in an
// [element][half][lane] layout (16-byte slots, lane-contiguous => conflict-free ds_read/write_b128),
// which keeps the round loops rolled (small I-cache footprint) and the VGPR count low.
// Dense MDS products are applied IN PLACE through the LU factors prepared on the host, the RP
// partial rounds through the sparse factorisation (host_util.hpp) — the same field values as the
// reference's dense rounds, hence identical bits.
//
// The sponge bodies are written against a state accessor `S` (ld/st) so that the very same code
// can be instantiated with a plain array on the host by the host-check library (CPU unit tests).
#pragma once
#include "fr.hpp"
#include "fr29.hpp"
#include "dev_common.hpp"
#include "poseidon_params.hpp"

namespace stark {

typedef PallasFr PF;   // the prover field

// Plain-array state (host-check library).
struct ArrayState {
    fr_t* a;
    FR_HD fr_t ld(int j) const { return a[j]; }
    FR_HD void st(int j, const fr_t& x) const { a[j] = x; }
};

// Dot product  sum_i c_i * x_i  with ONE Montgomery reduction: constants as nine 29-bit limbs in the 2^261
// domain, carry-free column sums (fr29.hpp).  A carry pass every 7 terms (Pallas) keeps the 64-bit columns in range;
// rows longer than 60 terms (t = 65, 129) are reduced in chunks.
struct DotAcc {
    fr_wide29 w; fr_t sum; int since, total; bool have_sum;
    FR_HD void init() { fr_wide29_zero(w); since = 0; total = 0; have_sum = false; }
    FR_HD void mac(const uint32_t* __restrict__ c29, const fr_t& x) {
        if (since == fr29_max_terms<PF>()) { fr_wide29_norm(w); since = 0; }
        fr_wide29_mac(w, c29, fr29_unpack(x)); ++since;
        if (++total == 60) { fr_t r = fr_wide29_reduce<PF>(w); sum = have_sum ? fr_add<PF>(sum, r) : r; have_sum = true; fr_wide29_zero(w); since = 0; total = 0; }
    }
    FR_HD fr_t finish() {
        if (!have_sum) return fr_wide29_reduce<PF>(w);
        return total ? fr_add<PF>(sum, fr_wide29_reduce<PF>(w)) : sum;
    }
};
FR_HD const uint32_t* c29(const uint32_t* base, size_t idx) { return base + 9 * idx; }

// y = L*(U*x) in place.  U: row i needs x[j>=i] (top-down); unit-lower L: row i needs y[j<i] (bottom-up).
// Every row is a dot product with wave-uniform constants: the 64-MAC partial products of all its terms
// are summed in one wide accumulator and reduced once (fr.hpp "wide").
template <class S> FR_HD void apply_lu(const S& s, const uint32_t* lu, int t) {
    for (int i = 0; i < t; ++i) {
        DotAcc d; d.init();
        for (int j = i; j < t; ++j) d.mac(c29(lu, i * t + j), s.ld(j));
        s.st(i, d.finish());
    }
    for (int i = t - 1; i >= 1; --i) {
        DotAcc d; d.init();
        for (int j = 0; j < i; ++j) d.mac(c29(lu, i * t + j), s.ld(j));
        s.st(i, fr_add<PF>(s.ld(i), d.finish()));
    }
}
// One Poseidon permutation of the state behind `s`.  Returns lane 0 of the result.  With `only0`
// the final MDS computes row 0 only (the sponge squeezes state[0]); the rest of the state is then dead.
template <class S> FR_HD fr_t permute_core(const S& s, const PoseidonDev& P, bool only0) {
    const int t = P.t, half = P.rf / 2;
    for (int r = 0; r < half; ++r) {
        for (int j = 0; j < t; ++j) s.st(j, fr_pow5_r29<PF>(fr_add<PF>(s.ld(j), P.rc_full[r * t + j])));   // x^5 / 2^20: the tables carry the 2^20
        apply_lu(s, (r == half - 1) ? P.lu_pre29 : P.lu29, t);
    }
    // Partial rounds in blocks of 4 (rp is a multiple of 4 for every supported width).  Within a block the
    // lanes 1..t-1 stay at their block-start value s_j; round q of the block computes
    //     x_q = (s0 + c_q)^5,   s0 <- a_q x_q + sum_{p<q} gamma_{q,p} x_p + sum_j u_{q,j} s_j      (ONE reduction)
    // and the lanes are brought up to date once per block:  s_j += sum_{p<4} w_{p,j} x_p           (one reduction per lane)
    // — the same values as updating s_j += w_j x after every round, with 20 reductions per block instead of 68.
    fr_t s0 = s.ld(0);
    const int w = 2 * t - 1;
    for (int b = 0; b < P.rp / 4; ++b) {
        const uint32_t* sp = c29(P.sparse29, (size_t)(4 * b) * w);
        const uint32_t* g = c29(P.gamma29, (size_t)b * 6);
        fr_t x0, x1, x2, x3;
#define STARK_PARTIAL_ROUND(q, XQ)                                                              \
        {                                                                                       \
            XQ = fr_pow5_r29<PF>(fr_add<PF>(s0, P.rc_partial[4 * b + q]));                       \
            DotAcc acc; acc.init();                                                             \
            acc.mac(c29(sp, q * w), XQ);                                                        \
            if (q > 0) acc.mac(c29(g, q * (q - 1) / 2 + 0), x0);                                 \
            if (q > 1) acc.mac(c29(g, q * (q - 1) / 2 + 1), x1);                                 \
            if (q > 2) acc.mac(c29(g, q * (q - 1) / 2 + 2), x2);                                 \
            for (int j = 1; j < t; ++j) acc.mac(c29(sp, q * w + j), s.ld(j));                    \
            s0 = acc.finish();                                                                  \
        }
        STARK_PARTIAL_ROUND(0, x0) STARK_PARTIAL_ROUND(1, x1) STARK_PARTIAL_ROUND(2, x2) STARK_PARTIAL_ROUND(3, x3)
#undef STARK_PARTIAL_ROUND
        for (int j = 1; j < t; ++j) {
            DotAcc u; u.init();
            u.mac(c29(sp, 0 * w + t - 1 + j), x0); u.mac(c29(sp, 1 * w + t - 1 + j), x1);
            u.mac(c29(sp, 2 * w + t - 1 + j), x2); u.mac(c29(sp, 3 * w + t - 1 + j), x3);
            s.st(j, fr_add<PF>(s.ld(j), u.finish()));
        }
    }
    s.st(0, s0);
    for (int r = half; r < P.rf; ++r) {
        for (int j = 0; j < t; ++j) s.st(j, fr_pow5_r29<PF>(fr_add<PF>(s.ld(j), P.rc_full[r * t + j])));   // x^5 / 2^20: the tables carry the 2^20
        if (only0 && r == P.rf - 1) {
            DotAcc acc; acc.init();
            for (int j = 0; j < t; ++j) acc.mac(c29(P.row0_29, j), s.ld(j));
            return acc.finish();
        }
        apply_lu(s, P.lu29, t);
    }
    return s.ld(0);
}

// ---- sponge bodies (one call = the work of one lane) ---------------------------------------------------
// hash_leaf_pair(f, s): one t=17 permutation of the constant template `init`
// with lanes 4,5 = (f, s).
template <class S> FR_HD fr_t leaf_pair_body(const S& s, const PoseidonDev& P, const fr_t* init, const fr_t& f, const fr_t& sv) {
    for (int j = 0; j < 17; ++j) s.st(j, init[j]);
    s.st(4, f); s.st(5, sv);
    return permute_core(s, P, true);
}
// hash_with_ds_dynamic([arity, level, pos0+k, label], children_k)  (merkle/src/lib.rs:167-176).
//   mode 0 (node level): children_k = in0[k*arity .. min((k+1)*arity, n_in))
//   mode 1 (pair leaf) : children_k = {in0[k], in1[k / cp_div]}  (merkle/src/lib.rs:380-388); in1 == nullptr: the second
//                        child is zero (fri.rs:266).  cp_div = m serves commit_pairs(f_l, s_l) with s_l the view f_{l+1}[i/m].
//   pos_list != nullptr: hash k carries DS position pos_list[k] instead of pos0 + k (the verifier's union-of-paths levels, whose
//                        parents are scattered; merkle/src/lib.rs:683-689).
struct DsJob { fr_t arity_f, level_f, label_f; uint64_t pos0; size_t arity, n_in, n_out; int mode; size_t cp_div = 1; const uint64_t* pos_list = nullptr; };
FR_HD uint64_t ds_position(const DsJob& J, size_t k) { return J.pos_list ? J.pos_list[k] : J.pos0 + k; }
FR_HD fr_t ds_pair_child(const DsJob& J, const fr_t* in0, const fr_t* in1, size_t k, size_t c) {
    if (c == 0) return ldg(in0 + k);
    return in1 ? ldg(in1 + k / J.cp_div) : fr_zero<PF>();
}
template <class S> FR_HD fr_t hash_ds_body(const S& s, const PoseidonDev& P, const DsJob& J, const fr_t* in0, const fr_t* in1, size_t k) {
    const int t = P.t, rate = t - 1;
    for (int j = 0; j < t; ++j) s.st(j, fr_zero<PF>());
    size_t cnt = J.mode == 1 ? 2 : ((k + 1) * J.arity <= J.n_in ? J.arity : J.n_in - k * J.arity);
    size_t total = 4 + cnt + 1;                       // ds || children || 1, then implicit zero padding
    size_t nperm = (total + rate - 1) / rate, done = 0;
    int cur = 0; fr_t res = fr_zero<PF>();
    for (size_t q = 0; q < total; ++q) {
        fr_t x;
        if (q == 0) x = J.arity_f; else if (q == 1) x = J.level_f; else if (q == 2) x = fr_from_u64<PF>(ds_position(J, k)); else if (q == 3) x = J.label_f;
        else if (q == total - 1) x = fr_one<PF>();
        else { size_t c = q - 4; x = J.mode == 1 ? ds_pair_child(J, in0, in1, k, c) : ldg(in0 + k * J.arity + c); }
        s.st(cur, fr_add<PF>(s.ld(cur), x));
        if (++cur == rate) { cur = 0; ++done; res = permute_core(s, P, done == nperm); }
    }
    if (cur != 0) res = permute_core(s, P, true);
    return res;
}
// Transcript-style duplex (t=17, rate 16, lazy permute) over the stream  prefix || fields_i || suffix,
// state[16] = cap.  Covers tr_hash_fields_tagged (fri.rs:28-35).
struct TrJob { const fr_t* prefix; int np; const fr_t* suffix; int ns; fr_t cap; size_t k; size_t n; };
template <class S> FR_HD fr_t tr_hash_body(const S& s, const PoseidonDev& P, const TrJob& J, const fr_t* fields, size_t i) {
    for (int j = 0; j < 16; ++j) s.st(j, fr_zero<PF>());
    s.st(16, J.cap);
    const size_t total = (size_t)J.np + J.k + (size_t)J.ns;
    int pos = 0;
    for (size_t q = 0; q < total; ++q) {
        fr_t x = q < (size_t)J.np ? J.prefix[q] : (q < (size_t)J.np + J.k ? ldg(fields + i * J.k + (q - J.np)) : J.suffix[q - J.np - J.k]);
        if (pos == 16) { permute_core(s, P, false); pos = 0; }
        s.st(pos, fr_add<PF>(s.ld(pos), x)); ++pos;
    }
    return permute_core(s, P, true);
}
// Generic eager sponge over explicit per-hash streams.
//   mode 0: hash_with_ds_dynamic — stream = a_k || b_k || 1, zero-padded (eager permute on a full rate).
//   mode 1: hash_with_ds (legacy) — state[t-1] = tag, b_k absorbed in rate-sized chunks, one permutation
//           per (possibly short) chunk, no padding.
template <class S> FR_HD fr_t hash_stream_body(const S& s, const PoseidonDev& P, int mode, const fr_t* a, size_t na, const fr_t* b, size_t nb, const fr_t& tag, size_t k) {
    const int t = P.t, rate = t - 1;
    for (int j = 0; j < t; ++j) s.st(j, fr_zero<PF>());
    int cur = 0;
    if (mode == 0) {
        const size_t total = na + nb + 1;
        for (size_t q = 0; q < total; ++q) {
            fr_t x = q < na ? ldg(a + k * na + q) : (q < na + nb ? ldg(b + k * nb + (q - na)) : fr_one<PF>());
            s.st(cur, fr_add<PF>(s.ld(cur), x));
            if (++cur == rate) { cur = 0; permute_core(s, P, false); }
        }
        if (cur != 0) permute_core(s, P, false);
    } else {
        s.st(t - 1, tag);
        for (size_t q = 0; q < nb; ++q) {
            s.st(cur, fr_add<PF>(s.ld(cur), ldg(b + k * nb + q)));
            if (++cur == rate || q + 1 == nb) { cur = 0; permute_core(s, P, false); }
        }
    }
    return s.ld(0);
}

#if defined(__HIPCC__)
// ---- LDS-resident per-lane state ---------------------------------------------------------------------
struct LdsState {
    uint4* base; int nl; int lane;
    __device__ __forceinline__ fr_t ld(int j) const {
        uint4 lo = base[(2 * j) * nl + lane], hi = base[(2 * j + 1) * nl + lane];
        fr_t x; x.v[0] = lo.x; x.v[1] = lo.y; x.v[2] = lo.z; x.v[3] = lo.w; x.v[4] = hi.x; x.v[5] = hi.y; x.v[6] = hi.z; x.v[7] = hi.w; return x;
    }
    __device__ __forceinline__ void st(int j, const fr_t& x) const {
        base[(2 * j) * nl + lane] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
        base[(2 * j + 1) * nl + lane] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
    }
};
// ---- kernels ---------------------------------------------------------------------------------------
// K3: h[i] = hash_leaf_pair(f[i], s_i), s_i = f_next[i/m] (zero when f_next == nullptr: fri.rs:266).
__global__ void __launch_bounds__(64) k_leaf_pair(PoseidonDev P, const fr_t* __restrict__ init, const fr_t* __restrict__ f,
                                                  const fr_t* __restrict__ f_next, size_t n, size_t m, fr_t* __restrict__ h) {
    extern __shared__ uint4 lds[];
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    LdsState s{lds, (int)blockDim.x, (int)threadIdx.x};
    stg(h + i, leaf_pair_body(s, P, init, ldg(f + i), f_next ? ldg(f_next + i / m) : fr_zero<PF>()));
}
// K4: one Merkle level / the pair-leaf level.
__global__ void __launch_bounds__(64) k_hash_ds(PoseidonDev P, DsJob J, const fr_t* __restrict__ in0, const fr_t* __restrict__ in1, fr_t* __restrict__ out) {
    extern __shared__ uint4 lds[];
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= J.n_out) return;
    LdsState s{lds, (int)blockDim.x, (int)threadIdx.x};
    stg(out + k, hash_ds_body(s, P, J, in0, in1, k));
}
// permute / permute_dynamic over a batch of AoS states (poseidon/src/lib.rs:31,219).
__global__ void __launch_bounds__(64) k_permute_batch(PoseidonDev P, fr_t* __restrict__ states, size_t n) {
    extern __shared__ uint4 lds[];
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    LdsState s{lds, (int)blockDim.x, (int)threadIdx.x};
    for (int j = 0; j < P.t; ++j) s.st(j, ldg(states + i * P.t + j));
    permute_core(s, P, false);
    for (int j = 0; j < P.t; ++j) stg(states + i * P.t + j, s.ld(j));
}
// Batch of independent transcript hashes (index seeds, z seeds); with n == 1 a whole column
// (the serial sponge of build_f0, fri.rs:551-554).
__global__ void __launch_bounds__(64) k_tr_hash(PoseidonDev P, TrJob J, const fr_t* __restrict__ fields, fr_t* __restrict__ out) {
    extern __shared__ uint4 lds[];
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= J.n) return;
    LdsState s{lds, (int)blockDim.x, (int)threadIdx.x};
    stg(out + i, tr_hash_body(s, P, J, fields, i));
}
__global__ void __launch_bounds__(64) k_hash_stream(PoseidonDev P, int mode, const fr_t* __restrict__ a, size_t na, const fr_t* __restrict__ b, size_t nb,
                                                    fr_t tag, size_t n, fr_t* __restrict__ out) {
    extern __shared__ uint4 lds[];
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    LdsState s{lds, (int)blockDim.x, (int)threadIdx.x};
    stg(out + k, hash_stream_body(s, P, mode, a, na, b, nb, tag, k));
}
#endif  // __HIPCC__

}  // namespace stark
