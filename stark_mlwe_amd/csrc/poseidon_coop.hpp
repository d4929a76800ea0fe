// stark_mlwe_amd/csrc/poseidon_coop.hpp — ONE wave per sponge: the latency-oriented form (gfx950).
//
// For chains that cannot be batched: the four serial column sponges of DeepAliRealBuilder::build_f0
// (crates/deep_ali/src/fri.rs:548-557 via tr_hash_fields_tagged, fri.rs:28-35 — n0/16 DEPENDENT t=17
// permutations per column) and the single Fiat-Shamir hashes (fri_sample_z_ell, fs_seed_from_roots,
// ali_sample_z_beta_fs).  A lane-per-sponge kernel would run such a chain on one lane (~1.4 ms per
// permutation measured); here the 64 lanes of one wave share one permutation:
//   * lane j < 17 holds state element j in registers;
//   * full rounds: S-boxes in parallel (3 product latencies), then the dense MDS with the row split
//     three ways over lanes (i, q), q = 0..2 (51 lanes): each sums ~6 terms of row i in a wide
//     accumulator, the three partials are added across lanes; the MDS matrix sits in LDS;
//   * partial rounds, in blocks of 4: the dot products of all four rounds are taken at once against the block-start
//     lanes (64 products in one slot + a butterfly sum), and a round is three product latencies: x^2 on one lane while
//     the others form c*x, then x^4, then (c*x)*x^4 = c*x^5 for c in {w_j, a, gamma} — x^5 itself is never formed.
// Same field values as the reference's dense rounds (the kernel-form constants of host_util.hpp).
#pragma once
#include "fr.hpp"
#include "dev_common.hpp"
#include "poseidon_params.hpp"
#include "poseidon_dev.hpp"   // TrJob, c29
#include "fr29.hpp"

#if defined(__HIPCC__)
namespace stark {

__device__ __forceinline__ fr_t shfl_fr(const fr_t& x, int src) {
    fr_t r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = (uint32_t)__shfl((int)x.v[i], src, 64);
    return r;
}
__device__ __forceinline__ fr_t shfl_down_fr(const fr_t& x, int d) {
    fr_t r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = (uint32_t)__shfl_down((int)x.v[i], d, 64);
    return r;
}
// broadcast of one lane's element through SGPRs (v_readlane: no LDS-crossbar round trip); src is wave-uniform
__device__ __forceinline__ fr_t bcast_fr(const fr_t& x, int src) {
    fr_t r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = (uint32_t)__builtin_amdgcn_readlane((int)x.v[i], src);
    return r;
}
// ---- nine-limb (radix 2^29) values on the dependent chain -------------------------------------------------------
// On a lone wave a radix-2^32 Montgomery product in a dependent chain takes 544 ns, the radix-2^29 product on nine-limb
// values 402 ns and the square 348 ns (tools/chain_bench.hip): no carry instructions, and values stay "lazy" (any value
// below 2^261 with limbs below 2^29 is a valid operand; a product of operands below 2^5 r comes out below ~9 r).
__device__ __forceinline__ fr29_t shfl29(const fr29_t& x, int src) {
    fr29_t r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = (uint32_t)__shfl((int)x.l[i], src, 64);
    return r;
}
__device__ __forceinline__ fr29_t shfl_xor29(const fr29_t& x, int m) {
    fr29_t r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = (uint32_t)__shfl_xor((int)x.l[i], m, 64);
    return r;
}
__device__ __forceinline__ fr29_t bcast29(const fr29_t& x, int src) {
    fr29_t r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        r.l[i] = (uint32_t)__builtin_amdgcn_readlane((int)x.l[i], src);
        asm volatile("" : "+v"(r.l[i]));       // keep the copy in a VGPR: otherwise the compiler computes the next product on the (slower) scalar unit
    }
    return r;
}
__device__ __forceinline__ void carry29(fr29_t& a) {                  // limbs back below 2^29 (signed: limbs may be negative, the value is not)
#pragma unroll
    for (int i = 0; i < 8; ++i) { a.l[i + 1] += (uint32_t)((int32_t)a.l[i] >> 29); a.l[i] &= FR_M29; }
}
__device__ __forceinline__ fr29_t add29(const fr29_t& a, const fr29_t& b) {     // limb-wise, no carry pass: at most 3 of these in a row
    fr29_t r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
}
// value below 2^261 -> below 2r + epsilon, still >= 0: subtract k * r with k = floor(value / 2^254) - 1  (r = 2^254 + t, t < 2^126).
// 64-bit signed limbs during the pass: k * r_i reaches 2^36.
template <class F> __device__ __forceinline__ void lazy_reduce29(fr29_t& a) {
    const uint32_t q = a.l[8] >> 22; const int64_t k = q ? (int64_t)q - 1 : 0;
    int64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int64_t d = (int64_t)a.l[i] - k * (int64_t)fr_p29<F>(i) + carry;
        if (i < 8) { a.l[i] = (uint32_t)d & FR_M29; carry = d >> 29; } else a.l[8] = (uint32_t)d;
    }
}
template <class F> __device__ __forceinline__ fr29_t ld29(const uint32_t* p) {   // nine limbs of a table entry; p == nullptr: zero
    fr29_t r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = p ? p[i] : 0u;
    return r;
}
struct CoopLds { uint32_t* mds; uint32_t* x; };   // [T*T][9] (nine 29-bit limbs, 2^261 domain, x 2^20), [T][9] (S-box outputs as nine 29-bit limbs)
__device__ __forceinline__ fr_t lds_get(const uint4* base, int idx) {
    uint4 lo = base[2 * idx], hi = base[2 * idx + 1];
    fr_t x; x.v[0] = lo.x; x.v[1] = lo.y; x.v[2] = lo.z; x.v[3] = lo.w; x.v[4] = hi.x; x.v[5] = hi.y; x.v[6] = hi.z; x.v[7] = hi.w; return x;
}
__device__ __forceinline__ void lds_put(uint4* base, int idx, const fr_t& x) {
    base[2 * idx] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]); base[2 * idx + 1] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
}
// LDS: x first (16-byte slots), then M as plain words, 36 B per entry (10.7 KiB for t = 17: LDS never limits the number of
// one-wave workgroups per CU).  B_1*M is used by ONE of the eight full rounds and is read from global memory there.
__host__ __device__ static inline size_t coop_lds_bytes(int t) { return (size_t)(t * 9 + 3) / 4 * 16 + (size_t)t * t * 9 * 4; }
__device__ __forceinline__ void lds_get29(const uint32_t* base, int idx, uint32_t* a) {
#pragma unroll
    for (int i = 0; i < 9; ++i) a[i] = base[9 * idx + i];
}
template <int T> __device__ __forceinline__ CoopLds coop_setup(uint4* lds, const PoseidonDev& P) {
    uint32_t* m = reinterpret_cast<uint32_t*>(lds + (T * 9 + 3) / 4);
    CoopLds L{m, reinterpret_cast<uint32_t*>(lds)};
    for (int k = threadIdx.x; k < T * T * 9; k += blockDim.x) L.mds[k] = P.mds29[k];                            // coalesced word copies
    __syncthreads();
    return L;
}

// One full round on one wave: S-boxes on the element lanes (the result stays in nine-limb form in LDS), then the dense MDS with every row
// split NS ways over the lanes.  s: this lane's state element (lanes 0..T-1 meaningful), updated in place.
template <int T>
__device__ __forceinline__ void coop_full_round(fr_t& s, int r, const uint32_t* M, const bool in_lds, const PoseidonDev& P, const CoopLds& L, int lane) {
    constexpr int NS = 64 / T, PER = (T + NS - 1) / NS;
    const int row = lane % T, q = lane / T;
    const int j0 = q * PER, j1 = (j0 + PER < T) ? j0 + PER : T;
    const bool elem = lane < T;
    // S-box on the element lanes; the result stays in nine-limb form (below 1.01 r: a valid multiplier operand) and goes to LDS
    // as such — no pack / conditional subtraction here, no unpack in front of each of the row segment's terms
    if (elem) {
        const fr29_t u = fr29_unpack(fr_add<PF>(s, ldg(P.rc_full + r * T + lane)));
        const fr29_t x2 = fr29_sqr_mont<PF, true>(u), x4 = fr29_sqr_mont<PF, true>(x2), x5 = fr29_mul_mont<PF, true>(u, x4);     // x^5 / 2^20: the matrices carry the 2^20
#pragma unroll
        for (int i = 0; i < 9; ++i) L.x[9 * lane + i] = x5.l[i];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    fr_t part = fr_zero<PF>();
    if (q < NS && j0 < T) {
        static_assert(PER <= fr29_max_terms<PF>(), "row segment within one carry-free run");
        fr_wide29 acc; fr_wide29_zero(acc);
        for (int j = j0; j < j1; ++j) {
            uint32_t a[9];
            if (in_lds) lds_get29(M, row * T + j, a);
            else { _Pragma("unroll") for (int i = 0; i < 9; ++i) a[i] = P.mds_pre29[9 * (row * T + j) + i]; }
            fr29_t xj;
#pragma unroll
            for (int i = 0; i < 9; ++i) xj.l[i] = L.x[9 * j + i];
            fr_wide29_mac_regs(acc, a, xj);
        }
        part = fr_wide29_reduce<PF, true>(acc);
    }
    fr_t tot = part;
#pragma unroll
    for (int k = 1; k < NS; ++k) { fr_t o = shfl_fr(part, (lane + k * T) & 63); tot = fr_add<PF>(tot, o); }
    if (elem) s = tot;
    __builtin_amdgcn_wave_barrier();
}

// s: this lane's state element (lanes 0..T-1 meaningful).  Returns the permuted element.
// T = 17: rows split 3 ways (51 lanes); T = 9: 7 ways (63 lanes).
template <int T>
__device__ __forceinline__ fr_t coop_permute(fr_t s, const PoseidonDev& P, const CoopLds& L, int lane) {
    const int half = P.rf / 2, w = 2 * T - 1;
    auto full_round = [&](int r, const uint32_t* M, const bool in_lds) { coop_full_round<T>(s, r, M, in_lds, P, L, lane); };
    for (int r = 0; r < half; ++r) full_round(r, L.mds, r != half - 1);
    // Partial rounds in blocks of 4 (the algebra of permute_core, poseidon_dev.hpp), three product latencies per round:
    //   block start : all 4*(T-1) products u_{q,j}*s_j at once (lane (q, j)), butterfly-summed per round -> D_q, parked
    //                 in the "accumulator" lanes 32+q of the state register;
    //   round q     : x = s0 + c.  slot 1: lane 48 squares x while lane j forms w_{q,j}*x and accumulator lane 32+q'
    //                 forms a_q*x (q' = q) or gamma_{q',q}*x (q' > q);  slot 2: x^4 on every lane;  slot 3: (slot 1) * x^4,
    //                 i.e. w x^5 / a x^5 / gamma x^5 without ever forming x^5 itself; one lane-wise add brings the lanes
    //                 s_j and the accumulators up to date, and accumulator lane 32+q now holds the next s0.
    // The chain runs on nine-limb values (see above): the multiplier tables are the radix-2^29 ones of the throughput
    // kernels (R' = 2^261 domain; a, w, gamma carry the 2^20 that x^2, x^4 and the last product lose: 2^5 each, compounded).
    constexpr int RATE = T - 1, LOG_RATE = RATE == 16 ? 4 : 3;
    static_assert(RATE == 16 || RATE == 8, "cooperative form: t = 17 or t = 9");
    fr29_t sl = fr29_unpack(s);                                                       // this lane's element; lanes 32..35: accumulators
    fr29_t s0l = bcast29(sl, 0);                                                      // s0, replicated on every lane
    const bool acc_lane = lane >= 32 && lane < 36, sq_lane = lane == 48;
    { const fr29_t c0 = fr29_unpack(ldg(P.rc_partial)); s0l = add29(s0l, c0); carry29(s0l); }      // x of the very first partial round
    for (int b = 0; b < P.rp / 4; ++b) {
        const size_t r0 = (size_t)(4 * b) * w;
        // this lane's multiplier for each of the four rounds, fetched up front (off the dependent chain)
        fr29_t cst[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t* p = nullptr;
            if (lane >= 1 && lane < T) p = c29(P.sparse29, r0 + q * w + T - 1 + lane);                       // w_{q,lane} * 2^20
            else if (acc_lane) { const int qq = lane - 32; if (qq == q) p = c29(P.sparse29, r0 + q * w); else if (qq > q) p = c29(P.gamma29, b * 6 + qq * (qq - 1) / 2 + q); }
            cst[q] = ld29<PF>(p);
        }
        const int rc_idx = 4 * b + (lane - 31);                                       // lane 32 -> round 4b+1, ..., lane 35 -> round 4b+4 (next block's first)
        const bool rc_ok = acc_lane && rc_idx < P.rp;
        const fr29_t rcu = fr29_unpack(ldg(P.rc_partial + (rc_ok ? rc_idx : 0)));        // one per-lane load, in flight during the products below
        {   // D_q = sum_j u_{q,j} s_j from the block-start lanes
            const int dq = lane >> LOG_RATE, dj = 1 + (lane & (RATE - 1));
            const fr29_t sj = shfl29(sl, dj);
            fr29_t v = fr29_mul_mont<PF, true>(ld29<PF>(dq < 4 ? c29(P.sparse29, r0 + dq * w + dj) : nullptr), sj);   // u_{q,j}: not scaled
            int pending = 0;
#pragma unroll
            for (int d = RATE / 2; d >= 1; d >>= 1) { v = add29(v, shfl_xor29(v, d)); if (++pending == 2) { carry29(v); pending = 0; } }
            if (pending) carry29(v);
            const fr29_t dv = shfl29(v, (lane - 32) << LOG_RATE);                     // lanes 32..35 fetch D_0..D_3 (other lanes: unused)
            // The round constant of the round that will READ accumulator lane 32+q as its s0 (round q+1 of this block; for q = 3
            // the first round of the next block) is added here, off the dependent chain: x = s0 + c then needs no add and no
            // carry pass of its own inside the round.
            fr29_t rcn;
#pragma unroll
            for (int i = 0; i < 9; ++i) rcn.l[i] = rc_ok ? rcu.l[i] : 0u;
#pragma unroll
            for (int i = 0; i < 9; ++i) sl.l[i] = acc_lane ? dv.l[i] + rcn.l[i] : sl.l[i];     // limbs < 2^30 until the first round's carry pass
        }
#define STARK_COOP_ROUND(q)                                                                       \
        {                                                                                         \
            const fr29_t x = s0l;                                   /* s0 + c: the constant is already in (block start) */ \
            fr29_t a1;                                                                            \
            _Pragma("unroll") for (int i = 0; i < 9; ++i) a1.l[i] = sq_lane ? x.l[i] : cst[q].l[i]; \
            const fr29_t m1 = fr29_mul_mont<PF, true>(a1, x);                        /* slot 1 */         \
            const fr29_t x2 = bcast29(m1, 48);                                                    \
            const fr29_t x4 = fr29_sqr_mont<PF, true>(x2);                           /* slot 2 */         \
            const fr29_t m3 = fr29_mul_mont<PF, true>(m1, x4);                       /* slot 3 (zero where cst is zero) */ \
            sl = add29(sl, m3); carry29(sl);                                                      \
            s0l = bcast29(sl, 32 + q);                                                            \
        }
        STARK_COOP_ROUND(0) STARK_COOP_ROUND(1) STARK_COOP_ROUND(2) STARK_COOP_ROUND(3)
#undef STARK_COOP_ROUND
        lazy_reduce29<PF>(sl);                                                        // the lanes gained ~4r in this block
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) sl.l[i] = lane == 0 ? s0l.l[i] : sl.l[i];
    lazy_reduce29<PF>(sl);
    {   // below 2r + epsilon: pack, then two conditional subtractions (the second one fires with probability ~2^-125)
        uint32_t tt[9];
#pragma unroll
        for (int wd = 0; wd < 8; ++wd) {
            const int lo = 32 * wd, i = lo / 29, sh = lo - 29 * i;
            uint32_t v = sl.l[i] >> sh;
            if (i + 1 < 9) v |= sl.l[i + 1] << (29 - sh);
            if (29 - sh + 29 < 32 && i + 2 < 9) v |= sl.l[i + 2] << (58 - sh);
            tt[wd] = v;
        }
        tt[8] = 0;
        fr_cond_sub<PF>(tt, 0u); fr_cond_sub<PF>(tt, 0u);
#pragma unroll
        for (int i = 0; i < 8; ++i) s.v[i] = tt[i];
    }
    for (int r = half; r < P.rf; ++r) full_round(r, L.mds, true);
    return s;
}

// tr_hash_fields_tagged over the stream prefix || fields_i || suffix, one 64-lane block per hash i.
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) k_tr_hash_coop(PoseidonDev P, TrJob J, const fr_t* __restrict__ fields, fr_t* __restrict__ out) {
    extern __shared__ uint4 lds[];
    CoopLds L = coop_setup<17>(lds, P);
    const int lane = threadIdx.x; const size_t i = blockIdx.x;
    fr_t s = lane == 16 ? J.cap : fr_zero<PF>();
    const size_t total = (size_t)J.np + J.k + (size_t)J.ns;
    for (size_t base = 0; base < total; base += 16) {                                // one rate-16 block per iteration
        if (base) s = coop_permute<17>(s, P, L, lane);                               // lazy permute: only before absorbing more
        const size_t e = base + lane;
        if (lane < 16 && e < total) {
            fr_t x = e < (size_t)J.np ? ldg(J.prefix + e) : (e < (size_t)J.np + J.k ? ldg(fields + i * J.k + (e - J.np)) : ldg(J.suffix + (e - J.np - J.k)));
            s = fr_add<PF>(s, x);
        }
    }
    s = coop_permute<17>(s, P, L, lane);
    if (lane == 0) stg(out + i, s);
}

// hash_leaf_pair (fri.rs:38-44), one wave per leaf: for layers too long for the five-wave kernel's one workgroup per CU and too short to fill the wave-pair
// throughput kernel, whose launch takes 0.77 ms whatever its size (2049 .. 8192 leaves).  init = the 17-element template of capi_core.hip ctx_leaf_init.
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) k_leaf_pair_coop(PoseidonDev P, const fr_t* __restrict__ init, const fr_t* __restrict__ f,
                                                                                                 const fr_t* __restrict__ f_next, size_t m, fr_t* __restrict__ h) {
    extern __shared__ uint4 lds[];
    CoopLds L = coop_setup<17>(lds, P);
    const int lane = threadIdx.x; const size_t i = blockIdx.x;
    fr_t s = fr_zero<PF>();
    if (lane < 17) s = lane == 4 ? ldg(f + i) : (lane == 5 ? (f_next ? ldg(f_next + i / m) : fr_zero<PF>()) : ldg(init + lane));
    s = coop_permute<17>(s, P, L, lane);
    if (lane == 0) stg(h + i, s);
}

// Up to 4 independent long sponges in ONE launch (one block each): the four column chains of build_f0.
// With `batch` set, block b hashes column b & 3 of trace b >> 2 (stark_deep_fri_prove_batch_dev: the 4 * B chains of B independent traces
// in one launch — each is serial, together they fill the chip); its fields pointer comes from the device array batch[b].
// With `stride` set (and batch null), block b hashes the k[0] fields at fields[0] + b * stride under tag 0 (tr_hash_dev: n equal-length sponges).
struct TrMultiJob { const fr_t* prefix[4]; int np[4]; const fr_t* suffix[4]; int ns[4]; const fr_t* fields[4]; size_t k[4]; fr_t cap; const fr_t* const* batch; size_t stride; };
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) k_tr_hash_coop_multi(PoseidonDev P, TrMultiJob J, fr_t* __restrict__ out) {
    extern __shared__ uint4 lds[];
    CoopLds L = coop_setup<17>(lds, P);
    const int lane = threadIdx.x, b = blockIdx.x, c = J.batch ? (b & 3) : (J.stride ? 0 : b);
    const fr_t* prefix = J.prefix[c]; const fr_t* suffix = J.suffix[c]; const fr_t* fields = J.batch ? J.batch[b] : (J.stride ? J.fields[0] + (size_t)b * J.stride : J.fields[c]);
    const size_t np = J.np[c], kk = J.k[c], total = np + kk + (size_t)J.ns[c];
    fr_t s = lane == 16 ? J.cap : fr_zero<PF>();
    auto fetch = [&](size_t base) -> fr_t {                                                // this lane's element of the rate block starting at `base`
        const size_t e = base + lane;
        if (lane < 16 && e < total) return e < np ? ldg(prefix + e) : (e < np + kk ? ldg(fields + (e - np)) : ldg(suffix + (e - np - kk)));
        return fr_zero<PF>();
    };
    fr_t nxt = fetch(0);
    for (size_t base = 0; base < total; base += 16) {
        const fr_t cur = nxt;
        if (base + 16 < total) nxt = fetch(base + 16);                                    // issued BEFORE the permutation: the HBM latency of the next block hides under it
        if (base) s = coop_permute<17>(s, P, L, lane);
        s = fr_add<PF>(s, cur);
    }
    s = coop_permute<17>(s, P, L, lane);
    if (lane == 0) stg(out + b, s);
}

// One Merkle node per wave (small levels: latency matters, not throughput).  Same job as k_hash_ds.
template <int T>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) k_hash_ds_coop(PoseidonDev P, DsJob J, const fr_t* __restrict__ in0, const fr_t* __restrict__ in1, fr_t* __restrict__ out) {
    extern __shared__ uint4 lds[];
    CoopLds L = coop_setup<T>(lds, P);
    const int lane = threadIdx.x, rate = T - 1; const size_t k = blockIdx.x;
    const size_t cnt = J.mode == 1 ? 2 : ((k + 1) * J.arity <= J.n_in ? J.arity : J.n_in - k * J.arity);
    const size_t total = 4 + cnt + 1;                                                // ds || children || 1, zero padded
    fr_t s = fr_zero<PF>();
    for (size_t base = 0; base < total; base += rate) {                              // eager sponge: permute after every full (or final) block
        const size_t q = base + lane;
        if (lane < rate && q < total) {
            fr_t x;
            if (q == 0) x = J.arity_f; else if (q == 1) x = J.level_f; else if (q == 2) x = fr_from_u64<PF>(ds_position(J, k)); else if (q == 3) x = J.label_f;
            else if (q == total - 1) x = fr_one<PF>();
            else { size_t c = q - 4; x = J.mode == 1 ? ds_pair_child(J, in0, in1, k, c) : ldg(in0 + k * J.arity + c); }
            s = fr_add<PF>(s, x);
        }
        s = coop_permute<T>(s, P, L, lane);
    }
    if (lane == 0) stg(out + k, s);
}

}  // namespace stark
#endif
