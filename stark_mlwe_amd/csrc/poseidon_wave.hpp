// stark_mlwe_amd/csrc/poseidon_wave.hpp — wide Poseidon states (t = 33, 65, 129: Merkle arities 32, 64, 128), ONE WAVE per sponge (gfx950).
//
// The reference's other bench schedules (crates/channel/benches/end_to_end.rs:195-210: uni32x3, uni64x2x8, hi64_32_8, the 128-fold ones) commit their
// layers with arity 32 / 64 / 128, i.e. hash_with_ds_dynamic over t = 33 / 65 / 129 (crates/poseidon/src/lib.rs:155-166, 219-312).  One lane per sponge
// (poseidon_dev.hpp) is all LDS allows for such states in a lane-per-sponge layout (a t = 65 state is 2 KB: one wave per CU), and a lane walks the 43 680
// products of a t = 65 permutation alone: 17 ms — every tree level, however small, costs two of those.  Here the 64 lanes of a wave share one sponge:
//   * state in LDS as nine 29-bit limbs per element, lazily reduced (limbs below 2^29, values below 2^261; never canonical inside a permutation);
//   * full rounds: S-boxes lane-parallel (lane l: elements l, l + 64, l + 128); the dense product cut into 8 t work items (row i, eighth of the row),
//     dealt round-robin to the lanes, wide-accumulated and Montgomery-reduced per item, the eight partials of a row summed by the row's lane;
//   * partial rounds in the sparse form of host_util.hpp, one at a time: every lane forms y = (s0 + c)^5 for itself (three products, no broadcast to wait
//     for), then u_j s_j and s_j += w_j y for its elements; the dot product's lane partials meet in a six-step butterfly.
// ~0.4 ms per t = 65 permutation instead of 17 ms; the same field values as the reference's dense rounds (GPU tests against the oracle: wide-arity
// Merkle trees, fri_build with 128-fold layers, the presets' goldens).
#pragma once
#include "fr.hpp"
#include "fr29.hpp"
#include "dev_common.hpp"
#include "poseidon_params.hpp"
#include "poseidon_dev.hpp"    // DsJob, ds_position
#include "poseidon_coop.hpp"   // nine-limb helpers (add29, carry29, lazy_reduce29, shfl_xor29)

#if defined(__HIPCC__)
namespace stark {

template <int T> struct WaveCfg {
    static constexpr int SEG = 8, TS = (T + SEG - 1) / SEG;            // a row's eighth: TS terms
    static constexpr int WORDS = T * 9 * 2 + T * SEG * 9;             // state, S-box outputs, row partials
    __host__ __device__ static constexpr size_t lds_bytes() { return (size_t)(WORDS + 3) / 4 * 16; }
};
static inline size_t wave_lds_bytes(int t) { return t == 33 ? WaveCfg<33>::lds_bytes() : t == 65 ? WaveCfg<65>::lds_bytes() : WaveCfg<129>::lds_bytes(); }

struct WaveLds { uint32_t* st; uint32_t* x; uint32_t* part; };
__device__ __forceinline__ fr29_t wv_ld(const uint32_t* p) { fr29_t r; _Pragma("unroll") for (int i = 0; i < 9; ++i) r.l[i] = p[i]; return r; }
__device__ __forceinline__ void wv_st(uint32_t* p, const fr29_t& v) { _Pragma("unroll") for (int i = 0; i < 9; ++i) p[i] = v.l[i]; }
__device__ __forceinline__ void wv_sync() { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); }      // lgkmcnt(0): one wave, LDS in order

// x (lazy, below ~20 r) -> x^5 / 2^20, below 1.01 r (the matrices carry the 2^20, host_util.hpp to_radix29)
__device__ __forceinline__ fr29_t wv_pow5(const fr29_t& u) {
    const fr29_t x2 = fr29_sqr_mont<PF, true>(u), x4 = fr29_sqr_mont<PF, true>(x2);
    return fr29_mul_mont<PF, true>(u, x4);
}

template <int T>
__device__ __forceinline__ void wave_full_round(const WaveLds& L, const PoseidonDev& P, int r, const uint32_t* M29, int lane) {
    typedef WaveCfg<T> Cfg;
    for (int e = lane; e < T; e += 64) {
        fr29_t u = add29(wv_ld(L.st + 9 * e), fr29_unpack(ldg(P.rc_full + r * T + e))); carry29(u);
        wv_st(L.x + 9 * e, wv_pow5(u));
    }
    wv_sync();
    for (int it = lane; it < T * Cfg::SEG; it += 64) {
        const int i = it / Cfg::SEG, seg = it - i * Cfg::SEG, j0 = seg * Cfg::TS, j1 = (j0 + Cfg::TS < T) ? j0 + Cfg::TS : T;
        fr_wide29 acc; fr_wide29_zero(acc);
        int run = 0;
        for (int j = j0; j < j1; ++j) {
            uint32_t a[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) a[k] = M29[9 * ((size_t)i * T + j) + k];
            fr_wide29_mac_regs(acc, a, wv_ld(L.x + 9 * j));
            if (++run == fr29_max_terms<PF>()) { fr_wide29_norm(acc); run = 0; }
        }
        fr29_t p; fr_wide29_mont<PF, true>(acc, p.l);
        wv_st(L.part + 9 * it, p);
    }
    wv_sync();
    for (int i = lane; i < T; i += 64) {
        fr29_t s = wv_ld(L.part + 9 * (i * Cfg::SEG));
#pragma unroll
        for (int seg = 1; seg < Cfg::SEG; ++seg) { s = add29(s, wv_ld(L.part + 9 * (i * Cfg::SEG + seg))); if (!(seg & 1)) carry29(s); }   // carry29 is a signed pass: limbs stay below 2^31
        carry29(s);
        wv_st(L.st + 9 * i, s);
    }
    wv_sync();
}

// One permutation of the LDS-resident state by the wave (kernel-form constants: dense M and B_1 M for the full rounds, [a, u, w] for the partial ones).
template <int T>
__device__ __forceinline__ void wave_permute(const WaveLds& L, const PoseidonDev& P, int lane) {
    const int half = P.rf / 2, W = 2 * T - 1;
    for (int r = 0; r < half; ++r) wave_full_round<T>(L, P, r, r == half - 1 ? P.mds_pre29 : P.mds29, lane);
    constexpr int EPL = (T - 1 + 63) / 64;                                  // elements 1..T-1 of a lane: j = 1 + lane + 64 k
    fr29_t sj[EPL];
#pragma unroll
    for (int k = 0; k < EPL; ++k) { const int j = 1 + lane + 64 * k; if (j < T) sj[k] = wv_ld(L.st + 9 * j); else { _Pragma("unroll") for (int i = 0; i < 9; ++i) sj[k].l[i] = 0; } }
    fr29_t s0 = wv_ld(L.st);                                               // every lane carries its own copy of s0
#pragma unroll 1
    for (int q = 0; q < P.rp; ++q) {
        const uint32_t* sp = P.sparse29 + 9 * (size_t)q * W;
        fr29_t x = add29(s0, fr29_unpack(ldg(P.rc_partial + q))); carry29(x);
        const fr29_t y = wv_pow5(x);
        fr_wide29 dot; fr_wide29_zero(dot);
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
            const int j = 1 + lane + 64 * k;
            if (j < T) {
                uint32_t u[9], w[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) { u[i] = sp[9 * j + i]; w[i] = sp[9 * (T - 1 + j) + i]; }
                fr_wide29_mac_regs(dot, u, sj[k]);                           // u_j s_j with the lane's OLD s_j (u is not scaled: it meets the lanes, not an S-box output)
                fr29_t wu; _Pragma("unroll") for (int i = 0; i < 9; ++i) wu.l[i] = w[i];
                sj[k] = add29(sj[k], fr29_mul_mont<PF, true>(wu, y)); carry29(sj[k]);
                if ((q & 31) == 31) lazy_reduce29<PF>(sj[k]);                // the lanes gain about r per round
            }
        }
        if (lane == 0) { uint32_t a[9]; _Pragma("unroll") for (int i = 0; i < 9; ++i) a[i] = sp[i]; fr_wide29_mac_regs(dot, a, y); }     // a_q y
        fr29_t d; fr_wide29_mont<PF, true>(dot, d.l);
        d = add29(d, shfl_xor29(d, 1)); d = add29(d, shfl_xor29(d, 2)); carry29(d);
        d = add29(d, shfl_xor29(d, 4)); d = add29(d, shfl_xor29(d, 8)); carry29(d);
        d = add29(d, shfl_xor29(d, 16)); d = add29(d, shfl_xor29(d, 32)); carry29(d);
        lazy_reduce29<PF>(d);                                               // 64 partials below 2 r each: below 2^261
        s0 = d;
    }
#pragma unroll
    for (int k = 0; k < EPL; ++k) { const int j = 1 + lane + 64 * k; if (j < T) { lazy_reduce29<PF>(sj[k]); wv_st(L.st + 9 * j, sj[k]); } }
    if (lane == 0) wv_st(L.st, s0);
    wv_sync();
    for (int r = half; r < P.rf; ++r) wave_full_round<T>(L, P, r, P.mds29, lane);
}

// K4 (wave form): one Merkle node per wave — hash_with_ds_dynamic([arity, level, position, label], children), eager sponge of rate T - 1.
template <int T>
__global__ void __launch_bounds__(64) k_hash_ds_wave(PoseidonDev P, DsJob J, const fr_t* __restrict__ in0, const fr_t* __restrict__ in1, fr_t* __restrict__ out) {
    extern __shared__ uint4 lds[];
    WaveLds L; L.st = reinterpret_cast<uint32_t*>(lds); L.x = L.st + T * 9; L.part = L.x + T * 9;
    const int lane = threadIdx.x, rate = T - 1; const size_t k = blockIdx.x;
    const size_t cnt = J.mode == 1 ? 2 : ((k + 1) * J.arity <= J.n_in ? J.arity : J.n_in - k * J.arity);
    const size_t total = 4 + cnt + 1;                                                // ds || children || 1, zero padded
    for (int e = lane; e < T; e += 64) { fr29_t z; _Pragma("unroll") for (int i = 0; i < 9; ++i) z.l[i] = 0; wv_st(L.st + 9 * e, z); }
    wv_sync();
    for (size_t base = 0; base < total; base += rate) {                              // eager sponge: permute after every full (or final) block
        for (int cur = lane; cur < rate; cur += 64) {
            const size_t q = base + cur;
            if (q < total) {
                fr_t x;
                if (q == 0) x = J.arity_f; else if (q == 1) x = J.level_f; else if (q == 2) x = fr_from_u64<PF>(ds_position(J, k)); else if (q == 3) x = J.label_f;
                else if (q == total - 1) x = fr_one<PF>();
                else { const size_t c = q - 4; x = J.mode == 1 ? ds_pair_child(J, in0, in1, k, c) : ldg(in0 + k * J.arity + c); }
                fr29_t s = add29(wv_ld(L.st + 9 * cur), fr29_unpack(x)); carry29(s);
                wv_st(L.st + 9 * cur, s);
            }
        }
        wv_sync();
        wave_permute<T>(L, P, lane);
    }
    if (lane == 0) {
        fr29_t v = wv_ld(L.st); carry29(v); lazy_reduce29<PF>(v);
        uint32_t tt[9];
#pragma unroll
        for (int wd = 0; wd < 8; ++wd) {
            const int lo = 32 * wd, i = lo / 29, sh = lo - 29 * i;
            uint32_t xw = v.l[i] >> sh;
            if (i + 1 < 9) xw |= v.l[i + 1] << (29 - sh);
            if (29 - sh + 29 < 32 && i + 2 < 9) xw |= v.l[i + 2] << (58 - sh);
            tt[wd] = xw;
        }
        tt[8] = 0;
        fr_cond_sub<PF>(tt, 0u); fr_cond_sub<PF>(tt, 0u);
        fr_t s; _Pragma("unroll") for (int i = 0; i < 8; ++i) s.v[i] = tt[i];
        stg(out + k, s);
    }
}

}  // namespace stark
#endif
