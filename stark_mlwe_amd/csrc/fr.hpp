// stark_mlwe_amd/csrc/fr.hpp — prime-field arithmetic for the MI355X hot path (product code).
//
// Replaces, on device, the arkworks `Fp<MontBackend<_,4>,4>` operations the reference's inner loops
// call (crates/field/src/lib.rs:13 `F = ark_pallas::Fr`; crates/fft/src/lib.rs:1 `ark_bls12_381::Fr`).
// Memory layout is the reference's: 4 little-endian u64 limbs, Montgomery form with R = 2^256, so a
// Rust `&[F]` is a `const uint64_t*` on the C-ABI with no conversion.  In registers an element is
// 8 x u32 (CDNA4 VALU is 32-bit; the 32x32+64 MAC is v_mad_u64_u32).
//
// The same inline functions compile for the host: the product's host logic (constant derivation,
// Fiat-Shamir scalars) uses them too, so the host side and the kernels share one definition.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FR_HD __host__ __device__ __forceinline__
#else
#define FR_HD inline
#endif

namespace stark {

struct PallasFr {
    // r = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001
    static FR_HD constexpr uint32_t P(int i) {
        constexpr uint32_t a[8] = {0x00000001u, 0x8c46eb21u, 0x0994a8ddu, 0x224698fcu, 0u, 0u, 0u, 0x40000000u};
        return a[i];
    }
    static FR_HD constexpr uint32_t R(int i) {   // 2^256 mod r  (Montgomery 1)
        constexpr uint32_t a[8] = {0xfffffffdu, 0x5b2b3e9cu, 0xe3420567u, 0x992c350bu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu};
        return a[i];
    }
    static FR_HD constexpr uint32_t R2(int i) {  // 2^512 mod r
        constexpr uint32_t a[8] = {0x0000000fu, 0xfc9678ffu, 0x891a16e3u, 0x67bb433du, 0x04ccf590u, 0x7fae2310u, 0x7ccfdaa9u, 0x096d41afu};
        return a[i];
    }
    static FR_HD constexpr uint32_t ROOT32(int i) {  // 5^((r-1)/2^32) in Montgomery form
        constexpr uint32_t a[8] = {0x8c9942deu, 0x21807742u, 0x21b60494u, 0xcc495789u, 0xb2efbee2u, 0xac2e5d27u, 0x7f2db056u, 0x0b79fa89u};
        return a[i];
    }
    static constexpr uint32_t NINV = 0xffffffffu;   // -r^-1 mod 2^32
    static constexpr uint32_t GENERATOR = 5;
    static constexpr int ID = 0;
};
struct Bls12381Fr {
    // r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    static FR_HD constexpr uint32_t P(int i) {
        constexpr uint32_t a[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
        return a[i];
    }
    static FR_HD constexpr uint32_t R(int i) {
        constexpr uint32_t a[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau, 0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
        return a[i];
    }
    static FR_HD constexpr uint32_t R2(int i) {
        constexpr uint32_t a[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu, 0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
        return a[i];
    }
    static FR_HD constexpr uint32_t ROOT32(int i) {  // 7^((r-1)/2^32) in Montgomery form
        constexpr uint32_t a[8] = {0x5f0e466au, 0xb9b58d8cu, 0x1819d7ecu, 0x5b1b4c80u, 0x52a31e64u, 0x0af53ae3u, 0x19e9b27bu, 0x5bf3addau};
        return a[i];
    }
    static constexpr uint32_t NINV = 0xffffffffu;
    static constexpr uint32_t GENERATOR = 7;
    static constexpr int ID = 1;
};

struct alignas(16) fr_t { uint32_t v[8]; };

template <class F> FR_HD fr_t fr_zero() { fr_t z; for (int i = 0; i < 8; ++i) z.v[i] = 0; return z; }
template <class F> FR_HD fr_t fr_one() { fr_t z; for (int i = 0; i < 8; ++i) z.v[i] = F::R(i); return z; }
FR_HD bool fr_eq(const fr_t& a, const fr_t& b) { uint32_t d = 0; for (int i = 0; i < 8; ++i) d |= a.v[i] ^ b.v[i]; return d == 0; }
FR_HD bool fr_is_zero(const fr_t& a) { uint32_t d = 0; for (int i = 0; i < 8; ++i) d |= a.v[i]; return d == 0; }

// t >= p ?  (t given as 8 limbs)
template <class F> FR_HD bool fr_geq_p(const uint32_t* t) {
    // borrow of t - p
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { uint64_t d = (uint64_t)t[i] - F::P(i) - br; br = (d >> 32) & 1; }
    return br == 0;
}
// conditional final subtraction: r = (t >= p || carry) ? t - p : t
// Eight-word add / subtract with carry.  Device: explicit carry chains (__builtin_addc / __builtin_subc -> v_add_co_u32 / v_addc_co_u32, one
// instruction per word).  Written with 64-bit intermediates the compiler zero-extends every word into a register pair and adds pairs
// (v_mov + v_ashrrev + 2 x v_lshl_add_u64 per word: a quarter of the hot loops' non-MAC instructions in the Poseidon kernels were that).
#if defined(__HIP_DEVICE_COMPILE__)
template <class F> FR_HD void fr_cond_sub(uint32_t* t, uint32_t carry) {
    uint32_t d[8]; unsigned c = 1;                       // t + ~P + 1: the carry out is 1 exactly when t >= P
#pragma unroll
    for (int i = 0; i < 8; ++i) { unsigned co; d[i] = __builtin_addc(t[i], ~F::P(i), c, &co); c = co; }
    const bool take = carry || c;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = take ? d[i] : t[i];
}
template <class F> FR_HD fr_t fr_add(const fr_t& a, const fr_t& b) {
    fr_t z; unsigned c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { unsigned co; z.v[i] = __builtin_addc(a.v[i], b.v[i], c, &co); c = co; }
    fr_cond_sub<F>(z.v, c);
    return z;
}
template <class F> FR_HD fr_t fr_sub(const fr_t& a, const fr_t& b) {
    fr_t z; unsigned br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { unsigned bo; z.v[i] = __builtin_subc(a.v[i], b.v[i], br, &bo); br = bo; }
    const uint32_t mask = br ? 0xffffffffu : 0u; unsigned c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { unsigned co; z.v[i] = __builtin_addc(z.v[i], F::P(i) & mask, c, &co); c = co; }
    return z;
}
#else
template <class F> FR_HD void fr_cond_sub(uint32_t* t, uint32_t carry) {
    uint32_t d[8]; uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { uint64_t x = (uint64_t)t[i] - F::P(i) - br; d[i] = (uint32_t)x; br = (x >> 32) & 1; }
    bool take = carry || (br == 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = take ? d[i] : t[i];
}
template <class F> FR_HD fr_t fr_add(const fr_t& a, const fr_t& b) {
    fr_t z; uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { uint64_t s = (uint64_t)a.v[i] + b.v[i] + c; z.v[i] = (uint32_t)s; c = s >> 32; }
    fr_cond_sub<F>(z.v, (uint32_t)c);
    return z;
}
template <class F> FR_HD fr_t fr_sub(const fr_t& a, const fr_t& b) {
    fr_t z; uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { uint64_t d = (uint64_t)a.v[i] - b.v[i] - br; z.v[i] = (uint32_t)d; br = (d >> 32) & 1; }
    uint32_t mask = br ? 0xffffffffu : 0u; uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { uint64_t s = (uint64_t)z.v[i] + (F::P(i) & mask) + c; z.v[i] = (uint32_t)s; c = s >> 32; }
    return z;
}
#endif
template <class F> FR_HD fr_t fr_neg(const fr_t& a) { return fr_sub<F>(fr_zero<F>(), a); }

// Montgomery product, word-serial CIOS over 32-bit limbs (portable C++: host build and reference for
// the device path).  NINV == 0xffffffff for both fields (r == 1 mod 2^32), so m = -t0 needs no
// multiply; the m*P(j) terms with P(j) in {0, 1, 2^30} fold away at compile time.
template <class F> FR_HD fr_t fr_mul_portable(const fr_t& a, const fr_t& b) {
    uint32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t c = 0;
        const uint32_t bi = b.v[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) { uint64_t x = (uint64_t)a.v[j] * bi + t[j] + c; t[j] = (uint32_t)x; c = x >> 32; }
        uint64_t x = (uint64_t)t[8] + c; t[8] = (uint32_t)x; uint32_t t9 = (uint32_t)(x >> 32);
        const uint32_t m = t[0] * F::NINV;
        c = ((uint64_t)m * F::P(0) + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < 8; ++j) { uint64_t y = (uint64_t)m * F::P(j) + t[j] + c; t[j - 1] = (uint32_t)y; c = y >> 32; }
        x = (uint64_t)t[8] + c; t[7] = (uint32_t)x; t[8] = t9 + (uint32_t)(x >> 32);
    }
    fr_cond_sub<F>(t, t[8]);
    fr_t z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z.v[i] = t[i];
    return z;
}

// ---- sums of products with ONE Montgomery reduction ("wide" accumulator) ----------------------------------
// 15 independent 96-bit column accumulators {hi : ml}; W += a*b adds the 64 partial products of one
// term, fr_wide_reduce folds in the Montgomery quotient terms and returns the fully reduced element.
// Inputs must be reduced (< r); at most 27 terms per accumulator (fr_reduce_wide_tail).  Used for the MDS / sparse-matrix
// dot products of the Poseidon rounds: per extra term only the 64 MACs are paid, not the reduction.
struct fr_wide { uint64_t ml[15]; uint32_t hi[15]; };
FR_HD void fr_wide_zero(fr_wide& w) {
#pragma unroll
    for (int c = 0; c < 15; ++c) { w.ml[c] = 0; w.hi[c] = 0; }
}
// t[0..8] < 2^(TAIL) r  ->  t[0..7] in [0, r): conditional subtraction of 2^(TAIL-1) r, ..., 2r, r.
// A sum of L products of reduced operands leaves the Montgomery step below r (L r/2^256 + 1):
// Pallas (r/2^256 ~ 1/4) stays under 8r up to L = 27, BLS12-381 (~0.45) under 16r up to L = 33.
template <class F> FR_HD void fr_reduce_wide_tail(uint32_t* t) {
    constexpr int TAIL = (F::P(7) >> 30) == 1 ? 3 : 4;
#pragma unroll
    for (int s = TAIL - 1; s >= 0; --s) {
        uint32_t d[9]; uint64_t br = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const uint32_t lo = i < 8 ? F::P(i) : 0u, below = i > 0 ? F::P(i - 1) : 0u;
            const uint32_t ps = s == 0 ? lo : ((lo << s) | (below >> (32 - s)));      // limb i of (r << s)
            uint64_t x = (uint64_t)t[i] - ps - br; d[i] = (uint32_t)x; br = (x >> 32) & 1;
        }
        const bool take = br == 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) t[i] = take ? d[i] : t[i];
    }
}
template <class F> FR_HD void fr_wide_mac_portable(fr_wide& w, const fr_t& a, const fr_t& b) {
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) { uint64_t pr = (uint64_t)a.v[i] * b.v[j]; uint64_t x = w.ml[i + j] + pr; w.hi[i + j] += x < pr ? 1u : 0u; w.ml[i + j] = x; }
}
template <class F> FR_HD fr_t fr_wide_reduce_portable(const fr_wide& w) {
    uint64_t ml = 0; uint32_t hi = 0; uint32_t m[8]; uint32_t t[9];
    for (int c = 0; c < 15; ++c) {
        { uint64_t x = ml + w.ml[c]; hi += w.hi[c] + (x < ml ? 1u : 0u); ml = x; }
        for (int k = (c > 7 ? c - 7 : 0); k <= (c - 1 < 7 ? c - 1 : 7); ++k) {
            const uint32_t pj = F::P(c - k);
            if (pj) { uint64_t pr = (uint64_t)m[k] * pj; uint64_t x = ml + pr; hi += x < pr ? 1u : 0u; ml = x; }
        }
        const uint32_t lo = (uint32_t)ml;
        if (c < 8) { m[c] = 0u - lo; ml = ((ml >> 32) | ((uint64_t)hi << 32)) + (lo != 0u ? 1u : 0u); hi = 0; }
        else { t[c - 8] = lo; ml = (ml >> 32) | ((uint64_t)hi << 32); hi = 0; }
    }
    t[7] = (uint32_t)ml; t[8] = (uint32_t)(ml >> 32);
    fr_reduce_wide_tail<F>(t);
    fr_t z; for (int i = 0; i < 8; ++i) z.v[i] = t[i]; return z;
}

#if defined(__HIP_DEVICE_COMPILE__)
// gfx950 device path (generated: tools/gen_fr_gfx950.py).  Every 32x32 partial product costs exactly
//     v_mad_u64_u32  ml, vcc, x, y, ml      (32x32+64 MAC, ~4 cycles per wave64 on CDNA4)
//     v_addc_co_u32  hi, vcc, 0, hi, vcc    (carry count)
// with no zero-extension moves (the portable form costs ~7 VALU instructions per product after
// instruction selection).  The Montgomery quotient digits m_c = -lo(column c) are free because
// r == 1 (mod 2^32); limbs of r that are zero contribute nothing (Pallas: 3 of 8).
template <class F> __device__ __forceinline__ fr_t fr_mul_dev(const fr_t& a, const fr_t& b);
template <class F> __device__ __forceinline__ fr_t fr_wide_reduce_dev(const fr_wide& w);
#include "fr_gfx950.inc"
#endif
template <class F> FR_HD fr_t fr_mul(const fr_t& a, const fr_t& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fr_mul_dev<F>(a, b);
#else
    return fr_mul_portable<F>(a, b);
#endif
}
template <class F> FR_HD fr_t fr_wide_reduce(const fr_wide& w) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fr_wide_reduce_dev<F>(w);
#else
    return fr_wide_reduce_portable<F>(w);
#endif
}
template <class F> FR_HD void fr_wide_mac_f(fr_wide& w, const fr_t& a, const fr_t& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    fr_wide_mac(w, a, b);
#else
    fr_wide_mac_portable<F>(w, a, b);
#endif
}
template <class F> FR_HD fr_t fr_sqr(const fr_t& a) { return fr_mul<F>(a, a); }
template <class F> FR_HD fr_t fr_pow5(const fr_t& x) {   // S-box x^5 = x * (x^2)^2  (poseidon/src/lib.rs:24-29)
    fr_t x2 = fr_sqr<F>(x); fr_t x4 = fr_sqr<F>(x2); return fr_mul<F>(x, x4);
}
template <class F> FR_HD fr_t fr_from_u64(uint64_t x) {   // F::from(u64): canonical -> Montgomery
    fr_t t; t.v[0] = (uint32_t)x; t.v[1] = (uint32_t)(x >> 32); for (int i = 2; i < 8; ++i) t.v[i] = 0;
    fr_t r2; for (int i = 0; i < 8; ++i) r2.v[i] = F::R2(i);
    return fr_mul<F>(t, r2);
}
template <class F> FR_HD fr_t fr_to_canonical(const fr_t& a) {  // into_bigint(): Montgomery -> canonical integer limbs
    fr_t o; o.v[0] = 1; for (int i = 1; i < 8; ++i) o.v[i] = 0;
    return fr_mul<F>(a, o);
}
template <class F> FR_HD fr_t fr_from_canonical(const fr_t& a) {
    fr_t r2; for (int i = 0; i < 8; ++i) r2.v[i] = F::R2(i);
    return fr_mul<F>(a, r2);
}
// Field::pow with a 64-bit exponent (the reference only ever uses `pow(&[n,0,0,0])`).
template <class F> FR_HD fr_t fr_pow_u64(const fr_t& a, uint64_t e) {
    fr_t acc = fr_one<F>(); bool started = false;
    for (int b = 63; b >= 0; --b) {
        if (started) acc = fr_sqr<F>(acc);
        if ((e >> b) & 1) { acc = fr_mul<F>(acc, a); started = true; }
    }
    return acc;
}
// Fermat inverse a^(r-2); zero maps to zero.
template <class F> FR_HD fr_t fr_inv(const fr_t& a) {
    uint32_t e[8];
    for (int i = 0; i < 8; ++i) e[i] = F::P(i);
    // P(0) == 1 for both fields: r - 2 borrows, so limb 0 becomes 0xffffffff and limb 1 loses 1.
    e[0] = 0xffffffffu; { int i = 1; while (e[i] == 0) { e[i] = 0xffffffffu; ++i; } e[i] -= 1; }
    fr_t acc = fr_one<F>(); bool started = false;
    for (int i = 7; i >= 0; --i)
        for (int b = 31; b >= 0; --b) {
            if (started) acc = fr_sqr<F>(acc);
            if ((e[i] >> b) & 1) { acc = fr_mul<F>(acc, a); started = true; }
        }
    return acc;
}
// get_root_of_unity(2^log_n) = ROOT32^(2^(32-log_n))   (field/src/lib.rs:46, fri.rs:54-55)
template <class F> FR_HD fr_t fr_root_of_unity(unsigned log_n) {
    fr_t w; for (int i = 0; i < 8; ++i) w.v[i] = F::ROOT32(i);
    for (unsigned i = log_n; i < 32; ++i) w = fr_sqr<F>(w);
    return w;
}

}  // namespace stark
