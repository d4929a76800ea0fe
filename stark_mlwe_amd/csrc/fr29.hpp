// stark_mlwe_amd/csrc/fr29.hpp — carry-free sums of products in radix 2^29 (gfx950: integer-VALU bound kernels).
//
// Why: in radix 2^32 every 32x32 partial product of a dot product costs TWO half-rate instructions
// (v_mad_u64_u32 + v_addc_co_u32: the 64-bit column accumulator can overflow on every add; measured
// 10.4 nominal cycles per pair at 2 waves/SIMD, profiles/r01_mac_carry_patterns.jsonl).  With 29-bit limbs
// a partial product is < 2^58, so a 64-bit column holds 63 of them: 9 x 9 = 81 products per term but
// NO carry instruction (5.85 cycles each) — 474 instead of 666 cycles per term, and the column walk of the
// reduction has no carry words either.
//
// Domain trick: data stays in the ark-ff representation x*R mod r, R = 2^256 (nothing changes at the ABI or
// in LDS).  The CONSTANT operand of every dot product (MDS / sparse-matrix entries) is stored as
// c*R' mod r with R' = 2^261 = 2^(9*29), split into nine 29-bit limbs; the Montgomery reduction below
// divides by R', so  sum_i (c_i R')(x_i R) / R' = (sum_i c_i x_i) R  — the result is again in the R domain.
// r = 1 (mod 2^29) in both fields, so the quotient digit is m = -t mod 2^29 with no multiply.
//
// Bounds (operands reduced, < r < 2^255, so limb 8 is below 2^23): a column gains < 8 * 2^58 per term (column 7:
// eight full products; column 8 has nine but two of them involve a top limb), the reduction adds < 6 * 2^58 per
// column (five non-zero limbs of r besides limb 0, plus the carry): at most 7 terms between carry passes, and
// at most 7 terms since the last pass when the reduction starts (7*8 + 6 = 62 < 64).  For a sum of K <= 64 terms the Montgomery step leaves < (K/128 + 1) r < 2r: ONE conditional
// subtraction.  Portable C++ (host build for the CPU checks; on the device the products compile to
// v_mad_u64_u32 with the 64-bit add folded in).
#pragma once
#include "fr.hpp"

namespace stark {

constexpr uint32_t FR_M29 = (1u << 29) - 1;
struct fr29_t { uint32_t l[9]; };             // value = sum l[i] * 2^(29 i), l[i] < 2^29
struct fr_wide29 { uint64_t c[18]; };         // column sums, weight 2^(29 k)

template <class F> constexpr uint32_t fr_p29(int i) {      // limb i of the modulus in radix 2^29
    const int lo = 29 * i, w = lo >> 5, s = lo & 31;
    const uint64_t x = (w < 8 ? (uint64_t)F::P(w) : 0ull) | (w + 1 < 8 ? ((uint64_t)F::P(w + 1) << 32) : 0ull);
    return (uint32_t)(x >> s) & FR_M29;
}

// Terms a column can take between carry passes (and before the reduction): 8 * 2^58 per term, (nz + 1) * 2^58 from the reduction.
template <class F> constexpr int fr29_max_terms() {
    int nz = 0;
    for (int j = 1; j < 9; ++j) nz += fr_p29<F>(j) != 0 ? 1 : 0;
    return (64 - (nz + 1)) / 8;
}

FR_HD fr29_t fr29_unpack(const fr_t& x) {
    fr29_t r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int lo = 29 * i, w = lo >> 5, s = lo & 31;
        uint32_t v = x.v[w] >> s;
        if (s > 3 && w + 1 < 8) v |= x.v[w + 1] << (32 - s);
        r.l[i] = v & FR_M29;
    }
    return r;
}
FR_HD void fr_wide29_zero(fr_wide29& w) {
#pragma unroll
    for (int k = 0; k < 18; ++k) w.c[k] = 0;
}
// W += a * b, a = nine 29-bit limbs (a constant in the R' domain), b unpacked data.
// Device: the constant tables are never written by a kernel and their address is wave-uniform, but after the
// first workgroup barrier the compiler can no longer prove "not clobbered" and falls back to per-lane vector
// loads (a VMEM round trip in front of every term).  Reading them through the constant address space keeps
// them on the scalar data path (s_load -> SGPR operand of the MAC).
FR_HD void fr_wide29_mac(fr_wide29& w, const uint32_t* __restrict__ a_, const fr29_t& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) uint32_t* cptr_t;
    cptr_t a = (cptr_t)a_;
#else
    const uint32_t* a = a_;
#endif
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const uint32_t ai = a[i];
#pragma unroll
        for (int j = 0; j < 9; ++j) w.c[i + j] += (uint64_t)ai * b.l[j];
    }
}
// the same with the constant's limbs already in registers (e.g. read from LDS)
FR_HD void fr_wide29_mac_regs(fr_wide29& w, const uint32_t* a, const fr29_t& b) {
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j) w.c[i + j] += (uint64_t)a[i] * b.l[j];
}
// carry pass: columns 0..16 back below 2^29, the excess moves up (column 17 absorbs the top)
FR_HD void fr_wide29_norm(fr_wide29& w) {
#pragma unroll
    for (int k = 0; k < 17; ++k) { w.c[k + 1] += w.c[k] >> 29; w.c[k] &= FR_M29; }
}
// Montgomery step: divides the column sums by R' = 2^261; l[0..8] = limbs of the quotient (l[8] keeps every bit
// above 2^232).  For a sum of K products of operands below 2r the quotient is below (K/32 + 1) r.
// MAC_POW2: a power-of-two limb of the modulus (Pallas: limb 8 = 2^22) is normally strength-reduced to a 64-bit shift plus a 64-bit
// add (two full-rate instructions per digit) — right for the throughput kernels, whose multiplier pipe is the saturated resource
// (measured: leaf kernel 3 % SLOWER with the MAC form in round 2, 1.3 % slower again in round 3 with the multiplier in an SGPR); the NTT butterflies,
// whose instruction mix has fewer MACs per instruction, take the MAC form too (LDE 2^20 -> 2^23 -2.3 %, 2^23 coset NTT -1.5 %).  On a lone wave every instruction costs one ~5-cycle issue slot, so the
// latency kernels (poseidon_coop.hpp) hold the limb in a register as an opaque multiplier: ONE v_mad_u64_u32 per digit instead of two
// instructions (sponge 145.6 -> 141.9 us per permutation).
template <class F, bool MAC_POW2 = false> FR_HD void fr_wide29_mont(fr_wide29& w, uint32_t* l) {
    uint32_t pj[9];
#pragma unroll
    for (int j = 1; j < 9; ++j) pj[j] = fr_p29<F>(j);
#if defined(__HIP_DEVICE_COMPILE__)
    if (MAC_POW2) {
#pragma unroll
        for (int j = 1; j < 9; ++j) if (fr_p29<F>(j) != 0 && (fr_p29<F>(j) & (fr_p29<F>(j) - 1)) == 0) asm("" : "+v"(pj[j]));
    }
#endif
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const uint64_t t = w.c[k];
        const uint32_t m = (0u - (uint32_t)t) & FR_M29;            // t + m * r == 0 (mod 2^29) because r == 1 (mod 2^29)
        w.c[k + 1] += (t + FR_M29) >> 29;                          // == (t + m) >> 29, the multiple of 2^29 above t; adding the constant needs no zero-extended m
#pragma unroll
        for (int j = 1; j < 9; ++j)
            if (fr_p29<F>(j) != 0) w.c[k + j] += (uint64_t)m * pj[j];
    }
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) { const uint64_t v = w.c[9 + i] + carry; l[i] = i < 8 ? ((uint32_t)v & FR_M29) : (uint32_t)v; carry = v >> 29; }
}
// nine 29-bit limbs (value < 2r) -> eight 32-bit limbs, fully reduced
template <class F> FR_HD fr_t fr29_pack_reduce(const uint32_t* l) {
    uint32_t t[9];
#pragma unroll
    for (int wd = 0; wd < 8; ++wd) {
        const int lo = 32 * wd, i = lo / 29, s = lo - 29 * i;        // word wd starts at bit s of limb i
        uint32_t v = l[i] >> s;
        if (i + 1 < 9) v |= l[i + 1] << (29 - s);
        if (29 - s + 29 < 32 && i + 2 < 9) v |= l[i + 2] << (58 - s);
        t[wd] = v;
    }
    t[8] = 0;                                                       // the value is below 2r < 2^256
    fr_cond_sub<F>(t, t[8]);
    fr_t z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z.v[i] = t[i];
    return z;
}
// The same WITHOUT the final conditional subtract: the eight words of the quotient (below 1.1 r for the short sums it is used on), not canonical.
template <class F> FR_HD fr_t fr_wide29_reduce_lazy(fr_wide29& w) {
    uint32_t l[9]; fr_wide29_mont<F>(w, l);
    fr_t z;
#pragma unroll
    for (int wd = 0; wd < 8; ++wd) {
        const int lo = 32 * wd, i = lo / 29, s = lo - 29 * i;
        uint32_t v = l[i] >> s;
        if (i + 1 < 9) v |= l[i + 1] << (29 - s);
        if (29 - s + 29 < 32 && i + 2 < 9) v |= l[i + 2] << (58 - s);
        z.v[wd] = v;
    }
    return z;
}
// Montgomery reduction by R' = 2^261 and return to eight 32-bit limbs, fully reduced.
template <class F, bool MAC_POW2 = false> FR_HD fr_t fr_wide29_reduce(fr_wide29& w) {
    uint32_t l[9]; fr_wide29_mont<F, MAC_POW2>(w, l);
    return fr29_pack_reduce<F>(l);
}

// ---- the S-box in radix 2^29 ---------------------------------------------------------------------------------
// x^5 = x * (x^2)^2 with two SQUARINGS: in radix 2^29 a doubled limb still fits the 32-bit multiplier operand, so a
// square is 36 cross products (against 2*l_j) + 9 squares = 45 MACs instead of 81 (in radix 2^32 doubling a limb
// overflows the operand, and doubling column sums costs what it saves).  Values stay as nine limbs, below 1.01 r,
// between the three steps; each step divides by R' = 2^261 while the operands carry R = 2^256, so the result is
//     fr_pow5_r29(x R) = x^5 R / 2^20      (three times a factor 2^5, compounded: 2^-5, 2^-15, 2^-20)
// and every constant that multiplies an S-box output is stored pre-multiplied by 2^20 (host_util.hpp to_radix29).
template <class F, bool MAC_POW2 = false> FR_HD fr29_t fr29_sqr_mont(const fr29_t& a) {
    fr_wide29 w; fr_wide29_zero(w);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        w.c[2 * i] += (uint64_t)a.l[i] * a.l[i];
#pragma unroll
        for (int j = i + 1; j < 9; ++j) w.c[i + j] += (uint64_t)a.l[i] * (a.l[j] << 1);
    }
    fr29_t r; fr_wide29_mont<F, MAC_POW2>(w, r.l); return r;
}
template <class F, bool MAC_POW2 = false> FR_HD fr29_t fr29_mul_mont(const fr29_t& a, const fr29_t& b) {
    fr_wide29 w; fr_wide29_zero(w);
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j) w.c[i + j] += (uint64_t)a.l[i] * b.l[j];
    fr29_t r; fr_wide29_mont<F, MAC_POW2>(w, r.l); return r;
}
constexpr int FR29_SBOX_SHIFT = 20;
template <class F> FR_HD fr_t fr_pow5_r29(const fr_t& x) {
    const fr29_t u = fr29_unpack(x);
    const fr29_t x2 = fr29_sqr_mont<F>(u), x4 = fr29_sqr_mont<F>(x2), x5 = fr29_mul_mont<F>(u, x4);
    return fr29_pack_reduce<F>(x5.l);
}

// ---- lazily reduced nine-limb values (NTT butterflies, ntt_dev.hpp) ----------------------------------------------
// Between two Montgomery steps a value may be any non-negative integer below 2^261 whose limbs 0..7 stay below
// 7 * 2^29 (limb 8 absorbs what is above 2^232).  fr29_mul_mont(tw, b) accepts b with limbs up to 6 * 2^29
// (9 * 2^29 * 6 * 2^29 + the reduction's 8 * 2^58 stays below 2^64 per column) and returns limbs below 2^29 and a
// value below (V_b * r / 2^261 + 1) r, i.e. below 1.6 r for V_b <= 41 in both fields.
FR_HD void fr29_norm(fr29_t& a) {               // carry pass: limbs 0..7 back below 2^29, the value unchanged
#pragma unroll
    for (int i = 0; i < 8; ++i) { a.l[i + 1] += a.l[i] >> 29; a.l[i] &= FR_M29; }
}
// Cheap reduction of a lazy value WITHOUT a product: carry pass, q = floor(top / (r_8 + 1)) by a magic multiply (exact for
// top < 2^29), x <- x - q r with signed carries.  q never exceeds x / r, and what is left is below (r_8 + q + 1) 2^232, i.e.
// below 1.00002 r: limbs below 2^29, the same residue.  (~45 simple instructions against 81 + 36 MACs and a column walk.)
template <class F> FR_HD void fr29_partial_reduce(fr29_t& x) {
    constexpr uint64_t d = (uint64_t)fr_p29<F>(8) + 1, magic = ((1ull << 52) + d - 1) / d;     // ceil(2^52 / d) < 2^31
    fr29_norm(x);
    const uint32_t q = (uint32_t)(((uint64_t)x.l[8] * magic) >> 52);
    int64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int64_t t = (int64_t)x.l[i] + carry - (int64_t)((uint64_t)q * fr_p29<F>(i));
        if (i < 8) { x.l[i] = (uint32_t)t & FR_M29; carry = t >> 29; } else x.l[8] = (uint32_t)t;
    }
}
// One radix-2 decimation-in-time butterfly  (a, b) <- (a + tw*b, a - tw*b)  on lazily reduced values.
// D = 4r written with every limb below the top lifted by 2^29 (and the matching borrow taken from the limb above),
// so that  a + D - p  never borrows limb-wise: p has limbs below 2^29 and a top limb below 1.6 * r_8 < D_8.
// Growth per stage: limbs of a by at most 2^29 (sum) / 2^30 (difference), the value by at most 4r.
template <class F> FR_HD void ntt29_butterfly(fr29_t& a, fr29_t& b, const fr29_t& tw, const uint32_t* __restrict__ D, bool norm) {
    if (norm) { fr29_norm(a); fr29_norm(b); }
    const fr29_t p = fr29_mul_mont<F, true>(tw, b);
#pragma unroll
    for (int k = 0; k < 9; ++k) { const uint32_t av = a.l[k]; a.l[k] = av + p.l[k]; b.l[k] = av + D[k] - p.l[k]; }
}
// The same with the twiddle 1: no product.  FRESH: b is a canonical input (limbs below 2^29, below r) and is taken as it is;
// otherwise b is first brought back below 1.00002 r — an unmultiplied operand would double the value at every stage.
template <class F, bool FRESH> FR_HD void ntt29_butterfly_w1(fr29_t& a, fr29_t& b, const uint32_t* __restrict__ D) {
    if (!FRESH) fr29_partial_reduce<F>(b);
#pragma unroll
    for (int k = 0; k < 9; ++k) { const uint32_t av = a.l[k], bv = b.l[k]; a.l[k] = av + bv; b.l[k] = av + D[k] - bv; }
}
// The first HEAD (<= 3) stages of a group of 2^HEAD consecutive rows, in registers: their twiddles have compile-time
// positions, so the trivial ones (all of stage 1, half of stage 2, a quarter of stage 3) cost no product:
// 5 products per 8 points instead of 12.  w1 = w_8, w2 = w_4, w3 = w_8^3 in the tables' domain.
template <class F, int HEAD> FR_HD void ntt29_head(fr29_t (&x)[1 << HEAD], const fr29_t& w1, const fr29_t& w2, const fr29_t& w3, const uint32_t* __restrict__ D) {
#pragma unroll
    for (int g = 0; g < (1 << (HEAD - 1)); ++g) ntt29_butterfly_w1<F, true>(x[2 * g], x[2 * g + 1], D);
    if constexpr (HEAD >= 2) {
#pragma unroll
        for (int g = 0; g < (1 << (HEAD - 2)); ++g) {
            ntt29_butterfly_w1<F, false>(x[4 * g], x[4 * g + 2], D);
            ntt29_butterfly<F>(x[4 * g + 1], x[4 * g + 3], w2, D, false);
        }
    }
    if constexpr (HEAD == 3) {
        ntt29_butterfly_w1<F, false>(x[0], x[4], D);
        ntt29_butterfly<F>(x[1], x[5], w1, D, false);
        ntt29_butterfly<F>(x[2], x[6], w2, D, false);
        ntt29_butterfly<F>(x[3], x[7], w3, D, false);
    }
}
// Stage s (1-based) of a tile starts from limbs below (1 + 2*(stages since the last carry pass)) * 2^29: a carry pass
// on both inputs before stages 4, 7, 10 keeps the multiplied operand at <= 5 * 2^29 and every sum below 2^32.
FR_HD bool ntt29_norm_before(int s) { return s > 1 && (s - 1) % 3 == 0; }
FR_HD bool ntt29_norm_after(int stages) { return stages >= 1 && (stages - 1) % 3 == 2; }   // limbs reach 7 * 2^29 after stages 3, 6, 9
// host: the constants of the lazy butterflies
template <class F> inline void ntt29_offset(uint32_t D[9]) {
    uint64_t cy = 0; uint32_t c[9];
    for (int i = 0; i < 9; ++i) { const uint64_t v = (uint64_t)fr_p29<F>(i) * 4 + cy; c[i] = i < 8 ? (uint32_t)(v & FR_M29) : (uint32_t)v; cy = v >> 29; }
    for (int i = 0; i < 9; ++i) D[i] = c[i] + (i < 8 ? (1u << 29) : 0u) - (i > 0 ? 1u : 0u);
}

// Host side: the nine limbs of c * R' mod r for a constant given in the R domain (c * R mod r).
template <class F> inline void fr29_const_from(const fr_t& c_R, uint32_t out[9]) {
    const fr_t c32 = fr_mul_portable<F>(c_R, fr_from_u64<F>(32));   // (cR)(32R)/R = 32 c R = c R'
    const fr29_t u = fr29_unpack(c32);
    for (int i = 0; i < 9; ++i) out[i] = u.l[i];
}

}  // namespace stark
