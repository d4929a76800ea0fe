// oracle/fr.hpp — TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product library.
//
// 4x u64 Montgomery (R = 2^256) prime-field element, restating what the reference gets from
// arkworks `Fp<MontBackend<FrConfig,4>,4>` (ark-ff 0.5.0, un-vendored dependency of
// /root/reference: Cargo.lock:45-172).  Reference call sites this stands in for:
//   crates/field/src/lib.rs:13      `pub use ark_pallas::Fr as F`      (PallasFr)
//   crates/fft/src/lib.rs:1         `use ark_bls12_381::Fr as F`       (Bls12381Fr)
// In-memory layout equals ark-ff's: 4 little-endian u64 limbs holding value*2^256 mod r.
// Constants: SURVEY.md Appendix A (byte patterns verified against the reference's build artefacts).
#pragma once
#include <cstdint>
#include <cstring>
#include <cstddef>

namespace oracle {

typedef unsigned __int128 u128;

struct PallasFrParams {
    // r = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001
    static constexpr uint64_t MOD[4] = {0x8c46eb2100000001ULL, 0x224698fc0994a8ddULL, 0x0ULL, 0x4000000000000000ULL};
    static constexpr uint64_t INV = 0x8c46eb20ffffffffULL;   // -r^-1 mod 2^64
    static constexpr uint64_t GENERATOR = 5;                  // ark-pallas fr.rs #[generator = "5"]
    static constexpr unsigned TWO_ADICITY = 32;
    static constexpr const char* NAME = "pallas_fr";
};
struct Bls12381FrParams {
    // r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    static constexpr uint64_t MOD[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
    static constexpr uint64_t INV = 0xfffffffeffffffffULL;
    static constexpr uint64_t GENERATOR = 7;                  // ark-bls12-381 fr.rs #[generator = "7"]
    static constexpr unsigned TWO_ADICITY = 32;
    static constexpr const char* NAME = "bls12_381_fr";
};

template <class P>
struct FrT {
    uint64_t l[4];   // Montgomery form, little-endian limbs (== ark-ff BigInt<4>.0)

    // ---- raw 256-bit helpers -------------------------------------------------------------
    static bool geq_mod(const uint64_t a[4]) {
        for (int i = 3; i >= 0; --i) {
            if (a[i] > P::MOD[i]) return true;
            if (a[i] < P::MOD[i]) return false;
        }
        return true;
    }
    static void sub_mod_inplace(uint64_t a[4]) {
        u128 br = 0;
        for (int i = 0; i < 4; ++i) {
            u128 d = (u128)a[i] - P::MOD[i] - br;
            a[i] = (uint64_t)d;
            br = (d >> 64) & 1;
        }
    }

    // ---- constructors -----------------------------------------------------------------------
    static FrT zero() { FrT z; z.l[0] = z.l[1] = z.l[2] = z.l[3] = 0; return z; }
    static FrT from_raw(const uint64_t x[4]) { FrT z; memcpy(z.l, x, 32); return z; }
    // R mod r (Montgomery form of 1), computed once: 2^256 mod r by repeated doubling.
    static FrT one() {
        static const FrT ONE = compute_r();
        return ONE;
    }
    static FrT compute_r() {
        uint64_t a[4] = {1, 0, 0, 0};
        for (int i = 0; i < 256; ++i) dbl_raw(a);
        return from_raw(a);
    }
    static FrT r2() {  // R^2 mod r = 2^512 mod r
        static const FrT R2 = [] {
            uint64_t a[4] = {1, 0, 0, 0};
            for (int i = 0; i < 512; ++i) dbl_raw(a);
            return from_raw(a);
        }();
        return R2;
    }
    static void dbl_raw(uint64_t a[4]) {  // a = 2a mod r, a < r
        uint64_t c = 0;
        for (int i = 0; i < 4; ++i) { uint64_t n = (a[i] << 1) | c; c = a[i] >> 63; a[i] = n; }
        if (c || geq_mod(a)) sub_mod_inplace(a);
    }
    // F::from(u64) (ark-ff `impl From<u64> for Fp`): canonical integer -> Montgomery.
    static FrT from_u64(uint64_t x) {
        FrT t; t.l[0] = x; t.l[1] = t.l[2] = t.l[3] = 0;
        return t * r2();
    }
    // from a canonical 256-bit integer that is already < r
    static FrT from_canonical(const uint64_t x[4]) { return from_raw(x) * r2(); }

    // F::from_le_bytes_mod_order(bytes): int_le(bytes) mod r  (ark-ff PrimeField; call sites
    // utils/src/lib.rs:12, transcript/src/lib.rs:18,25,38).  Horner over 8-byte limbs, top first.
    static FrT from_le_bytes_mod_order(const uint8_t* b, size_t n) {
        FrT acc = zero();
        const FrT base = from_u64(0).add_raw_pow64();  // 2^64 in Montgomery form
        size_t nl = (n + 7) / 8;
        for (size_t k = nl; k-- > 0;) {
            uint64_t w = 0;
            for (size_t j = 0; j < 8; ++j) {
                size_t idx = k * 8 + j;
                if (idx < n) w |= (uint64_t)b[idx] << (8 * j);
            }
            acc = acc * base + from_u64(w);
        }
        return acc;
    }
    FrT add_raw_pow64() const {  // helper: returns Montgomery(2^64)
        uint64_t x[4] = {0, 1, 0, 0};
        return from_canonical(x);
    }

    // ---- arithmetic ------------------------------------------------------------------------
    FrT operator+(const FrT& o) const {
        FrT z; u128 c = 0;
        for (int i = 0; i < 4; ++i) { u128 s = (u128)l[i] + o.l[i] + c; z.l[i] = (uint64_t)s; c = s >> 64; }
        if (c || geq_mod(z.l)) sub_mod_inplace(z.l);
        return z;
    }
    FrT operator-(const FrT& o) const {
        FrT z; u128 br = 0;
        for (int i = 0; i < 4; ++i) { u128 d = (u128)l[i] - o.l[i] - br; z.l[i] = (uint64_t)d; br = (d >> 64) & 1; }
        if (br) { u128 c = 0; for (int i = 0; i < 4; ++i) { u128 s = (u128)z.l[i] + P::MOD[i] + c; z.l[i] = (uint64_t)s; c = s >> 64; } }
        return z;
    }
    FrT neg() const { return zero() - *this; }
    // CIOS Montgomery product.
    FrT operator*(const FrT& o) const {
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) {
            u128 c = 0;
            for (int j = 0; j < 4; ++j) { u128 s = (u128)t[j] + (u128)l[j] * o.l[i] + c; t[j] = (uint64_t)s; c = s >> 64; }
            u128 s = (u128)t[4] + c; t[4] = (uint64_t)s; t[5] = (uint64_t)(s >> 64);
            uint64_t m = t[0] * P::INV;
            c = ((u128)t[0] + (u128)m * P::MOD[0]) >> 64;
            for (int j = 1; j < 4; ++j) { u128 s2 = (u128)t[j] + (u128)m * P::MOD[j] + c; t[j - 1] = (uint64_t)s2; c = s2 >> 64; }
            s = (u128)t[4] + c; t[3] = (uint64_t)s; t[4] = t[5] + (uint64_t)(s >> 64);
        }
        FrT z; memcpy(z.l, t, 32);
        if (t[4] || geq_mod(z.l)) sub_mod_inplace(z.l);
        return z;
    }
    FrT& operator+=(const FrT& o) { *this = *this + o; return *this; }
    FrT& operator-=(const FrT& o) { *this = *this - o; return *this; }
    FrT& operator*=(const FrT& o) { *this = *this * o; return *this; }
    FrT square() const { return *this * *this; }
    bool operator==(const FrT& o) const { return memcmp(l, o.l, 32) == 0; }
    bool operator!=(const FrT& o) const { return !(*this == o); }
    bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }

    // Field::pow(&[u64;4]) — square-and-multiply, MSB first.
    FrT pow(const uint64_t e[4]) const {
        FrT acc = one(); bool started = false;
        for (int i = 3; i >= 0; --i)
            for (int b = 63; b >= 0; --b) {
                if (started) acc = acc.square();
                if ((e[i] >> b) & 1) { acc = acc * *this; started = true; }
            }
        return acc;
    }
    FrT pow_u64(uint64_t e) const { uint64_t x[4] = {e, 0, 0, 0}; return pow(x); }
    // Field::inverse(): Fermat a^(r-2); returns zero for zero (callers `expect` non-zero).
    FrT inverse() const {
        uint64_t e[4]; memcpy(e, P::MOD, 32);
        // r - 2 (r's low limb ends in ...01, so no borrow beyond limb 0)
        e[0] -= 2;
        return pow(e);
    }

    // ---- canonical encodings ---------------------------------------------------------------
    // into_bigint(): canonical integer limbs.
    void to_canonical(uint64_t out[4]) const {
        FrT o; o.l[0] = 1; o.l[1] = o.l[2] = o.l[3] = 0;
        FrT c = *this * o;   // Montgomery reduce: (a * 1) / R
        memcpy(out, c.l, 32);
    }
    // CanonicalSerialize::{serialize_compressed, serialize_uncompressed}: 32-byte LE canonical
    // (field/src/lib.rs:206-215; fri.rs:65,184,491,514).
    void to_bytes_le(uint8_t out[32]) const {
        uint64_t c[4]; to_canonical(c);
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) out[8 * i + j] = (uint8_t)(c[i] >> (8 * j));
    }

    // ---- roots of unity (ark-ff FftField::get_root_of_unity) ------------------------------------
    // TWO_ADIC_ROOT_OF_UNITY = GENERATOR^((r-1)/2^TWO_ADICITY); get_root_of_unity(2^k) squares it
    // (TWO_ADICITY - k) times.  Used by field/src/lib.rs:46, fri.rs:54-55 (group_gen).
    static FrT two_adic_root() {
        static const FrT W = [] {
            uint64_t e[4]; memcpy(e, P::MOD, 32); e[0] -= 1;
            // shift right by TWO_ADICITY (=32)
            unsigned s = P::TWO_ADICITY;
            uint64_t o[4];
            for (int i = 0; i < 4; ++i) {
                uint64_t lo = e[i] >> s;
                uint64_t hi = (i + 1 < 4) ? (e[i + 1] << (64 - s)) : 0;
                o[i] = lo | hi;
            }
            return from_u64(P::GENERATOR).pow(o);
        }();
        return W;
    }
    static FrT root_of_unity_log(unsigned log_n) {
        FrT w = two_adic_root();
        for (unsigned i = log_n; i < P::TWO_ADICITY; ++i) w = w.square();
        return w;
    }
};

typedef FrT<PallasFrParams> Fr;          // the prover field (D1 in SURVEY.md)
typedef FrT<Bls12381FrParams> FrBls;     // the `fft` crate's field

}  // namespace oracle
