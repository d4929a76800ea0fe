// oracle/ntt.hpp — TEST INFRASTRUCTURE ONLY (CPU oracle).
//
// NTT / iNTT semantics of crates/fft/src/lib.rs:6-32 (thin wrappers over ark-poly 0.5.0
// `Radix2EvaluationDomain::{fft_in_place, ifft_in_place}`, an un-vendored dependency):
//   fft : out[i] = sum_j a[j] * w^(i*j),  w = get_root_of_unity(n), natural order in and out
//   ifft: exact inverse, including the n^-1 scaling.
// The reference holds no golden vectors for this path, so the oracle is the
// mathematical DFT: `dft_naive` (O(n^2), the definition) pins `ntt_radix2` (O(n log n)), which in
// turn checks the GPU kernels at large n.  Coset/LDE are defined by this build.
#pragma once
#include <vector>
#include "fr.hpp"

namespace oracle {

template <class F>
static inline std::vector<F> dft_naive(const std::vector<F>& a, bool inverse) {
    size_t n = a.size(); unsigned lg = 0; while (((size_t)1 << lg) < n) ++lg;
    F w = F::root_of_unity_log(lg); if (inverse) w = w.inverse();
    std::vector<F> out(n);
    F wi = F::one();
    for (size_t i = 0; i < n; ++i) {
        F acc = F::zero(), x = F::one();
        for (size_t j = 0; j < n; ++j) { acc += a[j] * x; x *= wi; }
        out[i] = acc; wi *= w;
    }
    if (inverse) { F ninv = F::from_u64((uint64_t)n).inverse(); for (auto& x : out) x *= ninv; }
    return out;
}

template <class F>
static inline void ntt_radix2(F* a, unsigned lg, bool inverse) {
    size_t n = (size_t)1 << lg;
    for (size_t i = 0; i < n; ++i) {   // bit reversal
        size_t r = 0; for (unsigned b = 0; b < lg; ++b) if (i >> b & 1) r |= (size_t)1 << (lg - 1 - b);
        if (r > i) { F t = a[i]; a[i] = a[r]; a[r] = t; }
    }
    F w = F::root_of_unity_log(lg); if (inverse) w = w.inverse();
    std::vector<F> tw(n / 2 ? n / 2 : 1);
    { F x = F::one(); for (size_t i = 0; i < n / 2; ++i) { tw[i] = x; x *= w; } }
    for (unsigned s = 1; s <= lg; ++s) {
        size_t half = (size_t)1 << (s - 1), step = n >> s;
        #pragma omp parallel for schedule(static)
        for (long blk = 0; blk < (long)(n >> s); ++blk) {
            size_t base = (size_t)blk << s;
            for (size_t j = 0; j < half; ++j) {
                F u = a[base + j], v = a[base + j + half] * tw[j * step];
                a[base + j] = u + v; a[base + j + half] = u - v;
            }
        }
    }
    if (inverse) { F ninv = F::from_u64((uint64_t)n).inverse(); for (size_t i = 0; i < n; ++i) a[i] *= ninv; }
}

// Coset forms (definition of this build): coset-NTT evaluates at shift*w^i (scale a[j] by shift^j
// first); coset-iNTT undoes it (iNTT, then scale by shift^-j).
template <class F>
static inline void ntt_coset(F* a, unsigned lg, bool inverse, const F& shift) {
    size_t n = (size_t)1 << lg;
    if (!inverse) { F x = F::one(); for (size_t j = 0; j < n; ++j) { a[j] *= x; x *= shift; } ntt_radix2(a, lg, false); }
    else { ntt_radix2(a, lg, true); F si = shift.inverse(), x = F::one(); for (size_t j = 0; j < n; ++j) { a[j] *= x; x *= si; } }
}
// LDE (definition of this build): evaluations on H (2^lg) -> coefficients -> zero-pad -> coset
// evaluations on shift*H' with |H'| = 2^(lg+lg_blowup).
template <class F>
static inline std::vector<F> lde(const std::vector<F>& evals, unsigned lg, unsigned lg_blowup, const F& shift) {
    std::vector<F> c = evals; ntt_radix2(c.data(), lg, true);
    c.resize((size_t)1 << (lg + lg_blowup), F::zero());
    ntt_coset(c.data(), lg + lg_blowup, false, shift);
    return c;
}

// Scalable synthetic generator (ii) (definition of this build; SURVEY.md §8(d) adapted): limb j of
// element (col, i) is mix64(seed + (col << 56) + 4*i + j), top limb masked to 62 bits (< 2^254 < r
// for both fields); the limbs ARE the stored (Montgomery) representation, as in ark's `Fp::rand`.
static inline uint64_t mix64(uint64_t x) {
    x += 0x9e3779b97f4a7c15ULL; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL; x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL; return x ^ (x >> 31);
}
static inline void synth_element(uint64_t seed, uint64_t col, uint64_t i, uint64_t out[4]) {
    for (uint64_t j = 0; j < 4; ++j) out[j] = mix64(seed + (col << 56) + 4 * i + j);
    out[3] &= 0x3FFFFFFFFFFFFFFFULL;
}

}  // namespace oracle
