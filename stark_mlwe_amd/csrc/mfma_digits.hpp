// stark_mlwe_amd/csrc/mfma_digits.hpp — HOST MIRROR of the scalar pieces of the int8-MFMA field products (poseidon_pair.hpp: recode_signed,
// mfma_fold_rows, the carry pass + Montgomery step of pair_apply_mds_mfma), statement by statement, for the host-check library: the fragment
// tables (host_util.hpp mfma_frags) and the fold arithmetic are exercised on the CPU against the L*U rows (tests/test_hostcheck.py).  The device
// keeps its own copies: routing the kernels through these templates compiled to a 2 % slower leaf kernel (same box, A/B), and the kernels'
// own results are pinned by the GPU parity tests.  Not included by any device translation unit.
#pragma once
#include "fr.hpp"
#include "fr29.hpp"

namespace stark {

// x (canonical) -> signed radix-256 digits in place of its bytes: add 0x80 to every byte with carries, flip every byte's top bit
// (digit b = byte b - 0x80 in [-128, 127]; x < r keeps the top byte below 0x80, so 32 digits hold it).  The int8 operand form of the MFMA product.
FR_HD fr_t recode_signed(const fr_t& x) {
    fr_t y; uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const uint64_t t = (uint64_t)x.v[i] + 0x80808080u + c; y.v[i] = (uint32_t)t ^ 0x80808080u; c = t >> 32; }
    return y;
}
// Digit sums of one 32-row tile pair -> 64-bit columns of weight 2^(29k).  lo[reg] = row (reg & 3) + 8 (reg >> 2) of tile rt (the rows the lower lane of a
// sponge's lane pair receives), hi[reg] = the same + 4 (the upper lane's rows); row = digit position c - 32 rt.  Pairs of adjacent digit sums
// (S0 + 256 S1 < 2^33) go into the column that holds the lower digit (shift < 29: no overflow, the columns are not normalised).
template <class V> FR_HD void mfma_fold_rows(int64_t* col, const V& lo, const V& hi, int rt) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const V& a = hh ? hi : lo;
                const int64_t pair = (int64_t)a[4 * q + 2 * p] + (int64_t)a[4 * q + 2 * p + 1] * 256;
                const int c = 32 * rt + 8 * q + 4 * hh + 2 * p, k = (8 * c) / 29, sh = 8 * c - 29 * k;
                col[k] += pair * ((int64_t)1 << sh);      // pair may be negative: a product, not a shift of a signed value (undefined before C++20)
            }
}
// columns (signed, the total a non-negative integer) -> canonical field element: signed carry pass, Montgomery step by 2^261
FR_HD fr_t mfma_finish_cols(int64_t* col) {
    fr_wide29 w;
#pragma unroll
    for (int k = 0; k < 17; ++k) { col[k + 1] += col[k] >> 29; w.c[k] = (uint64_t)col[k] & FR_M29; }
    w.c[17] = (uint64_t)col[17];
    return fr_wide29_reduce<PallasFr>(w);
}

}  // namespace stark
