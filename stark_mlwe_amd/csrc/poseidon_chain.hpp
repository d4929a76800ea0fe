// stark_mlwe_amd/csrc/poseidon_chain.hpp — the serial sponge as FIVE cooperating waves (gfx950): the latency form of round 3.
//
// The column sponges of DeepAliRealBuilder::build_f0 (crates/deep_ali/src/fri.rs:548-557 via tr_hash_fields_tagged, fri.rs:28-35) are
// n0/16 DEPENDENT t = 17 permutations per column; they are 99.7 % of an end-to-end prove.  One wave per sponge (poseidon_coop.hpp) spends
// 109 of its 143 us per permutation in the 64 partial rounds, three product latencies each, and on a lone wave a product is bound by the
// number of instructions one lane executes (81 + 36 multiply-accumulates and their carries: 400 ns).
//
// Here the partial rounds are UNROLLED over all 64 rounds (host_util.hpp chain tables):
//     X_{q+1} = [ c_{q+1} + sum_j u_{q,j} s_j^(0) ]  +  a_q y_q  +  Gamma_{q,q-1} y_{q-1}  +  sum_{p <= q-2} Gamma_{q,p} y_p ,   y_q = X_q^5
//                 E_q (wave C)                          wave A        wave A (previous round)   H_q - E_q (wave B)
// so that only X -> X^2 -> (a X) X^2 X^2 is on the dependent chain, and that chain runs in ROW FORM on wave A (namespace row, below): one
// product spread over the 16 lanes of a DPP row, limb c on lane c — 9 + 9 + 5 multiply-accumulates per lane-parallel product instead of 117
// on one lane: 233 ns per dependent product (tools/chain_row.hip, profiles/r03_chain_row_product.jsonl).  Four rows = four products per
// slot sharing one broadcast operand: row 0 the chain (X^2, X^3, X^5), row 1 a_q X -> a_q y_q, row 2 Gamma_{q+1,q} X -> Gamma_{q+1,q} y_q.
//   wave B  lane q keeps H_q: every round it adds Gamma_{q,p} y_p for all q >= p + 2 at once (one one-lane product, 64 lanes) and publishes
//           H_{p+2}; it has a full round of slack.
//   wave C  first E_q for all q (16-term dot products, four at a time, far ahead of the chain), then the lanes of the state,
//           s_j = s_j^(0) + sum_p w_{p,j} y_p (four rounds per product slot), finished one product after the chain ends.
// The waves talk through LDS mailboxes and monotonic counters (no barrier inside the 64 rounds).
// The eight FULL rounds keep the state in row form on all five waves (element e on row e & 3 of wave e >> 2): the 17 S-boxes run at once as
// row-form products whose broadcast operand is each row's own (ds_swizzle); the dense rows stay one-lane work on waves 0..2.
// 73 us per permutation (142 us on one wave).  The same field values as the reference's dense rounds: the host model below
// (chain_partial_model) is checked against permute_dense in tests/test_hostcheck.py, the kernel against the oracle on the GPU.
#pragma once
#include "fr.hpp"
#include "fr29.hpp"
#include "poseidon_params.hpp"
#if defined(__HIPCC__)
#include "poseidon_coop.hpp"      // full rounds on one wave, nine-limb helpers, TrMultiJob
#endif

namespace stark {

// ---- host + device: the unrolled partial rounds from the chain tables, with the one-lane product (the definition the kernel follows) ----
// s: the state after the first rf/2 full rounds (what permute_core / coop_permute hold when their partial loop starts); on return the state
// after the rp partial rounds.  Values are canonicalised between steps: this is the model of the algebra and of the tables' scaling, not of
// the lazy limb bounds (those are asserted by tools/chain_row_check.py and by the kernel's GPU tests).
FR_HD void chain_partial_model(fr_t* s, const PoseidonDev& P) {
    const int t = P.t, rp = P.rp, w = 2 * t - 1;
    auto mont = [](const uint32_t* c, const fr29_t& b) { fr29_t a; for (int i = 0; i < 9; ++i) a.l[i] = c[i]; return fr29_mul_mont<PF>(a, b); };
    auto canon = [](const fr29_t& v) { return fr29_pack_reduce<PF>(v.l); };
    fr_t H[64], F[16], prevG = fr_zero<PF>();
    for (int q = 0; q < rp; ++q) {                                                                            // wave C: E_q
        fr_t e = q + 1 < rp ? P.rc_partial[q + 1] : fr_zero<PF>();
        for (int j = 1; j < t; ++j) e = fr_add<PF>(e, canon(mont(P.sparse29 + 9 * ((size_t)q * w + j), fr29_unpack(s[j]))));
        H[q] = e;
    }
    for (int j = 0; j < t - 1; ++j) F[j] = s[j + 1];
    fr_t X = fr_add<PF>(s[0], P.rc_partial[0]);
    for (int q = 0; q < rp; ++q) {
        const fr29_t xs = fr29_unpack(X);
        const fr29_t x2 = fr29_mul_mont<PF>(xs, xs);                                                          // wave A, row 0
        const fr29_t y = fr29_mul_mont<PF>(fr29_mul_mont<PF>(xs, x2), x2);                                    //   y' = X^5 R^5 / R'^4 ... (three steps by 2^261)
        auto through = [&](const uint32_t* c) { return canon(fr29_mul_mont<PF>(fr29_mul_mont<PF>(mont(c, xs), x2), x2)); };   // rows 1, 2: (c X) X^2 X^2
        uint32_t ca[9], cg[9];
        for (int i = 0; i < 9; ++i) { ca[i] = P.chain_a[(size_t)q * 64 + 16 + i]; cg[i] = P.chain_a[(size_t)q * 64 + 32 + i]; }
        const fr_t aterm = through(ca), gterm = through(cg);
        for (int qq = q + 2; qq < rp; ++qq) {                                                                 // wave B
            uint32_t c[9]; for (int i = 0; i < 9; ++i) c[i] = P.chain_g[((size_t)q * 9 + i) * 64 + qq];
            H[qq] = fr_add<PF>(H[qq], canon(mont(c, y)));
        }
        for (int j = 0; j < t - 1; ++j) F[j] = fr_add<PF>(F[j], canon(mont(P.chain_w + ((size_t)q * (t - 1) + j) * 9, y)));   // wave C
        X = fr_add<PF>(fr_add<PF>(H[q], aterm), prevG);
        prevG = gterm;
    }
    s[0] = X;
    for (int j = 0; j < t - 1; ++j) s[j + 1] = F[j];
}

// Host: the constants of the row-form Montgomery step (row::mul below) for Pallas Fr, radix 2^29: ni = -r^-1 mod 2^261 (digit by digit: r = 1 mod 2^29,
// so the next digit is the negated current limb of 1 + ni * r), t = r - 2^254 (five limbs).
struct RowConstsHost { uint32_t ni[9]; uint32_t t[5]; };
inline RowConstsHost row_consts_host() {
    RowConstsHost K; uint64_t T[19]; for (auto& v : T) v = 0; T[0] = 1;
    for (int i = 0; i < 9; ++i) {
        const uint32_t d = (uint32_t)((0u - (uint32_t)T[i]) & FR_M29); K.ni[i] = d;
        for (int j = 0; j < 9; ++j) T[i + j] += (uint64_t)d * fr_p29<PF>(j);
        for (int k = i; k < 18; ++k) { T[k + 1] += T[k] >> 29; T[k] &= FR_M29; }
    }
    for (int k = 0; k < 5; ++k) K.t[k] = fr_p29<PF>(k);
    return K;
}

#if defined(__HIPCC__)
// ---- row form: one product on the 16 lanes of a DPP row -------------------------------------------------------------------------------------
// Lane c (= lane & 15) of a row holds limb c of the value, c = 0..8, lanes 9..15 hold 0; limbs below 2^30 + 8, value below 2^258.  Four rows per
// wave = four products per slot, all multiplied by ONE operand x whose nine limbs are wave-uniform (v_readlane -> SGPRs).
//   product     col_c = sum_k x_k y_(c-k): 9 multiply-accumulates, the row operand shifted by row_shr:k (zero fill); column 16 = x_8 y_8 on lane 8
//   split       three 29-bit pieces per column; the low nine limbs l (row_shr:1/2) and the high part h (row_shl:9/8/7)
//   Montgomery  m = l * (-r^-1) mod 2^261 (the low columns of a second 9x9 product), m r = m 2^254 + m t (9x5 product and two shifts), the exact
//               carry out of the nine low columns from columns 7 and 8; result = h + high columns + carry, pieces redistributed once more.
// x y / 2^261 (mod r), below 2^255: what fr29_mul_mont computes on one lane, in ~125 instead of ~480 instructions on the dependent chain.
namespace row {
constexpr uint32_t M29 = (1u << 29) - 1;
template <int K> __device__ __forceinline__ uint32_t shr(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + K, 0xf, 0xf, true); }   // lane i <- lane i-K of its row, else 0
template <int K> __device__ __forceinline__ uint32_t shl(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + K, 0xf, 0xf, true); }   // lane i <- lane i+K of its row, else 0
template <int K> __device__ __forceinline__ uint64_t shr64(uint64_t v) { return (uint64_t)shr<K>((uint32_t)v) | ((uint64_t)shr<K>((uint32_t)(v >> 32)) << 32); }
template <int K> __device__ __forceinline__ uint64_t shl64(uint64_t v) { return (uint64_t)shl<K>((uint32_t)v) | ((uint64_t)shl<K>((uint32_t)(v >> 32)) << 32); }
// a + (b moved along the row): written so that each shift folds into its add (v_add_u32_dpp, one instruction) — left alone the compiler
// builds v_add3_u32 out of two separate v_mov_b32_dpp (three instructions for the same sum)
__device__ __forceinline__ uint32_t keep(uint32_t v) { asm volatile("" : "+v"(v)); return v; }
struct Consts { uint32_t ni[9]; uint32_t t[5]; uint32_t dbg; };        // -r^-1 mod 2^261 and t = r - 2^254 in radix 2^29 (wave-uniform); dbg: timing experiments only (option sponge_debug; results are WRONG when set)
template <int K> struct Conv {
    static __device__ __forceinline__ void run(uint64_t& acc, const uint32_t* s, uint32_t v) { acc += (uint64_t)s[K] * shr<K>(v); Conv<K - 1>::run(acc, s, v); }
};
template <> struct Conv<0> { static __device__ __forceinline__ void run(uint64_t& acc, const uint32_t* s, uint32_t v) { acc += (uint64_t)s[0] * v; } };
__device__ __forceinline__ uint32_t mul(const uint32_t* xs, uint32_t y, const Consts& K, uint32_t cidx) {
    const uint32_t is8 = cidx == 8 ? ~0u : 0u, lt8 = cidx < 8 ? ~0u : 0u, lt9 = cidx < 9 ? ~0u : 0u, is9 = cidx == 9 ? ~0u : 0u;
    uint64_t col = 0; Conv<8>::run(col, xs, y);
    const uint64_t e = (uint64_t)xs[8] * (y & is8);
    const uint32_t p0 = (uint32_t)col & M29, p1 = (uint32_t)(col >> 29) & M29, p2 = (uint32_t)(col >> 58);
    const uint32_t e0 = (uint32_t)e & M29, e1 = (uint32_t)(e >> 29);                          // top limbs are below 2^26: column 16 is below 2^52
    const uint32_t l = keep(p0 + shr<1>(p1)) + shr<2>(p2);
    const uint32_t h = keep(keep(keep(e1 + shl<9>(p0)) + shl<8>(p1)) + shl<7>(p2)) + shl<1>(e0);
    uint64_t mc = 0; Conv<8>::run(mc, K.ni, l);
    const uint32_t m0 = (uint32_t)mc & M29, m1 = (uint32_t)(mc >> 29) & M29, m2 = (uint32_t)(mc >> 58);
    const uint32_t ml = (keep(m0 + shr<1>(m1)) + shr<2>(m2)) & lt9;
    uint64_t mt = 0; Conv<4>::run(mt, K.t, ml);
    // from here on everything fits 32 bits: the columns of m t are split into pieces like the others
    const uint32_t t0 = (uint32_t)mt & M29, t1 = (uint32_t)(mt >> 29) & M29, t2 = (uint32_t)(mt >> 58);
    const uint32_t lo7 = (ml & 127u) << 22, hi = ml >> 7;                                      // m 2^254 = m 2^22 X^8
    // the nine low limbs of V + m r (below 2^31.4 each) sum to C X^9 exactly, C <= 8: limbs 0..6 contribute less than 2^-25 to C X - (s_8 + floor(s_7 / X)),
    // an integer in {0, 1} — so C = (s_8 + floor(s_7 / X) + X - 1) >> 29
    const uint32_t sl = keep(keep(keep(l + t0) + shr<1>(t1)) + shr<2>(t2)) + shr<8>(lo7);
    const uint32_t Q = sl + shr<1>(sl >> 29);
    const uint32_t Cc = ((Q + M29) >> 29) & is8;                                               // on lane 8
    // high part, below 2^32: h + the high columns of m t and m 2^254 + the carry (brought from lane 8 to lane 0)
    const uint32_t Z = keep(keep(keep(keep(keep(h + hi) + shl<1>(lo7)) + shl<9>(t0)) + shl<8>(t1)) + shl<7>(t2)) + shl<8>(Cc);
    const uint32_t z0 = Z & ((M29 & lt8) | is8), z1 = (Z >> 29) & lt8;
    return (keep(z0 + shr<1>(z1)) + shl<1>((Z & is9) << 29)) & lt9;                            // limbs below 2^29 + 8; the top limb keeps what is above 2^232
}
// limbs below 2^32 (a sum of a few row-form values) -> limbs below 2^29 + 8, the top limb absorbing what is above 2^232
__device__ __forceinline__ uint32_t norm(uint32_t v, uint32_t cidx) {
    const uint32_t lt8 = cidx < 8 ? ~0u : 0u, is8 = cidx == 8 ? ~0u : 0u;
    return (v & ((M29 & lt8) | is8)) + shr<1>((v >> 29) & lt8);
}
__device__ __forceinline__ void bcast(uint32_t* xs, uint32_t v) {        // the nine limbs of row 0's value, wave-uniform
#pragma unroll
    for (int k = 0; k < 9; ++k) xs[k] = (uint32_t)__builtin_amdgcn_readlane((int)v, k);
}
// the nine limbs of EVERY row's own value, each row's lanes receiving their row's (ds_swizzle, bit-mask mode: lane <- (lane & 0x10) | k inside each
// group of 32 lanes): the broadcast operand of a product whose two factors both differ from row to row (the S-boxes of the full rounds)
template <int K> __device__ __forceinline__ uint32_t rowlane(uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x10 | (K << 5)); }
__device__ __forceinline__ void bcast_rows(uint32_t* xs, uint32_t v) {
    xs[0] = rowlane<0>(v); xs[1] = rowlane<1>(v); xs[2] = rowlane<2>(v); xs[3] = rowlane<3>(v); xs[4] = rowlane<4>(v);
    xs[5] = rowlane<5>(v); xs[6] = rowlane<6>(v); xs[7] = rowlane<7>(v); xs[8] = rowlane<8>(v);
}
__device__ __forceinline__ uint32_t mulg(uint32_t a, uint32_t b, const Consts& K, uint32_t cidx) { uint32_t xs[9]; bcast_rows(xs, a); return mul(xs, b, K, cidx); }
__device__ __forceinline__ uint32_t row1_to_row0(uint32_t v) { return (uint32_t)__builtin_amdgcn_permlane16_swap(v, v, false, false)[1]; }   // lanes 0..15 <- lanes 16..31
__device__ __forceinline__ uint32_t row2_to_row0(uint32_t v) { return (uint32_t)__builtin_amdgcn_permlane32_swap(v, v, false, false)[1]; }   // lanes 0..15 <- lanes 32..47
}  // namespace row

// ---- the five-wave sponge ------------------------------------------------------------------------------------------------------------------------
// LDS words behind the one-wave kernel's area (coop_lds_bytes): mailboxes of the partial rounds.  Rows of 16 words: a row-form value is written /
// read by the 16 lanes of a DPP row as they are (lanes 9..15 carry zeros).
struct ChainLds {
    uint32_t* s0;      // [17][16]  state entering the partial rounds (lazy limbs below 2^29 + 8, values below 11 r)
    uint32_t* sfin;    // [17][16]  state leaving them (values below 2 r)
    uint32_t* y;       // [64][16]  y_q = X_q^5 (three steps by 2^261), row form, as wave A produced it
    uint32_t* h;       // [64][16]  H_q - E_q from wave B (below 2r, limbs below 2^29)
    uint32_t* e;       // [64][16]  E_q from wave C
    uint32_t* ca;      // [64][64]  chain_a (wave A's row constants), copied once
    uint32_t* x;       // [17][16]  full rounds: the S-box outputs
    uint32_t* pm;      // [3][17][16] full rounds: the row sums of waves 0..2 (lazy nine-limb partials)
    uint32_t* in;      // [16][16]  the rate block being absorbed, nine limbs per field
    uint32_t* rc29;    // [8][17][16] the full rounds' constants as nine limbs and
    uint32_t* mpre;    // [17*17][9] B_1 M of the last first-half round, copied once: no global-memory latency on the chain
    uint32_t* rcp0;    // [16]      the first partial round's constant, nine limbs
    volatile uint32_t* flag;   // [0] y_ready, [1] h_ready, [2] e_ready, [3] timeout seen — monotonic counters, + 64 per permutation
};
constexpr int CHAIN_WORDS = 2 * 17 * 16 + 3 * 64 * 16 + 64 * 64 + 17 * 16 + 3 * 17 * 16 + 16 * 16 + 8 * 17 * 16 + 17 * 17 * 9 + 3 + 16 + 4;
__host__ __device__ static inline size_t chain_base_bytes() { return (coop_lds_bytes(17) + 63) / 64 * 64; }      // mailbox rows are read 16 bytes at a time
static inline size_t chain_lds_bytes() { return chain_base_bytes() + (size_t)(CHAIN_WORDS + 3) / 4 * 16; }
constexpr uint32_t CHAIN_SPIN_LIMIT = 1u << 22;     // a wave waits for a wave of its own workgroup (always resident): the bound only keeps a logic error from hanging the GPU

// Volatile accesses through a GENERIC pointer compile to flat_load / flat_store with system coherence (the address-space inference pass leaves
// volatile operations alone): hundreds of cycles on a path that is polled.  The mailboxes live in LDS, so say so: ds_read_b32 / ds_write_b32.
typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32;
__device__ __forceinline__ uint32_t lds_vload(const volatile uint32_t* p) { return *(lds_vu32*)(p); }
__device__ __forceinline__ void lds_vstore(volatile uint32_t* p, uint32_t v) { *(lds_vu32*)(p) = v; }
__device__ __forceinline__ bool chain_wait(volatile uint32_t* flags, int which, uint32_t target) {
    for (uint32_t spin = 0; spin < CHAIN_SPIN_LIMIT; ++spin) {
        if ((int32_t)(lds_vload(flags + which) - target) >= 0) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); return true; }
        if ((spin & 4095u) == 4095u && lds_vload(flags + 3)) return false;       // another wait has already timed out: do not queue up behind it
        __builtin_amdgcn_s_sleep(1);
    }
    lds_vstore(flags + 3, 1u); return false;
}
__device__ __forceinline__ void chain_post(volatile uint32_t* flags, int which, uint32_t value) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    lds_vstore(flags + which, value);
}
// the nine limbs of a mailbox row (16 words, 64-byte aligned): two 16-byte reads and one word
__device__ __forceinline__ void chain_read_row(const uint32_t* rowp, fr29_t& v) {
    const uint4 a = *reinterpret_cast<const uint4*>(rowp), b = *reinterpret_cast<const uint4*>(rowp + 4);
    v.l[0] = a.x; v.l[1] = a.y; v.l[2] = a.z; v.l[3] = a.w; v.l[4] = b.x; v.l[5] = b.y; v.l[6] = b.z; v.l[7] = b.w; v.l[8] = rowp[8];
}
// nine lazy limbs (any value below 2^261) -> canonical fr_t
__device__ __forceinline__ fr_t chain_canon(fr29_t v) {
    carry29(v); lazy_reduce29<PF>(v);
    uint32_t tt[9];
#pragma unroll
    for (int wd = 0; wd < 8; ++wd) {
        const int lo = 32 * wd, i = lo / 29, sh = lo - 29 * i;
        uint32_t x = v.l[i] >> sh;
        if (i + 1 < 9) x |= v.l[i + 1] << (29 - sh);
        if (29 - sh + 29 < 32 && i + 2 < 9) x |= v.l[i + 2] << (58 - sh);
        tt[wd] = x;
    }
    tt[8] = 0;
    fr_cond_sub<PF>(tt, 0u); fr_cond_sub<PF>(tt, 0u);
    fr_t s;
#pragma unroll
    for (int i = 0; i < 8; ++i) s.v[i] = tt[i];
    return s;
}

// State between the rounds of a sponge: ROW FORM on five waves — element e on row e & 3 of wave e >> 2, limb c on lane c of the row, lazily reduced
// (limbs below 2^31, values below 11 r), never canonical inside a sponge.
// One full round: all 17 S-boxes at once in row form, both factors of every product a row's own (three products of ~135 instructions + one LDS-crossbar
// broadcast each, against three one-lane products of ~190-230: 0.85 against 1.7 us); the dense rows on waves 0..2 as before (nine two-term
// segments, wide accumulation, one Montgomery step per segment), left as lazy nine-limb partials that the rows add up themselves.  Two barriers.
__device__ __forceinline__ void chain_full_round(uint32_t& sr, int r, const bool in_lds, const CoopLds& L, const ChainLds& C, const row::Consts& RK, int wave, int lane) {
    constexpr int T = 17;
    const uint32_t cidx = lane & 15; const int e = 4 * wave + (lane >> 4); const bool valid = e < T;
    {
        const uint32_t u = row::norm(sr + (valid ? C.rc29[(r * T + e) * 16 + cidx] : 0u), cidx);
        const uint32_t x2 = row::mulg(u, u, RK, cidx), x4 = row::mulg(x2, x2, RK, cidx), x5 = row::mulg(u, x4, RK, cidx);      // three steps by 2^261: x^5 / 2^20, as fr_pow5_r29
        if (valid) C.x[e * 16 + cidx] = x5;
    }
    __syncthreads();
    if (wave < 3) {
        const uint32_t* M = in_lds ? L.mds : C.mpre;
        const int row = lane % T, q = lane / T, g = 3 * wave + q, j0 = 2 * g, j1 = (j0 + 2 < T) ? j0 + 2 : T;
        fr29_t part; _Pragma("unroll") for (int i = 0; i < 9; ++i) part.l[i] = 0;
        if (q < 3 && j0 < T) {
            fr_wide29 acc; fr_wide29_zero(acc);
            for (int j = j0; j < j1; ++j) {
                uint32_t a[9]; lds_get29(M, row * T + j, a);
                fr29_t xj; chain_read_row(C.x + j * 16, xj);
                fr_wide29_mac_regs(acc, a, xj);
            }
            fr_wide29_mont<PF, true>(acc, part.l);
        }
        fr29_t tot = part;
#pragma unroll
        for (int k = 1; k < 3; ++k) tot = add29(tot, shfl29(part, (lane + k * T) & 63));
        if (lane < T) { carry29(tot); _Pragma("unroll") for (int i = 0; i < 9; ++i) C.pm[(wave * T + lane) * 16 + i] = tot.l[i]; }
    }
    __syncthreads();
    sr = valid ? C.pm[e * 16 + cidx] + C.pm[(T + e) * 16 + cidx] + C.pm[(2 * T + e) * 16 + cidx] : 0u;
}

// The first multipliers each helper wave needs are the same in every permutation: loaded ONCE per kernel (ChainPre), so that no global-memory latency
// stands between the barrier that opens the partial rounds and the first E_q.
struct ChainPre { fr29_t first; fr29_t first2; };      // wave B: Gamma_{lane,0}; wave C: u_{grp,j} of batch 0 and w_{grp,j} of slot 0
__device__ __forceinline__ ChainPre chain_preload(const PoseidonDev& P, int wave, int lane) {
    constexpr int T = 17, W = 2 * T - 1;
    ChainPre pre; _Pragma("unroll") for (int i = 0; i < 9; ++i) { pre.first.l[i] = 0; pre.first2.l[i] = 0; }
    if (wave == 1) { _Pragma("unroll") for (int i = 0; i < 9; ++i) pre.first.l[i] = P.chain_g[(size_t)i * 64 + lane]; }
    if (wave == 2) {
        const int j = 1 + (lane & 15), grp = lane >> 4;
#pragma unroll
        for (int i = 0; i < 9; ++i) { pre.first.l[i] = P.sparse29[9 * ((size_t)grp * W + j) + i]; pre.first2.l[i] = P.chain_w[((size_t)grp * (T - 1) + (j - 1)) * 9 + i]; }
    }
    return pre;
}
__device__ __forceinline__ void chain_partial_rounds(uint32_t& sr, const PoseidonDev& P, const ChainLds& C, const row::Consts& RK, const ChainPre& pre, int wave, int lane, uint32_t base) {
    constexpr int T = 17, W = 2 * T - 1, RP = 64;
    const uint32_t cidx = lane & 15; const int e = 4 * wave + (lane >> 4); const bool valid = e < T;
    const uint32_t s_norm = row::norm(sr, cidx);
    if (valid) C.s0[e * 16 + cidx] = s_norm;                                                          // the state entering the partial rounds, for wave C (elements 1..16)
    __syncthreads();                                                                                   // #1: s0 published, the previous permutation's mailboxes are free
    if (wave == 0) {
        // ---- A: the chain, row form ------------------------------------------------------------------------------------------------------------
        const uint32_t rw = lane >> 4;
        uint32_t x = rw == 0 ? row::norm(s_norm + C.rcp0[cidx], cidx) : 0u;                           // X_0 = s_0 + c_0: element 0 lives on this wave's row 0
        uint32_t prev = 0, kc_next = C.ca[lane];
#pragma unroll 1
        for (int q = 0; q < RP; ++q) {
            const uint32_t kc = kc_next;                                                               // this round's row constants were fetched a round ago
            kc_next = C.ca[((q + 1) & (RP - 1)) * 64 + lane];
            // E_q and H_q - E_q were posted long before this round ends (wave C runs far ahead, wave B had a full round): read the counters and the two
            // rows NOW, underneath the products, and fall back to waiting only if a counter was not there yet.  (Counter first, data after: LDS serves a
            // wave's requests in order, and the accesses are volatile, so a counter that reads "posted" vouches for the data read behind it.)
            const uint32_t fe = lds_vload(C.flag + 2), fh = lds_vload(C.flag + 1);
            uint32_t eq = rw == 0 ? lds_vload(C.e + q * 16 + cidx) : 0u, hq = rw == 0 ? lds_vload(C.h + q * 16 + cidx) : 0u;
            uint32_t xs[9]; row::bcast(xs, x);
            const uint32_t t1 = row::mul(xs, rw == 0 ? x : kc, RK, cidx);                             // X^2 | a X | Gamma X
            uint32_t x2[9]; row::bcast(x2, t1);
            const uint32_t t2 = row::mul(x2, rw == 0 ? x : t1, RK, cidx);                             // X^3 | a X^3 | Gamma X^3
            const uint32_t t3 = row::mul(x2, t2, RK, cidx);                                           // y_q | a_q y_q | Gamma_{q+1,q} y_q
            // y_q, then its counter: two volatile LDS stores in program order — LDS executes a wave's requests in order, so no s_waitcnt stands between
            // them (a release fence here would stall the chain for the store's round trip every round)
            if (rw == 0) lds_vstore(C.y + q * 16 + cidx, t3);
            if (lane == 0) lds_vstore(C.flag + 0, base + q + 1);
            const uint32_t need = base + q + 1;
            const bool early = (int32_t)((uint32_t)__builtin_amdgcn_readfirstlane((int)fe) - need) >= 0 && (int32_t)((uint32_t)__builtin_amdgcn_readfirstlane((int)fh) - need) >= 0;
            if (!early && !(RK.dbg & 1)) {
                chain_wait(C.flag, 2, need); chain_wait(C.flag, 1, need);
                eq = rw == 0 ? lds_vload(C.e + q * 16 + cidx) : 0u; hq = rw == 0 ? lds_vload(C.h + q * 16 + cidx) : 0u;
            }
            x = row::norm(hq + eq + row::row1_to_row0(t3) + row::row2_to_row0(prev), cidx);
            prev = t3;
        }
        if (rw == 0) C.sfin[cidx] = x;                                                                // s_0 after the last partial round (no constant follows)
    } else if (wave == 1) {
        // ---- B: H_q - E_q = sum_{p <= q-2} Gamma_{q,p} y_p on lane q ------------------------------------------------------------------------------
        fr29_t acc; _Pragma("unroll") for (int i = 0; i < 9; ++i) acc.l[i] = 0;
        fr29_t g = pre.first;
#pragma unroll 1
        for (int p = -2; p <= RP - 3; ++p) {
            if (p >= 0) {
                fr29_t gn = g;
                if (p + 1 <= RP - 3) { _Pragma("unroll") for (int i = 0; i < 9; ++i) gn.l[i] = P.chain_g[((size_t)(p + 1) * 9 + i) * 64 + lane]; }   // next round's multipliers, in flight during the product
                chain_wait(C.flag, 0, base + p + 1);
                fr29_t y; chain_read_row(C.y + p * 16, y);
                if (!(RK.dbg & 2)) {
                const fr29_t pr = fr29_mul_mont<PF, true>(g, y);
                acc = add29(acc, pr); carry29(acc); lazy_reduce29<PF>(acc);
                }
                g = gn;
            }
            const int qp = p + 2;
            if (lane == qp) { _Pragma("unroll") for (int i = 0; i < 9; ++i) C.h[qp * 16 + i] = acc.l[i]; }
            if (lane == 0) chain_post(C.flag, 1, base + qp + 1);
        }
    } else if (wave == 2) {
        // ---- C: E_q for every round, then the lanes of the state -------------------------------------------------------------------------------------
        const int j = 1 + (lane & 15), grp = lane >> 4;
        fr29_t sj; chain_read_row(C.s0 + 16 * j, sj);
        fr29_t un = pre.first;
#pragma unroll 1
        for (int b = 0; b < RP / 4; ++b) {
            const int q = 4 * b + grp;
            const fr29_t u = un;
            if (b + 1 < RP / 4) { _Pragma("unroll") for (int i = 0; i < 9; ++i) un.l[i] = P.sparse29[9 * ((size_t)(q + 4) * W + j) + i]; }   // the next batch's multipliers: in flight during this one
            fr29_t v = fr29_mul_mont<PF, true>(u, sj);
            v = add29(v, shfl_xor29(v, 8)); v = add29(v, shfl_xor29(v, 4)); carry29(v);
            v = add29(v, shfl_xor29(v, 2)); v = add29(v, shfl_xor29(v, 1)); carry29(v);
            if ((lane & 15) == 0) {
                if (q + 1 < RP) { const fr29_t c = fr29_unpack(ldg(P.rc_partial + q + 1)); v = add29(v, c); carry29(v); }
                lazy_reduce29<PF>(v);
#pragma unroll
                for (int i = 0; i < 9; ++i) C.e[q * 16 + i] = v.l[i];
            }
            if (lane == 0) chain_post(C.flag, 2, base + 4 * (b + 1));
        }
        fr29_t acc; _Pragma("unroll") for (int i = 0; i < 9; ++i) acc.l[i] = 0;
        fr29_t wn = pre.first2;
#pragma unroll 1
        for (int p4 = 0; p4 < RP / 4; ++p4) {
            const int p = 4 * p4 + grp;
            const fr29_t wv = wn;
            if (p4 + 1 < RP / 4) { _Pragma("unroll") for (int i = 0; i < 9; ++i) wn.l[i] = P.chain_w[((size_t)(p + 4) * (T - 1) + (j - 1)) * 9 + i]; }
            chain_wait(C.flag, 0, base + 4 * p4 + 4);
            fr29_t y; chain_read_row(C.y + p * 16, y);
            if (!(RK.dbg & 4)) {
            const fr29_t pr = fr29_mul_mont<PF, true>(wv, y);
            acc = add29(acc, pr); carry29(acc);
            }
        }
        acc = add29(acc, shfl_xor29(acc, 16)); acc = add29(acc, shfl_xor29(acc, 32)); carry29(acc);
        if (grp == 0) {
            acc = add29(acc, sj); carry29(acc); lazy_reduce29<PF>(acc);                                // 64 products and the old lane: below 76 r -> below 2 r
#pragma unroll
            for (int i = 0; i < 9; ++i) C.sfin[16 * j + i] = acc.l[i];
        }
    }
    __syncthreads();                                                                                   // #2: sfin complete
    sr = valid ? C.sfin[e * 16 + cidx] : 0u;
}

// One sponge by the workgroup's five waves: the stream elem(0), elem(1), ... (total of them; elem is asked by lanes 0..15 of wave 0 only, for q < total)
// into a state whose capacity element starts as `cap`, the lazy duplex of transcript/src/lib.rs:79-88 (permute only before absorbing more, once at
// the end) — which is also hash_with_ds_dynamic's eager sponge over ds || children || 1 (crates/poseidon/src/lib.rs:219-312: the same permutations at
// the same points, cap = 0).  Element 0 of the final state goes to *out_slot.
// state_in != nullptr: the sponge RESUMES from 17 stored elements (the streaming transcript of sumcheck_impl.hpp; cap is ignored); final_permute = false leaves
// the last block absorbed but not permuted; state_out != nullptr receives the 17 elements of the final state (canonical).
template <class Elem>
__device__ __forceinline__ void chain_sponge_ex(const PoseidonDev& P, const row::Consts& RK, uint4* lds, size_t total, const fr_t& cap, Elem elem, fr_t* out_slot,
                                                const fr_t* state_in, bool final_permute, fr_t* state_out) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    CoopLds L = coop_setup<17>(lds, P);                                  // ends with a workgroup barrier
    ChainLds C;
    { uint32_t* w = reinterpret_cast<uint32_t*>(lds) + chain_base_bytes() / 4;
      C.s0 = w; w += 17 * 16; C.sfin = w; w += 17 * 16; C.y = w; w += 64 * 16; C.h = w; w += 64 * 16; C.e = w; w += 64 * 16; C.x = w; w += 17 * 16; C.pm = w; w += 3 * 17 * 16;
      C.in = w; w += 16 * 16; C.rc29 = w; w += 8 * 17 * 16; C.ca = w; w += 64 * 64; C.mpre = w; w += 17 * 17 * 9 + 3; C.rcp0 = w; w += 16; C.flag = w;
      for (int k = threadIdx.x; k < 2 * 17 * 16 + 3 * 64 * 16 + 17 * 16 + 3 * 17 * 16 + 16 * 16; k += blockDim.x) C.s0[k] = 0u;      // s0 .. in: the pad lanes of the 16-word rows stay zero
      for (int k = threadIdx.x; k < 8 * 17; k += blockDim.x) { const fr29_t u = fr29_unpack(P.rc_full[k]); for (int i = 0; i < 16; ++i) C.rc29[k * 16 + i] = i < 9 ? u.l[i] : 0u; }
      for (int k = threadIdx.x; k < 64 * 64; k += blockDim.x) C.ca[k] = P.chain_a[k];
      for (int k = threadIdx.x; k < 17 * 17 * 9; k += blockDim.x) C.mpre[k] = P.mds_pre29[k];
      if (threadIdx.x == 0) { const fr29_t u = fr29_unpack(P.rc_partial[0]); for (int i = 0; i < 16; ++i) C.rcp0[i] = i < 9 ? u.l[i] : 0u; }
      if (threadIdx.x < 4) lds_vstore(C.flag + threadIdx.x, 0u); }
    __syncthreads();
    const ChainPre pre = chain_preload(P, wave, lane);
    const uint32_t cidx = lane & 15; const int e = 4 * wave + (lane >> 4); const bool valid = e < 17;
    // the state in row form; element 16 starts as the capacity constant (or all 17 elements come from state_in)
    if (state_in) { if (threadIdx.x < 17) { const fr29_t u = fr29_unpack(ldg(state_in + threadIdx.x)); for (int i = 0; i < 9; ++i) C.sfin[threadIdx.x * 16 + i] = u.l[i]; } }
    else if (threadIdx.x == 0) { const fr29_t u = fr29_unpack(cap); for (int i = 0; i < 9; ++i) C.sfin[16 * 16 + i] = u.l[i]; }
    __syncthreads();
    uint32_t sr = valid ? C.sfin[e * 16 + cidx] : 0u;
    auto fetch = [&](size_t base) -> fr_t {
        const size_t q = base + lane;
        if (wave == 0 && lane < 16 && q < total) return elem(q);
        return fr_zero<PF>();
    };
    auto permute = [&](uint32_t cbase) {
        const int half = P.rf / 2;
        if (!(RK.dbg & 8)) for (int r = 0; r < half; ++r) chain_full_round(sr, r, r != half - 1, L, C, RK, wave, lane);
        if (!(RK.dbg & 16)) chain_partial_rounds(sr, P, C, RK, pre, wave, lane, cbase);
        if (!(RK.dbg & 8)) for (int r = half; r < P.rf; ++r) chain_full_round(sr, r, true, L, C, RK, wave, lane);
    };
    uint32_t cbase = 0;
    fr_t nxt = fetch(0);
    for (size_t base = 0; base < total; base += 16) {
        const fr_t cur = nxt;
        if (base + 16 < total) nxt = fetch(base + 16);                   // in flight during the permutation
        if (base) { permute(cbase); cbase += 64; }
        if (wave == 0 && lane < 16) { const fr29_t u = fr29_unpack(cur); _Pragma("unroll") for (int i = 0; i < 9; ++i) C.in[lane * 16 + i] = u.l[i]; }
        __syncthreads();
        if (e < 16) sr += C.in[e * 16 + cidx];                            // absorbed lazily: limbs stay below 2^31
    }
    if (final_permute) permute(cbase);
    if (valid) C.sfin[e * 16 + cidx] = sr;                                // the state, lazy (pad lanes of a row: zeros)
    __syncthreads();
    if (threadIdx.x < 17 && (state_out || threadIdx.x == 0)) {
        fr29_t v; for (int i = 0; i < 9; ++i) v.l[i] = C.sfin[threadIdx.x * 16 + i];
        fr_t s = chain_canon(v);
        if (lds_vload(C.flag + 3)) s = fr_zero<PF>();                    // a timed-out wait (never seen) must not pass for a digest
        if (state_out) stg(state_out + threadIdx.x, s);
        if (threadIdx.x == 0 && out_slot) stg(out_slot, s);
    }
}
template <class Elem>
__device__ __forceinline__ void chain_sponge(const PoseidonDev& P, const row::Consts& RK, uint4* lds, size_t total, const fr_t& cap, Elem elem, fr_t* out_slot) {
    chain_sponge_ex(P, RK, lds, total, cap, elem, out_slot, nullptr, true, nullptr);
}

// The column sponges of build_f0 (and any other long tr_hash_fields_tagged chain): one workgroup of five waves per chain.  Same job description as
// k_tr_hash_coop_multi (block b: column b, or with J.batch column b & 3 of trace b >> 2).
__global__ void __launch_bounds__(320) __attribute__((amdgpu_waves_per_eu(1, 2))) k_tr_hash_chain(PoseidonDev P, TrMultiJob J, row::Consts RK, fr_t* __restrict__ out) {
    extern __shared__ uint4 lds[];
    const int b = blockIdx.x, c = J.batch ? (b & 3) : (J.stride ? 0 : b);
    const fr_t* prefix = J.prefix[c]; const fr_t* suffix = J.suffix[c]; const fr_t* fields = J.batch ? J.batch[b] : (J.stride ? J.fields[0] + (size_t)b * J.stride : J.fields[c]);
    const size_t np = J.np[c], kk = J.k[c], total = np + kk + (size_t)J.ns[c];
    chain_sponge(P, RK, lds, total, J.cap, [&](size_t q) -> fr_t { return q < np ? ldg(prefix + q) : (q < np + kk ? ldg(fields + (q - np)) : ldg(suffix + (q - np - kk))); }, out + b);
}

// SMALL Merkle levels and leaf layers, where the launch is one permutation's latency whatever the kernel: the same five waves per NODE (72 us per
// permutation against 142 us on one wave and ~0.4 ms in the wave-pair throughput form).  Jobs as k_hash_ds_coop / k_leaf_pair2.
__global__ void __launch_bounds__(320) __attribute__((amdgpu_waves_per_eu(1, 2))) k_hash_ds_chain(PoseidonDev P, DsJob J, row::Consts RK, const fr_t* __restrict__ in0, const fr_t* __restrict__ in1, fr_t* __restrict__ out) {
    extern __shared__ uint4 lds[];
    const size_t k = blockIdx.x;
    const size_t cnt = J.mode == 1 ? 2 : ((k + 1) * J.arity <= J.n_in ? J.arity : J.n_in - k * J.arity);
    const size_t total = 4 + cnt + 1;                                                // ds || children || 1, zero padded
    chain_sponge(P, RK, lds, total, fr_zero<PF>(), [&](size_t q) -> fr_t {
        if (q == 0) return J.arity_f; if (q == 1) return J.level_f; if (q == 2) return fr_from_u64<PF>(ds_position(J, k)); if (q == 3) return J.label_f;
        if (q == total - 1) return fr_one<PF>();
        const size_t c = q - 4; return J.mode == 1 ? ds_pair_child(J, in0, in1, k, c) : ldg(in0 + k * J.arity + c);
    }, out + k);
}
// hash_leaf_pair (fri.rs:38-44): init = the 17-element template of capi_core.hip ctx_leaf_init (elements 4, 5 are the slots of f_i and s_i, element 16 the capacity)
__global__ void __launch_bounds__(320) __attribute__((amdgpu_waves_per_eu(1, 2))) k_leaf_pair_chain(PoseidonDev P, row::Consts RK, const fr_t* __restrict__ init, const fr_t* __restrict__ f,
                                                                                                   const fr_t* __restrict__ f_next, size_t m, fr_t* __restrict__ h) {
    extern __shared__ uint4 lds[];
    const size_t i = blockIdx.x;
    chain_sponge(P, RK, lds, 9, ldg(init + 16), [&](size_t q) -> fr_t {
        if (q == 4) return ldg(f + i);
        if (q == 5) return f_next ? ldg(f_next + i / m) : fr_zero<PF>();
        return ldg(init + q);
    }, h + i);
}
#endif

}  // namespace stark
