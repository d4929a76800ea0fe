// stark_mlwe_amd/csrc/poseidon_pair.hpp — Poseidon sponges with TWO waves per batch of 64 states (gfx950).
//
// Why: one lane per sponge with the state in LDS (poseidon_dev.hpp) is capped by LDS capacity at
// 160 KiB / (t*32 B) = 301 states per CU for t = 17, i.e. ONE wave per SIMD, and a lone wave issues a
// VALU instruction only every ~5 cycles (measured: v_mad_u64_u32 11.0 vs 5.4 cycles/instr at 1 vs 2
// waves per SIMD).  Here a workgroup is a pair of waves (X, Y) that share the 64 LDS-resident states of
// the batch: lane l of both waves works on state l, each wave on its own part of the linear algebra.
// Same LDS per state, twice the waves per SIMD (4 workgroups of 40 KiB per CU for t = 17).
//   * full rounds : S-box on the wave's own elements; the dense MDS product
//                   - t = 17: on the MATRIX CORES (pair_apply_mds_mfma below): int8 MFMA over signed radix-256 digits, X the even
//                     output rows, Y the odd ones, each wave holding the whole state as operand registers;
//                   - t = 9: as in-place L*(U*x) on the VALU, X taking the even rows and Y the odd rows of each step (rows
//                     2k/2k+1 read only slots >= 2k, so a single barrier between the step's reads and its two writes keeps it race-free);
//   * partial rounds, in blocks of 4 (see permute_core in poseidon_dev.hpp for the algebra):
//       phase 1  X runs the S-box chain: x_q, then a_q x_q + sum_{p<q} gamma x_p + its share (lanes 1..NXD) of the
//                lane dot product from registers; Y computes the other lanes' part of every round's
//                dot product from the block-start state and posts it in an LDS mailbox (one barrier per round);
//       phase 2  both waves bring their half of the lanes up to date, s_j += sum_p w_{p,j} x_p (one
//                reduction per lane), X from registers, Y from the x mailboxes.
//     The mailboxes reuse state slots that are dead during the block (slot 0: X holds s0 in registers;
//     slots 1..NXD: X preloads those lanes), so the LDS budget stays at 4 workgroups per CU.
// Results are the same field values as poseidon_dev.hpp / the reference's dense rounds.
#pragma once
#include "fr.hpp"
#include "dev_common.hpp"
#include "poseidon_params.hpp"
#include "poseidon_dev.hpp"   // DsJob

#if defined(__HIPCC__)
namespace stark {

template <int T> struct PairCfg {
#ifndef STARK_NXD17
#define STARK_NXD17 5
#endif
    // X's share of the lanes in the per-round dot products: lanes 1..NXD, Y the other T - 1 - NXD.  Between two barriers X runs the S-box (three products),
    // a_q x_q, up to three gamma terms and its NXD lane terms, Y its lane terms: NXD = 5 balances the two for t = 17 (measured, 2^22 leaves: 61.0 ms at 3,
    // 60.2 at 4, 59.3 at 5, 60.5 at 6; a share that shrinks with the round as X's gamma terms grow — 6,5,5,4 / 6,6,5,4 / 7,6,5,4 — 61.4 / 61.2 / 60.7);
    // from eight terms on X's sum takes one carry pass in between (rounds 2 and 3 of a block).
    static constexpr int NXD = T == 17 ? STARK_NXD17 : 2;
#ifndef STARK_NXU17
#define STARK_NXU17 8
#endif
    static constexpr int NXU = T == 17 ? STARK_NXU17 : (T - 1) / 2;   // X updates lanes 1..NXU, Y lanes NXU+1..T-1
    static constexpr int NX = (T - 1) / 2;             // full rounds: X owns elements 0..NX-1 (S-box, absorb), Y the rest
    static constexpr int EXTRA = 3;                    // slots beyond the state: x mailboxes 1..3
    __host__ __device__ static constexpr int xslot(int p) { return p == 0 ? 0 : T + (p - 1); }
    // Dy mailboxes alias the state slots X preloaded (1..NXD), reused round-robin: Dy_q is read before barrier_{q+1},
    // Dy_{q+NXD} is written after barrier_{q+NXD-1} >= barrier_{q+1}
    __host__ __device__ static constexpr int dslot(int q) { return 1 + (q % NXD); }
    __host__ __device__ static constexpr size_t lds_bytes() { return (size_t)(T + EXTRA) * 2 * 64 * 16; }
};
static inline size_t pair_lds_bytes(int t) { return t == 17 ? PairCfg<17>::lds_bytes() : PairCfg<9>::lds_bytes(); }

struct PairState {
    uint4* st;      // [T + EXTRA][2][64]
    int lane; bool isY;
    __device__ __forceinline__ fr_t ld(int j) const {
        uint4 lo = st[(2 * j) * 64 + lane], hi = st[(2 * j + 1) * 64 + lane];
        fr_t x; x.v[0] = lo.x; x.v[1] = lo.y; x.v[2] = lo.z; x.v[3] = lo.w; x.v[4] = hi.x; x.v[5] = hi.y; x.v[6] = hi.z; x.v[7] = hi.w; return x;
    }
    __device__ __forceinline__ void sto(int j, const fr_t& x) const {
        st[(2 * j) * 64 + lane] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
        st[(2 * j + 1) * 64 + lane] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
    }
};

// In-place y = L*(U*x) over the shared state; one barrier per step.  Ends with the state consistent.
template <int T>
__device__ __forceinline__ void pair_apply_lu(const PairState& s, const uint32_t* lu) {
    const int yo = s.isY ? 1 : 0;
#pragma unroll 1
    for (int k = 0; 2 * k < T; ++k) {                       // U, top-down: rows 2k (X) and 2k+1 (Y)
        const int i = 2 * k + yo; const bool have = i < T;
        fr_t res;
        if (have) { DotAcc w; w.init(); _Pragma("unroll 1") for (int j = i; j < T; ++j) w.mac(c29(lu, i * T + j), s.ld(j)); res = w.finish(); }
        __syncthreads();
        if (have) s.sto(i, res);
    }
    __syncthreads();
#pragma unroll 1
    for (int k = 0; T - 1 - 2 * k >= 1; ++k) {              // unit-lower L, bottom-up: rows T-1-2k (X) and T-2-2k (Y)
        const int i = T - 1 - 2 * k - yo; const bool have = i >= 1;
        fr_t res;
        if (have) { DotAcc w; w.init(); _Pragma("unroll 1") for (int j = 0; j < i; ++j) w.mac(c29(lu, i * T + j), s.ld(j)); res = fr_add<PF>(s.ld(i), w.finish()); }
        __syncthreads();
        if (have) s.sto(i, res);
    }
    __syncthreads();
}
// x (canonical) -> signed radix-256 digits in place of its bytes: add 0x80 to every byte with carries, flip every byte's top bit
// (digit b = byte b - 0x80 in [-128, 127]; x < r keeps the top byte below 0x80, so 32 digits hold it).  The int8 operand form of the MFMA product.
__device__ __forceinline__ fr_t recode_signed(const fr_t& x) {
    fr_t y; uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const uint64_t t = (uint64_t)x.v[i] + 0x80808080u + c; y.v[i] = (uint32_t)t ^ 0x80808080u; c = t >> 32; }
    return y;
}
template <int T, bool RECODE = false>
__device__ __forceinline__ void pair_sbox_full(const PairState& s, const fr_t* rc) {
    const int j0 = s.isY ? PairCfg<T>::NX : 0, j1 = s.isY ? T : PairCfg<T>::NX;
    for (int j = j0; j < j1; ++j) { const fr_t x = fr_pow5_r29<PF>(fr_add<PF>(s.ld(j), rc[j])); s.sto(j, RECODE ? recode_signed(x) : x); }
    __syncthreads();
}

// ---- the dense full-round product on the matrix cores (T = 17) ---------------------------------------------------------------------
// y = M * x for the 64 sponges of the pair, x = the S-box outputs, stored RECODED (signed digits) in the state slots.  With both factors in
// signed radix-256 digits the digit-column sums S[(i,c)][n] = sum_{e,b} d_ie[c-b] * xd_e[n][b] are one int8 matrix product with a Toeplitz
// left factor (host_util.hpp mfma_frags), |S| < 2^24: exact in the i32 accumulators of v_mfma_i32_32x32x32_i8.
//   * B operand: lane l holds, for element e and column tile ct, the 16 bytes of half (l >> 5) of element e of sponge 32 ct + (l & 31) — exactly one
//     16-byte state slot.  A wave keeps the whole state in registers (17 x 2 x 4 VGPRs): every A fragment then feeds 4 MFMAs (1 KB of L2 traffic per
//     128 cycles of matrix pipe; less reuse is L1-bound, tools/mfma_dense.hip).
//   * X takes the even outputs, Y the odd ones.  Per output: 68 MFMAs into 2 x 2 tiles (digits 0..31 / 32..63 x sponges 0..31 / 32..63).
//   * D layout: lane l, register r = row (r & 3) + 8 (r >> 2) + 4 (l >> 5) of column l & 31: a sponge's 64 digit sums sit in lanes l and l + 32.
//     v_permlane32_swap(tile of sponges 0..31, tile of sponges 32..63) hands the lower lane the upper lane's rows of ITS sponge and vice versa, so
//     that afterwards lane l owns sponge l (the kernels' lane <-> sponge map) with the rows in a lane-uniform order.
//   * fold: pairs of adjacent digit sums (S0 + 256 S1 < 2^33) are shifted into the 64-bit column of weight 2^(29k) that holds the lower digit; a
//     signed carry pass (the total is a non-negative integer) leaves 29-bit limbs, and the usual Montgomery step by 2^261 returns y canonical.
// 95 k SIMD-cycles per product against ~180 k for the in-place L*U form (tools/mfma_mds.hip, profiles/r02_mfma_mds_end_to_end_prototype.jsonl).
typedef int mfma_v4i __attribute__((ext_vector_type(4)));
typedef int mfma_v16i __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void mfma_fold_rows(int64_t* col, const mfma_v16i& lo, const mfma_v16i& hi, int rt) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const mfma_v16i& a = hh ? hi : lo;
                const int64_t pair = (int64_t)a[4 * q + 2 * p] + (int64_t)a[4 * q + 2 * p + 1] * 256;
                const int c = 32 * rt + 8 * q + 4 * hh + 2 * p, k = (8 * c) / 29, sh = 8 * c - 29 * k;
                col[k] += pair * ((int64_t)1 << sh);      // pair may be negative: a product, not a shift of a signed value (undefined before C++20)
            }
}
// Precondition: every state slot holds a recoded S-box output and a barrier has passed.  Ends with the state canonical and consistent.
__device__ __forceinline__ void pair_apply_mds_mfma(const PairState& s, const void* frag) {
    constexpr int T = 17;
    const int lane = s.lane, h = lane >> 5;
    mfma_v4i b[T][2];
#pragma unroll
    for (int e = 0; e < T; ++e)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) { const uint4 u = s.st[(2 * e + h) * 64 + 32 * ct + (lane & 31)]; b[e][ct] = mfma_v4i{(int)u.x, (int)u.y, (int)u.z, (int)u.w}; }
    __syncthreads();                                   // both waves hold the state: the slots may be overwritten
    const mfma_v4i* A = reinterpret_cast<const mfma_v4i*>(frag) + lane;
#pragma unroll 1
    for (int i = s.isY ? 1 : 0; i < T; i += 2) {
        mfma_v16i acc[2][2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rt][ct][r] = 0;
        const mfma_v4i* Ai = A + (size_t)(i * 2) * T * 64;
#pragma unroll
        for (int e = 0; e < T; ++e) {
            const mfma_v4i a0 = Ai[(size_t)e * 64], a1 = Ai[(size_t)(T + e) * 64];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                acc[0][ct] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b[e][ct], acc[0][ct], 0, 0, 0);
                acc[1][ct] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b[e][ct], acc[1][ct], 0, 0, 0);
            }
        }
        mfma_v16i lo[2], hi[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)acc[rt][0][r], (unsigned)acc[rt][1][r], false, false);
                lo[rt][r] = (int)sw[0]; hi[rt][r] = (int)sw[1];
            }
        int64_t col[18];
#pragma unroll
        for (int k = 0; k < 18; ++k) col[k] = 0;
        mfma_fold_rows(col, lo[0], hi[0], 0); mfma_fold_rows(col, lo[1], hi[1], 1);
        fr_wide29 w;
#pragma unroll
        for (int k = 0; k < 17; ++k) { col[k + 1] += col[k] >> 29; w.c[k] = (uint64_t)col[k] & FR_M29; }
        w.c[17] = (uint64_t)col[17];
        s.sto(i, fr_wide29_reduce<PF>(w));
    }
    __syncthreads();
}

// s_j += w_{0,j} x0 + w_{1,j} x1 + w_{2,j} x2 + w_{3,j} x3   (one reduction)
template <int T>
__device__ __forceinline__ fr_t pair_lane_update(const uint32_t* sp, int j, const fr_t& base, const fr29_t& x0, const fr29_t& x1, const fr29_t& x2, const fr29_t& x3) {
    constexpr int W = 2 * T - 1;
    fr_wide29 u; fr_wide29_zero(u);
    fr_wide29_mac(u, c29(sp, 0 * W + T - 1 + j), x0); fr_wide29_mac(u, c29(sp, 1 * W + T - 1 + j), x1);
    fr_wide29_mac(u, c29(sp, 2 * W + T - 1 + j), x2); fr_wide29_mac(u, c29(sp, 3 * W + T - 1 + j), x3);
    // The lanes are NOT canonical between the blocks: the Montgomery quotient (below 1.1 r) is added as it is and r subtracted once, so a lane grows by
    // at most 0.1 r per block (below 2.7 r after the 16 blocks: eight words hold it); its consumers unpack and multiply (quotient bounds of fr29.hpp hold
    // for operands below 4 r), and the next full round's S-box returns canonical values.
    return fr_add<PF>(base, fr_wide29_reduce_lazy<PF>(u));
}

// One permutation by the wave pair.  Precondition: state consistent (a barrier since the last write).
// Returns lane 0 of the result in BOTH waves; with only0 the rest of the state is dead afterwards.
template <int T>
__device__ __forceinline__ fr_t pair_permute(const PairState& s, const PoseidonDev& P, bool only0, int r_begin = 0) {
    typedef PairCfg<T> Cfg;
    constexpr int NXD = Cfg::NXD, NXU = Cfg::NXU, W = 2 * T - 1;
    const int half = P.rf / 2;
    for (int r = r_begin; r < half; ++r) {
        if constexpr (T == 17) { pair_sbox_full<T, true>(s, P.rc_full + r * T); pair_apply_mds_mfma(s, (r == half - 1) ? P.mds_pre_frag : P.mds_frag); }
        else { pair_sbox_full<T>(s, P.rc_full + r * T); pair_apply_lu<T>(s, (r == half - 1) ? P.lu_pre29 : P.lu29); }
    }
    fr_t s0 = fr_zero<PF>();
    if (!s.isY) s0 = s.ld(0);
    for (int b = 0; b < P.rp / 4; ++b) {
        const uint32_t* sp = c29(P.sparse29, (size_t)(4 * b) * W);
        const uint32_t* g = c29(P.gamma29, (size_t)b * 6);
        if (!s.isY) {
            // ---- X: the S-box chain ------------------------------------------------------------------------
            fr_t keep[NXD];
#pragma unroll
            for (int j = 0; j < NXD; ++j) keep[j] = s.ld(1 + j);                  // their slots become Y's mailboxes for this block
            __syncthreads();                                           // S: mailboxes may be written from here on
            // x_0..x_3 live in their LDS mailboxes (this wave reads back its own writes in order); keeping them in
            // registers across the four rounds costs 36 VGPRs and pushes the allocator into scratch.
#define STARK_PAIR_ROUND(q)                                                                           \
            {                                                                                         \
                __builtin_amdgcn_sched_barrier(0);                     /* keep the rounds apart: less register pressure */ \
                const fr_t xq = fr_pow5_r29<PF>(fr_add<PF>(s0, P.rc_partial[4 * b + q]));              \
                s.sto(Cfg::xslot(q), xq);                                                             \
                fr_wide29 acc; fr_wide29_zero(acc);                                                   \
                fr_wide29_mac(acc, c29(sp, q * W), fr29_unpack(xq));                                  \
                if (q > 0) fr_wide29_mac(acc, c29(g, q * (q - 1) / 2 + 0), fr29_unpack(s.ld(Cfg::xslot(0)))); \
                if (q > 1) fr_wide29_mac(acc, c29(g, q * (q - 1) / 2 + 1), fr29_unpack(s.ld(Cfg::xslot(1)))); \
                if (q > 2) fr_wide29_mac(acc, c29(g, q * (q - 1) / 2 + 2), fr29_unpack(s.ld(Cfg::xslot(2)))); \
                if (1 + q + NXD > fr29_max_terms<PF>()) fr_wide29_norm(acc);   /* terms between carry passes (fr29.hpp); with NXD = 3 never */ \
                _Pragma("unroll") for (int j = 0; j < NXD; ++j) fr_wide29_mac(acc, c29(sp, q * W + 1 + j), fr29_unpack(keep[j])); \
                const fr_t part = fr_wide29_reduce<PF>(acc);                                          \
                __syncthreads();                                       /* barrier_q: Dy_q is posted */ \
                s0 = fr_add<PF>(part, s.ld(Cfg::dslot(q)));                                           \
            }
            STARK_PAIR_ROUND(0) STARK_PAIR_ROUND(1) STARK_PAIR_ROUND(2) STARK_PAIR_ROUND(3)
#undef STARK_PAIR_ROUND
            // ---- X: lanes 1..NXU up to date -------------------------------------------------------------------
            const fr29_t x0 = fr29_unpack(s.ld(Cfg::xslot(0))), x1 = fr29_unpack(s.ld(Cfg::xslot(1))), x2 = fr29_unpack(s.ld(Cfg::xslot(2))), x3 = fr29_unpack(s.ld(Cfg::xslot(3)));
#pragma unroll
            for (int j = 1; j <= NXD; ++j) { __builtin_amdgcn_sched_barrier(0); s.sto(j, pair_lane_update<T>(sp, j, keep[j - 1], x0, x1, x2, x3)); }
#pragma unroll 1
            for (int j = NXD + 1; j <= NXU; ++j) s.sto(j, pair_lane_update<T>(sp, j, s.ld(j), x0, x1, x2, x3));
        } else {
            // ---- Y: three quarters of every round's dot product, from the block-start state ----------------------
            __syncthreads();                                           // S
            for (int q = 0; q < 4; ++q) {
                DotAcc acc; acc.init();
#pragma unroll 1
                for (int j = NXD + 1; j < T; ++j) acc.mac(c29(sp, q * W + j), s.ld(j));
                const fr_t dy = acc.finish();
                s.sto(Cfg::dslot(q), dy);
                __syncthreads();                                       // barrier_q
            }
            // ---- Y: lanes NXU+1..T-1 up to date (x_0..x_3 were posted before barrier_0..3) ---------------------------
            const fr29_t x0 = fr29_unpack(s.ld(Cfg::xslot(0))), x1 = fr29_unpack(s.ld(Cfg::xslot(1))), x2 = fr29_unpack(s.ld(Cfg::xslot(2))), x3 = fr29_unpack(s.ld(Cfg::xslot(3)));
#pragma unroll 1
            for (int j = NXU + 1; j < T; ++j) s.sto(j, pair_lane_update<T>(sp, j, s.ld(j), x0, x1, x2, x3));
        }
        __syncthreads();                                               // E: lanes 1..T-1 consistent, mailboxes free
    }
    if (!s.isY) s.sto(0, s0);
    __syncthreads();
    for (int r = half; r < P.rf; ++r) {
        const bool squeeze = only0 && r == P.rf - 1;
        if constexpr (T == 17) { if (squeeze) pair_sbox_full<T>(s, P.rc_full + r * T); else pair_sbox_full<T, true>(s, P.rc_full + r * T); }
        else pair_sbox_full<T>(s, P.rc_full + r * T);
        if (squeeze) {                                      // squeeze: row 0 only, split over the two waves
            const int j0 = s.isY ? Cfg::NX : 0, j1 = s.isY ? T : Cfg::NX;
            DotAcc acc; acc.init();
#pragma unroll 1
            for (int j = j0; j < j1; ++j) acc.mac(c29(P.row0_29, j), s.ld(j));
            s.sto(T + (s.isY ? 1 : 0), acc.finish());                // two of the extra slots: not part of the state
            __syncthreads();
            fr_t out = fr_add<PF>(s.ld(T + 0), s.ld(T + 1));
            __syncthreads();                                 // the slots are reused by the next permutation
            return out;
        }
        if constexpr (T == 17) pair_apply_mds_mfma(s, P.mds_frag); else pair_apply_lu<T>(s, P.lu29);
    }
    return s.ld(0);
}

__device__ __forceinline__ PairState pair_setup(uint4* lds) {
    PairState s; s.st = lds; s.lane = threadIdx.x & 63;
    s.isY = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) != 0;      // wave-uniform by construction: tell the compiler (scalar branches, scalar constant loads)
    return s;
}

// K3 (pair form): h[i] = hash_leaf_pair(f[i], f_next[i/m] or 0).  Block = 128 threads = 64 states.
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2))) k_leaf_pair2(PoseidonDev P, const fr_t* __restrict__ leafc, const fr_t* __restrict__ f,
                                                    const fr_t* __restrict__ f_next, size_t n, size_t m, fr_t* __restrict__ h) {
    extern __shared__ uint4 lds[];
    PairState s = pair_setup(lds);
    const size_t i = (size_t)blockIdx.x * 64 + s.lane; const bool live = i < n; const size_t ii = live ? i : n - 1;   // tail lanes recompute the last leaf
    // Round 0 in closed form: 15 of the 17 lanes of the transcript template are constants, so after ARK and
    // S-box the MDS output is  K_i + M[i][4]*x4 + M[i][5]*x5  with K precomputed on the host
    // (leafc = [K(17) | M[:,4](17) | M[:,5](17)]).  Both waves compute x4, x5; each fills its own lanes.
    const fr_t x4 = fr_pow5_r29<PF>(fr_add<PF>(ldg(f + ii), P.rc_full[4]));
    const fr_t x5 = fr_pow5_r29<PF>(fr_add<PF>(f_next ? ldg(f_next + ii / m) : fr_zero<PF>(), P.rc_full[5]));
    const int j0 = s.isY ? PairCfg<17>::NX : 0, j1 = s.isY ? 17 : PairCfg<17>::NX;
    const uint32_t* m45 = reinterpret_cast<const uint32_t*>(leafc + 51);   // columns 4 and 5 of M in radix 2^29 (17 + 17 entries)
    const fr29_t x4u = fr29_unpack(x4), x5u = fr29_unpack(x5);
    for (int j = j0; j < j1; ++j) {
        fr_wide29 w; fr_wide29_zero(w);
        fr_wide29_mac(w, c29(m45, j), x4u); fr_wide29_mac(w, c29(m45, 17 + j), x5u);
        s.sto(j, fr_add<PF>(leafc[j], fr_wide29_reduce<PF>(w)));
    }
    __syncthreads();
    fr_t out = pair_permute<17>(s, P, true, 1);
    if (live && !s.isY) stg(h + i, out);
}

// K4 (pair form): one Merkle level / the pair-leaf level (DsJob as in poseidon_dev.hpp).
template <int T>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2))) k_hash_ds2(PoseidonDev P, DsJob J, const fr_t* __restrict__ in0, const fr_t* __restrict__ in1, fr_t* __restrict__ out) {
    extern __shared__ uint4 lds[];
    constexpr int rate = T - 1;
    PairState s = pair_setup(lds);
    const size_t k0 = (size_t)blockIdx.x * 64 + s.lane; const bool live = k0 < J.n_out; const size_t k = live ? k0 : J.n_out - 1;
    const int o0 = s.isY ? PairCfg<T>::NX : 0, o1 = s.isY ? T : PairCfg<T>::NX;      // elements this wave fills / absorbs into
    for (int j = o0; j < o1; ++j) s.sto(j, fr_zero<PF>());
    __syncthreads();
    const size_t cnt = J.mode == 1 ? 2 : ((k + 1) * J.arity <= J.n_in ? J.arity : J.n_in - k * J.arity);
    const size_t total = 4 + cnt + 1, nperm = (total + rate - 1) / rate;
    // The widest stream in the block decides how many permutations every lane walks through (barriers are wave-level: both waves
    // of the pair must execute the same sequence).  A lane with a SHORTER stream (the ragged last node of a level) sits out the
    // first max_perm - nperm of them — it permutes a dead state, clears it, and starts absorbing late — so that EVERY lane's result
    // is lane 0 of the last permutation: nothing has to be carried in registers across the permutations (the kernel sits at the
    // 256-VGPR limit; carrying a per-lane result across pair_permute spilled 118 VGPRs to scratch).
    const size_t max_total = 4 + (J.mode == 1 ? 2 : J.arity) + 1, max_perm = (max_total + rate - 1) / rate;
    const size_t skip = max_perm - nperm;
    size_t q = 0; fr_t res = fr_zero<PF>();
    for (size_t pidx = 0; pidx < max_perm; ++pidx) {
        if (pidx >= skip) {
            if (skip && pidx == skip) for (int j = o0; j < o1; ++j) s.sto(j, fr_zero<PF>());       // the dead permutations left garbage behind
            for (int cur = 0; cur < rate && q < total; ++cur, ++q) {
                if (cur < o0 || cur >= o1) continue;
                fr_t x;
                if (q == 0) x = J.arity_f; else if (q == 1) x = J.level_f; else if (q == 2) x = fr_from_u64<PF>(ds_position(J, k)); else if (q == 3) x = J.label_f;
                else if (q == total - 1) x = fr_one<PF>();
                else { size_t c = q - 4; x = J.mode == 1 ? ds_pair_child(J, in0, in1, k, c) : ldg(in0 + k * J.arity + c); }
                s.sto(cur, fr_add<PF>(s.ld(cur), x));
            }
        }
        __syncthreads();
        res = pair_permute<T>(s, P, pidx + 1 == max_perm);
        __syncthreads();
    }
    if (live && !s.isY) stg(out + k0, res);
}

}  // namespace stark
#endif
