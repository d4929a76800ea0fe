// stark_mlwe_amd/csrc/poseidon_pair.hpp — Poseidon sponges with TWO waves per batch of 64 states (gfx950).
//
// Why: one lane per sponge with the state in LDS (poseidon_dev.hpp) is capped by LDS capacity at
// 160 KiB / (t*32 B) = 301 states per CU for t = 17, i.e. ONE wave per SIMD, and a lone wave issues a
// VALU instruction only every ~5 cycles (measured: v_mad_u64_u32 11.0 vs 5.4 cycles/instr at 1 vs 2
// waves per SIMD).  Here a workgroup is a pair of waves (X, Y) that share the 64 LDS-resident states of
// the batch: lane l of both waves works on state l, each wave on its own part of the linear algebra.
// Same LDS per state, twice the waves per SIMD.
//   * full rounds : S-box on the wave's own elements; the dense MDS as in-place L*(U*x), X taking the
//                   even rows and Y the odd rows of each step (rows 2k/2k+1 read only slots >= 2k, so a
//                   single barrier between the step's reads and its two writes keeps it race-free);
//   * partial rounds (sparse form): X owns lanes 0..nx-1 incl. the S-box lane, Y the rest.  Y computes
//                   its share sum_{j>=nx} u_j s_j of the NEXT round's dot product while X runs the
//                   S-box, so the two waves are balanced; the hand-offs (s0 from X, partial dot from Y)
//                   go through one LDS slot each, bracketed by two back-to-back barriers per round.
// Results are the same field values as poseidon_dev.hpp / the reference's dense rounds.
#pragma once
#include "fr.hpp"
#include "dev_common.hpp"
#include "poseidon_params.hpp"
#include "poseidon_dev.hpp"   // DsJob

#if defined(__HIPCC__)
namespace stark {

struct PairState {
    uint4* st;      // [t][2][64]
    uint4* xs0;     // [2][64]  X -> Y : s0 after the S-box (one slot)
    uint4* xdot;    // [2][64]  Y -> X : Y's share of the dot product (one slot)
    int lane; bool isY; int nx;
    __device__ __forceinline__ static fr_t rd(const uint4* base, int slot, int lane) {
        uint4 lo = base[(2 * slot) * 64 + lane], hi = base[(2 * slot + 1) * 64 + lane];
        fr_t x; x.v[0] = lo.x; x.v[1] = lo.y; x.v[2] = lo.z; x.v[3] = lo.w; x.v[4] = hi.x; x.v[5] = hi.y; x.v[6] = hi.z; x.v[7] = hi.w; return x;
    }
    __device__ __forceinline__ static void wr(uint4* base, int slot, int lane, const fr_t& x) {
        base[(2 * slot) * 64 + lane] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
        base[(2 * slot + 1) * 64 + lane] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
    }
    __device__ __forceinline__ fr_t ld(int j) const { return rd(st, j, lane); }
    __device__ __forceinline__ void sto(int j, const fr_t& x) const { wr(st, j, lane, x); }
    __device__ __forceinline__ bool owns(int j) const { return isY ? j >= nx : j < nx; }
};
static inline size_t pair_lds_bytes(int t) { return ((size_t)t * 2 * 64 + 2 * 2 * 64) * 16; }

// In-place y = L*(U*x) over the shared state; one barrier per step (see header comment).  Ends with the
// state consistent for both waves.
__device__ __forceinline__ void pair_apply_lu(const PairState& s, const fr_t* lu, int t) {
    const int yo = s.isY ? 1 : 0;
    for (int k = 0; 2 * k < t; ++k) {                       // U, top-down: rows 2k (X) and 2k+1 (Y)
        const int i = 2 * k + yo; const bool have = i < t;
        fr_t res;
        if (have) { fr_wide w; fr_wide_zero(w); for (int j = i; j < t; ++j) fr_wide_mac_f<PF>(w, lu[i * t + j], s.ld(j)); res = fr_wide_reduce<PF>(w); }
        __syncthreads();
        if (have) s.sto(i, res);
    }
    __syncthreads();
    for (int k = 0; t - 1 - 2 * k >= 1; ++k) {              // unit-lower L, bottom-up: rows t-1-2k (X) and t-2-2k (Y)
        const int i = t - 1 - 2 * k - yo; const bool have = i >= 1;
        fr_t res;
        if (have) { fr_wide w; fr_wide_zero(w); for (int j = 0; j < i; ++j) fr_wide_mac_f<PF>(w, lu[i * t + j], s.ld(j)); res = fr_add<PF>(s.ld(i), fr_wide_reduce<PF>(w)); }
        __syncthreads();
        if (have) s.sto(i, res);
    }
    __syncthreads();
}
__device__ __forceinline__ void pair_sbox_full(const PairState& s, const fr_t* rc, int t) {
    const int j0 = s.isY ? s.nx : 0, j1 = s.isY ? t : s.nx;
    for (int j = j0; j < j1; ++j) s.sto(j, fr_pow5<PF>(fr_add<PF>(s.ld(j), rc[j])));
    __syncthreads();
}
// Y's share of a partial round's dot product: sum_{j>=nx} u_j s_j.
__device__ __forceinline__ fr_t pair_dot_y(const PairState& s, const fr_t* sp, int t) {
    fr_wide w; fr_wide_zero(w);
    for (int j = s.nx; j < t; ++j) fr_wide_mac_f<PF>(w, sp[j], s.ld(j));
    return fr_wide_reduce<PF>(w);
}
// One permutation by the wave pair.  Precondition: state consistent (a barrier since the last write).
// Returns lane 0 of the result in BOTH waves; with only0 the rest of the state is dead afterwards.
__device__ __forceinline__ fr_t pair_permute(const PairState& s, const PoseidonDev& P, bool only0, int r_begin = 0) {
    const int t = P.t, half = P.rf / 2, w = 2 * t - 1;
    for (int r = r_begin; r < half; ++r) {
        pair_sbox_full(s, P.rc_full + r * t, t);
        pair_apply_lu(s, (r == half - 1) ? P.lu_pre : P.lu, t);
    }
    // Partial rounds.  Two barriers per round bracket the hand-off slots (A: published, B: consumed), so one
    // slot each suffices (4 workgroups of 38.9 KB per CU = 2 waves per SIMD).
    fr_t s0 = fr_zero<PF>();
    if (!s.isY) s0 = s.ld(0);
    else PairState::wr(s.xdot, 0, s.lane, pair_dot_y(s, P.sparse, t));
    for (int r = 0; r < P.rp; ++r) {
        const fr_t* sp = P.sparse + (size_t)r * w;
        if (!s.isY) { s0 = fr_pow5<PF>(fr_add<PF>(s0, P.rc_partial[r])); PairState::wr(s.xs0, 0, s.lane, s0); }
        __syncthreads();                                                       // A_r
        const fr_t got = s.isY ? PairState::rd(s.xs0, 0, s.lane) : PairState::rd(s.xdot, 0, s.lane);
        __syncthreads();                                                       // B_r
        if (!s.isY) {
            fr_wide acc; fr_wide_zero(acc);
            fr_wide_mac_f<PF>(acc, sp[0], s0);
            for (int j = 1; j < s.nx; ++j) { fr_t sj = s.ld(j); fr_wide_mac_f<PF>(acc, sp[j], sj); s.sto(j, fr_add<PF>(sj, fr_mul<PF>(sp[t - 1 + j], s0))); }
            s0 = fr_add<PF>(fr_wide_reduce<PF>(acc), got);
        } else {
            for (int j = s.nx; j < t; ++j) { fr_t sj = s.ld(j); s.sto(j, fr_add<PF>(sj, fr_mul<PF>(sp[t - 1 + j], got))); }
            if (r + 1 < P.rp) PairState::wr(s.xdot, 0, s.lane, pair_dot_y(s, sp + w, t));
        }
    }
    if (!s.isY) s.sto(0, s0);
    __syncthreads();
    for (int r = half; r < P.rf; ++r) {
        pair_sbox_full(s, P.rc_full + r * t, t);
        if (only0 && r == P.rf - 1) {                       // squeeze: row 0 only, split over the two waves
            const int j0 = s.isY ? s.nx : 0, j1 = s.isY ? t : s.nx;
            fr_wide acc; fr_wide_zero(acc);
            for (int j = j0; j < j1; ++j) fr_wide_mac_f<PF>(acc, P.row0[j], s.ld(j));
            PairState::wr(s.isY ? s.xdot : s.xs0, 0, s.lane, fr_wide_reduce<PF>(acc));
            __syncthreads();
            fr_t out = fr_add<PF>(PairState::rd(s.xs0, 0, s.lane), PairState::rd(s.xdot, 0, s.lane));
            __syncthreads();                                 // the slots are reused by the next permutation
            return out;
        }
        pair_apply_lu(s, P.lu, t);
    }
    return s.ld(0);
}

__device__ __forceinline__ PairState pair_setup(uint4* lds, int t) {
    PairState s; s.st = lds; s.xs0 = lds + t * 2 * 64; s.xdot = s.xs0 + 2 * 64;
    s.lane = threadIdx.x & 63; s.isY = threadIdx.x >= 64; s.nx = (t - 1) / 2;
    return s;
}

// K3 (pair form): h[i] = hash_leaf_pair(f[i], f_next[i/m] or 0).  Block = 128 threads = 64 states.
__global__ void __launch_bounds__(128) k_leaf_pair2(PoseidonDev P, const fr_t* __restrict__ leafc, const fr_t* __restrict__ f,
                                                    const fr_t* __restrict__ f_next, size_t n, size_t m, fr_t* __restrict__ h) {
    extern __shared__ uint4 lds[];
    PairState s = pair_setup(lds, 17);
    const size_t i = (size_t)blockIdx.x * 64 + s.lane; const bool live = i < n; const size_t ii = live ? i : n - 1;   // tail lanes recompute the last leaf
    // Round 0 in closed form: 15 of the 17 lanes of the transcript template are constants, so after ARK and
    // S-box the MDS output is  K_i + M[i][4]*x4 + M[i][5]*x5  with K precomputed on the host
    // (leafc = [K(17) | M[:,4](17) | M[:,5](17)]).  Both waves compute x4, x5; each fills its own lanes.
    const fr_t x4 = fr_pow5<PF>(fr_add<PF>(ldg(f + ii), P.rc_full[4]));
    const fr_t x5 = fr_pow5<PF>(fr_add<PF>(f_next ? ldg(f_next + ii / m) : fr_zero<PF>(), P.rc_full[5]));
    const int j0 = s.isY ? s.nx : 0, j1 = s.isY ? 17 : s.nx;
    for (int j = j0; j < j1; ++j) {
        fr_wide w; fr_wide_zero(w);
        fr_wide_mac_f<PF>(w, leafc[17 + j], x4); fr_wide_mac_f<PF>(w, leafc[34 + j], x5);
        s.sto(j, fr_add<PF>(leafc[j], fr_wide_reduce<PF>(w)));
    }
    __syncthreads();
    fr_t out = pair_permute(s, P, true, 1);
    if (live && !s.isY) stg(h + i, out);
}

// K4 (pair form): one Merkle level / the pair-leaf level (DsJob as in poseidon_dev.hpp).
__global__ void __launch_bounds__(128) k_hash_ds2(PoseidonDev P, DsJob J, const fr_t* __restrict__ in0, const fr_t* __restrict__ in1, fr_t* __restrict__ out) {
    extern __shared__ uint4 lds[];
    const int t = P.t, rate = t - 1;
    PairState s = pair_setup(lds, t);
    const size_t k0 = (size_t)blockIdx.x * 64 + s.lane; const bool live = k0 < J.n_out; const size_t k = live ? k0 : J.n_out - 1;
    { const int j0 = s.isY ? s.nx : 0, j1 = s.isY ? t : s.nx; for (int j = j0; j < j1; ++j) s.sto(j, fr_zero<PF>()); }
    __syncthreads();
    const size_t cnt = J.mode == 1 ? 2 : ((k + 1) * J.arity <= J.n_in ? J.arity : J.n_in - k * J.arity);
    const size_t total = 4 + cnt + 1, nperm = (total + rate - 1) / rate;
    // the widest stream in the block decides how many absorb steps every lane walks through (barriers are
    // wave-level: both waves of the pair must execute the same sequence of permutations)
    const size_t max_total = 4 + (J.mode == 1 ? 2 : J.arity) + 1, max_perm = (max_total + rate - 1) / rate;
    size_t q = 0; fr_t res = fr_zero<PF>();
    for (size_t pidx = 0; pidx < max_perm; ++pidx) {
        const bool active = pidx < nperm;
        if (active) {
            for (int cur = 0; cur < rate && q < total; ++cur, ++q) {
                if (!s.owns(cur)) continue;
                fr_t x;
                if (q == 0) x = J.arity_f; else if (q == 1) x = J.level_f; else if (q == 2) x = fr_from_u64<PF>(J.pos0 + k); else if (q == 3) x = J.label_f;
                else if (q == total - 1) x = fr_one<PF>();
                else { size_t c = q - 4; x = J.mode == 1 ? ldg((c == 0 ? in0 : in1) + k) : ldg(in0 + k * J.arity + c); }
                s.sto(cur, fr_add<PF>(s.ld(cur), x));
            }
        }
        __syncthreads();
        // lanes whose stream ended earlier keep their result and permute a dead state (values unused)
        fr_t r2 = pair_permute(s, P, pidx + 1 == max_perm);
        if (active && pidx + 1 == nperm) res = (pidx + 1 == max_perm) ? r2 : s.ld(0);
        __syncthreads();
    }
    if (live && !s.isY) stg(out + k0, res);
}

}  // namespace stark
#endif
