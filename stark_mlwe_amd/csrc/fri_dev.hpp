// stark_mlwe_amd/csrc/fri_dev.hpp — streaming kernels of the FRI path on CDNA4 (gfx950).
//
//   k_fri_fold   fri_fold_layer            crates/deep_ali/src/fri.rs:85-102   out[b] = sum_t f[b*m+t] z^t
//                (compute_s_layer, fri.rs:123-143, is the view s[i] = out[i/m]; never materialised)
//   k_ali_*      deep_ali_merge_evals(_blinded)   crates/deep_ali/src/lib.rs:60-105
//   k_gather     levels[l][idx] reads for openings (merkle/src/lib.rs:261-291)
//   k_synth      synthetic trace columns for benchmarks (definition in DESIGN.md "Synthetic inputs")
// All HBM-streaming: every lane moves whole 32-byte elements as 2 x dwordx4, consecutive lanes touch
// consecutive elements.
#pragma once
#include <hip/hip_runtime.h>
#include "fr.hpp"
#include "dev_common.hpp"
#include "ntt_dev.hpp"   // PowTable / pow_lookup

namespace stark {

// Fold with m a power of two.  G = min(m,16) adjacent lanes cooperate on one output: each lane owns
// m/G consecutive inputs (Horner over its run, scaled by z^(lane_run_start)), then a butterfly sum.
// zp[t] = z^t for t < m (uniform table in global memory).
template <class F>
__global__ void __launch_bounds__(256) k_fri_fold_pow2(const fr_t* __restrict__ f, uint64_t n, const fr_t* __restrict__ zp, int log_m, int log_g, fr_t* __restrict__ out) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = 1 << (log_m - log_g);                 // inputs per lane
    const uint64_t first = tid << (log_m - log_g);        // index of this lane's first input
    const bool live = first < n;
    fr_t acc = fr_zero<F>();
    if (live) {
        const uint32_t t0 = (uint32_t)(first & ((1ull << log_m) - 1));
        // sum_{u<per} f[first+u] * z^(t0+u)
        for (int u = 0; u < per; ++u) acc = fr_add<F>(acc, fr_mul<F>(ldg(f + first + u), zp[t0 + u]));
    }
    for (int s = 0; s < log_g; ++s) acc = fr_add<F>(acc, shfl_xor_fr(acc, 1 << s));
    if (live && (tid & ((1u << log_g) - 1)) == 0) stg(out + (first >> log_m), acc);
}
// zp[t] = z^t, t < m (square-and-multiply per lane; m is a fold arity, at most a few hundred).
template <class F>
__global__ void __launch_bounds__(64) k_zpows(fr_t z, uint64_t m, fr_t* __restrict__ zp) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    fr_t acc = fr_one<F>(), b = z;
    for (uint64_t e = t; e; e >>= 1) { if (e & 1) acc = fr_mul<F>(acc, b); b = fr_sqr<F>(b); }
    stg(zp + t, acc);
}
// General m (>= 2, any): one lane per output.
template <class F>
__global__ void __launch_bounds__(256) k_fri_fold_any(const fr_t* __restrict__ f, uint64_t n_out, const fr_t* __restrict__ zp, uint64_t m, fr_t* __restrict__ out) {
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_out) return;
    fr_t acc = fr_zero<F>();
    for (uint64_t t = 0; t < m; ++t) acc = fr_add<F>(acc, fr_mul<F>(ldg(f + b * m + t), zp[t]));
    stg(out + b, acc);
}

// DEEP-ALI merge.  phi_j = a_j s_j + e_j - t_j (+ beta r_j);  f0_j = phi_j / (w^j - z).
// Each lane owns K elements j = tid + u*T (T = total lanes) so loads stay coalesced; ALL 256*K denominators of a
// workgroup are inverted with ONE Fermat inversion (Montgomery's trick, two levels: prefix products per lane, product
// scans across the lanes), as lib.rs's 2n inversions are n independent field inverses whose values do not depend
// on how they are obtained.
// Also accumulates the barycentric partial sum  sum_j phi_j w^j / (z - w^j)  per block (for c*).
#define ALI_K 8
template <class F>
__global__ void __launch_bounds__(256) k_ali_merge(const fr_t* __restrict__ a, const fr_t* __restrict__ s, const fr_t* __restrict__ e, const fr_t* __restrict__ t,
                                                   const fr_t* __restrict__ r_opt, fr_t beta, PowTable wpow, fr_t w_step /* w^T */, fr_t w_step_inv, fr_t z, uint64_t n, uint64_t j0 /* global position of element 0 */,
                                                   fr_t* __restrict__ f0, fr_t* __restrict__ block_sums) {
    __shared__ uint4 red[2 * 4];
    const uint64_t T = (uint64_t)gridDim.x * blockDim.x, tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    fr_t phi[ALI_K], pre[ALI_K];
    fr_t w = tid < n ? pow_lookup<F>(wpow, j0 + tid) : fr_one<F>();     // w^(j0 + tid)
    fr_t run = fr_one<F>();
#pragma unroll
    for (int u = 0; u < ALI_K; ++u) {                              // forward: phi_j and prefix products of (w^j - z)
        const uint64_t j = tid + (uint64_t)u * T;
        fr_t d = fr_one<F>(); phi[u] = fr_zero<F>();
        if (j < n) {
            fr_t p = fr_sub<F>(fr_add<F>(fr_mul<F>(ldg(a + j), ldg(s + j)), ldg(e + j)), ldg(t + j));
            if (r_opt) p = fr_add<F>(p, fr_mul<F>(beta, ldg(r_opt + j)));
            phi[u] = p; d = fr_sub<F>(w, z);
        }
        pre[u] = run; run = fr_mul<F>(run, d);
        w = fr_mul<F>(w, w_step);
    }
    // ONE Fermat inversion per WORKGROUP (it is ~380 products; per lane it would dwarf the ~9 products an element needs):
    // wave-level inclusive prefix / suffix products of the lanes' `run` (shuffles), the four wave totals through LDS, wave 0 inverts
    // their product, and every lane recovers  1/run = (1/P) * (other waves' totals) * (lanes before) * (lanes after).
    __shared__ uint4 tot[2 * 4], pinv[2];
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63, nwv = (int)(blockDim.x >> 6);
    fr_t pre_i = run, suf_i = run;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const fr_t a = shfl_up_fr(pre_i, d), b = shfl_dn_fr(suf_i, d);
        if (ln >= d) pre_i = fr_mul<F>(pre_i, a);
        if (ln + d < 64) suf_i = fr_mul<F>(suf_i, b);
    }
    fr_t before = shfl_up_fr(pre_i, 1), after = shfl_dn_fr(suf_i, 1);
    if (ln == 0) before = fr_one<F>();
    if (ln == 63) after = fr_one<F>();
    if (ln == 63) { tot[2 * wv] = make_uint4(pre_i.v[0], pre_i.v[1], pre_i.v[2], pre_i.v[3]); tot[2 * wv + 1] = make_uint4(pre_i.v[4], pre_i.v[5], pre_i.v[6], pre_i.v[7]); }
    __syncthreads();
    auto ld_tot = [&](const uint4* p, int i) { const uint4 lo = p[2 * i], hi = p[2 * i + 1]; fr_t x; x.v[0] = lo.x; x.v[1] = lo.y; x.v[2] = lo.z; x.v[3] = lo.w; x.v[4] = hi.x; x.v[5] = hi.y; x.v[6] = hi.z; x.v[7] = hi.w; return x; };
    if (wv == 0) {
        fr_t P = ld_tot(tot, 0); for (int i = 1; i < nwv; ++i) P = fr_mul<F>(P, ld_tot(tot, i));
        const fr_t Pi = fr_inv<F>(P);
        if (ln == 0) { pinv[0] = make_uint4(Pi.v[0], Pi.v[1], Pi.v[2], Pi.v[3]); pinv[1] = make_uint4(Pi.v[4], Pi.v[5], Pi.v[6], Pi.v[7]); }
    }
    __syncthreads();
    fr_t inv = ld_tot(pinv, 0);
    for (int i = 0; i < nwv; ++i) if (i != wv) inv = fr_mul<F>(inv, ld_tot(tot, i));                       // 1 / (this wave's total)
    inv = fr_mul<F>(fr_mul<F>(inv, before), after);                                                       // 1 / run of this lane
    fr_t bary = fr_zero<F>();
#pragma unroll
    for (int u = ALI_K - 1; u >= 0; --u) {                         // backward: peel the inverses off
        const uint64_t j = tid + (uint64_t)u * T;
        w = fr_mul<F>(w, w_step_inv);                              // back to w^j
        if (j < n) {
            fr_t dinv = fr_mul<F>(inv, pre[u]);                    // 1 / (w^j - z)
            inv = fr_mul<F>(inv, fr_sub<F>(w, z));
            fr_t q = fr_mul<F>(phi[u], dinv);
            stg(f0 + j, q);
            if (block_sums) bary = fr_sub<F>(bary, fr_mul<F>(q, w));   // phi w^j / (z - w^j) = -(phi / (w^j - z)) w^j  (only when c* is asked for)
        }
    }
    if (block_sums) {
        for (int sft = 1; sft < 64; sft <<= 1) bary = fr_add<F>(bary, shfl_xor_fr(bary, sft));
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (lane == 0) { red[2 * wave] = make_uint4(bary.v[0], bary.v[1], bary.v[2], bary.v[3]); red[2 * wave + 1] = make_uint4(bary.v[4], bary.v[5], bary.v[6], bary.v[7]); }
        __syncthreads();
        if (threadIdx.x == 0) {
            fr_t acc = fr_zero<F>();
            for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) {
                uint4 lo = red[2 * wv], hi = red[2 * wv + 1]; fr_t x;
                x.v[0] = lo.x; x.v[1] = lo.y; x.v[2] = lo.z; x.v[3] = lo.w; x.v[4] = hi.x; x.v[5] = hi.y; x.v[6] = hi.z; x.v[7] = hi.w;
                acc = fr_add<F>(acc, x);
            }
            stg(block_sums + blockIdx.x, acc);
        }
    }
}
// out[0] = scale * sum_{i<n} v[i]   (single 256-thread block; n is a few thousand block partials).
template <class F>
__global__ void __launch_bounds__(256) k_sum_single_block(const fr_t* __restrict__ v, uint64_t n, fr_t scale, fr_t* __restrict__ out) {
    __shared__ uint4 red[2 * 4];
    fr_t acc = fr_zero<F>();
    for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) acc = fr_add<F>(acc, ldg(v + i));
    for (int sft = 1; sft < 64; sft <<= 1) acc = fr_add<F>(acc, shfl_xor_fr(acc, sft));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[2 * wave] = make_uint4(acc.v[0], acc.v[1], acc.v[2], acc.v[3]); red[2 * wave + 1] = make_uint4(acc.v[4], acc.v[5], acc.v[6], acc.v[7]); }
    __syncthreads();
    if (threadIdx.x == 0) {
        fr_t tot = fr_zero<F>();
        for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) {
            uint4 lo = red[2 * wv], hi = red[2 * wv + 1]; fr_t x;
            x.v[0] = lo.x; x.v[1] = lo.y; x.v[2] = lo.z; x.v[3] = lo.w; x.v[4] = hi.x; x.v[5] = hi.y; x.v[6] = hi.z; x.v[7] = hi.w;
            tot = fr_add<F>(tot, x);
        }
        stg(out, fr_mul<F>(tot, scale));
    }
}

static __global__ void k_gather(const fr_t* __restrict__ src, const uint64_t* __restrict__ idx, uint64_t k, fr_t* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k) stg(out + i, ldg(src + idx[i]));
}

// Batched opening reads of the query phase: request i reads element index[i] of the array base[src[i]] (layers and tree levels of a
// whole proof in ONE launch instead of one synchronised gather per level).
static __global__ void k_gather_multi(const fr_t* const* __restrict__ base, const uint32_t* __restrict__ src, const uint64_t* __restrict__ index, uint64_t k, fr_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k) stg(out + i, ldg(base[src[i]] + index[i]));
}
// dst[a][b][c] (contiguous, dims Da x Db x Dc) = src[a*sa + b*sb + c*sc]: the layout changes around the all-to-all exchanges
// of the six-step NTT (32-byte elements, so even the transposing cases move whole 32-B units).
static __global__ void k_permute3(const fr_t* __restrict__ src, fr_t* __restrict__ dst, uint64_t Da, uint64_t Db, uint64_t Dc, uint64_t sa, uint64_t sb, uint64_t sc) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Da * Db * Dc) return;
    const uint64_t c = i % Dc, b = (i / Dc) % Db, a = i / (Dc * Db);
    stg(dst + i, ldg(src + a * sa + b * sb + c * sc));
}
// dst[k * stride + offset] = src[k]: interleaves the `stride` coset transforms of an LDE into natural order.
static __global__ void k_interleave(const fr_t* __restrict__ src, fr_t* __restrict__ dst, uint64_t n, uint64_t stride, uint64_t offset) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) stg(dst + k * stride + offset, ldg(src + k));
}

// Synthetic column: limb j of element i = mix64(seed + (col << 56) + 4*i + j), top limb masked to 62
// bits; the limbs are the stored (Montgomery) representation.  (DESIGN.md "Synthetic inputs".)
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x += 0x9e3779b97f4a7c15ull; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull; x = (x ^ (x >> 27)) * 0x94d049bb133111ebull; return x ^ (x >> 31);
}
static __global__ void k_synth(uint64_t seed, uint64_t col, uint64_t i0, uint64_t n, fr_t* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fr_t x;
#pragma unroll
    for (int j = 0; j < 4; ++j) { uint64_t v = mix64(seed + (col << 56) + 4 * (i0 + i) + j); if (j == 3) v &= 0x3FFFFFFFFFFFFFFFull; x.v[2 * j] = (uint32_t)v; x.v[2 * j + 1] = (uint32_t)(v >> 32); }
    stg(out + i, x);
}

}  // namespace stark
