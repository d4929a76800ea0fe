// stark_mlwe_amd/csrc/ntt_dev.hpp — radix-2 NTT / iNTT over a 255-bit prime field on CDNA4 (gfx950).
//
// Device replacement for crates/fft/src/lib.rs:6-32 (`fft`, `ifft`, `*_in_place`: thin wrappers over
// ark-poly `Radix2EvaluationDomain`): natural order in and out, out[i] = sum_j a[j] w^(ij) with
// w = get_root_of_unity(n); the inverse includes the n^-1 scaling.
//
// Decomposition (four-step, applied once or twice): n = B1*B2(*B3), each factor <= 2^10.
//   strided pass  : for every `rest` position, a size-B sub-NTT along the axis with stride S, staged
//                   through LDS as a [B points] x [C adjacent columns] tile (C*32 B contiguous per row
//                   keeps the outer-stride HBM traffic in whole 128-B segments), followed by the
//                   inter-pass twiddle w_m^(rest*k) on the way out.  In place.
//   last pass     : contiguous size-B sub-NTTs, C rows per tile chosen so that the digit-reversed
//                   (transposing) store writes C adjacent elements.
// Inside a tile: decimation-in-frequency butterflies in LDS ([element][half] 16-B slots => conflict-free
// ds_read/write_b128), stage twiddles w_B^e staged in LDS once per workgroup; the bit-reversed
// result order is undone by the LDS read address of the store phase.
#pragma once
#include <hip/hip_runtime.h>
#include "fr.hpp"
#include "dev_common.hpp"

namespace stark {

// Two-level power table of a base g:  g^e = hi[e >> LO_BITS] * lo[e & mask]   (one extra product).
struct PowTable { const fr_t* lo; const fr_t* hi; int lo_bits; };
template <class F> __device__ __forceinline__ fr_t pow_lookup(const PowTable& T, uint64_t e) {
    fr_t a = ldg(T.lo + (e & ((1ull << T.lo_bits) - 1)));
    uint64_t h = e >> T.lo_bits;
    if (h == 0) return a;
    return fr_mul<F>(a, ldg(T.hi + h));
}

struct NttPassArgs {
    int log_b;            // sub-NTT size B = 2^log_b
    int log_c;            // columns (strided) / rows (last) per tile, C = 2^log_c
    int log_n;            // full transform size (root table is for w_N, N = 2^log_n)
    uint64_t stride;      // strided pass: S (elements between consecutive points); last pass: unused
    int log_m;            // strided pass: log2(B*S), the size of the sub-problem this pass splits
    int log_b1, log_b2;   // last pass: sizes of the outer digits (P=3: o = k1*B2 + k2; P=2: log_b2 = 0; P=1: both 0)
    const fr_t* stage_tw; // B/2 entries: w_B^e
    PowTable root;        // powers of w_N (forward) or w_N^-1 (inverse)
    PowTable pre;         // coset pre-scale g^j on load (first pass, forward coset); lo == nullptr => none
    PowTable post;        // post-scale on the final store: entry e -> n^-1 * g^-e (inverse coset); lo == nullptr => none
    const fr_t* scale;    // plain inverse: n^-1 (nullptr => none)
    uint64_t rest0;       // strided pass: global index of this slab's first column (multi-GPU column blocks); 0 otherwise
    int log_vec;          // last pass: log2 of the length of one vector when several are batched (== log_b+log_b1+log_b2)
    // Direct tables (plans up to 2^24 points): one product per element instead of the two of the two-level lookup.
    const fr_t* tw_direct;   // strided pass: w_m^(rest*k) at index k*stride + rest (the layout of the sub-problem); nullptr => `root` lookup
    const fr_t* pre_direct;  // first pass: g^j at index j; nullptr => `pre` lookup
    uint32_t nz_points;      // first pass of a zero-padded transform (LDE): only the points p < nz_points of every sub-NTT are non-zero
                             // in memory; the rest is taken as zero without being read (0 = all points are read)
    uint64_t pre_row_stride; // != 0 (multi-GPU column slabs): the pre-scale exponent of tile element (p, column) is the GLOBAL natural index
                             // p * pre_row_stride + rest0 + column, not the position inside the local slab
};

__device__ __forceinline__ fr_t lds_ld(const uint4* lo, const uint4* hi, int slot) {
    uint4 a = lo[slot], b = hi[slot];
    fr_t x; x.v[0] = a.x; x.v[1] = a.y; x.v[2] = a.z; x.v[3] = a.w; x.v[4] = b.x; x.v[5] = b.y; x.v[6] = b.z; x.v[7] = b.w; return x;
}
__device__ __forceinline__ void lds_st(uint4* lo, uint4* hi, int slot, const fr_t& x) {
    lo[slot] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]); hi[slot] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
}
__device__ __forceinline__ uint32_t bitrev(uint32_t x, int bits) { return bits == 0 ? 0u : (__brev(x) >> (32 - bits)); }

// All log_b DIF stages over the tile in LDS; slot(p, c) = p*C + c.  Output X[bitrev(p)] lands at p.
template <class F>
__device__ __forceinline__ void lds_dif(uint4* dlo, uint4* dhi, const uint4* tlo, const uint4* thi, int log_b, int log_c, int nstages) {
    const int B = 1 << log_b, C = 1 << log_c, nbf = (B >> 1) << log_c;
    for (int s = 0; s < nstages; ++s) {
        const int log_half = log_b - 1 - s, half = 1 << log_half;
        for (int q = threadIdx.x; q < nbf; q += blockDim.x) {
            const int c = q & (C - 1), bq = q >> log_c;
            const int pos = bq & (half - 1), grp = bq >> log_half;
            const int i0 = ((grp << (log_half + 1)) + pos) * C + c, i1 = i0 + half * C;
            fr_t a = lds_ld(dlo, dhi, i0), b = lds_ld(dlo, dhi, i1);
            fr_t sum = fr_add<F>(a, b), dif = fr_sub<F>(a, b);
            if (pos != 0) dif = fr_mul<F>(dif, lds_ld(tlo, thi, pos << s));   // w_B^(pos*2^s); pos == 0 => 1
            lds_st(dlo, dhi, i0, sum); lds_st(dlo, dhi, i1, dif);
        }
        __syncthreads();
    }
}
// The last TAIL (<= 3) DIF stages of one column block, in registers: x[0..2^TAIL) are consecutive points of
// the block.  Their twiddles w_(2h)^pos have compile-time positions, so the trivial ones (pos == 0: all of
// the last stage, half of the one before, ...) cost nothing: 5 products per 8 points instead of 12.
template <class F, int TAIL>
__device__ __forceinline__ void dif_tail(fr_t* x, const fr_t& w1, const fr_t& w2, const fr_t& w3) {   // w_k = w_8^k (TAIL == 3); w2 = w_4 (TAIL == 2)
    if (TAIL == 3) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            fr_t a = x[p], b = x[p + 4]; x[p] = fr_add<F>(a, b); fr_t d = fr_sub<F>(a, b);
            x[p + 4] = p == 0 ? d : fr_mul<F>(d, p == 1 ? w1 : (p == 2 ? w2 : w3));
        }
    }
    if (TAIL >= 2) {
#pragma unroll
        for (int g = 0; g < (TAIL >= 2 ? (1 << (TAIL - 2)) : 0); ++g)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                fr_t a = x[4 * g + p], b = x[4 * g + p + 2]; x[4 * g + p] = fr_add<F>(a, b); fr_t d = fr_sub<F>(a, b);
                x[4 * g + p + 2] = p == 0 ? d : fr_mul<F>(d, w2);
            }
    }
#pragma unroll
    for (int g = 0; g < (1 << (TAIL - 1)); ++g) { fr_t a = x[2 * g], b = x[2 * g + 1]; x[2 * g] = fr_add<F>(a, b); x[2 * g + 1] = fr_sub<F>(a, b); }
}

template <class F>
__device__ __forceinline__ void load_stage_tw(uint4* tlo, uint4* thi, const fr_t* tw, int log_b) {
    const int nt = (1 << log_b) >> 1;
    for (int i = threadIdx.x; i < nt; i += blockDim.x) lds_st(tlo, thi, i, ldg(tw + i));
}

// Last TAIL stages of every (column, block) of the tile in registers, then the pass epilogue straight from
// the registers: STRIDED: inter-pass twiddle w_m^(rest*k) and in-place store; else the digit-reversing
// store of the last pass (optional post-scale).
template <class F, int TAIL, bool STRIDED>
__device__ __forceinline__ void tail_store(const NttPassArgs& A, const uint4* dlo, const uint4* dhi, const uint4* tlo, const uint4* thi,
                                           fr_t* dst, uint64_t tile, uint64_t k1_0, uint64_t k2) {
    constexpr int NP = 1 << TAIL;
    const int B = 1 << A.log_b, C = 1 << A.log_c, E = B << A.log_c, nblk = E >> TAIL;
    fr_t w1 = fr_zero<F>(), w2 = fr_zero<F>(), w3 = fr_zero<F>();
    if (TAIL == 3) { w1 = lds_ld(tlo, thi, B >> 3); w2 = lds_ld(tlo, thi, B >> 2); w3 = lds_ld(tlo, thi, 3 * (B >> 3)); }
    else if (TAIL == 2) w2 = lds_ld(tlo, thi, B >> 2);
    const int sh = A.log_n - A.log_m;                  // w_m = w_N^(2^sh)
    for (int idx = threadIdx.x; idx < nblk; idx += blockDim.x) {
        const int c = idx & (C - 1), blk = idx >> A.log_c;
        fr_t x[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) x[p] = lds_ld(dlo, dhi, (((blk << TAIL) + p) << A.log_c) + c);
        if (TAIL > 0) dif_tail<F, (TAIL > 0 ? TAIL : 1)>(x, w1, w2, w3);
        const uint64_t rest = A.rest0 + (tile << A.log_c) + c;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const uint32_t k = bitrev((uint32_t)((blk << TAIL) + p), A.log_b);
            fr_t y = x[p];
            if (STRIDED) {
                const uint64_t e = (rest * k) << sh;
                if (e) y = fr_mul<F>(y, A.tw_direct ? ldg(A.tw_direct + (uint64_t)k * A.stride + (rest - A.rest0)) : pow_lookup<F>(A.root, e));
                stg(dst + (uint64_t)k * A.stride + c, y);
            } else {
                const uint64_t out = ((uint64_t)k << (A.log_b1 + A.log_b2)) + (k2 << A.log_b1) + k1_0 + c;
                if (A.post.lo) y = fr_mul<F>(y, pow_lookup<F>(A.post, out));
                else if (A.scale) y = fr_mul<F>(y, *A.scale);
                stg(dst + out, y);
            }
        }
    }
}

// Strided pass.  grid.x = (n / (B*S)) * (S / C) tiles.
template <class F, int MINW>
__global__ void __launch_bounds__(MINW > 2 ? 256 : 512, MINW) k_ntt_strided(NttPassArgs A, const fr_t* src, fr_t* dst) {
    extern __shared__ uint4 lds[];
    const int B = 1 << A.log_b, C = 1 << A.log_c, E = B << A.log_c;
    uint4 *dlo = lds, *dhi = lds + E, *tlo = lds + 2 * E, *thi = tlo + (B >> 1);
    const uint64_t tiles_per_outer = A.stride >> A.log_c;
    const uint64_t outer = blockIdx.x / tiles_per_outer, tile = blockIdx.x % tiles_per_outer;
    const uint64_t base = (outer << A.log_m) + (tile << A.log_c);
    load_stage_tw<F>(tlo, thi, A.stage_tw, A.log_b);
    for (int idx = threadIdx.x; idx < E; idx += blockDim.x) {
        const int c = idx & (C - 1), p = idx >> A.log_c;
        const uint64_t g = base + (uint64_t)p * A.stride + c;
        if (A.nz_points && (uint32_t)p >= A.nz_points) { lds_st(dlo, dhi, idx, fr_zero<F>()); continue; }
        fr_t x = ldg(src + g);
        if (A.pre_direct) x = fr_mul<F>(x, ldg(A.pre_direct + (g & ((1ull << A.log_n) - 1))));
        else if (A.pre.lo) x = fr_mul<F>(x, pow_lookup<F>(A.pre, A.pre_row_stride ? (uint64_t)p * A.pre_row_stride + A.rest0 + (tile << A.log_c) + c : g));
        lds_st(dlo, dhi, idx, x);
    }
    __syncthreads();
    const int tail = A.log_b >= 3 ? 3 : A.log_b;
    lds_dif<F>(dlo, dhi, tlo, thi, A.log_b, A.log_c, A.log_b - tail);
    if (tail == 3) tail_store<F, 3, true>(A, dlo, dhi, tlo, thi, dst + base, tile, 0, 0);
    else if (tail == 2) tail_store<F, 2, true>(A, dlo, dhi, tlo, thi, dst + base, tile, 0, 0);
    else if (tail == 1) tail_store<F, 1, true>(A, dlo, dhi, tlo, thi, dst + base, tile, 0, 0);
    else tail_store<F, 0, true>(A, dlo, dhi, tlo, thi, dst + base, tile, 0, 0);
}

// Last (contiguous) pass with the digit-reversing store.  grid.x = n / (B*C) tiles.
template <class F, int MINW>
__global__ void __launch_bounds__(MINW > 2 ? 256 : 512, MINW) k_ntt_last(NttPassArgs A, const fr_t* src, fr_t* dst) {
    extern __shared__ uint4 lds[];
    const int B = 1 << A.log_b, C = 1 << A.log_c, E = B << A.log_c;
    uint4 *dlo = lds, *dhi = lds + E, *tlo = lds + 2 * E, *thi = tlo + (B >> 1);
    // tile -> (k2, k1 block): consecutive tiles walk k1 blocks first
    const uint64_t k1_blocks = (1ull << A.log_b1) >> A.log_c;                  // >= 1 (C divides B1)
    const uint64_t tiles_per_vec = k1_blocks << A.log_b2;
    const uint64_t vec = blockIdx.x / tiles_per_vec, vt = blockIdx.x % tiles_per_vec;
    const uint64_t k2 = vt / k1_blocks, k1_0 = (vt % k1_blocks) << A.log_c;
    src += vec << A.log_vec; dst += vec << A.log_vec;
    load_stage_tw<F>(tlo, thi, A.stage_tw, A.log_b);
    for (int idx = threadIdx.x; idx < E; idx += blockDim.x) {
        const int p = idx & (B - 1), c = idx >> A.log_b;               // p fastest: contiguous reads
        const uint64_t o = ((k1_0 + c) << A.log_b2) + k2;
        const uint64_t g = (o << A.log_b) + p;
        fr_t x = ldg(src + g);
        if (A.pre.lo) x = fr_mul<F>(x, pow_lookup<F>(A.pre, g));       // only when this is also the first pass
        lds_st(dlo, dhi, (p << A.log_c) + c, x);
    }
    __syncthreads();
    const int tail = A.log_b >= 3 ? 3 : A.log_b;
    lds_dif<F>(dlo, dhi, tlo, thi, A.log_b, A.log_c, A.log_b - tail);
    if (tail == 3) tail_store<F, 3, false>(A, dlo, dhi, tlo, thi, dst, 0, k1_0, k2);
    else if (tail == 2) tail_store<F, 2, false>(A, dlo, dhi, tlo, thi, dst, 0, k1_0, k2);
    else if (tail == 1) tail_store<F, 1, false>(A, dlo, dhi, tlo, thi, dst, 0, k1_0, k2);
    else tail_store<F, 0, false>(A, dlo, dhi, tlo, thi, dst, 0, k1_0, k2);
}

// Fill a PowTable: lo[i] = c0 * g^i (i < 2^lo_bits), hi[i] = g^(i << lo_bits) (i < 2^hi_bits).
template <class F>
__global__ void k_fill_pow_table(fr_t* lo, fr_t* hi, int lo_bits, int hi_bits, fr_t g, fr_t c0) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nlo = 1ull << lo_bits, nhi = 1ull << hi_bits;
    if (i < nlo) {
        fr_t acc = c0, b = g; uint64_t e = i;
        while (e) { if (e & 1) acc = fr_mul<F>(acc, b); b = fr_sqr<F>(b); e >>= 1; }
        stg(lo + i, acc);
    } else if (i < nlo + nhi) {
        fr_t acc = fr_one<F>(), b = g; uint64_t e = (i - nlo) << lo_bits;
        while (e) { if (e & 1) acc = fr_mul<F>(acc, b); b = fr_sqr<F>(b); e >>= 1; }
        stg(hi + (i - nlo), acc);
    }
}
// Direct inter-pass twiddle table of one strided pass: out[k*S + rest] = w_m^(rest*k), S = 2^(log_m - log_b).
template <class F>
__global__ void k_fill_tw_direct(PowTable root, int log_n, int log_m, int log_b, fr_t* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >> log_m) return;
    const int ls = log_m - log_b;
    const uint64_t k = i >> ls, rest = i & ((1ull << ls) - 1);
    stg(out + i, pow_lookup<F>(root, (rest * k) << (log_n - log_m)));
}
// out[j] = c0 * g^j via the two-level table (coset pre-scale, direct form)
template <class F>
__global__ void k_fill_pow_direct(PowTable t, uint64_t n, fr_t* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) stg(out + i, pow_lookup<F>(t, i));
}
template <class F>
__global__ void k_zero_fill(fr_t* p, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) stg(p + i, fr_zero<F>());
}

}  // namespace stark
