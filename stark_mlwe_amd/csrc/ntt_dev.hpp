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
// Inside a tile the elements live in LDS as NINE 29-BIT LIMBS, lazily reduced (fr29.hpp): decimation-in-time butterflies
// (a, b) <- (a + w b, a - w b + 4r), one carry-free 81-MAC product and one Montgomery step by 2^261 per butterfly, sums and
// differences limb-wise without carries, a carry pass every third stage.  In DIT every stage multiplies one operand, so values
// grow linearly (<= 4r per stage, 41 r over ten stages, inside the 2^261 / r = 128 (Pallas) / 70 (BLS12-381) head-room of nine
// limbs); decimation in frequency would double the unmultiplied sum every stage.  Against the eight-word exact form (8x8 MACs
// with a carry add each, conditional subtractions in every add/sub) a butterfly costs 995 instead of 1 420-1 530 SIMD cycles
// (tools/bf_bench.hip).  Points are stored bit-reversed on load, results leave in natural order; the last stage is fused with
// the pass epilogue (inter-pass twiddle / coset post-scale / n^-1), which is also the product that brings the value back
// below 2r for the canonical eight-word store.  Every table the kernels read (stage twiddles, inter-pass twiddles, coset
// powers, scales) carries the factor 32 = 2^261 / 2^256, so that one Montgomery step IS the product in the 2^256 domain.
#pragma once
#include <hip/hip_runtime.h>
#include "fr.hpp"
#include "fr29.hpp"
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
    const fr_t* pre_small;   // first strided pass of a coset transform, merged form: (g^S)^p by POINT index p (2^log_b entries, cache-resident); the other factor
                             // g^rest of g^(pS + rest) is common to a column and sits in that pass's twiddle table (tw_direct = w_m^(rest k) g^rest): one
                             // full-length table read per element less than with pre_direct
    uint32_t nz_points;      // first pass of a zero-padded transform (LDE): only the points p < nz_points of every sub-NTT are non-zero
                             // in memory; the rest is taken as zero without being read (0 = all points are read)
    uint32_t dlimb[9];       // 4r in borrow-proof nine-limb form (fr29.hpp ntt29_offset)
    uint32_t ntiles;         // tiles of the launch (a workgroup walks tiles blockIdx.x, + gridDim.x, ...)
    uint64_t pre_row_stride; // != 0 (multi-GPU column slabs): the pre-scale exponent of tile element (p, column) is the GLOBAL natural index
                             // p * pre_row_stride + rest0 + column, not the position inside the local slab
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL access of the wave
// (s_waitcnt vmcnt(0)), which would turn the prefetch of the next tile into a blocking load at the first stage barrier.  The pass
// kernels exchange data between threads through LDS alone (a thread's global loads and stores touch elements no other thread of
// the launch touches), so the LDS counter is the only one the barrier has to drain.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- the tile in LDS: nine 29-bit limbs per element, SoA over the slot index (two 16-byte halves + the top limb) ----
struct Tile29 { uint4* lo; uint4* mid; uint32_t* top; };
__device__ __forceinline__ fr29_t t29_ld(const Tile29& T, int slot) {
    const uint4 a = T.lo[slot], b = T.mid[slot]; fr29_t x;
    x.l[0] = a.x; x.l[1] = a.y; x.l[2] = a.z; x.l[3] = a.w; x.l[4] = b.x; x.l[5] = b.y; x.l[6] = b.z; x.l[7] = b.w; x.l[8] = T.top[slot]; return x;
}
__device__ __forceinline__ void t29_st(const Tile29& T, int slot, const fr29_t& x) {
    T.lo[slot] = make_uint4(x.l[0], x.l[1], x.l[2], x.l[3]); T.mid[slot] = make_uint4(x.l[4], x.l[5], x.l[6], x.l[7]); T.top[slot] = x.l[8];
}
__device__ __forceinline__ uint32_t bitrev(uint32_t x, int bits) { return bits == 0 ? 0u : (__brev(x) >> (32 - bits)); }
// LDS words of a tile with E elements and NT stage twiddles (both SoA blocks 16-byte aligned)
__host__ __device__ constexpr size_t ntt_tile_words(size_t E, size_t NT) { return ((9 * E + 3) & ~(size_t)3) + ((9 * NT + 3) & ~(size_t)3); }
struct TileLds { Tile29 data, tw; };
__device__ __forceinline__ TileLds tile_lds(uint4* lds, int E, int NT) {
    uint32_t* w = reinterpret_cast<uint32_t*>(lds);
    uint32_t* t = w + ((9 * (size_t)E + 3) & ~(size_t)3);
    return TileLds{Tile29{lds, lds + E, w + 8 * (size_t)E}, Tile29{reinterpret_cast<uint4*>(t), reinterpret_cast<uint4*>(t) + NT, t + 8 * (size_t)NT}};
}
template <class F>
__device__ __forceinline__ void load_stage_tw(const Tile29& T, const fr_t* tw, int nt) {
    for (int i = threadIdx.x; i < nt; i += blockDim.x) t29_st(T, i, fr29_unpack(ldg(tw + i)));
}
// The first HEAD = min(3, log_b) stages in registers (fr29.hpp ntt29_head), straight from the global loads: a thread takes the
// 2^HEAD points u + (B/2^HEAD) t of one column — in bit-reversed placement they are the consecutive rows (bitrev(u) << HEAD) + bitrev(t)
// — and stores the group into LDS.  `load(p, c)` delivers point p of tile column c as a lazy nine-limb value.
template <class F, int HEAD, class Load>
__device__ __forceinline__ void head_to_lds(const NttPassArgs& A, const Tile29& D, bool c_fastest, Load load) {
    constexpr int NP = 1 << HEAD;
    const int B = 1 << A.log_b, C = 1 << A.log_c, nu = B >> HEAD, ngrp = nu << A.log_c;
    fr29_t w1, w2, w3; w1 = w2 = w3 = fr29_unpack(ldg(A.stage_tw));
    if (HEAD >= 2) w2 = fr29_unpack(ldg(A.stage_tw + (B >> 2)));
    if (HEAD == 3) { w1 = fr29_unpack(ldg(A.stage_tw + (B >> 3))); w3 = fr29_unpack(ldg(A.stage_tw + 3 * (B >> 3))); }
    for (int idx = threadIdx.x; idx < ngrp; idx += blockDim.x) {
        int c, u;
        if (c_fastest) { c = idx & (C - 1); u = idx >> A.log_c; } else { u = idx & (nu - 1); c = idx >> (A.log_b - HEAD); }
        fr29_t x[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) x[k] = load(u + nu * (int)(__brev((uint32_t)k) >> (32 - HEAD)), c);
        ntt29_head<F, HEAD>(x, w1, w2, w3, A.dlimb);
        const int row0 = (int)(bitrev((uint32_t)u, A.log_b - HEAD) << HEAD);
#pragma unroll
        for (int k = 0; k < NP; ++k) t29_st(D, ((row0 + k) << A.log_c) + c, x[k]);
    }
}
template <class F, class Load>
__device__ __forceinline__ int head_dispatch(const NttPassArgs& A, const Tile29& D, bool c_fastest, Load load) {
    if (A.log_b >= 3) { head_to_lds<F, 3>(A, D, c_fastest, load); return 3; }
    if (A.log_b == 2) { head_to_lds<F, 2>(A, D, c_fastest, load); return 2; }
    head_to_lds<F, 1>(A, D, c_fastest, load); return 1;
}
// Stages head+1 .. log_b-1 of the decimation-in-time sub-NTTs of a tile, in LDS.  slot(row, c) = row*C + c; the points sit at
// bit-reversed rows, the outputs come out in natural order.  Stage s pairs rows j and j + 2^(s-1) of every group of 2^s rows
// with the twiddle w_(2^s)^j = w_B^(j * B / 2^s).  (The trivial twiddles j == 0 of these stages — one row in 8, 16, ... — are
// multiplied like the others: a branch on j would diverge inside a wave.)
template <class F>
__device__ __forceinline__ void lds_dit(const NttPassArgs& A, const Tile29& D, const Tile29& T, int head) {
    const int C = 1 << A.log_c, nbf = ((1 << A.log_b) >> 1) << A.log_c;
#pragma unroll 1
    for (int s = head + 1; s < A.log_b; ++s) {
        const int lh = s - 1, half = 1 << lh; const bool nrm = ntt29_norm_before(s);
#pragma unroll 1
        for (int q = threadIdx.x; q < nbf; q += blockDim.x) {
            const int c = q & (C - 1), bq = q >> A.log_c, j = bq & (half - 1), grp = bq >> lh;
            const int i0 = (((grp << s) + j) << A.log_c) + c, i1 = i0 + (half << A.log_c);
            fr29_t a = t29_ld(D, i0), b = t29_ld(D, i1);
            ntt29_butterfly<F>(a, b, t29_ld(T, j << (A.log_b - s)), A.dlimb, nrm);
            t29_st(D, i0, a); t29_st(D, i1, b);
        }
        lds_barrier();
    }
}
// The last stage (when the head left one) fused with the pass epilogue: emit(k, c, y, norm_out) receives output k of column c.
template <class F, class Emit>
__device__ __forceinline__ void last_stage(const NttPassArgs& A, const Tile29& D, const Tile29& T, int head, Emit emit) {
    const int B = 1 << A.log_b, C = 1 << A.log_c;
    const bool nrm_out = ntt29_norm_after(A.log_b);
    if (A.log_b > head) {
        const int nbf = (B >> 1) << A.log_c; const bool nrm = ntt29_norm_before(A.log_b);
#pragma unroll 1
        for (int q = threadIdx.x; q < nbf; q += blockDim.x) {
            const int c = q & (C - 1), j = q >> A.log_c;
            fr29_t a = t29_ld(D, (j << A.log_c) + c), b = t29_ld(D, ((j + (B >> 1)) << A.log_c) + c);
            ntt29_butterfly<F>(a, b, t29_ld(T, j), A.dlimb, nrm);
            emit((uint32_t)j, c, a, nrm_out); emit((uint32_t)(j + (B >> 1)), c, b, nrm_out);
        }
    } else {
        for (int q = threadIdx.x; q < (B << A.log_c); q += blockDim.x) { const int c = q & (C - 1), k = q >> A.log_c; emit((uint32_t)k, c, t29_ld(D, q), nrm_out); }
    }
}
// y * mult -> canonical eight-word value.  mult comes from a table in the kernels' domain (entries carry the factor 32 that
// turns the Montgomery step by 2^261 into a product in the 2^256 domain), y is lazily reduced.
template <class F>
__device__ __forceinline__ fr_t finish29(fr29_t y, const fr_t& mult, bool norm) {
    if (norm) fr29_norm(y);
    const fr29_t r = fr29_mul_mont<F, true>(fr29_unpack(mult), y);
    return fr29_pack_reduce<F>(r.l);
}
// an output that takes no factor: reduced without a product
template <class F>
__device__ __forceinline__ fr_t finish29_plain(fr29_t y) { fr29_partial_reduce<F>(y); return fr29_pack_reduce<F>(y.l); }

// ---- the pass kernels ------------------------------------------------------------------------------------------------------
// One workgroup per tile (the loop lets a launch use a smaller grid).  Tried and dropped: persistent workgroups that
// prefetch the next tile's points into registers underneath the LDS stages — the kernels are bound by VALU issue (SQ counters:
// the VALU pipe of a SIMD is busy 90-100 % of the time), not by an HBM phase that could be hidden.
//
// PRE: the pass applies a coset pre-scale on load (first pass of a coset transform): the scaled points go to LDS first and the head
// works from there — eight inlined table products would not unroll.
struct TileJob { uint64_t base, tile; };     // strided: element offset of the tile, tile index inside its outer block
template <class F, int MINW, bool PRE>
__global__ void __launch_bounds__(MINW > 2 ? 256 : 512, MINW) k_ntt_strided(NttPassArgs A, const fr_t* src, fr_t* dst) {
    extern __shared__ uint4 lds[];
    const int B = 1 << A.log_b, C = 1 << A.log_c, E = B << A.log_c, NT = B > 1 ? B >> 1 : 1;
    const TileLds L = tile_lds(lds, E, NT);
    const uint64_t tiles_per_outer = A.stride >> A.log_c;
    auto job = [&](uint64_t t) { const uint64_t outer = t / tiles_per_outer, tile = t % tiles_per_outer; return TileJob{(outer << A.log_m) + (tile << A.log_c), tile}; };
    load_stage_tw<F>(L.tw, A.stage_tw, NT);
    for (uint64_t t = blockIdx.x; t < A.ntiles; t += gridDim.x) {
        const TileJob J = job(t);
        int head;
        if (PRE && A.nz_points && A.log_b >= 3 && A.nz_points <= (uint32_t)(B >> 3)) {
            // Zero-padded input (LDE) whose non-zero points p < B/8 are the FIRST point of every head group: the three head stages pair a value with zeros,
            // (a, 0) -> (a, a), so the scaled point goes to the eight rows of its group as it is — no zeros written and read back, no head arithmetic.
            const int nzc = (int)A.nz_points << A.log_c;
            for (int idx = threadIdx.x; idx < nzc; idx += blockDim.x) {
                const int c = idx & (C - 1), p = idx >> A.log_c;
                const uint64_t g = J.base + (uint64_t)p * A.stride + c;
                const fr_t m = A.pre_small ? ldg(A.pre_small + p) : A.pre_direct ? ldg(A.pre_direct + (g & ((1ull << A.log_n) - 1)))
                                            : pow_lookup<F>(A.pre, A.pre_row_stride ? (uint64_t)p * A.pre_row_stride + A.rest0 + (J.tile << A.log_c) + c : g);
                const fr29_t x = fr29_mul_mont<F>(fr29_unpack(m), fr29_unpack(ldg(src + g)));
                const int row0 = (int)(bitrev((uint32_t)p, A.log_b - 3) << 3);
#pragma unroll
                for (int k = 0; k < 8; ++k) t29_st(L.data, ((row0 + k) << A.log_c) + c, x);
            }
            for (int idx = threadIdx.x + nzc; idx < ((B >> 3) << A.log_c); idx += blockDim.x) {      // groups whose first point is zero too (nz_points < B/8)
                const int c = idx & (C - 1), p = idx >> A.log_c; fr29_t z;
#pragma unroll
                for (int i = 0; i < 9; ++i) z.l[i] = 0;
                const int row0 = (int)(bitrev((uint32_t)p, A.log_b - 3) << 3);
#pragma unroll
                for (int k = 0; k < 8; ++k) t29_st(L.data, ((row0 + k) << A.log_c) + c, z);
            }
            head = 3;
        } else if (PRE) {
            for (int idx = threadIdx.x; idx < E; idx += blockDim.x) {
                const int c = idx & (C - 1), p = idx >> A.log_c;
                fr29_t x;
                if (A.nz_points && (uint32_t)p >= A.nz_points) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) x.l[i] = 0;
                } else {
                    const uint64_t g = J.base + (uint64_t)p * A.stride + c;
                    const fr_t m = A.pre_small ? ldg(A.pre_small + p) : A.pre_direct ? ldg(A.pre_direct + (g & ((1ull << A.log_n) - 1)))
                                                : pow_lookup<F>(A.pre, A.pre_row_stride ? (uint64_t)p * A.pre_row_stride + A.rest0 + (J.tile << A.log_c) + c : g);
                    x = fr29_mul_mont<F>(fr29_unpack(m), fr29_unpack(ldg(src + g)));
                }
                t29_st(L.data, (int)(bitrev((uint32_t)p, A.log_b) << A.log_c) + c, x);
            }
            lds_barrier();
            head = head_dispatch<F>(A, L.data, true, [&](int p, int c) -> fr29_t { return t29_ld(L.data, (int)(bitrev((uint32_t)p, A.log_b) << A.log_c) + c); });
        } else {
            head = head_dispatch<F>(A, L.data, true, [&](int p, int c) -> fr29_t {
                fr29_t x;
                if (A.nz_points && (uint32_t)p >= A.nz_points) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) x.l[i] = 0;
                } else x = fr29_unpack(ldg(src + J.base + (uint64_t)p * A.stride + c));
                return x;
            });
        }
        lds_barrier();
        lds_dit<F>(A, L.data, L.tw, head);
        // inter-pass twiddle w_m^(rest*k) and the in-place store
        const int sh = A.log_n - A.log_m;
        fr_t* out = dst + J.base;
        last_stage<F>(A, L.data, L.tw, head, [&](uint32_t k, int c, const fr29_t& y, bool nrm_out) {
            const uint64_t rest = A.rest0 + (J.tile << A.log_c) + c;
            const fr_t tw = A.tw_direct ? ldg(A.tw_direct + (uint64_t)k * A.stride + (rest - A.rest0)) : pow_lookup<F>(A.root, (rest * k) << sh);
            stg(out + (uint64_t)k * A.stride + c, finish29<F>(y, tw, nrm_out));
        });
        lds_barrier();                                                   // the tile in LDS is free again
    }
}

// Last (contiguous) pass with the digit-reversing store.  Tiles = n / (B*C).  PRE as above (single-pass coset transforms).
template <class F, int MINW, bool PRE>
__global__ void __launch_bounds__(MINW > 2 ? 256 : 512, MINW) k_ntt_last(NttPassArgs A, const fr_t* src0, fr_t* dst0) {
    extern __shared__ uint4 lds[];
    const int B = 1 << A.log_b, E = B << A.log_c, NT = B > 1 ? B >> 1 : 1;
    const TileLds L = tile_lds(lds, E, NT);
    // tile -> (vector, k2, k1 block): consecutive tiles walk k1 blocks first
    const uint64_t k1_blocks = (1ull << A.log_b1) >> A.log_c;                  // >= 1 (C divides B1)
    const uint64_t tiles_per_vec = k1_blocks << A.log_b2;
    load_stage_tw<F>(L.tw, A.stage_tw, NT);
    for (uint64_t t = blockIdx.x; t < A.ntiles; t += gridDim.x) {
        const uint64_t vec = t / tiles_per_vec, vt = t % tiles_per_vec, k2 = vt / k1_blocks, k1_0 = (vt % k1_blocks) << A.log_c;
        const fr_t* src = src0 + (vec << A.log_vec); fr_t* dst = dst0 + (vec << A.log_vec);
        int head;
        if (PRE) {
            for (int idx = threadIdx.x; idx < E; idx += blockDim.x) {
                const int p = idx & (B - 1), c = idx >> A.log_b;               // p fastest: contiguous reads
                const uint64_t g = ((((k1_0 + c) << A.log_b2) + k2) << A.log_b) + p;
                t29_st(L.data, (int)(bitrev((uint32_t)p, A.log_b) << A.log_c) + c, fr29_mul_mont<F>(fr29_unpack(pow_lookup<F>(A.pre, g)), fr29_unpack(ldg(src + g))));
            }
            lds_barrier();
            head = head_dispatch<F>(A, L.data, false, [&](int p, int c) -> fr29_t { return t29_ld(L.data, (int)(bitrev((uint32_t)p, A.log_b) << A.log_c) + c); });
        } else {
            head = head_dispatch<F>(A, L.data, false, [&](int p, int c) -> fr29_t {      // point index fastest: contiguous reads
                return fr29_unpack(ldg(src + (((((k1_0 + c) << A.log_b2) + k2) << A.log_b) + p)));
            });
        }
        lds_barrier();
        lds_dit<F>(A, L.data, L.tw, head);
        last_stage<F>(A, L.data, L.tw, head, [&](uint32_t k, int c, const fr29_t& y, bool nrm_out) {
            const uint64_t o = ((uint64_t)k << (A.log_b1 + A.log_b2)) + (k2 << A.log_b1) + k1_0 + c;
            fr_t r;
            if (A.post.lo) r = finish29<F>(y, pow_lookup<F>(A.post, o), nrm_out);
            else if (A.scale) r = finish29<F>(y, ldg(A.scale), nrm_out);
            else r = finish29_plain<F>(y);
            stg(dst + o, r);
        });
        lds_barrier();
    }
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
// Multi-GPU forward transform fed by the inverse's transposed output (stark_ntt_rows_coset_dev): the slab holds rows k1 = row0 + i of the
// [R][C] view c[k1 + R * k'] of a coefficient vector.
//   pre : dst[i][k'] = src[i][k'] * shift^(k' * R + k1)        — the coset pre-scale x[j] * shift^j at the element's natural index j
//   post: y[i][m]   *= w_n^(k1 * m)                             — the inter-step twiddle of the six-step split n = C x R
// Plain Montgomery tables (c0 = 1), not the x32 tables of the tiled NTT passes.
template <class F>
__global__ void k_rows_coset_pre(const fr_t* __restrict__ src, fr_t* __restrict__ dst, PowTable shift, uint64_t nrows, int log_cols, uint64_t row0, int log_rows) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (nrows << log_cols)) return;
    const uint64_t i = t >> log_cols, kp = t & ((1ull << log_cols) - 1);
    stg(dst + t, fr_mul<F>(ldg(src + t), pow_lookup<F>(shift, (kp << log_rows) + row0 + i)));
}
template <class F>
__global__ void k_rows_twiddle(fr_t* __restrict__ y, PowTable root, uint64_t nrows, int log_cols, uint64_t row0, int log_n) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (nrows << log_cols)) return;
    const uint64_t i = t >> log_cols, m = t & ((1ull << log_cols) - 1);
    const uint64_t e = ((row0 + i) * m) & ((1ull << log_n) - 1);
    if (e) stg(y + t, fr_mul<F>(ldg(y + t), pow_lookup<F>(root, e)));
}
// merged coset tables of a plan's first strided pass: small[p] = c32 (g^S)^p;  twc[k S + rest] = tw[k S + rest] * g^rest  (tw carries the factor 32 already)
template <class F>
__global__ void k_fill_coset_merged(PowTable g32, PowTable gplain, const fr_t* __restrict__ tw, int log_s, int log_b, fr_t* __restrict__ small, fr_t* __restrict__ twc) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (1ull << log_b)) stg(small + i, pow_lookup<F>(g32, i << log_s));
    if (i >> (log_s + log_b)) return;
    const uint64_t rest = i & ((1ull << log_s) - 1);
    stg(twc + i, rest ? fr_mul<F>(ldg(tw + i), pow_lookup<F>(gplain, rest)) : ldg(tw + i));
}
template <class F>
__global__ void k_zero_fill(fr_t* p, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) stg(p + i, fr_zero<F>());
}

}  // namespace stark
