// tools/chain_row.hip — the serial-sponge question of VERDICT r2 item 2: ONE field product spread over the 16 lanes of a DPP row
// ("limb per lane", radix 2^29, lazy carries) instead of one product per lane.  Measures the latency of a DEPENDENT chain of such
// products on a lone wave (the regime of the column sponges, crates/deep_ali/src/fri.rs:548-557) next to tools/chain_bench.hip's
// one-lane products.  Not product code; checked by tools/chain_row_check.py (big integers).
//
// Row form: lane c (= lane & 15) of a row holds limb c of the value, c = 0..8, lanes 9..15 hold 0; limbs below 2^30 + 8, value below
// 2^258.  Four rows per wave: four products at once, all sharing ONE operand x (broadcast through v_readlane -> SGPRs), which is how a
// partial round uses them (x*x, a*x, gamma*x, ... — DESIGN.md §7a).
//   product   col_c = sum_k x_k * y_(c-k): 9 MACs, the row operand shifted by DPP row_shr:k (zero fill); column 16 = x_8*y_8 separately
//   split     three 29-bit pieces per column, redistributed with row_shr:1/2 (low part l, limbs 0..8) and row_shl:9/8/7 (high part h)
//   Montgomery by 2^261: m = l * (-r^-1) mod 2^261 (low columns of a second 9x9 product), m*r = m*2^254 + m*t (9x5 product + shifts),
//             exact carry out of the nine low columns from columns 7 and 8, result = h + high columns + carry, pieces redistributed.
// Same value as fr29_mul_mont (x*y/2^261 mod r), congruent mod r, below 2^255.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define M29 0x1fffffffu
typedef uint32_t u32; typedef uint64_t u64;
template <int K> __device__ __forceinline__ u32 shr(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + K, 0xf, 0xf, true); }   // lane i <- lane i-K of its row, else 0
template <int K> __device__ __forceinline__ u32 shl(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + K, 0xf, 0xf, true); }   // lane i <- lane i+K of its row, else 0
template <int K> __device__ __forceinline__ u64 shr64(u64 v) { return (u64)shr<K>((u32)v) | ((u64)shr<K>((u32)(v >> 32)) << 32); }
template <int K> __device__ __forceinline__ u64 shl64(u64 v) { return (u64)shl<K>((u32)v) | ((u64)shl<K>((u32)(v >> 32)) << 32); }

struct RowConsts { u32 ni[9]; u32 t[5]; };   // -r^-1 mod 2^261 and t = r - 2^254, radix 2^29 (wave-uniform: SGPRs)

template <int K> struct Conv {
    static __device__ __forceinline__ void run(u64& acc, const u32* s, u32 v) { acc += (u64)s[K] * shr<K>(v); Conv<K - 1>::run(acc, s, v); }
};
template <> struct Conv<0> { static __device__ __forceinline__ void run(u64& acc, const u32* s, u32 v) { acc += (u64)s[0] * v; } };

// y <- x * y / 2^261 (mod r) in row form; xs = the nine limbs of x (wave-uniform)
__device__ __forceinline__ u32 mulrow(const u32* xs, u32 y, const RowConsts& K, u32 cidx) {
    const u32 is8 = cidx == 8 ? ~0u : 0u, lt8 = cidx < 8 ? ~0u : 0u, lt9 = cidx < 9 ? ~0u : 0u, is9 = cidx == 9 ? ~0u : 0u;
    u64 col = 0; Conv<8>::run(col, xs, y);
    const u64 e = (u64)xs[8] * (y & is8);                                                     // column 16, lane 8
    const u32 p0 = (u32)col & M29, p1 = (u32)(col >> 29) & M29, p2 = (u32)(col >> 58);
    const u32 e0 = (u32)e & M29, e1 = (u32)(e >> 29) & M29, e2 = (u32)(e >> 58);
    const u32 l = p0 + shr<1>(p1) + shr<2>(p2);
    const u32 h = shl<9>(p0) + shl<8>(p1) + shl<7>(p2) + shl<1>(e0) + e1 + shr<1>(e2);
    u64 mc = 0; Conv<8>::run(mc, K.ni, l);
    const u32 m0 = (u32)mc & M29, m1 = (u32)(mc >> 29) & M29, m2 = (u32)(mc >> 58);
    const u32 ml = (m0 + shr<1>(m1) + shr<2>(m2)) & lt9;
    u64 mt = 0; Conv<4>::run(mt, K.t, ml);
    const u32 t0 = (u32)mt & M29, t1 = (u32)(mt >> 29) & M29, t2 = (u32)(mt >> 58);            // from here on everything fits 32 bits
    const u32 lo7 = (ml & 127u) << 22, hi = ml >> 7;                                           // m * 2^254 = m * 2^22 * X^8
    const u32 sl = l + t0 + shr<1>(t1) + shr<2>(t2) + shr<8>(lo7);                             // the nine low limbs sum to C * X^9 exactly, C <= 8
    const u32 Q = sl + shr<1>(sl >> 29);
    const u32 Cc = ((Q + M29) >> 29) & is8;                                                    // lane 8
    const u32 Z = h + hi + shl<1>(lo7) + shl<9>(t0) + shl<8>(t1) + shl<7>(t2) + shl<8>(Cc);
    const u32 z0 = Z & ((M29 & lt8) | is8), z1 = (Z >> 29) & lt8;
    return (z0 + shr<1>(z1) + shl<1>((Z & is9) << 29)) & lt9;
}

__global__ void __launch_bounds__(64) k_chain_row(const u32* in, u32* out, RowConsts K, int iters) {
    const u32 lane = threadIdx.x, cidx = lane & 15;
    u32 v = in[blockIdx.x * 64 + lane];
    for (int it = 0; it < iters; ++it) {
        u32 xs[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) xs[k] = (u32)__builtin_amdgcn_readlane((int)v, k);          // row 0 is x
        v = mulrow(xs, v, K, cidx);
    }
    out[blockIdx.x * 64 + lane] = v;
}
// probe of the DPP semantics this file relies on
__global__ void k_probe(u32* out) { const u32 l = threadIdx.x; out[l] = shr<3>(l + 100); out[64 + l] = shl<9>(l + 100); }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    // r = 2^254 + t (Pallas Fr), X = 2^29
    const u32 ni[9] = {
#include "chain_row_consts.inc"
    };
    const u32 t5[5] = {0x00000001u, 0x0c642000u + 0u, 0, 0, 0};   // replaced below
    (void)t5;
    RowConsts K;
    for (int i = 0; i < 9; ++i) K.ni[i] = ni[i];
    { // t = 0x224698fc0994a8dd8c46eb2100000001 in radix 2^29
        const unsigned __int128 t = ((unsigned __int128)0x224698fc0994a8ddull << 64) | 0x8c46eb2100000001ull;
        for (int i = 0; i < 5; ++i) K.t[i] = (u32)((t >> (29 * i)) & M29);
    }
    std::vector<u32> h(64, 0); u64 s = 12345; auto nx = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int row = 0; row < 4; ++row) for (int c = 0; c < 9; ++c) h[16 * row + c] = (u32)nx() & (c == 8 ? 0x3fffffu : M29);   // values below 2^254
    u32 *din, *dout, *dpr; hipMalloc(&din, 256); hipMalloc(&dout, 4 * 256); hipMalloc(&dpr, 512); hipMemcpy(din, h.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dpr); std::vector<u32> pr(128); hipMemcpy(pr.data(), dpr, 512, hipMemcpyDeviceToHost);
    printf("PROBE shr3:"); for (int i = 0; i < 20; ++i) printf(" %u", pr[i]); printf("\nPROBE shl9:"); for (int i = 0; i < 20; ++i) printf(" %u", pr[64 + i]); printf("\n");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms = 0;
    for (int rep = 0; rep < 3; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL(k_chain_row, dim3(1), dim3(64), 0, 0, din, dout, K, iters); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); }
    std::vector<u32> o(64); hipMemcpy(o.data(), dout, 256, hipMemcpyDeviceToHost);
    printf("ITERS %d\n", iters);
    for (int row = 0; row < 4; ++row) { printf("IN %d", row); for (int c = 0; c < 16; ++c) printf(" %x", h[16 * row + c]); printf("\n"); }
    for (int row = 0; row < 4; ++row) { printf("OUT %d", row); for (int c = 0; c < 16; ++c) printf(" %x", o[16 * row + c]); printf("\n"); }
    printf("{\"chain\": \"row-form product (16 lanes per product, 4 products per wave, DPP column sums, lazy carries, Montgomery by 2^261)\", \"iters\": %d, \"ns_per_op\": %.1f, \"nominal_cycles_per_op_at_2.4GHz\": %.0f}\n",
           iters, ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
    return 0;
}
