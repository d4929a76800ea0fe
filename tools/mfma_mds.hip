// tools/mfma_mds.hip — PROTOTYPE, not product code: y = M * x (dense 17 x 17 over Pallas Fr, Montgomery form) for 64 sponges per wave pair
// END TO END on the matrix cores: signed radix-256 recoding of the state in LDS, v_mfma_i32_32x32x32_i8 against stored Toeplitz fragments
// of the constants, folding of the digit-column sums into 29-bit columns, the two-lane exchange (v_permlane32_swap), signed carry pass,
// Montgomery step by 2^261, canonical store.  The result is checked against sum_e M[i][e] * x_e computed with the product's portable field code.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I stark_mlwe_amd/csrc tools/mfma_mds.hip -o tools/bin/mfma_mds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include "fr.hpp"
#include "fr29.hpp"
using namespace stark;
typedef PallasFr F;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr int T = 17;

// x (canonical, below r) -> signed radix-256 digits in place of the bytes: add 0x80 to every byte with carries, flip every byte's top bit
__device__ __forceinline__ fr_t recode_signed(const fr_t& x) {
    fr_t y; uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const uint64_t s = (uint64_t)x.v[i] + 0x80808080u + c; y.v[i] = (uint32_t)s ^ 0x80808080u; c = s >> 32; }
    return y;
}
// After the half exchange a lane holds all 64 digit sums of ITS sponge: lo[rt][reg] = rows (reg&3) + 8(reg>>2) of tile rt (the lower lane's
// rows), hi[rt][reg] = the same + 4 (the upper lane's rows); row = digit position c - 32 rt.  Pairs of adjacent digits go into the 64-bit
// column of weight 2^(29k) that holds the lower one (shift < 29, pair below 2^33: no overflow).
__device__ __forceinline__ void fold_rows(int64_t* col, const v16i& lo, const v16i& hi, int rt) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const v16i& a = hh ? hi : lo;
                const int64_t pair = (int64_t)a[4 * q + 2 * p] + (int64_t)a[4 * q + 2 * p + 1] * 256;
                const int c = 32 * rt + 8 * q + 4 * hh + 2 * p, k = (8 * c) / 29, sh = 8 * c - 29 * k;
                col[k] += pair << sh;
            }
}
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2))) k_mds(const v4i* __restrict__ Atab, const fr_t* __restrict__ X, fr_t* __restrict__ Y, int reps) {
    __shared__ uint4 st[T * 2 * 64];
    const int lane = threadIdx.x & 63, h = lane >> 5; const bool isY = threadIdx.x >= 64;
    const size_t b0 = (size_t)blockIdx.x * T * 64;
    for (int e = isY ? 1 : 0; e < T; e += 2) {       // stage the recoded state: slot (2e + half)[sponge]
        const fr_t x = recode_signed(X[b0 + (size_t)e * 64 + lane]);
        st[(2 * e) * 64 + lane] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]); st[(2 * e + 1) * 64 + lane] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
    }
    __syncthreads();
    v4i b[T][2];
#pragma unroll
    for (int e = 0; e < T; ++e)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) { const uint4 u = st[(2 * e + h) * 64 + 32 * ct + (lane & 31)]; b[e][ct] = v4i{(int)u.x, (int)u.y, (int)u.z, (int)u.w}; }
    __syncthreads();
    for (int rep = 0; rep < reps; ++rep)
#pragma unroll 1
    for (int i = isY ? 1 : 0; i < T; i += 2) {
        v16i acc[2][2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rt][ct][r] = 0;
        const v4i* Ai = Atab + (size_t)(i * 2) * T * 64 + lane;
#pragma unroll
        for (int e = 0; e < T; ++e) {
            const v4i a0 = Ai[(size_t)e * 64], a1 = Ai[(size_t)(T + e) * 64];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                acc[0][ct] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b[e][ct], acc[0][ct], 0, 0, 0);
                acc[1][ct] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b[e][ct], acc[1][ct], 0, 0, 0);
            }
        }
        // half exchange: lane l (< 32) owns sponge l = column tile 0, lane 32 + l owns sponge 32 + l = column tile 1.  swap(vdst = tile 0, src0 = tile 1)
        // gives the lower lane the upper lane's tile-0 rows and the upper lane the lower lane's tile-1 rows: afterwards r[0] = the lower rows, r[1] = the upper rows of the own sponge
        v16i lo[2], hi[2];
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
        fold_rows(col, lo[0], hi[0], 0); fold_rows(col, lo[1], hi[1], 1);
        fr_wide29 w;
        // signed carry pass: the total is a non-negative integer, so every limb ends in [0, 2^29) and the top column non-negative
#pragma unroll
        for (int k = 0; k < 17; ++k) { col[k + 1] += col[k] >> 29; w.c[k] = (uint64_t)col[k] & FR_M29; }
        w.c[17] = (uint64_t)col[17];
        const fr_t y = fr_wide29_reduce<F>(w);
        if (rep == 0) Y[b0 + (size_t)i * 64 + lane] = y;
    }
}

int main() {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0); const int cus = prop.multiProcessorCount;
    uint64_t s = 0x243f6a8885a308d3ull; auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    auto rand_fr = [&]() { fr_t x; for (int i = 0; i < 8; ++i) x.v[i] = rnd(); x.v[7] &= 0x3fffffffu; return x; };
    std::vector<fr_t> M((size_t)T * T); for (auto& m : M) m = rand_fr();
    M[0] = fr_zero<F>(); for (int i = 0; i < 8; ++i) M[1].v[i] = F::P(i); M[1].v[0] -= 1;       // edge constants: 0 and r - 1
    // Toeplitz fragments of the signed digits of 32 * M (Montgomery form): lane (row r = l & 31, k half kh = l >> 5), byte j <-> digit (32 rt + r) - (16 kh + j)
    std::vector<int8_t> A((size_t)T * 2 * T * 64 * 16, 0);
    for (int i = 0; i < T; ++i) for (int e = 0; e < T; ++e) {
        const fr_t c32 = fr_mul_portable<F>(M[(size_t)i * T + e], fr_from_u64<F>(32));
        uint8_t by[32]; memcpy(by, c32.v, 32); int8_t d[32]; int cy = 0;
        for (int b = 0; b < 32; ++b) { const int v = by[b] + 0x80 + cy; cy = v >> 8; d[b] = (int8_t)((v & 0xff) - 0x80); }
        if (cy) { fprintf(stderr, "constant digit overflow\n"); return 1; }
        for (int rt = 0; rt < 2; ++rt) for (int l = 0; l < 64; ++l) for (int j = 0; j < 16; ++j) {
            const int idx = (32 * rt + (l & 31)) - (16 * (l >> 5) + j);
            A[((((size_t)(i * 2 + rt) * T + e) * 64 + l) * 16) + j] = (idx >= 0 && idx < 32) ? d[idx] : 0;
        }
    }
    const int batches = cus * 8;
    std::vector<fr_t> X((size_t)batches * T * 64); for (auto& x : X) x = rand_fr();
    for (int i = 0; i < 8; ++i) X[5].v[i] = F::P(i); X[5].v[0] -= 1; X[7] = fr_zero<F>();     // r - 1 and 0 among the inputs
    v4i* dA; fr_t *dX, *dY; (void)hipMalloc(&dA, A.size()); (void)hipMalloc(&dX, X.size() * 32); (void)hipMalloc(&dY, X.size() * 32);
    (void)hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); (void)hipMemcpy(dX, X.data(), X.size() * 32, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mds, dim3(batches), dim3(128), 0, 0, dA, dX, dY, 1);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
    std::vector<fr_t> Y(X.size()); (void)hipMemcpy(Y.data(), dY, Y.size() * 32, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int bt = 0; bt < batches; bt += 97) for (int n = 0; n < 64; ++n) for (int i = 0; i < T; ++i) {
        fr_t acc = fr_zero<F>();
        for (int e = 0; e < T; ++e) acc = fr_add<F>(acc, fr_mul_portable<F>(M[(size_t)i * T + e], X[((size_t)bt * T + e) * 64 + n]));
        if (!fr_eq(acc, Y[((size_t)bt * T + i) * 64 + n])) { if (bad < 5) fprintf(stderr, "mismatch batch %d sponge %d row %d\n", bt, n, i); ++bad; }
    }
    printf("{\"check\": \"y = M x (mod r) for all 64 sponges of sampled batches, incl. constants 0 and r-1, inputs 0 and r-1\", \"mismatches\": %ld}\n", bad);
    if (bad) return 1;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); const int reps = 8; float best = 1e9f;
    for (int r = 0; r < 3; ++r) { (void)hipEventRecord(e0); hipLaunchKernelGGL(k_mds, dim3(batches), dim3(128), 0, 0, dA, dX, dY, reps); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    const double products = (double)batches * reps;
    printf("{\"kernel\": \"dense 17x17 product of 64 sponges END TO END (recode, MFMA, fold, exchange, Montgomery step, canonical), wave pair\", \"ms\": %.3f, \"us_per_product_per_cu\": %.2f, "
           "\"simd_cycles_per_product_at_2.4GHz\": %.0f, \"valu_form_simd_cycles_per_product\": \"~180000\"}\n", best, best * 1e3 / (products / cus), best * 1e-3 * 2.4e9 * cus * 4 / products);
    return 0;
}
