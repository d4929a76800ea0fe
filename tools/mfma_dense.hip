// tools/mfma_dense.hip — PROTOTYPE, not product code: the dense t x t Poseidon matrix-vector product over a 255-bit field as an
// int8 matrix product on the MFMA units (v_mfma_i32_16x16x64_i8), for 64 sponges at a time.
//
//   y_i = sum_e M[i][e] * x_e        (t = 17; M constant, x the state)
// With both factors in SIGNED radix-256 digits (x = sum_b xd[b] 256^b, xd[b] in [-128, 127]; 32 digits each), the digit-column sums
//   S[(i,c)][n] = sum_{e,b} md[i][e][c-b] * xd[e][n][b]          c = 0..63, n = sponge
// are ONE integer matrix product  [17*64 rows] x [17*32] . [17*32] x [64 sponges]  with a Toeplitz left factor; |S| < 2^24, exact in
// the i32 accumulators, and y_i = sum_c S[(i,c)] 256^c (to be folded into the nine 29-bit-limb columns and Montgomery-reduced by the
// VALU).  68 row tiles x 4 column tiles x 9 K-steps = 2 448 MFMAs of 16 384 MACs per dense product of 64 sponges.
// This file measures the MFMA phase alone (A fragments streamed from global/L2, B fragments = the state, held in registers) and checks
// the digit sums and the reconstructed integers on the host.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_dense.hip -o tools/bin/mfma_dense
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int T = 17, ND = 32, NC = 64, RT = T * NC / 16, KS = 9, CT = 4;   // row tiles, K-steps (two elements each, the last half empty), column tiles

// one workgroup = 4 waves = one batch of 64 sponges; wave w takes row tiles w, w+4, ...
__global__ void __launch_bounds__(256) k_dense(const v4i* __restrict__ Atab, const v4i* __restrict__ Btab, int* __restrict__ S, int store_all, int reps) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const v4i* Bb = Btab + (size_t)blockIdx.x * KS * CT * 64;
    v4i b[KS][CT];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) b[ks][ct] = Bb[(ks * CT + ct) * 64 + lane];
    int check = 0;
    for (int rep = 0; rep < reps; ++rep) {
        for (int rt = w; rt < RT; rt += 4) {
            v4i acc[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[ct] = v4i{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const v4i a = Atab[((size_t)rt * KS + ks) * 64 + lane];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[ct] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b[ks][ct], acc[ct], 0, 0, 0);
            }
            if (store_all && rep == 0) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) *reinterpret_cast<v4i*>(S + ((((size_t)blockIdx.x * RT + rt) * CT + ct) * 64 + lane) * 4) = acc[ct];
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) check += acc[ct].x ^ acc[ct].y ^ acc[ct].z ^ acc[ct].w;
        }
    }
    if (!store_all && check == 0x7fffffff) S[0] = check;     // keeps the loop alive
}

// Variant W: the Toeplitz fragments are not stored — lane (row r, k-half hb) reads its 16 bytes as an UNALIGNED window of the reversed,
// zero-padded digit string of M[i][e] (96 bytes per constant: 27.7 KB for the whole matrix, L1-resident):
//   bytes j = 0..15  <->  digit index (c0 + r) - (16 hb + j),  string R[q] = digit[63 - q] for q in 32..63, zero elsewhere  =>  offset q0 = 63 - (c0 + r) + 16 hb
typedef int v4i_u __attribute__((ext_vector_type(4), aligned(4)));     // four byte-shifted copies of every string make each window DWORD-aligned
__global__ void __launch_bounds__(256) k_dense_w(const uint8_t* __restrict__ Rstr, const v4i* __restrict__ Btab, int* __restrict__ S, int store_all, int reps) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const v4i* Bb = Btab + (size_t)blockIdx.x * KS * CT * 64;
    v4i b[KS][CT];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) b[ks][ct] = Bb[(ks * CT + ct) * 64 + lane];
    const int r = lane & 15, g = lane >> 4, hb = g & 1, eo = g >> 1;
    int check = 0;
    for (int rep = 0; rep < reps; ++rep) {
        for (int rt = w; rt < RT; rt += 4) {
            const int i = rt >> 2, c0 = 16 * (rt & 3);
            const int q0 = 63 - (c0 + r) + 16 * hb, sh = q0 & 3;                 // copy `sh` holds the string shifted down by sh bytes
            const uint8_t* base = Rstr + (size_t)sh * (T * T * 96 + 64) + (size_t)i * T * 96 + (q0 - sh);
            v4i acc[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[ct] = v4i{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int e = 2 * ks + eo;
                v4i a = v4i{0, 0, 0, 0};
                if (e < T) a = *reinterpret_cast<const v4i_u*>(base + (size_t)e * 96);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[ct] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b[ks][ct], acc[ct], 0, 0, 0);
            }
            if (store_all && rep == 0) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) *reinterpret_cast<v4i*>(S + ((((size_t)blockIdx.x * RT + rt) * CT + ct) * 64 + lane) * 4) = acc[ct];
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) check += acc[ct].x ^ acc[ct].y ^ acc[ct].z ^ acc[ct].w;
        }
    }
    if (!store_all && check == 0x7fffffff) S[0] = check;
}

// ---- host big-integer helpers (little-endian 32-bit words) -------------------------------------------------------------------
struct U256 { uint32_t w[8]; };
static const uint32_t PALLAS_R[8] = {0x00000001u, 0x8c46eb21u, 0x0994a8ddu, 0x224698fcu, 0x00000000u, 0x00000000u, 0x00000000u, 0x40000000u};
static uint64_t rng_s = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return (uint32_t)(rng_s >> 16); }
static U256 rand_fr() { U256 x; for (int i = 0; i < 8; ++i) x.w[i] = rnd(); x.w[7] &= 0x3fffffffu; return x; }     // below 2^254 < r
static void signed_digits(const U256& x, int8_t d[ND]) {      // x = sum d[b] 256^b, d in [-128, 127]: add 0x80 to every byte (with carries), flip each byte's top bit
    uint8_t by[32]; memcpy(by, x.w, 32);
    int carry = 0;
    for (int b = 0; b < 32; ++b) { int v = by[b] + 0x80 + carry; carry = v >> 8; d[b] = (int8_t)((v & 0xff) - 0x80); }
    if (carry) { fprintf(stderr, "digit overflow\n"); exit(1); }
}

int main() {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0); const int cus = prop.multiProcessorCount;
    // constants and their Toeplitz fragments
    std::vector<int8_t> md((size_t)T * T * ND);
    std::vector<U256> M((size_t)T * T);
    for (int i = 0; i < T * T; ++i) { M[i] = rand_fr(); signed_digits(M[i], &md[(size_t)i * ND]); }
    std::vector<int8_t> A((size_t)RT * KS * 64 * 16, 0);
    for (int rt = 0; rt < RT; ++rt) { const int i = rt / 4, c0 = 16 * (rt % 4);
        for (int ks = 0; ks < KS; ++ks) for (int l = 0; l < 64; ++l) { const int r = l & 15, g = l >> 4, e = 2 * ks + (g >> 1), hb = g & 1;
            for (int j = 0; j < 16; ++j) { const int idx = (c0 + r) - (16 * hb + j); A[(((size_t)rt * KS + ks) * 64 + l) * 16 + j] = (e < T && idx >= 0 && idx < ND) ? md[((size_t)i * T + e) * ND + idx] : 0; } } }
    const size_t rcopy = (size_t)T * T * 96 + 64;
    std::vector<uint8_t> R(4 * rcopy, 0);
    for (int sh = 0; sh < 4; ++sh) for (int ie = 0; ie < T * T; ++ie) for (int q = 32; q < 64; ++q) R[sh * rcopy + (size_t)ie * 96 + q - sh] = (uint8_t)md[(size_t)ie * ND + (63 - q)];
    uint8_t* dR; (void)hipMalloc(&dR, R.size()); (void)hipMemcpy(dR, R.data(), R.size(), hipMemcpyHostToDevice);
    const int verify_batches = 2;
    for (int pass = 0; pass < 4; ++pass) {
        const bool windows = pass >= 2; if (windows) pass -= 2;
        const int batches = pass == 0 ? verify_batches : cus * 8;
        std::vector<U256> X((size_t)batches * T * 64); std::vector<int8_t> xd((size_t)batches * T * 64 * ND);
        for (size_t i = 0; i < X.size(); ++i) { X[i] = rand_fr(); signed_digits(X[i], &xd[i * ND]); }     // X[(batch*T + e)*64 + n]
        std::vector<int8_t> B((size_t)batches * KS * CT * 64 * 16, 0);
        for (int bt = 0; bt < batches; ++bt) for (int ks = 0; ks < KS; ++ks) for (int ct = 0; ct < CT; ++ct) for (int l = 0; l < 64; ++l) {
            const int n = 16 * ct + (l & 15), g = l >> 4, e = 2 * ks + (g >> 1), hb = g & 1;
            for (int j = 0; j < 16; ++j) B[((((size_t)bt * KS + ks) * CT + ct) * 64 + l) * 16 + j] = e < T ? xd[(((size_t)bt * T + e) * 64 + n) * ND + 16 * hb + j] : 0; }
        v4i *dA, *dB; int* dS; const size_t sbytes = (size_t)batches * RT * CT * 64 * 4 * 4;
        (void)hipMalloc(&dA, A.size()); (void)hipMalloc(&dB, B.size()); (void)hipMalloc(&dS, pass == 0 ? sbytes : 64);
        (void)hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
#define LAUNCH(SA, REPS) do { if (windows) hipLaunchKernelGGL(k_dense_w, dim3(batches), dim3(256), 0, 0, dR, dB, dS, SA, REPS); else hipLaunchKernelGGL(k_dense, dim3(batches), dim3(256), 0, 0, dA, dB, dS, SA, REPS); } while (0)
        if (pass == 0) {
            LAUNCH(1, 1); (void)hipDeviceSynchronize();
            std::vector<int> S(sbytes / 4); (void)hipMemcpy(S.data(), dS, sbytes, hipMemcpyDeviceToHost);
            // (1) digit sums against the definition, (2) sum_c S 256^c against the schoolbook integer sum_e M[i][e] * X[e] (mod 2^512, words)
            long bad = 0;
            for (int bt = 0; bt < batches; ++bt) for (int i = 0; i < T; ++i) for (int n = 0; n < 64; n += 7) {
                int64_t want[NC];
                for (int c = 0; c < NC; ++c) { int64_t s = 0; for (int e = 0; e < T; ++e) for (int b = 0; b < ND; ++b) { const int idx = c - b; if (idx >= 0 && idx < ND) s += (int64_t)md[((size_t)i * T + e) * ND + idx] * xd[(((size_t)bt * T + e) * 64 + n) * ND + b]; } want[c] = s; }
                for (int c = 0; c < NC; ++c) { const int rt = i * 4 + c / 16, r = c % 16, ct = n / 16, l = (n & 15) + 16 * (r / 4), reg = r % 4;
                    const int got = S[((((size_t)bt * RT + rt) * CT + ct) * 64 + l) * 4 + reg]; if (got != want[c]) { if (bad < 5) fprintf(stderr, "mismatch i=%d n=%d c=%d got %d want %lld\n", i, n, c, got, (long long)want[c]); ++bad; } }
                // integer check: fold the signed digit sums into 16 words with signed carries, compare with the schoolbook product sum
                uint32_t acc[17] = {0};
                for (int e = 0; e < T; ++e) { const U256& a = M[(size_t)i * T + e]; const U256& x = X[((size_t)bt * T + e) * 64 + n];
                    for (int p = 0; p < 8; ++p) { uint64_t cy = 0; for (int q = 0; q < 8; ++q) { uint64_t v = (uint64_t)a.w[p] * x.w[q] + acc[p + q] + cy; acc[p + q] = (uint32_t)v; cy = v >> 32; } for (int k = p + 8; cy && k < 17; ++k) { uint64_t v = (uint64_t)acc[k] + cy; acc[k] = (uint32_t)v; cy = v >> 32; } } }
                uint8_t fold[68] = {0}; int64_t cy = 0;
                for (int c = 0; c < 68; ++c) { int64_t v = (c < NC ? want[c] : 0) + cy; fold[c] = (uint8_t)(v & 0xff); cy = v >> 8; }
                if (memcmp(fold, acc, 68) != 0) { if (bad < 5) fprintf(stderr, "integer mismatch i=%d n=%d\n", i, n); ++bad; }
            }
            printf("{\"check\": \"digit sums and reconstructed integers, %d batches, sampled sponges\", \"a_operand\": \"%s\", \"mismatches\": %ld}\n", batches, windows ? "dword-aligned windows of digit strings" : "stored fragments", bad);
            if (bad) return 1;
        } else {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            const int reps = 8;
            LAUNCH(0, 1); (void)hipDeviceSynchronize();
            float best = 1e9f;
            for (int r = 0; r < 3; ++r) { (void)hipEventRecord(e0); LAUNCH(0, reps); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
            const double products = (double)batches * reps, macs = products * RT * CT * KS * 16384.0;
            printf("{\"kernel\": \"dense 17x17 product of 64 sponges, MFMA phase\", \"a_operand\": \"%s\", \"batches\": %d, \"reps\": %d, \"ms\": %.3f, \"us_per_product_per_cu\": %.2f, \"int8_TMAC_per_s\": %.1f, "
                   "\"simd_cycles_per_product_at_2.4GHz\": %.0f, \"valu_form_simd_cycles_per_product\": \"~180000 (289 terms x 98 instr + 34 reductions x 190 instr, 5.2 cycles each)\"}\n",
                   windows ? "dword-aligned windows of digit strings (4 shifted copies, 111 KB)" : "stored fragments (578 KB, L2)", batches, reps, best, best * 1e3 / (products / cus), macs / (best * 1e-3) / 1e12, best * 1e-3 * 2.4e9 * cus * 4 / products);
        }
        (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dS);
        if (windows) pass += 2;
    }
    return 0;
}
