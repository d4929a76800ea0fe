// tools/bf_bench.hip — radix-2 DIT butterflies on register-resident points, two arithmetic forms (not product code):
//   A  eight 32-bit limbs, exact: p = b*w (fr_mul), a+p, a-p with conditional corrections (what ntt_dev.hpp does today)
//   B  nine 29-bit limbs, lazy:   p = mont29(w, b), a+p limb-wise, a-p+D limb-wise (D = 4r in borrow-proof form), carry pass every 3rd stage
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I stark_mlwe_amd/csrc tools/bf_bench.hip -o tools/bin/bf_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "fr.hpp"
#include "fr29.hpp"
using namespace stark;
typedef PallasFr F;
#define ITERS 64

__global__ void __launch_bounds__(256) k_form_a(const fr_t* in, const fr_t* tw, fr_t* out) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    fr_t x[8]; for (int i = 0; i < 8; ++i) x[i] = in[g * 8 + i];
    fr_t w = tw[threadIdx.x & 63];
    for (int it = 0; it < ITERS; ++it) {
#define BFA(i, j) { fr_t p = fr_mul<F>(x[j], w); fr_t a = x[i]; x[i] = fr_add<F>(a, p); x[j] = fr_sub<F>(a, p); }
        BFA(0, 1) BFA(2, 3) BFA(4, 5) BFA(6, 7) BFA(0, 2) BFA(1, 3) BFA(4, 6) BFA(5, 7) BFA(0, 4) BFA(1, 5) BFA(2, 6) BFA(3, 7)
    }
    for (int i = 0; i < 8; ++i) out[g * 8 + i] = x[i];
}
struct D29 { uint32_t l[9]; };
__device__ __forceinline__ void norm29(fr29_t& a) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { a.l[i + 1] += a.l[i] >> 29; a.l[i] &= FR_M29; }
}
__global__ void __launch_bounds__(256) k_form_b(const fr_t* in, const fr_t* tw, fr_t* out, D29 D) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    fr29_t x[8]; for (int i = 0; i < 8; ++i) x[i] = fr29_unpack(in[g * 8 + i]);
    fr29_t w = fr29_unpack(tw[threadIdx.x & 63]);
    for (int it = 0; it < ITERS; ++it) {
#define BFB(i, j, N) { if (N) { norm29(x[i]); norm29(x[j]); } fr29_t p = fr29_mul_mont<F>(w, x[j]); \
            _Pragma("unroll") for (int k = 0; k < 9; ++k) { const uint32_t a = x[i].l[k]; x[i].l[k] = a + p.l[k]; x[j].l[k] = a + D.l[k] - p.l[k]; } }
        BFB(0, 1, 1) BFB(2, 3, 1) BFB(4, 5, 1) BFB(6, 7, 1) BFB(0, 2, 0) BFB(1, 3, 0) BFB(4, 6, 0) BFB(5, 7, 0) BFB(0, 4, 0) BFB(1, 5, 0) BFB(2, 6, 0) BFB(3, 7, 0)
    }
    for (int i = 0; i < 8; ++i) { norm29(x[i]); out[g * 8 + i] = fr29_pack_reduce<F>(x[i].l); }
}
int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0); const int cus = p.multiProcessorCount;
    for (int wps : {2, 4}) {
        const int threads = 256, blocks = cus * wps * 8;      // wps waves per SIMD resident, 8 rounds of blocks
        const size_t n = (size_t)threads * blocks * 8;
        fr_t *in, *out, *tw; (void)hipMalloc(&in, n * 32); (void)hipMalloc(&out, n * 32); (void)hipMalloc(&tw, 64 * 32);
        std::vector<uint32_t> h(n * 8); uint32_t s = 12345; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s; }
        for (size_t i = 0; i < n; ++i) h[i * 8 + 7] &= 0x1fffffffu;
        (void)hipMemcpy(in, h.data(), n * 32, hipMemcpyHostToDevice); (void)hipMemcpy(tw, h.data(), 64 * 32, hipMemcpyHostToDevice);
        D29 D{}; { // 4r, limbs lifted by 2^29 with the matching borrow from the limb above
            uint32_t c[9]; fr_t r4; uint64_t cy = 0; for (int i = 0; i < 8; ++i) { uint64_t v = (uint64_t)F::P(i) * 4 + cy; r4.v[i] = (uint32_t)v; cy = v >> 32; }
            fr29_t u = fr29_unpack(r4); for (int i = 0; i < 9; ++i) c[i] = u.l[i]; c[8] += (uint32_t)cy << 24;   // bits 256.. of 4r land in limb 8 (bit 256 = 232 + 24)
            for (int i = 0; i < 9; ++i) D.l[i] = c[i] + (i < 8 ? (1u << 29) : 0u) - (i > 0 ? 1u : 0u);
        }
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int form = 0; form < 2; ++form) {
            float best = 1e9f;
            for (int r = 0; r < 3; ++r) {
                (void)hipEventRecord(e0);
                if (form == 0) hipLaunchKernelGGL(k_form_a, dim3(blocks), dim3(threads), 0, 0, in, tw, out);
                else hipLaunchKernelGGL(k_form_b, dim3(blocks), dim3(threads), 0, 0, in, tw, out, D);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (r > 0 && ms < best) best = ms;
            }
            const double bfs = (double)threads * blocks * ITERS * 12;
            printf("{\"form\": \"%s\", \"blocks_per_cu\": %d, \"ms\": %.3f, \"G_butterflies_per_s\": %.2f, \"simd_cycles_per_wave_butterfly_at_2.4GHz\": %.0f}\n",
                   form == 0 ? "A 8x32 exact" : "B 9x29 lazy", wps * 8, best, bfs / best / 1e6, best * 1e-3 * 2.4e9 * cus * 4 / (bfs / 64));
        }
        (void)hipFree(in); (void)hipFree(out); (void)hipFree(tw);
    }
    return 0;
}
