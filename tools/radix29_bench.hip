// tools/radix29_bench.hip — dot-product throughput: radix 2^32 (mad+addc, fr_gfx950.inc) vs radix 2^29 (carry-free, fr29.hpp).
// Not product code.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Istark_mlwe_amd/csrc tools/radix29_bench.hip -o tools/bin/radix29_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "fr29.hpp"
using namespace stark;
typedef PallasFr F;
#define TERMS 6
#define ITERS 300

__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2))) k32(const fr_t* __restrict__ cst, const fr_t* __restrict__ in, fr_t* __restrict__ out) {
    extern __shared__ uint4 lds[];
    const size_t i = (size_t)blockIdx.x * 128 + threadIdx.x;
    fr_t x[TERMS];
    for (int k = 0; k < TERMS; ++k) x[k] = in[i * TERMS + k];
    for (int it = 0; it < ITERS; ++it) {
        fr_wide w; fr_wide_zero(w);
#pragma unroll
        for (int k = 0; k < TERMS; ++k) fr_wide_mac_f<F>(w, cst[(it % 8) * TERMS + k], x[k]);
        fr_t r = fr_wide_reduce<F>(w);
#pragma unroll
        for (int k = 0; k < TERMS - 1; ++k) x[k] = x[k + 1];
        x[TERMS - 1] = r;
    }
    out[i] = x[TERMS - 1];
    if (threadIdx.x == 9999) lds[0] = make_uint4(0, 0, 0, 0);
}
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2))) k29(const uint32_t* __restrict__ cst, const fr_t* __restrict__ in, fr_t* __restrict__ out) {
    extern __shared__ uint4 lds[];
    const size_t i = (size_t)blockIdx.x * 128 + threadIdx.x;
    fr_t x[TERMS];
    for (int k = 0; k < TERMS; ++k) x[k] = in[i * TERMS + k];
    for (int it = 0; it < ITERS; ++it) {
        fr_wide29 w; fr_wide29_zero(w);
#pragma unroll
        for (int k = 0; k < TERMS; ++k) fr_wide29_mac(w, cst + ((it % 8) * TERMS + k) * 9, fr29_unpack(x[k]));
        fr_t r = fr_wide29_reduce<F>(w);
#pragma unroll
        for (int k = 0; k < TERMS - 1; ++k) x[k] = x[k + 1];
        x[TERMS - 1] = r;
    }
    out[i] = x[TERMS - 1];
    if (threadIdx.x == 9999) lds[0] = make_uint4(0, 0, 0, 0);
}
int main() {
    const int blocks = 256 * 4 * 8, n = blocks * 128;
    std::vector<fr_t> hc(8 * TERMS), hin((size_t)n * TERMS); std::vector<uint32_t> hc29(8 * TERMS * 9);
    uint64_t s = 12345;
    auto nx = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (auto& c : hc) { for (int i = 0; i < 8; ++i) c.v[i] = (uint32_t)nx(); c.v[7] &= 0x3fffffffu; }
    for (auto& c : hin) { for (int i = 0; i < 8; ++i) c.v[i] = (uint32_t)nx(); c.v[7] &= 0x3fffffffu; }
    for (size_t k = 0; k < hc.size(); ++k) fr29_const_from<F>(hc[k], &hc29[k * 9]);
    fr_t *dc, *din, *o1, *o2; uint32_t* dc29;
    hipMalloc(&dc, hc.size() * 32); hipMalloc(&din, hin.size() * 32); hipMalloc(&o1, (size_t)n * 32); hipMalloc(&o2, (size_t)n * 32); hipMalloc(&dc29, hc29.size() * 4);
    hipMemcpy(dc, hc.data(), hc.size() * 32, hipMemcpyHostToDevice); hipMemcpy(din, hin.data(), hin.size() * 32, hipMemcpyHostToDevice); hipMemcpy(dc29, hc29.data(), hc29.size() * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k32, hipFuncAttributeMaxDynamicSharedMemorySize, 40960); hipFuncSetAttribute((const void*)k29, hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms32 = 0, ms29 = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(k32, dim3(blocks), dim3(128), 40960, 0, dc, din, o1); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms32, e0, e1);
        hipEventRecord(e0); hipLaunchKernelGGL(k29, dim3(blocks), dim3(128), 40960, 0, dc29, din, o2); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms29, e0, e1);
    }
    std::vector<fr_t> a(n), b(n); hipMemcpy(a.data(), o1, (size_t)n * 32, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o2, (size_t)n * 32, hipMemcpyDeviceToHost);
    size_t bad = 0; for (int i = 0; i < n; ++i) if (!fr_eq(a[i], b[i])) ++bad;
    const double dots = (double)n * ITERS;
    printf("{\"bench\": \"6-term dot + reduce\", \"radix32_ms\": %.3f, \"radix29_ms\": %.3f, \"speedup\": %.3f, \"radix32_ns_per_dot_per_cu\": %.2f, \"radix29_ns_per_dot_per_cu\": %.2f, \"mismatches\": %zu}\n",
           ms32, ms29, ms32 / ms29, ms32 * 1e6 / dots * 256, ms29 * 1e6 / dots * 256, bad);
    return bad != 0;
}
