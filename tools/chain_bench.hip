// tools/chain_bench.hip — latency of a DEPENDENT chain of field products on a lone wave (the serial-sponge regime):
// radix 2^32 Montgomery product (fr_gfx950.inc) vs radix 2^29 product / square on nine-limb values (fr29.hpp).  Not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "fr29.hpp"
using namespace stark;
typedef PallasFr F;
#define N 2000
__global__ void __launch_bounds__(64) c32(const fr_t* in, fr_t* out) {
    fr_t x = in[threadIdx.x], c = in[64 + threadIdx.x];
    for (int i = 0; i < N; ++i) x = fr_mul<F>(x, c);
    out[blockIdx.x * 64 + threadIdx.x] = x;
}
__global__ void __launch_bounds__(64) c29mul(const fr_t* in, fr_t* out) {
    fr29_t x = fr29_unpack(in[threadIdx.x]), c = fr29_unpack(in[64 + threadIdx.x]);
    for (int i = 0; i < N; ++i) x = fr29_mul_mont<F>(x, c);
    out[blockIdx.x * 64 + threadIdx.x] = fr29_pack_reduce<F>(x.l);
}
__global__ void __launch_bounds__(64) c29sqr(const fr_t* in, fr_t* out) {
    fr29_t x = fr29_unpack(in[threadIdx.x]);
    for (int i = 0; i < N; ++i) x = fr29_sqr_mont<F>(x);
    out[blockIdx.x * 64 + threadIdx.x] = fr29_pack_reduce<F>(x.l);
}
int main() {
    std::vector<fr_t> h(128); uint64_t s = 99; auto nx = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (auto& c : h) { for (int i = 0; i < 8; ++i) c.v[i] = (uint32_t)nx(); c.v[7] &= 0x3fffffffu; }
    fr_t *din, *dout; hipMalloc(&din, 128 * 32); hipMalloc(&dout, 256 * 64 * 32); hipMemcpy(din, h.data(), 128 * 32, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, void (*k)(const fr_t*, fr_t*)) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(4), dim3(64), 0, 0, din, dout); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); }
        printf("{\"chain\": \"%s\", \"ns_per_op\": %.1f, \"nominal_cycles_per_op_at_2.4GHz\": %.0f}\n", name, ms * 1e6 / N, ms * 1e6 / N * 2.4);
    };
    run("radix32 fr_mul", c32); run("radix29 mul (9 limbs in/out)", c29mul); run("radix29 sqr (9 limbs in/out)", c29sqr);
    return 0;
}
