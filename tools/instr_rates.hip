// tools/instr_rates.hip — measures VALU issue rates that size the Fr multiplier on gfx950 (not product code).
// Build: hipcc --offload-arch=gfx950 -O3 tools/instr_rates.hip -o gpurun_out/instr_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITERS 4096
// each kernel: 8 independent dependency chains per lane, ITERS iterations, 8 instrs per iteration
#define K8(NAME, BODY)                                                                             \
    __global__ void NAME(uint32_t* out, uint32_t seed) {                                           \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;       \
        uint32_t a4 = a0 * 9 + 4, a5 = a0 * 11 + 5, a6 = a0 * 13 + 6, a7 = a0 * 15 + 7;            \
        uint64_t d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;           \
        double f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7;             \
        uint32_t b = seed | 1;                                                                     \
        for (int i = 0; i < ITERS; ++i) { BODY }                                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(d0 ^ d1 ^ d2 ^ d3 ^ d4 ^ d5 ^ d6 ^ d7) ^ (uint32_t)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7); \
    }
#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

#define MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d##i) : "v"(a##i), "v"(b) : "vcc");
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define ADD32(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define ADDCO(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, 0, %2, vcc" : "+v"(a##i), "+v"(b) , "+v"(a##i) : : "vcc");
#define ADDC1(i) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a##i) : "v"(b) : "vcc");
#define LSHLADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(d##i) : "v"(d7));
#define MOV32(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a##i) : "v"(b));
#define MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a##i) : "v"(b));
#define MULHI24(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(f##i) : "v"(f7));
#define ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a##i) : "v"(b));
#define MADMIX(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc" : "+v"(d##i), "+v"(a##i) : "v"(a##i), "v"(b) : "vcc");

K8(k_mad64, R8(MAD64))
K8(k_mullo, R8(MULLO))
K8(k_mulhi, R8(MULHI))
K8(k_add32, R8(ADD32))
K8(k_addc, R8(ADDC1))
K8(k_lshladd64, R8(LSHLADD64))
K8(k_mov, R8(MOV32))
K8(k_mad24, R8(MAD24))
K8(k_mulhi24, R8(MULHI24))
K8(k_fma64, R8(FMA64))
K8(k_add3, R8(ADD3))
__global__ void k_madaddc(uint32_t* out, uint32_t seed) {   // mad + addc pairs (16 instrs / iteration)
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 9 + 4, a5 = a0 * 11 + 5, a6 = a0 * 13 + 6, a7 = a0 * 15 + 7;
    uint64_t d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7; uint32_t b = seed | 1;
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
#define MA(i) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(d##i), "+v"(c##i) : "v"(a##i), "v"(b) : "vcc");
    for (int i = 0; i < ITERS; ++i) { R8(MA) }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(d0 ^ d1 ^ d2 ^ d3 ^ d4 ^ d5 ^ d6 ^ d7) ^ c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
}

typedef void (*kern_t)(uint32_t*, uint32_t);
static void run(const char* name, kern_t k, int instr_per_iter, int waves_per_simd) {
    int cus = 256; hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); cus = p.multiProcessorCount;
    int threads = 64 * 4 * waves_per_simd, blocks = cus;      // one block per CU, waves spread over the 4 SIMDs
    uint32_t* out; hipMalloc(&out, (size_t)threads * blocks * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, 12345u); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, 12345u); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instrs = (double)ITERS * instr_per_iter * (threads / 64) * blocks;
    double per_simd_per_s = wave_instrs / (ms * 1e-3) / (cus * 4);
    printf("{\"instr\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"cycles_per_wave_instr_per_simd_at_2.4GHz\": %.3f, \"G_lane_ops_per_s\": %.1f}\n",
           name, waves_per_simd, ms, 2.4e9 / per_simd_per_s, wave_instrs * 64 / (ms * 1e-3) / 1e9);
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4}) {
        run("v_mad_u64_u32", k_mad64, 8, w); run("v_mul_lo_u32", k_mullo, 8, w); run("v_mul_hi_u32", k_mulhi, 8, w);
        run("v_add_u32", k_add32, 8, w); run("v_addc_co_u32", k_addc, 8, w); run("v_lshl_add_u64", k_lshladd64, 8, w); run("v_mov_b32", k_mov, 8, w);
        run("v_mad_u32_u24", k_mad24, 8, w); run("v_mul_hi_u32_u24", k_mulhi24, 8, w); run("v_fma_f64", k_fma64, 8, w); run("v_add3_u32", k_add3, 8, w);
        run("mad64+addc pair (2 instrs)", k_madaddc, 16, w);
    }
    return 0;
}
