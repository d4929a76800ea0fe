// tools/mix_rates.hip — marginal cost of non-MAC VALU instructions inside a v_mad_u64_u32-dominated stream on gfx950 (not product code).
// Each iteration: 16 MACs into 8 accumulators + K "other" instructions of one kind; the slope of time over K is what an extra
// instruction of that kind costs a kernel that is otherwise multiplying.  Build: hipcc --offload-arch=gfx950 -O3 tools/mix_rates.hip -o gpurun_out/mix_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 2048
#define MAC(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d##i) : "v"(a##i), "v"(b) : "vcc");
#define MAC16 MAC(0) MAC(1) MAC(2) MAC(3) MAC(4) MAC(5) MAC(6) MAC(7) MAC(0) MAC(1) MAC(2) MAC(3) MAC(4) MAC(5) MAC(6) MAC(7)
#define O_LSHLADD(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(e##i) : "v"(e7));
#define O_SHR64(i) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(e##i));
#define O_ALIGN(i) asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(a##i) : "v"(b));
#define O_ANDLIT(i) asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(a##i));
#define O_ANDREG(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a##i) : "v"(b));
#define O_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a##i) : "v"(b));
#define O_ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define O_CNDM(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(b));
#define R4(O) O(0) O(1) O(2) O(3)
#define R8(O) R4(O) O(4) O(5) O(6) O(7)
#define R16(O) R8(O) R8(O)
#define KERN(NAME, OTHERS)                                                                                          \
    __global__ void NAME(uint32_t* out, uint32_t seed) {                                                             \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 9 + 4, a5 = a0 * 11 + 5, a6 = a0 * 13 + 6, a7 = a0 * 15 + 7; \
        uint64_t d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;                             \
        uint64_t e0 = a0, e1 = a1, e2 = a2, e3 = a3, e4 = a4, e5 = a5, e6 = a6, e7 = a7; uint32_t b = seed | 1;      \
        for (int i = 0; i < ITERS; ++i) { MAC16 OTHERS }                                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(d0 ^ d1 ^ d2 ^ d3 ^ d4 ^ d5 ^ d6 ^ d7) ^ (uint32_t)(e0 ^ e1 ^ e2 ^ e3 ^ e4 ^ e5 ^ e6 ^ e7); \
    }
KERN(k_base, )
KERN(k_lshladd4, R4(O_LSHLADD)) KERN(k_lshladd8, R8(O_LSHLADD)) KERN(k_lshladd16, R16(O_LSHLADD))
KERN(k_shr4, R4(O_SHR64)) KERN(k_shr8, R8(O_SHR64)) KERN(k_shr16, R16(O_SHR64))
KERN(k_align4, R4(O_ALIGN)) KERN(k_align8, R8(O_ALIGN)) KERN(k_align16, R16(O_ALIGN))
KERN(k_andlit4, R4(O_ANDLIT)) KERN(k_andlit8, R8(O_ANDLIT)) KERN(k_andlit16, R16(O_ANDLIT))
KERN(k_andreg4, R4(O_ANDREG)) KERN(k_andreg8, R8(O_ANDREG)) KERN(k_andreg16, R16(O_ANDREG))
KERN(k_mov4, R4(O_MOV)) KERN(k_mov8, R8(O_MOV)) KERN(k_mov16, R16(O_MOV))
KERN(k_add4, R4(O_ADD)) KERN(k_add8, R8(O_ADD)) KERN(k_add16, R16(O_ADD))
KERN(k_cnd4, R4(O_CNDM)) KERN(k_cnd8, R8(O_CNDM)) KERN(k_cnd16, R16(O_CNDM))
KERN(k_mac4, MAC(0) MAC(1) MAC(2) MAC(3)) KERN(k_mac8, MAC(0) MAC(1) MAC(2) MAC(3) MAC(4) MAC(5) MAC(6) MAC(7)) KERN(k_mac16, MAC16)

typedef void (*kern_t)(uint32_t*, uint32_t);
static double run(kern_t k, int waves_per_simd) {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); int cus = p.multiProcessorCount;
    int threads = 64 * 4 * waves_per_simd, blocks = cus;
    uint32_t* out; hipMalloc(&out, (size_t)threads * blocks * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, 12345u); hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, 12345u); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    hipFree(out);
    return best * 1e-3 / ITERS / waves_per_simd * 2.4e9;      // nominal SIMD cycles per wave-iteration at 2.4 GHz
}
int main() {
    struct { const char* name; kern_t k[3]; } kinds[] = {
        {"v_lshl_add_u64", {k_lshladd4, k_lshladd8, k_lshladd16}}, {"v_lshrrev_b64", {k_shr4, k_shr8, k_shr16}}, {"v_alignbit_b32", {k_align4, k_align8, k_align16}},
        {"v_and_b32 literal", {k_andlit4, k_andlit8, k_andlit16}}, {"v_and_b32 reg", {k_andreg4, k_andreg8, k_andreg16}}, {"v_mov_b32", {k_mov4, k_mov8, k_mov16}},
        {"v_add_u32", {k_add4, k_add8, k_add16}}, {"v_cndmask_b32", {k_cnd4, k_cnd8, k_cnd16}}, {"v_mad_u64_u32", {k_mac4, k_mac8, k_mac16}}};
    for (int w : {1, 2, 4}) {
        double base = run(k_base, w);
        printf("{\"waves_per_simd\": %d, \"base_16_macs_cycles\": %.1f}\n", w, base);
        for (auto& kd : kinds) {
            double c4 = run(kd.k[0], w), c8 = run(kd.k[1], w), c16 = run(kd.k[2], w);
            printf("{\"waves_per_simd\": %d, \"other\": \"%s\", \"cycles_16mac_plus_4\": %.1f, \"plus_8\": %.1f, \"plus_16\": %.1f, \"marginal_cycles_per_instr\": %.2f}\n",
                   w, kd.name, c4, c8, c16, (c16 - base) / 16.0);
        }
    }
    return 0;
}
