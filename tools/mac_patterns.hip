// tools/mac_patterns.hip — which MAC+carry instruction pattern issues best on gfx950 (not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 2048
#define DECL uint32_t a = seed + threadIdx.x, b = seed | 1; \
    uint64_t d0 = a, d1 = a + 1, d2 = a + 2, d3 = a + 3, d4 = a + 4, d5 = a + 5, d6 = a + 6, d7 = a + 7; \
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
#define FIN out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(d0 ^ d1 ^ d2 ^ d3 ^ d4 ^ d5 ^ d6 ^ d7) ^ c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
#define OPS : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b)
// V1: dependent pairs through vcc
__global__ void v1(uint32_t* out, uint32_t seed) { DECL
    for (int i = 0; i < ITERS; ++i) asm volatile(
        "v_mad_u64_u32 %0, vcc, %16, %17, %0\n v_addc_co_u32 %8, vcc, 0, %8, vcc\n v_mad_u64_u32 %1, vcc, %16, %17, %1\n v_addc_co_u32 %9, vcc, 0, %9, vcc\n"
        "v_mad_u64_u32 %2, vcc, %16, %17, %2\n v_addc_co_u32 %10, vcc, 0, %10, vcc\n v_mad_u64_u32 %3, vcc, %16, %17, %3\n v_addc_co_u32 %11, vcc, 0, %11, vcc\n"
        "v_mad_u64_u32 %4, vcc, %16, %17, %4\n v_addc_co_u32 %12, vcc, 0, %12, vcc\n v_mad_u64_u32 %5, vcc, %16, %17, %5\n v_addc_co_u32 %13, vcc, 0, %13, vcc\n"
        "v_mad_u64_u32 %6, vcc, %16, %17, %6\n v_addc_co_u32 %14, vcc, 0, %14, vcc\n v_mad_u64_u32 %7, vcc, %16, %17, %7\n v_addc_co_u32 %15, vcc, 0, %15, vcc\n"
        OPS : "vcc");
    FIN }
// V2: 8 mads with distinct SGPR carries, then 8 VOP3 addc reading them
__global__ void v2(uint32_t* out, uint32_t seed) { DECL
    for (int i = 0; i < ITERS; ++i) asm volatile(
        "v_mad_u64_u32 %0, s[40:41], %16, %17, %0\n v_mad_u64_u32 %1, s[42:43], %16, %17, %1\n v_mad_u64_u32 %2, s[44:45], %16, %17, %2\n v_mad_u64_u32 %3, s[46:47], %16, %17, %3\n"
        "v_mad_u64_u32 %4, s[48:49], %16, %17, %4\n v_mad_u64_u32 %5, s[50:51], %16, %17, %5\n v_mad_u64_u32 %6, s[52:53], %16, %17, %6\n v_mad_u64_u32 %7, s[54:55], %16, %17, %7\n"
        "v_addc_co_u32_e64 %8, vcc, 0, %8, s[40:41]\n v_addc_co_u32_e64 %9, vcc, 0, %9, s[42:43]\n v_addc_co_u32_e64 %10, vcc, 0, %10, s[44:45]\n v_addc_co_u32_e64 %11, vcc, 0, %11, s[46:47]\n"
        "v_addc_co_u32_e64 %12, vcc, 0, %12, s[48:49]\n v_addc_co_u32_e64 %13, vcc, 0, %13, s[50:51]\n v_addc_co_u32_e64 %14, vcc, 0, %14, s[52:53]\n v_addc_co_u32_e64 %15, vcc, 0, %15, s[54:55]\n"
        OPS : "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55");
    FIN }
// V3: software-pipelined by one: mad k+1 (carry to sgpr) between mad k and its addc; alternate vcc / sgpr
__global__ void v3(uint32_t* out, uint32_t seed) { DECL
    for (int i = 0; i < ITERS; ++i) asm volatile(
        "v_mad_u64_u32 %0, s[40:41], %16, %17, %0\n v_mad_u64_u32 %1, s[42:43], %16, %17, %1\n v_addc_co_u32_e64 %8, vcc, 0, %8, s[40:41]\n v_mad_u64_u32 %2, s[40:41], %16, %17, %2\n v_addc_co_u32_e64 %9, vcc, 0, %9, s[42:43]\n"
        "v_mad_u64_u32 %3, s[42:43], %16, %17, %3\n v_addc_co_u32_e64 %10, vcc, 0, %10, s[40:41]\n v_mad_u64_u32 %4, s[40:41], %16, %17, %4\n v_addc_co_u32_e64 %11, vcc, 0, %11, s[42:43]\n"
        "v_mad_u64_u32 %5, s[42:43], %16, %17, %5\n v_addc_co_u32_e64 %12, vcc, 0, %12, s[40:41]\n v_mad_u64_u32 %6, s[40:41], %16, %17, %6\n v_addc_co_u32_e64 %13, vcc, 0, %13, s[42:43]\n"
        "v_mad_u64_u32 %7, s[42:43], %16, %17, %7\n v_addc_co_u32_e64 %14, vcc, 0, %14, s[40:41]\n v_addc_co_u32_e64 %15, vcc, 0, %15, s[42:43]\n"
        OPS : "vcc", "s40", "s41", "s42", "s43");
    FIN }
// V4: mads only (no carry handling) for reference;  V5: 8 independent VOP3 addc only
__global__ void v4(uint32_t* out, uint32_t seed) { DECL
    for (int i = 0; i < ITERS; ++i) asm volatile(
        "v_mad_u64_u32 %0, s[40:41], %16, %17, %0\n v_mad_u64_u32 %1, s[42:43], %16, %17, %1\n v_mad_u64_u32 %2, s[44:45], %16, %17, %2\n v_mad_u64_u32 %3, s[46:47], %16, %17, %3\n"
        "v_mad_u64_u32 %4, s[48:49], %16, %17, %4\n v_mad_u64_u32 %5, s[50:51], %16, %17, %5\n v_mad_u64_u32 %6, s[52:53], %16, %17, %6\n v_mad_u64_u32 %7, s[54:55], %16, %17, %7\n"
        OPS : "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55");
    FIN }
__global__ void v5(uint32_t* out, uint32_t seed) { DECL
    asm volatile("s_mov_b64 s[40:41], exec" ::: "s40", "s41");
    for (int i = 0; i < ITERS; ++i) asm volatile(
        "v_addc_co_u32_e64 %8, s[42:43], 0, %8, s[40:41]\n v_addc_co_u32_e64 %9, s[44:45], 0, %9, s[40:41]\n v_addc_co_u32_e64 %10, s[46:47], 0, %10, s[40:41]\n v_addc_co_u32_e64 %11, s[48:49], 0, %11, s[40:41]\n"
        "v_addc_co_u32_e64 %12, s[50:51], 0, %12, s[40:41]\n v_addc_co_u32_e64 %13, s[52:53], 0, %13, s[40:41]\n v_addc_co_u32_e64 %14, s[54:55], 0, %14, s[40:41]\n v_addc_co_u32_e64 %15, s[56:57], 0, %15, s[40:41]\n"
        OPS : "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57");
    FIN }
typedef void (*kern_t)(uint32_t*, uint32_t);
static void run(const char* name, kern_t k, int n_instr, int w) {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); int cus = p.multiProcessorCount, threads = 256 * w;
    uint32_t* out; hipMalloc(&out, (size_t)threads * cus * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(cus), dim3(threads), 0, 0, out, 12345u); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(cus), dim3(threads), 0, 0, out, 12345u); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double per_simd = (double)ITERS * (threads / 64) * cus / (cus * 4.0);
    printf("{\"pattern\": \"%s\", \"waves_per_simd\": %d, \"nominal_cycles_per_group_at_2.4GHz\": %.1f, \"instrs_per_group\": %d}\n", name, w, ms * 1e-3 * 2.4e9 / per_simd, n_instr);
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4}) {
        run("v1 8x(mad->addc via vcc)", v1, 16, w); run("v2 8 mads(sgpr carries) then 8 vop3 addc", v2, 16, w);
        run("v3 pipelined by one, vop3 addc", v3, 16, w); run("v4 8 mads only", v4, 8, w); run("v5 8 vop3 addc only", v5, 8, w);
    }
    return 0;
}
