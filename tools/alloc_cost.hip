// tools/alloc_cost.hip — cost of hipMalloc / hipFree for the buffer sizes one FRI build allocates (not product code).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main() {
    hipFree(0);
    for (size_t mb : {1, 16, 256, 1024}) {
        double ta = 0, tf = 0; const int reps = 5;
        for (int r = 0; r < reps; ++r) {
            void* p; auto t0 = std::chrono::steady_clock::now(); hipMalloc(&p, mb << 20); auto t1 = std::chrono::steady_clock::now();
            hipMemset(p, 0, mb << 20); hipDeviceSynchronize();
            auto t2 = std::chrono::steady_clock::now(); hipFree(p); auto t3 = std::chrono::steady_clock::now();
            ta += std::chrono::duration<double, std::milli>(t1 - t0).count(); tf += std::chrono::duration<double, std::milli>(t3 - t2).count();
        }
        printf("{\"MiB\": %zu, \"hipMalloc_ms\": %.3f, \"hipFree_ms\": %.3f}\n", mb, ta / reps, tf / reps);
    }
    return 0;
}
