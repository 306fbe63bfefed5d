// half_exec.hip -- does a wave64 VALU instruction whose upper (or lower) 32 lanes are all inactive issue faster
// on gfx950's 32-wide SIMDs?  Same instruction stream with EXEC = all 64 lanes, lanes 0-31, lanes 32-63, 16 lanes.
// Build: hipcc -O3 --offload-arch=gfx950 half_exec.hip -o half_exec
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X

template <int OP>
__global__ __launch_bounds__(256) void k(int* out, int iters, int mode) {
    int a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b0 = a0 ^ 0x55, b1 = a1 ^ 0x33;
    const int lane = threadIdx.x & 63;
    const bool on = mode == 0 ? true : mode == 1 ? lane < 32 : mode == 2 ? lane >= 32 : lane < 16;
    if (on) {
        for (int it = 0; it < iters; ++it) {
            if (OP == 0) { REP8(asm volatile("v_max3_i32 %0, %0, %4, %5\n v_max3_i32 %1, %1, %4, %5\n v_max3_i32 %2, %2, %4, %5\n v_max3_i32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) }
            if (OP == 1) { REP8(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}

template <int OP>
void run(const char* name, int mode) {
    const int blocks = 256 * 4;        // 4 waves per SIMD
    int* d; hipMalloc(&d, 4 * 256 * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, mode);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 4000, mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const char* m[] = {"all 64 lanes", "lanes 0-31", "lanes 32-63", "lanes 0-15"};
    printf("%-12s EXEC = %-12s %.3f ms  (%.2f cycles @2.4 GHz per wave-instruction per SIMD)\n", name, m[mode], ms,
           ms * 1e-3 * 2.4e9 / (4000.0 * 32 * 4));
    hipFree(d);
}
int main() {
    for (int mode = 0; mode < 4; ++mode) run<0>("v_max3_i32", mode);
    for (int mode = 0; mode < 4; ++mode) run<1>("v_add_u32", mode);
    return 0;
}
