// lone_wave.hip -- what ONE wave alone on its SIMD gets: issue interval of independent and of dependent
// VALU instructions, and the shader clock the chip holds when only a few waves run (s_memtime ticks per
// s_memrealtime tick of 10 ns).  Build: hipcc -O3 --offload-arch=gfx950 lone_wave.hip -o lone_wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X

template <int OP>
__global__ __launch_bounds__(64) void k(unsigned long long* out, int iters) {
    int a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b0 = a0 ^ 0x55, b1 = a1 ^ 0x33;
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) { REP8(asm volatile("v_max3_i32 %0, %0, %4, %5\n v_max3_i32 %1, %1, %4, %5\n v_max3_i32 %2, %2, %4, %5\n v_max3_i32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) }
        if (OP == 1) { REP8(asm volatile("v_max3_i32 %0, %0, %4, %5\n v_max3_i32 %0, %0, %4, %5\n v_max3_i32 %0, %0, %4, %5\n v_max3_i32 %0, %0, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) }
        if (OP == 2) { REP8(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
        if (OP == 3) { REP8(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %0, %0, %4\n v_add_u32 %0, %0, %4\n v_add_u32 %0, %0, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
        // the carried tagged cell's chain: and_or -> max3 -> add -> max, dependent
        if (OP == 4) { REP8(asm volatile("v_and_or_b32 %0, %0, %4, 21\n v_max3_i32 %0, %0, %4, %5\n v_add_u32 %0, %0, %4\n v_max_i32 %0, %0, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) }
        // DPP after a VALU write of its source (needs 2 wait states)
        if (OP == 5) { REP8(asm volatile("v_add_u32 %0, %0, %4\n s_nop 1\n v_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32 %0, %1, %4\n s_nop 1\n v_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) }
    }
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = t1 - t0; out[blockIdx.x * 4 + 1] = r1 - r0;
        out[blockIdx.x * 4 + 2] = (unsigned long long)(a0 + a1 + a2 + a3);
    }
}

template <int OP>
void run(const char* name, int blocks, int per_iter) {
    unsigned long long* d;
    hipMalloc(&d, sizeof(unsigned long long) * 4 * blocks);
    const int iters = 20000;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, 100);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(4 * blocks);
    hipMemcpy(h.data(), d, sizeof(unsigned long long) * 4 * blocks, hipMemcpyDeviceToHost);
    const double cyc = (double)h[0], ns = (double)h[1] * 10.0, n = (double)iters * per_iter;
    printf("%-34s %5d waves: %6.2f cycles / instruction, %6.2f ns / instruction, clock %.0f MHz\n", name, blocks, cyc / n,
           ns / n, cyc / ns * 1e3);
    hipFree(d);
}

int main() {
    for (int blocks : {1, 16, 64, 1024, 8192}) {
        run<0>("v_max3_i32 independent x4", blocks, 32);
        run<1>("v_max3_i32 dependent", blocks, 32);
        run<2>("v_add_u32 independent x4", blocks, 32);
        run<3>("v_add_u32 dependent", blocks, 32);
        run<4>("and_or/max3/add/max chain", blocks, 32);
        run<5>("add + s_nop 1 + dpp (3 per pair)", blocks, 48);
    }
    return 0;
}
