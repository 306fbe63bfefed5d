// cell16.hip -- does the 16-bit form of phase 1's carried cell pay?  Issue time of the cell's instruction mix
// (4 rows chained as in the kernel) in its 32-bit form (sdwa add, max3_i32, add, max_i32, max_i32) and with
// the two 2-operand maxima as v_max_u16; and what v_max_u16 / v_max3_u16 leave in the upper half.
// Build: hipcc -O3 --offload-arch=gfx950 cell16.hip -o cell16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(X) X X X X X X X X

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, int seed) {
    int d0 = 30000 + threadIdx.x, d1 = d0 + 3, d2 = d0 + 5, d3 = d0 + 7, x = d0 - 9;
    int h0 = d0 - 4, h1 = d0 - 6, h2 = d0 - 2, h3 = d0 - 8, prof = 0x0BFFFF0B + seed, go = -7;
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) { REP8(asm volatile(
            "v_add_u32_sdwa %0, %1, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_max3_i32 %1, %0, %4, %5\n v_add_u32 %0, %1, %10\n v_max_i32 %4, %0, %4\n v_max_i32 %5, %0, %5\n"
            "v_add_u32_sdwa %0, %2, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_max3_i32 %2, %0, %4, %6\n v_add_u32 %0, %2, %10\n v_max_i32 %4, %0, %4\n v_max_i32 %6, %0, %6\n"
            "v_add_u32_sdwa %0, %3, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_max3_i32 %3, %0, %4, %7\n v_add_u32 %0, %3, %10\n v_max_i32 %4, %0, %4\n v_max_i32 %7, %0, %7\n"
            "v_add_u32_sdwa %0, %1, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n v_max3_i32 %1, %0, %4, %8\n v_add_u32 %0, %1, %10\n v_max_i32 %4, %0, %4\n v_max_i32 %8, %0, %8"
            : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(x), "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(prof), "v"(go));) }
        if (OP == 1) { REP8(asm volatile(
            "v_add_u32_sdwa %0, %1, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_max3_i32 %1, %0, %4, %5\n v_add_u32 %0, %1, %10\n v_max_u16 %4, %0, %4\n v_max_u16 %5, %0, %5\n"
            "v_add_u32_sdwa %0, %2, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_max3_i32 %2, %0, %4, %6\n v_add_u32 %0, %2, %10\n v_max_u16 %4, %0, %4\n v_max_u16 %6, %0, %6\n"
            "v_add_u32_sdwa %0, %3, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_max3_i32 %3, %0, %4, %7\n v_add_u32 %0, %3, %10\n v_max_u16 %4, %0, %4\n v_max_u16 %7, %0, %7\n"
            "v_add_u32_sdwa %0, %1, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n v_max3_i32 %1, %0, %4, %8\n v_add_u32 %0, %1, %10\n v_max_u16 %4, %0, %4\n v_max_u16 %8, %0, %8"
            : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(x), "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(prof), "v"(go));) }
        // all-16-bit: add_u16 for the constant add too
        if (OP == 2) { REP8(asm volatile(
            "v_add_u32_sdwa %0, %1, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_max3_i32 %1, %0, %4, %5\n v_add_u16 %0, %1, %10\n v_max_u16 %4, %0, %4\n v_max_u16 %5, %0, %5\n"
            "v_add_u32_sdwa %0, %2, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_max3_i32 %2, %0, %4, %6\n v_add_u16 %0, %2, %10\n v_max_u16 %4, %0, %4\n v_max_u16 %6, %0, %6\n"
            "v_add_u32_sdwa %0, %3, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_max3_i32 %3, %0, %4, %7\n v_add_u16 %0, %3, %10\n v_max_u16 %4, %0, %4\n v_max_u16 %7, %0, %7\n"
            "v_add_u32_sdwa %0, %1, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n v_max3_i32 %1, %0, %4, %8\n v_add_u16 %0, %1, %10\n v_max_u16 %4, %0, %4\n v_max_u16 %8, %0, %8"
            : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(x), "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(prof), "v"(go));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (unsigned)(d0 + d1 + d2 + d3 + x + h0 + h1 + h2 + h3);
}

__global__ void sem(unsigned* out) {
    unsigned a = 0xAAAA1234u, b = 0xBBBB2345u, c = 0xCCCC0007u, r0, r1, r2;
    asm volatile("v_max_u16 %0, %1, %2" : "=v"(r0) : "v"(a), "v"(b));
    asm volatile("v_max3_u16 %0, %1, %2, %3" : "=v"(r1) : "v"(a), "v"(b), "v"(c));
    asm volatile("v_add_u16 %0, %1, %2" : "=v"(r2) : "v"(a), "v"(b));
    out[0] = r0; out[1] = r1; out[2] = r2;
}

template <int OP>
void run(const char* name, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd;            // 256-thread blocks: 4 waves = one per SIMD
    unsigned* d; hipMalloc(&d, 4 * 256 * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ninstr = (double)iters * 8 * 20 * waves_per_simd;     // per SIMD
    printf("%-44s %d w/SIMD: %.3f ms, %.3f ns per wave-instruction per SIMD, %.2f ns per cell\n", name, waves_per_simd, ms,
           ms * 1e6 / ninstr, ms * 1e6 / ninstr * 5);
    hipFree(d);
}

int main() {
    unsigned* d; hipMalloc(&d, 64); hipLaunchKernelGGL(sem, dim3(1), dim3(64), 0, 0, d); unsigned h[3];
    hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
    printf("v_max_u16(0xAAAA1234, 0xBBBB2345) = %08x   v_max3_u16(.., 0xCCCC0007) = %08x   v_add_u16 = %08x\n", h[0], h[1], h[2]);
    for (int w : {3, 4}) {
        run<0>("32-bit cell (sdwa, max3, add, max, max)", w);
        run<1>("two maxima as v_max_u16", w);
        run<2>("v_max_u16 + v_add_u16", w);
    }
    return 0;
}
