// mfma_valu.hip -- does VALU / transcendental work of one wave run beside the MFMAs of ANOTHER wave on the
// same SIMD?  One workgroup of 8 waves on one CU: waves 0..3 (one per SIMD) issue MFMAs back to back,
// waves 4..7 (their SIMD neighbours) issue gate-math-like VALU (v_fma_f32, v_exp_f32, v_rcp_f32).
// Each role is timed alone and together, for the f32-input MFMA the exact recurrence kernel uses and for
// the bf16 MFMA of the split-operand kernel.  Build: hipcc -O3 --offload-arch=gfx950 mfma_valu.hip -o mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define REP8(X) X X X X X X X X

// mode bit 0: waves 0..3 run MFMAs; bit 1: waves 4..7 run VALU.  KIND 0: f32 16x16x4, 1: bf16 16x16x32
template <int KIND>
__global__ __launch_bounds__(512) void k(unsigned long long* out, int iters, int mode) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool do_mfma = wave < 4 && (mode & 1), do_valu = wave >= 4 && (mode & 2);
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 1e-6f;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(a + i); bb[i] = (__bf16)(b - i); }
    float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3;
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (do_mfma) {
        for (int it = 0; it < iters; ++it) {
            if (KIND == 0) {
                REP8(acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
                     acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc1, 0, 0, 0);
                     acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc2, 0, 0, 0);
                     acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc3, 0, 0, 0);)
            } else {
                REP8(acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc0, 0, 0, 0);
                     acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc1, 0, 0, 0);
                     acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc2, 0, 0, 0);
                     acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc3, 0, 0, 0);)
            }
        }
    }
    if (do_valu) {
        for (int it = 0; it < iters; ++it) {
            // per group: 4 fma, 2 exp, 2 rcp on independent registers (the mix of a sigmoid / tanh)
            REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_exp_f32 %1, %1\n v_fma_f32 %2, %2, %4, %5\n v_rcp_f32 %3, %3\n"
                              "v_fma_f32 %0, %0, %4, %5\n v_exp_f32 %1, %1\n v_fma_f32 %2, %2, %4, %5\n v_rcp_f32 %3, %3"
                              : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a), "v"(b));)
        }
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if ((threadIdx.x & 63) == 0) {
        out[wave * 2] = t1 - t0;
        out[wave * 2 + 1] = (unsigned long long)(acc0[0] + acc1[1] + acc2[2] + acc3[3] + v0 + v1 + v2 + v3);
    }
}

template <int KIND>
void run(const char* name) {
    unsigned long long* d;
    hipMalloc(&d, sizeof(unsigned long long) * 16);
    const int iters = 4000;
    for (int mode : {1, 2, 3}) {
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(512), 0, 0, d, 50, mode);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(512), 0, 0, d, iters, mode);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(16);
        hipMemcpy(h.data(), d, sizeof(unsigned long long) * 16, hipMemcpyDeviceToHost);
        // s_memtime ticks at 100 MHz on gfx950?  report raw ticks per instruction and the ratio between modes
        printf("%-22s mode %s: MFMA wave %8.3f ticks / MFMA, VALU wave %8.3f ticks / instruction\n", name,
               mode == 1 ? "MFMA alone " : mode == 2 ? "VALU alone " : "both       ",
               (mode & 1) ? (double)h[0] / (iters * 32.0) : 0.0, (mode & 2) ? (double)h[8] / (iters * 64.0) : 0.0);
    }
    hipFree(d);
}

int main() {
    run<0>("v_mfma_f32_16x16x4_f32");
    run<1>("v_mfma_f32_16x16x32_bf16");
    return 0;
}
