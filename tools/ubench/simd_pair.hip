// simd_pair.hip -- how the work of the TWO waves of a SIMD composes in the float64 four-line recurrence (csrc/ta_lstm_f64.hip,
// lstm_seq4_f64_kernel): per step every wave issues NM v_mfma_f64_4x4x4_4b_f64 on three accumulator chains, then NV
// float64 VALU instructions (the cell update's class: fma chains), then the workgroup's barrier.  Eight waves (two per
// SIMD); the younger wave of a SIMD (waves 4..7) may carry EXTRA MFMAs.  Printed: cycles per step (s_memtime) and the end
// of each wave's MFMA phase, for MFMAs only, VALU only, both, and both with the older wave's VALU moved behind a wait for
// its partner.  Build: hipcc -O3 --offload-arch=gfx950 simd_pair.hip -o simd_pair
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define MF3 acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc0, 0, 0, 0); \
            acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1, 0, 0, 0); \
            acc2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc2, 0, 0, 0);
#define VA5 asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n v_fma_f64 %0, %0, %5, %4" \
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a), "v"(b));

// nm3: groups of three MFMAs per step for the older waves; extra3: more groups for waves 4..7; nv5: groups of five VALU
// instructions per step; mode bit 0: MFMAs, bit 1: VALU, bit 2: the older wave waits for its partner before its VALU
__global__ __launch_bounds__(512) void step_kernel(unsigned long long* out, int steps, int nm3, int extra3, int nv5, int mode) {
    __shared__ unsigned done[8];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 8) done[threadIdx.x] = 0;
    double acc0 = 0, acc1 = 0, acc2 = 0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    double v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3;
    const int my3 = nm3 + (wave >= 4 ? extra3 : 0);
    __syncthreads();
    unsigned long long t0, tm = 0, t1, msum = 0;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int s = 0; s < steps; ++s) {
        unsigned long long q0, q1;
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(q0) :: "memory");
        if (mode & 1)
            for (int i = 0; i < my3; ++i) { MF3 }
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(q1) : "v"(acc0), "v"(acc1), "v"(acc2) : "memory");
        msum += q1 - q0;
        if (mode & 4) {
            if (lane == 0) __hip_atomic_store(&done[wave], (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            for (int spin = 0; spin < 4096; ++spin) {
                if (__hip_atomic_load(&done[wave ^ 4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= (unsigned)(s + 1)) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (mode & 2)
            for (int i = 0; i < nv5; ++i) { VA5 }
        b = b * 0.999 + v0 * 1e-9 + acc0 * 1e-12;                    // the next step depends on this one
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    (void)tm;
    if (lane == 0) { out[wave * 2] = t1 - t0; out[wave * 2 + 1] = msum; }
    if (acc0 + acc1 + acc2 + v0 + v1 + v2 + v3 == 12345.678) out[20] = 1;
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 8 * 32);
    const int steps = 2000;
    auto run = [&](int nm3, int extra3, int nv5, int mode, const char* what) {
        hipLaunchKernelGGL(step_kernel, dim3(1), dim3(512), 0, 0, d, 50, nm3, extra3, nv5, mode);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(step_kernel, dim3(1), dim3(512), 0, 0, d, steps, nm3, extra3, nv5, mode);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(32);
        hipMemcpy(h.data(), d, 8 * 32, hipMemcpyDeviceToHost);
        printf("%-66s step %7.0f cycles;  MFMA phase: wave 0 %6.0f, wave 4 %6.0f, wave 3 %6.0f, wave 7 %6.0f\n", what,
               (double)h[0] / steps, (double)h[1] / steps, (double)h[9] / steps, (double)h[7] / steps, (double)h[15] / steps);
    };
    run(25, 0, 0, 1, "75 + 75 MFMAs per SIMD, no VALU");
    run(25, 3, 0, 1, "75 + 84 MFMAs, no VALU");
    run(25, 0, 27, 2, "no MFMAs, 135 + 135 f64 VALU");
    run(25, 0, 27, 3, "75 + 75 MFMAs, 135 + 135 VALU");
    run(25, 3, 27, 3, "75 + 84 MFMAs, 135 + 135 VALU");
    run(25, 3, 27, 7, "75 + 84 MFMAs, 135 + 135 VALU, older wave waits for its partner");
    run(25, 3, 20, 3, "75 + 84 MFMAs, 100 + 100 VALU");
    run(50, 6, 27, 3, "150 + 168 MFMAs (one wave's worth doubled), 135 + 135 VALU");
    hipFree(d);
    return 0;
}
