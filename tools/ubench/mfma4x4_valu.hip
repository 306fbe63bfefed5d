// mfma4x4_valu.hip -- can a wave hide gate-math-like VALU work inside its own stream of 4x4x1 f32 MFMAs?
// Eight waves on one CU (two per SIMD), each iteration = 298 MFMAs on two independent accumulator chains
// (two groups of four lines sharing the weights) + 96 VALU instructions (fma / exp / rcp mix).  Variants:
// MFMAs only; VALU only; MFMAs then VALU; VALU woven into the MFMA stream (one VALU after every third MFMA).
//   hipcc -O3 --offload-arch=gfx950 mfma4x4_valu.hip -o mfma4x4_valu
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MF(acc) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:4 abid:3" : "+v"(acc) : "v"(a), "v"(b));
#define VA1 asm volatile("v_fma_f32 %0, %0, %2, %3\n" : "+v"(v0), "+v"(v1) : "v"(a), "v"(b));
#define VA2 asm volatile("v_exp_f32 %1, %1\n" : "+v"(v0), "+v"(v1) : "v"(a), "v"(b));
#define VA3 asm volatile("v_fma_f32 %0, %0, %2, %3\n" : "+v"(v2), "+v"(v3) : "v"(a), "v"(b));
#define VA4 asm volatile("v_rcp_f32 %1, %1\n" : "+v"(v2), "+v"(v3) : "v"(a), "v"(b));
#define REP2(X) X X
#define REP4(X) REP2(X) REP2(X)
#define REP8(X) REP4(X) REP4(X)
#define REP16(X) REP8(X) REP8(X)
#define REP32(X) REP16(X) REP16(X)

template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long* out, int iters) {
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0;
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 1e-6f;
    float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3;
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { REP32(MF(acc0) MF(acc1) MF(acc0) MF(acc1) MF(acc0) MF(acc1) MF(acc0) MF(acc1) MF(acc0)) MF(acc1) }          // 289 MFMAs
        if (MODE == 1) { REP32(VA1 VA2 VA3) }                                                                                               // 96 VALU
        if (MODE == 2) { REP32(MF(acc0) MF(acc1) MF(acc0) MF(acc1) MF(acc0) MF(acc1) MF(acc0) MF(acc1) MF(acc0)) MF(acc1) REP32(VA1 VA2 VA3) }
        if (MODE == 3) { REP32(MF(acc0) MF(acc1) MF(acc0) VA1 MF(acc1) MF(acc0) MF(acc1) VA2 MF(acc0) MF(acc1) MF(acc0) VA3) MF(acc1) }
        if (MODE == 4) { REP32(MF(acc0) MF(acc1) MF(acc0) VA1 MF(acc1) MF(acc0) MF(acc1) VA4 MF(acc0) MF(acc1) MF(acc0) VA3) MF(acc1) }
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if ((threadIdx.x & 63) == 0) { out[threadIdx.x >> 6] = t1 - t0; out[8 + (threadIdx.x >> 6)] = (unsigned long long)(acc0[0] + acc1[1] + v0 + v1 + v2 + v3); }
}

template <int MODE>
void run(const char* name) {
    unsigned long long* d;
    (void)hipMalloc(&d, 16 * 8);
    const int iters = 500;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(512), 0, 0, d, iters); (void)hipDeviceSynchronize(); }
    unsigned long long h[16];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s %8.0f cycles per iteration (wave 0), %8.0f (wave 7)\n", name, (double)h[0] / iters, (double)h[7] / iters);
}

int main() {
    run<0>("289 MFMAs, two chains");
    run<1>("96 VALU (fma, exp, fma)");
    run<2>("289 MFMAs, then 96 VALU");
    run<3>("96 VALU woven into the MFMAs (fma/exp/fma)");
    run<4>("96 VALU woven into the MFMAs (fma/rcp/fma)");
    return 0;
}
