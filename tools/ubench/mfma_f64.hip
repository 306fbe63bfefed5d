// mfma_f64.hip -- v_mfma_f64_16x16x4_f64 on gfx950, for the float64 recurrence mode of csrc/ta_lstm_f64.hip:
//  (1) operand / result layout (checked against a host product, both candidate D layouts),
//  (2) cycles per MFMA: independent accumulators and one dependent chain, one and two waves per SIMD,
//  (3) v_fma_f64 issue interval alone and beside another wave's f64 MFMAs on the same SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_f64.hip -o mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

typedef double f64x4 __attribute__((ext_vector_type(4)));
#define REP8(X) X X X X X X X X

__global__ void layout_kernel(const double* A, const double* B, double* D) {
    const int lane = threadIdx.x;
    // hypothesis shared with the f32 16x16x4 form: A[i][k] in lane i + 16 k, B[k][j] in lane j + 16 k
    const double a = A[(lane & 15) * 4 + (lane >> 4)];
    const double b = B[(lane >> 4) * 16 + (lane & 15)];
    f64x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[lane * 4 + r] = acc[r];
}

// mode bit 0: waves 0..3 issue MFMAs (chain = 1: one accumulator); bit 1: waves 4..7 issue v_fma_f64
__global__ __launch_bounds__(512) void rate_kernel(unsigned long long* out, int iters, int mode, int chain, int nw) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool do_mfma = (mode & 1) && (nw == 8 ? true : wave < 4) && !((mode & 2) && wave >= 4);
    const bool do_valu = (mode & 2) && wave >= 4;
    f64x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    double a = threadIdx.x * 0.001, b = 1.0 + threadIdx.x * 1e-6;
    double v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3;
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (do_mfma) {
        if (chain) {
            for (int it = 0; it < iters; ++it) {
                REP8(acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
                     acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
                     acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
                     acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);)
            }
        } else {
            for (int it = 0; it < iters; ++it) {
                REP8(acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
                     acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
                     acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
                     acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);)
            }
        }
    }
    if (do_valu) {
        for (int it = 0; it < iters; ++it) {
            REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                              : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a), "v"(b));)
        }
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if ((threadIdx.x & 63) == 0) {
        out[wave * 2] = t1 - t0;
        out[wave * 2 + 1] = (unsigned long long)(acc0[0] + acc1[1] + acc2[2] + acc3[3] + v0 + v1 + v2 + v3);
    }
}

int main() {
    // (1) layout
    std::vector<double> A(64), B(64), D(256), ref(256, 0.0);
    for (int i = 0; i < 64; ++i) { A[i] = std::sin(1.0 + i) ; B[i] = std::cos(0.5 * i + 2.0); }
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s = std::fma(A[i * 4 + k], B[k * 16 + j], s);
            ref[i * 16 + j] = s;
        }
    double *dA, *dB, *dD;
    hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dD, 256 * 8);
    hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost);
    int okA = 0, okB = 0, exactA = 0, exactB = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
            const double v = D[lane * 4 + r];
            const double hA = ref[(4 * (lane >> 4) + r) * 16 + (lane & 15)];     // i = 4 (lane / 16) + r  (the f32 form's)
            const double hB = ref[(4 * r + (lane >> 4)) * 16 + (lane & 15)];     // i = 4 r + lane / 16
            okA += std::fabs(v - hA) < 1e-12; okB += std::fabs(v - hB) < 1e-12;
            exactA += v == hA; exactB += v == hB;
        }
    printf("layout: D[i][j] with j = lane %% 16;  i = 4 (lane / 16) + r: %d / 256 (bit-equal to the fma chain k = 0..3: %d);  "
           "i = 4 r + lane / 16: %d / 256 (bit-equal %d)\n", okA, exactA, okB, exactB);

    // (2), (3) rates
    unsigned long long* d;
    hipMalloc(&d, sizeof(unsigned long long) * 16);
    const int iters = 2000;
    auto run = [&](int mode, int chain, int nw, const char* what) {
        hipLaunchKernelGGL(rate_kernel, dim3(1), dim3(512), 0, 0, d, 20, mode, chain, nw);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(rate_kernel, dim3(1), dim3(512), 0, 0, d, iters, mode, chain, nw);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(16);
        hipMemcpy(h.data(), d, sizeof(unsigned long long) * 16, hipMemcpyDeviceToHost);
        printf("%-58s wave 0: %8.2f cycles / MFMA   wave 4: %8.2f cycles / %s\n", what,
               (mode & 1) ? (double)h[0] / (iters * 32.0) : 0.0,
               (double)h[8] / (iters * 32.0), (mode & 2) ? "v_fma_f64" : "MFMA");
    };
    run(1, 0, 4, "MFMA, 4 accumulators, one wave per SIMD");
    run(1, 1, 4, "MFMA, one dependent chain, one wave per SIMD");
    run(1, 0, 8, "MFMA, 4 accumulators, two waves per SIMD");
    run(1, 1, 8, "MFMA, one dependent chain, two waves per SIMD");
    run(2, 0, 4, "v_fma_f64 alone (waves 4..7)");
    run(3, 0, 4, "MFMA (waves 0..3) beside v_fma_f64 (waves 4..7)");
    hipFree(d);
    return 0;
}
