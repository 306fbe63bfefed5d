// mfma_f64_4x4.hip -- v_mfma_f64_4x4x4_4b_f64 on gfx950 (four 4 x 4 x 4 blocks per instruction), for the float64
// recurrence on groups of FOUR lines (csrc/ta_lstm_f64.hip, lstm_seq4_f64_kernel):
//  (1) operand / result layout, found by one-hot operands (no hypothesis needed) and printed as bit fields,
//  (2) bit-equality of a block's 4-term product with the host's fma chain k = 0..3 (what the 16 x 16 x 4 form gives),
//  (3) cycles per MFMA: independent accumulators and one dependent chain, one and two waves per SIMD,
//  (4) VALU work (v_fma_f64) beside another wave's 4 x 4 x 4 MFMAs on the same SIMD, and inside the MFMA wave's own stream,
//  (5) v_permlane16_swap / v_permlane32_swap semantics (the 4 x 4 transpose across the four 16-lane rows).
// Build: hipcc -O3 --offload-arch=gfx950 mfma_f64_4x4.hip -o mfma_f64_4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

#define REP8(X) X X X X X X X X

__global__ void onehot_kernel(unsigned long long* masks) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            const unsigned long long m = __ballot(d != 0.0);
            if (lane == 0) masks[la * 64 + lb] = m;
        }
}

__global__ void product_kernel(const double* A, const double* B, const double* C, double* D) {
    const int lane = threadIdx.x;
    D[lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[lane], B[lane], C[lane], 0, 0, 0);
}

__global__ void swap_kernel(unsigned* out) {
    const unsigned lane = threadIdx.x;
    unsigned v0 = 1000 + lane, v1 = 2000 + lane;
    unsigned w0 = v0, w1 = v1;
    // (through the builtins: after a VALU write of an operand the instruction needs wait states, which the compiler
    // provides; the same instruction in an asm statement right behind the adds read stale registers)
    auto r16 = __builtin_amdgcn_permlane16_swap(v0, v1, false, false);
    v0 = r16[0]; v1 = r16[1];
    auto r32 = __builtin_amdgcn_permlane32_swap(w0, w1, false, false);
    w0 = r32[0]; w1 = r32[1];
    out[lane] = v0; out[64 + lane] = v1; out[128 + lane] = w0; out[192 + lane] = w1;
}

// mode bit 0: MFMA waves (waves 0..3, or all 8 when nw == 8 and bit 1 clear); bit 1: waves 4..7 issue v_fma_f64;
// chain: 0 = four accumulators round-robin, 1 = one dependent chain, 3 = three accumulators;
// weave: one v_fma_f64 between every two MFMAs of the MFMA waves themselves
__global__ __launch_bounds__(512) void rate_kernel(unsigned long long* out, int iters, int mode, int chain, int nw, int weave) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // mode bit 2: roles swapped -- the OLDER waves 0..3 issue the VALU work, waves 4..7 the MFMAs
    const bool swapped = mode & 4;
    const bool do_valu = (mode & 2) && (swapped ? wave < 4 : wave >= 4);
    const bool do_mfma = (mode & 1) && !do_valu && (nw == 8 || (swapped ? wave >= 4 : wave < 4));
    double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    double a = threadIdx.x * 0.001, b = 1.0 + threadIdx.x * 1e-6;
    double v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3;
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (do_mfma) {
        if (weave) {
            for (int it = 0; it < iters; ++it) {
                REP8(asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %4, %5, %0\n v_fma_f64 %6, %6, %4, %5\n"
                                  "v_mfma_f64_4x4x4_4b_f64 %1, %4, %5, %1\n v_fma_f64 %7, %7, %4, %5\n"
                                  "v_mfma_f64_4x4x4_4b_f64 %2, %4, %5, %2\n v_fma_f64 %6, %6, %4, %5\n"
                                  "v_mfma_f64_4x4x4_4b_f64 %3, %4, %5, %3\n v_fma_f64 %7, %7, %4, %5"
                                  : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(a), "v"(b), "v"(v0), "v"(v1));)
            }
        } else if (chain == 1) {
            for (int it = 0; it < iters; ++it) {
                REP8(acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc0, 0, 0, 0);
                     acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc0, 0, 0, 0);
                     acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc0, 0, 0, 0);
                     acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc0, 0, 0, 0);)
            }
        } else if (chain == 3) {
            for (int it = 0; it < iters; ++it) {
                REP8(acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc0, 0, 0, 0);
                     acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1, 0, 0, 0);
                     acc2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc2, 0, 0, 0);
                     acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc0, 0, 0, 0);
                     acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1, 0, 0, 0);
                     acc2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc2, 0, 0, 0);
                     acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc0, 0, 0, 0);
                     acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1, 0, 0, 0);)
            }
        } else {
            for (int it = 0; it < iters; ++it) {
                REP8(acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc0, 0, 0, 0);
                     acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1, 0, 0, 0);
                     acc2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc2, 0, 0, 0);
                     acc3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc3, 0, 0, 0);)
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
        out[wave * 2 + 1] = (unsigned long long)(acc0 + acc1 + acc2 + acc3 + v0 + v1 + v2 + v3);
    }
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    // (1) layout by one-hot operands
    unsigned long long* dm;
    hipMalloc(&dm, 64 * 64 * 8);
    hipLaunchKernelGGL(onehot_kernel, dim3(1), dim3(64), 0, 0, dm);
    std::vector<unsigned long long> M(64 * 64);
    hipError_t e0 = hipMemcpy(M.data(), dm, 64 * 64 * 8, hipMemcpyDeviceToHost);
    printf("one-hot kernel: %s\n", hipGetErrorString(e0));
    for (int la = 0; la < 64; la += 5) {                        // raw sample: A lane -> (B lane : D lanes)
        printf("  A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb) if (M[la * 64 + lb]) printf("  B %d -> D mask %016llx", lb, M[la * 64 + lb]);
        printf("\n");
    }
    fflush(stdout);
    {
        int nz = 0;
        for (int i = 0; i < 4096; ++i) nz += M[i] != 0;
        if (nz != 256) { printf("unexpected: %d non-zero (A lane, B lane) pairs, expected 256 -- no layout analysis\n", nz); return 1; }
    }
    // Layout (read off the one-hot table, then checked against all 4 096 entries):
    //   A[i][k] of block b in lane i + 4 b + 16 k;  B[k][j] of block b in lane j + 4 b + 16 k;  D[i][j] of block b in lane j + 4 b + 16 i
    int ablock[64], ak[64], ai[64], bblock[64], bk[64], bj[64], dblock[64], di[64], dj[64];
    for (int l = 0; l < 64; ++l) {
        ai[l] = l & 3; ablock[l] = (l >> 2) & 3; ak[l] = l >> 4;
        bj[l] = l & 3; bblock[l] = (l >> 2) & 3; bk[l] = l >> 4;
        dj[l] = l & 3; dblock[l] = (l >> 2) & 3; di[l] = l >> 4;
    }
    {
        int bad = 0;
        for (int la = 0; la < 64; ++la)
            for (int lb = 0; lb < 64; ++lb) {
                unsigned long long want = 0;
                if (ablock[la] == bblock[lb] && ak[la] == bk[lb]) want = 1ull << (bj[lb] + 4 * ablock[la] + 16 * ai[la]);
                bad += M[la * 64 + lb] != want;
            }
        printf("layout: A[i][k].block b in lane i + 4 b + 16 k; B[k][j].block b in lane j + 4 b + 16 k; D[i][j].block b in lane j + 4 b + 16 i:"
               " %d of 4096 one-hot products disagree\n", bad);
    }

    // (2) a real product against the host's fma chain k = 0 .. 3, using the layout just found
    {
        std::vector<double> A(64), B(64), C(64), D(64);
        for (int i = 0; i < 64; ++i) { A[i] = std::sin(1.0 + i); B[i] = std::cos(0.5 * i + 2.0); C[i] = std::sin(0.3 * i) * 3.0; }
        double *dA, *dB, *dC, *dD;
        hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dC, 512); hipMalloc(&dD, 512);
        hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(product_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, 512, hipMemcpyDeviceToHost);
        int exact_up = 0, exact_down = 0, close = 0;
        for (int l = 0; l < 64; ++l) {
            double av[4], bv[4];
            for (int la = 0; la < 64; ++la) if (ablock[la] == dblock[l] && ai[la] == di[l]) av[ak[la]] = A[la];
            for (int lb = 0; lb < 64; ++lb) if (bblock[lb] == dblock[l] && bj[lb] == dj[l]) bv[bk[lb]] = B[lb];
            double up = C[l], down = C[l];
            for (int k = 0; k < 4; ++k) up = std::fma(av[k], bv[k], up);
            for (int k = 3; k >= 0; --k) down = std::fma(av[k], bv[k], down);
            exact_up += up == D[l]; exact_down += down == D[l]; close += std::fabs(up - D[l]) < 1e-12;
        }
        printf("product: C + sum_k A B of 64 results: within 1e-12: %d; bit-equal to the fma chain k = 0,1,2,3 from C: %d; to k = 3,2,1,0: %d\n",
               close, exact_up, exact_down);
    }

    // (5) lane swaps
    {
        unsigned* ds; hipMalloc(&ds, 256 * 4);
        hipLaunchKernelGGL(swap_kernel, dim3(1), dim3(64), 0, 0, ds);
        std::vector<unsigned> S(256);
        hipMemcpy(S.data(), ds, 1024, hipMemcpyDeviceToHost);
        printf("v_permlane16_swap v0, v1 (v0 = 1000 + lane, v1 = 2000 + lane): v0 at lanes 0,16,32,48 = %u %u %u %u; v1 = %u %u %u %u\n",
               S[0], S[16], S[32], S[48], S[64], S[80], S[96], S[112]);
        printf("v_permlane32_swap v0, v1:                                         v0 at lanes 0,16,32,48 = %u %u %u %u; v1 = %u %u %u %u\n",
               S[128], S[144], S[160], S[176], S[192], S[208], S[224], S[240]);
    }

    // (3), (4) rates
    unsigned long long* d;
    hipMalloc(&d, sizeof(unsigned long long) * 16);
    const int iters = 2000;
    auto run = [&](int mode, int chain, int nw, int weave, const char* what) {
        hipLaunchKernelGGL(rate_kernel, dim3(1), dim3(512), 0, 0, d, 20, mode, chain, nw, weave);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(rate_kernel, dim3(1), dim3(512), 0, 0, d, iters, mode, chain, nw, weave);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(16);
        hipMemcpy(h.data(), d, sizeof(unsigned long long) * 16, hipMemcpyDeviceToHost);
        if (mode & 4)
            printf("%-62s wave 4: %7.2f cycles / MFMA   wave 0: %7.2f cycles / v_fma_f64\n", what,
                   (double)h[8] / (iters * 32.0), (double)h[0] / (iters * 32.0));
        else
            printf("%-62s wave 0: %7.2f cycles / MFMA   wave 4: %7.2f cycles / %s\n", what,
                   (mode & 1) ? (double)h[0] / (iters * 32.0) : 0.0,
                   (double)h[8] / (iters * 32.0), (mode & 2) ? "v_fma_f64" : "MFMA");
    };
    run(1, 0, 4, 0, "MFMA 4x4x4, 4 accumulators, one wave per SIMD");
    run(1, 3, 4, 0, "MFMA 4x4x4, 3 accumulators, one wave per SIMD");
    run(1, 1, 4, 0, "MFMA 4x4x4, one dependent chain, one wave per SIMD");
    run(1, 0, 8, 0, "MFMA 4x4x4, 4 accumulators, two waves per SIMD");
    run(1, 3, 8, 0, "MFMA 4x4x4, 3 accumulators, two waves per SIMD");
    run(1, 1, 8, 0, "MFMA 4x4x4, one dependent chain, two waves per SIMD");
    run(2, 0, 4, 0, "v_fma_f64 alone (waves 4..7)");
    run(3, 0, 4, 0, "MFMA 4x4x4 (waves 0..3) beside v_fma_f64 (waves 4..7)");
    run(3, 1, 4, 0, "MFMA 4x4x4 one chain (waves 0..3) beside v_fma_f64 (4..7)");
    run(7, 0, 4, 0, "v_fma_f64 (OLDER waves 0..3) beside MFMA 4x4x4 (waves 4..7)");
    run(7, 1, 4, 0, "v_fma_f64 (waves 0..3) beside MFMA 4x4x4 one chain (4..7)");
    run(1, 0, 4, 1, "MFMA 4x4x4 + one v_fma_f64 each, own stream, one wave / SIMD");
    run(1, 0, 8, 1, "MFMA 4x4x4 + one v_fma_f64 each, own stream, two waves / SIMD");
    hipFree(d);
    return 0;
}
