// mfma4x4.hip -- is v_mfma_f32_4x4x1_16B_f32 (one k per instruction, 16 blocks of 4 x 4, A broadcast from the
// block `abid` names) a drop-in for v_mfma_f32_16x16x4_f32 in the recurrence kernel for groups of FOUR lines?
// (1) operand layout + broadcast semantics, (2) bit-equality of a 152-term accumulation with the 16x16x4 form
// and with a sequential fmaf chain on the host, (3) cycles per MFMA of ONE dependent accumulation chain per
// wave with one and two waves per SIMD.   hipcc -O3 --offload-arch=gfx950 mfma4x4.hip -o mfma4x4
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 152;

// out[line][col], line < 4, col < 64: sum_k A[line][k] * B[k][col], k ascending
__global__ void k4x4(const float* A, const float* B, float* out) {
    const int lane = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // a VGPR of A serves 16 k's: lane = 4 * (k % 16) + line; block `abid` = k % 16 is broadcast to all blocks
    float a[(K + 15) / 16];
    for (int c = 0; c < (K + 15) / 16; ++c) {
        const int k = 16 * c + lane / 4;
        a[c] = k < K ? A[(lane % 4) * K + k] : 0.f;
    }
#define STEP16(c)                                                                                      \\
    _Pragma("unroll") for (int kk = 0; kk < 16; ++kk) { }
    // (abid must be an immediate: spelled out)
#define M(c, kk) if (16 * (c) + (kk) < K) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[c], B[(16 * (c) + (kk)) * 64 + lane], acc, 4, kk, 0);
#define M16(c) M(c,0) M(c,1) M(c,2) M(c,3) M(c,4) M(c,5) M(c,6) M(c,7) M(c,8) M(c,9) M(c,10) M(c,11) M(c,12) M(c,13) M(c,14) M(c,15)
    M16(0) M16(1) M16(2) M16(3) M16(4) M16(5) M16(6) M16(7) M16(8) M16(9)
    // D: register r = row i (line), lane = 4 * block + j  ->  column = lane if B was laid out [k][lane]
    for (int r = 0; r < 4; ++r) out[r * 64 + lane] = acc[r];
}

// the same product with 16x16x4: M = 16 (rows 0..3 real), four column tiles of 16
__global__ void k16(const float* A, const float* B, float* out) {
    const int lane = threadIdx.x;
    for (int t = 0; t < 4; ++t) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int kk = 0; kk < K / 4; ++kk) {
            const int k = 4 * kk + lane / 16;                        // A: row = lane % 16, k = lane / 16
            const float av = (lane % 16) < 4 ? A[(lane % 16) * K + k] : 0.f;
            const float bv = B[k * 64 + 16 * t + lane % 16];         // B: col = lane % 16, k = lane / 16
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
        }
        // D: lane -> col = lane % 16, rows 4 * (lane / 16) + r
        if (lane / 16 == 0) for (int r = 0; r < 4; ++r) out[r * 64 + 16 * t + lane % 16] = acc[r];
    }
}

template <int KIND>
__global__ __launch_bounds__(512) void rate(unsigned long long* out, int iters, int waves) {
    const int wave = threadIdx.x >> 6;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = acc, acc3 = acc, acc4 = acc;
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 1e-6f;
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (wave < waves) {
        for (int it = 0; it < iters; ++it) {
            if (KIND == 0) {                    // one dependent chain of 4x4x1
#pragma unroll
                for (int q = 0; q < 32; ++q) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 4, 3, 0);
            } else {                            // four interleaved chains of 16x16x4 (the shipped kernel's shape)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc2, 0, 0, 0);
                    acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc3, 0, 0, 0);
                    acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4, 0, 0, 0);
                }
            }
        }
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if ((threadIdx.x & 63) == 0) { out[wave * 2] = t1 - t0; out[wave * 2 + 1] = (unsigned long long)(acc[0] + acc2[1] + acc3[2] + acc4[3]); }
}

int main() {
    std::vector<float> A(4 * K), B(K * 64), o4(256), o16(256), ref(256);
    srand(5);
    for (auto& v : A) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
    for (auto& v : B) v = (rand() / (float)RAND_MAX - 0.5f);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 64; ++j) {
            float s = 0.f;
            for (int k = 0; k < K; ++k) s = fmaf(A[i * K + k], B[k * 64 + j], s);
            ref[i * 64 + j] = s;
        }
    float *dA, *dB, *dO;
    (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dO, 256 * 4);
    (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k4x4, dim3(1), dim3(64), 0, 0, dA, dB, dO);
    (void)hipMemcpy(o4.data(), dO, 256 * 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dO);
    (void)hipMemcpy(o16.data(), dO, 256 * 4, hipMemcpyDeviceToHost);
    int d4 = 0, d16 = 0, d416 = 0;
    double e4 = 0;
    for (int i = 0; i < 256; ++i) {
        d4 += memcmp(&o4[i], &ref[i], 4) != 0; d16 += memcmp(&o16[i], &ref[i], 4) != 0; d416 += memcmp(&o4[i], &o16[i], 4) != 0;
        e4 = fmax(e4, fabs((double)o4[i] - ref[i]));
    }
    printf("4x4x1 vs host fmaf chain: %d of 256 differ (max abs %.3g); 16x16x4 vs chain: %d; 4x4x1 vs 16x16x4: %d\n", d4, e4, d16, d416);
    unsigned long long* d;
    (void)hipMalloc(&d, 16 * 8);
    for (int kind = 0; kind < 2; ++kind)
        for (int waves : {4, 8}) {
            const int iters = 2000;
            for (int rep = 0; rep < 2; ++rep) {
                if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(1), dim3(512), 0, 0, d, iters, waves);
                else hipLaunchKernelGGL(rate<1>, dim3(1), dim3(512), 0, 0, d, iters, waves);
                (void)hipDeviceSynchronize();
            }
            unsigned long long h[16];
            (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            printf("%s, %d wave(s) per SIMD: cycles per MFMA per wave:", kind == 0 ? "4x4x1 one chain" : "16x16x4 four chains", waves / 4);
            for (int w = 0; w < waves; ++w) printf(" %.2f", (double)h[2 * w] / (iters * 32.0));
            printf("\n");
        }
    return 0;
}
