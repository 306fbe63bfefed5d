// mfma_f64_ops.hip -- what sits between the f64 MFMAs of one dependent chain, one wave per SIMD (csrc/ta_lstm_f64.hip keeps its
// weights in AGPRs): (a) nothing, (b) two v_accvgpr_read_b32 per MFMA (the compiler's way of feeding an AGPR-resident
// operand), (c) two v_mov_b32, (d) the A operand read from AGPRs by the MFMA itself, (e) one ds_read_b64 per MFMA,
// (f) one v_fma_f64 per MFMA.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_f64_ops.hip -o mfma_f64_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double f64x4 __attribute__((ext_vector_type(4)));
#define REP16(X) X X X X X X X X X X X X X X X X

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* out, double* sink, int iters) {
    __shared__ double lds[256];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    lds[threadIdx.x] = threadIdx.x;
    f64x4 acc = {0, 0, 0, 0};
    double a = threadIdx.x * 0.001, b = 1.0 + threadIdx.x * 1e-6, aw = a * 0.5, v = 0.25, l = 0;
    int t0r = 1, t1r = 2, ag = threadIdx.x;
    const unsigned addr = (threadIdx.x & 63) * 8;
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { REP16(asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));) }
        if (MODE == 1) { REP16(asm volatile("v_accvgpr_read_b32 %2, %5\n v_accvgpr_read_b32 %3, %5\n s_nop 1\n v_mfma_f64_16x16x4_f64 %0, %1, %4, %0"
                                            : "+a"(acc), "+v"(a), "+v"(t0r), "+v"(t1r) : "v"(b), "a"(ag));) }
        if (MODE == 2) { REP16(asm volatile("v_mov_b32 %2, %5\n v_mov_b32 %3, %5\n s_nop 1\n v_mfma_f64_16x16x4_f64 %0, %1, %4, %0"
                                            : "+a"(acc), "+v"(a), "+v"(t0r), "+v"(t1r) : "v"(b), "v"(ag));) }
        if (MODE == 3) { REP16(asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc) : "a"(aw), "v"(b));) }
        if (MODE == 4) { REP16(asm volatile("ds_read_b64 %2, %4\n v_mfma_f64_16x16x4_f64 %0, %1, %3, %0\n s_waitcnt lgkmcnt(0)"
                                            : "+a"(acc), "+v"(a), "+v"(l) : "v"(b), "v"(addr));) }
        if (MODE == 5) { REP16(asm volatile("v_fma_f64 %2, %2, %1, %3\n v_mfma_f64_16x16x4_f64 %0, %1, %3, %0"
                                            : "+a"(acc), "+v"(a), "+v"(v) : "v"(b));) }
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
    sink[threadIdx.x] = acc[0] + acc[3] + v + l + t0r + t1r;
}

int main() {
    unsigned long long* out; double* sink;
    hipMalloc(&out, 64); hipMalloc(&sink, 256 * 8);
    const int iters = 2000;
    const char* names[6] = {"MFMAs only", "+ 2 v_accvgpr_read_b32 each", "+ 2 v_mov_b32 each", "A operand in AGPRs",
                            "+ 1 ds_read_b64 each", "+ 1 v_fma_f64 each"};
    auto run = [&](auto kern, int m) {
        hipLaunchKernelGGL(kern, dim3(1), dim3(256), 0, 0, out, sink, 10);
        hipLaunchKernelGGL(kern, dim3(1), dim3(256), 0, 0, out, sink, iters);
        unsigned long long h[4];
        hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
        printf("%-32s %7.2f cycles / MFMA\n", names[m], (double)h[0] / (iters * 16.0));
    };
    run(k<0>, 0); run(k<1>, 1); run(k<2>, 2); run(k<3>, 3); run(k<4>, 4); run(k<5>, 5);
    return 0;
}
