// cell_f64.hip -- what a tile of the float64 recurrence (csrc/ta_lstm_f64.hip) costs a wave alone on its SIMD:
// the 25 dependent f64 MFMAs, the float64 cell update (five exponentials, two reciprocals), and both in the
// kernel's order; shader cycles (s_memtime) and wall time (= the clock the chip holds under this load).
// Build: hipcc -O3 --offload-arch=gfx950 -I../../text_alignment_amd/csrc cell_f64.hip ../../text_alignment_amd/csrc/ta_common.cpp -o cell_f64
#include "../../text_alignment_amd/csrc/ta_lstm_f64.hip"
#include <cstdio>
#include <vector>

// what: bit 0 = MFMAs, bit 1 = cell
__global__ __launch_bounds__(256) void tile_kernel(unsigned long long* out, double* sink, int iters, int what) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double Aw[25];
    for (int k = 0; k < 25; ++k) Aw[k] = 1e-3 * (lane + 1) * (k + 1) * ((k & 1) ? -1.0 : 1.0);
    double b = 0.01 * (lane - 32);
    double c = 0.1, hsum = 0.0;
    f64x4 acc = {0.1 * lane, -0.2, 0.3, 0.05 * lane};
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
        if (what & 1) {
#pragma unroll
            for (int k = 0; k < 25; ++k) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Aw[k], b, acc, 0, 0, 0);
        }
        if (what & 2) {
            const double h = lstm_cell_f64(acc[0], acc[1], acc[2], acc[3], c, true, 0.3, -0.2, 0.1);
            hsum += h;
            b = h * 0.5;                                       // next tile's operand depends on this cell
            acc = (f64x4){h, -h, 0.5 * h, 0.25};
        } else {
            acc[0] *= 1e-3; acc[1] *= 1e-3; acc[2] *= 1e-3; acc[3] *= 1e-3;
        }
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = hsum + acc[0] + c;
}

int main() {
    unsigned long long* out; double* sink;
    const int nblk = 256;
    hipMalloc(&out, nblk * 4 * 8); hipMalloc(&sink, nblk * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    const char* names[4] = {"", "25 MFMAs (one chain)", "cell update", "25 MFMAs + cell update"};
    for (int nb : {1, 256}) {
        for (int what = 1; what <= 3; ++what) {
            hipLaunchKernelGGL(tile_kernel, dim3(nb), dim3(256), 0, 0, out, sink, 100, what);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(tile_kernel, dim3(nb), dim3(256), 0, 0, out, sink, iters, what);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(nb * 4);
            hipMemcpy(h.data(), out, nb * 4 * 8, hipMemcpyDeviceToHost);
            double cyc = 0; for (auto v : h) cyc += (double)v; cyc /= h.size();
            // s_memtime counts at a fixed 100 MHz on this part: convert with the wall time instead of trusting it
            printf("%3d workgroups  %-26s  %9.1f memtime ticks / tile   %8.3f us / tile (wall)\n", nb, names[what],
                   cyc / iters, ms * 1e3 / iters);
        }
    }
    return 0;
}
