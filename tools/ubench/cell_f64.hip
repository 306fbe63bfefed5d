// cell_f64.hip -- what a tile of the float64 recurrence (csrc/ta_lstm_f64.hip) costs a wave alone on its SIMD:
// the 25 dependent f64 MFMAs, the float64 cell update (five exponentials, two reciprocals), and both in the
// kernel's order; shader cycles (s_memtime) and wall time (= the clock the chip holds under this load).
// Build: hipcc -O3 --offload-arch=gfx950 -I../../text_alignment_amd/csrc cell_f64.hip ../../text_alignment_amd/csrc/ta_common.cpp -o cell_f64
#include "../../text_alignment_amd/csrc/ta_lstm_f64.hip"
#include <cstdio>
#include <cmath>
#include <vector>

// accuracy: the device cell update against the host's long double arithmetic on random pre-activations (round 5);
// and the seed of the reciprocal (v_rcp_f64) and the exponential alone
__global__ void cell_check_kernel(const double* in, double* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* v = in + 8 * i;
    double c = v[4];
    const double h = lstm_cell_f64(v[0], v[1], v[2], v[3], c, v[5], v[6], v[7]);
    out[4 * i] = h; out[4 * i + 1] = c;
    out[4 * i + 2] = __builtin_amdgcn_rcp(1.0 + std::fabs(v[0]) * 1e3);       // the raw seed
    out[4 * i + 3] = exp_f64(clamp20(v[1]));
}
static long double sigl(long double x) { x = -x; if (x < -20) x = -20; if (x > 20) x = 20; return 1.0L / (1.0L + expl(x)); }

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
            const double h = lstm_cell_f64(acc[0], acc[1], acc[2], acc[3], c, 0.3, -0.2, 0.1);
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
    {
        const int n = 1 << 16;
        std::vector<double> in(8 * n), got(4 * n);
        unsigned long long st = 88172645463325252ull;
        auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
        for (int i = 0; i < n; ++i) {
            const double span = i % 4 == 0 ? 60.0 : 8.0;                       // a quarter of the cases reach beyond the clips
            for (int k = 0; k < 4; ++k) in[8 * i + k] = (rnd() - 0.5) * span;
            in[8 * i + 4] = i % 7 == 0 ? 0.0 : (rnd() - 0.5) * (i % 5 == 0 ? 50.0 : 4.0);   // cell state
            for (int k = 5; k < 8; ++k) in[8 * i + k] = rnd() - 0.5;
        }
        double *din, *dout;
        hipMalloc(&din, in.size() * 8); hipMalloc(&dout, got.size() * 8);
        hipMemcpy(din, in.data(), in.size() * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(cell_check_kernel, dim3(n / 256), dim3(256), 0, 0, din, dout, n);
        hipMemcpy(got.data(), dout, got.size() * 8, hipMemcpyDeviceToHost);
        long double eh = 0, ec = 0, er = 0, ee = 0;
        for (int i = 0; i < n; ++i) {
            const double* v = &in[8 * i];
            const long double cp = v[4];
            const long double gi = sigl(v[0] + v[5] * cp), gf = sigl(v[1] + v[6] * cp), ci = tanhl((long double)v[2 + 1]);
            const long double cn = ci * gi + gf * cp;
            const long double go = sigl(v[2] + v[7] * cn), h = tanhl(cn) * go;
            eh = fmaxl(eh, fabsl(got[4 * i] - h));
            ec = fmaxl(ec, fabsl(got[4 * i + 1] - cn) / fmaxl(1.0L, fabsl(cn)));
            const long double d = 1.0L + fabsl((long double)v[0]) * 1e3L;
            er = fmaxl(er, fabsl(got[4 * i + 2] * d - 1.0L));
            long double x = v[1]; if (x < -20) x = -20; if (x > 20) x = 20;
            ee = fmaxl(ee, fabsl(got[4 * i + 3] / expl(x) - 1.0L));
        }
        printf("cell update against long double on %d random cases: max |h error| %.3Le, max relative c error %.3Le;  "
               "exp_f64 max relative error %.3Le;  v_rcp_f64 seed max relative error %.3Le (TA_F64_RCP_NEWTON = %d)\n",
               n, eh, ec, ee, er, TA_F64_RCP_NEWTON);
        hipFree(din); hipFree(dout);
    }
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
