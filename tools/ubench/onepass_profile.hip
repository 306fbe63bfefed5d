// onepass_profile.hip -- prices the one lever left for the latency-shaped one-pass fill (csrc/ta_nw.hip):
// a score profile in LDS for the TAGGED carried cell instead of the compare-select.  The tagged cell keeps scores as
// (score << 6) | tag, so a profile entry is 16 bits per row: one 32-bit word per lane and OCR symbol at R = 2 rows per
// lane, two at R = 4 (ds_read_b32 / ds_read_b64), unpacked by the add that forms M^.
//
// A lone wave (one per SIMD, as in the 1 x 4096^2 and 1 x 8192^2 launches) runs the steady step of the fill -- DPP pair,
// R cells, pointer bytes packed and stored once per group, the bottom-row write -- with (A) today's cell
// (cell_carried_tagged_c: v_cmp_eq + v_cndmask + add per cell) and (B) the profile cell; s_memtime cycles per step.
// Build: hipcc -O3 --offload-arch=gfx950 -I../../text_alignment_amd/csrc onepass_profile.hip -o onepass_profile
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#include "nw_hw.h"

using namespace ta;

template <bool SAMEGO>
__device__ __forceinline__ unsigned cell_profile(const CellRegs& k, int cs, int d_ul, int xg_u, int yg_l, int& d, int& xg, int& yg) {
    const int mr = (d_ul & k.clean) + cs;
    const int xr = (xg_u & k.clean) | kTagX;
    const int yr = yg_l & k.clean;
    d = max(max(mr, xr), yr);
    const int dgx = d + k.gox6;
    const int dgy = SAMEGO ? dgx : d + k.goy6;
    xg = max(dgx, xr);
    yg = max(dgy, yr);
    return v_bfi3((unsigned)d_ul, v_bfi12((unsigned)xg_u, (unsigned)yg_l));
}

template <int R, bool PROFILE>
__global__ __launch_bounds__(64) void step_kernel(unsigned long long* out, uint4* sink, int groups) {
    constexpr int SPG = 16 / R;
    __shared__ uint16_t ocode[4096 + 256];
    __shared__ __attribute__((aligned(16))) unsigned char tbl[28 * 64 * 8];
    __shared__ int2 hvd[64 * 4];
    const int lane = threadIdx.x;
    for (int j = lane; j < 4096 + 256; j += 64) ocode[j] = (uint16_t)((j * 7 + 3) % 27) * (PROFILE ? (R == 4 ? 512 : 256) : 1);
    for (int j = lane; j < 28 * 64 * 2; j += 64) reinterpret_cast<unsigned*>(tbl)[j] = 0xFEEAFEEAu + (unsigned)j;
    CellRegs kr;
    kr.cmis = (-1 * 64) | kTagM; kr.cmat = (11 * 64) | kTagM; kr.gox6 = -7 * 64; kr.goy6 = -7 * 64; kr.clean = ~kTagMask;
    asm volatile("" : "+v"(kr.cmis), "+v"(kr.cmat));
    int D[R], V[R], H[R], tc[R];
    for (int r = 0; r < R; ++r) { D[r] = -(lane * R + r) * 64; V[r] = D[r] - 448; H[r] = D[r] - 448; tc[r] = (lane * R + r) % 27; }
    int dsave = D[0] + 64;
    const unsigned char* tbl_lane = tbl + lane * (R == 4 ? 8 : 4);
    int2* wptr = hvd + lane;
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    int oc[SPG];
    unsigned long long pw[SPG];
    auto fetch = [&](int g) {
        const int idx = 64 + (g * SPG) % 3900 - lane;
#pragma unroll
        for (int q = 0; q < SPG; ++q) oc[q] = ocode[idx + q];
        if (PROFILE) {
#pragma unroll
            for (int q = 0; q < SPG; ++q) {
                if (R == 4) pw[q] = *reinterpret_cast<const unsigned long long*>(tbl_lane + oc[q]);
                else pw[q] = *reinterpret_cast<const unsigned*>(tbl_lane + oc[q]);
            }
        }
    };
    fetch(0);
    for (int g = 0; g < groups; ++g) {
        int ocn[SPG];
        unsigned long long pwn[SPG];
#pragma unroll
        for (int q = 0; q < SPG; ++q) { ocn[q] = oc[q]; pwn[q] = pw[q]; }
        fetch(g + 1);                                       // the next group's inputs fly under this group's cells
        unsigned bb[16];
#pragma unroll
        for (int q = 0; q < SPG; ++q) {
            int v_up = V[R - 1], d_next = D[R - 1];
            wave_shr1_pair<1>(v_up, V[R - 1], d_next, D[R - 1]);
            int d_ul = dsave, v_u = v_up;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int d_old = D[r];
                if (PROFILE) {
                    const int cs = (int)(short)(pwn[q] >> (16 * r));
                    bb[q * R + r] = cell_profile<true>(kr, cs, d_ul, v_u, H[r], D[r], V[r], H[r]);
                } else {
                    bb[q * R + r] = cell_carried_tagged_c<true>(kr, d_ul, v_u, H[r], tc[r], ocn[q], D[r], V[r], H[r]);
                }
                d_ul = d_old;
                v_u = V[r];
            }
            dsave = d_next;
            wptr[0] = make_int2(V[R - 1], D[R - 1]);
        }
        sink[(size_t)(g & 63) * 64 + lane] = make_uint4(pack4(bb[0], bb[1], bb[2], bb[3]), pack4(bb[4], bb[5], bb[6], bb[7]),
                                                        pack4(bb[8], bb[9], bb[10], bb[11]), pack4(bb[12], bb[13], bb[14], bb[15]));
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (lane == 0) out[blockIdx.x] = t1 - t0;
}

template <int R, bool PROFILE>
double run(int blocks) {
    unsigned long long* d; uint4* sink;
    hipMalloc(&d, 8 * blocks); hipMalloc(&sink, sizeof(uint4) * 64 * 64);
    const int groups = 20000;
    hipLaunchKernelGGL((step_kernel<R, PROFILE>), dim3(blocks), dim3(64), 0, 0, d, sink, 200);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((step_kernel<R, PROFILE>), dim3(blocks), dim3(64), 0, 0, d, sink, groups);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), d, 8 * blocks, hipMemcpyDeviceToHost);
    hipFree(d); hipFree(sink);
    return (double)h[0] / ((double)groups * (16 / R));
}

int main() {
    for (int blocks : {1, 256}) {
        const double a2 = run<2, false>(blocks), b2 = run<2, true>(blocks), a4 = run<4, false>(blocks), b4 = run<4, true>(blocks);
        printf("%4d lone waves: R = 2 rows per lane: compare-select %6.1f cycles / step, 16-bit profile %6.1f (%+.1f %%);  "
               "R = 4: compare-select %6.1f, profile %6.1f (%+.1f %%)\n", blocks, a2, b2, 100.0 * (b2 - a2) / a2, a4, b4,
               100.0 * (b4 - a4) / a4);
    }
    return 0;
}
