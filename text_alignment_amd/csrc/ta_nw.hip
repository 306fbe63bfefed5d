// ta_nw.hip -- affine-gap Needleman-Wunsch on MI355X (gfx950): wavefront fill + traceback.
//
// Replaces the arithmetic of textSeqCompare.perform_alignment (reference
// textSeqCompare.py:13-177): boundary rows :53-60, DP fill :62-88, traceback :96-170.
//
// Fill kernel (K1).  One workgroup per problem, W waves.  The DP table is cut into strips
// of 64*R rows; wave w takes strips w, w+W, ...  Inside a strip lane l owns R consecutive
// rows and sweeps the columns skewed by its lane id, so one step of the wave is one
// anti-diagonal band: the value a lane needs from the row above is what lane l-1 produced
// one step earlier and arrives with a single DPP wave_shr:1 (no LDS, no bpermute).  The
// three affine-gap score bands never leave registers; only the strip's bottom row (two
// ints per column) goes through an LDS hand-off row to the wave working on the next strip,
// which runs a few hundred columns behind it (progress words in LDS, polled).  HBM sees
// exactly one byte per cell: the packed pointers, collected 16 bytes per lane and stored as
// fully coalesced 1 KiB wave stores in a strip-major skewed layout (nw_cell.h).
//
// Traceback kernel (K2).  One wave per problem walks the pointer bytes from (n, m)
// (textSeqCompare.py:100-164) and writes the alignment columns right-aligned into the
// caller's buffer.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nw_cell.h"
#include "nw_hw.h"
#include "ta_common.h"

namespace ta {

// Hand-off rows in HBM for the wide launch (a problem spread over several workgroups): one row of
// (V, D) pairs below every kWideW-th strip, and one progress word per row, behind the pointer bytes.
constexpr int kWideW = 4;                         // waves (strips) per workgroup of the wide launch
template <int R>
struct WideWs {
    int64_t rows_off, prog_off, total;
    int row_elems, nrows;
    __host__ __device__ WideWs(int n, int m) {
        using L = PtrLayout<R>;
        nrows = L::nstrips(n) / kWideW;           // boundaries below strips kWideW-1, 2 kWideW-1, ...
        row_elems = (m + 2 + 1) & ~1;             // int2 entries, 16-byte multiple
        rows_off = L::total_bytes(n, m);
        prog_off = rows_off + (int64_t)nrows * row_elems * 8;
        total = (prog_off + (int64_t)nrows * 4 + 1023) & ~(int64_t)1023;
    }
};

template <int R>
__global__ __launch_bounds__(64) void nw_wide_init_kernel(NwArgs a) {
    const int p = blockIdx.x;
    const int n = (int)(a.t_off[p + 1] - a.t_off[p]);
    const int m = (int)(a.o_off[p + 1] - a.o_off[p]);
    if (n <= 0 || m <= 0) return;
    const WideWs<R> wl(n, m);
    int* gprog = reinterpret_cast<int*>(a.ws + a.ws_off[p] + wl.prog_off);
    for (int i = threadIdx.x; i < wl.nrows; i += 64) gprog[i] = 0;
}

// WIDE = false: one workgroup per problem, wave w takes strips w, w+W, ... (hand-off in LDS only).
// WIDE = true:  workgroup (chunk, p) takes strips chunk*W .. chunk*W+W-1 of problem p; the bottom
// row of a workgroup's last strip goes to the next workgroup through HBM (exported from the LDS
// hand-off row every kCheck groups, progress word released at agent scope).  A workgroup only ever
// waits for a workgroup with a smaller block index, so in-order dispatch guarantees progress.
template <int R, int W, bool WIDE>
__global__ __launch_bounds__(W * 64) void nw_fill_kernel(NwArgs a) {
    using L = PtrLayout<R>;
    constexpr int SPG = L::SPG;
    // groups between two looks at the LDS progress word of the strip above: a strip follows the one
    // above at CHK + 17 groups (the wide launch exports to HBM every kCheck groups regardless)
    constexpr int CHK = 4;
    constexpr int DW = R / 4;                     // dwords of pointer bytes per step
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int p = WIDE ? (int)(blockIdx.x % a.wide_stride) : (int)blockIdx.x;
    const int chunk = WIDE ? (int)(blockIdx.x / a.wide_stride) : 0;
    if (WIDE && p >= a.nprob) return;
    const int64_t t0 = a.t_off[p], o0 = a.o_off[p];
    const int n = (int)(a.t_off[p + 1] - t0);
    const int m = (int)(a.o_off[p + 1] - o0);
    if (n <= 0 || m <= 0) return;                 // nothing to fill; traceback emits pure gaps
    if (WIDE && chunk * W >= L::nstrips(n)) return;

    const int32_t* prm = a.params + (size_t)p * a.params_stride;
    const CellConsts c = make_consts(prm[0], prm[1], prm[2], prm[3], prm[4], prm[5]);
    CellRegs kr;
    kr.cmis = c.cmismatch; kr.cmat = c.cmatch; kr.gox6 = c.gox6; kr.goy6 = c.goy6;
    kr.clean = ~kTagMask;
    // keep the two select constants resident in VGPRs (hipcc otherwise re-materialises them with
    // two v_mov per step)
    asm volatile("" : "+v"(kr.cmis), "+v"(kr.cmat));

    const NwLds lds(m);
    int2* hvd = reinterpret_cast<int2*>(smem);
    int2* dummy = reinterpret_cast<int2*>(smem + lds.hvd_bytes);
    uint16_t* ocode = reinterpret_cast<uint16_t*>(smem + lds.hvd_bytes + lds.dummy_bytes);
    int* prog = reinterpret_cast<int*>(smem + lds.hvd_bytes + lds.dummy_bytes + lds.oc_bytes);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // stage the OCR codes and the row-0 boundary (textSeqCompare.py:57-60) into LDS
    for (int j = tid; j < kOPad + m + kOTail; j += W * 64) {
        const int src = j - kOPad;
        ocode[j] = (src >= 0 && src < m) ? (uint16_t)a.o_codes[o0 + src] : (uint16_t)0xFFFF;
    }
    for (int j = tid; j <= m; j += W * 64) hvd[j] = make_int2(bnd_V_row0(c, j), bnd_D_row0(c, j));
    if (tid < 16) prog[tid] = 0;
    __syncthreads();

    const int nstrips = L::nstrips(n);
    const int ngroups = L::ngroups(m);
    const int64_t strip_bytes = L::strip_bytes(m);
    uint8_t* const ws_p = a.ws + a.ws_off[p];
    const WideWs<R> wl(n, m);
    int2* const xrows = reinterpret_cast<int2*>(ws_p + wl.rows_off);
    int* const gprog = reinterpret_cast<int*>(ws_p + wl.prog_off);
    int imp_hi = 0, exp_hi = 0;                   // hand-off columns imported / exported so far (WIDE)
    const int prev_wave = (wave + W - 1) % W;
    // groups [g_lo, g_hi) are "steady": every lane is inside 1 <= j <= m on every step
    const int g_lo = (63 + SPG - 1) / SPG;
    const int g_hi = m / SPG;
    int pass = 0;

    for (int s = chunk * W + wave; s < nstrips; s += (WIDE ? nstrips : W), ++pass) {
        // ---- per-strip lane state: column-0 boundary (textSeqCompare.py:53-56) ----
        int D[R], V[R], H[R], tc[R];
        const int row0 = s * L::SR + lane * R;            // 0-based index of this lane's first row
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = row0 + r + 1;
            D[r] = bnd_D_col0(c, i);
            H[r] = bnd_H_col0(c, i);
            V[r] = 0;
            tc[r] = (i <= n) ? a.t_codes[t0 + i - 1] : -1;
        }
        int dsave = bnd_D_col0(c, row0);
        // retire the transcript-code loads before the group loops (keeps s_waitcnt vmcnt(0), which
        // would also drain the pointer stores, out of the loop body)
#pragma unroll
        for (int r = 0; r < R; ++r) asm volatile("" :: "v"(tc[r]));
        const bool lane_has_rows = row0 < n;
        uint8_t* out = ws_p + (int64_t)s * strip_bytes + (int64_t)lane * 16;
        const int prod_pass = (wave == 0) ? pass - 1 : pass;   // pass in which prev_wave did strip s-1

        // the strip above must be kCheck+1 groups ahead before this wave touches a span:
        // the hand-off entries of group g+1 are prefetched while group g is computed
        auto wait_span = [&](int g_first) {
            if (s == 0 || (W == 1 && !WIDE)) return;
            // the last step of groups [g_first, g_first + CHK] reads hand-off column
            // min(k_last + 1, m), written by the producer's lane 63 at its step col + 62
            const int k_last = min((g_first + CHK + 1) * SPG - 1, L::nsteps(m) - 1);
            const int col = min(k_last + 1, m);
            const int need_groups = min(ngroups, (col + 62) / SPG + 1);
            if (WIDE && wave == 0) {
                // the strip above belongs to the previous workgroup: wait for its progress word,
                // then pull every newly final column from its HBM row into the LDS hand-off row
                if (imp_hi >= col) return;
                const int row = s / W - 1;
                int have;
                while (true) {
                    have = __builtin_amdgcn_readfirstlane(
                        __hip_atomic_load(&gprog[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if (have >= need_groups) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                const int cfin = (have >= ngroups) ? m : min(m, have * SPG - 1 - 62);
                const int2* src = xrows + (int64_t)row * wl.row_elems;
                for (int j = imp_hi + 1 + lane; j <= cfin; j += 64) hvd[j] = src[j];
                imp_hi = cfin;
                return;
            }
            const int need = prod_pass * ngroups + need_groups;
            while (true) {
                const int have = __hip_atomic_load(&prog[prev_wave], __ATOMIC_ACQUIRE,
                                                   __HIP_MEMORY_SCOPE_WORKGROUP);
                if (__builtin_amdgcn_readfirstlane(have) >= need) break;
                __builtin_amdgcn_s_sleep(2);
            }
        };
        // progress is published after groups 0, CHK, 2 CHK, ...: the consumer's needs are 1 mod CHK
        auto publish = [&](int g) {
            if (WIDE && wave == W - 1) {
                if (((g % kCheck) == 0 || g == ngroups - 1) && s + 1 < nstrips) {
                    // lane 63 has finished columns <= k - 62 of this strip's bottom row
                    const int cfin = (g == ngroups - 1) ? m : min(m, (g + 1) * SPG - 1 - 62);
                    int2* dst = xrows + (int64_t)(s / W) * wl.row_elems;
                    for (int j = exp_hi + 1 + lane; j <= cfin; j += 64) dst[j] = hvd[j];
                    exp_hi = max(exp_hi, cfin);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    if (lane == 0)
                        __hip_atomic_store(&gprog[s / W], g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return;
            }
            if (W == 1) return;
            if ((g % CHK) == 0 || g == ngroups - 1) {
                if (lane == 63)
                    __hip_atomic_store(&prog[wave], pass * ngroups + g + 1, __ATOMIC_RELEASE,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };
        // per-group inputs: this lane's OCR codes and (lane 0) the row above, from LDS
        int oc_next[SPG];
        int2 hd_next[SPG];
        auto load_group = [&](int g) {
            const int idx = kOPad + g * SPG - lane;        // o index of step k is k - lane
#pragma unroll
            for (int q = 0; q < SPG; ++q) {
                oc_next[q] = ocode[idx + q];
                hd_next[q] = hvd[min(g * SPG + q + 1, m)]; // lane 0's column at step k is k + 1
            }
        };
        auto prefetch = [&](int g) {
            if (g + 1 < ngroups) {
                if (((g + 1) % CHK) == 0) wait_span(g + 1);
                load_group(g + 1);
            }
        };
        // one group with per-lane activity tests (ramp-up, ramp-down, short rows)
        auto group_edge = [&](int g) {
            int oc[SPG];
            int2 hd[SPG];
#pragma unroll
            for (int q = 0; q < SPG; ++q) { oc[q] = oc_next[q]; hd[q] = hd_next[q]; }
            prefetch(g);
            unsigned acc[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int q = 0; q < SPG; ++q) {
                const int k = g * SPG + q;
                const int j = k - lane + 1;               // this lane's column at step k (1-based)
                const bool active = (j >= 1) && (j <= m) && lane_has_rows;
                // lane 0 takes the row above from the hand-off row, lanes 1..63 from lane-1
                int v_up = hd[q].x, d_next = hd[q].y;
                wave_shr1_pair<4>(v_up, V[R - 1], d_next, D[R - 1]);
                if (active) {
                    int d_ul = dsave, v_u = v_up;
                    unsigned b[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int d_old = D[r];
                        b[r] = cell_hw(kr, d_ul, v_u, H[r], tc[r], oc[q], D[r], V[r], H[r]);
                        d_ul = d_old;
                        v_u = V[r];
                    }
#pragma unroll
                    for (int x = 0; x < DW; ++x)
                        acc[q * DW + x] = pack4(b[4 * x], b[4 * x + 1], b[4 * x + 2], b[4 * x + 3]);
                    dsave = d_next;
                    if (lane == 63) hvd[j] = make_int2(V[R - 1], D[R - 1]);
                }
            }
            *reinterpret_cast<uint4*>(out + (int64_t)g * 1024) = make_uint4(acc[0], acc[1], acc[2], acc[3]);
            publish(g);
        };

        wait_span(0);
        load_group(0);
        int g = 0;
        const int e1 = min(g_lo, ngroups);
        for (; g < e1; ++g) group_edge(g);

        if (g < g_hi) {
            // ---- steady state: straight-line code, no EXEC changes.  Lanes whose rows lie
            // below row n compute don't-care values that never reach a valid row.  Two groups per
            // iteration with two input buffers (A / B): the LDS prefetch of the next group lands in
            // the other buffer, so the loop back-edge needs no register copies. ----
            // lane 63 publishes its bottom row to hvd[j], j = k - 62; the other lanes write a
            // private dummy slot (8-byte lane stride: bank-conflict free) so the store needs no EXEC mask
            int2* wptr = (lane == 63) ? (hvd + (g * SPG - 62)) : (dummy + lane);
            const int winc = (lane == 63) ? SPG : 0;
            int ocA[SPG], ocB[SPG];
            int2 hdA[SPG], hdB[SPG];
#pragma unroll
            for (int q = 0; q < SPG; ++q) { ocA[q] = oc_next[q]; hdA[q] = hd_next[q]; }
            auto fetch = [&](int gn, int (&oc)[SPG], int2 (&hd)[SPG]) {       // inputs of group gn
                if (gn < ngroups) {
                    if ((gn % CHK) == 0) wait_span(gn);
                    const int idx = kOPad + gn * SPG - lane;
#pragma unroll
                    for (int q = 0; q < SPG; ++q) {
                        oc[q] = ocode[idx + q];
                        hd[q] = hvd[min(gn * SPG + q + 1, m)];
                    }
                }
            };
            auto steady = [&](int gg, const int (&oc)[SPG], const int2 (&hd)[SPG]) {
                unsigned acc[4];
#pragma unroll
                for (int q = 0; q < SPG; ++q) {
                    int v_up = hd[q].x, d_next = hd[q].y;
                    wave_shr1_pair<1>(v_up, V[R - 1], d_next, D[R - 1]);
                    int d_ul = dsave, v_u = v_up;
                    unsigned b[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int d_old = D[r];
                        b[r] = cell_hw(kr, d_ul, v_u, H[r], tc[r], oc[q], D[r], V[r], H[r]);
                        d_ul = d_old;
                        v_u = V[r];
                    }
#pragma unroll
                    for (int x = 0; x < DW; ++x)
                        acc[q * DW + x] = pack4(b[4 * x], b[4 * x + 1], b[4 * x + 2], b[4 * x + 3]);
                    dsave = d_next;
                    wptr[q] = make_int2(V[R - 1], D[R - 1]);
                }
                wptr += winc;
                *reinterpret_cast<uint4*>(out + (int64_t)gg * 1024) = make_uint4(acc[0], acc[1], acc[2], acc[3]);
                publish(gg);
            };
            while (g + 1 < g_hi) {
                fetch(g + 1, ocB, hdB);
                steady(g, ocA, hdA);
                fetch(g + 2, ocA, hdA);
                steady(g + 1, ocB, hdB);
                g += 2;
            }
            if (g < g_hi) {
                fetch(g + 1, ocB, hdB);
                steady(g, ocA, hdA);
                ++g;
#pragma unroll
                for (int q = 0; q < SPG; ++q) { oc_next[q] = ocB[q]; hd_next[q] = hdB[q]; }
            } else {
#pragma unroll
                for (int q = 0; q < SPG; ++q) { oc_next[q] = ocA[q]; hd_next[q] = hdA[q]; }
            }
        }
        for (; g < ngroups; ++g) group_edge(g);
    }
}

// K2: pointer walk, textSeqCompare.py:96-170.  One wave per problem.
//
// A walk step needs one pointer byte, and the next address depends on it, so reading HBM per
// step costs a full memory latency (~200 ns) each.  The wave instead pulls a WINDOW of the
// strip layout into LDS -- all 64 lanes (the strip's 256 rows) x kTbGroups groups (4 skewed
// steps each), i.e. kTbGroups fully coalesced 1 KiB loads -- and walks inside it from LDS.
// Both window coordinates only ever decrease along the walk (row up => lane down, and the
// skewed step k = (j-1) + lane never grows), so a window is left exactly once: through the
// top of the strip or through its low-k side.
constexpr int kTbGroups = 32;                 // window depth in groups (128 skewed steps)
constexpr int kTbOps = 512;                   // alignment columns buffered per window

template <int R>
__global__ __launch_bounds__(64) void nw_traceback_kernel(NwArgs a) {
    static_assert(R == 4, "window walk is written for 4 rows per lane");
    using L = PtrLayout<R>;
    __shared__ uint4 win[kTbGroups * 64];
    __shared__ uint8_t opsbuf[kTbOps];
    const int p = blockIdx.x, lane = threadIdx.x;
    const int n = (int)(a.t_off[p + 1] - a.t_off[p]);
    const int m = (int)(a.o_off[p + 1] - a.o_off[p]);
    const uint8_t* ws_p = a.ws + a.ws_off[p];
    uint8_t* ops = a.ops_out + a.ops_off[p];
    const int cap = n + m;
    const int64_t strip_bytes = L::strip_bytes(m);
    int x = n, y = m, len = 0;
    int st = 0;
    bool first = true;
    while (x > 0 && y > 0) {
        // position in layout coordinates
        int strip = (x - 1) / L::SR;
        int l = ((x - 1) % L::SR) / R;
        int r = (x - 1) % R;
        int k = (y - 1) + l;
        const int g_hi = k >> 2;
        const int g_lo = max(0, g_hi - (kTbGroups - 1));
        {   // load the window: piece (g, lane) -> win[(g - g_lo) * 64 + lane]
            const uint8_t* base = ws_p + (int64_t)strip * strip_bytes + (int64_t)lane * 16;
#pragma unroll 8
            for (int it = 0; it <= g_hi - g_lo; ++it)
                win[it * 64 + lane] = *reinterpret_cast<const uint4*>(base + (int64_t)(g_lo + it) * 1024);
        }
        __syncthreads();
        const uint8_t* wb = reinterpret_cast<const uint8_t*>(win);
        if (first) {                                              // start state, textSeqCompare.py:102
            st = ptr_pm(wb[(((k >> 2) - g_lo) * 64 + l) * 16 + (k & 3) * R + r]);
            first = false;
        }
        // walk while inside this window, a run at a time (nw_hw.h: walk_window_vec)
        const int cnt = walk_window_vec(win, g_lo, g_lo * 4, strip * L::SR, x, y, st, opsbuf, kTbOps, lane);
        __syncthreads();
        for (int i = lane; i < cnt; i += 64) ops[cap - 1 - (len + i)] = opsbuf[i];
        len += cnt;
        __syncthreads();
    }
    while (y > 0) { if (lane == 0) ops[cap - 1 - len] = 2; ++len; --y; }          // textSeqCompare.py:154-158
    while (x > 0) { if (lane == 0) ops[cap - 1 - len] = 1; ++len; --x; }          // textSeqCompare.py:160-164
    if (lane == 0) a.ops_len[p] = len;
}

}  // namespace ta

using namespace ta;

constexpr int kR = 4;          // rows per lane of the production kernel

extern "C" int64_t ta_nw_workspace_bytes(int32_t n, int32_t m) {
    if (n <= 0 || m <= 0) return 0;
    return WideWs<kR>(n, m).total;
}

extern "C" int32_t ta_nw_max_m(void) {
    // LDS per workgroup = 10 bytes per OCR token + ~2.5 KiB; 160 KiB per CU
    return 16000;
}

template <int W>
static hipError_t launch_fill(const NwArgs& a, int max_m, hipStream_t st) {
    const size_t lds = NwLds(max_m).total;
    // allow > 64 KiB of dynamic LDS: once per process, thread-safe (function-local static initialiser)
    static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(&nw_fill_kernel<kR, W, false>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (once != hipSuccess) return once;
    hipLaunchKernelGGL((nw_fill_kernel<kR, W, false>), dim3(a.nprob), dim3(W * 64), lds, st, a);
    return hipGetLastError();
}

// few tall problems: spread each over ceil(nstrips / kWideW) workgroups so that every strip has a
// SIMD of its own.  Block index = chunk * stride + p with stride a multiple of 8: workgroups are
// dealt round-robin to the 8 XCDs, so all chunks of a problem share one L2 for the HBM hand-off rows.
static hipError_t launch_fill_wide(NwArgs a, int nstrips, int max_m, hipStream_t st) {
    const size_t lds = NwLds(max_m).total;
    static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(&nw_fill_kernel<kR, kWideW, true>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (once != hipSuccess) return once;
    a.wide_stride = (a.nprob + 7) & ~7;
    const int chunks = (nstrips + kWideW - 1) / kWideW;
    hipLaunchKernelGGL((nw_wide_init_kernel<kR>), dim3(a.nprob), dim3(64), 0, st, a);
    hipLaunchKernelGGL((nw_fill_kernel<kR, kWideW, true>), dim3(a.wide_stride * chunks), dim3(kWideW * 64),
                       lds, st, a);
    return hipGetLastError();
}

extern "C" int ta_nw_batch(const int32_t* t_codes, const int64_t* t_off,
                           const int32_t* o_codes, const int64_t* o_off, int32_t nprob,
                           const int32_t* params, int32_t params_stride,
                           uint8_t* ws, const int64_t* ws_off,
                           uint8_t* ops_out, const int64_t* ops_off, int32_t* ops_len,
                           int32_t max_n, int32_t max_m, int64_t score_bound,
                           uint32_t flags, void* stream) {
    if (nprob < 0 || max_n < 0 || max_m < 0) return ta_fail(TA_EINVAL, "negative size");
    if (nprob == 0) return TA_OK;
    if (!t_off || !o_off || !params || !ws_off || !ops_off || !ops_len)
        return ta_fail(TA_EINVAL, "null pointer argument");
    if (params_stride != 0 && params_stride != 6) return ta_fail(TA_EINVAL, "params_stride must be 0 or 6");
    if (score_bound < 0 || score_bound >= (1ll << 23))
        return ta_fail(TA_ERANGE, "(n+m+2)*max|param| does not fit the 32-bit encoded scores");
    if (max_m > ta_nw_max_m()) return ta_fail(TA_ELIMIT, "m exceeds the LDS hand-off row capacity");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    NwArgs a{t_codes, t_off, o_codes, o_off, params, params_stride, ws, ws_off,
             ops_out, ops_off, ops_len, nprob};
    if (flags & TA_NW_FILL) {
        if (max_n > 0 && max_m > 0) {
            if (!t_codes || !o_codes || !ws) return ta_fail(TA_EINVAL, "null code/workspace pointer");
            const int nstrips = PtrLayout<kR>::nstrips(max_n);
            hipError_t e;
            const bool wide = (flags & TA_NW_WIDE) ? true : (flags & TA_NW_NARROW) ? false
                              : (nprob < 256 && nstrips > kWideW);
            if (wide && nstrips > kWideW) e = launch_fill_wide(a, nstrips, max_m, st);
            else if (nstrips >= 8) e = launch_fill<8>(a, max_m, st);
            else if (nstrips >= 4) e = launch_fill<4>(a, max_m, st);
            else if (nstrips >= 2) e = launch_fill<2>(a, max_m, st);
            else e = launch_fill<1>(a, max_m, st);
            if (e != hipSuccess) return ta_fail_hip(e, "nw_fill_kernel launch");
        }
    }
    if (flags & TA_NW_TRACEBACK) {
        if (!ops_out && (max_n + max_m) > 0) return ta_fail(TA_EINVAL, "null ops_out");
        hipLaunchKernelGGL((nw_traceback_kernel<kR>), dim3(nprob), dim3(64), 0, st, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return ta_fail_hip(e, "nw_traceback_kernel launch");
    }
    return TA_OK;
}
