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
#include <algorithm>
#include <atomic>
#include <type_traits>

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
    constexpr int CHK = (R == 4) ? 4 : 2;                  // 16 .. 32 steps of look-ahead (R = 4: 16, R = 2: 16, R = 1: 32)
    // export grain of the wide launch (groups between two copies of the bottom row to HBM): 32 .. 64 steps
    constexpr int XCHK = (R == 4) ? kCheck : (R == 2 ? 8 : 2);
    static_assert(R == 1 || R == 2 || R == 4, "rows per lane");
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
    // Non-positive gap opens (every system of the reference's grid): the strip state and the hand-off
    // rows are kept in CARRIED form, XG = V~ + gox and YG = H~ + goy, and the cell is
    // cell_update_carried_tagged (nw_cell.h: 13 instructions instead of 16, same pointers); with equal
    // gap opens one add fewer still.  Decided per problem (uniform in the workgroup).
    const bool carried = opens_nonpositive(c.gox, c.goy);
    const int form = carried ? (c.gox == c.goy ? 2 : 1) : 0;
    const int xadj6 = carried ? c.gox6 : 0, yadj6 = carried ? c.goy6 : 0;

    const NwLds lds(m);
    int2* hvd = reinterpret_cast<int2*>(smem) + kHvdPad;
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
    for (int j = tid; j <= m; j += W * 64) hvd[j] = make_int2(bnd_V_row0(c, j) + xadj6, bnd_D_row0(c, j));
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
    int exp_pending = 0;                          // progress value of an export whose stores are still in flight (WIDE)
    const int prev_wave = (wave + W - 1) % W;
    // groups [g_lo, g_hi) are "steady": every lane is inside 1 <= j <= m on every step
    const int g_lo = (63 + SPG - 1) / SPG;
    const int g_hi = m / SPG;
    // The start-up groups of a strip (lanes start one step apart) through the steady body instead of the EXEC-predicated
    // one: a lone problem's strips are a CHAIN -- a strip can start only when the one above is 62 steps + a block ahead,
    // so the 64 start-up steps of every strip are on the critical path, and the predicated body takes ~2.5 x a steady
    // step.  A lane that has not reached column 1 runs over VIRTUAL columns j <= 0 whose "mismatch" score is
    // -(1 + gex) (tag M): with the carried cell that leaves its column-0 boundary state exactly in place -- D = b_i | M
    // (M^ = b_(i-1) - (1 + gex) = b_i ties with YG = b_i and the M tag wins, as at the boundary), YG' = max(b_i + goy, b_i)
    // with the boundary's own tag, the XG chain of a virtual column (from -2^24 at the strip's edge) stays <= b_i + gox
    // because b falls with i iff gex <= -1 (the condition) -- and the D the DPP shift hands down is the b of the lane
    // above.  Pointer bytes of virtual cells are garbage nobody reads; lane 63's bottom-row writes for j <= 0 land in
    // the pad in front of the hand-off row.
    const bool from_zero = carried && c.gex <= -1 && g_lo + 2 < g_hi;
    const int cpad = ((-(1 + c.gex)) * 64) | kTagM;
    int pass = 0;

    for (int s = chunk * W + wave; s < nstrips; s += (WIDE ? nstrips : W), ++pass) {
        // ---- per-strip lane state: column-0 boundary (textSeqCompare.py:53-56) ----
        int D[R], V[R], H[R], tc[R];
        const int row0 = s * L::SR + lane * R;            // 0-based index of this lane's first row
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = row0 + r + 1;
            D[r] = bnd_D_col0(c, i);
            H[r] = bnd_H_col0(c, i) + yadj6;
            V[r] = from_zero ? -(1 << 30) : 0;
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
                // then pull every newly final column from its HBM row into the LDS hand-off row.
                // The row was stored write-through (sc1) and drained before the word was set, and it is
                // read here with sc1 loads by the wave that polled, after its poll matched: no acquire
                // fence (buffer_inv, ~1.7 us each, on the critical path of every workgroup's first strip).
                if (imp_hi >= col) return;
                const int row = s / W - 1;
                int have;
                while (true) {
                    have = __builtin_amdgcn_readfirstlane(
                        __hip_atomic_load(&gprog[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if (have >= need_groups) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       // no instruction: keeps the loads below the poll
                const int cfin = (have >= ngroups) ? m : min(m, have * SPG - 1 - 62);
                const unsigned long long* src = reinterpret_cast<const unsigned long long*>(xrows + (int64_t)row * wl.row_elems);
                for (int j = imp_hi + 1 + lane; j <= cfin; j += 64) {
                    const unsigned long long e = __hip_atomic_load(&src[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    hvd[j] = make_int2((int)(unsigned)e, (int)(unsigned)(e >> 32));
                }
                imp_hi = cfin;
                return;
            }
            // Strips of one workgroup hand their rows over in LDS, and the LDS executes a wave's operations
            // in order: a relaxed read of the progress word, then the entries.
            const int need = prod_pass * ngroups + need_groups;
            while (true) {
                const int have = __hip_atomic_load(&prog[prev_wave], __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_WORKGROUP);
                if (__builtin_amdgcn_readfirstlane(have) >= need) break;
                __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        };
        // progress is published after groups 0, CHK, 2 CHK, ...: the consumer's needs are 1 mod CHK
        auto publish = [&](int g) {
            if (WIDE && wave == W - 1) {
                if (((g % XCHK) == 0 || g == ngroups - 1) && s + 1 < nstrips) {
                    // Export: the newly final columns of this strip's bottom row go to the row in HBM with
                    // WRITE-THROUGH stores (8-byte sc1: no release fence, whose L2 write-back stalled this
                    // wave ~2 us per export).  The progress word may only follow stores that have completed;
                    // instead of draining the queue here (the group's pointer-byte store has just been
                    // issued), the word of the PREVIOUS export is set now: XCHK groups -- one vector-memory
                    // operation each -- have been issued since its stores, so once all but the XCHK / 2
                    // youngest operations are done, they are.  The last export of a strip drains.
                    int* const word = &gprog[s / W];
                    // The invariant this rests on, spelled out so that a change trips over it: between two exports
                    // this wave issues AT LEAST kVmemPerGroup vector-memory operations per group (today exactly one:
                    // the group's 16-byte pointer store; more per group only makes the wait stricter), vmcnt retires
                    // them in issue order, so "all but the XCHK / 2 youngest are done" implies the previous export's
                    // stores -- XCHK * kVmemPerGroup >= XCHK / 2 operations older -- are done.  Batching the pointer
                    // stores of several groups into one instruction would break it: lower XCHK / 2 with it.
                    constexpr int kVmemPerGroup = 1;
                    static_assert(XCHK % 2 == 0 && XCHK * kVmemPerGroup >= XCHK / 2 && XCHK / 2 >= 1 && XCHK / 2 <= 63,
                                  "the progress word of an export follows its stores by a counted vmcnt wait");
                    if (exp_pending > 0 && g != ngroups - 1) {    // (the last export may follow the one before closely: it drains)
                        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(XCHK / 2) : "memory");
                        if (lane == 0) __hip_atomic_store(word, exp_pending, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    // lane 63 has finished columns <= k - 62 of this strip's bottom row
                    const int cfin = (g == ngroups - 1) ? m : min(m, (g + 1) * SPG - 1 - 62);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // lane 63's entries are in LDS
                    unsigned long long* dst = reinterpret_cast<unsigned long long*>(xrows + (int64_t)(s / W) * wl.row_elems);
                    for (int j = exp_hi + 1 + lane; j <= cfin; j += 64) {
                        const int2 e = hvd[j];
                        __hip_atomic_store(&dst[j], ((unsigned long long)(unsigned)e.y << 32) | (unsigned)e.x,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    exp_hi = max(exp_hi, cfin);
                    exp_pending = g + 1;
                    if (g == ngroups - 1) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (lane == 0) __hip_atomic_store(word, g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        exp_pending = 0;
                    }
                }
                return;
            }
            if (W == 1) return;
            if ((g % CHK) == 0 || g == ngroups - 1) {
                // lane 63's hand-off entries and the progress word are LDS writes of one wave: in order.
                // (An atomic RELEASE store would also drain this wave's pointer-byte stores -- a round trip
                // to HBM per publish, which a lone wave on its SIMD has nothing to hide behind.)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 63)
                    __hip_atomic_store(&prog[wave], pass * ngroups + g + 1, __ATOMIC_RELAXED,
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
        // the strip's groups with the cell in the problem's form (FORM: 0 general, 1 carried, 2 carried
        // with one gap open)
        auto run = [&](auto form_tag) {
        constexpr int FORM = decltype(form_tag)::value;
        auto cell = [&](int d_ul, int x_u, int y_l, int t, int o, int& d, int& x, int& y) -> unsigned {
            if constexpr (FORM == 0) return cell_c(kr, d_ul, x_u, y_l, t, o, d, x, y);
            else return cell_carried_tagged_c<FORM == 2>(kr, d_ul, x_u, y_l, t, o, d, x, y);
        };
        // (start-up groups under from_zero: the mismatch score of a step is the virtual columns' while k < lane)
        auto cell_su = [&](int miss, int d_ul, int x_u, int y_l, int t, int o, int& d, int& x, int& y) -> unsigned {
            if constexpr (FORM == 0) return cell_c(kr, d_ul, x_u, y_l, t, o, d, x, y);
            else return cell_carried_tagged_miss<FORM == 2>(kr, miss, d_ul, x_u, y_l, t, o, d, x, y);
        };
        // 16 pointer bytes of a group (byte q * R + r: step q, row r) -> one 16-byte piece
        auto store_piece = [&](int gg, const unsigned (&bb)[16]) {
            *reinterpret_cast<uint4*>(out + (int64_t)gg * 1024) =
                make_uint4(pack4(bb[0], bb[1], bb[2], bb[3]), pack4(bb[4], bb[5], bb[6], bb[7]),
                           pack4(bb[8], bb[9], bb[10], bb[11]), pack4(bb[12], bb[13], bb[14], bb[15]));
        };
        // one group with per-lane activity tests (ramp-up, ramp-down, short rows)
        auto group_edge = [&](int g) {
            int oc[SPG];
            int2 hd[SPG];
#pragma unroll
            for (int q = 0; q < SPG; ++q) { oc[q] = oc_next[q]; hd[q] = hd_next[q]; }
            prefetch(g);
            unsigned bb[16];
#pragma unroll
            for (int x = 0; x < 16; ++x) bb[x] = 0u;
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
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int d_old = D[r];
                        bb[q * R + r] = cell(d_ul, v_u, H[r], tc[r], oc[q], D[r], V[r], H[r]);
                        d_ul = d_old;
                        v_u = V[r];
                    }
                    dsave = d_next;
                    if (lane == 63) hvd[j] = make_int2(V[R - 1], D[R - 1]);
                }
            }
            store_piece(g, bb);
            publish(g);
        };

        wait_span(0);
        load_group(0);
        int g = 0;
        const int e1 = from_zero ? 0 : min(g_lo, ngroups);
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
                unsigned bb[16];
                const bool su = from_zero && gg < g_lo;               // a start-up group: some lanes are in virtual columns
#pragma unroll
                for (int q = 0; q < SPG; ++q) {
                    int v_up = hd[q].x, d_next = hd[q].y;
                    wave_shr1_pair<1>(v_up, V[R - 1], d_next, D[R - 1]);
                    int d_ul = dsave, v_u = v_up;
                    const int miss = (su && gg * SPG + q < lane) ? cpad : kr.cmis;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int d_old = D[r];
                        bb[q * R + r] = cell_su(miss, d_ul, v_u, H[r], tc[r], oc[q], D[r], V[r], H[r]);
                        d_ul = d_old;
                        v_u = V[r];
                    }
                    dsave = d_next;
                    wptr[q] = make_int2(V[R - 1], D[R - 1]);
                }
                wptr += winc;
                store_piece(gg, bb);
                publish(gg);
            };
            // Blocks of CHK groups with running pointers: one progress wait and one publish per block at
            // fixed places in straight-line code, no per-group conditions, clamps or index arithmetic.  A
            // wave ALONE on its SIMD -- the latency-shaped launches, a few problems spread over the chip --
            // issues one instruction of ANY kind per ~5.5 cycles (tools/ubench/lone_wave.hip), so the ~25
            // scalar instructions per step that the per-group form spent on control were a quarter of
            // such a launch's critical path.
            if (from_zero) {                                        // the start-up groups, two at a time (g_lo is even)
                static_assert(((63 + SPG - 1) / SPG) % 2 == 0, "start-up groups in pairs");
                while (g < g_lo) {
                    fetch(g + 1, ocB, hdB);
                    steady(g, ocA, hdA);
                    fetch(g + 2, ocA, hdA);
                    steady(g + 1, ocB, hdB);
                    g += 2;
                }
            }
            if constexpr (CHK % 2 == 0) {
                if ((g % CHK) == 0 && g + CHK < g_hi) {
                    const uint16_t* ocp = ocode + (kOPad + (g + 1) * SPG - lane);   // codes of the next group to fetch
                    const int2* hvp = hvd + ((g + 1) * SPG + 1);                    // its hand-off entries: all <= m here
                    uint8_t* outp = out + (int64_t)g * 1024;
                    int* const pw = (lane == 63) ? &prog[wave] : reinterpret_cast<int*>(dummy + lane);
                    int pv = pass * ngroups + g + 1;
                    auto fetch_fast = [&](int (&oc)[SPG], int2 (&hd)[SPG]) {
#pragma unroll
                        for (int q = 0; q < SPG; ++q) { oc[q] = ocp[q]; hd[q] = hvp[q]; }
                        ocp += SPG; hvp += SPG;
                    };
                    auto steady_fast = [&](const int (&oc)[SPG], const int2 (&hd)[SPG]) {
                        unsigned bb[16];
#pragma unroll
                        for (int q = 0; q < SPG; ++q) {
                            int v_up = hd[q].x, d_next = hd[q].y;
                            wave_shr1_pair_sched(v_up, V[R - 1], d_next, D[R - 1]);
                            int d_ul = dsave, v_u = v_up;
#pragma unroll
                            for (int r = 0; r < R; ++r) {
                                const int d_old = D[r];
                                bb[q * R + r] = cell(d_ul, v_u, H[r], tc[r], oc[q], D[r], V[r], H[r]);
                                d_ul = d_old;
                                v_u = V[r];
                            }
                            dsave = d_next;
                            wptr[q] = make_int2(V[R - 1], D[R - 1]);
                        }
                        wptr += winc;
                        *reinterpret_cast<uint4*>(outp) =
                            make_uint4(pack4(bb[0], bb[1], bb[2], bb[3]), pack4(bb[4], bb[5], bb[6], bb[7]),
                                       pack4(bb[8], bb[9], bb[10], bb[11]), pack4(bb[12], bb[13], bb[14], bb[15]));
                        outp += 1024;
                    };
                    while (g + CHK < g_hi) {
                        wait_span(g);                       // covers the fetches of groups g + 1 .. g + CHK
                        fetch_fast(ocB, hdB);
                        steady_fast(ocA, hdA);
                        // progress after the block's first group (g = 0 mod CHK): the consumer's needs are 1 mod CHK
                        if (WIDE && wave == W - 1) publish(g);
                        else if (W > 1) {
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            __hip_atomic_store(pw, pv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        pv += CHK;
                        fetch_fast(ocA, hdA);
                        steady_fast(ocB, hdB);
#pragma unroll
                        for (int b2 = 2; b2 < CHK; b2 += 2) {
                            fetch_fast(ocB, hdB);
                            steady_fast(ocA, hdA);
                            fetch_fast(ocA, hdA);
                            steady_fast(ocB, hdB);
                        }
                        g += CHK;
                    }
                    // the per-group loop below waits when it FETCHES a group that is 0 mod CHK; group g was
                    // fetched by the last block without that wait, so take it here (covers g + 1 .. g + CHK)
                    wait_span(g);
                }
            }
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
        };   // run
        if (form == 2) run(std::integral_constant<int, 2>{});
        else if (form == 1) run(std::integral_constant<int, 1>{});
        else run(std::integral_constant<int, 0>{});
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
    using L = PtrLayout<R>;
    constexpr int SPG = L::SPG;
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
        const int g_hi = k / SPG;
        const int g_lo = max(0, g_hi - (kTbGroups - 1));
        {   // load the window: piece (g, lane) -> win[(g - g_lo) * 64 + lane].  A group's 64 pieces are 1 KiB
            // contiguous both in the workspace and in the window, which is exactly the shape of the
            // direct-to-LDS load (wave-uniform LDS base + 16 B per lane): no registers in between, so every
            // load of the window is in flight at once and the window costs ONE memory latency (batches of
            // eight through registers cost four, ~6 us per window of a wave that has nothing to hide them
            // behind; a 32-deep register array spilled).
            const uint8_t* base = ws_p + (int64_t)strip * strip_bytes + (int64_t)lane * 16;
            for (int it = 0; it <= g_hi - g_lo; ++it)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(base + (int64_t)(g_lo + it) * 1024),
                    (__attribute__((address_space(3))) void*)(win + it * 64), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        const uint8_t* wb = reinterpret_cast<const uint8_t*>(win);
        if (first) {                                              // start state, textSeqCompare.py:102
            st = ptr_pm(wb[(((k / SPG) - g_lo) * 64 + l) * 16 + (k % SPG) * R + r]);
            first = false;
        }
        // walk while inside this window, a run at a time (nw_hw.h: walk_window_vec)
        const int cnt = walk_window_vec<false, 64, false, R>(win, g_lo, g_lo * SPG, strip * L::SR, x, y, st,
                                                             opsbuf, kTbOps, lane);
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

// Rows per lane.  R = 4 (256-row strips) is the throughput shape: two DPP shifts and one 16-byte store per
// 16 cells.  A batch that cannot fill the chip with 256-row strips -- fewer than ~one strip per SIMD -- is
// bound by the latency of its wavefronts instead (a strip's steps are issued by ONE wave, one instruction
// per ~5.6 cycles, and a strip can start only ~86 steps behind the strip above): R = 2 (128-row strips)
// halves the work of a step and doubles the strips in flight.  Chosen per launch from the batch size;
// TA_NW_ROWS(r) in `flags` overrides (tests, timing).  R = 1 (64-row strips) is instantiated and tested but
// never chosen: one 4096^2 problem fills in 0.92 ms against 0.76 ms at R = 2 (the 62-step lag between
// strips is paid 64 times, and a step still costs ~30 instructions).
static int rows_per_lane(int nprob, int max_n, uint32_t flags) {
    const int forced = (int)((flags >> TA_NW_ROWS_SHIFT) & 0x7u);
    if (forced == 1 || forced == 2 || forced == 4) return forced;
    const int64_t strips4 = (int64_t)nprob * PtrLayout<4>::nstrips(max_n);
    return strips4 < 1024 ? 2 : 4;                 // 1024 SIMDs
}

template <int R>
static int64_t workspace_bytes_r(int n, int m) { return WideWs<R>(n, m).total; }

extern "C" int64_t ta_nw_workspace_bytes(int32_t n, int32_t m) {
    if (n <= 0 || m <= 0) return 0;
    const int64_t a = workspace_bytes_r<4>(n, m), b = workspace_bytes_r<2>(n, m), c = workspace_bytes_r<1>(n, m);
    return std::max(a, std::max(b, c));            // whichever strip height the launch picks
}

extern "C" int32_t ta_nw_max_m(void) {
    // LDS per workgroup = 10 bytes per OCR token + ~2.5 KiB; 160 KiB per CU
    return 16000;
}

template <int R, int W>
static hipError_t launch_fill(const NwArgs& a, int max_m, hipStream_t st) {
    const size_t lds = NwLds(max_m).total;
    hipError_t e = allow_full_lds(&nw_fill_kernel<R, W, false>);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((nw_fill_kernel<R, W, false>), dim3(a.nprob), dim3(W * 64), lds, st, a);
    return hipGetLastError();
}

// few tall problems: spread each over ceil(nstrips / kWideW) workgroups so that every strip has a
// SIMD of its own.  Block index = chunk * stride + p with stride a multiple of 8: workgroups are
// dealt round-robin to the 8 XCDs, so all chunks of a problem share one L2 for the HBM hand-off rows.
template <int R>
static hipError_t launch_fill_wide(NwArgs a, int nstrips, int max_m, hipStream_t st) {
    const size_t lds = NwLds(max_m).total;
    hipError_t e = allow_full_lds(&nw_fill_kernel<R, kWideW, true>);
    if (e != hipSuccess) return e;
    a.wide_stride = (a.nprob + 7) & ~7;
    const int chunks = (nstrips + kWideW - 1) / kWideW;
    hipLaunchKernelGGL((nw_wide_init_kernel<R>), dim3(a.nprob), dim3(64), 0, st, a);
    hipLaunchKernelGGL((nw_fill_kernel<R, kWideW, true>), dim3(a.wide_stride * chunks), dim3(kWideW * 64),
                       lds, st, a);
    return hipGetLastError();
}

template <int R>
static hipError_t launch_fill_r(const NwArgs& a, int max_n, int max_m, int nprob, uint32_t flags, hipStream_t st) {
    const int nstrips = PtrLayout<R>::nstrips(max_n);
    const bool wide = (flags & TA_NW_WIDE) ? true : (flags & TA_NW_NARROW) ? false
                      : (nprob < 256 && nstrips > kWideW);
    if (wide && nstrips > kWideW) return launch_fill_wide<R>(a, nstrips, max_m, st);
    if (nstrips >= 8) return launch_fill<R, 8>(a, max_m, st);
    if (nstrips >= 4) return launch_fill<R, 4>(a, max_m, st);
    if (nstrips >= 2) return launch_fill<R, 2>(a, max_m, st);
    return launch_fill<R, 1>(a, max_m, st);
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
    // fill and traceback of one batch must agree on the strip height: both derive it from the same
    // (nprob, max_n, flags), so a caller that issues them as two calls passes the same three
    const int rows = rows_per_lane(nprob, max_n, flags);
    if (flags & TA_NW_FILL) {
        if (max_n > 0 && max_m > 0) {
            if (!t_codes || !o_codes || !ws) return ta_fail(TA_EINVAL, "null code/workspace pointer");
            const hipError_t e = rows == 1 ? launch_fill_r<1>(a, max_n, max_m, nprob, flags, st)
                                 : rows == 2 ? launch_fill_r<2>(a, max_n, max_m, nprob, flags, st)
                                           : launch_fill_r<4>(a, max_n, max_m, nprob, flags, st);
            if (e != hipSuccess) return ta_fail_hip(e, "nw_fill_kernel launch");
        }
    }
    if (flags & TA_NW_TRACEBACK) {
        if (!ops_out && (max_n + max_m) > 0) return ta_fail(TA_EINVAL, "null ops_out");
        if (rows == 1) hipLaunchKernelGGL((nw_traceback_kernel<1>), dim3(nprob), dim3(64), 0, st, a);
        else if (rows == 2) hipLaunchKernelGGL((nw_traceback_kernel<2>), dim3(nprob), dim3(64), 0, st, a);
        else hipLaunchKernelGGL((nw_traceback_kernel<4>), dim3(nprob), dim3(64), 0, st, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return ta_fail_hip(e, "nw_traceback_kernel launch");
    }
    return TA_OK;
}
