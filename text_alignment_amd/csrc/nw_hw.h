// nw_hw.h -- device-side helpers shared by the NW kernels (ta_nw.hip, ta_nw2.hip): kernel
// argument block, single-instruction inline-asm helpers, LDS carve of the fill kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nw_cell.h"

namespace ta {

struct NwArgs {
    const int32_t* t_codes; const int64_t* t_off;
    const int32_t* o_codes; const int64_t* o_off;
    const int32_t* params; int32_t params_stride;
    uint8_t* ws; const int64_t* ws_off;
    uint8_t* ops_out; const int64_t* ops_off; int32_t* ops_len;
    int32_t nprob;
    int32_t wide_stride;          // wide one-pass launch only: block index = chunk * wide_stride + p
    int32_t apad;                 // phase 1 with a score profile in LDS: alphabet size + 1 (pad row)
};

// ---- single-instruction helpers.  Inline asm pins the instruction selection: left to
// itself hipcc un-folds the pre-shifted constants and splits max3 / bfi (25 VALU per cell
// instead of 16, see DESIGN.md "K1 instruction budget").  Non-volatile: each is a pure
// register op the compiler may schedule freely.
__device__ __forceinline__ int v_max3(int a, int b, int c) {
    int d;
    asm("v_max3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ int v_and_or_x(int a, int mask) {            // (a & mask) | kTagX
    int d;
    asm("v_and_or_b32 %0, %1, %2, 21" : "=v"(d) : "v"(a), "s"(mask));
    return d;
}
__device__ __forceinline__ unsigned v_bfi3(unsigned a, unsigned b) {    // (a & 3) | (b & ~3)
    unsigned d;
    asm("v_bfi_b32 %0, 3, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ unsigned v_bfi12(unsigned a, unsigned b) {   // (a & 12) | (b & ~12)
    unsigned d;
    asm("v_bfi_b32 %0, 12, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// t == o ? hit : miss   (textSeqCompare.py:32).  The lane mask goes to an SGPR pair of the
// compiler's choosing, not VCC: the compares of a group depend only on codes that are known
// before the group starts, so they can be hoisted off the max3 dependency chain, and the selects
// do not serialise on one mask register (5 % on the score-only cell, tools/ubench/valu_rate.hip).
__device__ __forceinline__ int v_score(int t, int o, int miss, int hit) {
    unsigned long long mask;
    int d;
    asm("v_cmp_eq_u32 %0, %1, %2" : "=s"(mask) : "v"(t), "v"(o));
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(d) : "v"(miss), "v"(hit), "s"(mask));
    return d;
}
// The carried score-only cell of phase 1 is written in plain C: hipcc selects v_max3_i32 for the
// nested max and folds the byte unpack into v_add_u32_sdwa (sext, BYTE_b) by itself, and -- unlike
// an asm statement -- needs no wait state between these instructions and their consumers.
// a + (signed byte b of `packed`): unpack of the LDS score profile's entry (slow-class issue like
// max3, but no v_cmp / v_cndmask per cell).  `b` is a constant after unrolling.
__device__ __forceinline__ int c_add_sbyte(int a, int packed, int b) {
    return a + (int)(int8_t)((unsigned)packed >> (8 * b));
}
__device__ __forceinline__ int c_max3(int a, int b, int c) { return max(max(a, b), c); }
// lane l receives lane l-1's value; lane 0 keeps what the destination held (DPP wave_shr:1,
// bound_ctrl off).  The leading s_nop covers the VALU-write -> DPP-read wait states (2) that
// hipcc cannot see across asm statements; NOPS = 4 also covers an EXEC write before a DPP.
template <int NOPS>
__device__ __forceinline__ void wave_shr1_pair(int& a_io, int a_src, int& b_io, int b_src) {
    asm volatile("s_nop %4\n\t"
                 "v_mov_b32_dpp %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %1, %3 wave_shr:1 row_mask:0xf bank_mask:0xf"
                 : "+v"(a_io), "+v"(b_io) : "v"(a_src), "v"(b_src), "n"(NOPS));
}

// The same shift through the compiler's DPP builtin: hipcc then sees the VALU-write -> DPP-read wait
// states itself and fills them with independent work of the next step where it has any.
__device__ __forceinline__ void wave_shr1_pair_sched(int& a_io, int a_src, int& b_io, int b_src) {
    a_io = __builtin_amdgcn_update_dpp(a_io, a_src, 0x138, 0xf, 0xf, false);      // 0x138 = wave_shr:1
    b_io = __builtin_amdgcn_update_dpp(b_io, b_src, 0x138, 0xf, 0xf, false);
}

constexpr int kOPad = 64;     // o-code padding in front (lanes that have not started yet)
constexpr int kOTail = 80;    // steps run to m + 62 (+ group round-up) past the last code
constexpr int kCheck = 16;    // hand-off progress is checked / published every kCheck groups

// LDS carve (dynamic): int2 pad[kHvdPad] hvd[m+2] | int2 dummy[64*4] | code ocode[kOPad+m+kOTail] | int prog[16]
//                      | uint32 profile[waves][apad][64]   (phase 1 with a score profile only)
constexpr int kHvdPad = 64;   // entries in front of the hand-off row: lane 63's bottom-row writes for the virtual columns
                              // j <= 0 of a strip's start-up groups land here (ta_nw.hip, from_zero)
struct NwLds {
    size_t hvd_bytes, dummy_bytes, oc_bytes, tbl_off, total;
    __host__ __device__ explicit NwLds(int m, int code_bytes = 2, int tbl_bytes = 0) {
        hvd_bytes = ((size_t)(kHvdPad + m + 2) * 8 + 15) & ~(size_t)15;
        dummy_bytes = 64 * 4 * 8;
        oc_bytes = ((size_t)(kOPad + m + kOTail) * code_bytes + 15) & ~(size_t)15;
        tbl_off = hvd_bytes + dummy_bytes + oc_bytes + 64;
        total = tbl_off + (size_t)tbl_bytes;
    }
};

// One interior cell on the encoded values: the arithmetic of ta::cell_update (nw_cell.h,
// checked on the CPU by the lane simulator), one VALU instruction per line.
struct CellRegs {
    int cmis, cmat, gox6, goy6, clean;    // clean = ~kTagMask, wave-uniform
};
__device__ __forceinline__ unsigned cell_hw(const CellRegs& k, int d_ul, int v_u, int h_l,
                                            int t, int o, int& d, int& v, int& h) {
    const int cs = v_score(t, o, k.cmis, k.cmat);        // v_cmp_eq + v_cndmask
    const int mr = (d_ul & k.clean) + cs;                 // v_and, v_add
    const int xr = v_and_or_x(v_u, k.clean);              // v_and_or
    const int yr = h_l & k.clean;                         // v_and
    const int xg = xr + k.gox6;                           // v_add
    const int yg = yr + k.goy6;                           // v_add
    d = v_max3(mr, xg, yg);
    v = v_max3(mr, xr, yg);
    h = v_max3(mr, xg, yr);
    return v_bfi3((unsigned)d_ul, v_bfi12((unsigned)v_u, (unsigned)h_l));
}

// the carried cell on encoded values (ta::cell_update_carried_tagged), one VALU instruction per line
template <bool SAMEGO = false>
__device__ __forceinline__ unsigned cell_carried_tagged_hw(const CellRegs& k, int d_ul, int xg_u, int yg_l,
                                                           int t, int o, int& d, int& xg, int& yg) {
    const int cs = v_score(t, o, k.cmis, k.cmat);        // v_cmp_eq + v_cndmask
    const int mr = (d_ul & k.clean) + cs;                 // v_and, v_add
    const int xr = v_and_or_x(xg_u, k.clean);             // v_and_or
    const int yr = yg_l & k.clean;                        // v_and
    d = v_max3(mr, xr, yr);
    const int dgx = d + k.gox6;                           // v_add
    const int dgy = SAMEGO ? dgx : d + k.goy6;            // v_add (none when the two gap opens are equal)
    xg = max(dgx, xr);                                    // v_max
    yg = max(dgy, yr);                                    // v_max
    return v_bfi3((unsigned)d_ul, v_bfi12((unsigned)xg_u, (unsigned)yg_l));
}

// The same cell in plain C.  hipcc selects v_and_or_b32 / v_max3_i32 / v_bfi_b32 for these forms by itself,
// and -- unlike between asm statements, where it pads every def-use with an s_nop (two per cell around the
// max3: issue slots a VALU-bound kernel pays for, and a lone wave pays 5.5 cycles each) -- schedules them
// without wait states.  The constants stay in registers (kr), so nothing is un-folded.
template <bool SAMEGO = false>
__device__ __forceinline__ unsigned cell_carried_tagged_c(const CellRegs& k, int d_ul, int xg_u, int yg_l,
                                                          int t, int o, int& d, int& xg, int& yg) {
    const int cs = (t == o) ? k.cmat : k.cmis;
    const int mr = (d_ul & k.clean) + cs;
    const int xr = (xg_u & k.clean) | kTagX;
    const int yr = yg_l & k.clean;
    d = max(max(mr, xr), yr);
    const int dgx = d + k.gox6;
    const int dgy = SAMEGO ? dgx : d + k.goy6;
    xg = max(dgx, xr);
    yg = max(dgy, yr);
    // (the pointer byte through the asm v_bfi_b32 helpers: hipcc would split each into two v_and + an or;
    // their results are not needed before the group's bytes are packed, so no pad lands behind them)
    return v_bfi3((unsigned)d_ul, v_bfi12((unsigned)xg_u, (unsigned)yg_l));
}
// the same with the mismatch score given per call: the virtual columns j <= 0 of a strip's start-up groups score
// -(1 + gex) (encoded, tag M), which keeps a lane's column-0 boundary state in place (ta_nw.hip, from_zero)
template <bool SAMEGO = false>
__device__ __forceinline__ unsigned cell_carried_tagged_miss(const CellRegs& k, int miss, int d_ul, int xg_u, int yg_l,
                                                             int t, int o, int& d, int& xg, int& yg) {
    const int cs = (t == o) ? k.cmat : miss;
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
__device__ __forceinline__ unsigned cell_c(const CellRegs& k, int d_ul, int v_u, int h_l, int t, int o,
                                           int& d, int& v, int& h) {
    const int cs = (t == o) ? k.cmat : k.cmis;
    const int mr = (d_ul & k.clean) + cs;
    const int xr = (v_u & k.clean) | kTagX;
    const int yr = h_l & k.clean;
    const int xg = xr + k.gox6;
    const int yg = yr + k.goy6;
    d = max(max(mr, xg), yg);
    v = max(max(mr, xr), yg);
    h = max(max(mr, xg), yr);
    return v_bfi3((unsigned)d_ul, v_bfi12((unsigned)v_u, (unsigned)h_l));
}

// pack the low bytes of four values into one dword (3 v_perm_b32)
__device__ __forceinline__ unsigned pack4(unsigned b0, unsigned b1, unsigned b2, unsigned b3) {
    const unsigned lo = __builtin_amdgcn_perm(b1, b0, 0x0C0C0400u);
    const unsigned hi = __builtin_amdgcn_perm(b3, b2, 0x0C0C0400u);
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}

// ---- vectorised traceback walk inside an LDS-staged window of the strip layout (R = 4) ----
// textSeqCompare.py:110-145.  In state st a step emits op = st, moves up unless st == 2 and left
// unless st == 1, and the next state is the 2-bit pointer field of the cell's byte that belongs to
// st (bits 2*st, 2*st+1; field f means state 2 - f).  Alignments are made of RUNS -- matches stay
// in state 0 along a diagonal, gaps in state 1 / 2 along a column / row -- so instead of one
// dependent step at a time, lane i looks at the cell i steps further along the current direction
// and reports whether the walk would still be in state st there; one ballot gives the run
// length, and the whole run (plus the step that changes state) is taken at once.
//
// win: window of 16-byte pieces [(group - gw_lo) * 64 + lane]; x_lo: row index just above the
// strip (cells with x > x_lo are in it); klow: smallest valid skewed step.  Returns the number
// of ops appended to opsbuf; updates x, y, st.  The walk stops when it leaves the strip, the
// valid steps (klow, in steps), the table (x == 0 or y == 0) or after max_ops.
//
// TOP_PENDING (two-phase aligner): the window's first-row cells hold no PM / PX (the row above came
// without winner tags).  A step that leaves a strip below the first one upwards is still taken,
// and ends the walk with st = 3 + (the state it was taken in): the caller resolves it.
//
// WL < 64 (two-phase aligner): the window holds only the lanes l_lo .. l_lo + WL - 1 of every group
// (pieces at [(group - gw_lo) * WL + (lane - l_lo)]); the walk also stops when it needs a lane above
// l_lo (the caller re-fills around the new position).  WL = 64: whole strips, l_lo = 0.
//
// OPS_REV: opsbuf points at the slot of the FIRST column this call emits in the caller's
// right-aligned output and columns go to descending addresses (opsbuf[-i]): the walk writes the
// alignment straight to memory, no staging buffer.  Otherwise opsbuf[i], ascending (LDS staging).
// R: rows per lane of the strip layout the window is in (PtrLayout<R>: 16 / R steps per 16-byte piece).
template <bool TOP_PENDING = false, int WL = 64, bool OPS_REV = false, int R = 4>
__device__ __forceinline__ int walk_window_vec(const uint4* win, int gw_lo, int klow, int x_lo,
                                               int& x, int& y, int& st, uint8_t* opsbuf, int max_ops,
                                               int lane, long long* iterations = nullptr, int l_lo = 0) {
    const uint8_t* wb = reinterpret_cast<const uint8_t*>(win);
    constexpr int SPG = 16 / R;                                   // steps per piece
    constexpr int RSH = (R == 4) ? 2 : (R == 2 ? 1 : 0), SSH = (SPG == 4) ? 2 : (SPG == 8 ? 3 : 4);
    static_assert(R == 1 || R == 2 || R == 4, "rows per lane");
    int cnt = 0;
    while (true) {
        if (iterations) ++*iterations;
        const int up = (st != 2), left = (st != 1);
        const int xi = x - lane * up, yi = y - lane * left;
        const int li = (xi - 1 - x_lo) >> RSH;
        const int ki = (yi - 1) + li;
        const bool valid = (xi > x_lo) & (yi > 0) & (ki >= klow) & (WL == 64 || li >= l_lo);
        unsigned b = 0;
        if (valid) b = wb[((ki >> SSH) - gw_lo) * (WL * 16) + (li - l_lo) * 16 + (ki & (SPG - 1)) * R + ((xi - 1) & (R - 1))];
        int nxt = 2 - (int)((b >> (2 * st)) & 3u);
        if (TOP_PENDING && up && x_lo > 0 && xi == x_lo + 1) nxt = 3 + st;
        const unsigned long long vmask = __ballot(valid);
        if ((vmask & 1ull) == 0ull) break;                       // lane 0 (the current cell) is out
        const unsigned long long cmask = __ballot(valid && nxt == st);
        int run = (~cmask == 0ull) ? 64 : (int)__builtin_ctzll(~cmask);   // lanes 0..run-1 stay in st
        int steps = run, st_new = st;
        if (run < 64 && ((vmask >> run) & 1ull)) {               // the step that leaves state st
            steps = run + 1;
            st_new = __builtin_amdgcn_readlane(nxt, run);
        }
        steps = min(steps, max_ops - cnt);
        if (steps < run + 1) st_new = st;                         // truncated inside the run
        if (lane < steps) {
            if (OPS_REV) opsbuf[-(cnt + lane)] = (uint8_t)st;
            else opsbuf[cnt + lane] = (uint8_t)st;
        }
        cnt += steps;
        x -= steps * up;
        y -= steps * left;
        st = st_new;
        if (cnt >= max_ops) break;
    }
    return cnt;
}

}  // namespace ta
