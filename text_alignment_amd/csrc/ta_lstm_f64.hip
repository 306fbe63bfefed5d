// ta_lstm_f64.hip -- the line recogniser's recurrence in FLOAT64 on the f64 matrix cores of MI355X (gfx950).
//
// Why: the reference's recogniser (`ocropus-rpred`, reference alignToOCR.py:142-147; arithmetic of the
// third-party ocropy 1.3.3) computes in float64 numpy.  The random-weight model of
// SURVEY.md section 8(d) is chaotic: a float32 rounding difference grows up to ten-thousandfold over a line of
// 800 .. 2000 columns, so no float32 recurrence (ta_lstm.hip, either mode) holds "logits within 1e-3" free-running
// on every line.  This mode accumulates, carries state and evaluates the gate functions in float64.
//
// One direction's float64 weights (4 gates x 100 units x 149 inputs = 477 KB) do not fit a CU, so the product is
// split where the recurrence allows it:
//
//  Kx lstm_xproj_f64_kernel   Gx[dir][row][gx_index(unit, gate)] = W[:, :49] . [1; x_row] for EVERY row of the batch at
//                             once (no dependence between timesteps): a weight-stationary f64 GEMM on
//                             v_mfma_f64_16x16x4_f64, rows x 48 x 400 per direction (+ the bias the accumulators start from).
//  Kr lstm_seq_f64_kernel     one workgroup of FOUR waves (one per SIMD, 512 registers each) per (16 lines,
//                             direction).  The recurrent weights W[:, 49:] (4 x 100 x 100 float64 = 320 KB) live in
//                             the CU's registers as A fragments for the whole kernel (the first 120 doubles of a wave
//                             pinned in AGPRs, which the MFMA reads itself); h_{t-1} of the 16 lines is the B operand,
//                             kept in LDS (float64) and read into registers once per step; per step the accumulators
//                             start from Gx (the loads are issued a step ahead) and take 25 k-steps of
//                             v_mfma_f64_16x16x4_f64; cell state and gate functions in float64 (own exp: range
//                             reduction + degree-9 polynomial, 1.9e-14).
//  Kr4 lstm_seq4_f64_kernel   (round 5; the product's choice) the same recurrence, bit for bit, on groups of FOUR lines:
//                             eight waves (two per SIMD, 256 registers each), v_mfma_f64_4x4x4_4b_f64, the gates of a cell
//                             brought into one lane by the gfx950 row swaps -- see the comment at the kernel.
//
// Tiling of the 400 pre-activations of a step: 25 tiles of 16 = (4 units) x (4 gates).  The WEIGHTS are the A operand
// (M = 16 tile rows, row i = 4 * gate + unit-in-tile), the 16 lines the B operand's columns.  The f64 MFMA returns
// D[i][j] in lane j + 16 (i mod 4), register i / 4 (tools/ubench/mfma_f64.hip): lane (j, q) therefore holds, in its
// four accumulator registers, the four GATES of (line j, unit 4 tile + q) -- the cell update needs no exchange between
// lanes, and the four values are 32 contiguous bytes of Gx (two 16-byte loads / stores).  No padding: 25 tiles x 25
// k-steps are exactly 400 x 100.  Every wave owns six tiles; the 25th is split along k between waves 1..3 and summed by
// wave 0 (seq_f64_body).  In the projection kernel waves take 7, 6, 6, 6 column tiles.
//
// Measured (tools/ubench/mfma_f64.hip, mfma_f64_ops.hip, cell_f64.hip; profiles/r04_mfma_f64.txt, r04_mfma_f64_ops.txt,
// r04_cell_f64.txt): 64 cycles per MFMA per SIMD (= the 78.6 TF float64 peak), the same for one dependent accumulator
// chain; the matrix instruction holds the SIMD's vector issue while it runs -- a wave's v_fma_f64 beside another
// wave's f64 MFMAs gets one issue per 69 cycles, and any VALU instruction between two MFMAs of a chain costs its full
// ~6 .. 11 cycles (LDS reads do not) -- so the gate math cannot hide under the MFMAs and a step costs MFMA time + gate
// time: (25 x 1623 + 25 x ~700) / 4 SIMDs = 14 500 cycles; the kernel takes 15 600.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ta_common.h"

namespace ta64 {

constexpr int kNi = 48;
constexpr int kNs = 100;
constexpr int kLines = 16;          // lines per workgroup (MFMA M)
constexpr int kCols = 4 * kNs;      // 400 pre-activations per step, column = 4 * unit + gate
constexpr int kTiles = kCols / 16;  // 25 column tiles
constexpr int kKH = kNs / 4;        // 25 k-steps over h
constexpr int kKXM = kNi / 4;       // projection: 12 k-steps over x; the bias is what the accumulators start from
constexpr int kKX = kKXM + 1;       // fragments per tile in the packed input weights: the 12 k-steps + the bias fragment
constexpr int kW = 4;               // waves per workgroup
constexpr int kMaxNT = 7;           // tile slots per wave in the packed recurrent weights: six own tiles + the 25th
constexpr int kOwnTiles = 6;        // recurrence: tiles a wave owns (6 wave .. 6 wave + 5); the 25th is split along k

typedef double f64x4 __attribute__((ext_vector_type(4)));

#ifdef TA_F64_PROFILE
#ifndef TA_F64_PROF_WAVE
#define TA_F64_PROF_WAVE 0          // the four-line kernel: the wave that reports ([0] MFMA phase, [1] transpose + cell update + stores, [2] barrier)
#endif
// -DTA_F64_PROFILE: cycle counters of wave 0 of every workgroup (tools/f64_time.py): [0] tiles' MFMAs + cell update,
// [1] h to LDS / output store / next accumulator loads, [2] the step's barrier, [3] steps
__device__ unsigned long long g_prof[4];
__device__ __forceinline__ unsigned long long prof_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
#endif

// Gx layout inside a row (400 doubles): per tile of 4 units 16 doubles = [gate pair 2][unit-in-tile 4][2]: a lane owns
// the four gates of one (row, unit) and moves them with two 16-byte accesses; the four lanes of a row's tile make each
// access a contiguous 64-byte piece.
__host__ __device__ constexpr int gx_index(int unit, int gate) { return 16 * (unit / 4) + 8 * (gate / 2) + 2 * (unit % 4) + (gate % 2); }
// The projection's column tiles per wave: 0..6, 7..12, 13..18, 19..24 -- four waves, one per SIMD, ~290 registers each.
// (Giving the waves the 25th tile in turn, row tile by row tile, measured SLOWER: 6.1 against 5.4 ms per 2.76 M rows.
// Round 6, -DTA_XPROJ_WAVES=8: EIGHT waves of 4 / 3 tiles at 180 registers, two per SIMD, so that one wave's row stores
// could issue under its SIMD partner's MFMAs -- bit-identical Gx, 6.25 against 5.17 ms: slower.  A wave then writes
// 384 .. 512 contiguous bytes of a row instead of 896 and every row tile's x values are fetched and converted by twice
// as many waves; the kernel is bound by what the memory system takes for this write pattern (3.4 TB/s of the 4.7 TB/s
// a plain fill reaches), not by its own issue port.  The switch stays for A/B builds: tools/f64_variants.sh.)
#ifndef TA_XPROJ_WAVES
#define TA_XPROJ_WAVES 4
#endif
#ifndef TA_XPROJ_ORDER
#define TA_XPROJ_ORDER 1
#endif
constexpr int kXW = TA_XPROJ_WAVES;
__host__ __device__ constexpr int tile0_of(int wave) { return kXW == 8 ? (wave == 0 ? 0 : 1 + 3 * wave) : (wave == 0 ? 0 : 1 + 6 * wave); }
__host__ __device__ constexpr int ntiles_of(int wave) { return kXW == 8 ? (wave == 0 ? 4 : 3) : (wave == 0 ? 7 : 6); }

// ---------------------------------------------------------------------------------------------
// float64 gate functions.  exp: x = n ln2 + r, |r| <= ln2 / 2; e^r by a degree-7 polynomial -- the interpolant at the
// Chebyshev nodes of the interval, 5.5e-11 relative (round 6; rounds 4-5 carried degree 9 / 10 at 1.9e-14 / 3e-13).  What the
// path is asked for is logits within 1e-3 of the float64 restatement; this model amplifies a perturbation of the recurrence
// by up to 1e4 over a line, so 5.5e-11 reaches the logits as < 1e-6 -- below the 1e-5 the float32 output
// layer contributes -- and two multiply-adds fewer per exponential are ten fewer float64 VALU instructions per cell, on a
// SIMD whose f64 MFMAs and f64 VALU do not overlap.  |x| <= 40 where the scaled result is built by hand (the callers
// clamp), so n ln2 is exact enough with ln2 as ONE double (|n| <= 58: 1.3e-15).  SCALE = 2 evaluates e^(2 x') from
// x' = x / 2 (the tanh's e^(-2 |c|)): the same reduction on r / 2 with the coefficients scaled by powers of two, which is
// exact -- no instruction for the doubling.
#ifndef TA_F64_RCP_NEWTON
#define TA_F64_RCP_NEWTON 1
#endif
// The cell update is a LATENCY chain, not an instruction count: a float64 VALU instruction issues in 4 cycles but its
// result is ready for a dependent one after ~8, the exponentials are Horner chains, and while a wave updates its cells its
// SIMD partner is in its MFMA phase (which holds the vector issue) -- nobody fills the bubbles.  Round 5's 160 instructions
// took 680 cycles on a wave alone; cutting them to 129 in the same order took 704 (tools/ubench/cell_f64.hip,
// profiles/r06_cell_f64.txt).  So the three exponentials that do not depend on each other (input gate, forget gate, tanh of
// the candidate) are evaluated IN LOCKSTEP, level by level -- three independent instructions per level cover the latency --
// and so are the two of the output half (tanh of the new state, output gate).  Each level is one asm block: the order is
// the point, and left to itself the compiler emits the chains one after the other.
struct ExpConsts {                   // c_k S^k of the degree-7 interpolant, for SCALE = S
    double c0, c1, c2, c3, c4, c5, c6, c7, log2e_s, mln2_s;
};
template <int SCALE> __device__ __forceinline__ constexpr ExpConsts exp_consts() {
    constexpr double s1 = SCALE, s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2, s5 = s4 * s1, s6 = s4 * s2, s7 = s4 * s3;
    return {0.9999999999595618, 0.999999999995509 * s1, 0.5000000107729166 * s2, 0.16666666786308587 * s3,
            0.041666218319291945 * s4, 0.008333283538708528 * s5, 0.0013948578326459795 * s6, 0.00019907569310848288 * s7,
            SCALE * 1.4426950408889634074, -0.69314718055994530942 / SCALE};
}
constexpr double kExpMagic = 6755399441055744.0;                          // 1.5 * 2^52: t = x log2e + magic has n = rint(..) in its low dword
// One level for THREE chains (A with the SCALE = 2 constant, B and C with the SCALE = 1 one): d = d * r + c
#define TA_LEVEL3(da, db, dc, ra, rb, rc, ca, cbc)                                                                        \
    asm("v_fma_f64 %0, %0, %3, %6\n\tv_fma_f64 %1, %1, %4, %7\n\tv_fma_f64 %2, %2, %5, %7"                               \
        : "+v"(da), "+v"(db), "+v"(dc) : "v"(ra), "v"(rb), "v"(rc), "s"(ca), "s"(cbc))
#define TA_LEVEL2(da, db, ra, rb, ca, cb)                                                                                 \
    asm("v_fma_f64 %0, %0, %2, %4\n\tv_fma_f64 %1, %1, %3, %5" : "+v"(da), "+v"(db) : "v"(ra), "v"(rb), "s"(ca), "s"(cb))
// 2^n into the exponent field of p (p in [0.70, 1.42], |n| <= 29: a clamped sigmoid argument) with ONE 32-bit instruction
__device__ __forceinline__ double scale_small_f64(double p, double t) {
    const unsigned long long pb = __builtin_bit_cast(unsigned long long, p);
    unsigned hi;
    asm("v_lshl_add_u32 %0, %1, 20, %2" : "=v"(hi) : "v"((unsigned)__builtin_bit_cast(unsigned long long, t)), "v"((unsigned)(pb >> 32)));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (pb & 0xffffffffull));
}
// ea = e^(2 xa) for ANY xa <= 0 (the tanh's argument: unclamped, the hardware scales, underflow included);
// eb = e^xb, ec = e^xc for |xb|, |xc| <= 20 (clamped sigmoid arguments)
__device__ __forceinline__ void exp3_f64(double xa, double xb, double xc, double& ea, double& eb, double& ec) {
    constexpr ExpConsts A = exp_consts<2>(), B = exp_consts<1>();
    double magic = kExpMagic, ta, tb, tc;
    asm("v_fma_f64 %0, %3, %6, %8\n\tv_fma_f64 %1, %4, %7, %8\n\tv_fma_f64 %2, %5, %7, %8"
        : "=&v"(ta), "=&v"(tb), "=&v"(tc) : "v"(xa), "v"(xb), "v"(xc), "s"(A.log2e_s), "s"(B.log2e_s), "v"(magic));
    double na, nb, nc;
    asm("v_add_f64 %0, %3, %6\n\tv_add_f64 %1, %4, %6\n\tv_add_f64 %2, %5, %6"
        : "=&v"(na), "=&v"(nb), "=&v"(nc) : "v"(ta), "v"(tb), "v"(tc), "s"(-kExpMagic));
    double ra, rb, rc;                                                      // r = (S x - n ln2) / S
    asm("v_fma_f64 %0, %3, %9, %6\n\tv_fma_f64 %1, %4, %10, %7\n\tv_fma_f64 %2, %5, %10, %8"
        : "=&v"(ra), "=&v"(rb), "=&v"(rc) : "v"(na), "v"(nb), "v"(nc), "v"(xa), "v"(xb), "v"(xc), "s"(A.mln2_s), "s"(B.mln2_s));
    double c6a = A.c6, c6b = B.c6, pa, pb, pc;
    asm("v_fma_f64 %0, %3, %6, %8\n\tv_fma_f64 %1, %4, %7, %9\n\tv_fma_f64 %2, %5, %7, %9"
        : "=&v"(pa), "=&v"(pb), "=&v"(pc) : "v"(ra), "v"(rb), "v"(rc), "s"(A.c7), "s"(B.c7), "v"(c6a), "v"(c6b));
    TA_LEVEL3(pa, pb, pc, ra, rb, rc, A.c5, B.c5);
    TA_LEVEL3(pa, pb, pc, ra, rb, rc, A.c4, B.c4);
    TA_LEVEL3(pa, pb, pc, ra, rb, rc, A.c3, B.c3);
    TA_LEVEL3(pa, pb, pc, ra, rb, rc, A.c2, B.c2);
    TA_LEVEL3(pa, pb, pc, ra, rb, rc, A.c1, B.c1);
    TA_LEVEL3(pa, pb, pc, ra, rb, rc, A.c0, B.c0);
    ea = __builtin_ldexp(pa, (int)__builtin_bit_cast(long long, ta));
    eb = scale_small_f64(pb, tb);
    ec = scale_small_f64(pc, tc);
}
// the same for the two exponentials of the output half: ea = e^(2 xa), xa <= 0 unclamped; eb = e^xb, |xb| <= 20
__device__ __forceinline__ void exp2_f64(double xa, double xb, double& ea, double& eb) {
    constexpr ExpConsts A = exp_consts<2>(), B = exp_consts<1>();
    double magic = kExpMagic, ta, tb;
    asm("v_fma_f64 %0, %2, %4, %6\n\tv_fma_f64 %1, %3, %5, %6"
        : "=&v"(ta), "=&v"(tb) : "v"(xa), "v"(xb), "s"(A.log2e_s), "s"(B.log2e_s), "v"(magic));
    double na, nb;
    asm("v_add_f64 %0, %2, %4\n\tv_add_f64 %1, %3, %4" : "=&v"(na), "=&v"(nb) : "v"(ta), "v"(tb), "s"(-kExpMagic));
    double ra, rb;
    asm("v_fma_f64 %0, %2, %6, %4\n\tv_fma_f64 %1, %3, %7, %5"
        : "=&v"(ra), "=&v"(rb) : "v"(na), "v"(nb), "v"(xa), "v"(xb), "s"(A.mln2_s), "s"(B.mln2_s));
    double c6a = A.c6, c6b = B.c6, pa, pb;
    asm("v_fma_f64 %0, %2, %4, %6\n\tv_fma_f64 %1, %3, %5, %7"
        : "=&v"(pa), "=&v"(pb) : "v"(ra), "v"(rb), "s"(A.c7), "s"(B.c7), "v"(c6a), "v"(c6b));
    TA_LEVEL2(pa, pb, ra, rb, A.c5, B.c5);
    TA_LEVEL2(pa, pb, ra, rb, A.c4, B.c4);
    TA_LEVEL2(pa, pb, ra, rb, A.c3, B.c3);
    TA_LEVEL2(pa, pb, ra, rb, A.c2, B.c2);
    TA_LEVEL2(pa, pb, ra, rb, A.c1, B.c1);
    TA_LEVEL2(pa, pb, ra, rb, A.c0, B.c0);
    ea = __builtin_ldexp(pa, (int)__builtin_bit_cast(long long, ta));
    eb = scale_small_f64(pb, tb);
}
// e^x alone (tools/ubench/cell_f64.hip checks it against long double; the cell update itself uses the lockstep forms)
__device__ __forceinline__ double exp_f64(double x) {
    double ea, eb;
    exp2_f64(-1.0, x, ea, eb);
    return eb;
}
// 1 / d for d in [1, 1e40]: the hardware reciprocal (measured 4.5e-8 relative: tools/ubench/cell_f64.hip,
// profiles/r05_cell_f64.txt) refined by ONE Newton step (quadratic: 2e-15)
__device__ __forceinline__ double rcp_f64(double d) {
#pragma clang fp contract(off)
    double y = __builtin_amdgcn_rcp(d);
#pragma unroll
    for (int i = 0; i < TA_F64_RCP_NEWTON; ++i) {
        const double e = __builtin_fma(-d, y, 1.0);
        y = __builtin_fma(y, e, y);
    }
    return y;
}
__device__ __forceinline__ double clamp20(double v) { return __builtin_fmin(__builtin_fmax(v, -20.0), 20.0); }
// ocropy's sigmoid is 1 / (1 + exp(clip(-x, -20, 20)))   -- in float64 the clip is visible
// (sigma(-25) = 2.06e-9 with it, 1.4e-11 without), so it is kept.  tanh(x) = sign(x) (1 - e) / (1 + e) with e = exp(-2 |x|):
// e <= 1 for every x, so nothing needs clamping (numpy's tanh is not clipped either; round 5 clamped |x| to 20 at two
// float64 min / max per tanh, and the compiler added a canonicalising v_max_f64 in front of each); the sign goes onto the
// numerator with one v_bfi_b32.

// One LSTM cell update (`forward_py`) from the four pre-activations of a (line, unit) pair.
// c: the cell state, ZERO before the first step of a sequence (the input / forget peepholes and the old state's share then
// vanish by themselves); wop_t: the output peephole, zero at the first step of the whole sequence (skipped at t = 0).
// Every operation is spelled out (no contraction left to the compiler): the 16-line and the 4-line kernel round alike.
// A NaN in ci_pre (how a wait that ran out poisons a cell, see the kernels) reaches c and h: that path has no min / max.
__device__ __forceinline__ double lstm_cell_f64(double gi, double gf, double go, double ci_pre, double& c,
                                                double wip, double wfp, double wop_t) {
#pragma clang fp contract(off)
    // every gate is a ratio with denominator 1 + e^z; the three of the cell-state update share ONE reciprocal
    // (of the product of their denominators, at most 2 (1 + e^20)^2 ~ 5e17), the two of the output another
    const double cp = c;
    double ea, eb, ef;     // tanh(ci_pre) = sign (1 - ea) / (1 + ea);  sigma = 1 / (1 + e^clip(-z)), ocropy's clip
    exp3_f64(-__builtin_fabs(ci_pre), clamp20(-__builtin_fma(wip, cp, gi)), clamp20(-__builtin_fma(wfp, cp, gf)), ea, eb, ef);
    const double da = 1.0 + ea;
    const double pa = __builtin_fma(da, eb, da), pf = 1.0 + ef;                  // (1 + ea) (1 + eb)
    const double r3 = rcp_f64(pa * pf);
    const double cn = __builtin_fma(__builtin_copysign(1.0 - ea, ci_pre) * pf, r3, (pa * r3) * cp);   // ci * gi + gf * c
    double ec, eo;
    exp2_f64(-__builtin_fabs(cn), clamp20(-__builtin_fma(wop_t, cn, go)), ec, eo);
    const double dc = 1.0 + ec;
    c = cn;
    return __builtin_copysign(1.0 - ec, cn) * rcp_f64(__builtin_fma(dc, eo, dc));                     // tanh(c) * go
}

// ---------------------------------------------------------------------------------------------
// Kx: Gx[dir][row][4 unit + gate] = bias + sum_k W_gate[unit][1 + k] * x_row[k]     (k < 48, float64)
// The accumulators START from the bias and twelve k-steps run over x alone.  Until round 6 the constant 1 was input 0 of a
// 52-wide padded row (13 k-steps): its first fma was fma(1, b, 0) = b and its last three added 0 * 0 -- the chain of
// fmas over x1 .. x48 in between is the same one, so Gx is the same to the bit (but for the sign of an exact zero) with
// a thirteenth of the matrix instructions gone.
struct XprojArgs {
    const float* x;        // [rows][48], the rows of this call
    int64_t rows;
    const double* wx;      // [dir 2][tile 25][slot 13][lane 64]: B fragments, column j = lane % 16 of the tile = the (unit, gate) whose
                           // gx_index is 16 tile + j: slots 0..11 W_gate(2 (j / 8) + j % 2)[unit 4 tile + (j % 8) / 2][1 + 4 slot + lane / 16]
                           // (the 48 weights of x), slot 12 the column's BIAS W_gate(..)[unit ..][0] in every lane of the column
    double* gx;            // [dir 2][rows][400], gx_index(unit, gate) inside a row
};
typedef double f64x2 __attribute__((ext_vector_type(2)));

// Here the ROWS are the A operand and the weights the B operand: D[i][j] = (row i, column j) sits in lane j + 16 (i % 4),
// register i / 4, so one store instruction writes, for four rows, the 16 doubles of a tile row -- 128 contiguous bytes
// each (with the weights as A, as in the recurrence, a lane would own 32 bytes of a row and every store instruction
// would write half lines: measured 6.5 against 5.4 ms per 2.76 M rows).
template <int NT>
__device__ __forceinline__ void xproj_body(const XprojArgs& a, int dir, int wave, int lane) {
    const int tile0 = tile0_of(wave);
    double Bx[NT][kKXM], bias[NT];
    {
        const double* wp = a.wx + ((size_t)(dir * kTiles + tile0) * kKX) * 64 + lane;
#pragma unroll
        for (int s = 0; s < NT; ++s) {
#pragma unroll
            for (int kk = 0; kk < kKXM; ++kk) Bx[s][kk] = wp[((size_t)s * kKX + kk) * 64];
            bias[s] = wp[((size_t)s * kKX + kKXM) * 64];
        }
    }
    const int64_t ntiles = (a.rows + 15) / 16;
    const int li = lane & 15, kq = lane >> 4;
    double* gxd = a.gx + (size_t)dir * a.rows * kCols;
    // this lane's A values of a row tile: x[4 kk + kq] of row tile * 16 + li (rows past the end: clamped)
    auto load_tile = [&](int64_t tile, float (&dst)[kKXM]) {
        const int64_t row = min(min(tile, ntiles - 1) * 16 + li, a.rows - 1);
        const float* xr = a.x + row * kNi + kq;
#pragma unroll
        for (int kk = 0; kk < kKXM; ++kk) dst[kk] = xr[4 * kk];
    };
    float A[kKXM], An[kKXM];
    load_tile(blockIdx.x, A);
#if TA_XPROJ_ORDER == 1
    // Column tile by column tile: the 12 k-steps of ONE tile (a dependent chain: 64 cycles per MFMA all the same), then the
    // four row stores of the tile BEFORE it -- issued in the shadow of the chain's last MFMA, so a wave's stores are spread
    // over its MFMA stream (4 per 832 cycles) instead of coming as 28 in a burst when all four waves of the CU have
    // finished a row tile together and the matrix pipes idle until the shared store path has drained.
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        load_tile(tile + gridDim.x, An);                       // flies under this tile's MFMAs
        double Ad[kKXM];
#pragma unroll
        for (int kk = 0; kk < kKXM; ++kk) Ad[kk] = (double)A[kk];
        double* orow[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // rows past the end are CLAMPED here as in load_tile: such a lane computes the last row's values again and writes
            // them to the last row again -- no branch around any store
            const int64_t row = min(tile * 16 + 4 * r + kq, a.rows - 1);
            orow[r] = gxd + row * kCols + 16 * tile0 + li;
        }
        f64x4 prev = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            f64x4 cur = {bias[s], bias[s], bias[s], bias[s]};
#pragma unroll
            for (int kk = 0; kk < kKXM; ++kk)
#if defined(TA_XPROJ_ABL) && (TA_XPROJ_ABL & 2)       // timing ablation: one MFMA per tile instead of 12
                if (kk == 0)
#endif
                cur = __builtin_amdgcn_mfma_f64_16x16x4f64(Ad[kk], Bx[s][kk], cur, 0, 0, 0);
#if defined(TA_XPROJ_ABL) && (TA_XPROJ_ABL & 1)       // timing ablation: no stores (one that never happens keeps the values alive)
            if (s > 0 && a.rows < 0) {
#else
            if (s > 0) {
#endif
#pragma unroll
                for (int r = 0; r < 4; ++r) orow[r][16 * (s - 1)] = prev[r];
            }
            __builtin_amdgcn_sched_barrier(0);
            prev = cur;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) orow[r][16 * (NT - 1)] = prev[r];
#pragma unroll
        for (int kk = 0; kk < kKXM; ++kk) A[kk] = An[kk];
    }
    return;
#endif
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        load_tile(tile + gridDim.x, An);                       // flies under this tile's MFMAs
        f64x4 acc[NT];
#pragma unroll
        for (int s = 0; s < NT; ++s) acc[s] = (f64x4){bias[s], bias[s], bias[s], bias[s]};
#pragma unroll
        for (int kk = 0; kk < kKXM; ++kk) {
            const double av = (double)A[kk];
#pragma unroll
            for (int s = 0; s < NT; ++s) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Bx[s][kk], acc[s], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = tile * 16 + 4 * r + kq;
            if (row < a.rows) {
                double* o = gxd + row * kCols + 16 * tile0 + li;
#pragma unroll
                for (int s = 0; s < NT; ++s) o[16 * s] = acc[s][r];
            }
        }
#pragma unroll
        for (int kk = 0; kk < kKXM; ++kk) A[kk] = An[kk];
    }
}

#ifdef TA_XPROJ_OCC2            // timing builds: two four-wave workgroups per CU (the compiler must fit 256 registers: it spills)
#define TA_XPROJ_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#else
#define TA_XPROJ_ATTR
#endif
__global__ __launch_bounds__(kXW * 64) TA_XPROJ_ATTR void lstm_xproj_f64_kernel(XprojArgs a) {
    const int dir = blockIdx.y, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == 0) xproj_body<ntiles_of(0)>(a, dir, wave, lane);
    else xproj_body<ntiles_of(1)>(a, dir, wave, lane);
}

// ---------------------------------------------------------------------------------------------
// Kr: the recurrence
struct Seq64Args {
    const double* gx;          // [dir 2][gx_rows][400], row r of it = absolute row gx_row0 + r
    int64_t gx_row0, gx_rows;
    const int64_t* row_off;    // per line: first (absolute) row
    const int32_t* T;          // per line: timesteps
    const int32_t* group_lines;// [ngroups][16] line ids, -1 = empty slot
    const double* wh;          // [dir 2][wave 4][slot 7][k-step 25][lane 64]: A fragments, W_gate(i / 4)[unit 4 tile + i % 4][49 + 4 kstep + lane / 16],
                               // i = lane % 16, tile = 6 wave + slot (slots 0..5) or 24 (slot 6: the tile split along k)
    const double* peep;        // [dir 2][3: WIP, WFP, WOP][100]
    float* hout;               // [rows][200] (absolute rows)
    const double* h0;          // optional [lines][2][100]: outputs before the first step
    const double* c0;          // optional [lines][2][100]: cell states before the first step
    const int32_t* tstart;     // optional [lines][2]: steps of the sequence already done
    int32_t* status;           // optional: one word, OR-ed with TA_LSTM_F64_PARTS_LATE if a wait for the split tile's parts runs out
};

// The f64 MFMA holds the SIMD's vector issue for its 64 cycles: any VALU instruction between two MFMAs of a chain
// is paid in full (tools/ubench/mfma_f64_ops.hip: 64 cycles per MFMA back to back, 93 with the two v_accvgpr_read_b32
// the compiler uses to feed an operand it keeps in AGPRs; an LDS read between them is free, and so is an A operand the
// MFMA reads from AGPRs itself).  A wave's weights (300 .. 318 registers) do not fit the 256 architectural VGPRs next
// to everything else, so the first kAgprDoubles of them are PINNED in AGPRs and named as such in the instruction; the
// accumulators stay in VGPRs (Gx arrives there, the cell update reads them there).  Written as inline assembly, so the
// wait states the compiler would insert after a matrix instruction are ours to provide (mfma_settle).
constexpr int kAgprDoubles = 120;
template <bool IN_AGPR>
__device__ __forceinline__ void mfma_f64(f64x4& acc, double aw, double b) {
    if (IN_AGPR) asm("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "a"(aw), "v"(b));
    else asm("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(aw), "v"(b));
}
// before the first MFMA of a chain: a VALU instruction may just have written one of its operands (the compiler
// leaves two wait states between such a write and the builtin)
__device__ __forceinline__ void mfma_begin(f64x4& acc) { asm volatile("s_nop 3" : "+v"(acc)); }
// after the last MFMA of a chain, before anything else reads its result: 16 passes = 18 wait states (what the
// compiler puts after the builtin: s_nop 15, s_nop 2)
__device__ __forceinline__ void mfma_settle(f64x4& acc) { asm volatile("s_nop 15\n s_nop 2" : "+v"(acc)); }

// Tiles of a step: every wave owns six (tiles 6 wave .. 6 wave + 5); the 25th (units 96 .. 99) is split along k --
// waves 1, 2, 3 take k-steps 0..8, 9..16, 17..24 of it at the start of their step and leave the partial sums in LDS,
// wave 0 (which takes none) adds them to Gx in that fixed order when its own six tiles are done and updates those
// four units' cells.  A step therefore costs a SIMD six tiles + ~0.35 of one instead of seven (25 tiles on four SIMDs).
// K0, NK: this wave's k-steps of the split tile (NK = 0: the wave that sums the parts).
template <int K0, int NK>
__device__ __forceinline__ void seq_f64_body(const Seq64Args& a, double (&hs)[2][kNs][kLines], const double (&peep_s)[3][kNs],
                                             double (&part)[kW - 1][4][64], unsigned& part_flag,
                                             const int (&s_line)[kLines], const int (&s_T)[kLines],
                                             const long long (&s_row)[kLines], int dir, int wave, int lane, int Tmax) {
    constexpr int NT = kOwnTiles;
    constexpr bool kSums = NK == 0;
    const int tile0 = NT * wave;
    // recurrent weights of this wave's tiles: A fragments, constant over time
    double Aw[NT][kKH];
    double Ap[NK > 0 ? NK : 1];
    {
        const double* wp = a.wh + ((size_t)(dir * kW + wave) * kMaxNT * kKH) * 64 + lane;
#pragma unroll
        for (int s = 0; s < NT; ++s)
#pragma unroll
            for (int kk = 0; kk < kKH; ++kk) Aw[s][kk] = wp[((size_t)s * kKH + kk) * 64];
#pragma unroll
        for (int i = 0; i < NK; ++i) Ap[i] = wp[((size_t)NT * kKH + K0 + i) * 64];
    }
    // this lane's accumulators: the four gates of (line slot li, unit 4 (tile0 + s) + kq), s < NT; wave 0 also those
    // of unit 96 + kq
    const int li = lane & 15, kq = lane >> 4;
    const int myT = s_T[li];
    const long long myrow = s_row[li];
    const int myid = s_line[li];
    const int ubase = 4 * tile0 + kq;
    constexpr int kSplitTile = kTiles - 1;
    const int usplit = 4 * kSplitTile + kq;
    const double* gxd = a.gx + (size_t)dir * a.gx_rows * kCols + 2 * kq;
    auto gx_row = [&](int t) -> const double* {
        int tt = t < myT ? t : myT - 1;
        if (dir) tt = myT - 1 - tt;                            // Reversed(LSTM): run on xs[::-1]
        const long long row = myT > 0 ? myrow - a.gx_row0 + tt : 0;     // an empty slot reads row 0 of the buffer (never used)
        return gxd + row * kCols;
    };
    double c[NT], csplit = 0.0;
    int ts = 0;
#pragma unroll
    for (int s = 0; s < NT; ++s) c[s] = 0.0;
    if (myid >= 0) {
        if (a.tstart) ts = a.tstart[(size_t)myid * 2 + dir];
        if (a.c0 && ts > 0) {                                  // (a sequence that starts here starts from c = 0)
#pragma unroll
            for (int s = 0; s < NT; ++s) c[s] = a.c0[((size_t)myid * 2 + dir) * kNs + ubase + 4 * s];
            if (kSums) csplit = a.c0[((size_t)myid * 2 + dir) * kNs + usplit];
        }
    }
    // the output peephole is skipped at the first step of the whole sequence: 0 / 1 factor, 1 from the second step on
    double wop_on = ts > 0 ? 1.0 : 0.0;
    float* houtp = a.hout + dir * kNs;

    auto load_gx = [&](const double* g, int tile) -> f64x4 {
        const f64x2 lo = *reinterpret_cast<const f64x2*>(g + 16 * tile);
        const f64x2 hi = *reinterpret_cast<const f64x2*>(g + 16 * tile + 8);
        return (f64x4){lo[0], lo[1], hi[0], hi[1]};
    };
    f64x4 acc[NT], accs = {0.0, 0.0, 0.0, 0.0};
    {
        const double* g = gx_row(0);
#pragma unroll
        for (int s = 0; s < NT; ++s) acc[s] = load_gx(g, tile0 + s);
    }
#ifdef TA_F64_PROFILE
    unsigned long long p_comp = 0, p_move = 0, p_bar = 0;
#endif
    for (int t = 0; t < Tmax; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        const double* gnext = gx_row(t + 1);
        const int tt = dir ? myT - 1 - t : t;
        float* hrow = houtp + (myrow + tt) * (2 * kNs);       // (dereferenced only while t < myT)
        // the split tile's Gx of THIS step: asked for first, so that when it is needed (after the six tiles) the
        // wait counts only loads issued after it -- a load carried across the loop edge is waited for with
        // vmcnt(0), which here would also wait for the tile loads issued moments before
        if (kSums) accs = load_gx(gx_row(t), kSplitTile);
        // The B operand of the step (h_{t-1}: B[k][j] = unit k of line j, lane j + 16 (k % 4)): once from LDS into
        // registers, for all tiles.
        double B[kKH];
#pragma unroll
        for (int kk = 0; kk < kKH; ++kk) B[kk] = hs[cur][4 * kk + kq][li];
        if (NK > 0) {
            f64x4 pacc = {0.0, 0.0, 0.0, 0.0};
            mfma_begin(pacc);
#pragma unroll
            for (int i = 0; i < NK; ++i) mfma_f64<false>(pacc, Ap[i], B[K0 + i]);
            mfma_settle(pacc);
#pragma unroll
            for (int r = 0; r < 4; ++r) part[wave - 1][r][lane] = pacc[r];
            // a wave's LDS instructions execute in order: the count follows the four stores of all its lanes.  The fences
            // name the LDS only -- a plain workgroup release / acquire also drains the Gx loads in flight (vmcnt(0))
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            if (lane == 0) __hip_atomic_fetch_add(&part_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // Tile by tile: the 25 k-steps of a tile (one dependent accumulator chain: 64 cycles per MFMA either way),
        // its cell update, then the loads of ITS accumulators for the next step (they have a whole step to arrive).
#pragma unroll
        for (int s = 0; s < NT; ++s) {
#ifdef TA_F64_PROFILE
            const unsigned long long p0 = prof_now();
#endif
            mfma_begin(acc[s]);
#pragma unroll
            for (int kk = 0; kk < kKH; ++kk) {
                if (s * kKH + kk < kAgprDoubles) mfma_f64<true>(acc[s], Aw[s][kk], B[kk]);
                else mfma_f64<false>(acc[s], Aw[s][kk], B[kk]);
            }
            mfma_settle(acc[s]);
            const int unit = ubase + 4 * s;
            const double h = lstm_cell_f64(acc[s][0], acc[s][1], acc[s][2], acc[s][3], c[s],
                                           peep_s[0][unit], peep_s[1][unit], wop_on * peep_s[2][unit]);
#ifdef TA_F64_PROFILE
            unsigned long long p1;                                 // (h as an input: the stamp follows the cell update)
            asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(p1) : "v"(h) : "memory");
#endif
            hs[nxt][unit][li] = h;
            if (t < myT) hrow[unit] = (float)h;
            acc[s] = load_gx(gnext, tile0 + s);            // (unconditional: gx_row clamps past the end; a conditional load would
                                                           // make every later wait count it as possibly absent)
#ifdef TA_F64_PROFILE
            p_comp += p1 - p0;
            p_move += prof_now() - p1;
#endif
        }
        if (kSums) {
#ifdef TA_F64_PROFILE
            const unsigned long long p0 = prof_now();
#endif
            // the parts were posted at the start of the other waves' step; the wait is bounded all the same
            const unsigned want = (unsigned)(kW - 1) * (unsigned)(t + 1);
            bool arrived = false;
            for (int spin = 0; spin < (1 << 22) && !arrived; ++spin)
                arrived = __hip_atomic_load(&part_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= want;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            if (!arrived) {                                    // a wait that ran out must show: NaN outputs, not plausible ones,
                accs[3] = __builtin_nan("");                   // (ci_pre: the one pre-activation no min / max stands behind), and a
                                                               // word the host reads with the decoder's counts
                if (a.status && lane == 0) atomicOr(a.status, TA_LSTM_F64_PARTS_LATE);
            }
#pragma unroll
            for (int w = 0; w < kW - 1; ++w)
#pragma unroll
                for (int r = 0; r < 4; ++r) accs[r] += part[w][r][lane];
            const double h = lstm_cell_f64(accs[0], accs[1], accs[2], accs[3], csplit,
                                           peep_s[0][usplit], peep_s[1][usplit], wop_on * peep_s[2][usplit]);
            hs[nxt][usplit][li] = h;
            if (t < myT) hrow[usplit] = (float)h;
#ifdef TA_F64_PROFILE
            p_move += prof_now() - p0;
#endif
        }
        wop_on = 1.0;
#ifdef TA_F64_PROFILE
        const unsigned long long pb = prof_now();
#endif
        __syncthreads();
#ifdef TA_F64_PROFILE
        p_bar += prof_now() - pb;
#endif
    }
#ifdef TA_F64_PROFILE
    if (wave == 0 && lane == 0) {
        atomicAdd(&g_prof[0], p_comp);
        atomicAdd(&g_prof[1], p_move);
        atomicAdd(&g_prof[2], p_bar);
        atomicAdd(&g_prof[3], (unsigned long long)Tmax);
    }
#endif
}

__global__ __launch_bounds__(kW * 64) void lstm_seq_f64_kernel(Seq64Args a) {
    __shared__ __attribute__((aligned(16))) double hs[2][kNs][kLines];       // h of the 16 lines, [unit][line]
    __shared__ double peep_s[3][kNs];
    __shared__ double part[kW - 1][4][64];                                   // the split tile's partial sums
    __shared__ unsigned part_flag;                                           // parts posted so far (3 per step)
    __shared__ int s_line[kLines];
    __shared__ int s_T[kLines];
    __shared__ long long s_row[kLines];

    const int grp = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) part_flag = 0;
    if (tid < kLines) {
        const int id = a.group_lines[grp * kLines + tid];
        s_line[tid] = id;
        s_T[tid] = id >= 0 ? a.T[id] : 0;
        s_row[tid] = id >= 0 ? a.row_off[id] : 0;
    }
    for (int e = tid; e < 2 * kNs * kLines; e += kW * 64) (&hs[0][0][0])[e] = 0.0;
    for (int e = tid; e < 3 * kNs; e += kW * 64) (&peep_s[0][0])[e] = a.peep[(size_t)dir * 3 * kNs + e];
    __syncthreads();
    int Tmax = 0;
#pragma unroll
    for (int s = 0; s < kLines; ++s) Tmax = max(Tmax, s_T[s]);
    if (a.h0) {                                            // h_{-1} of continued sequences
        for (int e = tid; e < kLines * kNs; e += kW * 64) {
            const int slot = e / kNs, u = e % kNs;
            const int id = s_line[slot];
            if (id >= 0) hs[0][u][slot] = a.h0[((size_t)id * 2 + dir) * kNs + u];
        }
    }
    __syncthreads();
    if (wave == 0) seq_f64_body<0, 0>(a, hs, peep_s, part, part_flag, s_line, s_T, s_row, dir, wave, lane, Tmax);
    else if (wave == 1) seq_f64_body<0, 9>(a, hs, peep_s, part, part_flag, s_line, s_T, s_row, dir, wave, lane, Tmax);
    else if (wave == 2) seq_f64_body<9, 8>(a, hs, peep_s, part, part_flag, s_line, s_T, s_row, dir, wave, lane, Tmax);
    else seq_f64_body<17, 8>(a, hs, peep_s, part, part_flag, s_line, s_T, s_row, dir, wave, lane, Tmax);
}


// ---------------------------------------------------------------------------------------------
// Kr4: the recurrence on groups of FOUR lines (round 5) -- the same arithmetic, bit for bit, on v_mfma_f64_4x4x4_4b_f64.
//
// The recurrence is a chain of T dependent steps per workgroup and a step of Kr costs what its 16 x 16 x 4 tiles cost
// however few of the 16 columns are lines: a page of 30 lines is two groups per direction (four workgroups, 14 ms), and
// even 1 920 lines are ONE round of 240 workgroups on 256 CUs whose time is the longest line's chain.  The 4 x 4 x 4 form
// computes four independent 4 x 4 x 4 blocks per instruction in 16 cycles (tools/ubench/mfma_f64_4x4.hip,
// profiles/r05_mfma_f64_4x4.txt: the same 16 multiply-adds per cycle and SIMD as the 16 x 16 x 4 form; a dependent chain
// needs four wait states, hidden by three chains per wave and two waves per SIMD).  Operand layout (measured there):
//     A[i][k] of block b in lane i + 4 b + 16 k,   B[k][j] of block b in lane j + 4 b + 16 k,   D[i][j] of block b in lane j + 4 b + 16 i.
// A tile is the same 16 weight rows as in Kr -- (4 gates) x (4 units) -- laid out as block b = unit-in-tile, row i = gate;
// B is h_{t-1} of the four lines, the same for all four blocks (an LDS broadcast read); D puts gate i of (unit b, line j)
// into lane j + 4 b + 16 i: the four gates of a cell sit in the four 16-lane ROWS of the wave, at the same place in each.
// A wave owns three tiles (three accumulator chains); a 4 x 4 transpose of (accumulator, row) by the gfx950 row swaps
// (v_permlane16_swap, v_permlane32_swap: 8 instructions for doubles) gives the lanes of row r the four gates of tile r,
// whose cell they update -- the function Kr calls, from accumulators that took the same fma chain (Gx, then k ascending:
// one block's 4-term product is the fma chain k = 0..3 of the 16 x 16 x 4 form, checked in the microbenchmark), so the two
// kernels' outputs are equal to the bit.  Eight waves (two per SIMD, 256 registers each: 150 hold the wave's weights) cover
// 24 tiles; the 25th is wave 7's fourth, computed as the three partial chains Kr splits it into (k-steps 0..8, 9..16,
// 17..24 from zero, added to Gx in that order) and updated by the 16 lanes of its fourth row.
constexpr int kG4 = 4;              // lines per workgroup
constexpr int kW4 = 8;              // waves per workgroup

struct Seq64G4Args {
    const double* gx;          // as Seq64Args
    int64_t gx_row0, gx_rows;
    const int64_t* row_off;
    const int32_t* T;
    const int32_t* group_lines;// [ngroups][4] line ids, -1 = empty slot
    const double* wh4;         // [dir 2][tile 25][k-step 25][lane 64]: A fragments of the 4 x 4 x 4 form,
                               // W_gate(lane % 4)[unit 4 tile + (lane / 4) % 4][49 + 4 kstep + lane / 16]
    const double* peep;
    float* hout;
    const double* h0;
    const double* c0;
    const int32_t* tstart;
    int32_t* status;           // as Seq64Args
};

__device__ __forceinline__ void swap_rows16(double& a, double& b) {       // a.row1 <-> b.row0, a.row3 <-> b.row2
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const unsigned long long ua = __builtin_bit_cast(unsigned long long, a), ub = __builtin_bit_cast(unsigned long long, b);
    const u32x2 lo = __builtin_amdgcn_permlane16_swap((unsigned)ua, (unsigned)ub, false, false);
    const u32x2 hi = __builtin_amdgcn_permlane16_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
    a = __builtin_bit_cast(double, (unsigned long long)lo[0] | ((unsigned long long)hi[0] << 32));
    b = __builtin_bit_cast(double, (unsigned long long)lo[1] | ((unsigned long long)hi[1] << 32));
}
__device__ __forceinline__ void swap_rows32(double& a, double& b) {       // a.rows 2, 3 <-> b.rows 0, 1
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const unsigned long long ua = __builtin_bit_cast(unsigned long long, a), ub = __builtin_bit_cast(unsigned long long, b);
    const u32x2 lo = __builtin_amdgcn_permlane32_swap((unsigned)ua, (unsigned)ub, false, false);
    const u32x2 hi = __builtin_amdgcn_permlane32_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
    a = __builtin_bit_cast(double, (unsigned long long)lo[0] | ((unsigned long long)hi[0] << 32));
    b = __builtin_bit_cast(double, (unsigned long long)lo[1] | ((unsigned long long)hi[1] << 32));
}

// Every wave owns tiles 3 wave .. 3 wave + 2.  The 25th tile (units 96..99) is computed as the three partial chains Kr splits
// it into -- k-steps 0..8, 9..16, 17..24 from zero -- by waves 4, 5, 6 (one per SIMD 0, 1, 2; K0, NK: this wave's k-steps),
// at the START of their step; the parts go through LDS to wave 3 (SUMS; it shares SIMD 3 with wave 7 and has no part of its
// own), whose fourth row of lanes adds them to Gx in Kr's order and updates those four units' cells.  A SIMD's step is
// then 150 + 9 / 8 / 8 / 0 MFMAs (the first form gave wave 7 all 25: 175 on SIMD 3 and a step 10 % longer).
template <int K0, int NK, bool SUMS>
__device__ __forceinline__ void seq4_f64_body(const Seq64G4Args& a, double (&hs)[2][kNs][kG4], double (&part)[3][64],
                                              const double (&ap_s)[kKH][64], unsigned& part_flag, const int (&s_line)[kG4], const int (&s_T)[kG4],
                                              const long long (&s_row)[kG4], int dir, int wave, int lane, int Tmax) {
    constexpr int kSplitTile = kTiles - 1;
    const int tile0 = 3 * wave;
    double Aw[3][kKH];
    {
        const double* wp = a.wh4 + ((size_t)(dir * kTiles + tile0) * kKH) * 64 + lane;
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int kk = 0; kk < kKH; ++kk) Aw[s][kk] = wp[((size_t)s * kKH + kk) * 64];
    }
    // roles of this lane.  Before the transpose (accumulators, Gx): gate `row`, unit-in-tile ub, line j of each of the wave's
    // tiles; as B operand: k = row, line j; after the transpose: the cell (tile slot `row`, unit ub, line j).
    const int j = lane & 3, ub = (lane >> 2) & 3, row = lane >> 4;
    const int myT = s_T[j];
    const long long myrow = s_row[j];
    const int myid = s_line[j];
    const bool has_cell = SUMS || row < 3;
    const int mytile = row < 3 ? tile0 + row : kSplitTile;
    const int unit = 4 * mytile + ub;
    const double* pp = a.peep + (size_t)dir * 3 * kNs;
    const double wip = pp[unit], wfp = pp[kNs + unit], wop = pp[2 * kNs + unit];
    // Gx of (gate `row`, unit ub of a tile): position 16 tile + 8 (gate / 2) + 2 ub + gate % 2 of the row (gx_index);
    // the summing lanes take all four gates of (unit ub of tile 24): two 16-byte pieces at 16 * 24 + 2 ub (+ 8)
    const double* gxrow0 = a.gx + (size_t)dir * a.gx_rows * kCols;
    const int gx_own = 8 * (row >> 1) + 2 * ub + (row & 1);
    // running pointers: the Gx row of the current step (stepping +-400 doubles while the line lasts, then staying on its
    // last row -- what it computes past its end is never stored) and the hout element of this lane's cell (+-200 floats)
    const long long gstep = dir ? -(long long)kCols : (long long)kCols;
    const double* gcur = gxrow0 + (myT > 0 ? myrow - a.gx_row0 + (dir ? myT - 1 : 0) : 0) * kCols;
    double c = 0.0;
    int ts = 0;
    if (myid >= 0) {
        if (a.tstart) ts = a.tstart[(size_t)myid * 2 + dir];
        if (a.c0 && has_cell && ts > 0) c = a.c0[((size_t)myid * 2 + dir) * kNs + unit];
    }
    double wop_t = ts > 0 ? wop : 0.0;                       // the output peephole is skipped at the sequence's first step
    float* hptr = a.hout + dir * kNs + unit + (myrow + (dir ? myT - 1 : 0)) * (2 * kNs);
    const long long hstep = dir ? -2 * kNs : 2 * kNs;
    double acc[3];
    f64x2 gs01 = {0.0, 0.0}, gs23 = {0.0, 0.0};
    {
        const double* g = gcur;
#pragma unroll
        for (int s = 0; s < 3; ++s) acc[s] = g[16 * (tile0 + s) + gx_own];
        if (SUMS) {
            gs01 = *reinterpret_cast<const f64x2*>(g + 16 * kSplitTile + 2 * ub);
            gs23 = *reinterpret_cast<const f64x2*>(g + 16 * kSplitTile + 2 * ub + 8);
        }
    }
#ifdef TA_F64_PROFILE
    unsigned long long q_mfma = 0, q_cell = 0, q_bar = 0;
#endif
    for (int t = 0; t < Tmax; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        const double* gnext = t + 1 < myT ? gcur + gstep : gcur;
        gcur = gnext;
        const double* hb = &hs[cur][row][j];
#ifdef TA_F64_PROFILE
        const unsigned long long q0 = prof_now();
#endif
        // The B operand of the step (h_{t-1}[unit 4 kk + row][line j]) comes from LDS in chunks of five k-steps, two chunks
        // ahead of the MFMAs that use them (15 MFMAs = 244 cycles cover an LDS read): with the reads between the MFMAs, one
        // group of three ahead, a SIMD's 159 MFMAs took 3 170 cycles instead of 2 590 -- the younger wave of a SIMD runs the
        // second half of its chains alone, every wait exposed; all 25 values at once do not fit beside the 150 weight registers
        // The next step's accumulators (Gx) are asked for HERE, a whole step ahead: Gx streams from HBM (6 400 bytes per
        // timestep, no reuse) and a miss takes longer than the cell update that used to cover it -- the younger wave of a
        // SIMD, whose cell update is the last thing in a step, then waited ~600 cycles at the start of the next
        double gn[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) gn[s] = gnext[16 * (tile0 + s) + gx_own];
        constexpr int CH = 5, NCH = kKH / CH;
        double B[2][CH];
#ifdef TA_F64_SETPRIO                            // measured: no gain (4 395 against 4 265 cycles per step) -- timing builds only
        __builtin_amdgcn_s_setprio(3);
#endif
#pragma unroll
        for (int q = 0; q < 2 * CH; ++q) B[q / CH][q % CH] = hb[4 * q * kG4];
        if (NK > 0) {
            // this wave's part of the split tile first (its B values straight from LDS: nine or eight reads, once)
            double pacc = 0.0;
#pragma unroll
            for (int i = 0; i < NK; ++i)                             // (the split tile's weights come from LDS too: 150 registers hold the wave's own)
                pacc = __builtin_amdgcn_mfma_f64_4x4x4f64(ap_s[K0 + i][lane], hb[4 * (K0 + i) * kG4], pacc, 0, 0, 0);
            part[wave - 4][lane] = pacc;
            // a wave's LDS instructions execute in order: the count follows the store of all its lanes.  LDS-only fences --
            // a plain workgroup release / acquire would also drain the Gx loads in flight
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            if (lane == 0) __hip_atomic_fetch_add(&part_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
            for (int q = 0; q < CH; ++q)
#pragma unroll
                for (int s = 0; s < 3; ++s)
#if defined(TA_F64_ABL) && (TA_F64_ABL & 2)       // timing ablation: one MFMA per chunk and chain instead of five
                    if (q == 0)
#endif
                    acc[s] = __builtin_amdgcn_mfma_f64_4x4x4f64(Aw[s][ch * CH + q], B[ch & 1][q], acc[s], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ch + 2 < NCH) {
#pragma unroll
                for (int q = 0; q < CH; ++q) B[ch & 1][q] = hb[4 * ((ch + 2) * CH + q) * kG4];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef TA_F64_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        double g0 = acc[0], g1 = acc[1], g2 = acc[2], g3 = 0.0;
#ifdef TA_F64_PROFILE
        unsigned long long q1;                                       // (g0 as an input: the stamp follows the MFMAs)
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(q1) : "v"(g0), "v"(g2) : "memory");
#endif
        // The next step's accumulators take their Gx values HERE, before this step's hout store is issued.  Left to itself
        // the compiler sinks these copies to the top of the next iteration, where the loads and the store are both
        // pending; with mixed kinds outstanding it cannot count and waits with vmcnt(0) -- i.e. for the store's
        // acknowledgement by the L2, every step, exposed on the wave whose cell update is the last thing in the step
        // (~800 of 4 200 cycles).  Here only the loads (issued a whole step ago) are outstanding: the wait is free.
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            acc[s] = gn[s];
            asm volatile("" : "+v"(acc[s]));
        }
        // (accumulator a, row b) -> (accumulator b, row a): row r then holds gates 0..3 of tile slot r in g0..g3
        swap_rows16(g0, g1);
        swap_rows16(g2, g3);
        swap_rows32(g0, g2);
        swap_rows32(g1, g3);
        if (SUMS) {
            // the parts were posted at the start of the other waves' step; the wait is bounded all the same
            const unsigned want = 3u * (unsigned)(t + 1);
            bool arrived = false;
            for (int spin = 0; spin < (1 << 22) && !arrived; ++spin)
                arrived = __hip_atomic_load(&part_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= want;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            const double* pl = &part[0][lane & 15];
            double v[4] = {gs01[0], gs01[1], gs23[0], gs23[1]};
            if (!arrived) {                                    // a wait that ran out must show: NaN outputs, not plausible ones,
                v[3] = __builtin_nan("");                      // (ci_pre: the one pre-activation no min / max stands behind), and a
                                                               // word the host reads with the decoder's counts
                if (a.status && lane == 0) atomicOr(a.status, TA_LSTM_F64_PARTS_LATE);
            }
#pragma unroll
            for (int w = 0; w < 3; ++w)                              // Kr's order: Gx, then the parts of k-steps 0..8, 9..16, 17..24
#pragma unroll
                for (int g = 0; g < 4; ++g) v[g] += pl[w * 64 + 16 * g];
            if (row == 3) { g0 = v[0]; g1 = v[1]; g2 = v[2]; g3 = v[3]; }
            gs01 = *reinterpret_cast<const f64x2*>(gnext + 16 * kSplitTile + 2 * ub);
            gs23 = *reinterpret_cast<const f64x2*>(gnext + 16 * kSplitTile + 2 * ub + 8);
        }
#if defined(TA_F64_ABL) && (TA_F64_ABL & 1)       // timing ablation: no cell update
        const double h = ((g0 + g1) + (g2 + g3)) * 1e-3 + c * wip;
#else
        const double h = lstm_cell_f64(g0, g1, g2, g3, c, wip, wfp, wop_t);
#endif
        wop_t = wop;
        if (has_cell) {
#if defined(TA_F64_ABL) && (TA_F64_ABL & 4)       // timing ablation: the cell update runs, but the MFMAs see a constant operand
            hs[nxt][unit][j] = h * 0.0 + 0.001 * (lane & 3);
#else
            hs[nxt][unit][j] = h;
#endif
            if (t < myT) *hptr = (float)h;
        }
        hptr += hstep;
#ifdef TA_F64_PROFILE
        unsigned long long q2;
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(q2) : "v"(h) : "memory");
#endif
        // The step's barrier orders the LDS only (h of this step for every wave's next B operand).  __syncthreads() is a
        // workgroup fence over ALL memory: it made every wave wait for its hout store to be acknowledged by the L2 -- a
        // write latency on the critical path of every step (the last wave's cell update, its store, the wait, the barrier).
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#ifdef TA_F64_PROFILE
        q_mfma += q1 - q0; q_cell += q2 - q1; q_bar += prof_now() - q2;
#endif
    }
#ifdef TA_F64_PROFILE
    if (wave == TA_F64_PROF_WAVE && lane == 0) {
        atomicAdd(&g_prof[0], q_mfma);
        atomicAdd(&g_prof[1], q_cell);
        atomicAdd(&g_prof[2], q_bar);
        atomicAdd(&g_prof[3], (unsigned long long)Tmax);
    }
#endif
}

__global__ __launch_bounds__(kW4 * 64) void lstm_seq4_f64_kernel(Seq64G4Args a) {
    __shared__ __attribute__((aligned(16))) double hs[2][kNs][kG4];          // h of the four lines, [unit][line]
    __shared__ double part[3][64];                                           // the 25th tile's partial sums
    __shared__ double ap_s[kKH][64];                                         // the 25th tile's A fragments (12.8 KB)
    __shared__ unsigned part_flag;                                           // parts posted so far (3 per step)
    __shared__ int s_line[kG4];
    __shared__ int s_T[kG4];
    __shared__ long long s_row[kG4];

    const int grp = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) part_flag = 0;
    if (tid < kG4) {
        const int id = a.group_lines[grp * kG4 + tid];
        s_line[tid] = id;
        s_T[tid] = id >= 0 ? a.T[id] : 0;
        s_row[tid] = id >= 0 ? a.row_off[id] : 0;
    }
    for (int e = tid; e < 2 * kNs * kG4; e += kW4 * 64) (&hs[0][0][0])[e] = 0.0;
    for (int e = tid; e < kKH * 64; e += kW4 * 64) (&ap_s[0][0])[e] = a.wh4[((size_t)(dir * kTiles + kTiles - 1) * kKH) * 64 + e];
    __syncthreads();
    int Tmax = 0;
#pragma unroll
    for (int s = 0; s < kG4; ++s) Tmax = max(Tmax, s_T[s]);
    if (a.h0) {
        for (int e = tid; e < kG4 * kNs; e += kW4 * 64) {
            const int slot = e / kNs, u = e % kNs;
            const int id = s_line[slot];
            if (id >= 0) hs[0][u][slot] = a.h0[((size_t)id * 2 + dir) * kNs + u];
        }
    }
    __syncthreads();
    if (wave == 3) seq4_f64_body<0, 0, true>(a, hs, part, ap_s, part_flag, s_line, s_T, s_row, dir, wave, lane, Tmax);
    else if (wave == 4) seq4_f64_body<0, 9, false>(a, hs, part, ap_s, part_flag, s_line, s_T, s_row, dir, wave, lane, Tmax);
    else if (wave == 5) seq4_f64_body<9, 8, false>(a, hs, part, ap_s, part_flag, s_line, s_T, s_row, dir, wave, lane, Tmax);
    else if (wave == 6) seq4_f64_body<17, 8, false>(a, hs, part, ap_s, part_flag, s_line, s_T, s_row, dir, wave, lane, Tmax);
    else seq4_f64_body<0, 0, false>(a, hs, part, ap_s, part_flag, s_line, s_T, s_row, dir, wave, lane, Tmax);
}

}  // namespace ta64

using namespace ta64;

extern "C" int64_t ta_lstm_f64_weight_doubles(int32_t which) {
    // 0: wh [2][4][7][25][64]; 1: wx [2][25][13][64]; 2: peep [2][3][100]; 3: wh4 [2][25][25][64]
    if (which == 0) return (int64_t)2 * kW * kMaxNT * kKH * 64;
    if (which == 1) return (int64_t)2 * kTiles * kKX * 64;
    if (which == 2) return (int64_t)2 * 3 * kNs;
    if (which == 3) return (int64_t)2 * kTiles * kKH * 64;          // wh4 [2][25][25][64]: the four-line kernel's
    return 0;
}

#ifdef TA_F64_PROFILE
extern "C" int ta_lstm_f64_profile(unsigned long long* out4, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_prof), sizeof(g_prof));
    if (e == hipSuccess && reset) {
        const unsigned long long z[4] = {0, 0, 0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z));
    }
    return e == hipSuccess ? 0 : ta_fail_hip(e, "ta_lstm_f64_profile");
}
#endif

extern "C" int64_t ta_lstm_f64_gx_bytes(int64_t rows) { return rows < 0 ? 0 : (int64_t)2 * rows * kCols * 8; }

extern "C" int ta_lstm_xproj_f64(const float* x, int64_t rows, const double* wx, double* gx, void* stream) {
    if (rows < 0) return ta_fail(TA_EINVAL, "negative row count");
    if (rows == 0) return TA_OK;
    if (!x || !wx || !gx) return ta_fail(TA_EINVAL, "null pointer argument");
    XprojArgs a{x, rows, wx, gx};
    const int64_t ntiles = (rows + 15) / 16;
    const dim3 grid((unsigned)(ntiles < 1024 ? ntiles : 1024), 2);
    hipLaunchKernelGGL(lstm_xproj_f64_kernel, grid, dim3(kXW * 64), 0, reinterpret_cast<hipStream_t>(stream), a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ta_fail_hip(e, "lstm_xproj_f64_kernel launch");
    return TA_OK;
}

extern "C" int ta_lstm_forward_f64(const double* gx, int64_t gx_row0, int64_t gx_rows, const int64_t* row_off,
                                   const int32_t* T, const int32_t* group_lines, int32_t ngroups, const double* wh,
                                   const double* peep, float* hout, const double* h0, const double* c0,
                                   const int32_t* tstart, int32_t* status, void* stream) {
    if (ngroups < 0 || gx_rows < 0 || gx_row0 < 0) return ta_fail(TA_EINVAL, "negative count");
    if (ngroups == 0) return TA_OK;
    if (!gx || !row_off || !T || !group_lines || !wh || !peep || !hout)
        return ta_fail(TA_EINVAL, "null pointer argument");
    if ((h0 != nullptr) != (c0 != nullptr) || (h0 != nullptr) != (tstart != nullptr))
        return ta_fail(TA_EINVAL, "h0, c0 and tstart go together (all null, or all given)");
    Seq64Args a{gx, gx_row0, gx_rows, row_off, T, group_lines, wh, peep, hout, h0, c0, tstart, status};
    hipLaunchKernelGGL(lstm_seq_f64_kernel, dim3(2 * ngroups), dim3(kW * 64), 0,
                       reinterpret_cast<hipStream_t>(stream), a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ta_fail_hip(e, "lstm_seq_f64_kernel launch");
    return TA_OK;
}

extern "C" int ta_lstm_forward_f64_g4(const double* gx, int64_t gx_row0, int64_t gx_rows, const int64_t* row_off,
                                      const int32_t* T, const int32_t* group_lines, int32_t ngroups, const double* wh4,
                                      const double* peep, float* hout, const double* h0, const double* c0,
                                      const int32_t* tstart, int32_t* status, void* stream) {
    if (ngroups < 0 || gx_rows < 0 || gx_row0 < 0) return ta_fail(TA_EINVAL, "negative count");
    if (ngroups == 0) return TA_OK;
    if (!gx || !row_off || !T || !group_lines || !wh4 || !peep || !hout)
        return ta_fail(TA_EINVAL, "null pointer argument");
    if ((h0 != nullptr) != (c0 != nullptr) || (h0 != nullptr) != (tstart != nullptr))
        return ta_fail(TA_EINVAL, "h0, c0 and tstart go together (all null, or all given)");
    Seq64G4Args a{gx, gx_row0, gx_rows, row_off, T, group_lines, wh4, peep, hout, h0, c0, tstart, status};
    hipLaunchKernelGGL(lstm_seq4_f64_kernel, dim3(2 * ngroups), dim3(kW4 * 64), 0,
                       reinterpret_cast<hipStream_t>(stream), a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ta_fail_hip(e, "lstm_seq4_f64_kernel launch");
    return TA_OK;
}
