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
//                             v_mfma_f64_16x16x4_f64, rows x 52 x 400 per direction.
//  Kr lstm_seq_f64_kernel     one workgroup of FOUR waves (one per SIMD, 512 registers each) per (16 lines,
//                             direction).  The recurrent weights W[:, 49:] (4 x 100 x 100 float64 = 320 KB) live in
//                             the CU's registers as A fragments for the whole kernel (the first 120 doubles of a wave
//                             pinned in AGPRs, which the MFMA reads itself); h_{t-1} of the 16 lines is the B operand,
//                             kept in LDS (float64) and read into registers once per step; per step the accumulators
//                             start from Gx (the loads are issued a step ahead) and take 25 k-steps of
//                             v_mfma_f64_16x16x4_f64; cell state and gate functions in float64 (own exp: range
//                             reduction + degree-10 polynomial, 3e-13).
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
constexpr int kXK = 52;             // [1, x(48), 3 zeros]
constexpr int kKX = kXK / 4;        // 13 k-steps over [1, x]
constexpr int kW = 4;               // waves per workgroup
constexpr int kMaxNT = 7;           // tile slots per wave in the packed recurrent weights: six own tiles + the 25th
constexpr int kOwnTiles = 6;        // recurrence: tiles a wave owns (6 wave .. 6 wave + 5); the 25th is split along k

typedef double f64x4 __attribute__((ext_vector_type(4)));

#ifdef TA_F64_PROFILE
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
// the projection's column tiles per wave: 0..6, 7..12, 13..18, 19..24 (giving the waves the 25th in turn, row tile by
// row tile, measured SLOWER: 6.1 against 5.4 ms per 2.76 M rows -- the kernel waits on its stores, not on the MFMAs)
__host__ __device__ constexpr int tile0_of(int wave) { return wave == 0 ? 0 : 1 + 6 * wave; }
__host__ __device__ constexpr int ntiles_of(int wave) { return wave == 0 ? 7 : 6; }

// ---------------------------------------------------------------------------------------------
// float64 gate functions.  exp: x = n ln2 + r, |r| <= ln2 / 2; e^r by its Taylor polynomial of degree 10
// (remainder < 3e-13 relative: four orders below what the chaotic spec model needs), scaled by 2^n with
// v_ldexp_f64.  Valid for |x| < 700.
__device__ __forceinline__ double exp_f64(double x) {
    // n = rint(x log2 e) by the magic-number add: the integer lands in the low mantissa bits of t (|n| < 2^31 here)
    const double kMagic = 6755399441055744.0;                             // 1.5 * 2^52
    const double t = __builtin_fma(x, 1.4426950408889634074, kMagic);
    const double n = t - kMagic;
    double r = __builtin_fma(-n, 6.93147180369123816490e-01, x);          // ln2 split as in fdlibm: hi has 32 bits
    r = __builtin_fma(-n, 1.90821492927058770002e-10, r);
    double p = 1.0 / 3628800.0;                                           // degree 10: remainder r^11 / 11! < 3e-13 relative
    p = __builtin_fma(p, r, 1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, 1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)__builtin_bit_cast(long long, t));     // low dword of t = n (two's complement)
}
// 1 / d for d in [1, 1 + e^20]: the hardware reciprocal refined by two Newton steps (quadratic: whatever the
// seed's accuracy above 2^-14, the result is within an ulp or two)
__device__ __forceinline__ double rcp_f64(double d) {
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-d, y, 1.0);
    return __builtin_fma(y, e, y);
}
// ocropy's sigmoid is 1 / (1 + exp(clip(-x, -20, 20)))   -- in float64 the clip is visible
// (sigma(-25) = 2.06e-9 with it, 1.4e-11 without), so it is kept; tanh(x) = (1 - e) / (1 + e) with e = exp(-2x), |x|
// clamped to 20 (tanh(20) = 1 - 8e-18 rounds to 1).

// One LSTM cell update (`forward_py`) from the four pre-activations of a (line, unit) pair.
// past0: not the first step of the whole sequence (the peepholes on the old cell state and the output peephole are
// skipped at t = 0).
__device__ __forceinline__ double lstm_cell_f64(double gi, double gf, double go, double ci_pre, double& c, bool past0,
                                                double wip, double wfp, double wop) {
    // every gate is a ratio with denominator 1 + e^z; the three of the cell-state update share ONE reciprocal
    // (of the product of their denominators, at most (1 + e^20)^3 ~ 1e26), the two of the output another
    const double cp = past0 ? c : 0.0;
    const double ea = exp_f64(-2.0 * __builtin_fmin(__builtin_fmax(ci_pre, -20.0), 20.0));               // tanh(ci_pre) = (1 - ea) / (1 + ea)
    const double eb = exp_f64(__builtin_fmin(__builtin_fmax(-__builtin_fma(wip, cp, gi), -20.0), 20.0)); // sigma = 1 / (1 + eb), ocropy's clip
    const double ef = exp_f64(__builtin_fmin(__builtin_fmax(-__builtin_fma(wfp, cp, gf), -20.0), 20.0));
    const double pa = (1.0 + ea) * (1.0 + eb), pf = 1.0 + ef;
    const double r3 = rcp_f64(pa * pf);
    const double cn = __builtin_fma((1.0 - ea) * pf, r3, (pa * r3) * cp);      // ci * gi + gf * c
    const double ec = exp_f64(-2.0 * __builtin_fmin(__builtin_fmax(cn, -20.0), 20.0));
    const double eo = exp_f64(__builtin_fmin(__builtin_fmax(-__builtin_fma(past0 ? wop : 0.0, cn, go), -20.0), 20.0));
    c = cn;
    return (1.0 - ec) * rcp_f64((1.0 + ec) * (1.0 + eo));                       // tanh(c) * go
}

// ---------------------------------------------------------------------------------------------
// Kx: Gx[dir][row][4 unit + gate] = sum_kp W_gate[unit][kp] * [1, x_row, 0, 0, 0][kp]     (kp < 52, float64)
struct XprojArgs {
    const float* x;        // [rows][48], the rows of this call
    int64_t rows;
    const double* wx;      // [dir 2][tile 25][k-step 13][lane 64]: B fragments, column j = lane % 16 of the tile = the (unit, gate) whose
                           // gx_index is 16 tile + j: W_gate(2 (j / 8) + j % 2)[unit 4 tile + (j % 8) / 2][kp 4 kstep + lane / 16]
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
    double Bx[NT][kKX];
    {
        const double* wp = a.wx + ((size_t)(dir * kTiles + tile0) * kKX) * 64 + lane;
#pragma unroll
        for (int s = 0; s < NT; ++s)
#pragma unroll
            for (int kk = 0; kk < kKX; ++kk) Bx[s][kk] = wp[((size_t)s * kKX + kk) * 64];
    }
    const int64_t ntiles = (a.rows + 15) / 16;
    const int li = lane & 15, kq = lane >> 4;
    double* gxd = a.gx + (size_t)dir * a.rows * kCols;
    // this lane's A values of a row tile: [1, x, 0 0 0][4 kk + kq] of row tile * 16 + li (rows past the end: clamped)
    auto load_tile = [&](int64_t tile, float (&dst)[kKX]) {
        const int64_t row = min(min(tile, ntiles - 1) * 16 + li, a.rows - 1);
        const float* xr = a.x + row * kNi;
#pragma unroll
        for (int kk = 0; kk < kKX; ++kk) {
            const int kp = 4 * kk + kq;                        // kp = 0: the constant 1; 49 .. 51: zero padding
            dst[kk] = (kp >= 1 && kp <= kNi) ? xr[kp - 1] : (kp == 0 ? 1.0f : 0.0f);
        }
    };
    float A[kKX], An[kKX];
    load_tile(blockIdx.x, A);
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        load_tile(tile + gridDim.x, An);                       // flies under this tile's MFMAs
        f64x4 acc[NT];
#pragma unroll
        for (int s = 0; s < NT; ++s) acc[s] = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < kKX; ++kk) {
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
        for (int kk = 0; kk < kKX; ++kk) A[kk] = An[kk];
    }
}

__global__ __launch_bounds__(kW * 64) void lstm_xproj_f64_kernel(XprojArgs a) {
    const int dir = blockIdx.y, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == 0) xproj_body<7>(a, dir, wave, lane);
    else xproj_body<6>(a, dir, wave, lane);
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
        if (a.c0) {
#pragma unroll
            for (int s = 0; s < NT; ++s) c[s] = a.c0[((size_t)myid * 2 + dir) * kNs + ubase + 4 * s];
            if (kSums) csplit = a.c0[((size_t)myid * 2 + dir) * kNs + usplit];
        }
        if (a.tstart) ts = a.tstart[(size_t)myid * 2 + dir];
    }
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
        const bool past0 = (t > 0) | (ts > 0);
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
            const double h = lstm_cell_f64(acc[s][0], acc[s][1], acc[s][2], acc[s][3], c[s], past0,
                                           peep_s[0][unit], peep_s[1][unit], peep_s[2][unit]);
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
            for (int spin = 0; spin < (1 << 22); ++spin)
                if (__hip_atomic_load(&part_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= want) break;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#pragma unroll
            for (int w = 0; w < kW - 1; ++w)
#pragma unroll
                for (int r = 0; r < 4; ++r) accs[r] += part[w][r][lane];
            const double h = lstm_cell_f64(accs[0], accs[1], accs[2], accs[3], csplit, past0,
                                           peep_s[0][usplit], peep_s[1][usplit], peep_s[2][usplit]);
            hs[nxt][usplit][li] = h;
            if (t < myT) hrow[usplit] = (float)h;
#ifdef TA_F64_PROFILE
            p_move += prof_now() - p0;
#endif
        }
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

}  // namespace ta64

using namespace ta64;

extern "C" int64_t ta_lstm_f64_weight_doubles(int32_t which) {
    // 0: wh [2][4][7][25][64]; 1: wx [2][25][13][64]; 2: peep [2][3][100]
    if (which == 0) return (int64_t)2 * kW * kMaxNT * kKH * 64;
    if (which == 1) return (int64_t)2 * kTiles * kKX * 64;
    if (which == 2) return (int64_t)2 * 3 * kNs;
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
    hipLaunchKernelGGL(lstm_xproj_f64_kernel, grid, dim3(kW * 64), 0, reinterpret_cast<hipStream_t>(stream), a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ta_fail_hip(e, "lstm_xproj_f64_kernel launch");
    return TA_OK;
}

extern "C" int ta_lstm_forward_f64(const double* gx, int64_t gx_row0, int64_t gx_rows, const int64_t* row_off,
                                   const int32_t* T, const int32_t* group_lines, int32_t ngroups, const double* wh,
                                   const double* peep, float* hout, const double* h0, const double* c0,
                                   const int32_t* tstart, void* stream) {
    if (ngroups < 0 || gx_rows < 0 || gx_row0 < 0) return ta_fail(TA_EINVAL, "negative count");
    if (ngroups == 0) return TA_OK;
    if (!gx || !row_off || !T || !group_lines || !wh || !peep || !hout)
        return ta_fail(TA_EINVAL, "null pointer argument");
    if ((h0 != nullptr) != (c0 != nullptr) || (h0 != nullptr) != (tstart != nullptr))
        return ta_fail(TA_EINVAL, "h0, c0 and tstart go together (all null, or all given)");
    Seq64Args a{gx, gx_row0, gx_rows, row_off, T, group_lines, wh, peep, hout, h0, c0, tstart};
    hipLaunchKernelGGL(lstm_seq_f64_kernel, dim3(2 * ngroups), dim3(kW * 64), 0,
                       reinterpret_cast<hipStream_t>(stream), a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ta_fail_hip(e, "lstm_seq_f64_kernel launch");
    return TA_OK;
}
