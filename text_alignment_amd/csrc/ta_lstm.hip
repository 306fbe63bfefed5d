// ta_lstm.hip -- the line recogniser `alignToOCR.py` obtains by shelling out to
// `ocropus-rpred` (reference alignToOCR.py:142-147), on MI355X (gfx950).
//
// Arithmetic restated from ocropy 1.3.3 (third-party, not in the reference tree; SURVEY.md
// Appendix B): 1-layer bidirectional peephole LSTM (ni = 48, ns = 100), softmax output layer,
// threshold / arg-max decode.  Three kernels:
//
//  K3 lstm_seq_kernel   one workgroup (7 waves) per (group of 16 lines, direction).  The whole
//                       weight matrix of a direction [4 gates x 100 units x 149 inputs] lives in
//                       registers as f32 MFMA B-fragments (152 VGPRs per lane), the recurrent
//                       input [1, x_t, h_{t-1}] of the 16 lines is the A operand, staged in a
//                       double-buffered LDS tile; per timestep 38 k-steps x 4 gates of
//                       v_mfma_f32_16x16x4_f32 per wave, the gate non-linearities on the
//                       accumulator registers, one barrier.  f32 in / f32 accumulate (exact
//                       fmaf chain), so logits stay within 1e-3 of the float64 restatement.
//  K4 lstm_output_kernel  Y[rows x 200] . W2^T + bias as an MFMA GEMM with W2 resident in LDS,
//                       fused clip(-100,100) + softmax on the accumulator tile.
//  K5 decode_kernel     one wave per line: runs of P(blank) < threshold, first maximum of the
//                       run over (t, class).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ta_common.h"

namespace ta {

constexpr int kNi = 48;            // input rows (normalised line height)
constexpr int kNs = 100;           // LSTM states per direction
constexpr int kLines = 16;         // lines per workgroup (MFMA M)
constexpr int kWaves = 7;          // 7 x 16 = 112 >= 100 units
constexpr int kXK = 52;            // [1, x(48), 3 zero pads]
constexpr int kKP = kXK + kNs;     // 152 padded inputs
constexpr int kKS = kKP / 4;       // 38 k-steps of 4
constexpr int kKSP = 40;           // k-steps per LDS row, padded for 16-byte reads

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct LstmArgs {
    const float* x;            // [rows][48]
    const int64_t* row_off;    // per line: first row
    const int32_t* T;          // per line: timesteps
    const int32_t* group_lines;// [ngroups][16] line ids, -1 = empty slot
    const float* wp;           // packed B fragments [2][7][4][38][64]
    const float* peep;         // [2][3][112]
    float* hout;               // [rows][200]
    // optional continuation of sequences that were started elsewhere (all three null: fresh lines)
    const float* h0;           // [lines][2][100] outputs before the first step, per direction
    const float* c0;           // [lines][2][100] cell states before the first step
    const int32_t* tstart;     // [lines][2] steps of the sequence already done (> 0: the "[t > 0]" rules apply from step 0)
};

// exp(z) for |z| <= 40 on the hardware exp2 unit (v_exp_f32, ~1 ulp) with a compensated
// argument: z*log2(e) is formed as hi + lo so the 2^-24 relative rounding of the product (which
// the exponential would amplify by |z|) is put back to first order.  ~7 VALU instead of the
// ~20 of the library expf, at the same accuracy for this range.
__device__ __forceinline__ float exp_fast(float z) {
    const float L = 1.44269502162933349609375f;          // float(log2 e)
    const float Llo = 1.925963033500011e-8f;             // log2 e - L
    const float hi = z * L;
    const float lo = fmaf(z, L, -hi) + z * Llo;
    return __builtin_amdgcn_exp2f(hi) * fmaf(lo, 0.693147182464599609375f, 1.0f);
}
// Gate non-linearities: outputs in (-1, 1) that are summed into pre-activations of order one, so
// ABSOLUTE accuracy is what counts.  exp2(z log2 e) on the hardware unit has a relative error of
// ~|z| 2^-24; through sigma (slope <= 1/4) and tanh (slope <= 1) that is < 3e-8 absolute at any z --
// below the rounding of the accumulations.  Minimum instruction count, because the gate math
// (not the matrix pipe) sets the step time of the split-operand kernel and adds to it in the f32
// kernel: no clip (exp2 saturates to 0 / inf and the reciprocal to 1 / 0 exactly where the
// reference's clip(-x, -20, 20) has long stopped mattering in float32: |sigma(20) - 1| = 2e-9), and
// tanh(x) = 2 sigma(2x) - 1 -- four and five instructions, two of them transcendental.  Absolute
// error < 1.2e-7 (the cancellation near 0 is absolute, too).
__device__ __forceinline__ float sigmoid_min(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269502162933349609375f));
}
__device__ __forceinline__ float tanh_min(float x) {
    return fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -2.8853900432586669921875f)), -1.0f);
}

// One LSTM cell update from the four pre-activations of a (line, unit) pair, shared by
// both exact-f32 recurrence kernels so that they round alike: every multiply-add is spelled as the fmaf it is
// meant to be, which leaves the compiler nothing to contract (which of `ci * gi + gf * c`'s two products is
// fused was its choice before, and it chose differently in the two kernels).  past0: not the first step of the
// whole sequence -- the peepholes on the old cell state and the output peephole are skipped at t = 0.
__device__ __forceinline__ float lstm_cell_f32(float gi, float gf, float go, float ci_pre, float& c, bool past0,
                                               float wip, float wfp, float wop) {
    // (the "[t > 0]" rules as selects on the operands: fmaf(w, 0, g) is g, so the arithmetic is straight-line)
    const float ci = tanh_min(ci_pre);
    const float cp = past0 ? c : 0.0f;
    gi = sigmoid_min(fmaf(wip, cp, gi));
    gf = sigmoid_min(fmaf(wfp, cp, gf));
    const float cn = fmaf(ci, gi, gf * cp);                       // (the product with the old state is the rounded one)
    go = sigmoid_min(fmaf(past0 ? wop : 0.0f, cn, go));
    c = cn;
    return tanh_min(cn) * go;
}

// Seven waves own 16 units each (4 gates x 38 k-steps = 152 MFMAs per timestep); on four SIMDs that
// is 2, 2, 2, 1 waves and the SIMDs with two waves set the pace.  An eighth wave takes the CI gate of
// waves 4, 5 and 6 (3 x 38 = 114 MFMAs, the same operands in the same order, so the sums are
// bit-identical) and hands the three accumulator tiles over through LDS: every SIMD then issues
// 266 MFMAs per timestep instead of 304 on the busiest (11.7 -> 11.2 ms per 1920 lines).
constexpr int kSeqWaves = kWaves + 1;
__global__ __launch_bounds__(kSeqWaves * 64) void lstm_seq_kernel(LstmArgs a) {
    __shared__ __attribute__((aligned(16))) float src[2][4][kLines][kKSP];
    __shared__ f32x4 cibuf[3][64];                         // CI accumulator tiles of waves 4..6
    __shared__ int s_line[kLines];
    __shared__ int s_T[kLines];
    __shared__ long long s_row[kLines];

    // groups arrive longest first; both directions of a group are neighbours in dispatch order, so
    // with more workgroups than CUs the long ones start first (longest-processing-time order)
    const int grp = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    if (tid < kLines) {
        const int id = a.group_lines[grp * kLines + tid];
        s_line[tid] = id;
        s_T[tid] = id >= 0 ? a.T[id] : 0;
        s_row[tid] = id >= 0 ? a.row_off[id] : 0;
    }
    for (int e = tid; e < 2 * 4 * kLines * kKSP; e += kSeqWaves * 64) (&src[0][0][0][0])[e] = 0.0f;
    __syncthreads();
    int Tmax = 0;
#pragma unroll
    for (int s = 0; s < kLines; ++s) Tmax = max(Tmax, s_T[s]);

    // weights of this wave's 16 units: B fragments, constant over time
    const bool helper = (wave == kWaves);                   // the eighth wave: no units of its own
    const bool lean = (wave >= 4 && !helper);               // waves 4..6: CI comes from the helper
    float Bf[4][kKS];
    {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            // helper: slot j holds the CI fragments (gate 3) of wave 4 + j; slot 3 stays unused
            const int w = helper ? min(4 + g4, kWaves - 1) : wave;
            const int gate = helper ? 3 : g4;
            const float* wp = a.wp + (((size_t)(dir * kWaves + w) * 4 + gate) * kKS) * 64 + lane;
#pragma unroll
            for (int kk = 0; kk < kKS; ++kk) Bf[g4][kk] = wp[(size_t)kk * 64];
        }
    }
    const int unit = helper ? 127 : wave * 16 + (lane & 15);
    const int pu = helper ? 0 : unit;
    const float wip = a.peep[(dir * 3 + 0) * 112 + pu];
    const float wfp = a.peep[(dir * 3 + 1) * 112 + pu];
    const float wop = a.peep[(dir * 3 + 2) * 112 + pu];

    // rows of the accumulator tile this lane owns: slot = (lane>>4)*4 + r
    int myT[4];
    long long myrow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int slot = (lane >> 4) * 4 + r;
        myT[r] = s_T[slot];
        myrow[r] = s_row[slot];
    }
    float c[4] = {0.f, 0.f, 0.f, 0.f};
    int ts[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int id = s_line[(lane >> 4) * 4 + r];
        if (id >= 0 && !helper) {
            if (a.c0 && unit < kNs) c[r] = a.c0[((size_t)id * 2 + dir) * kNs + unit];
            if (a.tstart) ts[r] = a.tstart[(size_t)id * 2 + dir];
        }
    }
    if (a.h0) {                                            // h_{-1} of continued sequences into the first A tile
        for (int e = tid; e < kLines * kNs; e += kSeqWaves * 64) {
            const int slot = e / kNs, u = e % kNs, kp = kXK + u;
            const int id = s_line[slot];
            if (id >= 0) src[0][kp & 3][slot][kp >> 2] = a.h0[((size_t)id * 2 + dir) * kNs + u];
        }
    }

    // loader role: element e -> (slot, kp) of the x part [16][52]
    constexpr int kXE = kLines * kXK;                      // 832
    auto x_value = [&](int e, int t) -> float {
        const int slot = e / kXK, kp = e % kXK;
        if (kp == 0) return 1.0f;
        if (kp > kNi) return 0.0f;
        const int Tl = s_T[slot];
        if (Tl <= 0) return 0.0f;
        int tt = t < Tl ? t : Tl - 1;
        if (dir) tt = Tl - 1 - tt;                          // Reversed(LSTM): run on xs[::-1]
        return a.x[(s_row[slot] + tt) * kNi + (kp - 1)];
    };
    auto x_store = [&](int e, int buf, float v) {
        const int slot = e / kXK, kp = e % kXK;
        src[buf][kp & 3][slot][kp >> 2] = v;
    };
    {
        for (int e = tid; e < kXE; e += kSeqWaves * 64) x_store(e, 0, x_value(e, 0));
    }
    __syncthreads();

    const int e0 = tid, e1 = tid + kSeqWaves * 64;         // 512 + 512 >= 832
    for (int t = 0; t < Tmax; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        // prefetch next step's inputs (global loads fly under the MFMAs)
        float xn0 = 0.f, xn1 = 0.f;
        if (t + 1 < Tmax) {
            xn0 = x_value(e0, t + 1);
            if (e1 < kXE) xn1 = x_value(e1, t + 1);
        }
        // A fragments: src[cur][k = lane>>4][line = lane&15][kk]
        float A[kKS];                                        // 9 x 16 bytes + 8 bytes: no dead registers
        {
            const float* arow = &src[cur][lane >> 4][lane & 15][0];
            const f32x4* ap = reinterpret_cast<const f32x4*>(arow);
#pragma unroll
            for (int q = 0; q < kKS / 4; ++q) {
                const f32x4 v = ap[q];
                A[4 * q] = v[0]; A[4 * q + 1] = v[1]; A[4 * q + 2] = v[2]; A[4 * q + 3] = v[3];
            }
            const float2 tail = *reinterpret_cast<const float2*>(arow + 4 * (kKS / 4));
            A[kKS - 2] = tail.x; A[kKS - 1] = tail.y;
        }
        f32x4 acc[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) acc[g4] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (wave < 4) {
#pragma unroll
            for (int kk = 0; kk < kKS; ++kk) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
                    acc[g4] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[kk], Bf[g4][kk], acc[g4], 0, 0, 0);
            }
        } else {                                            // waves 4..6: GI, GF, GO; helper: three CI tiles
#pragma unroll
            for (int kk = 0; kk < kKS; ++kk) {
#pragma unroll
                for (int g4 = 0; g4 < 3; ++g4)
                    acc[g4] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[kk], Bf[g4][kk], acc[g4], 0, 0, 0);
            }
        }
        // next step's inputs go to LDS BEFORE this step's output stores are issued: the wait for
        // the prefetched loads would otherwise also wait for those stores (vmcnt counts both)
        if (t + 1 < Tmax) {
            x_store(e0, nxt, xn0);
            if (e1 < kXE) x_store(e1, nxt, xn1);
        }
        // hand-over of the CI tiles computed by the helper wave.  A full barrier on purpose: with a
        // progress word that only waves 4..6 wait for, the early waves' gate math contends with the
        // late waves' MFMAs (the f32 MFMA shares the VALU datapath) and the step gets slower
        if (helper) {
#pragma unroll
            for (int j = 0; j < 3; ++j) cibuf[j][lane] = acc[j];
        }
        __syncthreads();
        if (lean) acc[3] = cibuf[wave - 4][lane];
        // gates: acc[0..3] = WGI, WGF, WGO, WCI . src
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool past0 = (t > 0) | (ts[r] > 0);             // not the first step of the whole sequence
            const float h = lstm_cell_f32(acc[0][r], acc[1][r], acc[2][r], acc[3][r], c[r], past0, wip, wfp, wop);
            if (unit < kNs) {
                const int slot = (lane >> 4) * 4 + r;
                const int kp = kXK + unit;
                src[nxt][kp & 3][slot][kp >> 2] = h;
                if (t < myT[r]) {
                    const int tt = dir ? myT[r] - 1 - t : t;
                    a.hout[(myrow[r] + tt) * (2 * kNs) + dir * kNs + unit] = h;
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// K3 for SMALL batches: the same recurrence, bit for bit, on groups of FOUR lines.
//
// The recurrence is a chain of T dependent steps per group, and a step of the 16-line kernel above costs
// what its 16 x 16 x 4 MFMA tiles cost however many of the 16 rows are lines: a page of 30 lines is two
// groups, and even 1 920 lines are only 240 workgroups on 256 CUs -- the longest group sets the time and
// most CUs idle for part of it.  v_mfma_f32_4x4x1_16B_f32 computes sixteen 4 x 4 outer products per
// instruction (one k per instruction, 8 cycles instead of 32): with A broadcast from one block (cbsz = 4:
// the four LINES' inputs at step k) and B holding the weights of 16 units x 4 gates at that k, a wave
// advances four lines at a quarter of the cost -- four times as many, four times shorter workgroups, which
// pack onto the CUs (and a lone page finishes in a quarter of the time).  One k per instruction in
// ascending order is the fmaf chain the 16 x 16 x 4 form computes as well (tools/ubench/mfma4x4.hip: all
// 256 outputs of a 152-term product identical to the host's fmaf chain in both forms), the gate functions
// are the same instructions, so the outputs are the 16-line kernel's to the bit
// (tests/test_ocr_gpu.py::test_four_line_groups_equal_sixteen_line_groups).
//
// Layout.  A wave owns 16 units; lane = 4 * b + j holds, of the B operand, the weight of (unit 16w + b,
// gate j) -- blocks are units, the 4 columns of a block the 4 gates -- so the accumulator register r of lane
// (b, j) is the pre-activation of gate j, unit b, LINE r.  A 4 x 4 transpose inside each quad of lanes (two
// rounds of DPP quad permutes) turns that into: lane (b, j) holds all four gates of unit b for line j, and
// does the gate math of that one (line, unit) pair.  The A operand of 16 consecutive k is ONE register
// (lane = 4 * (k % 16) + line; `abid` = k % 16 selects the block to broadcast): ten LDS reads per step.
constexpr int kG4 = 4;              // lines per workgroup
constexpr int kKC4 = (kKP + 15) / 16;   // 10 A registers of 16 k each

template <int CTRL>
__device__ __forceinline__ float quad_perm(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

__global__ __launch_bounds__(kWaves * 64) void lstm_seq4_kernel(LstmArgs a) {
    // A tile in LDS, [k][line]: the x rows (k < 52: 1, x_t, zeros) three deep, the h rows (k >= 52) two deep
    __shared__ __attribute__((aligned(16))) float xs[3][256];              // 52 x 4 used; a 64-lane read of chunk 3 stays inside
    __shared__ __attribute__((aligned(16))) float hs[2][448];              // 112 units x 4 lines
    __shared__ int s_line[kG4];
    __shared__ int s_T[kG4];
    __shared__ long long s_row[kG4];

    const int grp = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < kG4) {
        const int id = a.group_lines[grp * kG4 + tid];
        s_line[tid] = id;
        s_T[tid] = id >= 0 ? a.T[id] : 0;
        s_row[tid] = id >= 0 ? a.row_off[id] : 0;
    }
    for (int e = tid; e < 3 * 256; e += kWaves * 64) (&xs[0][0])[e] = 0.0f;
    for (int e = tid; e < 2 * 448; e += kWaves * 64) (&hs[0][0])[e] = 0.0f;
    __syncthreads();
    int Tmax = 0;
#pragma unroll
    for (int s = 0; s < kG4; ++s) Tmax = max(Tmax, s_T[s]);

    // weights: Bf[k][lane] = W_gate(lane % 4)[unit 16 * wave + lane / 4][k]
    float Bf[kKP];
    {
        const float* wp = a.wp + ((size_t)(dir * kWaves + wave) * kKP) * 64 + lane;
#pragma unroll
        for (int k = 0; k < kKP; ++k) Bf[k] = wp[(size_t)k * 64];
    }
    const int unit = wave * 16 + (lane >> 2), slot = lane & 3;      // this lane's (unit, line) after the transpose
    const float wip = a.peep[(dir * 3 + 0) * 112 + unit];
    const float wfp = a.peep[(dir * 3 + 1) * 112 + unit];
    const float wop = a.peep[(dir * 3 + 2) * 112 + unit];
    const int myT = s_T[slot];
    const long long myrow = s_row[slot];
    float c = 0.f;
    int ts = 0;
    {
        const int id = s_line[slot];
        if (id >= 0) {
            if (a.c0 && unit < kNs) c = a.c0[((size_t)id * 2 + dir) * kNs + unit];
            if (a.tstart) ts = a.tstart[(size_t)id * 2 + dir];
        }
    }
    if (a.h0) {                                            // h_{-1} of continued sequences into the first A tile
        for (int e = tid; e < kG4 * kNs; e += kWaves * 64) {
            const int sl = e / kNs, u = e % kNs;
            const int id = s_line[sl];
            if (id >= 0) hs[0][u * kG4 + sl] = a.h0[((size_t)id * 2 + dir) * kNs + u];
        }
    }
    // loader role: thread e < 4 * 48 -> (line e / 48, input row e % 48); the constant row k = 0 is set once
    if (tid < 3 * kG4) xs[tid / kG4][tid % kG4] = 1.0f;
    // loader role: wave 3 -- alone on its SIMD (waves w and w + 4 share one), so the wait for its loads costs
    // nobody else's issue slots -- lane l takes elements l, l + 64, l + 128 of the 4 x 48 inputs of a step:
    // element e -> (line e / 48, input row e % 48)
    const bool loader = wave == 3;
    auto x_value = [&](int i, int t) -> float {
        const int e = lane + 64 * i, sl = e / kNi;
        const int Tl = s_T[sl];
        if (Tl <= 0) return 0.0f;
        int tt = t < Tl ? t : Tl - 1;
        if (dir) tt = Tl - 1 - tt;                          // Reversed(LSTM): run on xs[::-1]
        return a.x[(s_row[sl] + tt) * kNi + e % kNi];
    };
    auto x_store = [&](int buf, int i, float v) {
        const int e = lane + 64 * i;
        xs[buf][(1 + e % kNi) * kG4 + e / kNi] = v;
    };
    if (loader) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            x_store(0, i, x_value(i, 0));
            if (Tmax > 1) x_store(1, i, x_value(i, 1));
        }
    }
    __syncthreads();

    // A step's products come in two parts: k < 49 over [1, x_t] (k = 49 .. 51 are zero padding of both operands:
    // adding +0 to an accumulator that is never -0 changes nothing, so those three are not issued) and k >= 52 over
    // h_{t-1}; only the second waits for the previous step.  A wave's accumulator is ONE dependent chain of MFMAs
    // (12.6 cycles each when the wave is alone on the pipe, 8 when something fills the gaps), so the x part of step
    // t + 1 -- a second, independent accumulator -- is issued INTERLEAVED with the h part of step t, two h products
    // to one x product: x_{t+1} sits in LDS one step early for that (xs is three deep: x_t is gone, x_{t+1} is being
    // multiplied, x_{t+2} is being stored) and the row prefetched from HBM is x_{t+2}.  Either accumulator takes
    // its x part first (into zeros), then its h part, k ascending -- the order of the 16-line kernel: same bits.
#define TA_M4(ac, q, kk) ac = __builtin_amdgcn_mfma_f32_4x4x1f32(A[q], Bf[16 * (q) + (kk)], ac, 4, kk, 0);
    static_assert(kXK == 52 && kKP == 152 && kNi == 48, "the spelled-out k ranges below");
    f32x4 accn = {0.f, 0.f, 0.f, 0.f};
    {                                                         // x part of step 0
        float A[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) A[q] = xs[0][64 * q + lane];
        f32x4 ac = accn;
        TA_M4(ac, 0, 0) TA_M4(ac, 0, 1) TA_M4(ac, 0, 2) TA_M4(ac, 0, 3)
        TA_M4(ac, 0, 4) TA_M4(ac, 0, 5) TA_M4(ac, 0, 6) TA_M4(ac, 0, 7)
        TA_M4(ac, 0, 8) TA_M4(ac, 0, 9) TA_M4(ac, 0, 10) TA_M4(ac, 0, 11)
        TA_M4(ac, 0, 12) TA_M4(ac, 0, 13) TA_M4(ac, 0, 14) TA_M4(ac, 0, 15)
        TA_M4(ac, 1, 0) TA_M4(ac, 1, 1) TA_M4(ac, 1, 2) TA_M4(ac, 1, 3)
        TA_M4(ac, 1, 4) TA_M4(ac, 1, 5) TA_M4(ac, 1, 6) TA_M4(ac, 1, 7)
        TA_M4(ac, 1, 8) TA_M4(ac, 1, 9) TA_M4(ac, 1, 10) TA_M4(ac, 1, 11)
        TA_M4(ac, 1, 12) TA_M4(ac, 1, 13) TA_M4(ac, 1, 14) TA_M4(ac, 1, 15)
        TA_M4(ac, 2, 0) TA_M4(ac, 2, 1) TA_M4(ac, 2, 2) TA_M4(ac, 2, 3)
        TA_M4(ac, 2, 4) TA_M4(ac, 2, 5) TA_M4(ac, 2, 6) TA_M4(ac, 2, 7)
        TA_M4(ac, 2, 8) TA_M4(ac, 2, 9) TA_M4(ac, 2, 10) TA_M4(ac, 2, 11)
        TA_M4(ac, 2, 12) TA_M4(ac, 2, 13) TA_M4(ac, 2, 14) TA_M4(ac, 2, 15)
        TA_M4(ac, 3, 0)
        accn = ac;
    }
    int b1 = 1, b2 = 2, b3 = 0;                               // xs buffers of x_{t+1}, x_{t+2} and the one after

    for (int t = 0; t < Tmax; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        float xn[3] = {0.f, 0.f, 0.f};
        if (loader && t + 2 < Tmax) {                         // flies under the MFMAs
#pragma unroll
            for (int i = 0; i < 3; ++i) xn[i] = x_value(i, t + 2);
        }
        f32x4 acc = accn;
        accn = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            float A[kKC4];                                    // chunks 0 .. 3 (lanes < 16 of chunk 3): x rows; 3 .. 9: h rows
#pragma unroll
            for (int q = 0; q < 3; ++q) A[q] = xs[b1][64 * q + lane];
            // chunk 3 = k 48 .. 63: lanes < 16 are x rows (k 48 .. 51), lanes >= 16 h rows 0 .. 11
            A[3] = lane < 16 ? xs[b1][192 + lane] : hs[cur][max(lane - 16, 0)];
#pragma unroll
            for (int q = 4; q < kKC4; ++q) A[q] = hs[cur][64 * q - 4 * kXK + lane];
            TA_M4(acc, 3, 4) TA_M4(acc, 3, 5) TA_M4(accn, 0, 0) TA_M4(acc, 3, 6)
            TA_M4(acc, 3, 7) TA_M4(accn, 0, 1) TA_M4(acc, 3, 8) TA_M4(acc, 3, 9)
            TA_M4(accn, 0, 2) TA_M4(acc, 3, 10) TA_M4(acc, 3, 11) TA_M4(accn, 0, 3)
            TA_M4(acc, 3, 12) TA_M4(acc, 3, 13) TA_M4(accn, 0, 4) TA_M4(acc, 3, 14)
            TA_M4(acc, 3, 15) TA_M4(accn, 0, 5) TA_M4(acc, 4, 0) TA_M4(acc, 4, 1)
            TA_M4(accn, 0, 6) TA_M4(acc, 4, 2) TA_M4(acc, 4, 3) TA_M4(accn, 0, 7)
            TA_M4(acc, 4, 4) TA_M4(acc, 4, 5) TA_M4(accn, 0, 8) TA_M4(acc, 4, 6)
            TA_M4(acc, 4, 7) TA_M4(accn, 0, 9) TA_M4(acc, 4, 8) TA_M4(acc, 4, 9)
            TA_M4(accn, 0, 10) TA_M4(acc, 4, 10) TA_M4(acc, 4, 11) TA_M4(accn, 0, 11)
            TA_M4(acc, 4, 12) TA_M4(acc, 4, 13) TA_M4(accn, 0, 12) TA_M4(acc, 4, 14)
            TA_M4(acc, 4, 15) TA_M4(accn, 0, 13) TA_M4(acc, 5, 0) TA_M4(acc, 5, 1)
            TA_M4(accn, 0, 14) TA_M4(acc, 5, 2) TA_M4(acc, 5, 3) TA_M4(accn, 0, 15)
            TA_M4(acc, 5, 4) TA_M4(acc, 5, 5) TA_M4(accn, 1, 0) TA_M4(acc, 5, 6)
            TA_M4(acc, 5, 7) TA_M4(accn, 1, 1) TA_M4(acc, 5, 8) TA_M4(acc, 5, 9)
            TA_M4(accn, 1, 2) TA_M4(acc, 5, 10) TA_M4(acc, 5, 11) TA_M4(accn, 1, 3)
            TA_M4(acc, 5, 12) TA_M4(acc, 5, 13) TA_M4(accn, 1, 4) TA_M4(acc, 5, 14)
            TA_M4(acc, 5, 15) TA_M4(accn, 1, 5) TA_M4(acc, 6, 0) TA_M4(acc, 6, 1)
            TA_M4(accn, 1, 6) TA_M4(acc, 6, 2) TA_M4(acc, 6, 3) TA_M4(accn, 1, 7)
            TA_M4(acc, 6, 4) TA_M4(acc, 6, 5) TA_M4(accn, 1, 8) TA_M4(acc, 6, 6)
            TA_M4(acc, 6, 7) TA_M4(accn, 1, 9) TA_M4(acc, 6, 8) TA_M4(acc, 6, 9)
            TA_M4(accn, 1, 10) TA_M4(acc, 6, 10) TA_M4(acc, 6, 11) TA_M4(accn, 1, 11)
            TA_M4(acc, 6, 12) TA_M4(acc, 6, 13) TA_M4(accn, 1, 12) TA_M4(acc, 6, 14)
            TA_M4(acc, 6, 15) TA_M4(accn, 1, 13) TA_M4(acc, 7, 0) TA_M4(acc, 7, 1)
            TA_M4(accn, 1, 14) TA_M4(acc, 7, 2) TA_M4(acc, 7, 3) TA_M4(accn, 1, 15)
            TA_M4(acc, 7, 4) TA_M4(acc, 7, 5) TA_M4(accn, 2, 0) TA_M4(acc, 7, 6)
            TA_M4(acc, 7, 7) TA_M4(accn, 2, 1) TA_M4(acc, 7, 8) TA_M4(acc, 7, 9)
            TA_M4(accn, 2, 2) TA_M4(acc, 7, 10) TA_M4(acc, 7, 11) TA_M4(accn, 2, 3)
            TA_M4(acc, 7, 12) TA_M4(acc, 7, 13) TA_M4(accn, 2, 4) TA_M4(acc, 7, 14)
            TA_M4(acc, 7, 15) TA_M4(accn, 2, 5) TA_M4(acc, 8, 0) TA_M4(acc, 8, 1)
            TA_M4(accn, 2, 6) TA_M4(acc, 8, 2) TA_M4(acc, 8, 3) TA_M4(accn, 2, 7)
            TA_M4(acc, 8, 4) TA_M4(acc, 8, 5) TA_M4(accn, 2, 8) TA_M4(acc, 8, 6)
            TA_M4(acc, 8, 7) TA_M4(accn, 2, 9) TA_M4(acc, 8, 8) TA_M4(acc, 8, 9)
            TA_M4(accn, 2, 10) TA_M4(acc, 8, 10) TA_M4(acc, 8, 11) TA_M4(accn, 2, 11)
            TA_M4(acc, 8, 12) TA_M4(acc, 8, 13) TA_M4(accn, 2, 12) TA_M4(acc, 8, 14)
            TA_M4(acc, 8, 15) TA_M4(accn, 2, 13) TA_M4(acc, 9, 0) TA_M4(acc, 9, 1)
            TA_M4(accn, 2, 14) TA_M4(acc, 9, 2) TA_M4(acc, 9, 3) TA_M4(accn, 2, 15)
            TA_M4(acc, 9, 4) TA_M4(acc, 9, 5) TA_M4(accn, 3, 0) TA_M4(acc, 9, 6)
            TA_M4(acc, 9, 7)
        }
        // before this step's output stores are issued: the wait for the prefetched load would otherwise also wait
        // for those stores (vmcnt counts both)
        if (loader && t + 2 < Tmax) {
#pragma unroll
            for (int i = 0; i < 3; ++i) x_store(b2, i, xn[i]);
        }
        // transpose inside the quad: acc[r] of lane j = (gate j, line r)  ->  g[q] of lane j = (gate q, line j)
        float g0 = acc[0], g1 = acc[1], g2 = acc[2], g3 = acc[3];
        {
            const bool odd = lane & 1, hi = lane & 2;
            float s, r;
            s = odd ? g0 : g1; r = quad_perm<0xB1>(s); if (odd) g0 = r; else g1 = r;      // lanes j ^ 1, registers 0 <-> 1
            s = odd ? g2 : g3; r = quad_perm<0xB1>(s); if (odd) g2 = r; else g3 = r;      //             registers 2 <-> 3
            s = hi ? g0 : g2; r = quad_perm<0x4E>(s); if (hi) g0 = r; else g2 = r;        // lanes j ^ 2, registers 0 <-> 2
            s = hi ? g1 : g3; r = quad_perm<0x4E>(s); if (hi) g1 = r; else g3 = r;        //             registers 1 <-> 3
        }
        // gates, the 16-line kernel's arithmetic: g0..g3 = WGI, WGF, WGO, WCI . src
        {
            const bool past0 = (t > 0) | (ts > 0);                  // not the first step of the whole sequence
            const float h = lstm_cell_f32(g0, g1, g2, g3, c, past0, wip, wfp, wop);
            if (unit < kNs) {
                hs[nxt][unit * kG4 + slot] = h;
                if (t < myT) {
                    const int tt = dir ? myT - 1 - t : t;
                    a.hout[(myrow + tt) * (2 * kNs) + dir * kNs + unit] = h;
                }
            }
        }
        __syncthreads();
        const int bt = b1; b1 = b2; b2 = b3; b3 = bt;
    }
#undef TA_M4
}

// ---------------------------------------------------------------------------------------------
// K3 (split operands): the same recurrence on the 16-bit matrix cores.  v_mfma_f32_16x16x32_bf16 /
// _f16 run at 16x the rate of the f32-input MFMA, so f32 operands are split into 16-bit terms whose
// products are accumulated in f32:
//   weights      W = W_hi (bf16, 8 significant bits) + W_r (fp16 of the rest, 11 more: |W - W_hi - W_r| <= 2^-20 |W|)
//   activations  a = a_hi + a_mid + a_lo (bf16, exact to 24 bits);  a16 = fp16(a) (11 bits, only ever multiplied by W_r)
//   W . a  ~  W_hi . (a_lo + a_mid + a_hi)  +  W_r . a16          4 MFMAs per k-step, relative error ~2^-19
// i.e. 80 MFMAs of 16 cycles per wave and timestep instead of 152 of 32.  (The first version of this
// mode kept W_r in bf16 too -- 16 weight bits, five products -- and was 1.25x slower and 8x less
// exact.)  The split of h is done once where it is produced (gate math) and stored as four 16-bit
// planes in LDS, so consumers load ready A fragments (one ds_read_b128 per plane and k-step).
constexpr int kKP2 = 160;          // [1, x(48), 3 pads, h(100), 8 pads]
constexpr int kKS2 = kKP2 / 32;    // 5 k-steps of 32
constexpr int kRS2 = 168;          // LDS row stride in bf16 (336 B: conflict-free b128 reads)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int kPlanes = 4;         // activation planes in LDS: bf16 hi, mid, lo and fp16

struct Split4 { unsigned short hi, mid, lo, h16; };
__device__ __forceinline__ Split4 split4(float v) {
    const __bf16 h = (__bf16)v;
    const float r1 = v - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    const __bf16 l = (__bf16)r2;
    const _Float16 f = (_Float16)v;
    return {__builtin_bit_cast(unsigned short, h), __builtin_bit_cast(unsigned short, m),
            __builtin_bit_cast(unsigned short, l), __builtin_bit_cast(unsigned short, f)};
}

__global__ __launch_bounds__(kWaves * 64) void lstm_seq_split_kernel(LstmArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short srcp[2][kPlanes][kLines][kRS2];
    __shared__ int s_line[kLines];
    __shared__ int s_T[kLines];
    __shared__ long long s_row[kLines];

    // groups arrive longest first; both directions of a group are neighbours in dispatch order, so
    // with more workgroups than CUs the long ones start first (longest-processing-time order)
    const int grp = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    if (tid < kLines) {
        const int id = a.group_lines[grp * kLines + tid];
        s_line[tid] = id;
        s_T[tid] = id >= 0 ? a.T[id] : 0;
        s_row[tid] = id >= 0 ? a.row_off[id] : 0;
    }
    for (int e = tid; e < 2 * kPlanes * kLines * kRS2; e += kWaves * 64) (&srcp[0][0][0][0])[e] = 0;
    __syncthreads();
    int Tmax = 0;
#pragma unroll
    for (int s = 0; s < kLines; ++s) Tmax = max(Tmax, s_T[s]);

    // B fragments: [dir][wave][plane 2: W_hi bf16, W_r fp16][gate 4][kstep 5][lane 64] x 8 halves (16 bytes)
    bf16x8 Bf[2][4][kKS2];
    {
        const uint4* wp = reinterpret_cast<const uint4*>(a.wp) +
                          ((size_t)(dir * kWaves + wave) * 2 * 4 * kKS2) * 64 + lane;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                for (int ks = 0; ks < kKS2; ++ks) {
                    const uint4 v = wp[((size_t)(pl * 4 + g4) * kKS2 + ks) * 64];
                    Bf[pl][g4][ks] = __builtin_bit_cast(bf16x8, v);
                }
    }
    const int unit = wave * 16 + (lane & 15);
    const float wip = a.peep[(dir * 3 + 0) * 112 + unit];
    const float wfp = a.peep[(dir * 3 + 1) * 112 + unit];
    const float wop = a.peep[(dir * 3 + 2) * 112 + unit];

    int myT[4];
    long long myrow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int slot = (lane >> 4) * 4 + r;
        myT[r] = s_T[slot];
        myrow[r] = s_row[slot];
    }
    float c[4] = {0.f, 0.f, 0.f, 0.f};
    int ts[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int id = s_line[(lane >> 4) * 4 + r];
        if (id >= 0) {
            if (a.c0 && unit < kNs) c[r] = a.c0[((size_t)id * 2 + dir) * kNs + unit];
            if (a.tstart) ts[r] = a.tstart[(size_t)id * 2 + dir];
        }
    }
    if (a.h0) {
        for (int e = tid; e < kLines * kNs; e += kWaves * 64) {
            const int slot = e / kNs, u = e % kNs;
            const int id = s_line[slot];
            if (id >= 0) {
                const Split4 sp = split4(a.h0[((size_t)id * 2 + dir) * kNs + u]);
                srcp[0][0][slot][kXK + u] = sp.hi;
                srcp[0][1][slot][kXK + u] = sp.mid;
                srcp[0][2][slot][kXK + u] = sp.lo;
                srcp[0][3][slot][kXK + u] = sp.h16;
            }
        }
    }

    constexpr int kXE = kLines * kXK;                      // 832 elements of [16][52]
    auto x_value = [&](int e, int t) -> float {
        const int slot = e / kXK, kp = e % kXK;
        if (kp == 0) return 1.0f;
        if (kp > kNi) return 0.0f;
        const int Tl = s_T[slot];
        if (Tl <= 0) return 0.0f;
        int tt = t < Tl ? t : Tl - 1;
        if (dir) tt = Tl - 1 - tt;
        return a.x[(s_row[slot] + tt) * kNi + (kp - 1)];
    };
    auto x_store = [&](int e, int buf, float v) {
        const int slot = e / kXK, kp = e % kXK;
        const Split4 sp = split4(v);
        srcp[buf][0][slot][kp] = sp.hi;
        srcp[buf][1][slot][kp] = sp.mid;
        srcp[buf][2][slot][kp] = sp.lo;
        srcp[buf][3][slot][kp] = sp.h16;
    };
    for (int e = tid; e < kXE; e += kWaves * 64) x_store(e, 0, x_value(e, 0));
    __syncthreads();

    const int e0 = tid, e1 = tid + kWaves * 64;
    for (int t = 0; t < Tmax; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        float xn0 = 0.f, xn1 = 0.f;
        if (t + 1 < Tmax) {
            xn0 = x_value(e0, t + 1);
            if (e1 < kXE) xn1 = x_value(e1, t + 1);
        }
        f32x4 acc[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) acc[g4] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < kKS2; ++ks) {
            const int off = 32 * ks + 8 * (lane >> 4);
            const bf16x8 ahi = *reinterpret_cast<const bf16x8*>(&srcp[cur][0][lane & 15][off]);
            const bf16x8 amid = *reinterpret_cast<const bf16x8*>(&srcp[cur][1][lane & 15][off]);
            const bf16x8 alo = *reinterpret_cast<const bf16x8*>(&srcp[cur][2][lane & 15][off]);
            const f16x8 a16 = *reinterpret_cast<const f16x8*>(&srcp[cur][3][lane & 15][off]);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                f32x4 v = acc[g4];                          // small terms first
                v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo, Bf[0][g4][ks], v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16, __builtin_bit_cast(f16x8, Bf[1][g4][ks]), v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(amid, Bf[0][g4][ks], v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, Bf[0][g4][ks], v, 0, 0, 0);
                acc[g4] = v;
            }
        }
        if (t + 1 < Tmax) {                    // before the output stores (see the f32 kernel)
            x_store(e0, nxt, xn0);
            if (e1 < kXE) x_store(e1, nxt, xn1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float gi = acc[0][r], gf = acc[1][r], go = acc[2][r];
            const float ci = tanh_min(acc[3][r]);
            const bool past0 = (t > 0) | (ts[r] > 0);
            if (past0) { gi += wip * c[r]; gf += wfp * c[r]; }
            gi = sigmoid_min(gi);
            gf = sigmoid_min(gf);
            float cn = ci * gi;
            if (past0) { cn += gf * c[r]; go += wop * cn; }
            go = sigmoid_min(go);
            const float h = tanh_min(cn) * go;
            c[r] = cn;
            if (unit < kNs) {
                const int slot = (lane >> 4) * 4 + r;
                const Split4 sp = split4(h);
                srcp[nxt][0][slot][kXK + unit] = sp.hi;
                srcp[nxt][1][slot][kXK + unit] = sp.mid;
                srcp[nxt][2][slot][kXK + unit] = sp.lo;
                srcp[nxt][3][slot][kXK + unit] = sp.h16;
                if (t < myT[r]) {
                    const int tt = dir ? myT[r] - 1 - t : t;
                    a.hout[(myrow[r] + tt) * (2 * kNs) + dir * kNs + unit] = h;
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// K4: z = W2 . [1, y_t]  (No x 201), p = softmax(clip(z, -100, 100))   (Appendix B.4)
//
// One wave per tile of 16 timesteps.  The sum over k may run in any order, so lane
// (row = lane & 15, kq = lane >> 4) takes the CONTIGUOUS quarter y[row][50*kq .. 50*kq + 49] of
// its row straight from HBM into registers (no LDS staging of A); k-step kk of the MFMA then
// multiplies y[.][50*kq + kk] with the matching row of W2^T, which sits in LDS in that order.
// The bias column seeds the accumulators.  Softmax runs on the accumulator tile; besides the
// probabilities the kernel can emit a 16-byte per-timestep summary (P(blank), best probability,
// best class) that is all the decoder needs.
constexpr int kOKS = 50;           // k-steps: 200 = 4 x 50
constexpr int kOWaves = 4;
constexpr int kMaxCT = 8;          // up to 128 classes

struct OutArgs {
    const float* y;        // [rows][200]
    int64_t rows;
    const float* w2p;      // [1 + 200][nop]: row 0 = bias, row 1 + 4*kk + kq = W2[:, 1 + 50*kq + kk]
    int no, nct;
    float* probs;          // optional [rows][no]
    float* logits;         // optional [rows][no]
    float4* summary;       // optional [rows]: {P(class 0), best P, best class as float bits, 0}
};

// reductions over the 16 lanes of a DPP row (= the 16 class columns of an accumulator row): two
// quad permutes, then half-row and row mirrors; every lane ends with the result.  No LDS traffic
// (__shfl_xor compiles to ds_bpermute_b32).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f<kDppXor1>(v));
    v = fmaxf(v, dpp_f<kDppXor2>(v));
    v = fmaxf(v, dpp_f<kDppHalfMirror>(v));
    return fmaxf(v, dpp_f<kDppMirror>(v));
}
__device__ __forceinline__ float row16_sum(float v) {
    // fixed association ((lane pairs) quads) halves) row: the same tree on every lane
    v += dpp_f<kDppXor1>(v);
    v += dpp_f<kDppXor2>(v);
    v += dpp_f<kDppHalfMirror>(v);
    return v + dpp_f<kDppMirror>(v);
}
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_max_u64(unsigned long long k) {
    const unsigned lo = dpp_u<CTRL>((unsigned)k), hi = dpp_u<CTRL>((unsigned)(k >> 32));
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    return o > k ? o : k;
}
__device__ __forceinline__ unsigned long long row16_max_u64(unsigned long long k) {
    k = dpp_max_u64<kDppXor1>(k);
    k = dpp_max_u64<kDppXor2>(k);
    k = dpp_max_u64<kDppHalfMirror>(k);
    return dpp_max_u64<kDppMirror>(k);
}

// clip + softmax (+ summary) of one 16-row accumulator tile: rows (lane>>4)*4 + r, classes
// ct*16 + (lane&15).  Shared by the f32 and the split-operand output kernels.
struct OutSinks { int64_t rows; int no; float* probs; float* logits; float4* summary; };
template <int NCT>
__device__ __forceinline__ void softmax_tile(f32x4 (&acc)[NCT], int64_t r0, const OutSinks& a, int lane) {
    // Only the LAST class tile can hold classes >= no, and whether a lane's class there exists is one bit per
    // lane: it is applied with selects (a masked class enters the maximum as -3e38, the sum and the arg-max as
    // nothing), so the arithmetic below is straight-line code -- the first form tested `cls < no` around every
    // use and compiled to some seventy exec-masked branch regions per tile, as long as the tile's MFMAs.
    const int rr = lane & 15;
    const bool lastv = (NCT - 1) * 16 + rr < a.no;
    const bool want_logits = a.logits != nullptr, want_probs = a.probs != nullptr;      // wave-uniform
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t row = r0 + (lane >> 4) * 4 + r;
        const bool row_ok = row < a.rows;
        if (want_logits && row_ok) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
                if (ct < NCT - 1 || lastv) a.logits[row * a.no + ct * 16 + rr] = acc[ct][r];
        }
        float zmax = -3.0e38f;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            float z = fminf(fmaxf(acc[ct][r], -100.f), 100.f);
            if (ct == NCT - 1) z = lastv ? z : -3.0e38f;
            acc[ct][r] = z;
            zmax = fmaxf(zmax, z);
        }
        zmax = row16_max(zmax);
        float sum = 0.f;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            float e = exp_fast(fmaxf(acc[ct][r] - zmax, -87.0f));
            if (ct == NCT - 1) e = lastv ? e : 0.0f;
            acc[ct][r] = e;
            sum += e;
        }
        sum = row16_sum(sum);
        const float inv = 1.0f / sum;
        unsigned long long key = 0ull;           // larger P wins, then the smaller class
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const float pr = acc[ct][r] * inv;
            acc[ct][r] = pr;
            unsigned long long k = ((unsigned long long)__float_as_uint(pr) << 32) | (unsigned)(0xFFFFFFFFu - (ct * 16 + rr));
            if (ct == NCT - 1) k = lastv ? k : 0ull;
            key = k > key ? k : key;
        }
        if (want_probs && row_ok) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
                if (ct < NCT - 1 || lastv) a.probs[row * a.no + ct * 16 + rr] = acc[ct][r];
        }
        if (a.summary) {
            key = row16_max_u64(key);
            if (rr == 0 && row_ok) {             // lane rr == 0 holds P(class 0) in acc[0]
                const unsigned cls = 0xFFFFFFFFu - (unsigned)key;
                a.summary[row] = make_float4(acc[0][r], __uint_as_float((unsigned)(key >> 32)),
                                             __uint_as_float(cls), 0.f);
            }
        }
    }
}
template <int NCT>
__global__ __launch_bounds__(kOWaves * 64) void lstm_output_kernel(OutArgs a) {
    extern __shared__ __attribute__((aligned(16))) float osm[];
    constexpr int nop = NCT * 16;
    float* w2 = osm;                                        // [201][nop]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < 201 * nop; e += kOWaves * 64) w2[e] = a.w2p[e];
    __syncthreads();
    const int kq = lane >> 4, rr = lane & 15;
    const OutSinks sinks{a.rows, a.no, a.probs, a.logits, a.summary};

    const int64_t ntiles = (a.rows + 15) / 16;
    const int64_t stride = (int64_t)gridDim.x * kOWaves;
    // this lane's quarter row of a tile; rows past the end are clamped (read, never stored)
    auto load_tile = [&](int64_t tile, float (&dst)[kOKS]) {
        const int64_t row = min(min(tile, ntiles - 1) * 16 + rr, a.rows - 1);
        const float2* src = reinterpret_cast<const float2*>(a.y + row * 200 + kq * kOKS);
#pragma unroll
        for (int q = 0; q < kOKS / 2; ++q) { const float2 v = src[q]; dst[2 * q] = v.x; dst[2 * q + 1] = v.y; }
    };
    float A[kOKS], An[kOKS];
    load_tile((int64_t)blockIdx.x * kOWaves + wave, A);
    for (int64_t tile = (int64_t)blockIdx.x * kOWaves + wave; tile < ntiles; tile += stride) {
        const int64_t r0 = tile * 16;
        load_tile(tile + stride, An);          // next tile's rows fly under this tile's MFMAs
        f32x4 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const float b = w2[ct * 16 + rr];                                   // bias row
            acc[ct] = (f32x4){b, b, b, b};
        }
#pragma unroll
        for (int kk = 0; kk < kOKS; ++kk) {
            const float* bp = w2 + (size_t)(1 + 4 * kk + kq) * nop + rr;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[kk], bp[ct * 16], acc[ct], 0, 0, 0);
        }
        softmax_tile<NCT>(acc, r0, sinks, lane);
#pragma unroll
        for (int q = 0; q < kOKS; ++q) A[q] = An[q];
    }
}

// ---------------------------------------------------------------------------------------------
// K4 (split operands): the same product on the 16-bit matrix cores, operands split as in the
// recurrence above -- W2 = W_hi (bf16) + W_r (fp16 of the rest), a row of hout = three bf16 terms +
// one fp16 -- four v_mfma_f32_16x16x32 per k-step of 32 and class tile, f32 accumulation, relative
// error of a product sum ~2^-19.  The f32-input kernel above needs 300 MFMAs of 32 cycles per
// 16-row tile; this one 168 of 16 cycles, and the split of the hout rows (once per row tile, reused
// by every class tile) is VALU work that runs beside them.  A wave takes two row tiles at a time (four with up to 16
// classes, one with more than 96) so that each B fragment it reads from LDS (W2 sits there as ready
// fragments, 14 KiB per class tile) feeds eight MFMAs, and twelve waves per CU (168 VGPRs) keep
// enough row loads in flight: with everything but the loads removed the kernel still takes 0.57 ms
// per 2.76 M rows (3.9 TB/s), 0.76 ms complete (the f32-input form: 1.5 ms).
constexpr int kO2KS = 7;           // k-steps of 32 over the 200 inputs (+ 24 zeros)
// 16-row tiles per wave and pass: as many as the register file takes beside the accumulators
#ifdef TA_O2T
constexpr int o2_tiles(int) { return TA_O2T; }
#else
constexpr int o2_tiles(int nct) { return nct <= 1 ? 4 : nct <= 6 ? 2 : 1; }
#endif
#ifndef TA_O2_ABL
#define TA_O2_ABL 0     // timing experiments only: 1 no MFMAs, 2 no softmax, 4 no operand split
#endif
#ifndef TA_O2W
#define TA_O2W 12
#endif
constexpr int kO2Waves = TA_O2W;

struct Out2Args {
    const float* y;        // [rows][200]
    int64_t rows;
    const uint4* w2s;      // [plane 2: hi, r][nct][k-step 7][lane 64]: 8 x 16 bit = W2[ct*16 + lane%16][1 + 32*ks + 4*(lane/16) + 16*(j/4) + j%4]
    const float* bias;     // [nct * 16]: W2[:, 0]
    int no, nct;
    float* probs; float* logits; float4* summary;
};

__device__ __forceinline__ void split8(const float4& v0, const float4& v1, bf16x8& hi, bf16x8& mid, bf16x8& lo, f16x8& h16) {
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)v[i];
        const float r1 = v[i] - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        hi[i] = h; mid[i] = m; lo[i] = (__bf16)r2; h16[i] = (_Float16)v[i];
    }
}

template <int NCT>
__global__ __launch_bounds__(kO2Waves * 64) void lstm_output_split_kernel(Out2Args a) {
    constexpr int kO2T = o2_tiles(NCT);
    extern __shared__ __attribute__((aligned(16))) uint4 wsm[];       // [2][NCT][7][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < 2 * NCT * kO2KS * 64; e += kO2Waves * 64) wsm[e] = a.w2s[e];
    __syncthreads();
    const int kb = lane >> 4, rr = lane & 15;
    const OutSinks sinks{a.rows, a.no, a.probs, a.logits, a.summary};
    float bias[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) bias[ct] = a.bias[ct * 16 + rr];

    const int64_t nsuper = (a.rows + 16 * kO2T - 1) / (16 * kO2T);
    for (int64_t sup = (int64_t)blockIdx.x * kO2Waves + wave; sup < nsuper; sup += (int64_t)gridDim.x * kO2Waves) {
        const int64_t r0 = sup * (16 * kO2T);
        // this lane's 8 inputs of its row per k-step: 32*ks + 4*kb + {0..3} and + 16 (the order of the
        // sum over k is free, and this one makes every load instruction read 64 contiguous bytes
        // per row; the B fragments are packed in the same order); rows past the end are clamped
        // (read, never stored); the last k-step holds inputs 192..199 only
        const float* rowp[kO2T];
#pragma unroll
        for (int t = 0; t < kO2T; ++t)
            rowp[t] = a.y + min(r0 + t * 16 + rr, a.rows - 1) * 200 + kb * 4;
        f32x4 acc[kO2T][NCT];
#pragma unroll
        for (int t = 0; t < kO2T; ++t)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[t][ct] = (f32x4){bias[ct], bias[ct], bias[ct], bias[ct]};
        float4 raw[kO2T][2], nxt[kO2T][2];
#pragma unroll
        for (int t = 0; t < kO2T; ++t) {
            raw[t][0] = *reinterpret_cast<const float4*>(rowp[t]);
            raw[t][1] = *reinterpret_cast<const float4*>(rowp[t] + 16);
        }
#pragma unroll 1
        for (int ks = 0; ks < kO2KS; ++ks) {
            if (ks + 1 < kO2KS) {
                const bool have1 = (ks + 1 < kO2KS - 1);            // the last k-step: inputs 192..199 only
                const bool have0 = have1 || kb < 2;
#pragma unroll
                for (int t = 0; t < kO2T; ++t) {
                    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                    nxt[t][0] = have0 ? *reinterpret_cast<const float4*>(rowp[t] + 32 * (ks + 1)) : z;
                    nxt[t][1] = have1 ? *reinterpret_cast<const float4*>(rowp[t] + 32 * (ks + 1) + 16) : z;
                }
            }
            bf16x8 ahi[kO2T], amid[kO2T], alo[kO2T];
            f16x8 a16[kO2T];
#pragma unroll
            for (int t = 0; t < kO2T; ++t) {
                if (TA_O2_ABL & 4) {
                    ahi[t] = __builtin_bit_cast(bf16x8, raw[t][0]); amid[t] = __builtin_bit_cast(bf16x8, raw[t][1]);
                    alo[t] = ahi[t]; a16[t] = __builtin_bit_cast(f16x8, raw[t][1]);
                } else split8(raw[t][0], raw[t][1], ahi[t], amid[t], alo[t], a16[t]);
            }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const bf16x8 bh = __builtin_bit_cast(bf16x8, wsm[((0 * NCT + ct) * kO2KS + ks) * 64 + lane]);
                const f16x8 br = __builtin_bit_cast(f16x8, wsm[((1 * NCT + ct) * kO2KS + ks) * 64 + lane]);
#pragma unroll
                for (int t = 0; t < kO2T; ++t) {
                    f32x4 v = acc[t][ct];
                    if (TA_O2_ABL & 1) {
                        v[0] += (float)alo[t][0] * (float)bh[0] + (float)a16[t][1] * (float)br[1] + (float)amid[t][2] + (float)ahi[t][3];
                        acc[t][ct] = v;
                        continue;
                    }
                    v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo[t], bh, v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[t], br, v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(amid[t], bh, v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi[t], bh, v, 0, 0, 0);
                    acc[t][ct] = v;
                }
            }
#pragma unroll
            for (int t = 0; t < kO2T; ++t) { raw[t][0] = nxt[t][0]; raw[t][1] = nxt[t][1]; }
        }
#pragma unroll
        for (int t = 0; t < kO2T; ++t) {
            if (TA_O2_ABL & 2) {
                float sacc = 0.f;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) sacc += acc[t][ct][0] + acc[t][ct][1] + acc[t][ct][2] + acc[t][ct][3];
                const int64_t row = r0 + t * 16 + (lane >> 4) * 4;
                if (rr == 0 && row < a.rows && a.summary) a.summary[row] = make_float4(sacc, 0.f, 0.f, 0.f);
            } else softmax_tile<NCT>(acc[t], r0 + t * 16, sinks, lane);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K5: translate_back(outputs, threshold) (Appendix B.5).  One wave per line.
struct DecArgs {
    const float* probs; const int64_t* row_off; const int32_t* T; int nlines, no;
    float threshold;
    int32_t* dec_t; int32_t* dec_c; int32_t* dec_n; const int64_t* dec_off;
};

__global__ __launch_bounds__(64) void decode_kernel(DecArgs a) {
    const int line = blockIdx.x, lane = threadIdx.x;
    const int T = a.T[line];
    const float* p = a.probs + a.row_off[line] * a.no;
    int32_t* out_t = a.dec_t + a.dec_off[line];
    int32_t* out_c = a.dec_c + a.dec_off[line];
    int n = 0;
    bool in_run = false;
    unsigned long long best = 0ull;
    int best_t = 0;
    for (int t = 0; t < T; ++t) {
        const float* row = p + (size_t)t * a.no;
        const float p0 = row[0];
        // per-lane best over classes lane, lane+64: larger p wins, then the smaller class
        unsigned long long key = 0ull;
        for (int cls = lane; cls < a.no; cls += 64) {
            const unsigned long long k =
                ((unsigned long long)__float_as_uint(row[cls]) << 32) | (unsigned)(0xFFFFFFFFu - cls);
            key = k > key ? k : key;
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned lo = __shfl_xor((unsigned)key, d, 64);
            const unsigned hi = __shfl_xor((unsigned)(key >> 32), d, 64);
            const unsigned long long other = ((unsigned long long)hi << 32) | lo;
            key = other > key ? other : key;
        }
        if (p0 < a.threshold) {
            if (!in_run) { in_run = true; best = 0ull; best_t = t; }
            if (key > best) { best = key; best_t = t; }       // strictly greater: earlier t wins ties
        } else if (in_run) {
            if (lane == 0) { out_t[n] = best_t; out_c[n] = (int)(0xFFFFFFFFu - (unsigned)best); }
            ++n;
            in_run = false;
        }
    }
    if (in_run) {
        if (lane == 0) { out_t[n] = best_t; out_c[n] = (int)(0xFFFFFFFFu - (unsigned)best); }
        ++n;
    }
    if (lane == 0) a.dec_n[line] = n;
}

// K5': the same decode from K4's per-timestep summaries (16 B per timestep instead of a row of
// probabilities).  One lane per line would do; a wave scans 64 timesteps at a time.
struct DecSumArgs {
    const float4* summary; const int64_t* row_off; const int32_t* T; int nlines;
    float threshold;
    int32_t* dec_t; int32_t* dec_c; int32_t* dec_n; const int64_t* dec_off;
};

__global__ __launch_bounds__(64) void decode_summary_kernel(DecSumArgs a) {
    const int line = blockIdx.x, lane = threadIdx.x;
    const int T = a.T[line];
    const float4* s = a.summary + a.row_off[line];
    int32_t* out_t = a.dec_t + a.dec_off[line];
    int32_t* out_c = a.dec_c + a.dec_off[line];
    int n = 0;
    bool in_run = false;
    unsigned long long best = 0ull;
    int best_t = 0;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        float4 v = make_float4(1.f, 0.f, 0.f, 0.f);          // beyond T: not in a run
        if (t < T) v = s[t];
        const unsigned long long key =
            ((unsigned long long)__float_as_uint(v.y) << 32) | (0xFFFFFFFFu - __float_as_uint(v.z));
        const unsigned long long below = __ballot(v.x < a.threshold);
        // wave-uniform scan of the chunk, run by run: only the entries inside runs are visited (entries beyond T
        // read as "not below", so a run that reaches T ends there)
        int q = 0;
        while (q < 64) {
            if (!in_run) {
                const unsigned long long rest = below >> q;
                if (rest == 0ull) break;
                q += (int)__builtin_ctzll(rest);               // first entry of the next run
                in_run = true; best = 0ull; best_t = t0 + q;
            }
            const unsigned long long inv = ~(below >> q);       // (bits shifted in from above read as "not below")
            const int stop = inv == 0ull ? 64 : min(q + (int)__builtin_ctzll(inv), 64);   // end of the run of ones at q
            for (; q < stop; ++q) {
                const unsigned klo = __builtin_amdgcn_readlane((unsigned)key, q);
                const unsigned khi = __builtin_amdgcn_readlane((unsigned)(key >> 32), q);
                const unsigned long long kq = ((unsigned long long)khi << 32) | klo;
                if (kq > best) { best = kq; best_t = t0 + q; }
            }
            if (q < 64) {                                       // the entry at q is not below: the run is over
                if (lane == 0) { out_t[n] = best_t; out_c[n] = (int)(0xFFFFFFFFu - (unsigned)best); }
                ++n;
                in_run = false;
            }
        }
    }
    if (in_run) {
        if (lane == 0) { out_t[n] = best_t; out_c[n] = (int)(0xFFFFFFFFu - (unsigned)best); }
        ++n;
    }
    if (lane == 0) a.dec_n[line] = n;
}

}  // namespace ta

using namespace ta;

extern "C" int32_t ta_lstm_packed_weight_floats(int32_t mode) {
    // mode 0: f32 fragments [2][7][4][38][64]; mode 1: 16-bit planes [2][7][2][4][5][64][8] (as 4-byte units);
    // mode 2: f32, one k per instruction [2][7][152][64]
    if (mode == 2) return 2 * kWaves * kKP * 64;
    return mode == 0 ? 2 * kWaves * 4 * kKS * 64 : 2 * kWaves * 2 * 4 * kKS2 * 64 * 4;
}

extern "C" int ta_lstm_forward(const float* x, const int64_t* row_off, const int32_t* T,
                               const int32_t* group_lines, int32_t ngroups,
                               const float* wp, const float* peep, float* hout, int32_t mode,
                               const float* h0, const float* c0, const int32_t* tstart,
                               void* stream) {
    if (ngroups < 0) return ta_fail(TA_EINVAL, "negative group count");
    if (ngroups == 0) return TA_OK;
    if (!x || !row_off || !T || !group_lines || !wp || !peep || !hout)
        return ta_fail(TA_EINVAL, "null pointer argument");
    if (mode != 0 && mode != 1 && mode != 2)
        return ta_fail(TA_EINVAL, "mode must be 0 (f32 MFMA), 1 (split 16-bit operands) or 2 (f32 MFMA, groups of four lines)");
    if ((h0 != nullptr) != (c0 != nullptr) || (h0 != nullptr) != (tstart != nullptr))
        return ta_fail(TA_EINVAL, "h0, c0 and tstart go together (all null, or all given)");
    LstmArgs a{x, row_off, T, group_lines, wp, peep, hout, h0, c0, tstart};
    if (mode == 1)
        hipLaunchKernelGGL(lstm_seq_split_kernel, dim3(2 * ngroups), dim3(kWaves * 64), 0,
                           reinterpret_cast<hipStream_t>(stream), a);
    else if (mode == 2)
        hipLaunchKernelGGL(lstm_seq4_kernel, dim3(2 * ngroups), dim3(kWaves * 64), 0,
                           reinterpret_cast<hipStream_t>(stream), a);
    else
        hipLaunchKernelGGL(lstm_seq_kernel, dim3(2 * ngroups), dim3(kSeqWaves * 64), 0,
                           reinterpret_cast<hipStream_t>(stream), a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ta_fail_hip(e, "lstm_seq_kernel launch");
    return TA_OK;
}

template <int NCT>
static hipError_t launch_output(const OutArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    if (lds > 64 * 1024) {          // above the default dynamic-LDS limit: raise it (once per device, ta_common.h)
        const hipError_t e = allow_full_lds(&lstm_output_kernel<NCT>);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(lstm_output_kernel<NCT>, grid, dim3(kOWaves * 64), lds, st, a);
    return hipSuccess;
}

extern "C" int ta_lstm_output(const float* y, int64_t rows, const float* w2p, int32_t no,
                              float* probs, float* logits, float* summary, void* stream) {
    if (rows < 0 || no <= 0 || no > 16 * kMaxCT) return ta_fail(TA_EINVAL, "bad rows / class count");
    if (rows == 0) return TA_OK;
    if (!y || !w2p || !(probs || summary)) return ta_fail(TA_EINVAL, "null pointer argument");
    const int nct = (no + 15) / 16;
    OutArgs a{y, rows, w2p, no, nct, probs, logits, reinterpret_cast<float4*>(summary)};
    const size_t lds = (size_t)201 * nct * 16 * sizeof(float);
    const int64_t ntiles = (rows + 15) / 16;
    const int64_t want = (ntiles + kOWaves - 1) / kOWaves;
    const dim3 grid((unsigned)(want < 1024 ? want : 1024));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipError_t pre = hipSuccess;
    switch (nct) {
        case 1: pre = launch_output<1>(a, grid, lds, st); break;
        case 2: pre = launch_output<2>(a, grid, lds, st); break;
        case 3: pre = launch_output<3>(a, grid, lds, st); break;
        case 4: pre = launch_output<4>(a, grid, lds, st); break;
        case 5: pre = launch_output<5>(a, grid, lds, st); break;
        case 6: pre = launch_output<6>(a, grid, lds, st); break;
        case 7: pre = launch_output<7>(a, grid, lds, st); break;
        default: pre = launch_output<8>(a, grid, lds, st); break;
    }
    if (pre != hipSuccess) return ta_fail_hip(pre, "hipFuncSetAttribute(lstm_output_kernel)");
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ta_fail_hip(e, "lstm_output_kernel launch");
    return TA_OK;
}

template <int NCT>
static hipError_t launch_output_split(const Out2Args& a, dim3 grid, size_t lds, hipStream_t st) {
    const hipError_t e = allow_full_lds(&lstm_output_split_kernel<NCT>);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(lstm_output_split_kernel<NCT>, grid, dim3(kO2Waves * 64), lds, st, a);
    return hipSuccess;
}

extern "C" int64_t ta_lstm_output_split_weight_bytes(int32_t no) {
    if (no <= 0 || no > 16 * kMaxCT) return 0;
    return (int64_t)2 * ((no + 15) / 16) * kO2KS * 64 * 16;
}

extern "C" int ta_lstm_output_split(const float* y, int64_t rows, const void* w2s, const float* bias, int32_t no,
                                    float* probs, float* logits, float* summary, void* stream) {
    if (rows < 0 || no <= 0 || no > 16 * kMaxCT) return ta_fail(TA_EINVAL, "bad rows / class count");
    if (rows == 0) return TA_OK;
    if (!y || !w2s || !bias || !(probs || summary)) return ta_fail(TA_EINVAL, "null pointer argument");
    const int nct = (no + 15) / 16;
    Out2Args a{y, rows, reinterpret_cast<const uint4*>(w2s), bias, no, nct, probs, logits,
               reinterpret_cast<float4*>(summary)};
    const size_t lds = (size_t)ta_lstm_output_split_weight_bytes(no);
    const int64_t nsuper = (rows + 16 * o2_tiles(nct) - 1) / (16 * o2_tiles(nct));
    const int64_t want = (nsuper + kO2Waves - 1) / kO2Waves;
    const dim3 grid((unsigned)(want < 256 ? want : 256));     // the fragments of W2 take most of a CU's LDS: one workgroup per CU
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipError_t pre = hipSuccess;
    switch (nct) {
        case 1: pre = launch_output_split<1>(a, grid, lds, st); break;
        case 2: pre = launch_output_split<2>(a, grid, lds, st); break;
        case 3: pre = launch_output_split<3>(a, grid, lds, st); break;
        case 4: pre = launch_output_split<4>(a, grid, lds, st); break;
        case 5: pre = launch_output_split<5>(a, grid, lds, st); break;
        case 6: pre = launch_output_split<6>(a, grid, lds, st); break;
        case 7: pre = launch_output_split<7>(a, grid, lds, st); break;
        default: pre = launch_output_split<8>(a, grid, lds, st); break;
    }
    if (pre != hipSuccess) return ta_fail_hip(pre, "hipFuncSetAttribute(lstm_output_split_kernel)");
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ta_fail_hip(e, "lstm_output_split_kernel launch");
    return TA_OK;
}

extern "C" int ta_decode_summary(const float* summary, const int64_t* row_off, const int32_t* T,
                                 int32_t nlines, float threshold,
                                 int32_t* dec_t, int32_t* dec_c, int32_t* dec_n, const int64_t* dec_off,
                                 void* stream) {
    if (nlines < 0) return ta_fail(TA_EINVAL, "bad line count");
    if (nlines == 0) return TA_OK;
    if (!summary || !row_off || !T || !dec_t || !dec_c || !dec_n || !dec_off)
        return ta_fail(TA_EINVAL, "null pointer argument");
    DecSumArgs a{reinterpret_cast<const float4*>(summary), row_off, T, nlines, threshold,
                 dec_t, dec_c, dec_n, dec_off};
    hipLaunchKernelGGL(decode_summary_kernel, dim3(nlines), dim3(64), 0,
                       reinterpret_cast<hipStream_t>(stream), a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ta_fail_hip(e, "decode_summary_kernel launch");
    return TA_OK;
}

extern "C" int ta_decode(const float* probs, const int64_t* row_off, const int32_t* T,
                         int32_t nlines, int32_t no, float threshold,
                         int32_t* dec_t, int32_t* dec_c, int32_t* dec_n, const int64_t* dec_off,
                         void* stream) {
    if (nlines < 0 || no <= 0) return ta_fail(TA_EINVAL, "bad line / class count");
    if (nlines == 0) return TA_OK;
    if (!probs || !row_off || !T || !dec_t || !dec_c || !dec_n || !dec_off)
        return ta_fail(TA_EINVAL, "null pointer argument");
    DecArgs a{probs, row_off, T, nlines, no, threshold, dec_t, dec_c, dec_n, dec_off};
    hipLaunchKernelGGL(decode_kernel, dim3(nlines), dim3(64), 0,
                       reinterpret_cast<hipStream_t>(stream), a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ta_fail_hip(e, "decode_kernel launch");
    return TA_OK;
}
