// ta_lineest.hip -- text-line height normalisation on the GPU (SURVEY.md section 8f, row N1,
// "GPU later"): what `ocropus-rpred` does to every PNG strip that reference alignToOCR.py:131-147
// hands it before the LSTM sees it -- ocropy 1.3.3 CenterNormalizer.measure / dewarp / normalize
// and prepare_line (third-party arithmetic, parity unpinned).
//
// The host restatement (text_alignment_amd/lineest.py, scipy.ndimage in float64) is followed
// operation by operation, in the same order of floating-point operations as scipy's C loops
// (correlate1d: centre tap, then tap pairs from the outside in; uniform_filter1d: running sum;
// integer output of gaussian_filter: truncation; geometric transform: 2 x 2 taps in row-major
// order), with explicit non-fused multiplies and adds, so that the integer decisions in the
// middle -- the per-column arg-max of the smoothed image, the truncated centre line, the band
// half-height r -- come out identical and the resampled line agrees to float32 rounding.
//
// Data: one greyscale uint8 strip per line (h x w, white background), concatenated; three float64
// planes of h x w per line as workspace.  Everything here is bandwidth / latency trivia next to
// the recogniser: the point is to take 16-50 ms of host CPU per strip off the page pipeline.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ta_common.h"

namespace ta {

constexpr int kLnThreads = 256;
#ifndef TA_LN_ROW_NO
#define TA_LN_ROW_NO 8      // adjacent outputs per thread of the row gaussian (4: 4.3 ms per 960 strips, 8: see DESIGN)
#endif
constexpr int kLnTarget = 48;      // target height
constexpr int kLnPad = 16;         // prepare_line pad

struct LnArgs {
    const uint8_t* pix; const int64_t* pix_off;      // [h][w] per line
    const int32_t* hh; const int32_t* ww;
    int32_t nlines;
    const double* gw;                                // gaussian weights, all lines' kernels
    const int64_t* gw_off;                           // [nlines][3] offset of the CENTRE tap of wy, wx, wc
    const int32_t* gr;                               // [nlines][3] radii ry, rx, rc
    double* ws; const int64_t* ws_off;               // 3 planes of h*w doubles per line
    int32_t* arg; int32_t* center; const int64_t* col_off;   // [w] per line
    int32_t* minmax;                                 // [nlines][2]
    int32_t* r_out; int32_t* wout;                   // [nlines]
};

__device__ __forceinline__ double dmul(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double dadd(double a, double b) { return __dadd_rn(a, b); }

// B.0 / prepare_raw_strip: min and max pixel of each strip
__global__ __launch_bounds__(kLnThreads) void ln_minmax_kernel(LnArgs a) {
    __shared__ int smin[kLnThreads], smax[kLnThreads];
    const int line = blockIdx.x, tid = threadIdx.x;
    const int64_t n = (int64_t)a.hh[line] * a.ww[line];
    const uint8_t* p = a.pix + a.pix_off[line];
    int lo = 255, hi = 0;
    for (int64_t e = tid; e < n; e += kLnThreads) { const int v = p[e]; lo = min(lo, v); hi = max(hi, v); }
    smin[tid] = lo; smax[tid] = hi;
    __syncthreads();
    for (int s = kLnThreads / 2; s > 0; s >>= 1) {
        if (tid < s) { smin[tid] = min(smin[tid], smin[tid + s]); smax[tid] = max(smax[tid], smax[tid + s]); }
        __syncthreads();
    }
    if (tid == 0) { a.minmax[2 * line] = smin[0]; a.minmax[2 * line + 1] = smax[0]; }
}

// temp = amax(line) - line; temp = temp / amax(temp), line = pix / 255.0   (plane 0)
__global__ __launch_bounds__(kLnThreads) void ln_temp_kernel(LnArgs a) {
    const int line = blockIdx.x;
    const int64_t n = (int64_t)a.hh[line] * a.ww[line];
    const uint8_t* p = a.pix + a.pix_off[line];
    double* A = a.ws + a.ws_off[line];
    const double amax = (double)a.minmax[2 * line + 1] / 255.0;
    const double tmax = dadd(amax, -((double)a.minmax[2 * line] / 255.0));
    for (int64_t e = (int64_t)blockIdx.y * kLnThreads + threadIdx.x; e < n; e += (int64_t)gridDim.y * kLnThreads)
        A[e] = dadd(amax, -((double)p[e] / 255.0)) / tmax;
}

// scipy correlate1d with a symmetric kernel, mode 'constant' (zeros): centre tap, then the pairs
// from the outermost inwards.  AXIS 0: along rows (stride w), AXIS 1: along columns (stride 1).
template <int AXIS, int SRC, int DST, int WSEL>
__global__ __launch_bounds__(kLnThreads) void ln_gauss_kernel(LnArgs a) {
    const int line = blockIdx.x;
    const int h = a.hh[line], w = a.ww[line];
    const int64_t n = (int64_t)h * w;
    const double* S = a.ws + a.ws_off[line] + (int64_t)SRC * n;
    double* D = a.ws + a.ws_off[line] + (int64_t)DST * n;
    const double* wc = a.gw + a.gw_off[3 * line + WSEL];
    const int rad = a.gr[3 * line + WSEL];
    const int len = AXIS == 0 ? h : w;
    const int64_t stride = AXIS == 0 ? w : 1;
    for (int64_t e = (int64_t)blockIdx.y * kLnThreads + threadIdx.x; e < n; e += (int64_t)gridDim.y * kLnThreads) {
        const int pos = AXIS == 0 ? (int)(e / w) : (int)(e % w);
        const double* c = S + e;
        double t = dmul(c[0], wc[0]);
        const int reach = min(rad, len - 1);               // beyond it both taps of a pair are zeros
        for (int jj = -reach; jj < 0; ++jj) {
            const double lo = (pos + jj >= 0) ? c[(int64_t)jj * stride] : 0.0;
            const double hi = (pos - jj < len) ? c[-(int64_t)jj * stride] : 0.0;
            t = dadd(t, dmul(dadd(lo, hi), wc[jj]));
        }
        D[e] = t;
    }
}

// The same correlation along a row (the expensive one: up to 8 h + 1 taps), NO adjacent outputs
// per thread: the two tap windows x[j + jj .. j + jj + 3] and x[j - jj .. j - jj + 3] slide by one
// element per tap pair, so each pair costs two new loads for NO outputs.  Every output still sums
// its own products in scipy's order.
template <int SRC, int DST, int WSEL, bool ONLY_WIDE>
__global__ __launch_bounds__(kLnThreads) void ln_gauss_row_kernel(LnArgs a) {
    constexpr int NO = TA_LN_ROW_NO;
    const int line = blockIdx.x;
    const int h = a.hh[line], w = a.ww[line];
    const int64_t n = (int64_t)h * w;
    const double* S = a.ws + a.ws_off[line] + (int64_t)SRC * n;
    double* D = a.ws + a.ws_off[line] + (int64_t)DST * n;
    const double* wc = a.gw + a.gw_off[3 * line + WSEL];
    const int rad = a.gr[3 * line + WSEL];
    const int reach = min(rad, w - 1);
    if (ONLY_WIDE && reach <= 640) return;                  // (kRowMaxReach: those strips are ln_gauss_row_lds_kernel's)
    const int per_row = (w + NO - 1) / NO;
    const int64_t ngroups = (int64_t)h * per_row;
    for (int64_t gidx = (int64_t)blockIdx.y * kLnThreads + threadIdx.x; gidx < ngroups;
         gidx += (int64_t)gridDim.y * kLnThreads) {
        const int i = (int)(gidx / per_row), j0 = (int)(gidx % per_row) * NO;
        const double* row = S + (int64_t)i * w;
        auto X = [&](int k) -> double { return (k >= 0 && k < w) ? row[k] : 0.0; };
        double t[NO], lo[NO], hi[NO];
#pragma unroll
        for (int q = 0; q < NO; ++q) {
            t[q] = dmul(X(j0 + q), wc[0]);
            lo[q] = X(j0 - reach + q);
            hi[q] = X(j0 + reach + q);
        }
        // the windows are RINGS: at the u-th tap of a round of NO, element q of the low window sits in
        // lo[(q + u) % NO] and of the high window in hi[(q - u) mod NO] -- a slide costs one load each and no
        // register moves (shifting 2 x NO doubles per tap pair cost as much as the arithmetic)
        for (int jb = -reach; jb < 0; jb += NO) {
#pragma unroll
            for (int u = 0; u < NO; ++u) {
                const int jj = jb + u;
                if (jj < 0) {
                    const double wj = wc[jj];
#pragma unroll
                    for (int q = 0; q < NO; ++q)
                        t[q] = dadd(t[q], dmul(dadd(lo[(q + u) % NO], hi[(q - u + NO) % NO]), wj));
                    lo[u % NO] = X(j0 + jj + NO);                  // enters as element NO - 1 of tap jj + 1's window
                    hi[(NO - 1 - u) % NO] = X(j0 - jj - 1);        // enters as element 0
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NO; ++q)
            if (j0 + q < w) D[(int64_t)i * w + j0 + q] = t[q];
    }
}

// Round 6: the same row correlation with the row staged in LDS.  The kernel above runs at a fifth of the float64 VALU
// rate: a thread's two new window elements per tap pair are 8-byte loads at a lane stride of 64 bytes -- 64 cache lines per
// load instruction -- and the texture path, not the arithmetic, sets the pace (6.6-7.0 ms per 960 strips of 60 x 800..2000
// at sigma = 60: 481 taps).  Here a workgroup copies a tile of a row (its outputs + `reach` elements either side, zeros
// outside the strip: no bounds test in the tap loop) into LDS with coalesced loads, and the windows slide over LDS.  An ODD
// number of adjacent outputs per thread: an odd lane stride in doubles (40 bytes at five) puts the 16 lanes of a quarter-wave
// on 16 different pairs of banks -- no conflicts, no padding.  Every output still sums its own products in scipy's order (centre tap, then the
// pairs from the outermost inwards, explicit non-fused operations): results equal to the bit, checked by the same tests.
// Strips whose reach exceeds the tile's halo (taller than 160 rows) take the kernel above.
// The outputs per thread follow the row's width -- 5 up to 1 280 columns, 7 up to 1 792, 9 beyond -- so that one pass of the
// workgroup covers the row (a second pass over a 120-column rest costs as much as the first); all three odd.
constexpr int kRowMaxNO = 9;
constexpr int kRowMaxReach = 640;                   // 4 sigma + 0.5 at sigma = h = 160
template <int NO>
__device__ __forceinline__ void gauss_row_lds_body(const double* S, double* D, const double* wc, int h, int w, int reach, double* L) {
    constexpr int kTile = kLnThreads * NO;
    const int tid = threadIdx.x;
    const int span = kTile + 2 * reach + NO;
    for (int i = blockIdx.y; i < h; i += gridDim.y) {
        const double* row = S + (int64_t)i * w;
        for (int jt = 0; jt < w; jt += kTile) {
            __syncthreads();                                // the tile before this one has been read
            for (int k = tid; k < span; k += kLnThreads) {
                const int src = jt - reach + k;
                L[k] = (src >= 0 && src < w) ? row[src] : 0.0;
            }
            __syncthreads();
            const int j0 = jt + NO * tid;
            if (j0 < w) {
                const double* C = L + reach + NO * tid;     // C[k] = X(j0 + k)
                double t[NO], lo[NO], hi[NO];
#pragma unroll
                for (int q = 0; q < NO; ++q) {
                    t[q] = dmul(C[q], wc[0]);
                    lo[q] = C[q - reach];
                    hi[q] = C[q + reach];
                }
                for (int jb = -reach; jb < 0; jb += NO) {   // ring windows, as in the kernel above
#pragma unroll
                    for (int u = 0; u < NO; ++u) {
                        const int jj = jb + u;
                        if (jj < 0) {
                            const double wj = wc[jj];
#pragma unroll
                            for (int q = 0; q < NO; ++q)
                                t[q] = dadd(t[q], dmul(dadd(lo[(q + u) % NO], hi[(q - u + NO) % NO]), wj));
                            lo[u % NO] = C[jj + NO];
                            hi[(NO - 1 - u) % NO] = C[-jj - 1];
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < NO; ++q)
                    if (j0 + q < w) D[(int64_t)i * w + j0 + q] = t[q];
            }
        }
    }
}

template <int SRC, int DST, int WSEL>
__global__ __launch_bounds__(kLnThreads) void ln_gauss_row_lds_kernel(LnArgs a) {
    __shared__ double L[kLnThreads * kRowMaxNO + 2 * kRowMaxReach + kRowMaxNO];       // 28.7 KB
    const int line = blockIdx.x;
    const int h = a.hh[line], w = a.ww[line];
    const int64_t n = (int64_t)h * w;
    const double* S = a.ws + a.ws_off[line] + (int64_t)SRC * n;
    double* D = a.ws + a.ws_off[line] + (int64_t)DST * n;
    const double* wc = a.gw + a.gw_off[3 * line + WSEL];
    const int rad = a.gr[3 * line + WSEL];
    const int reach = min(rad, w - 1);
    if (reach > kRowMaxReach) return;                       // (such strips are done by ln_gauss_row_kernel, launched beside this one)
    if (w <= kLnThreads * 5) gauss_row_lds_body<5>(S, D, wc, h, w, reach, L);
    else if (w <= kLnThreads * 7) gauss_row_lds_body<7>(S, D, wc, h, w, reach, L);
    else gauss_row_lds_body<9>(S, D, wc, h, w, reach, L);
}

// The correlation down the columns (AXIS 0 of ln_gauss_kernel; reach = h - 1: every row of the strip), four
// vertically adjacent outputs per thread with the same sliding tap windows -- a thread's loads per tap pair go
// from eight to two, and the threads of a wave sit on neighbouring columns, so every load is one coalesced row
// segment.  Every output sums its own products in scipy's order.
template <int SRC, int DST, int WSEL, bool ONLY_TALL>
__global__ __launch_bounds__(kLnThreads) void ln_gauss_col_kernel(LnArgs a) {
    constexpr int NO = 4;
    const int line = blockIdx.x;
    const int h = a.hh[line], w = a.ww[line];
    if (ONLY_TALL && h <= 96) return;                       // (kColMaxH: those strips are ln_gauss_col_lds_kernel's)
    const int64_t n = (int64_t)h * w;
    const double* S = a.ws + a.ws_off[line] + (int64_t)SRC * n;
    double* D = a.ws + a.ws_off[line] + (int64_t)DST * n;
    const double* wc = a.gw + a.gw_off[3 * line + WSEL];
    const int rad = a.gr[3 * line + WSEL];
    const int reach = min(rad, h - 1);
    const int64_t ngroups = (int64_t)((h + NO - 1) / NO) * w;
    for (int64_t gidx = (int64_t)blockIdx.y * kLnThreads + threadIdx.x; gidx < ngroups;
         gidx += (int64_t)gridDim.y * kLnThreads) {
        const int j = (int)(gidx % w), i0 = (int)(gidx / w) * NO;
        const double* col = S + j;
        auto X = [&](int k) -> double { return (k >= 0 && k < h) ? col[(int64_t)k * w] : 0.0; };
        double t[NO], lo[NO], hi[NO];
#pragma unroll
        for (int q = 0; q < NO; ++q) {
            t[q] = dmul(X(i0 + q), wc[0]);
            lo[q] = X(i0 - reach + q);
            hi[q] = X(i0 + reach + q);
        }
        for (int jb = -reach; jb < 0; jb += NO) {              // ring windows, as in the row kernel above
#pragma unroll
            for (int u = 0; u < NO; ++u) {
                const int jj = jb + u;
                if (jj < 0) {
                    const double wj = wc[jj];
#pragma unroll
                    for (int q = 0; q < NO; ++q)
                        t[q] = dadd(t[q], dmul(dadd(lo[(q + u) % NO], hi[(q - u + NO) % NO]), wj));
                    lo[u % NO] = X(i0 + jj + NO);
                    hi[(NO - 1 - u) % NO] = X(i0 - jj - 1);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NO; ++q)
            if (i0 + q < h) D[(int64_t)(i0 + q) * w + j] = t[q];
    }
}

// Round 6: the column correlation over an LDS tile -- all h rows of 64 neighbouring columns (h <= kColMaxH; taller strips
// take the kernel above).  A workgroup's four waves take the groups of four rows in turn; a lane owns one column, so every
// LDS access of a wave is 64 consecutive doubles (no bank conflicts), and rows outside the strip are ONE row of zeros
// the index is clamped to.  Same operations per output, in the same order: results equal to the bit.
constexpr int kColTile = 64;
constexpr int kColMaxH = 96;                        // (96 + 1 zero row) x 64 x 8 B = 49 664 B of LDS: three workgroups per CU
template <int SRC, int DST, int WSEL>
__global__ __launch_bounds__(kLnThreads) void ln_gauss_col_lds_kernel(LnArgs a) {
    constexpr int NO = 4;
    __shared__ double L[(kColMaxH + 1) * kColTile];
    const int line = blockIdx.x, tid = threadIdx.x;
    const int h = a.hh[line], w = a.ww[line];
    if (h > kColMaxH) return;                               // (done by ln_gauss_col_kernel, launched beside this one)
    const int64_t n = (int64_t)h * w;
    const double* S = a.ws + a.ws_off[line] + (int64_t)SRC * n;
    double* D = a.ws + a.ws_off[line] + (int64_t)DST * n;
    const double* wc = a.gw + a.gw_off[3 * line + WSEL];
    const int rad = a.gr[3 * line + WSEL];
    const int reach = min(rad, h - 1);
    const int c = tid & (kColTile - 1), wave = tid >> 6;
    for (int jt = blockIdx.y * kColTile; jt < w; jt += gridDim.y * kColTile) {
        __syncthreads();                                    // the tile before this one has been read
        for (int e = tid; e < (h + 1) * kColTile; e += kLnThreads) {
            const int r = e / kColTile, cc = e % kColTile;
            L[e] = (r < h && jt + cc < w) ? S[(int64_t)r * w + jt + cc] : 0.0;      // row h: the zeros outside the strip
        }
        __syncthreads();
        if (jt + c >= w) continue;
        const double* col = L + c;
        auto X = [&](int k) -> double { return col[((unsigned)k < (unsigned)h ? k : h) * kColTile]; };
        for (int i0 = NO * wave; i0 < h; i0 += NO * (kLnThreads / 64)) {
            double t[NO], lo[NO], hi[NO];
#pragma unroll
            for (int q = 0; q < NO; ++q) {
                t[q] = dmul(X(i0 + q), wc[0]);
                lo[q] = X(i0 - reach + q);
                hi[q] = X(i0 + reach + q);
            }
            for (int jb = -reach; jb < 0; jb += NO) {       // ring windows, as in the kernels above
#pragma unroll
                for (int u = 0; u < NO; ++u) {
                    const int jj = jb + u;
                    if (jj < 0) {
                        const double wj = wc[jj];
#pragma unroll
                        for (int q = 0; q < NO; ++q)
                            t[q] = dadd(t[q], dmul(dadd(lo[(q + u) % NO], hi[(q - u + NO) % NO]), wj));
                        lo[u % NO] = X(i0 + jj + NO);
                        hi[(NO - 1 - u) % NO] = X(i0 - jj - 1);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < NO; ++q)
                if (i0 + q < h) D[(int64_t)(i0 + q) * w + jt + c] = t[q];
        }
    }
}

// scipy uniform_filter1d, mode 'constant': running sum over the zero-extended line.
// AXIS 0: one thread per column, size int(0.5 h); AXIS 1: one thread per row, size w.
template <int AXIS, int SRC, int DST>
__global__ __launch_bounds__(kLnThreads) void ln_uniform_kernel(LnArgs a) {
    const int line = blockIdx.x;
    const int h = a.hh[line], w = a.ww[line];
    const int64_t n = (int64_t)h * w;
    const double* S = a.ws + a.ws_off[line] + (int64_t)SRC * n;
    double* D = a.ws + a.ws_off[line] + (int64_t)DST * n;
    const int len = AXIS == 0 ? h : w;
    const int lines = AXIS == 0 ? w : h;
    const int64_t stride = AXIS == 0 ? w : 1, lstride = AXIS == 0 ? 1 : w;
    const int size = AXIS == 0 ? (int)(h * 0.5) : w;
    for (int q = blockIdx.y * kLnThreads + threadIdx.x; q < lines; q += gridDim.y * kLnThreads) {
        const double* s = S + (int64_t)q * lstride;
        double* d = D + (int64_t)q * lstride;
        if (size <= 1) {                                    // scipy skips axes of size <= 1
            for (int k = 0; k < len; ++k) d[(int64_t)k * stride] = s[(int64_t)k * stride];
            continue;
        }
        const int size1 = size / 2;
        auto ext = [&](int k) -> double {                   // extended line: size1 zeros in front
            const int src = k - size1;
            return (src >= 0 && src < len) ? s[(int64_t)src * stride] : 0.0;
        };
        // the running sum is one dependent chain per line (scipy's order); its INPUTS are not: eight steps'
        // worth are loaded first, so the chain waits for one memory latency per eight steps, not per step
        constexpr int kU = 8;
        double tmp = 0.0;
        for (int l0 = 0; l0 < size; l0 += kU) {
            double in[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) in[u] = (l0 + u < size) ? ext(l0 + u) : 0.0;
#pragma unroll
            for (int u = 0; u < kU; ++u) if (l0 + u < size) tmp = dadd(tmp, in[u]);
        }
        d[0] = tmp / (double)size;
        for (int l0 = 1; l0 < len; l0 += kU) {
            double in[kU], out[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                in[u] = (l0 + u < len) ? ext(l0 + u + size - 1) : 0.0;
                out[u] = (l0 + u < len) ? ext(l0 + u - 1) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                if (l0 + u < len) {
                    tmp = dadd(tmp, dadd(in[u], -out[u]));
                    d[(int64_t)(l0 + u) * stride] = tmp / (double)size;
                }
            }
        }
    }
}

// smoothed = G2 + 0.001 * U; a = argmax over rows (first maximum), per column
template <int G2, int U>
__global__ __launch_bounds__(kLnThreads) void ln_argmax_kernel(LnArgs a) {
    const int line = blockIdx.x;
    const int h = a.hh[line], w = a.ww[line];
    const int64_t n = (int64_t)h * w;
    const double* G = a.ws + a.ws_off[line] + (int64_t)G2 * n;
    const double* Uf = a.ws + a.ws_off[line] + (int64_t)U * n;
    int32_t* arg = a.arg + a.col_off[line];
    for (int j = blockIdx.y * kLnThreads + threadIdx.x; j < w; j += gridDim.y * kLnThreads) {
        double best = dadd(G[j], dmul(0.001, Uf[j]));
        int bi = 0;
        for (int i = 1; i < h; ++i) {
            const double v = dadd(G[(int64_t)i * w + j], dmul(0.001, Uf[(int64_t)i * w + j]));
            if (v > best) { best = v; bi = i; }
        }
        arg[j] = bi;
    }
}

// center = int32(gaussian_filter(a, 0.3 h)): integer input, default mode 'reflect', the float64
// result is truncated towards zero on the way into the integer output array
__global__ __launch_bounds__(kLnThreads) void ln_center_kernel(LnArgs a) {
    const int line = blockIdx.x;
    const int w = a.ww[line];
    const int32_t* arg = a.arg + a.col_off[line];
    int32_t* cen = a.center + a.col_off[line];
    const double* wc = a.gw + a.gw_off[3 * line + 2];
    const int rad = a.gr[3 * line + 2];
    auto at = [&](int k) -> double {                        // reflect: (d c b a | a b c d | d c b a)
        while (k < 0 || k >= w) k = (k < 0) ? (-k - 1) : (2 * w - k - 1);
        return (double)arg[k];
    };
    for (int j = blockIdx.y * kLnThreads + threadIdx.x; j < w; j += gridDim.y * kLnThreads) {
        double t = dmul((double)arg[j], wc[0]);
        for (int jj = -rad; jj < 0; ++jj) t = dadd(t, dmul(dadd(at(j + jj), at(j - jj)), wc[jj]));
        cen[j] = (int32_t)(long long)t;
    }
}

// mad = mean |row - center[col]| over ink pixels (temp != 0 <=> pixel below the strip's maximum)
__global__ __launch_bounds__(kLnThreads) void ln_mad_kernel(LnArgs a) {
    __shared__ unsigned long long ssum[kLnThreads], scnt[kLnThreads];
    const int line = blockIdx.x, tid = threadIdx.x;
    const int h = a.hh[line], w = a.ww[line];
    const int64_t n = (int64_t)h * w;
    const uint8_t* p = a.pix + a.pix_off[line];
    const int32_t* cen = a.center + a.col_off[line];
    const int pmax = a.minmax[2 * line + 1];
    unsigned long long sum = 0, cnt = 0;
    for (int64_t e = tid; e < n; e += kLnThreads) {
        if (p[e] != pmax) {
            const int i = (int)(e / w), j = (int)(e % w);
            sum += (unsigned long long)abs(i - cen[j]);
            ++cnt;
        }
    }
    ssum[tid] = sum; scnt[tid] = cnt;
    __syncthreads();
    for (int s = kLnThreads / 2; s > 0; s >>= 1) {
        if (tid < s) { ssum[tid] += ssum[tid + s]; scnt[tid] += scnt[tid + s]; }
        __syncthreads();
    }
    if (tid == 0) {
        const double mad = (double)ssum[0] / (double)scnt[0];
        const int r = (int)dadd(1.0, dmul(4.0, mad));
        a.r_out[line] = r;
        const double scale = dmul((double)kLnTarget, 1.0) / (double)(2 * r);
        a.wout[line] = (int)dmul(scale, (double)w);
    }
}

struct LnOutArgs {
    const uint8_t* pix; const int64_t* pix_off;
    const int32_t* hh; const int32_t* ww;
    int32_t nlines;
    const int32_t* center; const int64_t* col_off;
    const int32_t* minmax; const int32_t* r; const int32_t* wout;
    float* tmp; const int64_t* tmp_off;              // [48][W'] per line
    unsigned int* omax;                              // [nlines] bits of the (positive) maximum
    float* x; const int64_t* row_off;                // LSTM input rows, [W' + 32][48] per line
};

// dewarp (2r rows around the centre line, white outside) + affine_transform(eye/scale, order 1,
// mode 'constant', cval = white) to height 48
__global__ __launch_bounds__(kLnThreads) void ln_resample_kernel(LnOutArgs a) {
    __shared__ float smax[kLnThreads];
    const int line = blockIdx.x, tid = threadIdx.x;
    const int h = a.hh[line], w = a.ww[line], r = a.r[line], wo = a.wout[line];
    const uint8_t* p = a.pix + a.pix_off[line];
    const int32_t* cen = a.center + a.col_off[line];
    float* out = a.tmp + a.tmp_off[line];
    const float cval = (float)((double)a.minmax[2 * line + 1] / 255.0);
    const int dh = 2 * r;
    const double scale = dmul((double)kLnTarget, 1.0) / (double)dh;
    const double mdiag = 1.0 / scale;
    auto dewarped = [&](int y, int xx) -> double {          // float32 array in the reference
        const int row = cen[xx] + y - r;
        const float v = (row >= 0 && row < h) ? (float)((double)p[(int64_t)row * w + xx] / 255.0) : cval;
        return (double)v;
    };
    float vmax = 0.0f;
    const int64_t nout = (int64_t)kLnTarget * wo;
    for (int64_t e = (int64_t)blockIdx.y * kLnThreads + tid; e < nout; e += (int64_t)gridDim.y * kLnThreads) {
        const int i = (int)(e / wo), j = (int)(e % wo);
        const double cy = dadd(dadd(0.0, dmul((double)i, mdiag)), dmul((double)j, 0.0));
        const double cx = dadd(dadd(0.0, dmul((double)i, 0.0)), dmul((double)j, mdiag));
        float v;
        if (cy < 0.0 || cy > (double)(dh - 1) || cx < 0.0 || cx > (double)(w - 1)) {
            v = cval;
        } else {
            const int y0 = (int)floor(cy), x0 = (int)floor(cx);
            const double ty = dadd(cy, -(double)y0), tx = dadd(cx, -(double)x0);
            const double wy[2] = {dadd(1.0, -ty), ty}, wx[2] = {dadd(1.0, -tx), tx};
            double t = 0.0;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int yy = min(y0 + dy, dh - 1), xq = min(x0 + dx, w - 1);
                    t = dadd(t, dmul(dmul(dewarped(yy, xq), wy[dy]), wx[dx]));
                }
            v = (float)t;
        }
        out[e] = v;
        vmax = fmaxf(vmax, v);
    }
    smax[tid] = vmax;
    __syncthreads();
    for (int s = kLnThreads / 2; s > 0; s >>= 1) {
        if (tid < s) smax[tid] = fmaxf(smax[tid], smax[tid + s]);
        __syncthreads();
    }
    if (tid == 0) atomicMax(a.omax + line, __float_as_uint(smax[0]));   // values are >= 0
}

// prepare_line: line / amax(line); amax(line) - line; transpose; 16 zero rows before and after
__global__ __launch_bounds__(kLnThreads) void ln_finish_kernel(LnOutArgs a) {
    const int line = blockIdx.x;
    const int wo = a.wout[line];
    const float* in = a.tmp + a.tmp_off[line];
    float* x = a.x + a.row_off[line] * kLnTarget;
    const float amax = __uint_as_float(a.omax[line]);
    const float top = __fdiv_rn(amax, amax);                 // amax of the divided line
    const int64_t rows = wo + 2 * kLnPad;
    for (int64_t e = (int64_t)blockIdx.y * kLnThreads + threadIdx.x; e < rows * kLnTarget;
         e += (int64_t)gridDim.y * kLnThreads) {
        const int t = (int)(e / kLnTarget), i = (int)(e % kLnTarget);
        float v = 0.0f;
        if (t >= kLnPad && t < kLnPad + wo) v = __fsub_rn(top, __fdiv_rn(in[(int64_t)i * wo + (t - kLnPad)], amax));
        x[e] = v;
    }
}

}  // namespace ta

using namespace ta;

extern "C" int ta_linenorm_measure(const uint8_t* pix, const int64_t* pix_off, const int32_t* hh,
                                   const int32_t* ww, int32_t nlines, const double* gw,
                                   const int64_t* gw_off, const int32_t* gr, double* ws,
                                   const int64_t* ws_off, int32_t* arg, int32_t* center,
                                   const int64_t* col_off, int32_t* minmax, int32_t* r_out,
                                   int32_t* wout, void* stream) {
    if (nlines < 0) return ta_fail(TA_EINVAL, "negative line count");
    if (nlines == 0) return TA_OK;
    if (!pix || !pix_off || !hh || !ww || !gw || !gw_off || !gr || !ws || !ws_off || !arg || !center ||
        !col_off || !minmax || !r_out || !wout)
        return ta_fail(TA_EINVAL, "null pointer argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    LnArgs a{pix, pix_off, hh, ww, nlines, gw, gw_off, gr, ws, ws_off, arg, center, col_off, minmax,
             r_out, wout};
    const dim3 one(nlines), wide(nlines, 32), cols(nlines, 8);
    hipLaunchKernelGGL(ln_minmax_kernel, one, dim3(kLnThreads), 0, st, a);
    hipLaunchKernelGGL(ln_temp_kernel, wide, dim3(kLnThreads), 0, st, a);
    hipLaunchKernelGGL((ln_gauss_col_lds_kernel<0, 1, 0>), wide, dim3(kLnThreads), 0, st, a); // plane 0 -> 1
    hipLaunchKernelGGL((ln_gauss_col_kernel<0, 1, 0, true>), dim3(nlines, 8), dim3(kLnThreads), 0, st, a);   // ... strips taller than 96 rows
    hipLaunchKernelGGL((ln_gauss_row_lds_kernel<1, 2, 1>), wide, dim3(kLnThreads), 0, st, a);     // plane 1 -> 2
    hipLaunchKernelGGL((ln_gauss_row_kernel<1, 2, 1, true>), dim3(nlines, 8), dim3(kLnThreads), 0, st, a);   // ... strips taller than 160 rows
    hipLaunchKernelGGL((ln_uniform_kernel<0, 2, 0>), cols, dim3(kLnThreads), 0, st, a);       // plane 2 -> 0
    hipLaunchKernelGGL((ln_uniform_kernel<1, 0, 1>), one, dim3(kLnThreads), 0, st, a);        // plane 0 -> 1
    hipLaunchKernelGGL((ln_argmax_kernel<2, 1>), cols, dim3(kLnThreads), 0, st, a);
    hipLaunchKernelGGL(ln_center_kernel, cols, dim3(kLnThreads), 0, st, a);
    hipLaunchKernelGGL(ln_mad_kernel, one, dim3(kLnThreads), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ta_fail_hip(e, "line normaliser (measure) launch");
    return TA_OK;
}

extern "C" int ta_linenorm_resample(const uint8_t* pix, const int64_t* pix_off, const int32_t* hh,
                                    const int32_t* ww, int32_t nlines, const int32_t* center,
                                    const int64_t* col_off, const int32_t* minmax, const int32_t* r,
                                    const int32_t* wout, float* tmp, const int64_t* tmp_off,
                                    uint32_t* omax, float* x, const int64_t* row_off, void* stream) {
    if (nlines < 0) return ta_fail(TA_EINVAL, "negative line count");
    if (nlines == 0) return TA_OK;
    if (!pix || !pix_off || !hh || !ww || !center || !col_off || !minmax || !r || !wout || !tmp ||
        !tmp_off || !omax || !x || !row_off)
        return ta_fail(TA_EINVAL, "null pointer argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    LnOutArgs a{pix, pix_off, hh, ww, nlines, center, col_off, minmax, r, wout, tmp, tmp_off, omax, x, row_off};
    hipError_t e = hipMemsetAsync(omax, 0, sizeof(uint32_t) * (size_t)nlines, st);
    if (e != hipSuccess) return ta_fail_hip(e, "line normaliser memset");
    hipLaunchKernelGGL(ln_resample_kernel, dim3(nlines, 16), dim3(kLnThreads), 0, st, a);
    hipLaunchKernelGGL(ln_finish_kernel, dim3(nlines, 16), dim3(kLnThreads), 0, st, a);
    e = hipGetLastError();
    if (e != hipSuccess) return ta_fail_hip(e, "line normaliser (resample) launch");
    return TA_OK;
}
