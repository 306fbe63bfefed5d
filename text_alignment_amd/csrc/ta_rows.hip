// ta_rows.hip -- prepared text-line rows from wherever they already lie on the device into the recogniser's row layout.
//
// The recogniser (ta_lstm.hip, ta_lstm_f64.hip) wants the (T, 48) float32 rows of a batch's lines in ONE tensor, longest
// line first (the order its groups of lines take them in).  A caller whose rows already sit in device memory -- a block
// of rows it keeps resident, or the device copy of a page-locked host block that one DMA transfer brought over as it
// was (text_alignment_amd.page.RowBlock) -- hands over one source address per line; this kernel is the permutation.
// It replaces, for such inputs, the host-side memcpy of every line into a staging buffer that the page pipeline of
// alignToOCR.process_batch spent most of its host time in (the seam being fed: reference alignToOCR.py:131-147, where
// the strips go to the recogniser as PNG files).
//
// HBM-bound by construction: 192 bytes read and written per row, 16 bytes per lane, rows of a line contiguous on both
// sides -- every wave moves 1 KiB pieces of both streams.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ta_common.h"

namespace {

constexpr int kRowFloats = 48;                 // ni of the line models (include/text_alignment_amd.h)
constexpr int kRowVec = kRowFloats / 4;        // float4 per row
constexpr int kRowsPerBlock = 64;              // 12 KiB per workgroup
constexpr int kThreads = 256;

struct GatherArgs {
    const int64_t* src;        // [nlines] device ADDRESS of line b's first row (16-byte aligned)
    const int64_t* dst_row;    // [nlines] first row of line b in x
    const int32_t* T;          // [nlines] rows of line b
    float* x;                  // [rows][48]
};

__global__ __launch_bounds__(kThreads) void rows_gather_kernel(GatherArgs a) {
    const int b = blockIdx.y;
    const int T = a.T[b];
    const int r0 = blockIdx.x * kRowsPerBlock;
    if (r0 >= T) return;
    const int nvec = min(kRowsPerBlock, T - r0) * kRowVec;
    const float4* s = reinterpret_cast<const float4*>(a.src[b]) + (size_t)r0 * kRowVec;
    float4* d = reinterpret_cast<float4*>(a.x) + ((size_t)a.dst_row[b] + r0) * kRowVec;
#pragma unroll
    for (int i = threadIdx.x; i < kRowsPerBlock * kRowVec; i += kThreads)
        if (i < nvec) d[i] = s[i];
}

}  // namespace

extern "C" int ta_rows_gather(const int64_t* src, const int64_t* dst_row, const int32_t* T, int32_t nlines,
                              int32_t max_T, float* x, void* stream) {
    if (nlines < 0 || max_T < 0) return ta_fail(TA_EINVAL, "negative count");
    if (nlines == 0 || max_T == 0) return TA_OK;
    if (!src || !dst_row || !T || !x) return ta_fail(TA_EINVAL, "null pointer argument");
    for (int32_t b0 = 0; b0 < nlines; b0 += 65535) {           // (grid.y is 16 bits wide)
        const int32_t nb = nlines - b0 < 65535 ? nlines - b0 : 65535;
        GatherArgs a{src + b0, dst_row + b0, T + b0, x};
        const dim3 grid((unsigned)((max_T + kRowsPerBlock - 1) / kRowsPerBlock), (unsigned)nb);
        hipLaunchKernelGGL(rows_gather_kernel, grid, dim3(kThreads), 0, reinterpret_cast<hipStream_t>(stream), a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return ta_fail_hip(e, "rows_gather_kernel launch");
    }
    return TA_OK;
}
