// ta_nw_general.hip -- float64 / substitution-table form of the NW aligner.
//
// Covers the scoring systems the integer wavefront kernel (ta_nw.hip) does not: a caller-
// supplied scoring function (reference textSeqCompare.py:27-29, tabulated on the host over
// the distinct tokens) and non-integral numbers.  It keeps the reference's arithmetic
// literally -- IEEE float64 scores, -1e100 boundary sentinels (textSeqCompare.py:55,60),
// left-to-right additions (textSeqCompare.py:75-85), first maximum wins -- so results are
// bit-identical to the reference for any inputs.  It is a correctness path, not a fast one:
// one workgroup per problem sweeps anti-diagonals with a barrier per diagonal
// (ta_nw_general_batch: many problems in one launch).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ta_common.h"

namespace ta {

struct GenArgs {
    const int32_t* t; const int32_t* o; int n, m;
    const double* params;        // match, mismatch, gox, goy, gex, gey
    const double* table; int tm; // optional [t id][o id] scores, row length tm
    double* sc;                  // 9 * (n + 1) doubles: 3 diagonals x {M, X, Y}
    uint8_t* ptr;                // (n + 1) * (m + 1) bytes, row-major, PM | PX<<2 | PY<<4
    uint8_t* ops; int32_t* ops_len;
};

__device__ __forceinline__ int first_max3(double a, double b, double c, double& mx) {
    mx = a;
    if (b > mx) mx = b;
    if (c > mx) mx = c;
    return a == mx ? 0 : (b == mx ? 1 : 2);
}

// boundary values, textSeqCompare.py:53-60 (G = module-global gap_extend = -1)
__device__ __forceinline__ void boundary(int i, int j, double& M, double& X, double& Y) {
    if (i == 0) { M = -1.0 * j; X = -1.0 * j; Y = -1e100; }
    else        { M = -1.0 * i; X = -1e100;   Y = -1.0 * i; }
}

__device__ __forceinline__ void general_one(const GenArgs& a) {
    const int n = a.n, m = a.m, tid = threadIdx.x;
    const double match = a.params[0], mismatch = a.params[1];
    const double gox = a.params[2], goy = a.params[3], gex = a.params[4], gey = a.params[5];
    const size_t S = (size_t)n + 1, W = (size_t)m + 1;
    for (int d = 2; d <= n + m; ++d) {
        double* cur = a.sc + (size_t)(d % 3) * 3 * S;
        const double* p1 = a.sc + (size_t)((d + 2) % 3) * 3 * S;   // diagonal d-1
        const double* p2 = a.sc + (size_t)((d + 1) % 3) * 3 * S;   // diagonal d-2
        const int lo = max(1, d - m), hi = min(n, d - 1);
        for (int i = lo + tid; i <= hi; i += blockDim.x) {
            const int j = d - i;
            double Md, Xd, Yd, Ml, Xl, Yl, Mu, Xu, Yu;
            if (i - 1 == 0 || j - 1 == 0) boundary(i - 1, j - 1, Md, Xd, Yd);
            else { Md = p2[i - 1]; Xd = p2[S + i - 1]; Yd = p2[2 * S + i - 1]; }
            if (j - 1 == 0) boundary(i, 0, Ml, Xl, Yl);
            else { Ml = p1[i]; Xl = p1[S + i]; Yl = p1[2 * S + i]; }
            if (i - 1 == 0) boundary(0, j, Mu, Xu, Yu);
            else { Mu = p1[i - 1]; Xu = p1[S + i - 1]; Yu = p1[2 * S + i - 1]; }
            const int ti = a.t[i - 1], oj = a.o[j - 1];
            const double s = a.table ? a.table[(size_t)ti * a.tm + oj]
                                     : (ti == oj ? match : mismatch);
            double mx;
            const int pm = first_max3(Md, Xd, Yd, mx);                       // :70-72
            cur[i] = mx + s;
            const int py = first_max3(Ml + goy + gey, Xl + goy + gey, Yl + gey, mx);   // :75-80
            cur[2 * S + i] = mx;
            const int px = first_max3(Mu + gox + gex, Xu + gex, Yu + gox + gex, mx);   // :83-88
            cur[S + i] = mx;
            a.ptr[(size_t)i * W + j] = (uint8_t)(pm | (px << 2) | (py << 4));
        }
        __syncthreads();
    }
    if (tid != 0) return;
    // traceback, textSeqCompare.py:96-170
    const int cap = n + m;
    int x = n, y = m, len = 0, st = 0;
    if (n > 0 && m > 0) st = a.ptr[(size_t)n * W + m] & 3;
    while (x > 0 && y > 0) {
        const unsigned b = a.ptr[(size_t)x * W + y];
        int op;
        if (st == 0) { op = 0; st = b & 3; --x; --y; }
        else if (st == 1) { op = 1; st = (b >> 2) & 3; --x; }
        else { op = 2; st = (b >> 4) & 3; --y; }
        a.ops[cap - 1 - len] = (uint8_t)op; ++len;
    }
    while (y > 0) { a.ops[cap - 1 - len] = 2; ++len; --y; }
    while (x > 0) { a.ops[cap - 1 - len] = 1; ++len; --x; }
    *a.ops_len = len;
}

__global__ __launch_bounds__(1024) void nw_general_kernel(GenArgs a) { general_one(a); }

// many problems, one workgroup each (a parameter grid with non-integral numbers: the reference's
// evaluate_text_alignment.py:178-198 loop, one launch instead of one per scoring system)
struct GenBatchArgs {
    const int32_t* t_codes; const int64_t* t_off;
    const int32_t* o_codes; const int64_t* o_off;
    const double* params; int params_stride;          // 0: one system for all, 6: one per problem
    double* sc; const int64_t* sc_off;                 // in doubles
    uint8_t* ptr; const int64_t* ptr_off;              // in bytes
    uint8_t* ops; const int64_t* ops_off; int32_t* ops_len;
};

__global__ __launch_bounds__(1024) void nw_general_batch_kernel(GenBatchArgs b) {
    const int p = blockIdx.x;
    GenArgs a;
    a.t = b.t_codes + b.t_off[p];
    a.o = b.o_codes + b.o_off[p];
    a.n = (int)(b.t_off[p + 1] - b.t_off[p]);
    a.m = (int)(b.o_off[p + 1] - b.o_off[p]);
    a.params = b.params + (size_t)p * b.params_stride;
    a.table = nullptr; a.tm = 0;
    a.sc = b.sc + b.sc_off[p];
    a.ptr = b.ptr + b.ptr_off[p];
    a.ops = b.ops + b.ops_off[p];
    a.ops_len = b.ops_len + p;
    general_one(a);
}

}  // namespace ta

extern "C" int64_t ta_nw_general_score_bytes(int32_t n) { return (int64_t)9 * ((int64_t)n + 1) * 8; }
extern "C" int64_t ta_nw_general_ptr_bytes(int32_t n, int32_t m) {
    return ((int64_t)n + 1) * ((int64_t)m + 1);
}

extern "C" int ta_nw_general(const int32_t* t, int32_t n, const int32_t* o, int32_t m,
                             const double* params, const double* table, int32_t tm,
                             double* score_ws, uint8_t* ptr_ws,
                             uint8_t* ops_out, int32_t* ops_len, void* stream) {
    if (n < 0 || m < 0) return ta_fail(TA_EINVAL, "negative size");
    if (!params || !score_ws || !ptr_ws || !ops_len || (!ops_out && n + m > 0))
        return ta_fail(TA_EINVAL, "null pointer argument");
    if ((n > 0 && !t) || (m > 0 && !o)) return ta_fail(TA_EINVAL, "null code pointer");
    ta::GenArgs a{t, o, n, m, params, table, tm, score_ws, ptr_ws, ops_out, ops_len};
    hipLaunchKernelGGL(ta::nw_general_kernel, dim3(1), dim3(1024), 0,
                       reinterpret_cast<hipStream_t>(stream), a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ta_fail_hip(e, "nw_general_kernel launch");
    return TA_OK;
}

extern "C" int ta_nw_general_batch(const int32_t* t_codes, const int64_t* t_off,
                                   const int32_t* o_codes, const int64_t* o_off, int32_t nprob,
                                   const double* params, int32_t params_stride,
                                   double* score_ws, const int64_t* score_off,
                                   uint8_t* ptr_ws, const int64_t* ptr_off,
                                   uint8_t* ops_out, const int64_t* ops_off, int32_t* ops_len, void* stream) {
    if (nprob < 0) return ta_fail(TA_EINVAL, "negative problem count");
    if (nprob == 0) return TA_OK;
    if (params_stride != 0 && params_stride != 6) return ta_fail(TA_EINVAL, "params_stride must be 0 or 6");
    if (!t_codes || !t_off || !o_codes || !o_off || !params || !score_ws || !score_off || !ptr_ws || !ptr_off ||
        !ops_out || !ops_off || !ops_len)
        return ta_fail(TA_EINVAL, "null pointer argument");
    ta::GenBatchArgs b{t_codes, t_off, o_codes, o_off, params, params_stride, score_ws, score_off,
                       ptr_ws, ptr_off, ops_out, ops_off, ops_len};
    hipLaunchKernelGGL(ta::nw_general_batch_kernel, dim3(nprob), dim3(1024), 0,
                       reinterpret_cast<hipStream_t>(stream), b);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ta_fail_hip(e, "nw_general_batch_kernel launch");
    return TA_OK;
}
