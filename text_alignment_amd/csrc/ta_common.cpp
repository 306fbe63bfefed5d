// ta_common.cpp -- ta_version / ta_last_error and the per-thread error record.
#include "ta_common.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static thread_local char g_err[256] = "";

int ta_fail(int code, const char* what) {
    std::snprintf(g_err, sizeof(g_err), "%s", what);
    return code;
}

int ta_fail_hip(hipError_t e, const char* where) {
    std::snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    return TA_EHIP;
}

extern "C" int ta_version(void) { return 100; }   // 0.1.0

extern "C" const char* ta_last_error(void) { return g_err; }

// PCI address ("0000:c1:00.0") of HIP device `device`: the key under /sys/bus/pci/devices/ from which a rank reads the
// NUMA node of its GPU (text_alignment_amd.sharding.bind_to_gpu_node)
extern "C" int ta_device_pci_bus_id(int32_t device, char* out, int32_t len) {
    if (!out || len < 13) return ta_fail(TA_EINVAL, "ta_device_pci_bus_id needs a buffer of at least 13 bytes");
    const hipError_t e = hipDeviceGetPCIBusId(out, len, device);
    if (e != hipSuccess) return ta_fail_hip(e, "hipDeviceGetPCIBusId");
    return TA_OK;
}

// Many small host arrays into one (page-locked) staging buffer in ONE native call: piece k = nbytes[k] bytes from src[k]
// to dst + dst_off[k].  The page pipeline's copy threads stage a chunk's strips / rows with it (275 KB per text line, 480
// lines per chunk): a Python loop of slice assignments takes and drops the interpreter lock once per piece and fights the
// pipeline's own thread for it; one foreign call holds no lock at all.  [host only: no device work]
extern "C" int ta_host_copy_pieces(void* dst, const void* const* src, const int64_t* dst_off, const int64_t* nbytes, int32_t n) {
    if (n < 0) return ta_fail(TA_EINVAL, "negative piece count");
    if (n == 0) return TA_OK;
    if (!dst || !src || !dst_off || !nbytes) return ta_fail(TA_EINVAL, "null pointer argument");
    char* base = static_cast<char*>(dst);
    for (int32_t k = 0; k < n; ++k)
        if (nbytes[k] > 0) std::memcpy(base + dst_off[k], src[k], (size_t)nbytes[k]);
    return TA_OK;
}

// Characters and boxes of every decoded line of a batch (reference alignToOCR.py:160-182, after the `.llocs` text the
// reference parses there: x is carried with ONE decimal, "%.1f", and the box edge is int(np.round(float(x) + offset_x)),
// half to even).  Host arithmetic, not a kernel: the page pipeline's thread spent 1.6 ms per chunk of 480 lines on the
// fifteen numpy passes this loop replaces.  Entry i of line b: (dec_t, dec_c)[dec_off[b] + i], i < dec_n[b];
// x = (t - pad) * raw_w[b] / (T[b] - 2 pad); a character's box runs from the previous character's position (the strip's
// x_min for the first) to its own; classes whose code point is < 0 ('~' and '') are dropped but still move the edge.
// Outputs (capacity sum dec_n): out_line, out_cp, out_boxes [k][4] = ulx, uly, lrx, lry; *out_count = characters kept.
static inline int64_t ta_edge_position(double x, double x_min) {
    const double t = x * 10.0, f = std::floor(t), frac = t - f;
    double one_dec = (frac > 0.5 ? f + 1.0 : f) / 10.0;
    if (std::fabs(frac - 0.5) < 1e-6) {                     // (near-)ties: the decimal conversion decides, as "%.1f" does
        char buf[64];
        std::snprintf(buf, sizeof(buf), "%.1f", x);
        one_dec = std::strtod(buf, nullptr);
    }
    return (int64_t)std::nearbyint(one_dec + x_min);        // round half to even (the default rounding mode)
}

extern "C" int ta_host_chars_of_batch(const int32_t* dec_t, const int32_t* dec_c, const int64_t* dec_n, const int64_t* dec_off,
                                      const int64_t* T, const int64_t* raw_w, const int64_t* x_min, const int64_t* y_min,
                                      const int64_t* y_max, const int64_t* cps, int32_t ncps, int32_t pad, int32_t nlines,
                                      int64_t dec_len, int64_t* out_line, int64_t* out_cp, int64_t* out_boxes, int64_t* out_count) {
    if (nlines < 0 || ncps < 0) return ta_fail(TA_EINVAL, "negative count");
    if (!out_count) return ta_fail(TA_EINVAL, "null pointer argument");
    *out_count = 0;
    if (nlines == 0) return TA_OK;
    if (!dec_t || !dec_c || !dec_n || !dec_off || !T || !raw_w || !x_min || !y_min || !y_max || !cps || !out_line || !out_cp || !out_boxes)
        return ta_fail(TA_EINVAL, "null pointer argument");
    int64_t k = 0;
    for (int32_t b = 0; b < nlines; ++b) {
        // (the counts come from the device: a line that claims more entries than the arrays hold is refused, not read)
        if (dec_n[b] < 0 || dec_off[b] < 0 || dec_off[b] + dec_n[b] > dec_len)
            return ta_fail(TA_EINVAL, "a line's decoded entries lie outside the decoder's arrays");
        const double scale = (double)raw_w[b] / (double)(T[b] - 2 * pad);
        const double xm = (double)x_min[b];
        int64_t left = x_min[b];
        const int32_t* tt = dec_t + dec_off[b];
        const int32_t* cc = dec_c + dec_off[b];
        for (int64_t i = 0; i < dec_n[b]; ++i) {
            const int64_t right = ta_edge_position(((double)tt[i] - (double)pad) * scale, xm);
            const int32_t c = cc[i];
            if (c < 0 || c >= ncps) return ta_fail(TA_EINVAL, "a decoded class is outside the codec");
            if (cps[c] >= 0) {
                out_line[k] = b;
                out_cp[k] = cps[c];
                out_boxes[4 * k] = left; out_boxes[4 * k + 1] = y_min[b]; out_boxes[4 * k + 2] = right; out_boxes[4 * k + 3] = y_max[b];
                ++k;
            }
            left = right;
        }
    }
    *out_count = k;
    return TA_OK;
}

// Union of the OCR character boxes under every syllable of a batch of pages (reference alignToOCR.py:285-324, after the
// alignment): host arithmetic.  ops: the alignment columns of all pages laid end to end (0 pair, 1 transcript character
// over a gap, 2 gap over an OCR character); idx: for every OCR-carrying column, in order, the row of `boxes` ([.][4] =
// ulx, uly, lrx, lry) of its character; syllable s covers the transcript characters first_t[s] .. last_t[s] (positions in
// the concatenated transcripts; ranges disjoint and ascending), i.e. the columns from the one of its first character to
// the one of its last.  Per syllable: out_low[s] = the largest uly under it (INT64_MIN if no OCR character is: the
// reference skips such a syllable, :313-314) and out_box[s] = union of the boxes whose uly IS that value (a syllable
// that spans two text lines keeps the lower one, :318-320).
extern "C" int ta_host_syllable_boxes(const uint8_t* ops, int64_t ncol, const int64_t* idx, int64_t nidx, const int64_t* boxes,
                                      int64_t nboxes, const int64_t* first_t, const int64_t* last_t, int64_t nsyl,
                                      int64_t* out_low, int64_t* out_box) {
    if (ncol < 0 || nidx < 0 || nsyl < 0 || nboxes < 0) return ta_fail(TA_EINVAL, "negative count");
    if (nsyl == 0) return TA_OK;
    if (!ops || !first_t || !last_t || !out_low || !out_box || (nidx > 0 && (!idx || !boxes)))
        return ta_fail(TA_EINVAL, "null pointer argument");
    std::vector<int64_t> col_of_t, o_at((size_t)ncol, -1);
    col_of_t.reserve((size_t)ncol);
    int64_t no = 0;
    for (int64_t c = 0; c < ncol; ++c) {
        if (ops[c] != 2) col_of_t.push_back(c);
        if (ops[c] != 1) o_at[(size_t)c] = no++;
    }
    if (no != nidx) return ta_fail(TA_EINVAL, "all_chars not same length as alignment");
    const int64_t nt = (int64_t)col_of_t.size();
    const int64_t kMin = INT64_MIN, kMax = INT64_MAX;
    for (int64_t s = 0; s < nsyl; ++s) {
        if (first_t[s] < 0 || last_t[s] < first_t[s] || last_t[s] >= nt) return ta_fail(TA_EINVAL, "a syllable lies outside the transcript");
        const int64_t c0 = col_of_t[(size_t)first_t[s]], c1 = col_of_t[(size_t)last_t[s]] + 1;
        int64_t low = kMin;
        for (int64_t c = c0; c < c1; ++c) {
            const int64_t o = o_at[(size_t)c];
            if (o < 0) continue;
            const int64_t r = idx[o];
            if (r < 0 || r >= nboxes) return ta_fail(TA_EINVAL, "a character index is outside the box table");
            const int64_t uly = boxes[4 * r + 1];
            if (uly > low) low = uly;
        }
        int64_t ulx = kMax, uly_ = kMax, lrx = kMin, lry = kMin;
        if (low != kMin) {
            for (int64_t c = c0; c < c1; ++c) {
                const int64_t o = o_at[(size_t)c];
                if (o < 0) continue;
                const int64_t* b = boxes + 4 * idx[o];
                if (b[1] != low) continue;
                if (b[0] < ulx) ulx = b[0];
                if (b[1] < uly_) uly_ = b[1];
                if (b[2] > lrx) lrx = b[2];
                if (b[3] > lry) lry = b[3];
            }
        }
        out_low[s] = low;
        out_box[4 * s] = ulx; out_box[4 * s + 1] = uly_; out_box[4 * s + 2] = lrx; out_box[4 * s + 3] = lry;
    }
    return TA_OK;
}

// ---- host arithmetic of the page preprocessing (no device work): the per-page numpy passes between two stages of
// csrc/ta_preproc.hip's pipeline, as plain loops that hold no interpreter lock.  Each reproduces the numpy expression
// it replaces operation by operation (this file is compiled without FMA contraction); the Python forms stay in
// preproc_gpu.py / textAlignPreprocessing.py as cross-checks (tests/test_preprocessing.py).

// Otsu's threshold of n 256-bin histograms (preproc_gpu.otsu_from_histogram): first maximum over t of
// (mean_all * cum[t] - mean_cum[t] * total)^2 / (cum[t] * (total - cum[t])), non-finite values counted as 0.
extern "C" int ta_host_otsu_batch(const int32_t* hist, int32_t n, int32_t* thr) {
    if (n < 0) return ta_fail(TA_EINVAL, "negative count");
    if (n == 0) return TA_OK;
    if (!hist || !thr) return ta_fail(TA_EINVAL, "null pointer argument");
    for (int32_t k = 0; k < n; ++k) {
        const int32_t* h = hist + (size_t)k * 256;
        double cum[256], mean_cum[256], total = 0.0, c = 0.0, m = 0.0;
        for (int t = 0; t < 256; ++t) {
            const double v = (double)h[t];
            c += v; cum[t] = c;                              // np.cumsum: left to right
            m += v * (double)t; mean_cum[t] = m;
        }
        // hist.sum(): pairwise in numpy, but a sum of integers below 2^53 is exact in any order
        total = cum[255];
        const double mean_all = mean_cum[255];
        int best = 0;
        double best_v = -1.0;
        for (int t = 0; t < 256; ++t) {
            const double num = mean_all * cum[t] - mean_cum[t] * total;
            double b = (num * num) / (cum[t] * (total - cum[t]));
            if (!std::isfinite(b)) b = 0.0;
            if (b > best_v) { best_v = b; best = t; }
        }
        thr[k] = best;
    }
    return TA_OK;
}

// numpy's pairwise summation of a contiguous float64 run (numpy/_core/src/umath/loops_utils.h.src, DOUBLE_pairwise_sum):
// what np.add.reduce does along a contiguous axis, so that sums agree with numpy's to the last bit
static double ta_np_pairwise_sum(const double* a, int64_t n) {
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return ta_np_pairwise_sum(a, n2) + ta_np_pairwise_sum(a + n2, n - n2);
}

// The skew sweep's choice for n pages (preproc_gpu._skew_search): page k's histograms are hist + off[k], nang[k] rows of
// hs[k] int32 counts; best[k] = int(np.argmax(np.var(rows, axis=1))) -- per row the mean (a sum of integers: exact),
// the squared deviations, their pairwise sum over n -- and any[k] = whether the page has a count at all.
// var_out (may be NULL): the variances, rows of all pages laid end to end.
extern "C" int ta_host_sharpest_rows(const int32_t* hist, const int64_t* off, const int32_t* nang, const int32_t* hs, int32_t n,
                                     int32_t* best, uint8_t* any, double* var_out) {
    if (n < 0) return ta_fail(TA_EINVAL, "negative count");
    if (n == 0) return TA_OK;
    if (!hist || !off || !nang || !hs || !best || !any) return ta_fail(TA_EINVAL, "null pointer argument");
    std::vector<double> sq;
    int64_t vpos = 0;
    for (int32_t k = 0; k < n; ++k) {
        if (nang[k] < 0 || hs[k] < 0 || off[k] < 0) return ta_fail(TA_EINVAL, "negative size");
        const int32_t* page = hist + off[k];
        const int64_t len = hs[k];
        sq.resize((size_t)len);
        int b = 0;
        double bv = 0.0;
        bool some = false, first = true;
        for (int32_t a = 0; a < nang[k]; ++a) {
            const int32_t* row = page + (int64_t)a * len;
            int64_t s = 0;
            for (int64_t i = 0; i < len; ++i) { s += row[i]; some = some || row[i] != 0; }
            double v;
            if (len == 0) {
                v = std::nan("");                            // (np.var of an empty row; argmax then stays at the first)
            } else {
                const double mean = (double)s / (double)len;
                for (int64_t i = 0; i < len; ++i) { const double x = (double)row[i] - mean; sq[(size_t)i] = x * x; }
                v = ta_np_pairwise_sum(sq.data(), len) / (double)len;
            }
            if (var_out) var_out[vpos++] = v;
            // np.argmax: the first maximum, a NaN counting as one
            if (first) { b = 0; bv = v; first = false; }
            else if (v > bv || (std::isnan(v) && !std::isnan(bv))) { b = a; bv = v; }
        }
        best[k] = b;
        any[k] = some ? 1 : 0;
    }
    return TA_OK;
}

// Bounding boxes of the text lines of a page (textAlignPreprocessing.line_boxes; reference :253-276): for every peak
// location the union of the components [ulx, uly, lrx, lry] that vertically coincide with the strip of half height
// `half` around it.  out_boxes [npeaks][4]; out_hit[p] = whether line p has a component at all.
extern "C" int ta_host_line_boxes(const int64_t* comps, int64_t ncomp, const int64_t* peaks, int64_t npeaks, int64_t half,
                                  int64_t* out_boxes, uint8_t* out_hit) {
    if (ncomp < 0 || npeaks < 0) return ta_fail(TA_EINVAL, "negative count");
    if (npeaks == 0) return TA_OK;
    if (!peaks || !out_boxes || !out_hit || (ncomp > 0 && !comps)) return ta_fail(TA_EINVAL, "null pointer argument");
    for (int64_t p = 0; p < npeaks; ++p) {
        const int64_t st = peaks[p] - half, sb = peaks[p] + half;
        int64_t ulx = INT64_MAX, uly = INT64_MAX, lrx = INT64_MIN, lry = INT64_MIN;
        bool hit = false;
        for (int64_t c = 0; c < ncomp; ++c) {
            const int64_t* b = comps + 4 * c;
            const int64_t top = b[1], bottom = b[1] + (b[3] - b[1] + 1);
            const bool above = top < st && bottom < st, below = top > sb && bottom > sb;
            if (above || below) continue;
            hit = true;
            if (b[0] < ulx) ulx = b[0];
            if (b[1] < uly) uly = b[1];
            if (b[2] > lrx) lrx = b[2];
            if (b[3] > lry) lry = b[3];
        }
        out_hit[p] = hit ? 1 : 0;
        out_boxes[4 * p] = ulx; out_boxes[4 * p + 1] = uly; out_boxes[4 * p + 2] = lrx; out_boxes[4 * p + 3] = lry;
    }
    return TA_OK;
}
