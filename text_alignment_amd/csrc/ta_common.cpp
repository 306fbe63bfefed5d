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
