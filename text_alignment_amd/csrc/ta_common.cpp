// ta_common.cpp -- ta_version / ta_last_error and the per-thread error record.
#include "ta_common.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

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
                                      int64_t* out_line, int64_t* out_cp, int64_t* out_boxes, int64_t* out_count) {
    if (nlines < 0 || ncps < 0) return ta_fail(TA_EINVAL, "negative count");
    if (!out_count) return ta_fail(TA_EINVAL, "null pointer argument");
    *out_count = 0;
    if (nlines == 0) return TA_OK;
    if (!dec_t || !dec_c || !dec_n || !dec_off || !T || !raw_w || !x_min || !y_min || !y_max || !cps || !out_line || !out_cp || !out_boxes)
        return ta_fail(TA_EINVAL, "null pointer argument");
    int64_t k = 0;
    for (int32_t b = 0; b < nlines; ++b) {
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
