// ta_common.cpp -- ta_version / ta_last_error and the per-thread error record.
#include "ta_common.h"

#include <cstdio>
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
