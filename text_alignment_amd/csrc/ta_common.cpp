// ta_common.cpp -- ta_version / ta_last_error and the per-thread error record.
#include "ta_common.h"

#include <cstdio>

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
