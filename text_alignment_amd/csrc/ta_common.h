// ta_common.h -- error plumbing shared by the C-ABI translation units of libta_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/text_alignment_amd.h"

// record a failure for ta_last_error() and return `code`
int ta_fail(int code, const char* what);
int ta_fail_hip(hipError_t e, const char* where);
