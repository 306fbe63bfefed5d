// ta_common.h -- error plumbing shared by the C-ABI translation units of libta_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#include "../../include/text_alignment_amd.h"

// record a failure for ta_last_error() and return `code`
int ta_fail(int code, const char* what);
int ta_fail_hip(hipError_t e, const char* where);

// Raise of a kernel's dynamic-LDS limit above the default 64 KiB, once per kernel instantiation AND device (the
// attribute belongs to the device's copy of the code object; a process that drives several GPUs needs it on each).
// Thread-safe; the only mutable state the library keeps besides the lazily loaded code object.
template <typename K>
static inline hipError_t allow_full_lds(K kernel) {
    constexpr int kMaxDev = 64;
    static std::atomic<int> done[kMaxDev];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < kMaxDev && done[dev].load(std::memory_order_acquire)) return hipSuccess;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess && dev >= 0 && dev < kMaxDev) done[dev].store(1, std::memory_order_release);
    return e;
}
