// ta_common.h -- error plumbing shared by the C-ABI translation units of libta_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#include "../../include/text_alignment_amd.h"

// record a failure for ta_last_error() and return `code`
int ta_fail(int code, const char* what);
int ta_fail_hip(hipError_t e, const char* where);

// Raise of a kernel's dynamic-LDS limit above the default 64 KiB, once per KERNEL and device (the attribute belongs to the
// device's copy of the code object; a process that drives several GPUs needs it on each).  The guard is keyed on the
// kernel's ADDRESS: instantiations of one template share a function-pointer TYPE (all lstm_output_kernel<NCT> are
// void (*)(OutArgs)), so a guard per type would skip the attribute for every instantiation after the first.
// Thread-safe; the only mutable state the library keeps besides the lazily loaded code object.
#include <mutex>
#include <set>
#include <utility>
template <typename K>
static inline hipError_t allow_full_lds(K kernel) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const std::pair<int, const void*> key(dev, reinterpret_cast<const void*>(kernel));
    std::lock_guard<std::mutex> lock(mu);
    if (done.count(key)) return hipSuccess;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) done.insert(key);
    return e;
}
