// Shared by the C-ABI translation units: per-thread error message and stride decoding.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

namespace fa_capi {

inline thread_local char g_err[512] = "";

inline int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// NULL = contiguous [B][H][S][D]; otherwise element strides (batch, head, seq), head_dim stride 1
inline bool set_strides(const int64_t* s, int H, int S, int D, long long& sb, long long& sh, long long& ss)
{
    if (s == nullptr) { ss = D; sh = (long long)S * D; sb = (long long)H * S * D; return true; }
    sb = s[0]; sh = s[1]; ss = s[2];
    return sb >= 0 && sh >= 0 && ss >= D;
}

// One-time, PER-DEVICE opt-in of a kernel to more than 64 KiB of dynamic LDS (the attribute belongs to the function on
// the current device: a process that drives several GPUs needs it on each).  KernelTag gives every kernel instantiation
// its own flags; lock-free on the fast path, and harmless if two threads race to set the same attribute.
// Returns hipSuccess or the error of hipGetDevice / hipFuncSetAttribute.
template <class KernelTag>
inline hipError_t ensure_dynamic_lds(const void* kernel, int lds_bytes)
{
    constexpr int kMaxDevices = 64;
    static std::atomic<bool> done[kMaxDevices];        // zero-initialised
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const bool tracked = dev >= 0 && dev < kMaxDevices;
    if (tracked && done[dev].load(std::memory_order_acquire)) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e == hipSuccess && tracked) done[dev].store(true, std::memory_order_release);
    return e;
}

}  // namespace fa_capi
