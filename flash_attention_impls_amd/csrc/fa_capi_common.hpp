// Shared by the C-ABI translation units: per-thread error message and stride decoding.
#pragma once
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

}  // namespace fa_capi
