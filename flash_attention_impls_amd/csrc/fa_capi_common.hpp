// Shared by the C-ABI translation units: per-thread error message and stride decoding.
#pragma once
#include "fa_build_guard.hpp"      // first: it looks at the switches the command line set, before any header defines defaults
#include <hip/hip_runtime.h>
#include <algorithm>
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

// Compute units of the current device (hipDeviceProp_t::multiProcessorCount, read once per device; 256 if the query fails):
// the launch heuristics below count rounds of workgroups against it instead of assuming a whole MI355X (SPX mode, 256 CUs),
// so a partitioned device (DPX / QPX / CPX: 128 / 64 / 32 CUs) is scheduled by its own size.  The XCD count stays 8 in the
// kernels' workgroup decode (blockIdx % 8): on a partition with fewer XCDs that is only a permutation of the blocks.
inline int device_cus()
{
    constexpr int kMaxDevices = 64;
    static std::atomic<int> cus[kMaxDevices];          // zero-initialised; 0 = not read yet
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 256;
    int n = cus[dev].load(std::memory_order_acquire);
    if (n > 0) return n;
    hipDeviceProp_t prop;
    n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    cus[dev].store(n, std::memory_order_release);
    return n;
}

// Grid mapping shared by all kernels (wg_decode in fa_fwd_kernel.hpp): blockIdx % 8 = XCD, all blocks of a (batch, head)
// slice on one XCD.  A head count that is small and not a multiple of 8 would leave XCDs idle (4 heads: half the chip):
// each head then becomes 2, 4 or 8 virtual heads that share its blocks, so that the virtual heads fill all 8 XCDs evenly.
inline int head_split(long long heads, long long per_head)
{
    if (heads <= 0) return 1;
    // the smallest split that leaves at most 10 % of the (virtual head, XCD) slots empty: every doubling also doubles
    // the number of L2s a head's K / V are fetched into
    int f = 1;
    for (; f < 8; f *= 2) {
        const long long v = heads * f, slots = ((v + 7) / 8) * 8;
        if ((slots - v) * 10 <= slots) break;
    }
    while (f > 1 && f > per_head) f /= 2;            // never more virtual heads than blocks to share
    return f;
}
inline long long grid_blocks(long long heads, long long per_head, int hsplit)
{
    const long long vheads = ((heads * hsplit + 7) / 8) * 8;
    return vheads * ((per_head + hsplit - 1) / hsplit);
}

// Causal launches: pair the blocks of a head (nb-1-t, t) into workgroups of equal work, or launch them one by one, longest
// first?  Pairs are perfectly balanced but quantised: G workgroups on `cus` CUs (256 on a whole MI355X) take ceil(G / cus) rounds of nb + 1 block
// units.  Single blocks take about max(nb, total work / 256) with some slack for the greedy order.  Returns true when the
// single-block launch is expected to finish first (small grids, and grids whose last round of pairs would be mostly empty).
// `lone_waves`: 128-row workgroups (one wave per SIMD; grids that do not fill the chip): there single blocks win at equal estimates
// (round 4, tools/exp_rows.py: (1,2,16384,128) 602 against 575 TFLOP/s, (1,4,4096,128) 273 against 247; equal where pairs fill the chip).
inline bool causal_unpaired(long long heads, long long nb, int cus = 256, bool lone_waves = false)
{
    if (nb <= 1) return false;
    const long long pairs = heads * ((nb + 1) / 2);
    const double t_pair = (double)((pairs + cus - 1) / cus) * (double)(nb + 1);
    const double work = (double)heads * (double)nb * (double)(nb + 1) / 2.0;
    const double t_single = std::max((double)nb, 1.2 * work / (double)cus);
    if (lone_waves) return t_single <= t_pair;
    return t_single < 0.85 * t_pair;                 // (measured: at equal estimates the pairs win, the greedy order is no LPT)
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
