// cspn_debug.hip — the poisoned-LDS debugging aid (include/cspn_hip.h: cspn_debug_set_lds_poison).
//
// LDS is not cleared between kernels: a workgroup inherits what the previous one on its CU left there.  A kernel that reads a
// word it never wrote is therefore correct or not depending on what ran before it — invisible to any test that runs the kernel
// in isolation (round 4's flake probe: 900 clean repetitions of a case that failed once in ten suite runs).  With the switch on,
// every launch of the library (CSPN_PRELAUNCH, cspn_common.hpp) is preceded by this fill of the whole LDS of every CU.
#include "cspn_common.hpp"

#include <atomic>

namespace cspn_detail {

int g_lds_poison_on = 0;
static std::atomic<unsigned> g_pattern{0x7fc00000u};

namespace {
constexpr int POISON_LDS_BYTES = 160 * 1024;
// one workgroup per CU at a time (the allocation is the whole LDS); the short hold makes the dispatcher hand the first
// n_cu workgroups to n_cu different CUs even when the fill itself would be over before the last one is placed
__global__ __launch_bounds__(256) void lds_poison_kernel(unsigned pattern, unsigned long long ticks, unsigned* sink) {
    extern __shared__ unsigned hold[];
    for (int i = threadIdx.x; i < POISON_LDS_BYTES / 4; i += 256) hold[i] = pattern;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (sink && hold[(threadIdx.x * 97) % (POISON_LDS_BYTES / 4)] != pattern) sink[0] = 1;      // never true: keeps the stores alive
}
}  // namespace

void lds_poison(hipStream_t st) {
    static std::atomic<int> granted[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    if (!granted[dev & 63].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(lds_poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, POISON_LDS_BYTES) != hipSuccess)
            return;
        granted[dev & 63].store(1, std::memory_order_release);
    }
    hipDeviceProp_t prop;
    static std::atomic<int> ncu[64];
    int n = ncu[dev & 63].load(std::memory_order_relaxed);
    if (n <= 0) {
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return;
        n = prop.multiProcessorCount;
        ncu[dev & 63].store(n, std::memory_order_relaxed);
    }
    // 2 x CUs workgroups, each holding its CU for ~3 us (300 ticks of the 100 MHz clock)
    hipLaunchKernelGGLInternal(lds_poison_kernel, dim3(2 * n), dim3(256), POISON_LDS_BYTES, st, g_pattern.load(std::memory_order_relaxed), 300ull,
                               static_cast<unsigned*>(nullptr));
    (void)hipGetLastError();
}

}  // namespace cspn_detail

extern "C" int cspn_debug_set_lds_poison(int enabled, unsigned pattern, int* previous_or_null) {
    if (previous_or_null) *previous_or_null = cspn_detail::g_lds_poison_on;
    cspn_detail::g_pattern.store(pattern, std::memory_order_relaxed);
    cspn_detail::g_lds_poison_on = enabled ? 1 : 0;
    return 1;
}
