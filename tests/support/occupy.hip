// occupy.hip — TEST SUPPORT (not product): a co-tenant for the contention tests of the weight-resident launches.
// `occupy(n_wg, lds_bytes, ticks, stream)` launches n_wg workgroups of 1024 threads that each hold `lds_bytes` of LDS (120 KB:
// no resident workgroup fits next to one on a CU) and spin until `ticks` of the 100 MHz wall clock have passed.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tests/support/libcspn_occupy.so tests/support/occupy.hip
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(1024) void occupy_kernel(unsigned long long ticks, unsigned* sink) {
    extern __shared__ unsigned hold[];
    hold[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    unsigned n = 0;
    while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); ++n; }
    if (n == 0xffffffffu) sink[0] = hold[(threadIdx.x + 1) & 1023];      // never true: keeps the LDS allocation alive
}

extern "C" int occupy(int n_wg, int lds_bytes, unsigned long long ticks, unsigned* sink, void* stream) {
    if (lds_bytes > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
        return 0;
    hipLaunchKernelGGL(occupy_kernel, dim3(n_wg), dim3(1024), lds_bytes, static_cast<hipStream_t>(stream), ticks, sink);
    return hipGetLastError() == hipSuccess ? 1 : 0;
}

// lds_poison: fill the whole LDS (160 KB) of every CU with `pattern`.  Kernels do not get a cleared LDS — they inherit
// whatever the previous workgroup on the CU left — so a kernel that reads an LDS word it never wrote gives results that depend
// on what ran before it (round 5: the root cause of the one bit mismatch of the default path at KITTI B = 8 with sparse depth).
// The poisoned-LDS tests run this in front of every engine call: a NaN pattern makes any such read visible.
// One workgroup per CU at a time (the allocation is the whole LDS), `n_wg` >= 4 x CUs of them, each holding its CU for `ticks`
// so that the dispatcher has to spread them over every CU.
__global__ __launch_bounds__(256) void lds_poison_kernel(unsigned pattern, int words, unsigned long long ticks, unsigned* sink) {
    extern __shared__ unsigned hold[];
    for (int i = threadIdx.x; i < words; i += 256) hold[i] = pattern;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    if (hold[(threadIdx.x * 97) % words] != pattern) sink[1] = 1;      // never true: keeps the stores alive
}

extern "C" int lds_poison(int n_wg, unsigned pattern, unsigned long long ticks, unsigned* sink, void* stream) {
    const int lds_bytes = 160 * 1024;
    static bool granted = false;
    if (!granted) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(lds_poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
            return 0;
        granted = true;
    }
    hipLaunchKernelGGL(lds_poison_kernel, dim3(n_wg), dim3(256), lds_bytes, static_cast<hipStream_t>(stream), pattern, lds_bytes / 4, ticks, sink);
    return hipGetLastError() == hipSuccess ? 1 : 0;
}
