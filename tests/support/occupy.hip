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
