// cspn_metrics.hip — depth-metrics reduction (libs/metrics.py:49-83) + the library's error channel / version.
#include "cspn_common.hpp"

namespace cspn_detail {

static thread_local char g_err[512] = "";

int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 0;
}

const char* last_error() { return g_err; }

}  // namespace cspn_detail

namespace {

// ------------------------------------------------------------------------------------------------
// evaluation metrics: masked sums (libs/metrics.py:49-83)
// ------------------------------------------------------------------------------------------------
template <typename DT>
__global__ __launch_bounds__(1024) void cspn_metrics_kernel(const DT* __restrict__ pred, const DT* __restrict__ target,
                                                            size_t n, int vec_ok, double* __restrict__ acc,
                                                            int nslots) {
    // Per-thread partial sums stay in fp32 (a thread sees ~a dozen pixels); fp64 starts at the wave reduction.
    float f[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) f[k] = 0.f;
    auto one = [&](float o, float t) { metric_terms(o, t, f); };
    const size_t gtid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * blockDim.x;
    const size_t nq = vec_ok ? n / 4 : 0;
    for (size_t q = gtid; q < nq; q += gsz) {
        const float4 o = ld4(pred + 4 * q), t = ld4(target + 4 * q);
        one(o.x, t.x); one(o.y, t.y); one(o.z, t.z); one(o.w, t.w);
    }
    for (size_t i = 4 * nq + gtid; i < n; i += gsz) one(ld1(pred + i), ld1(target + i));
    double s[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) s[k] = (double)f[k];
    // wavefront sum by DPP (no LDS crossbar) -> LDS -> one atomic per block and quantity (10 per block)
    __shared__ double part[16][10];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const double v = wave_sum_to_lane63(s[k]);
        if (lane == 63) part[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 10) {
        double v = 0.0;
        const int nw = (blockDim.x + 63) >> 6;
        for (int w = 0; w < nw; ++w) v += part[w][threadIdx.x];
        // contention on one address costs ~45 ns per atomic: spread the blocks over `nslots` accumulator rows
        if (v != 0.0) atomicAdd(acc + (size_t)(blockIdx.x % nslots) * 10 + threadIdx.x, v);
    }
}

}  // namespace

extern "C" {

int cspn_abi_version(void) { return CSPN_ABI_VERSION; }
const char* cspn_last_error(void) { return cspn_detail::last_error(); }

int cspn_metrics_accumulate(const void* pred, const void* target, int dtype, size_t n, double* acc, int nslots,
                            cspn_stream_t stream) {
    if (!pred || !target || !acc || nslots < 1) return fail("cspn_metrics_accumulate: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = grid_for(n / 4 + 1, 1024) > 256 ? 256 : grid_for(n / 4 + 1, 1024);   // <= one block per CU
    const int vec_ok = aligned16(pred) && aligned16(target);
    if (dtype == CSPN_F32)
        hipLaunchKernelGGL((cspn_metrics_kernel<float>), dim3(grid), dim3(1024), 0, st,
                           static_cast<const float*>(pred), static_cast<const float*>(target), n, vec_ok, acc, nslots);
    else if (dtype == CSPN_F16)
        hipLaunchKernelGGL((cspn_metrics_kernel<__half>), dim3(grid), dim3(1024), 0, st,
                           static_cast<const __half*>(pred), static_cast<const __half*>(target), n, vec_ok, acc, nslots);
    else
        return fail("cspn_metrics_accumulate: unsupported dtype %d", dtype);
    HIP_OK(hipGetLastError());
    return 1;
}

}  // extern "C"
