// cspn_prepare.hip — one-off passes around the propagation loop: weight preparation (3x3 abs/shift/normalise,
// K x K softmax) and the transposed tap volume for the backward recurrence.  See DESIGN.md §4.2.
#include "cspn_common.hpp"

namespace {

// ------------------------------------------------------------------------------------------------
// prepare kernels (run once per forward)
// ------------------------------------------------------------------------------------------------
// 3x3: w_j[p] = |g_{7-j}[p+off_j]| / S[p],  S[p] = sum_{k=0..7} |g_k[p+o_k]| summed in the
// reference's channel order k = 0..7 (CSPN_new.py:29-70, :124-127).  True IEEE division.
template <typename GT, typename WT>
__global__ void cspn3_prepare_kernel(const GT* __restrict__ g, long bs, long cs, int B, int H, int W,
                                     WT* __restrict__ w8, float* __restrict__ s_out) {
    const size_t HW = (size_t)H * W;
    const size_t total = (size_t)B * HW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / HW);
        const int p = (int)(i - (size_t)b * HW);
        const int y = p / W, x = p - y * W;
        const GT* gb = g + (size_t)b * bs;
        float a[8];
        float S = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            // reference plane k samples at o_k = -(off of tap k) ... tap j = 7-k, off_j row-major
            const int j = 7 - k;
            const int lin = j < 4 ? j : j + 1;
            const int dy = lin / 3 - 1, dx = lin % 3 - 1;
            const int yy = y + dy, xx = x + dx;
            float v = 0.f;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = fabsf(ld1(gb + (size_t)k * cs + (size_t)yy * W + xx));
            a[j] = v;
            S = (k == 0) ? v : S + v;
        }
        float qv[8];
        div8_shared_reciprocal(a, S, qv);
#pragma unroll
        for (int j = 0; j < 8; ++j) st1(w8 + (size_t)b * Taps<WT>::image_elems(8, HW) + Taps<WT>::idx(j, p, HW), qv[j]);
        if (s_out) s_out[i] = S;
    }
}

// K x K: softmax over the K*K-1 channels at the centre pixel (CSPN_ours.py:35); tap j = channel j.
template <int K, typename GT, typename WT>
__global__ void cspn_pac_prepare_kernel(const GT* __restrict__ g, int B, int H, int W, WT* __restrict__ wk) {
    constexpr int NT = K * K - 1;
    const size_t HW = (size_t)H * W;
    const size_t total = (size_t)B * HW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / HW);
        const size_t p = i - (size_t)b * HW;
        const GT* gb = g + (size_t)b * NT * HW + p;
        float v[NT];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < NT; ++c) { v[c] = ld1(gb + (size_t)c * HW); mx = fmaxf(mx, v[c]); }
        float den = 0.f;
#pragma unroll
        for (int c = 0; c < NT; ++c) { v[c] = expf(v[c] - mx); den += v[c]; }
#pragma unroll
        for (int c = 0; c < NT; ++c) st1(wk + (size_t)b * Taps<WT>::image_elems(NT, HW) + Taps<WT>::idx(c, p, HW), v[c] / den);
    }
}

// wT_j[q] = w_{NT-1-j}[q + off_j]  (0 outside)
template <int K, typename WT>
__global__ void cspn_transpose_kernel(const WT* __restrict__ w, WT* __restrict__ wT, int B, int H, int W) {
    constexpr int R = K / 2;
    constexpr int NT = K * K - 1;
    const size_t HW = (size_t)H * W;
    const size_t total = (size_t)B * HW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / HW);
        const int p = (int)(i - (size_t)b * HW);
        const int y = p / W, x = p - y * W;
        int j = 0;
        for (int dy = -R; dy <= R; ++dy)
            for (int dx = -R; dx <= R; ++dx) {
                if (dy == 0 && dx == 0) continue;
                const int yy = y + dy, xx = x + dx;
                float v = 0.f;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W)
                    v = ld1(w + (size_t)b * Taps<WT>::image_elems(NT, HW) + Taps<WT>::idx(NT - 1 - j, (size_t)yy * W + xx, HW));
                st1(wT + (size_t)b * Taps<WT>::image_elems(NT, HW) + Taps<WT>::idx(j, p, HW), v);
                ++j;
            }
    }
}

}  // namespace

extern "C" {

int cspn3_prepare(const void* guidance, int g_dtype, long bs, long cs, int B, int H, int W, void* w8,
                  int w_dtype, float* s_or_null, cspn_stream_t stream) {
    if (!guidance || !w8 || B <= 0 || H <= 0 || W <= 0) return fail("cspn3_prepare: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = grid_for((size_t)B * H * W, 256);
    if (g_dtype == CSPN_F32 && w_dtype == CSPN_F32)
        hipLaunchKernelGGL((cspn3_prepare_kernel<float, float>), dim3(grid), dim3(256), 0, st,
                           static_cast<const float*>(guidance), bs, cs, B, H, W, static_cast<float*>(w8), s_or_null);
    else if (g_dtype == CSPN_F16 && w_dtype == CSPN_F16)
        hipLaunchKernelGGL((cspn3_prepare_kernel<__half, __half>), dim3(grid), dim3(256), 0, st,
                           static_cast<const __half*>(guidance), bs, cs, B, H, W, static_cast<__half*>(w8), s_or_null);
    else if (g_dtype == CSPN_F16 && w_dtype == CSPN_F32)
        hipLaunchKernelGGL((cspn3_prepare_kernel<__half, float>), dim3(grid), dim3(256), 0, st,
                           static_cast<const __half*>(guidance), bs, cs, B, H, W, static_cast<float*>(w8), s_or_null);
    else
        return fail("cspn3_prepare: unsupported dtypes g=%d w=%d", g_dtype, w_dtype);
    HIP_OK(hipGetLastError());
    return 1;
}

int cspn_pac_prepare(const void* guided, int g_dtype, int B, int H, int W, int K, void* wk, int w_dtype,
                     cspn_stream_t stream) {
    if (!guided || !wk || B <= 0 || H <= 0 || W <= 0) return fail("cspn_pac_prepare: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = grid_for((size_t)B * H * W, 256);
#define PAC_PREP(KV)                                                                                           \
    if (K == KV) {                                                                                             \
        if (g_dtype == CSPN_F32 && w_dtype == CSPN_F32)                                                        \
            hipLaunchKernelGGL((cspn_pac_prepare_kernel<KV, float, float>), dim3(grid), dim3(256), 0, st,      \
                               static_cast<const float*>(guided), B, H, W, static_cast<float*>(wk));           \
        else if (g_dtype == CSPN_F16 && w_dtype == CSPN_F16)                                                   \
            hipLaunchKernelGGL((cspn_pac_prepare_kernel<KV, __half, __half>), dim3(grid), dim3(256), 0, st,    \
                               static_cast<const __half*>(guided), B, H, W, static_cast<__half*>(wk));         \
        else                                                                                                   \
            return fail("cspn_pac_prepare: unsupported dtypes g=%d w=%d", g_dtype, w_dtype);                   \
        HIP_OK(hipGetLastError());                                                                             \
        return 1;                                                                                              \
    }
    PAC_PREP(3) PAC_PREP(5) PAC_PREP(7)
#undef PAC_PREP
    return fail("cspn_pac_prepare: unsupported K=%d (3, 5, 7)", K);
}

int cspn_transpose_weights(const void* w, void* wT, int w_dtype, int B, int H, int W, int K, cspn_stream_t stream) {
    if (!w || !wT) return fail("cspn_transpose_weights: NULL pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = grid_for((size_t)B * H * W, 256);
#define TR(KV)                                                                                              \
    if (K == KV) {                                                                                          \
        if (w_dtype == CSPN_F32)                                                                            \
            hipLaunchKernelGGL((cspn_transpose_kernel<KV, float>), dim3(grid), dim3(256), 0, st,            \
                               static_cast<const float*>(w), static_cast<float*>(wT), B, H, W);             \
        else                                                                                                \
            hipLaunchKernelGGL((cspn_transpose_kernel<KV, __half>), dim3(grid), dim3(256), 0, st,           \
                               static_cast<const __half*>(w), static_cast<__half*>(wT), B, H, W);           \
        HIP_OK(hipGetLastError());                                                                          \
        return 1;                                                                                           \
    }
    TR(3) TR(5) TR(7)
#undef TR
    return fail("cspn_transpose_weights: unsupported K=%d", K);
}

}  // extern "C"
