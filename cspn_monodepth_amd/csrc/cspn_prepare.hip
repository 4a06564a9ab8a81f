// cspn_prepare.hip — one-off passes around the propagation loop: weight preparation (3x3 abs/shift/normalise,
// K x K softmax) and the transposed tap volume for the backward recurrence.  See DESIGN.md §4.2.
#include "cspn_common.hpp"

namespace {

// ------------------------------------------------------------------------------------------------
// prepare kernels (run once per forward)
// ------------------------------------------------------------------------------------------------
// 3x3: w_j[p] = |g_{7-j}[p+off_j]| / S[p],  S[p] = sum_{k=0..7} |g_k[p+o_k]| summed in the
// reference's channel order k = 0..7 (CSPN_new.py:29-70, :124-127); quotients by div8_shared_reciprocal.
template <typename GT, typename WT>
__global__ void cspn3_prepare_kernel(const GT* __restrict__ g, long bs, long cs, int B, int H, int W, int Wv,
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
        if (x >= Wv) {               // row-padding column: no weights, and a divisor the backward can divide by
#pragma unroll
            for (int j = 0; j < 8; ++j) qv[j] = 0.f;
            S = 1.f;
        }
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
        for (int c = 0; c < NT; ++c) { v[c] = softmax_exp<WT>(v[c] - mx); den += v[c]; }
        // one reciprocal + NT multiplies instead of NT divisions (the pass is VALU-bound); den >= 1 unless NaN
        const float inv = reciprocal_refined(den);
#pragma unroll
        for (int c = 0; c < NT; ++c) st1(wk + (size_t)b * Taps<WT>::image_elems(NT, HW) + Taps<WT>::idx(c, p, HW), softmax_weight<WT>(v[c], inv));
    }
}

// The same, four pixels per thread: NT aligned quad loads, four softmaxes in registers, tap-volume quad stores
// (16-byte stores; for f16 a store carries two taps).  Needs H*W % 4 == 0 and 16-byte aligned tensors.
template <int K, typename GT, typename WT>
__global__ __launch_bounds__(256) void cspn_pac_prepare_vec_kernel(const GT* __restrict__ g, int B, size_t HW,
                                                                    WT* __restrict__ wk) {
    constexpr int NT = K * K - 1;
    const size_t nquads = (size_t)B * (HW >> 2);
    for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < nquads; q += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(q / (HW >> 2));
        const size_t p = (q - (size_t)b * (HW >> 2)) << 2;
        const GT* gb = g + (size_t)b * NT * HW + p;
        float v[NT][4];
        float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            const float4 t = ld4(gb + (size_t)c * HW);
            v[c][0] = t.x; v[c][1] = t.y; v[c][2] = t.z; v[c][3] = t.w;
#pragma unroll
            for (int e = 0; e < 4; ++e) mx[e] = fmaxf(mx[e], v[c][e]);
        }
        float den[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[c][e] = softmax_exp<WT>(v[c][e] - mx[e]); den[e] += v[c][e]; }
        float inv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) inv[e] = reciprocal_refined(den[e]);
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[c][e] = softmax_weight<WT>(v[c][e], inv[e]);
        store_taps_quad<NT>(wk + (size_t)b * Taps<WT>::image_elems(NT, HW), p, HW, v);
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

int cspn3_prepare(const void* guidance, int g_dtype, long bs, long cs, int B, int H, int W, int W_valid, void* w8,
                  int w_dtype, float* s_or_null, cspn_stream_t stream) {
    if (!guidance || !w8 || B <= 0 || H <= 0 || W <= 0 || W_valid < 0 || W_valid > W)
        return fail("cspn3_prepare: bad arguments");
    const int Wv = W_valid > 0 ? W_valid : W;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = grid_for((size_t)B * H * W, 256);
    if (g_dtype == CSPN_F32 && w_dtype == CSPN_F32)
        hipLaunchKernelGGL((cspn3_prepare_kernel<float, float>), dim3(grid), dim3(256), 0, st,
                           static_cast<const float*>(guidance), bs, cs, B, H, W, Wv, static_cast<float*>(w8), s_or_null);
    else if (g_dtype == CSPN_F16 && w_dtype == CSPN_F16)
        hipLaunchKernelGGL((cspn3_prepare_kernel<__half, __half>), dim3(grid), dim3(256), 0, st,
                           static_cast<const __half*>(guidance), bs, cs, B, H, W, Wv, static_cast<__half*>(w8), s_or_null);
    else if (g_dtype == CSPN_F16 && w_dtype == CSPN_F32)
        hipLaunchKernelGGL((cspn3_prepare_kernel<__half, float>), dim3(grid), dim3(256), 0, st,
                           static_cast<const __half*>(guidance), bs, cs, B, H, W, Wv, static_cast<float*>(w8), s_or_null);
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
    const size_t HW = (size_t)H * W;
    // f16: four pixels per thread (8-byte loads, 16-byte pair-interleaved stores): 38 -> 33 us at config 3.
    // f32 stays on the one-pixel kernel, which already runs at the HBM floor (320 MB in 56 us).
    if ((K == 3 || K == 5) && g_dtype == CSPN_F16 && (HW % 4 == 0) && aligned16(guided) && aligned16(wk)) {
        const int vgrid = grid_for((size_t)B * (HW / 4), 256);
#define PAC_PREP_V(KV, GTT)                                                                                       \
    hipLaunchKernelGGL((cspn_pac_prepare_vec_kernel<KV, GTT, GTT>), dim3(vgrid), dim3(256), 0, st,               \
                       static_cast<const GTT*>(guided), B, HW, static_cast<GTT*>(wk))
        if (g_dtype != w_dtype) return fail("cspn_pac_prepare: unsupported dtypes g=%d w=%d", g_dtype, w_dtype);
        if (K == 3 && g_dtype == CSPN_F32) PAC_PREP_V(3, float);
        else if (K == 3) PAC_PREP_V(3, __half);
        else if (g_dtype == CSPN_F32) PAC_PREP_V(5, float);
        else PAC_PREP_V(5, __half);
#undef PAC_PREP_V
        HIP_OK(hipGetLastError());
        return 1;
    }
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
