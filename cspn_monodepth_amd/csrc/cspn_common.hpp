// cspn_common.hpp — shared device/host helpers of the CSPN HIP engine (included by every translation unit).
// Everything here has internal linkage (anonymous namespace / inline) except the thread-local error channel.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <type_traits>

#include "cspn_hip.h"

namespace cspn_detail {
// thread-local message of the last failing entry point on this thread (defined in cspn_metrics.hip)
int fail(const char* fmt, ...);
const char* last_error();
// cspn_resident.hip: the K = 3 softmax-weight form of the quad-based resident kernel (used by cspnk_forward_resident)
int resident_pac3_f32(const void* guided, const void* x0, const void* sparse, void* out, void* history, void* wk_out, void* work,
                      unsigned seq, unsigned* host_err, int B, int H, int W, int T, int blend, const void* target, double* acc,
                      int nslots, const cspn_resident_plan* plan, void* stream);
// cspnk_d2.hip: the K = 5 fp16 form with the state packed as fp16 pairs in LDS and v_dot2_f32_f16 steps (launched by cspnk_forward_resident)
int kres_d2_row_stride(int wo);
size_t kres_d2_lds_bytes(int dr, int ls, int threads, int npf);
int kres_d2_prefetch_channels(int dr, int ls, int threads, int rounds);      // guidance channels staged through LDS (0 or 8)
int kres_d2_launch(const void* kres_args, int threads, int grid, size_t lds_bytes, int blend, int mode, int clean, int npf, void* stream);   // mode 0 plain, 1 scored, 2 history
// pac_conv2d_s2.hip: the pixel-adaptive convolution and its gradients for stride 2 x 2, dilation 1, K in {3, 5}, padding K / 2,
// W % 8 == 0 (launched by the cspn_pac_conv2d* entry points of pac_conv2d.hip when every base pointer is 16-byte aligned)
struct PacS2Args {
    int B, C, CK, H, W, Ho, Wo, WQ;      // WQ = Wo / 4 = W / 8 quads per output row
    int cchunk, nchunk, gx;              // channels per chunk, chunks, spatial workgroups per image (set by the launcher)
};
bool pac_s2_geometry(int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int W);
int pac_s2_forward(const void* in, const void* kern, void* out, int dtype, int K, const PacS2Args& a, void* stream);
int pac_s2_grad_input(const void* gout, const void* kern, void* gin, int dtype, int K, const PacS2Args& a, void* stream);
int pac_s2_grad_kernel(const void* gout, const void* in, void* gk, int dtype, int K, const PacS2Args& a, void* stream);
// cspn_repair.hip: the guard behind a plain resident inference launch (cspn_resident_plan.guard): re-computes the call's result
// on the stream when — and only when — its launches gave up (abort word == seq)
bool resident_repair_fits(int T);
int resident_repair_launch(const float* g, long bs, long cs, const float* d0, const float* sparse, float* out, float* hist, float* s_out,
                           float* w_out, const float* s_in, int mode, const unsigned* abort_word, unsigned seq, int B, int H, int W, int Wv,
                           int T, int blend, int n_cu, void* stream, const float* target = nullptr, double* acc = nullptr, int nslots = 0);
                           // mode 0: inference, 1: inference + fused metrics, 2: training forward, 3 / 4: reverse sweep from a tap volume / from
                                                                                 // guidance + S, 10 / 12: softmax-weight (CSPN_ours K = 3) inference / training forward
// ... and of cspnk_forward_resident's unscored inference calls (round_every: the steps between two roundings of the state to the plane dtype)
bool kres_repair_fits(int K, int T);
int kres_repair_launch(const void* g, int g_dtype, int K, const void* x0, const void* sparse, void* out, int state_dtype,
                       const unsigned* abort_word, unsigned seq, int B, int H, int W, int T, int round_every, int blend, int n_cu, void* stream);
// ... and of the K = 5 fp16 training forms: cspnk_forward_resident_history's dot-product launch and cspnk_transposed_resident
int kres_history_repair_launch(const void* g, const void* x0, const void* sparse, void* hist, void* wk_out, const unsigned* abort_word, unsigned seq,
                               int B, int H, int W, int T, int blend, int n_cu, void* stream);
int kres_sweep_repair_launch(const void* wk, const void* g_T, const void* sparse, int in_dtype, float* g32_out, float* hist, const unsigned* abort_word,
                             unsigned seq, int B, int H, int W, int T, int premask, int n_cu, void* stream);
// cspn_debug.hip: the poisoned-LDS debugging aid (include/cspn_hip.h: cspn_debug_set_lds_poison)
extern int g_lds_poison_on;
void lds_poison(hipStream_t st);
}  // namespace cspn_detail

// Every launch of the library goes through this hook: with the poisoned-LDS switch on, the whole LDS of every CU is filled with
// a pattern in front of the kernel (a kernel that reads LDS it never wrote then shows it).  Off: one predictable branch.
#define CSPN_PRELAUNCH(st)                                                                \
    do {                                                                                  \
        if (__builtin_expect(cspn_detail::g_lds_poison_on, 0)) cspn_detail::lds_poison(st);   \
    } while (0)
#define CSPN_PRE(st) (__builtin_expect(cspn_detail::g_lds_poison_on, 0) ? cspn_detail::lds_poison(st) : (void)0)   // expression form: `CSPN_PRE(st), kernel<<<...>>>(...)`
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kern, grid, block, lds, st, ...)                                \
    do {                                                                                  \
        CSPN_PRELAUNCH(st);                                                               \
        hipLaunchKernelGGLInternal((kern), (grid), (block), (lds), (st), __VA_ARGS__);    \
    } while (0)

namespace {

using cspn_detail::fail;

#define HIP_OK(expr)                                                                  \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) return fail("%s: %s", #expr, hipGetErrorString(e_));    \
    } while (0)

// ------------------------------------------------------------------------------------------------
// element access helpers: everything is computed in fp32, storage is f32 or f16
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const __half* p) {
    const uint2 raw = *reinterpret_cast<const uint2*>(p);
    const float2 a = __half22float2(*reinterpret_cast<const __half2*>(&raw.x));
    const float2 b = __half22float2(*reinterpret_cast<const __half2*>(&raw.y));
    return make_float4(a.x, a.y, b.x, b.y);
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(__half* p, float4 v) {
    uint2 raw;
    *reinterpret_cast<__half2*>(&raw.x) = __floats2half2_rn(v.x, v.y);
    *reinterpret_cast<__half2*>(&raw.y) = __floats2half2_rn(v.z, v.w);
    *reinterpret_cast<uint2*>(p) = raw;
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const __half* p) { return __half2float(*p); }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(__half* p, float v) { *p = __float2half_rn(v); }

__device__ __forceinline__ float sgnf(float v) { return (float)((v > 0.f) - (v < 0.f)); }
__device__ __forceinline__ float4 sgn4(float4 v) { return make_float4(sgnf(v.x), sgnf(v.y), sgnf(v.z), sgnf(v.w)); }

// q[k] = a[k] / S for the 8 gates of one pixel (0 <= a[k] <= S), sharing one reciprocal:
//   r = v_rcp_f32(S) refined by one Newton step (r <- r + r (1 - S r)), q[k] = a[k] * r
// — 3 + 8 VALU operations.  Each quotient is within ~1.5 ulp of the IEEE one; the reference itself never forms these
// quotients (it divides the weighted SUM by S every step, CSPN_new.py:124-127), so neither rounding is "the" reference
// one: measured against it on full frames the refined depth moves from 4.6e-7 to 5.7e-7 max relative error (bar: 1e-5)
// while the derive launch loses 9 us of division sequences (v_div_scale / v_div_fmas / v_div_fixup + two residual
// corrections per quotient: 43 operations, kept below under CSPN_IEEE_NORMALISE for A/B runs).
// S = 0 (all gates zero, also every padding pixel) needs no branch: rcp(0) = inf, the Newton step makes it NaN and
// 0 * NaN = NaN = 0 / 0.  Only a denormal-range or huge S — where v_rcp_f32 flushes — takes the true division.
__device__ __forceinline__ void div8_shared_reciprocal(const float (&a)[8], float S, float (&q)[8]) {
#ifndef CSPN_IEEE_NORMALISE
    const bool okr = (S <= 0x1p+100f) && (S >= 0x1p-100f || S == 0.f);
    if (okr) {
        float r = __builtin_amdgcn_rcpf(S);
        const float e = fmaf(-S, r, 1.0f);
        r = fmaf(e, r, r);
#pragma unroll
        for (int k = 0; k < 8; ++k) q[k] = a[k] * r;
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) q[k] = a[k] / S;
    }
#else
    // bit-identical to IEEE division: the compiler's fp32 division sequence with the divisor-only part shared
    bool fast = (S >= 0x1p-60f) && (S <= 0x1p+60f);
#pragma unroll
    for (int k = 0; k < 8; ++k) fast = fast && (a[k] == 0.f || a[k] >= S * 0x1p-40f);
    if (fast) {
        float r = __builtin_amdgcn_rcpf(S);
        const float e = fmaf(-S, r, 1.0f);
        r = fmaf(e, r, r);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float qq = a[k] * r;
            const float e2 = fmaf(-S, qq, a[k]);
            qq = fmaf(e2, r, qq);
            const float e3 = fmaf(-S, qq, a[k]);
            q[k] = fmaf(e3, r, qq);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) q[k] = a[k] / S;
    }
#endif
}

// exp(x) for x <= 0 (the max-subtracted logits) on the hardware exponential: v_exp_f32 is 2^t to 1 ulp; the product
// x * log2(e) is formed in two pieces (fma recovers its rounding error) so that the result stays within ~2 ulp of
// expf for every x instead of drifting by |x| * 2^-24.  libm's expf made both prepare kernels VALU-bound (~25 VALU
// operations per call, 24 calls per pixel at K = 5: 33.6 us for a pass whose HBM floor is 20 us).
__device__ __forceinline__ float exp_nonpositive(float x) {
    const float L2E = 1.44269502162933349609375f;          // float(log2(e))
    const float L2E_LO = 1.925963033500011e-8f;            // log2(e) - float(log2(e))
    const float t = x * L2E;
    const float r = fmaf(x, L2E, -t) + x * L2E_LO;         // what t lost
    const float e = __builtin_amdgcn_exp2f(t);
    return fmaf(e, r * 0.693147180559945f, e);             // 2^(t+r) = 2^t (1 + r ln 2 + ...)
}
__device__ __forceinline__ float reciprocal_refined(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    return fmaf(fmaf(-d, r, 1.0f), r, r);
}

// The softmax numerator exp(v - max) for a tap volume of type WT (CSPN_ours.py:35).  fp32 weights: the two-piece form above.
// fp16 weights: one multiply + v_exp_f32 — the argument's error (<= 24 * 1.4e-7 for every result that does not underflow in
// fp16) moves the numerator by ~2e-6 relative, 200x below half an fp16 ulp, and the derive of the weight-resident K x K
// launch (cspnk_resident.hip: 24 exponentials per pixel over tile + halo) is VALU-bound on exactly these operations.
// Every producer of fp16 softmax weights goes through this one function, so they all agree bit for bit.
// half(a * b) with ONE rounding (v_fma_mixlo_f16 / v_fma_mixhi_f16 round the exact product to fp16, and write the low / high
// half of `dst` leaving the other half alone): how every producer of fp16 softmax weights forms numerator * 1/sum.  Left to
// the compiler, `__float2half_rn(e * inv)` is fused into this instruction in some kernels and stays a multiply + a conversion
// (two roundings) in others — 6e-5 of the weights then differ by an fp16 ulp between kernels that must agree bit for bit.
__device__ __forceinline__ unsigned mul_into_half_lo(unsigned dst, float a, float b) {
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(dst) : "v"(a), "v"(b));
    return dst;
}
__device__ __forceinline__ unsigned mul_into_half_hi(unsigned dst, float a, float b) {
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(dst) : "v"(a), "v"(b));
    return dst;
}
// numerator * (1 / sum) as the value the tap volume of type WT will hold
template <typename WT> __device__ __forceinline__ float softmax_weight(float e, float inv) { return e * inv; }
template <> __device__ __forceinline__ float softmax_weight<__half>(float e, float inv) {
    const unsigned r = mul_into_half_lo(0u, e, inv);
    return __half2float(__ushort_as_half((unsigned short)(r & 0xffffu)));
}

template <typename WT> __device__ __forceinline__ float softmax_exp(float d) { return exp_nonpositive(d); }
template <> __device__ __forceinline__ float softmax_exp<__half>(float d) {
    return __builtin_amdgcn_exp2f(d * 1.44269502162933349609375f);
}

// The 10 masked terms of Result.evaluate (libs/metrics.py:49-83) for one pixel, added to f[0..9]:
// {inv^2, inv, diff^2, diff, diff/t, |log10 o - log10 t|, #(r<1.25), #(r<1.25^2), #(r<1.25^3), 1} over t > 0.
// Algebraically equal forms that avoid cancellation and redundant divisions:
//   |1/o - 1/t| = |o-t| / |o t|,   |log10 o - log10 t| = |log10(o/t)| = |log2(o/t)| log10(2),
//   max(o/t, t/o) < c  <=>  |log2(o/t)| < log2(c)  for o > 0;  o < 0 counts (both ratios are negative), o = 0 and NaN do not
//   (torch.max semantics) — the one logarithm of the log-error term serves the three thresholds.
// Reciprocals and the logarithm use the hardware v_rcp_f32 / v_log_f32 (1 ulp): the terms are summed over
// ~10^5..10^6 pixels and compared at 1e-5, and IEEE divisions + log10f made this reduction VALU-bound.
// Branch- and select-free masking: an invalid pixel (t <= 0 or NaN) is scored as o = t = 1, which adds exactly 0 to the six
// error sums, and the four counters take the validity as a condition.  ~30 VALU operations + 3 transcendentals per pixel;
// the form with six products and twelve compares per pixel (86 instructions with its masking) was 5.7 us of every scored
// config-2 forward.
__device__ __forceinline__ void metric_terms(float o, float t, float (&f)[10]) {
    const bool valid = t > 0.f;
    const float tt = valid ? t : 1.f, oo = valid ? o : 1.f;
    const float ad = fabsf(oo - tt);
    const float rt = __builtin_amdgcn_rcpf(tt);
    const float inv = ad * __builtin_amdgcn_rcpf(fabsf(oo * tt));
    f[0] = fmaf(inv, inv, f[0]);
    f[1] += inv;
    f[2] = fmaf(ad, ad, f[2]);
    f[3] += ad;
    f[4] = fmaf(ad, rt, f[4]);
    const float r = oo * rt;
    const float lg = fabsf(__builtin_amdgcn_logf(r));                       // |log2(o/t)|; NaN for o < 0 (as log10 of a negative)
    f[5] = fmaf(lg, 0.30102999566398120f, f[5]);
    const bool neg = r < 0.f;
    constexpr float L1 = 0.32192809488736235f;                             // log2(1.25)
    f[6] += (valid && (lg < L1 || neg)) ? 1.f : 0.f;
    f[7] += (valid && (lg < 2.f * L1 || neg)) ? 1.f : 0.f;
    f[8] += (valid && (lg < 3.f * L1 || neg)) ? 1.f : 0.f;
    f[9] += valid ? 1.f : 0.f;
}

// ------------------------------------------------------------------------------------------------
// Tap-volume layout (the [B, K*K-1, H, W] weight volume the propagation streams)
//   f32: planar, tap plane j of image b at ((b*NT + j)*HW + p).
//   f16: taps interleaved in PAIRS per 4-pixel quad — [B][NT/2][ceil(HW/4)][2][4] — so that one 16-byte load
//        returns taps (2i, 2i+1) of a quad.  8-byte loads run at about half the per-byte rate of 16-byte loads
//        on gfx950; with planar f16 planes the kernel was slower than its f32 twin.
// p = y*W + x is the linear pixel index inside an image; every kernel goes through Taps<WT>.
// ------------------------------------------------------------------------------------------------
template <typename WT> struct Taps;
template <> struct Taps<float> {
    __host__ __device__ static size_t image_elems(int NT, size_t HW) { return (size_t)NT * HW; }
    __device__ static size_t idx(int j, size_t p, size_t HW) { return (size_t)j * HW + p; }
};
template <> struct Taps<__half> {
    __host__ __device__ static size_t hw4(size_t HW) { return (HW + 3) & ~(size_t)3; }
    __host__ __device__ static size_t image_elems(int NT, size_t HW) { return (size_t)NT * hw4(HW); }
    __device__ static size_t idx(int j, size_t p, size_t HW) {
        return (size_t)(j >> 1) * 2 * hw4(HW) + ((p >> 2) << 3) + ((size_t)(j & 1) << 2) + (p & 3);
    }
};

#ifndef CSPN_DPP_BOUND_CTRL
#define CSPN_DPP_BOUND_CTRL true
#endif
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const volatile __attribute__((address_space(3))) v4f* lds_cv4f_ptr;   // LDS (addrspace 3) volatile b128

// Wavefront-level halo exchange: value held by lane-1 / lane+1 (DPP wave shift, VALU only).  Lanes without
// a source (0 / 63) or with an exec-masked source get 0 (bound_ctrl) and are patched from LDS by the caller.
__device__ __forceinline__ float dpp_from_prev_lane(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, CSPN_DPP_BOUND_CTRL));
}
__device__ __forceinline__ float dpp_from_next_lane(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, CSPN_DPP_BOUND_CTRL));
}

// Sum of v over the 64 lanes of a wavefront, valid in lane 63.  VALU only (DPP row shifts + the gfx9 row broadcasts):
// the __shfl_down ladder is 6 ds_bpermute_b32 per value and the LDS crossbar became the cost of the fused metrics
// reduction (10 values x 16 waves per workgroup).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add_step(float x) {
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v = dpp_add_step<0x111, 0xf>(v);      // row_shr:1
    v = dpp_add_step<0x112, 0xf>(v);      // row_shr:2
    v = dpp_add_step<0x114, 0xf>(v);      // row_shr:4
    v = dpp_add_step<0x118, 0xf>(v);      // row_shr:8   -> lane 15 of every row of 16 holds its row sum
    v = dpp_add_step<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v = dpp_add_step<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wavefront sum
    return v;
}

// The same for fp64 (the standalone metrics kernel accumulates arbitrarily many pixels per thread): DPP moves the two
// 32-bit halves, the add is a plain v_add_f64.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_step(double x) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, true);
    return x + __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum_to_lane63(double v) {
    v = dpp_add_step<0x111, 0xf>(v);
    v = dpp_add_step<0x112, 0xf>(v);
    v = dpp_add_step<0x114, 0xf>(v);
    v = dpp_add_step<0x118, 0xf>(v);
    v = dpp_add_step<0x142, 0xa>(v);
    v = dpp_add_step<0x143, 0xc>(v);
    return v;
}

// All NT taps of the quad starting at pixel p (p % 4 == 0) of one image's tap volume -> out[NT][4] (fp32).
template <int NT>
__device__ __forceinline__ void load_taps_quad(const float* img, size_t p, size_t HW, bool ok, float (&out)[NT][4]) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const float4 v = ok ? ld4(img + (size_t)j * HW + p) : make_float4(0.f, 0.f, 0.f, 0.f);
        out[j][0] = v.x; out[j][1] = v.y; out[j][2] = v.z; out[j][3] = v.w;
    }
}
template <int NT>
__device__ __forceinline__ void load_taps_quad(const __half* img, size_t p, size_t HW, bool ok, float (&out)[NT][4]) {
    const size_t pair_stride = 2 * Taps<__half>::hw4(HW);
#pragma unroll
    for (int jp = 0; jp < NT / 2; ++jp) {
        uint4 raw = make_uint4(0u, 0u, 0u, 0u);
        if (ok) raw = *reinterpret_cast<const uint4*>(img + (size_t)jp * pair_stride + 2 * p);
        const float2 a = __half22float2(*reinterpret_cast<const __half2*>(&raw.x));
        const float2 b = __half22float2(*reinterpret_cast<const __half2*>(&raw.y));
        const float2 c = __half22float2(*reinterpret_cast<const __half2*>(&raw.z));
        const float2 d = __half22float2(*reinterpret_cast<const __half2*>(&raw.w));
        out[2 * jp][0] = a.x; out[2 * jp][1] = a.y; out[2 * jp][2] = b.x; out[2 * jp][3] = b.y;
        out[2 * jp + 1][0] = c.x; out[2 * jp + 1][1] = c.y; out[2 * jp + 1][2] = d.x; out[2 * jp + 1][3] = d.y;
    }
}
// acc + w * x with w = element e of tap (2i + odd) taken straight from a packed pair register set (see the f16
// tap-volume layout: words x,y hold the even tap's pixels (0,1),(2,3), words z,w the odd tap's).  One
// v_fma_mix_f32 (f16 source selected by op_sel) — written as asm because, given C++ conversions, the optimiser
// hoists all of them out of the step loop and materialises the weights as fp32 again (197 instead of ~110 VGPRs).
__device__ __forceinline__ float fma_packed_tap(const uint4& r, int odd, int e, float x, float acc) {
    const unsigned w = odd ? ((e >> 1) ? r.w : r.z) : ((e >> 1) ? r.y : r.x);
    float out;
    if (e & 1) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(out) : "v"(w), "v"(x), "v"(acc));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(out) : "v"(w), "v"(x), "v"(acc));
    return out;
}

// acc + w * x with w = half e (0..7) of an OCT register set: r.x = pixels (0,1), r.y = (2,3), r.z = (4,5), r.w = (6,7) of one
// tap — the layout of a 16-byte load from a planar fp16 tap / kernel plane.  One v_fma_mix_f32 (see fma_packed_tap).
__device__ __forceinline__ float fma_h8(const uint4& r, int e, float x, float acc) {
    const unsigned w = (e >> 1) == 0 ? r.x : ((e >> 1) == 1 ? r.y : ((e >> 1) == 2 ? r.z : r.w));
    float out;
    if (e & 1) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(out) : "v"(w), "v"(x), "v"(acc));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(out) : "v"(w), "v"(x), "v"(acc));
    return out;
}

// the matching quad stores
template <int NT>
__device__ __forceinline__ void store_taps_quad(float* img, size_t p, size_t HW, const float (&v)[NT][4]) {
#pragma unroll
    for (int j = 0; j < NT; ++j) st4(img + (size_t)j * HW + p, make_float4(v[j][0], v[j][1], v[j][2], v[j][3]));
}
template <int NT>
__device__ __forceinline__ void store_taps_quad(__half* img, size_t p, size_t HW, const float (&v)[NT][4]) {
    const size_t pair_stride = 2 * Taps<__half>::hw4(HW);
#pragma unroll
    for (int jp = 0; jp < NT / 2; ++jp) {
        uint4 raw;
        *reinterpret_cast<__half2*>(&raw.x) = __floats2half2_rn(v[2 * jp][0], v[2 * jp][1]);
        *reinterpret_cast<__half2*>(&raw.y) = __floats2half2_rn(v[2 * jp][2], v[2 * jp][3]);
        *reinterpret_cast<__half2*>(&raw.z) = __floats2half2_rn(v[2 * jp + 1][0], v[2 * jp + 1][1]);
        *reinterpret_cast<__half2*>(&raw.w) = __floats2half2_rn(v[2 * jp + 1][2], v[2 * jp + 1][3]);
        *reinterpret_cast<uint4*>(img + (size_t)jp * pair_stride + 2 * p) = raw;
    }
}

// blockIdx -> logical tile id such that XCD x (= blockIdx % 8, observed dispatch order; speed only,
// never correctness) processes one contiguous range of tiles.
__device__ __forceinline__ int xcd_contiguous_id(int bid, int nb) {
    const int q = nb >> 3, r = nb & 7;
    const int x = bid & 7, j = bid >> 3;
    return x * q + (x < r ? x : r) + j;
}

// ------------------------------------------------------------------------------------------------
// small host helpers
// ------------------------------------------------------------------------------------------------
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int round_up4(int a) { return (a + 3) & ~3; }
inline size_t esize(int dt) { return dt == CSPN_F16 ? 2 : 4; }

int grid_for(size_t n, int block) {
    size_t g = (n + block - 1) / block;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}


}  // namespace
