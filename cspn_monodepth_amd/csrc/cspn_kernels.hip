// cspn_kernels.hip — CDNA4 (gfx950) kernels + C ABI of the CSPN affinity-propagation engine.
//
// The hot path is the recurrence  d_{t+1}[p] = blend( sum_j w_j[p] * d_t[p + off_j] )  over the
// K*K-1 non-centre taps of a K x K window (reference: network/libs/post_process/CSPN_new.py:80-90
// for K=3 with sum-normalised neighbour-indexed gates, CSPN_ours.py:47-53 + base/pac.py:89-92 for
// softmax-normalised centre-indexed taps).  It is a (K*K+1)*sizeof(T) bytes/pixel/step stream with
// ~0.4 flop/byte: HBM/L2-bandwidth bound, no MFMA.  Design:
//   * one launch = S consecutive propagation steps of one tile ("temporal blocking", S >= 1);
//   * each thread owns NQ vertically consecutive 4-pixel quads and keeps their K*K-1 weights in
//     VGPRs for all S steps (the weight volume is the only large stream: it is read once per launch
//     with 16-byte coalesced loads issued before anything else);
//   * the depth tile + halo lives in LDS (ping-pong), neighbours are exchanged through LDS between
//     the wavefronts of the workgroup, one barrier per step;
//   * blockIdx -> tile mapping is XCD-aware: every XCD (b % 8) walks a contiguous range of tiles, so
//     halo re-reads and the depth written by the previous launch hit that XCD's private L2.
// No fast-math: the reference's 0/0 = NaN semantics (CSPN_new.py:127) must survive.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <type_traits>

#include "cspn_hip.h"

namespace {

thread_local char g_err[512] = "";

int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 0;
}

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

// q[k] = a[k] / S (k < 8, 0 <= a[k] <= S), bit-identical to IEEE division: this IS the arithmetic of the compiler's
// fp32 division (v_div_scale, v_rcp, one Newton step on the reciprocal, q = a r, two residual corrections,
// v_div_fmas, v_div_fixup) with the part that depends only on the divisor shared by the eight quotients —
// 4 + 8*5 VALU operations instead of 8 * ~14.  The scale / fixup stages only act on extreme exponents and on
// inf / nan / 0 operands, so anything outside a comfortable normal range takes the plain division.
__device__ __forceinline__ void div8_shared_reciprocal(const float (&a)[8], float S, float (&q)[8]) {
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
}

// The 10 masked terms of Result.evaluate (libs/metrics.py:49-83) for one pixel, added to f[0..9]:
// {inv^2, inv, diff^2, diff, diff/t, |log10 o - log10 t|, #(r<1.25), #(r<1.25^2), #(r<1.25^3), 1} over t > 0.
// Algebraically equal forms that avoid cancellation and redundant divisions:
//   |1/o - 1/t| = |o-t| / |o t|,   |log10 o - log10 t| = |log10(o/t)|,
//   max(o/t, t/o) < c  <=>  o < c t  and  (o > 0 ? t < c o : o < 0)      (t > 0; NaN -> false as torch.max)
// Reciprocals and the logarithm use the hardware v_rcp_f32 / v_log_f32 (1 ulp): the terms are summed over
// ~10^5..10^6 pixels and compared at 1e-5, and IEEE divisions + log10f made this reduction VALU-bound.
__device__ __forceinline__ void metric_terms(float o, float t, float (&f)[10]) {
    if (!(t > 0.f)) return;
    const float ad = fabsf(o - t);
    const float rt = __builtin_amdgcn_rcpf(t);
    const float inv = ad * __builtin_amdgcn_rcpf(fabsf(o * t));
    f[0] = fmaf(inv, inv, f[0]);
    f[1] += inv;
    f[2] = fmaf(ad, ad, f[2]);
    f[3] += ad;
    f[4] = fmaf(ad, rt, f[4]);
    f[5] += fabsf(__builtin_amdgcn_logf(o * rt)) * 0.30102999566398120f;     // |log10(o/t)| = |log2(o/t)| log10(2)
    const float c1 = 1.25f, c2 = 1.25f * 1.25f, c3 = 1.25f * 1.25f * 1.25f;
    const bool pos = o > 0.f, neg = o < 0.f;
    f[6] += (o < c1 * t && (pos ? t < c1 * o : neg)) ? 1.f : 0.f;
    f[7] += (o < c2 * t && (pos ? t < c2 * o : neg)) ? 1.f : 0.f;
    f[8] += (o < c3 * t && (pos ? t < c3 * o : neg)) ? 1.f : 0.f;
    f[9] += 1.f;
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

typedef float v4f __attribute__((ext_vector_type(4)));
typedef const volatile __attribute__((address_space(3))) v4f* lds_cv4f_ptr;   // LDS (addrspace 3) volatile b128

// Wavefront-level halo exchange: value held by lane-1 / lane+1 (DPP wave shift, VALU only).  Lanes without
// a source (0 / 63) or with an exec-masked source get 0 and are patched from LDS by the caller.
__device__ __forceinline__ float dpp_from_prev_lane(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_next_lane(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
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
// the fused propagation kernel
// ------------------------------------------------------------------------------------------------
struct PropArgs {
    const void* w;       // [B,NT,H,W] tap planes, or (WSRC=1) the guidance tensor itself
    long g_bs, g_cs;     // WSRC=1: guidance batch / channel strides in elements
    void* w_out;         // WSRC=1: optional tap volume receiving the derived weights of the interior quads
    const void* target;  // SCORE=1: ground-truth depth [B,H,W] (DT) scored against the final state
    double* macc;        // SCORE=1: [nslots][10] metric accumulators (cspn_metrics_accumulate layout)
    int nslots;
    const void* d_in;    // [B,H,W]
    void* d_out;         // [B,H,W] state after the last fused step (may be null when hist != null)
    void* hist;          // null, or plane s-1 (stride B*H*W) receives the state after fused step s
    const void* sparse;  // [B,H,W] (blend 1, 2)
    const void* d0;      // [B,H,W] (blend 1)
    int B, H, W, S;
    int tw, th, tiles_x, tiles_y;
    int wq, wr;          // weight region: quad columns, rows
    int hxw, hyw;        // weight-region halo (pixels) left/right, top/bottom
    int dr, ls;          // depth region rows, LDS row stride (floats)
};

// Minimum waves per SIMD requested from the register allocator: the one-quad 3x3 instance is held to
// 64 VGPRs (8 waves/SIMD, i.e. 8/4/2 workgroups of 256/512/1024 threads per CU); the others take what they need.
template <int K, int NQ> struct MinWaves { static constexpr int value = (K == 3 && NQ == 1) ? 8 : 1; };

// WSRC = 0: weights are read from prepared tap planes.  WSRC = 1 (3x3 only): the launch derives them from the
// raw guidance itself — |g| of the 8 shifted channels, their sum, the IEEE divisions (exactly the arithmetic of
// cspn3_prepare_kernel, CSPN_new.py:29-70/:124-127) — so inference needs no prepare pass and never
// materialises the 8 weight planes (saves 53 MB written + 53 MB re-read per forward at config 2).
// SCORE = 1: the launch that produces the final state also accumulates the depth metrics of its interior pixels
// against `target` (the reduction cspn_metrics_kernel would do in a separate pass over the output).
template <int K, int NQ, int NTHREADS, typename WT, typename DT, int BLEND, int WSRC, int SCORE = 0>
__global__ __launch_bounds__(NTHREADS, (MinWaves<K, NQ>::value)) void cspn_prop_fused(const PropArgs a) {
    static_assert(WSRC == 0 || K == 3, "on-the-fly weights exist for the 3x3 variant only");
    constexpr int R = K / 2;
    constexpr int NT = K * K - 1;
    constexpr int WIN = 4 + 2 * R;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int tile = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int b = tile / tiles_per_img;
    const int trem = tile - b * tiles_per_img;
    const int ty = trem / a.tiles_x;
    const int tx = trem - ty * a.tiles_x;
    const int H = a.H, W = a.W;
    const int y0 = ty * a.th, x0 = tx * a.tw;
    const size_t HW = (size_t)H * W;
    const size_t plane = (size_t)a.B * HW;

    const WT* __restrict__ wg = static_cast<const WT*>(a.w) + (size_t)b * Taps<WT>::image_elems(NT, HW);
    const DT* __restrict__ din = static_cast<const DT*>(a.d_in) + (size_t)b * HW;
    const DT* __restrict__ spg = BLEND ? static_cast<const DT*>(a.sparse) + (size_t)b * HW : nullptr;

    // ---- ownership: strip (sx, sy) = NQ vertically consecutive quads of the weight region -------
    const int wq = a.wq, wr = a.wr;
    const int sy = tid / wq;
    const int sx = tid - sy * wq;
    const int r0 = sy * NQ;                    // first weight-region row of this strip
    const int xq = x0 - a.hxw + 4 * sx;        // image x of the quad
    const int yq0 = y0 - a.hyw + r0;           // image y of the first quad
    const bool x_in = (xq >= 0) && (xq < W);   // W % 4 == 0: a quad is fully inside or outside
    const int lane = tid & 63;
    const bool fix_left = (sx == 0) || (lane == 0);          // left neighbour quad is not lane-1's
    const bool fix_right = (sx == wq - 1) || (lane == 63);   // right neighbour quad is not lane+1's

    // ---- 1. issue the weight stream first (independent of LDS): NQ x NT 16-byte loads -----------
    float wreg[NQ][NT][4];
    unsigned in_img = 0, interior = 0;
    // Blend operands of the owned quads, om = 1 - m and md0 = m * d0 (m = sign(sparse); both products are exact),
    // are parked in two private LDS planes instead of 8 VGPRs per quad: each thread only ever touches its
    // own slots, so no barrier is involved, and the one-quad instances stay within 64 VGPRs.
    float* const om_lds = lds + (size_t)2 * a.dr * a.ls;
    float* const md_lds = om_lds + (size_t)a.wr * 4 * a.wq;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int r = r0 + i, y = yq0 + i;
        const bool ok = (r < wr) && x_in && (y >= 0) && (y < H);
        if (ok) in_img |= 1u << i;
        if (ok && r >= a.hyw && r < a.hyw + a.th && xq >= x0 && xq < x0 + a.tw) interior |= 1u << i;
        const size_t off = (size_t)(ok ? y : 0) * W + (ok ? xq : 0);
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (WSRC == 0) {
            load_taps_quad<NT>(wg, off, HW, ok, wreg[i]);
        } else {
            // tap j = (dy,dx) row-major without the centre reads channel 7-j at p+off_j.  The aligned quad of
            // row y+dy gives three of the four shifted values, the fourth is the neighbouring lane's quad
            // (DPP wave shift) or, at strip ends, one scalar load.
            const WT* __restrict__ gq = static_cast<const WT*>(a.w) + (size_t)b * a.g_bs;
            float left[NT], right[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int lin = j < 4 ? j : j + 1;
                const int dy = lin / 3 - 1, dx = lin % 3 - 1;
                const int row = y + dy;
                const bool rok = ok && row >= 0 && row < H;
                const WT* src = gq + (size_t)(7 - j) * a.g_cs + (size_t)(rok ? row : 0) * W;
                const float4 v = rok ? ld4(src + xq) : z4;
                wreg[i][j][0] = v.x; wreg[i][j][1] = v.y; wreg[i][j][2] = v.z; wreg[i][j][3] = v.w;
                left[j] = 0.f; right[j] = 0.f;
                if (dx < 0) left[j] = dpp_from_prev_lane(v.w);
                if (dx > 0) right[j] = dpp_from_next_lane(v.x);
            }
            if (fix_left) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int lin = j < 4 ? j : j + 1;
                    const int dy = lin / 3 - 1, dx = lin % 3 - 1;
                    if (dx < 0) {
                        const int row = y + dy;
                        const bool c = ok && row >= 0 && row < H && xq >= 1;
                        left[j] = c ? ld1(gq + (size_t)(7 - j) * a.g_cs + (size_t)row * W + xq - 1) : 0.f;
                    }
                }
            }
            if (fix_right) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int lin = j < 4 ? j : j + 1;
                    const int dy = lin / 3 - 1, dx = lin % 3 - 1;
                    if (dx > 0) {
                        const int row = y + dy;
                        const bool c = ok && row >= 0 && row < H && xq + 4 < W;
                        right[j] = c ? ld1(gq + (size_t)(7 - j) * a.g_cs + (size_t)row * W + xq + 4) : 0.f;
                    }
                }
            }
            // a_j[e] = |g_{7-j}[p_e + off_j]|
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int lin = j < 4 ? j : j + 1;
                const int dx = lin % 3 - 1;
                const float q0 = wreg[i][j][0], q1 = wreg[i][j][1], q2 = wreg[i][j][2], q3 = wreg[i][j][3];
                if (dx < 0) { wreg[i][j][0] = fabsf(left[j]); wreg[i][j][1] = fabsf(q0); wreg[i][j][2] = fabsf(q1); wreg[i][j][3] = fabsf(q2); }
                else if (dx > 0) { wreg[i][j][0] = fabsf(q1); wreg[i][j][1] = fabsf(q2); wreg[i][j][2] = fabsf(q3); wreg[i][j][3] = fabsf(right[j]); }
                else { wreg[i][j][0] = fabsf(q0); wreg[i][j][1] = fabsf(q1); wreg[i][j][2] = fabsf(q2); wreg[i][j][3] = fabsf(q3); }
            }
            // S in the reference's channel order k = 0..7 (tap 7..0), then true division; 0 for padding quads
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float S = wreg[i][7][e];
#pragma unroll
                for (int k = 1; k < 8; ++k) S += wreg[i][7 - k][e];
                float av[8], qv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) av[j] = wreg[i][j][e];
                div8_shared_reciprocal(av, S, qv);
#pragma unroll
                for (int j = 0; j < NT; ++j) wreg[i][j][e] = ok ? qv[j] : 0.f;
            }
            // the launch that derives the weights can also publish them (tap-volume layout) for the launches
            // that follow, which then stream them like a prepared volume
            if (a.w_out && ((interior >> i) & 1u)) {
                WT* wo = static_cast<WT*>(a.w_out) + (size_t)b * Taps<WT>::image_elems(NT, HW);
                store_taps_quad<NT>(wo, off, HW, wreg[i]);
            }
        }
        if (BLEND && r < wr) {
            const float4 m = ok ? sgn4(ld4(spg + off)) : z4;
            const int qoff = (r * wq + sx) * 4;
            *reinterpret_cast<float4*>(om_lds + qoff) = make_float4(1.f - m.x, 1.f - m.y, 1.f - m.z, 1.f - m.w);
            if (BLEND == CSPN_BLEND_SPARSE) {
                const float4 v = ok ? ld4(static_cast<const DT*>(a.d0) + (size_t)b * HW + off) : z4;
                *reinterpret_cast<float4*>(md_lds + qoff) = make_float4(m.x * v.x, m.y * v.y, m.z * v.z, m.w * v.w);
            }
        }
    }

    // ---- 2. stage the depth region (weight region + R halo) into LDS ----------------------------
    const int dr = a.dr, ls = a.ls;
    float* cur = lds;
    float* nxt = lds + (size_t)dr * ls;
    const int yd0 = y0 - a.hyw - R;            // image y of depth-region row 0
    const int xd0 = x0 - a.hxw - 4;            // image x of LDS column 0
    for (int idx = tid; idx < dr * wq; idx += NTHREADS) {
        const int row = idx / wq, qx = idx - row * wq;
        const int y = yd0 + row, x = xd0 + 4 + 4 * qx;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y >= 0 && y < H && x >= 0 && x < W) {
            v = ld4(din + (size_t)y * W + x);
            if (BLEND == CSPN_BLEND_PREMASK) {
                const float4 m = sgn4(ld4(spg + (size_t)y * W + x));
                v.x *= 1.f - m.x; v.y *= 1.f - m.y; v.z *= 1.f - m.z; v.w *= 1.f - m.w;
            }
        }
        *reinterpret_cast<float4*>(&cur[row * ls + 4 + 4 * qx]) = v;
    }
    for (int idx = tid; idx < dr * 2 * R; idx += NTHREADS) {
        const int row = idx / (2 * R), c = idx - row * (2 * R);
        const int lc = (c < R) ? (4 - R + c) : (4 + 4 * wq + (c - R));
        const int y = yd0 + row, x = xd0 + lc;
        float v = 0.f;
        if (y >= 0 && y < H && x >= 0 && x < W) {
            v = ld1(din + (size_t)y * W + x);
            if (BLEND == CSPN_BLEND_PREMASK) v *= 1.f - sgnf(ld1(spg + (size_t)y * W + x));
        }
        cur[row * ls + lc] = v;
        nxt[row * ls + lc] = 0.f;   // the outer halo ring of the second buffer is never computed
    }
    __syncthreads();

    // ---- 3. S propagation steps in LDS ------------------------------------------------------------
    const bool active = (r0 < wr);
    const int cb = 4 + 4 * sx;
    DT* __restrict__ dout = a.d_out ? static_cast<DT*>(a.d_out) + (size_t)b * HW : nullptr;
    DT* __restrict__ hist = a.hist ? static_cast<DT*>(a.hist) + (size_t)b * HW : nullptr;

    float mf[10];
    if (SCORE) {
#pragma unroll
        for (int k = 0; k < 10; ++k) mf[k] = 0.f;
    }
    for (int s = 1; s <= a.S; ++s) {
        const bool last = (s == a.S);
        if (active) {
            // Window fetch.  One aligned ds_read_b128 per row gives the thread's own 4 pixels; the R pixels
            // to the left / right are the neighbouring lanes' quads, taken with DPP wave shifts (no LDS
            // traffic, no bank conflicts).  Only the lanes at the ends of a strip row (and wave lanes 0 / 63)
            // fetch their halo from LDS, in two exec-masked blocks.
            float win[NQ + 2 * R][WIN];
            // rows r0 .. r0+NQ-1+2R of the depth region; only quads below the weight region (NQ > 1, never
            // computed) can point past its last row, so clamp those.
            auto row_ptr = [&](int rr) -> const float* {
                int drow = r0 + rr;
                if (NQ > 1) drow = drow < dr ? drow : dr - 1;
                return cur + drow * ls + cb;
            };
#pragma unroll
            for (int rr = 0; rr < NQ + 2 * R; ++rr) {
                // volatile: keep this ONE ds_read_b128.  Left alone, the optimiser re-loads overlapping
                // dword pairs from LDS (bank-conflicted ds_read2_b32) to feed v_pk_fma_f32 operand pairs.
                const v4f mid = *(lds_cv4f_ptr)(row_ptr(rr));
                const float m4[4] = {mid.x, mid.y, mid.z, mid.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) win[rr][R + c] = m4[c];
#pragma unroll
                for (int c = 0; c < R; ++c) {
                    win[rr][c] = dpp_from_prev_lane(m4[4 - R + c]);
                    win[rr][R + 4 + c] = dpp_from_next_lane(m4[c]);
                }
            }
            if (fix_left) {
#pragma unroll
                for (int rr = 0; rr < NQ + 2 * R; ++rr)
#pragma unroll
                    for (int c = 0; c < R; ++c) win[rr][c] = row_ptr(rr)[c - R];
            }
            if (fix_right) {
#pragma unroll
                for (int rr = 0; rr < NQ + 2 * R; ++rr)
#pragma unroll
                    for (int c = 0; c < R; ++c) win[rr][R + 4 + c] = row_ptr(rr)[4 + c];
            }
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                if (r0 + i < wr) {
                    float u[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int dy = -R; dy <= R; ++dy)
#pragma unroll
                        for (int dx = -R; dx <= R; ++dx) {
                            if (dy == 0 && dx == 0) continue;
                            const int lin = (dy + R) * K + (dx + R);
                            const int j = lin < (K * K) / 2 ? lin : lin - 1;
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                u[e] = fmaf(wreg[i][j][e], win[i + dy + R][e + dx + R], u[e]);
                        }
                    float keep[4];   // value carried to the next step through LDS
                    float om[4] = {1.f, 1.f, 1.f, 1.f}, md[4] = {0.f, 0.f, 0.f, 0.f};
                    if (BLEND) {
                        const int qoff = ((r0 + i) * wq + sx) * 4;
                        const float4 o4 = *reinterpret_cast<const float4*>(om_lds + qoff);
                        om[0] = o4.x; om[1] = o4.y; om[2] = o4.z; om[3] = o4.w;
                        if (BLEND == CSPN_BLEND_SPARSE) {
                            const float4 m4 = *reinterpret_cast<const float4*>(md_lds + qoff);
                            md[0] = m4.x; md[1] = m4.y; md[2] = m4.z; md[3] = m4.w;
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (BLEND == CSPN_BLEND_SPARSE) {
                            u[e] = om[e] * u[e] + md[e];          // (1-m) u + m d0     CSPN_new.py:90
                            keep[e] = u[e];
                        } else if (BLEND == CSPN_BLEND_PREMASK) {
                            keep[e] = om[e] * u[e];
                        } else {
                            keep[e] = u[e];
                        }
                        if (!((in_img >> i) & 1u)) { u[e] = 0.f; keep[e] = 0.f; }   // zero padding stays exactly zero
                    }
                    if (!last)
                        *reinterpret_cast<float4*>(&nxt[(r0 + i + R) * ls + cb]) =
                            make_float4(keep[0], keep[1], keep[2], keep[3]);
                    if ((interior >> i) & 1u) {
                        const size_t off = (size_t)(yq0 + i) * W + xq;
                        const float4 uv = make_float4(u[0], u[1], u[2], u[3]);
                        if (hist) st4(hist + (size_t)(s - 1) * plane + off, uv);
                        else if (last) st4(dout + off, uv);
                        if (SCORE && last) {
                            const float4 tg = ld4(static_cast<const DT*>(a.target) + (size_t)b * HW + off);
                            const float t4[4] = {tg.x, tg.y, tg.z, tg.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float o = u[e];
                                if (sizeof(DT) == 2) o = __half2float(__float2half_rn(o));   // score the stored value
                                metric_terms(o, t4[e], mf);
                            }
                        }
                    }
                }
            }
        }
        if (!last) {
            __syncthreads();
            float* t = cur; cur = nxt; nxt = t;
        }
    }
    if (SCORE) {
        // fp32 wave reduction (<= 256 pixels per wave), fp64 across the waves, 10 atomics per workgroup
        float* part = lds + (size_t)2 * a.dr * a.ls + (size_t)(BLEND == CSPN_BLEND_SPARSE ? 2 : (BLEND ? 1 : 0)) * a.wr * 4 * a.wq;
        const int wave = tid >> 6;
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            float v = mf[k];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            if (lane == 0) part[wave * 10 + k] = v;
        }
        __syncthreads();
        if (tid < 10) {
            double v = 0.0;
            for (int w = 0; w < NTHREADS / 64; ++w) v += (double)part[w * 10 + tid];
            if (v != 0.0) atomicAdd(a.macc + (size_t)(blockIdx.x % a.nslots) * 10 + tid, v);
        }
    }
}

// Generic one-pixel-per-thread step (any W, any alignment, S = 1).  Correctness path for shapes the
// vector kernel cannot take (W % 4 != 0); not the tuned path.
template <int K, typename WT, typename DT, int BLEND>
__global__ void cspn_prop_scalar(const void* w_, const void* din_, void* dout_, const void* sp_,
                                 const void* d0_, int B, int H, int W) {
    constexpr int R = K / 2;
    constexpr int NT = K * K - 1;
    const size_t HW = (size_t)H * W;
    const size_t total = (size_t)B * HW;
    const WT* w = static_cast<const WT*>(w_);
    const DT* din = static_cast<const DT*>(din_);
    const DT* sp = static_cast<const DT*>(sp_);
    const DT* d0 = static_cast<const DT*>(d0_);
    DT* dout = static_cast<DT*>(dout_);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / HW);
        const int p = (int)(i - (size_t)b * HW);
        const int y = p / W, x = p - y * W;
        float u = 0.f;
        int j = 0;
        for (int dy = -R; dy <= R; ++dy)
            for (int dx = -R; dx <= R; ++dx) {
                if (dy == 0 && dx == 0) continue;
                const int yy = y + dy, xx = x + dx;
                float dv = 0.f;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                    dv = ld1(din + (size_t)b * HW + (size_t)yy * W + xx);
                    if (BLEND == CSPN_BLEND_PREMASK) dv *= 1.f - sgnf(ld1(sp + (size_t)b * HW + (size_t)yy * W + xx));
                }
                u = fmaf(ld1(w + (size_t)b * Taps<WT>::image_elems(NT, HW) + Taps<WT>::idx(j, p, HW)), dv, u);
                ++j;
            }
        if (BLEND == CSPN_BLEND_SPARSE) {
            const float m = sgnf(ld1(sp + i));
            u = (1.f - m) * u + m * ld1(d0 + i);
        }
        st1(dout + i, u);
    }
}

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

// ------------------------------------------------------------------------------------------------
// backward helpers
// ------------------------------------------------------------------------------------------------
// gw_j[p] = (1-m) sum_t G_{t+1}[p] d_t[p+off_j];  gd0[p] = G_0[p] + m sum_{t>=1} G_t[p]
template <int K, typename DT>
__global__ void cspn_grad_weights_kernel(const DT* __restrict__ d0, const DT* __restrict__ dhist,
                                         const float* __restrict__ ghist, const DT* __restrict__ sparse,
                                         float* __restrict__ gw, float* __restrict__ gd0,
                                         int B, int H, int W, int T) {
    constexpr int R = K / 2;
    constexpr int NT = K * K - 1;
    const size_t HW = (size_t)H * W;
    const size_t total = (size_t)B * HW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / HW);
        const int p = (int)(i - (size_t)b * HW);
        const int y = p / W, x = p - y * W;
        float acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = 0.f;
        float gsum = 0.f;
        for (int t = 0; t < T; ++t) {
            const DT* d = (t == 0) ? d0 : dhist + (size_t)(t - 1) * total;
            const float G = ghist[(size_t)(T - 1 - t) * total + i];   // G_{t+1}
            gsum += G;
            int j = 0;
#pragma unroll
            for (int dy = -R; dy <= R; ++dy)
#pragma unroll
                for (int dx = -R; dx <= R; ++dx) {
                    if (dy == 0 && dx == 0) continue;
                    const int yy = y + dy, xx = x + dx;
                    float dv = 0.f;
                    if (yy >= 0 && yy < H && xx >= 0 && xx < W) dv = ld1(d + (size_t)b * HW + (size_t)yy * W + xx);
                    acc[j] = fmaf(G, dv, acc[j]);
                    ++j;
                }
        }
        const float m = sparse ? sgnf(ld1(sparse + i)) : 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) gw[((size_t)b * NT + j) * HW + p] = (1.f - m) * acc[j];
        gd0[i] = ghist[(size_t)T * total + i] + m * gsum;        // G_0 + m sum_{t>=1} G_t
    }
}

// dL/dg_{7-j}[q] = sign(g) * gA_j[q - off_j],  gA_j = (gw_j - sum_k gw_k w_k) / S
template <typename GT, typename WT>
__global__ void cspn3_grad_guidance_kernel(const GT* __restrict__ g, long bs, long cs, int C,
                                           const WT* __restrict__ w8, const float* __restrict__ S,
                                           const float* __restrict__ gw, GT* __restrict__ gg,
                                           int B, int H, int W) {
    const size_t HW = (size_t)H * W;
    const size_t total = (size_t)B * HW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / HW);
        const int q = (int)(i - (size_t)b * HW);
        const int y = q / W, x = q - y * W;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int lin = j < 4 ? j : j + 1;
            const int dy = lin / 3 - 1, dx = lin % 3 - 1;
            const int yy = y - dy, xx = x - dx;       // p = q - off_j
            float val = 0.f;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                const size_t p = (size_t)yy * W + xx;
                float dot = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    dot = fmaf(gw[((size_t)b * 8 + k) * HW + p],
                               ld1(w8 + (size_t)b * Taps<WT>::image_elems(8, HW) + Taps<WT>::idx(k, p, HW)), dot);
                const float gA = (gw[((size_t)b * 8 + j) * HW + p] - dot) / S[(size_t)b * HW + p];
                val = sgnf(ld1(g + (size_t)b * bs + (size_t)(7 - j) * cs + q)) * gA;
            }
            st1(gg + (size_t)b * bs + (size_t)(7 - j) * cs + q, val);
        }
        for (int c = 8; c < C; ++c) st1(gg + (size_t)b * bs + (size_t)c * cs + q, 0.f);
    }
}

template <int K, typename WT, typename GT>
__global__ void cspn_pac_grad_guided_kernel(const WT* __restrict__ wk, const float* __restrict__ gw,
                                            GT* __restrict__ gg, int B, int H, int W) {
    constexpr int NT = K * K - 1;
    const size_t HW = (size_t)H * W;
    const size_t total = (size_t)B * HW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / HW);
        const size_t p = i - (size_t)b * HW;
        float sm[NT], gv[NT];
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            sm[c] = ld1(wk + (size_t)b * Taps<WT>::image_elems(NT, HW) + Taps<WT>::idx(c, p, HW));
            gv[c] = gw[((size_t)b * NT + c) * HW + p];
            dot = fmaf(sm[c], gv[c], dot);
        }
#pragma unroll
        for (int c = 0; c < NT; ++c) st1(gg + ((size_t)b * NT + c) * HW + p, sm[c] * (gv[c] - dot));
    }
}

// Fused backward tail (vector path, W % 4 == 0).  One thread owns a 4-pixel quad p and, in ONE pass over the two
// histories (8 B/px/step of HBM traffic), accumulates
//     acc_j[p] = sum_t G_{t+1}[p] * d_t[p+off_j]        (dL/dw_j up to the (1-m) factor)
//     gsum[p]  = sum_t G_{t+1}[p]
// in registers: per step one aligned 16-byte load of G and one per window row of d, horizontal neighbours by DPP
// wave shifts (strip-end lanes patch with scalar loads).  The epilogue then never writes dL/dw at all:
//   VARIANT 1 (3x3):  gA_j = ((1-m) acc_j - dot) / S, dot = sum_k (1-m) acc_k w_k   (quotient rule of w = A/S)
//                     dL/dg_{7-j}[p+off_j] = sign(g) * gA_j[p]   scattered from the source side; targets
//                     without a source (image border) and channels >= 8 are zero-filled here as well.
//   VARIANT 2 (KxK):  dL/dguided_c = sm_c ((1-m) acc_c - sum_k (1-m) acc_k sm_k)    (softmax backward)
//   VARIANT 0:        dL/dw_j = (1-m) acc_j   (raw, for callers that want it)
// and gd0 = G_0 + m * gsum for all variants.
struct TailArgs {
    const void* d0;       // [B,H,W]   DT
    const void* dhist;    // [T,B,H,W] DT  (d_1..d_T)
    const float* ghist;   // [T+1,B,H,W] backward order: ghist[s] = G_{T-s}
    const void* sparse;   // [B,H,W] DT or null
    const void* w;        // [B,NT,H,W] WT tap planes (variants 1, 2)
    const float* S;       // [B,H,W] (variant 1)
    const void* guidance; // variant 1: [B,C,H,W] WT through strides
    void* gout;           // variant 0: gw f32 [B,NT,H,W]; 1: grad_guidance WT (guidance strides); 2: grad_guided WT
    float* gd0;           // [B,H,W]
    long g_bs, g_cs;
    int B, H, W, T, C;
};

template <int K, typename DT, typename WT, int VARIANT>
__global__ __launch_bounds__(256) void cspn_grad_tail(const TailArgs a) {
    constexpr int R = K / 2;
    constexpr int NT = K * K - 1;
    constexpr int WIN = 4 + 2 * R;
    const int H = a.H, W = a.W, T = a.T;
    const int WQ = W >> 2;
    const size_t HW = (size_t)H * W;
    const size_t plane = (size_t)a.B * HW;
    const size_t nquads = (size_t)a.B * H * WQ;
    // XCD-contiguous block order: vertically adjacent rows (shared window rows) stay within one XCD's L2
    const size_t q = (size_t)xcd_contiguous_id(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    const bool live = q < nquads;
    const size_t qq = live ? q : 0;
    const int b = (int)(qq / ((size_t)H * WQ));
    const int rem = (int)(qq - (size_t)b * H * WQ);
    const int y = rem / WQ, qx = rem - y * WQ, x = qx * 4;
    const int lane = threadIdx.x & 63;
    const bool fix_left = (qx == 0) || (lane == 0);
    const bool fix_right = (qx == WQ - 1) || (lane == 63);
    const size_t off = (size_t)b * HW + (size_t)y * W + x;
    const DT* d0 = static_cast<const DT*>(a.d0);
    const DT* dh = static_cast<const DT*>(a.dhist);

    float acc[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][e] = 0.f;
    float gsum[4] = {0.f, 0.f, 0.f, 0.f};

    // Software-pipelined stream over the histories: the loads of UNR steps (G quad, 2R+1 row quads, and the
    // strip-end lanes' scalar halo patches) are all issued before the first one is consumed; otherwise every
    // step (and every patch load) is a serialised HBM/L2 round trip and the pass is latency-bound.
    constexpr int UNR = (K == 3) ? 4 : (K == 5 ? 2 : 1);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t0 = 0; t0 < T; t0 += UNR) {
        float4 Gq[UNR], midq[UNR][2 * R + 1];
        float lf[UNR][2 * R + 1][R], rf[UNR][2 * R + 1][R];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int t = t0 + u;
            const bool tv = live && t < T;
            const DT* d = (t == 0) ? d0 : dh + (size_t)(tv ? t - 1 : 0) * plane;
            Gq[u] = tv ? ld4(a.ghist + (size_t)(T - 1 - t) * plane + off) : z4;
#pragma unroll
            for (int rr = 0; rr < 2 * R + 1; ++rr) {
                const int row = y + rr - R;
                const bool rok = tv && row >= 0 && row < H;
                const DT* rp = d + (size_t)b * HW + (size_t)(rok ? row : 0) * W;
                midq[u][rr] = rok ? ld4(rp + x) : z4;
#pragma unroll
                for (int c = 0; c < R; ++c) { lf[u][rr][c] = 0.f; rf[u][rr][c] = 0.f; }
                if (fix_left) {
#pragma unroll
                    for (int c = 0; c < R; ++c) {
                        const int xx = x + c - R;
                        lf[u][rr][c] = (rok && xx >= 0) ? ld1(rp + xx) : 0.f;
                    }
                }
                if (fix_right) {
#pragma unroll
                    for (int c = 0; c < R; ++c) {
                        const int xx = x + 4 + c;
                        rf[u][rr][c] = (rok && xx < W) ? ld1(rp + xx) : 0.f;
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const float g4[4] = {Gq[u].x, Gq[u].y, Gq[u].z, Gq[u].w};
            float win[2 * R + 1][WIN];
#pragma unroll
            for (int rr = 0; rr < 2 * R + 1; ++rr) {
                const float m4[4] = {midq[u][rr].x, midq[u][rr].y, midq[u][rr].z, midq[u][rr].w};
#pragma unroll
                for (int c = 0; c < 4; ++c) win[rr][R + c] = m4[c];
#pragma unroll
                for (int c = 0; c < R; ++c) {
                    const float l = dpp_from_prev_lane(m4[4 - R + c]);
                    const float r = dpp_from_next_lane(m4[c]);
                    win[rr][c] = fix_left ? lf[u][rr][c] : l;
                    win[rr][R + 4 + c] = fix_right ? rf[u][rr][c] : r;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) gsum[e] += g4[e];
#pragma unroll
            for (int dy = -R; dy <= R; ++dy)
#pragma unroll
                for (int dx = -R; dx <= R; ++dx) {
                    if (dy == 0 && dx == 0) continue;
                    const int lin = (dy + R) * K + (dx + R);
                    const int j = lin < (K * K) / 2 ? lin : lin - 1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j][e] = fmaf(g4[e], win[dy + R][e + dx + R], acc[j][e]);
                }
        }
    }
    if (!live) return;

    float om[4] = {1.f, 1.f, 1.f, 1.f}, mm[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.sparse) {
        const float4 sp = sgn4(ld4(static_cast<const DT*>(a.sparse) + off));
        mm[0] = sp.x; mm[1] = sp.y; mm[2] = sp.z; mm[3] = sp.w;
#pragma unroll
        for (int e = 0; e < 4; ++e) om[e] = 1.f - mm[e];
    }
    {
        const float4 G0 = ld4(a.ghist + (size_t)T * plane + off);
        st4(a.gd0 + off, make_float4(G0.x + mm[0] * gsum[0], G0.y + mm[1] * gsum[1], G0.z + mm[2] * gsum[2],
                                     G0.w + mm[3] * gsum[3]));
    }
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][e] *= om[e];

    if constexpr (VARIANT == 0) {
        float* gw = static_cast<float*>(a.gout) + (size_t)b * NT * HW + (size_t)y * W + x;
#pragma unroll
        for (int j = 0; j < NT; ++j) st4(gw + (size_t)j * HW, make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]));
    } else if constexpr (VARIANT == 2) {
        const WT* wk = static_cast<const WT*>(a.w) + (size_t)b * Taps<WT>::image_elems(NT, HW);
        WT* gg = static_cast<WT*>(a.gout) + (size_t)b * NT * HW + (size_t)y * W + x;   // plain [B,NT,H,W] gradient
        float dot[4] = {0.f, 0.f, 0.f, 0.f};
        float sm[NT][4];
        load_taps_quad<NT>(wk, (size_t)y * W + x, HW, true, sm);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) dot[e] = fmaf(sm[j][e], acc[j][e], dot[e]);
#pragma unroll
        for (int j = 0; j < NT; ++j)
            st4(gg + (size_t)j * HW, make_float4(sm[j][0] * (acc[j][0] - dot[0]), sm[j][1] * (acc[j][1] - dot[1]),
                                                 sm[j][2] * (acc[j][2] - dot[2]), sm[j][3] * (acc[j][3] - dot[3])));
    } else {
        static_assert(VARIANT != 1 || K == 3, "guidance epilogue is the 3x3 variant");
        const WT* w8 = static_cast<const WT*>(a.w) + (size_t)b * Taps<WT>::image_elems(8, HW);
        const WT* g = static_cast<const WT*>(a.guidance) + (size_t)b * a.g_bs;
        WT* gg = static_cast<WT*>(a.gout) + (size_t)b * a.g_bs;
        float dot[4] = {0.f, 0.f, 0.f, 0.f};
        {
            float w4[8][4];
            load_taps_quad<8>(w8, (size_t)y * W + x, HW, true, w4);
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) dot[e] = fmaf(acc[j][e], w4[j][e], dot[e]);
        }
        const float4 Sv = ld4(a.S + off);
        const float S4[4] = {Sv.x, Sv.y, Sv.z, Sv.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int lin = j < 4 ? j : j + 1;
            const int dy = lin / 3 - 1, dx = lin % 3 - 1;
            const size_t cplane = (size_t)(7 - j) * a.g_cs;
            const int ty = y + dy;
            float gA[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) gA[e] = (acc[j][e] - dot[e]) / S4[e];
            // Scatter gA_j[p] to p + off_j as ALIGNED quads of the target row ty: the target quad [x, x+4) takes
            // source columns [x-dx, x+4-dx), i.e. three own values and one from the neighbouring lane (DPP).
            // Where that neighbour sits in another wave (lane 0 / 63 inside a row) the element is written by its
            // owner with one scalar store and skipped here; where it does not exist (image border) it is 0.
            const float from_prev = dpp_from_prev_lane(gA[3]);
            const float from_next = dpp_from_next_lane(gA[0]);
            if (ty >= 0 && ty < H) {
                const size_t o = cplane + (size_t)ty * W + x;
                const float4 gs = sgn4(ld4(g + o));
                if (dx == 0) {
                    st4(gg + o, make_float4(gs.x * gA[0], gs.y * gA[1], gs.z * gA[2], gs.w * gA[3]));
                } else if (dx > 0) {
                    const float v0 = (qx == 0) ? 0.f : from_prev;              // column 0 has no source
                    if (lane == 0 && qx > 0) {                                   // left neighbour lives in another wave
                        st1(gg + o + 1, gs.y * gA[0]); st1(gg + o + 2, gs.z * gA[1]); st1(gg + o + 3, gs.w * gA[2]);
                    } else {
                        st4(gg + o, make_float4(gs.x * v0, gs.y * gA[0], gs.z * gA[1], gs.w * gA[2]));
                    }
                    if (lane == 63 && qx < WQ - 1) st1(gg + o + 4, sgnf(ld1(g + o + 4)) * gA[3]);
                } else {
                    const float v3 = (qx == WQ - 1) ? 0.f : from_next;         // column W-1 has no source
                    if (lane == 63 && qx < WQ - 1) {
                        st1(gg + o, gs.x * gA[1]); st1(gg + o + 1, gs.y * gA[2]); st1(gg + o + 2, gs.z * gA[3]);
                    } else {
                        st4(gg + o, make_float4(gs.x * gA[1], gs.y * gA[2], gs.z * gA[3], gs.w * v3));
                    }
                    if (lane == 0 && qx > 0) st1(gg + o - 1, sgnf(ld1(g + o - 1)) * gA[0]);
                }
            }
            // rows of this plane that no source row reaches: row 0 (dy=+1) / row H-1 (dy=-1)
            if ((dy > 0 && y == 0) || (dy < 0 && y == H - 1))
                st4(gg + cplane + (size_t)y * W + x, make_float4(0.f, 0.f, 0.f, 0.f));
        }
        for (int c = 8; c < a.C; ++c) st4(gg + (size_t)c * a.g_cs + (size_t)y * W + x, make_float4(0.f, 0.f, 0.f, 0.f));
    }
}

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
    // wave64 shuffle reduction -> LDS -> one atomic per block and quantity (10 per block)
    __shared__ double part[16][10];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        double v = s[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0) part[wave][k] = v;
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

// ------------------------------------------------------------------------------------------------
// host side: plan selection and launches
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

struct Launch {
    PropArgs a;
    int grid, threads, nq;
    size_t lds_bytes;
};

// Geometry of one fused launch.  Returns false if (plan, S) does not fit the machine limits.
bool make_geometry(int K, int B, int H, int W, int S, int tw, int th, int nq, int threads, int blend, Launch* L) {
    const int R = K / 2;
    if (tw <= 0 || th <= 0 || (tw & 3) || nq <= 0) return false;
    PropArgs& a = L->a;
    a.B = B; a.H = H; a.W = W; a.S = S;
    a.tw = tw; a.th = th;
    a.tiles_x = ceil_div(W, tw);
    a.tiles_y = ceil_div(H, th);
    a.hyw = (S - 1) * R;
    a.hxw = round_up4((S - 1) * R);
    a.wq = (tw + 2 * a.hxw) / 4;
    a.wr = th + 2 * a.hyw;
    a.dr = a.wr + 2 * R;
    a.ls = 4 * a.wq + 8;
    if ((long)a.wq * ceil_div(a.wr, nq) > threads) return false;
    const int blend_planes = blend == CSPN_BLEND_SPARSE ? 2 : (blend == CSPN_BLEND_PREMASK ? 1 : 0);
    L->lds_bytes = ((size_t)2 * a.dr * a.ls + (size_t)blend_planes * a.wr * 4 * a.wq + 16 * 10 /* SCORE partials */) *
                   sizeof(float);
    if (L->lds_bytes > 160 * 1024) return false;
    L->grid = B * a.tiles_x * a.tiles_y;
    L->threads = threads;
    L->nq = nq;
    return true;
}

// Built-in plan heuristic (overridable through cspn_plan, or replaced by the host-side autotuner).
// Rules distilled from plan sweeps on MI355X (profiles/r01_plan_sweep_*.txt, DESIGN.md "plan selection"):
//   * temporal blocking pays until the halo work (ratio ~1.8) eats the saved launches: S0 = 8 / 3 / 2 steps
//     per launch for K = 3 / 5 / 7, balanced over ceil(T/S0) launches;
//   * one quad per thread (<= 64 VGPRs at K=3 -> 8 waves/SIMD) and the largest workgroup: two 1024-thread
//     workgroups per CU overlap one tile's weight stream with the other's LDS steps;
//   * tile width = W split into n equal parts (rounded up to whole quads), tile height = every row the
//     workgroup can own, evened out over the image; the (n, height) pair with the fewest total
//     weight-region pixels (tiles x (tile + halo)) wins, wider tile on ties.
void default_plan(int K, int B, int H, int W, int T, int keep_history, cspn_plan* p) {
    (void)B; (void)keep_history;
    const int R = K / 2;
    p->force_scalar = 0;
    p->threads = (K == 3) ? 1024 : 256;
    p->quads_per_thread = 1;
    int S = (K == 3) ? 8 : (K == 5 ? 3 : 2);
    if (T < 1) T = 1;
    if (S > T) S = T;
    S = ceil_div(T, ceil_div(T, S));                 // balance the launches (T=24, S0=8 -> 3 x 8)
    for (;; --S) {                                   // shrink S until some tiling fits the workgroup
        const int hyw = (S - 1) * R, hxw = round_up4(hyw);
        long best_cost = -1;
        for (int n = 1; n <= 64; ++n) {
            const int tw = round_up4(ceil_div(W, n));
            if (n > 1 && tw < 16) break;
            const int wq = (tw + 2 * hxw) / 4;
            if (wq > p->threads) continue;
            int th = p->quads_per_thread * (p->threads / wq) - 2 * hyw;
            if (th > H) th = H;
            if (th < 1 || (th < 8 && th < H)) continue;
            th = ceil_div(H, ceil_div(H, th));       // even out the tile rows
            const long cost = (long)ceil_div(W, tw) * ceil_div(H, th) * (4L * wq) * (th + 2 * hyw);
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; p->tile_w = tw; p->tile_h = th; }
        }
        if (best_cost >= 0 || S == 1) break;
    }
    p->steps_per_launch = S;
    if (p->tile_w <= 0) { p->tile_w = round_up4(W < 64 ? W : 64); p->tile_h = p->threads / (p->tile_w / 4); }
}

void resolve_plan(int K, int B, int H, int W, int T, int keep_history, const cspn_plan* user, cspn_plan* p) {
    default_plan(K, B, H, W, T, keep_history, p);
    if (user) {
        if (user->steps_per_launch > 0) p->steps_per_launch = user->steps_per_launch;
        if (user->tile_w > 0) p->tile_w = user->tile_w;
        if (user->tile_h > 0) p->tile_h = user->tile_h;
        if (user->quads_per_thread > 0) p->quads_per_thread = user->quads_per_thread;
        if (user->threads > 0) p->threads = user->threads;
        p->force_scalar = user->force_scalar;
    }
    if (W % 4 != 0) p->force_scalar = 1;
    if (p->force_scalar) p->steps_per_launch = 1;
    if (p->steps_per_launch > T && T > 0) p->steps_per_launch = T;
    // do not own rows far below the image
    const int nq = p->quads_per_thread;
    const int hmax = ceil_div(H, nq) * nq;
    if (p->tile_h > hmax) p->tile_h = hmax;
}

template <int K, int NQ, int NTHREADS, typename WT, typename DT, int WSRC, int SCORE = 0>
int launch_fused_blend(const Launch& L, int blend, hipStream_t st) {
#define CSPN_LAUNCH(BL)                                                                               \
    do {                                                                                              \
        auto kern = cspn_prop_fused<K, NQ, NTHREADS, WT, DT, BL, WSRC, SCORE>;                        \
        if (L.lds_bytes > 64 * 1024)                                                                  \
            HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                           \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds_bytes)); \
        hipLaunchKernelGGL(kern, dim3(L.grid), dim3(NTHREADS), L.lds_bytes, st, L.a);                 \
    } while (0)
    switch (blend) {
        case CSPN_BLEND_NONE: CSPN_LAUNCH(CSPN_BLEND_NONE); break;
        case CSPN_BLEND_SPARSE: CSPN_LAUNCH(CSPN_BLEND_SPARSE); break;
        case CSPN_BLEND_PREMASK: CSPN_LAUNCH(CSPN_BLEND_PREMASK); break;
        default: return fail("bad blend mode %d", blend);
    }
#undef CSPN_LAUNCH
    HIP_OK(hipGetLastError());
    return 1;
}

template <int K, typename WT, typename DT>
int launch_fused(const Launch& L, int blend, int wsrc, hipStream_t st) {
    if (L.a.macc) {   // scoring launch: the one-quad instances the built-in plans use
        if (wsrc || blend == CSPN_BLEND_PREMASK)
            return fail("scoring needs a launch that streams prepared (or published) weights, forward only");
        if constexpr (K == 3 || K == 5) {
            if constexpr (std::is_same<WT, DT>::value) {
#define CSPN_CASE_S(NQV, NTV) \
    if (L.nq == NQV && L.threads == NTV) return launch_fused_blend<K, NQV, NTV, WT, DT, 0, 1>(L, blend, st)
                CSPN_CASE_S(1, 256); CSPN_CASE_S(1, 512);
                if constexpr (K == 3) { CSPN_CASE_S(1, 1024); CSPN_CASE_S(2, 512); }
#undef CSPN_CASE_S
            }
        }
        return fail("no scoring kernel instance for K=%d quads_per_thread=%d threads=%d", K, L.nq, L.threads);
    }
    if constexpr (K == 3) {
        if (wsrc) {      // on-the-fly weights: one- and two-quad instances only
#define CSPN_CASE_G(NQV, NTV) \
    if (L.nq == NQV && L.threads == NTV) return launch_fused_blend<K, NQV, NTV, WT, DT, 1>(L, blend, st)
            CSPN_CASE_G(1, 256); CSPN_CASE_G(2, 256); CSPN_CASE_G(1, 512); CSPN_CASE_G(2, 512);
            CSPN_CASE_G(1, 1024); CSPN_CASE_G(2, 1024);
#undef CSPN_CASE_G
            return fail("no from-guidance kernel instance for quads_per_thread=%d threads=%d", L.nq, L.threads);
        }
    } else if (wsrc) {
        return fail("on-the-fly weights exist for K=3 only");
    }
#define CSPN_CASE(NQV, NTV) \
    if (L.nq == NQV && L.threads == NTV) return launch_fused_blend<K, NQV, NTV, WT, DT, 0>(L, blend, st)
    if constexpr (K == 3) {
        CSPN_CASE(1, 256); CSPN_CASE(2, 256); CSPN_CASE(4, 256); CSPN_CASE(8, 256);
        CSPN_CASE(1, 512); CSPN_CASE(2, 512); CSPN_CASE(4, 512); CSPN_CASE(1, 1024); CSPN_CASE(2, 1024);
    } else if constexpr (K == 5) {
        CSPN_CASE(1, 256); CSPN_CASE(2, 256); CSPN_CASE(3, 256); CSPN_CASE(1, 512);
    } else {
        CSPN_CASE(1, 256);
    }
#undef CSPN_CASE
    return fail("no kernel instance for K=%d quads_per_thread=%d threads=%d", K, L.nq, L.threads);
}

template <int K, typename WT, typename DT>
int launch_scalar(const void* w, const void* din, void* dout, const void* sp, const void* d0, int B, int H,
                  int W, int blend, hipStream_t st) {
    const int grid = grid_for((size_t)B * H * W, 256);
    switch (blend) {
        case CSPN_BLEND_NONE:
            hipLaunchKernelGGL((cspn_prop_scalar<K, WT, DT, CSPN_BLEND_NONE>), dim3(grid), dim3(256), 0, st, w, din, dout, sp, d0, B, H, W);
            break;
        case CSPN_BLEND_SPARSE:
            hipLaunchKernelGGL((cspn_prop_scalar<K, WT, DT, CSPN_BLEND_SPARSE>), dim3(grid), dim3(256), 0, st, w, din, dout, sp, d0, B, H, W);
            break;
        case CSPN_BLEND_PREMASK:
            hipLaunchKernelGGL((cspn_prop_scalar<K, WT, DT, CSPN_BLEND_PREMASK>), dim3(grid), dim3(256), 0, st, w, din, dout, sp, d0, B, H, W);
            break;
        default: return fail("bad blend mode %d", blend);
    }
    HIP_OK(hipGetLastError());
    return 1;
}

template <int K, typename WT, typename DT>
int propagate_typed(const void* w, const void* d0, const void* sparse, void* out, void* history, void* work,
                    int B, int H, int W, int T, int blend, const cspn_plan* user, hipStream_t st,
                    int wsrc = 0, long g_bs = 0, long g_cs = 0, const void* target = nullptr, double* macc = nullptr,
                    int nslots = 0, void* w_out = nullptr) {
    const size_t plane_bytes = (size_t)B * H * W * sizeof(DT);
    if (T == 0) {
        if (out) HIP_OK(hipMemcpyAsync(out, d0, plane_bytes, hipMemcpyDeviceToDevice, st));
        return 1;
    }
    cspn_plan p;
    resolve_plan(K, B, H, W, T, history != nullptr, user, &p);
    // the vector kernel needs whole, 16-byte (8-byte for f16) aligned quads
    bool vec = !p.force_scalar && (W % 4 == 0) && aligned16(w) && aligned16(d0) && (!out || aligned16(out)) &&
               (!history || aligned16(history)) && (!work || aligned16(work)) && (!sparse || aligned16(sparse));
    if (p.steps_per_launch > 1 && !vec) p.steps_per_launch = 1;

    // destination chain: d0 -> (work0 <-> work1)* -> out,  or history planes
    char* wk = static_cast<char*>(work);
    const void* src = d0;
    int t = 0, launch_idx = 0;
    const int n_launch = vec ? ceil_div(T, p.steps_per_launch) : T;
    if (!history && n_launch > 1 && !work) return fail("workspace required (T=%d, launches=%d)", T, n_launch);
    if (!history && !out) return fail("out is NULL and no history requested");
    while (t < T) {
        const int S = vec ? (T - t < p.steps_per_launch ? T - t : p.steps_per_launch) : 1;
        void* dst;
        void* hist_base = nullptr;
        if (history) {
            hist_base = static_cast<char*>(history) + (size_t)t * plane_bytes;
            dst = static_cast<char*>(history) + (size_t)(t + S - 1) * plane_bytes;
        } else {
            dst = (t + S >= T) ? out : static_cast<void*>(wk + (size_t)(launch_idx & 1) * plane_bytes);
        }
        if (vec) {
            Launch L{};
            if (!make_geometry(K, B, H, W, S, p.tile_w, p.tile_h, p.quads_per_thread, p.threads, blend, &L))
                return fail("plan does not fit: K=%d S=%d tile=%dx%d nq=%d threads=%d", K, S, p.tile_w, p.tile_h,
                            p.quads_per_thread, p.threads);
            // from-guidance with a weight buffer: the first launch derives + publishes the weights, the rest stream them
            const bool derive = wsrc && (launch_idx == 0 || !w_out);
            L.a.w = (wsrc && !derive) ? w_out : w;
            L.a.w_out = (derive && n_launch > 1) ? w_out : nullptr;
            L.a.g_bs = g_bs; L.a.g_cs = g_cs; L.a.d_in = src; L.a.sparse = sparse; L.a.d0 = d0;
            L.a.d_out = history ? nullptr : dst;
            L.a.hist = hist_base;
            const bool final_launch = (t + S >= T);
            if (final_launch && macc && derive)
                return fail("scored from-guidance propagation needs more than one launch (T > steps_per_launch)");
            L.a.target = final_launch ? target : nullptr;
            L.a.macc = final_launch ? macc : nullptr;
            L.a.nslots = nslots;
            if (!launch_fused<K, WT, DT>(L, blend, derive ? 1 : 0, st)) return 0;
        } else {
            if (wsrc) return fail("from-guidance propagation needs W %% 4 == 0 and 16-byte aligned tensors; "
                                  "use cspn3_prepare + cspn_propagate");
            if (macc) return fail("scored propagation needs W %% 4 == 0 and 16-byte aligned tensors; use "
                                  "cspn_propagate + cspn_metrics_accumulate");
            if (!launch_scalar<K, WT, DT>(w, src, dst, sparse, d0, B, H, W, blend, st)) return 0;
        }
        src = dst;
        t += S;
        ++launch_idx;
    }
    return 1;
}

template <int K>
int propagate_k(const void* w, int w_dtype, const void* d0, const void* sparse, void* out, void* history,
                void* work, int d_dtype, int B, int H, int W, int T, int blend, const cspn_plan* plan,
                hipStream_t st) {
    if (w_dtype == CSPN_F32 && d_dtype == CSPN_F32)
        return propagate_typed<K, float, float>(w, d0, sparse, out, history, work, B, H, W, T, blend, plan, st);
    if (w_dtype == CSPN_F16 && d_dtype == CSPN_F16)
        return propagate_typed<K, __half, __half>(w, d0, sparse, out, history, work, B, H, W, T, blend, plan, st);
    if (w_dtype == CSPN_F16 && d_dtype == CSPN_F32)
        return propagate_typed<K, __half, float>(w, d0, sparse, out, history, work, B, H, W, T, blend, plan, st);
    return fail("unsupported dtype combination w=%d d=%d", w_dtype, d_dtype);
}

template <int K, typename DT, typename WT>
int launch_tail(const TailArgs& a, int variant, hipStream_t st) {
    const size_t nquads = (size_t)a.B * a.H * (a.W / 4);
    const int grid = (int)((nquads + 255) / 256);
    if (variant == 0) hipLaunchKernelGGL((cspn_grad_tail<K, DT, WT, 0>), dim3(grid), dim3(256), 0, st, a);
    else if (variant == 2) hipLaunchKernelGGL((cspn_grad_tail<K, DT, WT, 2>), dim3(grid), dim3(256), 0, st, a);
    else if constexpr (K == 3) hipLaunchKernelGGL((cspn_grad_tail<3, DT, WT, 1>), dim3(grid), dim3(256), 0, st, a);
    else return fail("backward tail variant %d unsupported for K=%d", variant, K);
    HIP_OK(hipGetLastError());
    return 1;
}

template <int K>
int launch_tail_typed(const TailArgs& a, int variant, int d_dtype, int w_dtype, hipStream_t st) {
    if (d_dtype == CSPN_F32 && w_dtype == CSPN_F32) return launch_tail<K, float, float>(a, variant, st);
    if (d_dtype == CSPN_F16 && w_dtype == CSPN_F16) return launch_tail<K, __half, __half>(a, variant, st);
    if (d_dtype == CSPN_F32 && w_dtype == CSPN_F16) return launch_tail<K, float, __half>(a, variant, st);
    return fail("backward tail: unsupported dtypes d=%d w=%d", d_dtype, w_dtype);
}

bool tail_vector_ok(const TailArgs& a) {
    return (a.W % 4 == 0) && aligned16(a.d0) && (!a.dhist || aligned16(a.dhist)) && aligned16(a.ghist) &&
           (!a.sparse || aligned16(a.sparse)) && (!a.w || aligned16(a.w)) && (!a.S || aligned16(a.S)) &&
           (!a.guidance || aligned16(a.guidance)) && aligned16(a.gout) && aligned16(a.gd0) &&
           (a.g_bs % 4 == 0) && (a.g_cs % 4 == 0);
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int cspn_abi_version(void) { return CSPN_ABI_VERSION; }

int cspn_plan_resolve(int K, int B, int H, int W, int T, int keep_history, const cspn_plan* plan_or_null,
                      cspn_plan* resolved) {
    if (!resolved) return fail("cspn_plan_resolve: NULL output");
    if (K != 3 && K != 5 && K != 7) return fail("cspn_plan_resolve: unsupported K=%d", K);
    resolve_plan(K, B, H, W, T, keep_history, plan_or_null, resolved);
    if (!resolved->force_scalar) {
        Launch L;
        if (!make_geometry(K, B, H, W, resolved->steps_per_launch, resolved->tile_w, resolved->tile_h,
                           resolved->quads_per_thread, resolved->threads, CSPN_BLEND_SPARSE /* worst-case LDS */, &L))
            return fail("plan does not fit: K=%d S=%d tile=%dx%d nq=%d threads=%d", K, resolved->steps_per_launch,
                        resolved->tile_w, resolved->tile_h, resolved->quads_per_thread, resolved->threads);
    }
    return 1;
}
const char* cspn_last_error(void) { return g_err; }

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

size_t cspn_propagate_workspace_bytes(int B, int H, int W, int T, int d_dtype, int keep_history) {
    if (keep_history || T <= 1) return 0;
    return (size_t)2 * B * H * W * esize(d_dtype);
}

int cspn_propagate(const void* w, int w_dtype, const void* d0, const void* sparse, void* out, void* history,
                   void* work, int d_dtype, int B, int H, int W, int K, int T, int blend, const cspn_plan* plan,
                   cspn_stream_t stream) {
    if (!w || !d0 || B <= 0 || H <= 0 || W <= 0 || T < 0) return fail("cspn_propagate: bad arguments");
    if (blend != CSPN_BLEND_NONE && !sparse) return fail("cspn_propagate: blend=%d needs sparse", blend);
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (K) {
        case 3: return propagate_k<3>(w, w_dtype, d0, sparse, out, history, work, d_dtype, B, H, W, T, blend, plan, st);
        case 5: return propagate_k<5>(w, w_dtype, d0, sparse, out, history, work, d_dtype, B, H, W, T, blend, plan, st);
        case 7: return propagate_k<7>(w, w_dtype, d0, sparse, out, history, work, d_dtype, B, H, W, T, blend, plan, st);
        default: return fail("cspn_propagate: unsupported K=%d (3, 5, 7)", K);
    }
}

int cspn_propagate_scored(const void* w, int w_dtype, const void* d0, const void* sparse, void* out, void* work,
                          int d_dtype, int B, int H, int W, int K, int T, int blend, const void* target, double* acc,
                          int nslots, const cspn_plan* plan, cspn_stream_t stream) {
    if (!w || !d0 || !out || !target || !acc || nslots < 1 || B <= 0 || H <= 0 || W <= 0 || T < 1)
        return fail("cspn_propagate_scored: bad arguments");
    if (blend != CSPN_BLEND_NONE && blend != CSPN_BLEND_SPARSE) return fail("cspn_propagate_scored: blend %d", blend);
    if (blend != CSPN_BLEND_NONE && !sparse) return fail("cspn_propagate_scored: blend needs sparse");
    if (!aligned16(target)) return fail("cspn_propagate_scored: target must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
#define SCORED(KV, WTT, DTT) \
    return propagate_typed<KV, WTT, DTT>(w, d0, sparse, out, nullptr, work, B, H, W, T, blend, plan, st, 0, 0, 0, target, acc, nslots)
    if (K == 3 && w_dtype == CSPN_F32 && d_dtype == CSPN_F32) SCORED(3, float, float);
    if (K == 3 && w_dtype == CSPN_F16 && d_dtype == CSPN_F16) SCORED(3, __half, __half);
    if (K == 5 && w_dtype == CSPN_F32 && d_dtype == CSPN_F32) SCORED(5, float, float);
    if (K == 5 && w_dtype == CSPN_F16 && d_dtype == CSPN_F16) SCORED(5, __half, __half);
#undef SCORED
    return fail("cspn_propagate_scored: unsupported K=%d / dtypes w=%d d=%d", K, w_dtype, d_dtype);
}

int cspn3_propagate_from_guidance(const void* guidance, int g_dtype, long bs, long cs, void* w8_out, const void* d0,
                                  const void* sparse, void* out, void* history, void* work, int d_dtype, int B, int H,
                                  int W, int T, int blend, const void* target, double* acc, int nslots,
                                  const cspn_plan* plan, cspn_stream_t stream) {
    if ((target || acc) && (!target || !acc || nslots < 1 || !w8_out || history || !aligned16(target) || g_dtype != d_dtype))
        return fail("cspn3_propagate_from_guidance: scoring needs target, acc, nslots >= 1, w8_out, no history, "
                    "one dtype and a 16-byte aligned target");
    if (!guidance || !d0 || B <= 0 || H <= 0 || W <= 0 || T < 0) return fail("cspn3_propagate_from_guidance: bad arguments");
    if (blend != CSPN_BLEND_NONE && blend != CSPN_BLEND_SPARSE) return fail("cspn3_propagate_from_guidance: blend %d", blend);
    if (blend != CSPN_BLEND_NONE && !sparse) return fail("cspn3_propagate_from_guidance: blend needs sparse");
    if ((cs & 3) || (bs & 3)) return fail("cspn3_propagate_from_guidance: guidance strides must be multiples of 4 elements");
    if (w8_out && !aligned16(w8_out)) return fail("cspn3_propagate_from_guidance: w8_out must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (g_dtype == CSPN_F32 && d_dtype == CSPN_F32)
        return propagate_typed<3, float, float>(guidance, d0, sparse, out, history, work, B, H, W, T, blend, plan, st, 1, bs, cs, target, acc, nslots, w8_out);
    if (g_dtype == CSPN_F16 && d_dtype == CSPN_F16)
        return propagate_typed<3, __half, __half>(guidance, d0, sparse, out, history, work, B, H, W, T, blend, plan, st, 1, bs, cs, target, acc, nslots, w8_out);
    if (g_dtype == CSPN_F16 && d_dtype == CSPN_F32)
        return propagate_typed<3, __half, float>(guidance, d0, sparse, out, history, work, B, H, W, T, blend, plan, st, 1, bs, cs, target, acc, nslots, w8_out);
    return fail("cspn3_propagate_from_guidance: unsupported dtypes g=%d d=%d", g_dtype, d_dtype);
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

int cspn_grad_weights(const void* d0, const void* dhist, const float* ghist, const void* sparse, float* gw,
                      float* gd0, int d_dtype, int B, int H, int W, int K, int T, cspn_stream_t stream) {
    if (!d0 || !ghist || !gw || !gd0 || (T > 1 && !dhist)) return fail("cspn_grad_weights: NULL pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    {   // vector path: the fused tail without an epilogue (dL/dw written as is)
        TailArgs a{};
        a.d0 = d0; a.dhist = dhist; a.ghist = ghist; a.sparse = sparse; a.gout = gw; a.gd0 = gd0;
        a.B = B; a.H = H; a.W = W; a.T = T;
        if (tail_vector_ok(a) && (K == 3 || K == 5 || K == 7)) {
            switch (K) {
                case 3: return launch_tail_typed<3>(a, 0, d_dtype, d_dtype, st);
                case 5: return launch_tail_typed<5>(a, 0, d_dtype, d_dtype, st);
                default: return launch_tail_typed<7>(a, 0, d_dtype, d_dtype, st);
            }
        }
    }
    const int grid = grid_for((size_t)B * H * W, 256);
#define GWK(KV)                                                                                                \
    if (K == KV) {                                                                                             \
        if (d_dtype == CSPN_F32)                                                                               \
            hipLaunchKernelGGL((cspn_grad_weights_kernel<KV, float>), dim3(grid), dim3(256), 0, st,            \
                               static_cast<const float*>(d0), static_cast<const float*>(dhist), ghist,         \
                               static_cast<const float*>(sparse), gw, gd0, B, H, W, T);                        \
        else                                                                                                   \
            hipLaunchKernelGGL((cspn_grad_weights_kernel<KV, __half>), dim3(grid), dim3(256), 0, st,           \
                               static_cast<const __half*>(d0), static_cast<const __half*>(dhist), ghist,       \
                               static_cast<const __half*>(sparse), gw, gd0, B, H, W, T);                       \
        HIP_OK(hipGetLastError());                                                                             \
        return 1;                                                                                              \
    }
    GWK(3) GWK(5) GWK(7)
#undef GWK
    return fail("cspn_grad_weights: unsupported K=%d", K);
}

int cspn3_grad_guidance(const void* guidance, int g_dtype, long bs, long cs, int C, const void* w8, int w_dtype,
                        const float* s, const float* gw, void* grad_guidance, int B, int H, int W,
                        cspn_stream_t stream) {
    if (!guidance || !w8 || !s || !gw || !grad_guidance) return fail("cspn3_grad_guidance: NULL pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = grid_for((size_t)B * H * W, 256);
    if (g_dtype == CSPN_F32 && w_dtype == CSPN_F32)
        hipLaunchKernelGGL((cspn3_grad_guidance_kernel<float, float>), dim3(grid), dim3(256), 0, st,
                           static_cast<const float*>(guidance), bs, cs, C, static_cast<const float*>(w8), s, gw,
                           static_cast<float*>(grad_guidance), B, H, W);
    else if (g_dtype == CSPN_F16 && w_dtype == CSPN_F16)
        hipLaunchKernelGGL((cspn3_grad_guidance_kernel<__half, __half>), dim3(grid), dim3(256), 0, st,
                           static_cast<const __half*>(guidance), bs, cs, C, static_cast<const __half*>(w8), s, gw,
                           static_cast<__half*>(grad_guidance), B, H, W);
    else
        return fail("cspn3_grad_guidance: unsupported dtypes g=%d w=%d", g_dtype, w_dtype);
    HIP_OK(hipGetLastError());
    return 1;
}

int cspn_pac_grad_guided(const void* wk, int w_dtype, const float* gw, void* grad_guided, int g_dtype, int B,
                         int H, int W, int K, cspn_stream_t stream) {
    if (!wk || !gw || !grad_guided) return fail("cspn_pac_grad_guided: NULL pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = grid_for((size_t)B * H * W, 256);
#define PG(KV)                                                                                                 \
    if (K == KV) {                                                                                             \
        if (w_dtype == CSPN_F32 && g_dtype == CSPN_F32)                                                        \
            hipLaunchKernelGGL((cspn_pac_grad_guided_kernel<KV, float, float>), dim3(grid), dim3(256), 0, st,  \
                               static_cast<const float*>(wk), gw, static_cast<float*>(grad_guided), B, H, W);  \
        else if (w_dtype == CSPN_F16 && g_dtype == CSPN_F16)                                                   \
            hipLaunchKernelGGL((cspn_pac_grad_guided_kernel<KV, __half, __half>), dim3(grid), dim3(256), 0, st,\
                               static_cast<const __half*>(wk), gw, static_cast<__half*>(grad_guided), B, H, W);\
        else                                                                                                   \
            return fail("cspn_pac_grad_guided: unsupported dtypes");                                           \
        HIP_OK(hipGetLastError());                                                                             \
        return 1;                                                                                              \
    }
    PG(3) PG(5) PG(7)
#undef PG
    return fail("cspn_pac_grad_guided: unsupported K=%d", K);
}

int cspn3_backward_tail(const void* d0, const void* dhist, const float* ghist, const void* sparse,
                        const void* guidance, long bs, long cs, int C, const void* w8, const float* s,
                        void* grad_guidance, float* gd0, int dtype, int B, int H, int W, int T, cspn_stream_t stream) {
    if (!d0 || !ghist || !guidance || !w8 || !s || !grad_guidance || !gd0 || (T > 1 && !dhist))
        return fail("cspn3_backward_tail: NULL pointer");
    TailArgs a{};
    a.d0 = d0; a.dhist = dhist; a.ghist = ghist; a.sparse = sparse; a.w = w8; a.S = s; a.guidance = guidance;
    a.gout = grad_guidance; a.gd0 = gd0; a.g_bs = bs; a.g_cs = cs; a.B = B; a.H = H; a.W = W; a.T = T; a.C = C;
    if (!tail_vector_ok(a)) return fail("cspn3_backward_tail needs W %% 4 == 0 and 16-byte aligned tensors; use "
                                        "cspn_grad_weights + cspn3_grad_guidance");
    return launch_tail_typed<3>(a, 1, dtype, dtype, static_cast<hipStream_t>(stream));
}

int cspn_pac_backward_tail(const void* d0, const void* dhist, const float* ghist, const void* sparse, const void* wk,
                           void* grad_guided, float* gd0, int d_dtype, int w_dtype, int B, int H, int W, int K, int T,
                           cspn_stream_t stream) {
    if (!d0 || !ghist || !wk || !grad_guided || !gd0 || (T > 1 && !dhist)) return fail("cspn_pac_backward_tail: NULL pointer");
    TailArgs a{};
    a.d0 = d0; a.dhist = dhist; a.ghist = ghist; a.sparse = sparse; a.w = wk; a.gout = grad_guided; a.gd0 = gd0;
    a.B = B; a.H = H; a.W = W; a.T = T;
    if (!tail_vector_ok(a)) return fail("cspn_pac_backward_tail needs W %% 4 == 0 and 16-byte aligned tensors; use "
                                        "cspn_grad_weights + cspn_pac_grad_guided");
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (K) {
        case 3: return launch_tail_typed<3>(a, 2, d_dtype, w_dtype, st);
        case 5: return launch_tail_typed<5>(a, 2, d_dtype, w_dtype, st);
        case 7: return launch_tail_typed<7>(a, 2, d_dtype, w_dtype, st);
        default: return fail("cspn_pac_backward_tail: unsupported K=%d", K);
    }
}

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
