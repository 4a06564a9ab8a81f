// cspn_backward.hip — backward tails of the propagation loop (the reverse recurrence itself reuses
// cspn_prop_fused on the transposed weights).  See DESIGN.md §4.2.
#include "cspn_common.hpp"

#include <cstdlib>

namespace {

// ------------------------------------------------------------------------------------------------
// backward helpers
// ------------------------------------------------------------------------------------------------
// gw_j[p] = (1-m) sum_t G_{t+1}[p] d_t[p+off_j];  gd0[p] = G_0[p] + m sum_{t>=1} G_t[p]
template <int K, typename DT>
__global__ void cspn_grad_weights_kernel(const DT* __restrict__ d0, const DT* __restrict__ dhist,
                                         const float* __restrict__ g_T, const float* __restrict__ ghist,
                                         const DT* __restrict__ sparse, float* __restrict__ gw, float* __restrict__ gd0,
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
            const float G = (t == T - 1) ? g_T[i] : ghist[(size_t)(T - 2 - t) * total + i];   // G_{t+1}
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
        gd0[i] = (T > 0 ? ghist[(size_t)(T - 1) * total + i] : g_T[i]) + m * gsum;        // G_0 + m sum_{t>=1} G_t
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
    const float* g_T;     // [B,H,W] G_T = dL/d(d_T), the incoming gradient (read in place, never copied)
    const float* ghist;   // [T,B,H,W] backward order: ghist[s] = G_{T-1-s}, what the reverse sweep wrote
    const void* sparse;   // [B,H,W] DT or null
    const void* w;        // [B,NT,H,W] WT tap planes (variants 1, 2)
    const float* S;       // [B,H,W] (variant 1)
    const void* guidance; // variant 1: [B,C,H,W] WT through strides
    void* gout;           // variant 0: gw f32 [B,NT,H,W]; 1: grad_guidance WT (guidance strides); 2: grad_guided WT
    void* gd0;            // [B,H,W] f32, or fp16 when gd0_half (the training step on half planes: no cast launch behind the tail)
    long g_bs, g_cs;
    int B, H, W, T, C;
    int gd0_half;
};

// History loads of the tail as RAW bits, converted where they are consumed.  For fp32 planes this is the plain load.  For fp16
// planes a guarded `ok ? ld(p) : 0` makes an exec-masked block, and the compiler sinks the f16 -> f32 conversion INTO that block
// (cvt(0) = 0), right behind the load: every block then waits for its own round trip — 10 row quads + the strip-end patches per
// trip of the stream loop, serialised (cspn_grad_tail<5, __half, ...> ran at 302 us against 127 us for its fp32-history twin,
// profiles/r04_kernel_stats_train_leg_pac5_state16.csv).  Raw loads from a safe address + a select after the conversion keep
// every load of a trip in flight together; the empty asm pins the conversion behind the load phase.
// Cache policy of the tails' single-use streams.  Every G quad is read by exactly one thread, every d row by the threads of three
// rows: G loads that allocate in the L2 push out the d lines the neighbouring rows are about to re-read.  Non-temporal G loads
// (bit 2), epilogue stores (4) and epilogue guidance loads (8): cspn_grad_tail<3, float, float, 1> 107.5-108.7 -> 89.5-93.0 us in a
// same-box A/B (two alternating repetitions; G loads alone: 92.0-96.8); non-temporal loads of the d history (1) cost 4 us —
// they are the re-used stream.  The K = 5 split tail does not care (66-69 us either way).  Results are bit-identical.
#ifndef CSPN_TAIL_NT
#define CSPN_TAIL_NT 14
#endif
__device__ __forceinline__ float4 ld4_nt(const float* p) {
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4_nt(float* p, float4 q) {
    const v4f v = {q.x, q.y, q.z, q.w};
    __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p));
}
template <typename WT>
__device__ __forceinline__ float4 ld4_ep(const WT* p) {
    if constexpr (std::is_same<WT, float>::value && (CSPN_TAIL_NT & 8) != 0) return ld4_nt(p);
    else return ld4(p);
}
template <typename WT>
__device__ __forceinline__ void st4_ep(WT* p, float4 v) {
    if constexpr (std::is_same<WT, float>::value && (CSPN_TAIL_NT & 4) != 0) st4_nt(p, v);
    else st4(p, v);
}
template <typename DT> struct TailRaw;
template <> struct TailRaw<float> {
    static constexpr bool RAW = false;        // fp32: the guarded loads themselves (no conversion to sink; 124 VGPRs = 4 waves per SIMD)
    typedef float4 Q;
    typedef float S;
    static __device__ __forceinline__ Q ldq(const float* p) { return (CSPN_TAIL_NT & 1) ? ld4_nt(p) : ld4(p); }
    static __device__ __forceinline__ S lds(const float* p) { return *p; }
    static __device__ __forceinline__ float4 f4(Q q) { return q; }
    static __device__ __forceinline__ float f1(S s) { return s; }
    static __device__ __forceinline__ S zero() { return 0.f; }
};
template <> struct TailRaw<__half> {
    static constexpr bool RAW = true;
    typedef uint2 Q;
    typedef unsigned S;
    static __device__ __forceinline__ Q ldq(const __half* p) { return *reinterpret_cast<const uint2*>(p); }
    static __device__ __forceinline__ S lds(const __half* p) { return *reinterpret_cast<const unsigned short*>(p); }
    static __device__ __forceinline__ float4 f4(Q q) {
        asm volatile("" : "+v"(q.x), "+v"(q.y));
        const float2 a = __half22float2(*reinterpret_cast<const __half2*>(&q.x));
        const float2 b = __half22float2(*reinterpret_cast<const __half2*>(&q.y));
        return make_float4(a.x, a.y, b.x, b.y);
    }
    static __device__ __forceinline__ float f1(S s) {
        asm volatile("" : "+v"(s));
        return __half2float(__ushort_as_half((unsigned short)s));
    }
    static __device__ __forceinline__ S zero() { return 0u; }
};

#ifndef CSPN_TAIL_REBUILD_W
#define CSPN_TAIL_REBUILD_W 1      // 3x3 epilogue: rebuild w_j from the guidance quads + S instead of reading the tap volume
#endif
#ifndef CSPN_TAIL_PROBE
#define CSPN_TAIL_PROBE 0          // developer A/B: 1 = no scatter epilogue (stream + epilogue loads only), 2 = no stream loop
#endif
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
#ifndef CSPN_TAIL_UNR3
#define CSPN_TAIL_UNR3 3       // 124 VGPRs -> 4 waves/SIMD; 4 steps in flight need 146 (3 waves) and measured 2.5 % slower
#endif
    constexpr int UNR = (K == 3) ? CSPN_TAIL_UNR3 : (K == 5 ? 2 : 1);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t0 = 0; t0 < (CSPN_TAIL_PROBE == 2 ? 0 : T); t0 += UNR) {
        typedef TailRaw<DT> RW;
        float4 Gq[UNR];
        typename RW::Q midq[UNR][2 * R + 1];
        typename RW::S lf[UNR][2 * R + 1][R], rf[UNR][2 * R + 1][R];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int t = t0 + u;
            const bool tv = live && t < T;
            const DT* d = (t == 0) ? d0 : dh + (size_t)(tv ? t - 1 : 0) * plane;
            const float* gsrc = (t == T - 1) ? a.g_T : a.ghist + (size_t)(tv ? T - 2 - t : 0) * plane;      // G_{t+1}
            Gq[u] = tv ? ((CSPN_TAIL_NT & 2) ? ld4_nt(gsrc + off) : ld4(gsrc + off)) : z4;
#pragma unroll
            for (int rr = 0; rr < 2 * R + 1; ++rr) {
                const int row = y + rr - R;
                const bool rok = tv && row >= 0 && row < H;
                const DT* rp = d + (size_t)b * HW + (size_t)(rok ? row : 0) * W;
                if constexpr (RW::RAW) midq[u][rr] = RW::ldq(rp + x);     // always a valid address; zeroed at the consumer when !rok
                else midq[u][rr] = rok ? RW::ldq(rp + x) : z4;
#pragma unroll
                for (int c = 0; c < R; ++c) { lf[u][rr][c] = RW::zero(); rf[u][rr][c] = RW::zero(); }
                if (fix_left) {
#pragma unroll
                    for (int c = 0; c < R; ++c) {
                        const int xx = x + c - R;
                        if constexpr (RW::RAW) lf[u][rr][c] = RW::lds(rp + (xx >= 0 ? xx : 0));
                        else lf[u][rr][c] = (rok && xx >= 0) ? RW::lds(rp + xx) : RW::zero();
                    }
                }
                if (fix_right) {
#pragma unroll
                    for (int c = 0; c < R; ++c) {
                        const int xx = x + 4 + c;
                        if constexpr (RW::RAW) rf[u][rr][c] = RW::lds(rp + (xx < W ? xx : W - 1));
                        else rf[u][rr][c] = (rok && xx < W) ? RW::lds(rp + xx) : RW::zero();
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const float g4[4] = {Gq[u].x, Gq[u].y, Gq[u].z, Gq[u].w};
            const bool tvu = live && (t0 + u) < T;
            (void)tvu;
            float win[2 * R + 1][WIN];
#pragma unroll
            for (int rr = 0; rr < 2 * R + 1; ++rr) {
                const int row = y + rr - R;
                const bool rok = !RW::RAW || (tvu && row >= 0 && row < H);      // fp32: the loads were guarded already
                const float4 mq = RW::f4(midq[u][rr]);
                const float m4[4] = {rok ? mq.x : 0.f, rok ? mq.y : 0.f, rok ? mq.z : 0.f, rok ? mq.w : 0.f};
#pragma unroll
                for (int c = 0; c < 4; ++c) win[rr][R + c] = m4[c];
#pragma unroll
                for (int c = 0; c < R; ++c) {
                    const float l = dpp_from_prev_lane(m4[4 - R + c]);
                    const float r = dpp_from_next_lane(m4[c]);
                    const float lv = (!RW::RAW || (rok && x + c - R >= 0)) ? RW::f1(lf[u][rr][c]) : 0.f;
                    const float rv = (!RW::RAW || (rok && x + 4 + c < W)) ? RW::f1(rf[u][rr][c]) : 0.f;
                    win[rr][c] = fix_left ? lv : l;
                    win[rr][R + 4 + c] = fix_right ? rv : r;
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
        const float4 G0 = ld4((T > 0 ? a.ghist + (size_t)(T - 1) * plane : a.g_T) + off);
        const float4 gd = make_float4(G0.x + mm[0] * gsum[0], G0.y + mm[1] * gsum[1], G0.z + mm[2] * gsum[2], G0.w + mm[3] * gsum[3]);
        if (a.gd0_half) st4(static_cast<__half*>(a.gd0) + off, gd);
        else st4(static_cast<float*>(a.gd0) + off, gd);
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
        const WT* g = static_cast<const WT*>(a.guidance) + (size_t)b * a.g_bs;
        WT* gg = static_cast<WT*>(a.gout) + (size_t)b * a.g_bs;
        const float4 Sv = ld4(a.S + off);
        const float S4[4] = {Sv.x, Sv.y, Sv.z, Sv.w};
#if CSPN_TAIL_REBUILD_W
        // w_j[p] = |g_{7-j}[p + off_j]| / S[p] is REBUILT from the guidance quads this epilogue loads anyway (for the signs of the
        // scatter targets) instead of being read back from the forward's tap volume: the aligned quad of channel 7-j on row
        // y + dy_j, shifted by dx_j with one DPP move (strip-end lanes patch with the scalar they load for their extra store).
        // Same recipe as the forward (div8_shared_reciprocal: |g| * refined 1/S), so fp32 taps come out bit-identical; 53 MB
        // less to read per backward at config 2.
        float gq[8][4], gl[8], gr[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int lin = j < 4 ? j : j + 1;
            const int dy = lin / 3 - 1, dx = lin % 3 - 1;
            const int ty = y + dy;
            const bool tok = ty >= 0 && ty < H;
            const size_t o = (size_t)(7 - j) * a.g_cs + (size_t)(tok ? ty : 0) * W + x;
            const float4 q4 = ld4_ep(g + o);
            gq[j][0] = tok ? q4.x : 0.f; gq[j][1] = tok ? q4.y : 0.f; gq[j][2] = tok ? q4.z : 0.f; gq[j][3] = tok ? q4.w : 0.f;
            gl[j] = gr[j] = 0.f;
            if (dx > 0 && lane == 63 && qx < WQ - 1 && tok) gr[j] = ld1(g + o + 4);
            if (dx < 0 && lane == 0 && qx > 0 && tok) gl[j] = ld1(g + o - 1);
        }
#endif
        float rS[4];          // 1/S by the forward's recipe (rcp + one Newton step): 4 reciprocals instead of 32 divisions
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float r = __builtin_amdgcn_rcpf(S4[e]);
            rS[e] = fmaf(fmaf(-S4[e], r, 1.0f), r, r);
        }
        float dot[4] = {0.f, 0.f, 0.f, 0.f};
#if CSPN_TAIL_REBUILD_W
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int lin = j < 4 ? j : j + 1;
            const int dx = lin % 3 - 1;
            float ax[4];
            if (dx == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) ax[e] = gq[j][e];
            } else if (dx > 0) {
                float nx = dpp_from_next_lane(gq[j][0]);
                if (fix_right) nx = gr[j];                       // 0 at the row end: the gate beyond the image is the zero padding
                ax[0] = gq[j][1]; ax[1] = gq[j][2]; ax[2] = gq[j][3]; ax[3] = nx;
            } else {
                float pv = dpp_from_prev_lane(gq[j][3]);
                if (fix_left) pv = gl[j];
                ax[0] = pv; ax[1] = gq[j][0]; ax[2] = gq[j][1]; ax[3] = gq[j][2];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) dot[e] = fmaf(acc[j][e], fabsf(ax[e]) * rS[e], dot[e]);
        }
#else
        {
            const WT* w8 = static_cast<const WT*>(a.w) + (size_t)b * Taps<WT>::image_elems(8, HW);
            float w4[8][4];
            load_taps_quad<8>(w8, (size_t)y * W + x, HW, true, w4);
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) dot[e] = fmaf(acc[j][e], w4[j][e], dot[e]);
        }
#endif
#if CSPN_TAIL_PROBE == 1
        if (dot[0] == 12345.678f) st4(gg + (size_t)y * W + x, make_float4(dot[0], dot[1], dot[2], dot[3]));      // stream + loads only
        return;
#endif
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int lin = j < 4 ? j : j + 1;
            const int dy = lin / 3 - 1, dx = lin % 3 - 1;
            const size_t cplane = (size_t)(7 - j) * a.g_cs;
            const int ty = y + dy;
            float gA[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) gA[e] = (acc[j][e] - dot[e]) * rS[e];
            // Scatter gA_j[p] to p + off_j as ALIGNED quads of the target row ty: the target quad [x, x+4) takes
            // source columns [x-dx, x+4-dx), i.e. three own values and one from the neighbouring lane (DPP).
            // Where that neighbour sits in another wave (lane 0 / 63 inside a row) the element is written by its
            // owner with one scalar store and skipped here; where it does not exist (image border) it is 0.
            const float from_prev = dpp_from_prev_lane(gA[3]);
            const float from_next = dpp_from_next_lane(gA[0]);
            if (ty >= 0 && ty < H) {
                const size_t o = cplane + (size_t)ty * W + x;
#if CSPN_TAIL_REBUILD_W
                const float4 gs = sgn4(make_float4(gq[j][0], gq[j][1], gq[j][2], gq[j][3]));
                const float gs_r = sgnf(gr[j]), gs_l = sgnf(gl[j]);
#else
                const float4 gs = sgn4(ld4(g + o));
#endif
                if (dx == 0) {
                    st4_ep(gg + o, make_float4(gs.x * gA[0], gs.y * gA[1], gs.z * gA[2], gs.w * gA[3]));
                } else if (dx > 0) {
                    const float v0 = (qx == 0) ? 0.f : from_prev;              // column 0 has no source
                    if (lane == 0 && qx > 0) {                                   // left neighbour lives in another wave
                        st1(gg + o + 1, gs.y * gA[0]); st1(gg + o + 2, gs.z * gA[1]); st1(gg + o + 3, gs.w * gA[2]);
                    } else {
                        st4_ep(gg + o, make_float4(gs.x * v0, gs.y * gA[0], gs.z * gA[1], gs.w * gA[2]));
                    }
#if CSPN_TAIL_REBUILD_W
                    if (lane == 63 && qx < WQ - 1) st1(gg + o + 4, gs_r * gA[3]);
#else
                    if (lane == 63 && qx < WQ - 1) st1(gg + o + 4, sgnf(ld1(g + o + 4)) * gA[3]);
#endif
                } else {
                    const float v3 = (qx == WQ - 1) ? 0.f : from_next;         // column W-1 has no source
                    if (lane == 63 && qx < WQ - 1) {
                        st1(gg + o, gs.x * gA[1]); st1(gg + o + 1, gs.y * gA[2]); st1(gg + o + 2, gs.z * gA[3]);
                    } else {
                        st4_ep(gg + o, make_float4(gs.x * gA[1], gs.y * gA[2], gs.z * gA[3], gs.w * v3));
                    }
#if CSPN_TAIL_REBUILD_W
                    if (lane == 0 && qx > 0) st1(gg + o - 1, gs_l * gA[0]);
#else
                    if (lane == 0 && qx > 0) st1(gg + o - 1, sgnf(ld1(g + o - 1)) * gA[0]);
#endif
                }
            }
            // rows of this plane that no source row reaches: row 0 (dy=+1) / row H-1 (dy=-1)
            if ((dy > 0 && y == 0) || (dy < 0 && y == H - 1))
                st4_ep(gg + cplane + (size_t)y * W + x, make_float4(0.f, 0.f, 0.f, 0.f));
        }
        for (int c = 8; c < a.C; ++c) st4_ep(gg + (size_t)c * a.g_cs + (size_t)y * W + x, make_float4(0.f, 0.f, 0.f, 0.f));
    }
}

// ---- K = 5: the tail split by TAP ROWS over two half-workgroups -------------------------------------------------------------
// A quad of the 5 x 5 tail carries 24 x 4 accumulators: with the window and two steps of loads in flight that is the whole
// 256-register file, ONE wavefront per SIMD, and every load trip of the stream is exposed (cspn_grad_tail<5, half, half, 2>:
// 142 us for 294 MB = 0.26 of the HBM peak, profiles/r04_kernel_stats_train_leg_pac5.csv).  Here threads 0..127 of a workgroup
// own the taps above the centre (rows dy = -2, -1 and the left half of row 0: j = 0..11) of 128 quads, threads 128..255 the
// mirror half (j = 12..23) of the SAME quads: 48 accumulators and 3 window rows per thread.  The softmax backward needs
// sum_k (1-m) acc_k sm_k over all 24 taps: each half sums its 12 and the two partial sums meet in LDS (one barrier per
// workgroup, after the stream).  Strip-end patches are ONE aligned pair load per row and side (x - 2 / x + 4) instead of two
// scalar loads.  Loads are raw bits from always-valid addresses, zeroed where they are consumed (see TailRaw).
template <typename DT> struct TailPair;
template <> struct TailPair<float> {
    typedef float2 P;
    static __device__ __forceinline__ P ld(const float* p) { return *reinterpret_cast<const float2*>(p); }
    static __device__ __forceinline__ void f2(P v, float& a, float& b) { a = v.x; b = v.y; }
    static __device__ __forceinline__ P zero() { return make_float2(0.f, 0.f); }
};
template <> struct TailPair<__half> {
    typedef unsigned P;
    static __device__ __forceinline__ P ld(const __half* p) { return *reinterpret_cast<const unsigned*>(p); }
    static __device__ __forceinline__ void f2(P v, float& a, float& b) {
        asm volatile("" : "+v"(v));
        const float2 f = __half22float2(*reinterpret_cast<const __half2*>(&v));
        a = f.x; b = f.y;
    }
    static __device__ __forceinline__ P zero() { return 0u; }
};
template <typename DT> struct TailQuad;
template <> struct TailQuad<float> {
    typedef float4 Q;
    static __device__ __forceinline__ Q ld(const float* p) { return ld4(p); }
    static __device__ __forceinline__ float4 f4(Q q) { return q; }
};
template <> struct TailQuad<__half> {
    typedef uint2 Q;
    static __device__ __forceinline__ Q ld(const __half* p) { return *reinterpret_cast<const uint2*>(p); }
    static __device__ __forceinline__ float4 f4(Q q) { return TailRaw<__half>::f4(q); }
};

#ifndef CSPN_TAIL5_UNR
#define CSPN_TAIL5_UNR 2
#endif

template <int NT0, int CNT>
__device__ __forceinline__ void load_tap_range_quad(const float* img, size_t p, size_t HW, float (&out)[CNT][4]) {
#pragma unroll
    for (int j = 0; j < CNT; ++j) {
        const float4 v = ld4(img + (size_t)(NT0 + j) * HW + p);
        out[j][0] = v.x; out[j][1] = v.y; out[j][2] = v.z; out[j][3] = v.w;
    }
}
template <int NT0, int CNT>
__device__ __forceinline__ void load_tap_range_quad(const __half* img, size_t p, size_t HW, float (&out)[CNT][4]) {
    static_assert(NT0 % 2 == 0 && CNT % 2 == 0, "whole tap pairs");
    const size_t pair_stride = 2 * Taps<__half>::hw4(HW);
#pragma unroll
    for (int jp = 0; jp < CNT / 2; ++jp) {
        const uint4 raw = *reinterpret_cast<const uint4*>(img + (size_t)(NT0 / 2 + jp) * pair_stride + 2 * p);
        const float2 a = __half22float2(*reinterpret_cast<const __half2*>(&raw.x));
        const float2 b = __half22float2(*reinterpret_cast<const __half2*>(&raw.y));
        const float2 c = __half22float2(*reinterpret_cast<const __half2*>(&raw.z));
        const float2 d = __half22float2(*reinterpret_cast<const __half2*>(&raw.w));
        out[2 * jp][0] = a.x; out[2 * jp][1] = a.y; out[2 * jp][2] = b.x; out[2 * jp][3] = b.y;
        out[2 * jp + 1][0] = c.x; out[2 * jp + 1][1] = c.y; out[2 * jp + 1][2] = d.x; out[2 * jp + 1][3] = d.y;
    }
}

template <typename DT, typename WT, int VARIANT, int HALF>
__device__ __forceinline__ void tail5_half(const TailArgs& a, float4 (*xdot)[128]) {
    constexpr int K = 5, R = 2, NT = 24, NH = 12, RR = R + 1, WIN = 4 + 2 * R;
    constexpr int DY0 = HALF ? 0 : -R;                  // window row rr holds image row y + DY0 + rr
    constexpr int UNR = CSPN_TAIL5_UNR;
    typedef TailQuad<DT> QD;
    typedef TailPair<DT> PR;
    const int H = a.H, W = a.W, T = a.T;
    const int WQ = W >> 2;
    const size_t HW = (size_t)H * W;
    const size_t plane = (size_t)a.B * HW;
    const size_t nquads = (size_t)a.B * H * WQ;
    const int ql = threadIdx.x & 127;
    const size_t q = (size_t)xcd_contiguous_id(blockIdx.x, gridDim.x) * 128 + ql;
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

    float acc[NH][4];
#pragma unroll
    for (int j = 0; j < NH; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][e] = 0.f;
    float gsum[4] = {0.f, 0.f, 0.f, 0.f};

    // row offsets of the window, clamped to valid rows (invalid ones are zeroed at the consumer), and the patch columns
    unsigned rowoff[RR];
    bool rowok[RR];
#pragma unroll
    for (int rr = 0; rr < RR; ++rr) {
        const int row = y + DY0 + rr;
        rowok[rr] = row >= 0 && row < H;
        rowoff[rr] = (unsigned)((rowok[rr] ? row : y) * W);
    }
    const bool has_left = x > 0, has_right = x + 4 < W;
    const int xl = has_left ? x - 2 : x, xr = has_right ? x + 4 : x;

    for (int t0 = 0; t0 < T; t0 += UNR) {
        float4 Gq[UNR];
        typename QD::Q midq[UNR][RR];
        typename PR::P lf[UNR][RR], rf[UNR][RR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int t = t0 + u;
            const int tc = t < T ? t : T - 1;           // a step past the end re-reads the last one; its G is zeroed below
            const DT* d = ((tc == 0) ? d0 : dh + (size_t)(tc - 1) * plane) + (size_t)b * HW;
            const float* gsrc = (tc == T - 1) ? a.g_T : a.ghist + (size_t)(T - 2 - tc) * plane;      // G_{t+1}
            Gq[u] = (CSPN_TAIL_NT & 2) ? ld4_nt(gsrc + off) : ld4(gsrc + off);
#pragma unroll
            for (int rr = 0; rr < RR; ++rr) {
                const DT* rp = d + rowoff[rr];
                midq[u][rr] = QD::ld(rp + x);
                lf[u][rr] = PR::zero(); rf[u][rr] = PR::zero();
                if (HALF == 1 && rr == 0) {
                    if (fix_right) rf[u][rr] = PR::ld(rp + xr);                     // row 0 of the lower half: taps right of the centre only
                } else if (HALF == 0 && rr == RR - 1) {
                    if (fix_left) lf[u][rr] = PR::ld(rp + xl);                      // row 0 of the upper half: taps left of the centre only
                } else {
                    if (fix_left) lf[u][rr] = PR::ld(rp + xl);
                    if (fix_right) rf[u][rr] = PR::ld(rp + xr);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const bool tv = live && (t0 + u) < T;
            const float g4[4] = {tv ? Gq[u].x : 0.f, tv ? Gq[u].y : 0.f, tv ? Gq[u].z : 0.f, tv ? Gq[u].w : 0.f};
            float win[RR][WIN];
#pragma unroll
            for (int rr = 0; rr < RR; ++rr) {
                const bool rok = rowok[rr];
                const float4 mq = QD::f4(midq[u][rr]);
                const float m4[4] = {rok ? mq.x : 0.f, rok ? mq.y : 0.f, rok ? mq.z : 0.f, rok ? mq.w : 0.f};
#pragma unroll
                for (int c = 0; c < 4; ++c) win[rr][R + c] = m4[c];
                float l0, l1, r0, r1;
                PR::f2(lf[u][rr], l0, l1);
                PR::f2(rf[u][rr], r0, r1);
                const bool lok = rok && has_left, rgt = rok && has_right;
                const float lv[2] = {lok ? l0 : 0.f, lok ? l1 : 0.f}, rv[2] = {rgt ? r0 : 0.f, rgt ? r1 : 0.f};
#pragma unroll
                for (int c = 0; c < R; ++c) {
                    const float l = dpp_from_prev_lane(m4[4 - R + c]);
                    const float r = dpp_from_next_lane(m4[c]);
                    win[rr][c] = fix_left ? lv[c] : l;
                    win[rr][R + 4 + c] = fix_right ? rv[c] : r;
                }
            }
            if (HALF == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) gsum[e] += g4[e];
            }
#pragma unroll
            for (int jl = 0; jl < NH; ++jl) {
                constexpr int KK2 = (K * K) / 2;
                const int j = HALF * NH + jl;
                const int lin = j < KK2 ? j : j + 1;
                const int dy = lin / K - R, dx = lin % K - R;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[jl][e] = fmaf(g4[e], win[dy - DY0][e + dx + R], acc[jl][e]);
            }
        }
    }

    float om[4] = {1.f, 1.f, 1.f, 1.f}, mm[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.sparse && live) {
        const float4 sp = sgn4(ld4(static_cast<const DT*>(a.sparse) + off));
        mm[0] = sp.x; mm[1] = sp.y; mm[2] = sp.z; mm[3] = sp.w;
#pragma unroll
        for (int e = 0; e < 4; ++e) om[e] = 1.f - mm[e];
    }
    if (HALF == 0 && live) {
        const float4 G0 = ld4((T > 0 ? a.ghist + (size_t)(T - 1) * plane : a.g_T) + off);
        const float4 gd = make_float4(G0.x + mm[0] * gsum[0], G0.y + mm[1] * gsum[1], G0.z + mm[2] * gsum[2], G0.w + mm[3] * gsum[3]);
        if (a.gd0_half) st4(static_cast<__half*>(a.gd0) + off, gd);
        else st4(static_cast<float*>(a.gd0) + off, gd);
    }
#pragma unroll
    for (int j = 0; j < NH; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][e] *= om[e];

    if constexpr (VARIANT == 0) {
        if (live) {
            float* gw = static_cast<float*>(a.gout) + (size_t)b * NT * HW + (size_t)y * W + x;
#pragma unroll
            for (int j = 0; j < NH; ++j) st4(gw + (size_t)(HALF * NH + j) * HW, make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]));
        }
    } else {
        const WT* wk = static_cast<const WT*>(a.w) + (size_t)b * Taps<WT>::image_elems(NT, HW);
        float sm[NH][4];
        load_tap_range_quad<HALF * NH, NH>(wk, live ? (size_t)y * W + x : 0, HW, sm);
        float dot[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NH; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) dot[e] = fmaf(sm[j][e], acc[j][e], dot[e]);
        xdot[HALF][ql] = make_float4(dot[0], dot[1], dot[2], dot[3]);
        __syncthreads();
        const float4 od = xdot[1 - HALF][ql];
        // (upper half's sum) + (lower half's sum), in that order in both halves: the two agree bit for bit
        if (HALF == 0) { dot[0] += od.x; dot[1] += od.y; dot[2] += od.z; dot[3] += od.w; }
        else { dot[0] = od.x + dot[0]; dot[1] = od.y + dot[1]; dot[2] = od.z + dot[2]; dot[3] = od.w + dot[3]; }
        if (live) {
            WT* gg = static_cast<WT*>(a.gout) + (size_t)b * NT * HW + (size_t)y * W + x;   // plain [B,NT,H,W] gradient
#pragma unroll
            for (int j = 0; j < NH; ++j)
                st4(gg + (size_t)(HALF * NH + j) * HW, make_float4(sm[j][0] * (acc[j][0] - dot[0]), sm[j][1] * (acc[j][1] - dot[1]),
                                                                   sm[j][2] * (acc[j][2] - dot[2]), sm[j][3] * (acc[j][3] - dot[3])));
        }
    }
}

template <typename DT, typename WT, int VARIANT>
__global__ __launch_bounds__(256) void cspn_grad_tail5(const TailArgs a) {
    static_assert(VARIANT == 0 || VARIANT == 2, "raw dL/dw or the softmax epilogue");
    __shared__ float4 xdot[2][128];
    if (threadIdx.x < 128) tail5_half<DT, WT, VARIANT, 0>(a, xdot);          // wave-uniform: waves 0, 1 / 2, 3
    else tail5_half<DT, WT, VARIANT, 1>(a, xdot);
}

bool tail5_split_enabled() {
    static const int on = [] { const char* e = getenv("CSPN_TAIL5_SPLIT"); return (e && e[0] == '0') ? 0 : 1; }();
    return on != 0;
}

template <int K, typename DT, typename WT>
int launch_tail(const TailArgs& a, int variant, hipStream_t st) {
    const size_t nquads = (size_t)a.B * a.H * (a.W / 4);
#ifndef CSPN_TAIL_BLOCK
#define CSPN_TAIL_BLOCK 256
#endif
    const int grid = (int)((nquads + 255) / 256);
    const int grid3 = (int)((nquads + CSPN_TAIL_BLOCK - 1) / CSPN_TAIL_BLOCK);
    if constexpr (K == 5) {
        if ((variant == 0 || variant == 2) && tail5_split_enabled()) {
            const int grid5 = (int)((nquads + 127) / 128);
            if (variant == 0) hipLaunchKernelGGL((cspn_grad_tail5<DT, WT, 0>), dim3(grid5), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((cspn_grad_tail5<DT, WT, 2>), dim3(grid5), dim3(256), 0, st, a);
            HIP_OK(hipGetLastError());
            return 1;
        }
    }
    if (variant == 0) hipLaunchKernelGGL((cspn_grad_tail<K, DT, WT, 0>), dim3(grid), dim3(256), 0, st, a);
    else if (variant == 2) hipLaunchKernelGGL((cspn_grad_tail<K, DT, WT, 2>), dim3(grid), dim3(256), 0, st, a);
    else if constexpr (K == 3) hipLaunchKernelGGL((cspn_grad_tail<3, DT, WT, 1>), dim3(grid3), dim3(CSPN_TAIL_BLOCK), 0, st, a);
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
    return (a.W % 4 == 0) && aligned16(a.d0) && (!a.dhist || aligned16(a.dhist)) && (!a.ghist || aligned16(a.ghist)) && aligned16(a.g_T) &&
           (!a.sparse || aligned16(a.sparse)) && (!a.w || aligned16(a.w)) && (!a.S || aligned16(a.S)) &&
           (!a.guidance || aligned16(a.guidance)) && aligned16(a.gout) && aligned16(a.gd0) &&
           (a.g_bs % 4 == 0) && (a.g_cs % 4 == 0);
}

}  // namespace

extern "C" {

int cspn_grad_weights(const void* d0, const void* dhist, const float* g_T, const float* ghist, const void* sparse,
                      float* gw, float* gd0, int d_dtype, int B, int H, int W, int K, int T, cspn_stream_t stream) {
    if (!d0 || !g_T || !gw || !gd0 || (T > 1 && !dhist) || (T > 0 && !ghist)) return fail("cspn_grad_weights: NULL pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    {   // vector path: the fused tail without an epilogue (dL/dw written as is)
        TailArgs a{};
        a.d0 = d0; a.dhist = dhist; a.g_T = g_T; a.ghist = ghist; a.sparse = sparse; a.gout = gw; a.gd0 = gd0;
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
                               static_cast<const float*>(d0), static_cast<const float*>(dhist), g_T, ghist,    \
                               static_cast<const float*>(sparse), gw, gd0, B, H, W, T);                        \
        else                                                                                                   \
            hipLaunchKernelGGL((cspn_grad_weights_kernel<KV, __half>), dim3(grid), dim3(256), 0, st,           \
                               static_cast<const __half*>(d0), static_cast<const __half*>(dhist), g_T, ghist,  \
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

int cspn3_backward_tail(const void* d0, const void* dhist, const float* g_T, const float* ghist, const void* sparse,
                        const void* guidance, long bs, long cs, int C, const void* w8, const float* s,
                        void* grad_guidance, float* gd0, int dtype, int B, int H, int W, int T, cspn_stream_t stream) {
    if (!d0 || !g_T || !guidance || !s || !grad_guidance || !gd0 || (T > 1 && !dhist) || (T > 0 && !ghist))
        return fail("cspn3_backward_tail: NULL pointer");
#if !CSPN_TAIL_REBUILD_W
    if (!w8) return fail("cspn3_backward_tail: this build reads the tap volume (CSPN_TAIL_REBUILD_W=0): w8 is required");
#endif
    TailArgs a{};
    a.d0 = d0; a.dhist = dhist; a.g_T = g_T; a.ghist = ghist; a.sparse = sparse; a.w = w8; a.S = s; a.guidance = guidance;
    a.gout = grad_guidance; a.gd0 = gd0; a.g_bs = bs; a.g_cs = cs; a.B = B; a.H = H; a.W = W; a.T = T; a.C = C;
    if (!tail_vector_ok(a)) return fail("cspn3_backward_tail needs W %% 4 == 0 and 16-byte aligned tensors; use "
                                        "cspn_grad_weights + cspn3_grad_guidance");
    return launch_tail_typed<3>(a, 1, dtype, dtype, static_cast<hipStream_t>(stream));
}

int cspn_pac_backward_tail(const void* d0, const void* dhist, const float* g_T, const float* ghist, const void* sparse, const void* wk,
                           void* grad_guided, void* gd0, int gd0_dtype, int d_dtype, int w_dtype, int B, int H, int W, int K, int T,
                           cspn_stream_t stream) {
    if (!d0 || !g_T || !wk || !grad_guided || !gd0 || (T > 1 && !dhist) || (T > 0 && !ghist)) return fail("cspn_pac_backward_tail: NULL pointer");
    TailArgs a{};
    a.d0 = d0; a.dhist = dhist; a.g_T = g_T; a.ghist = ghist; a.sparse = sparse; a.w = wk; a.gout = grad_guided; a.gd0 = gd0;
    a.B = B; a.H = H; a.W = W; a.T = T;
    if (gd0_dtype != CSPN_F32 && gd0_dtype != CSPN_F16) return fail("cspn_pac_backward_tail: bad gd0_dtype %d", gd0_dtype);
    a.gd0_half = gd0_dtype == CSPN_F16;
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

}  // extern "C"
