// pac_conv2d_s2.hip — the pixel-adaptive convolution (network/libs/base/pac.py:89-92 forward, :96-121 backward) for the
// down-sampling geometry: stride 2 x 2, dilation 1, square K in {3, 5}, padding K / 2, W % 8 == 0.  SURVEY.md §8 f-3.
//
// Why its own kernels.  The any-geometry LDS-tiled forward stages a (2*64 + K) x (2*16 + K) input patch per 64 x 16 output
// tile: 17 KB per channel, three channels per batch, two barriers per batch, and 30 % of the threads of a 152 x 114 plane sit
// in ragged edge tiles (0.28 of the HBM peak); the gradients of strided windows had no tiled form at all (0.15).  At stride 2
// the op is a plain stream of the INPUT-sized tensor (4x the output): every input element is used by (K/2)^2 .. ((K+1)/2)^2
// outputs, all of them inside an 8-pixel octet + a 1-2 pixel fringe.  So there is nothing to stage:
//   * a thread owns one quad of 4 consecutive OUTPUT pixels (oy, 4qx..4qx+3) = one aligned octet 8qx..8qx+7 of each of the K
//     input rows 2oy-P .. 2oy-P+K-1 (two 16-byte loads in fp32, one in fp16), flat index over (oy, qx): no tile quantisation;
//   * the fringe (P columns to the left, P-1 to the right) is fetched by the lane itself with scalar loads of lines the
//     neighbouring lanes' octets bring in anyway (row ends are the zero padding); every load of a batch is issued before the
//     first is consumed (see RowRaw);
//   * the window walk is pure register arithmetic with compile-time indices: out[e] += k[i][j][e] * ext_i[2e + j];
//   * dL/dkernel is the same walk with the roles of kernel and grad_out swapped; a shared kernel (kernel_ch = 1) sums over
//     channels in registers, the wavefronts of a workgroup split the channels and meet in LDS in a fixed order (deterministic);
//   * dL/dinput runs on the same thread grid turned round: a thread owns the input rows 2yp, 2yp+1 x the octet 8qx..8qx+7.
//     Tap row i feeds input row parity (i + P) & 1 from grad_out row yp + dy_i, tap column j feeds column parity (j + P) & 1
//     from grad_out column 4qx + e + dx_j with dy, dx in {-1, 0, +1} at compile time: products are formed on the aligned
//     grad_out / kernel quads and shifted by at most one lane.  Every input pixel is written exactly once (no zero fill,
//     no atomics).
// Accumulation is fp32 for both storage types.  A zero of the padding still multiplies the kernel value in the forward
// (0 * inf = NaN, as F.unfold * kernel does); in dL/dinput a window position outside grad_out contributes no term at all.
#include "cspn_common.hpp"

#include <algorithm>

namespace {

using cspn_detail::PacS2Args;

template <int K> struct S2 {
    static constexpr int P = K / 2;
    static constexpr int LH = P;            // fringe columns to the left of the octet
    static constexpr int RH = P - 1;        // ... and to the right
    static constexpr int EXT = 8 + LH + RH;
    static constexpr int NT = K * K;
    // dL/dinput: tap index t (row or column) -> parity of the input coordinate it feeds, offset of the grad_out coordinate
    static constexpr int par(int t) { return (t + P) & 1; }
    static constexpr int del(int t) { return (par(t) + P - t) / 2; }      // exact: the numerator is even
    static constexpr int DMIN = del(K - 1), DMAX = del(0);
    static constexpr int NR = DMAX - DMIN + 1;
};

__device__ __forceinline__ void ld8(const float* p, float (&o)[8]) {
    const float4 a = ld4(p), b = ld4(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void ld8(const __half* p, float (&o)[8]) {
    uint4 raw = *reinterpret_cast<const uint4*>(p);
    asm volatile("" : "+v"(raw.x), "+v"(raw.y), "+v"(raw.z), "+v"(raw.w));     // conversions stay behind the load phase
    const float2 a = __half22float2(*reinterpret_cast<const __half2*>(&raw.x));
    const float2 b = __half22float2(*reinterpret_cast<const __half2*>(&raw.y));
    const float2 c = __half22float2(*reinterpret_cast<const __half2*>(&raw.z));
    const float2 d = __half22float2(*reinterpret_cast<const __half2*>(&raw.w));
    o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y; o[4] = c.x; o[5] = c.y; o[6] = d.x; o[7] = d.y;
}
__device__ __forceinline__ void st8(float* p, const float (&v)[8]) {
    st4(p, make_float4(v[0], v[1], v[2], v[3]));
    st4(p + 4, make_float4(v[4], v[5], v[6], v[7]));
}
__device__ __forceinline__ void st8(__half* p, const float (&v)[8]) {
    uint4 raw;
    *reinterpret_cast<__half2*>(&raw.x) = __floats2half2_rn(v[0], v[1]);
    *reinterpret_cast<__half2*>(&raw.y) = __floats2half2_rn(v[2], v[3]);
    *reinterpret_cast<__half2*>(&raw.z) = __floats2half2_rn(v[4], v[5]);
    *reinterpret_cast<__half2*>(&raw.w) = __floats2half2_rn(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = raw;
}
template <typename T>
__device__ __forceinline__ void ldq(const T* p, float (&v)[4]) {
    const float4 q = ld4(p);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
}

// One input row of the window walk in two phases, so that EVERY load of a batch (all rows, all channels in flight) is issued
// before the first one is consumed: RowRaw holds the raw bits of the aligned octet and of the fringe columns (P to the left,
// P - 1 to the right), each fringe column fetched by the lane itself with a 4- / 2-byte load from a clamped (always valid)
// address — the lines are the ones the neighbouring lanes' octets bring in anyway.  (The first version took the fringe from the
// neighbouring lanes by DPP and patched the first / last lane of a wavefront under a branch: the DPP moves sat right behind each
// row's loads and the patch blocks fenced the scheduler in, so a K-row window cost K serialised round trips per channel —
// `s_waitcnt vmcnt(1)` after every pair of octet loads; the K = 5 forward ran at a quarter of the HBM peak.)
template <typename T> struct Raw8;
template <> struct Raw8<float> { float4 a, b; };
template <> struct Raw8<__half> { uint4 r; };
__device__ __forceinline__ void ld8_raw(const float* p, Raw8<float>& o) { o.a = ld4(p); o.b = ld4(p + 4); }
__device__ __forceinline__ void ld8_raw(const __half* p, Raw8<__half>& o) { o.r = *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ unsigned ld1_raw(const float* p) { return __float_as_uint(*p); }
__device__ __forceinline__ unsigned ld1_raw(const __half* p) { return *reinterpret_cast<const unsigned short*>(p); }
__device__ __forceinline__ void cvt8(const Raw8<float>& r, float (&o)[8]) {
    o[0] = r.a.x; o[1] = r.a.y; o[2] = r.a.z; o[3] = r.a.w; o[4] = r.b.x; o[5] = r.b.y; o[6] = r.b.z; o[7] = r.b.w;
}
__device__ __forceinline__ void cvt8(const Raw8<__half>& r, float (&o)[8]) {
    const float2 a = __half22float2(*reinterpret_cast<const __half2*>(&r.r.x));
    const float2 b = __half22float2(*reinterpret_cast<const __half2*>(&r.r.y));
    const float2 c = __half22float2(*reinterpret_cast<const __half2*>(&r.r.z));
    const float2 d = __half22float2(*reinterpret_cast<const __half2*>(&r.r.w));
    o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y; o[4] = c.x; o[5] = c.y; o[6] = d.x; o[7] = d.y;
}
template <typename T> __device__ __forceinline__ float cvt1(unsigned raw);
template <> __device__ __forceinline__ float cvt1<float>(unsigned raw) { return __uint_as_float(raw); }
template <> __device__ __forceinline__ float cvt1<__half>(unsigned raw) { return __half2float(__ushort_as_half((unsigned short)raw)); }

template <typename T, int K> struct RowRaw {
    Raw8<T> oct;
    unsigned fringe[S2<K>::LH + S2<K>::RH];
};
// `row` is always a valid address (the caller clamps the row index)
template <typename T, int K>
__device__ __forceinline__ void issue_row(const T* row, int qx, int WQ, RowRaw<T, K>& r) {
    typedef S2<K> G;
    ld8_raw(row + 8 * qx, r.oct);
#pragma unroll
    for (int m = 0; m < G::LH; ++m) r.fringe[m] = ld1_raw(row + (qx > 0 ? 8 * qx - G::LH + m : 0));
#pragma unroll
    for (int m = 0; m < G::RH; ++m) r.fringe[G::LH + m] = ld1_raw(row + (qx < WQ - 1 ? 8 * qx + 8 + m : 0));
}
// ext[m] = row[8qx - P + m], 0 where the window sees padding (rok: the row exists)
template <typename T, int K>
__device__ __forceinline__ void finish_row(const RowRaw<T, K>& r, bool rok, int qx, int WQ, float (&ext)[S2<K>::EXT]) {
    typedef S2<K> G;
    float o[8];
    cvt8(r.oct, o);
#pragma unroll
    for (int m = 0; m < 8; ++m) ext[G::LH + m] = rok ? o[m] : 0.f;
#pragma unroll
    for (int m = 0; m < G::LH; ++m) ext[m] = (rok && qx > 0) ? cvt1<T>(r.fringe[m]) : 0.f;
#pragma unroll
    for (int m = 0; m < G::RH; ++m) ext[G::LH + 8 + m] = (rok && qx < WQ - 1) ? cvt1<T>(r.fringe[G::LH + m]) : 0.f;
}

// Workgroup -> (image, spatial block, channel chunk) of the chunked launches.  Every chunk of a spatial block reads the same
// K*K kernel quads per thread; with a (spatial, chunk, image) grid the chunks of one block are 17 workgroups apart and land on
// eight DIFFERENT XCDs (the dispatcher deals workgroups out round-robin), i.e. on eight private L2s: the kernel planes were
// fetched from memory once per chunk — at K = 5 more bytes than the input itself.  Here the linear id is decoded so that the
// chunks of a spatial block are the workgroups L, L + 8, L + 16, ...: same XCD, consecutive in time, the taps of all but the
// first come out of that XCD's L2.
struct Chunked {
    int b, bx, c0;
    bool any;
};
__device__ __forceinline__ Chunked chunked_id(const PacS2Args& a) {
    const int L = blockIdx.x, r = L & 7, q = L >> 3;
    const int y = q % a.nchunk, sg = q / a.nchunk;
    const int sidx = sg * 8 + r;
    Chunked c;
    c.any = sidx < a.gx * a.B;
    c.b = c.any ? sidx / a.gx : 0;
    c.bx = sidx - c.b * a.gx;
    c.c0 = y * a.cchunk;
    return c;
}

struct Where {
    int oy, qx, lane;
    bool live, fixl, fixr;
};
__device__ __forceinline__ Where where_am_i(int qraw, int Ho, int WQ) {
    Where w;
    w.live = qraw < Ho * WQ;
    const int q = w.live ? qraw : 0;
    w.oy = q / WQ;
    w.qx = q - w.oy * WQ;
    w.lane = threadIdx.x & 63;
    w.fixl = w.lane == 0 || w.qx == 0;
    w.fixr = w.lane == 63 || w.qx == WQ - 1;
    return w;
}

// ------------------------------------------------------------------------------------------------ forward
template <typename T, int K, bool SHARED, int CB>
__global__ __launch_bounds__(256) void pac_s2_fwd(const T* __restrict__ in, const T* __restrict__ kern, T* __restrict__ out,
                                                  const PacS2Args a) {
    typedef S2<K> G;
    const Chunked ck = chunked_id(a);
    if (!ck.any) return;
    const Where w = where_am_i(ck.bx * 256 + threadIdx.x, a.Ho, a.WQ);
    const int b = ck.b;
    const int c0 = ck.c0, c1 = min(a.C, c0 + a.cchunk);
    const size_t iplane = (size_t)a.H * a.W, oplane = (size_t)a.Ho * a.Wo;
    const size_t opix = (size_t)w.oy * a.Wo + 4 * w.qx;
    int rowoff[K];
    bool rok[K];
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const int yi = 2 * w.oy - G::P + i;
        rok[i] = yi >= 0 && yi < a.H;
        rowoff[i] = (rok[i] ? yi : 0) * a.W;
    }
    float kv[G::NT][4];
    if constexpr (SHARED) {
#pragma unroll
        for (int t = 0; t < G::NT; ++t) ldq(kern + ((size_t)b * G::NT + t) * oplane + opix, kv[t]);
    }
    for (int c = c0; c < c1; c += CB) {
        RowRaw<T, K> raw[CB][K];
#pragma unroll
        for (int cc = 0; cc < CB; ++cc) {
            const int cu = min(c + cc, c1 - 1);
            const T* pl = in + ((size_t)b * a.C + cu) * iplane;
#pragma unroll
            for (int i = 0; i < K; ++i) issue_row<T, K>(pl + rowoff[i], w.qx, a.WQ, raw[cc][i]);
        }
#pragma unroll
        for (int cc = 0; cc < CB; ++cc) {
            const int cu = min(c + cc, c1 - 1);
            if constexpr (!SHARED) {
#pragma unroll
                for (int t = 0; t < G::NT; ++t) ldq(kern + (((size_t)b * a.C + cu) * G::NT + t) * oplane + opix, kv[t]);
            }
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < K; ++i) {
                float ext[G::EXT];
                finish_row<T, K>(raw[cc][i], rok[i], w.qx, a.WQ, ext);
#pragma unroll
                for (int j = 0; j < K; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = fmaf(kv[i * K + j][e], ext[2 * e + j], acc[e]);
            }
            if (w.live && c + cc < c1)
                st4(out + ((size_t)b * a.C + cu) * oplane + opix, make_float4(acc[0], acc[1], acc[2], acc[3]));
        }
    }
}

// ------------------------------------------------------------------------------------------------ dL/dkernel
// kernel_ch = C: one gradient window per channel, nothing to sum.
template <typename T, int K, int CB>
__global__ __launch_bounds__(256) void pac_s2_gk_perch(const T* __restrict__ gout, const T* __restrict__ in, T* __restrict__ gk,
                                                       const PacS2Args a) {
    typedef S2<K> G;
    const Chunked ck = chunked_id(a);
    if (!ck.any) return;
    const Where w = where_am_i(ck.bx * 256 + threadIdx.x, a.Ho, a.WQ);
    const int b = ck.b;
    const int c0 = ck.c0, c1 = min(a.C, c0 + a.cchunk);
    const size_t iplane = (size_t)a.H * a.W, oplane = (size_t)a.Ho * a.Wo;
    const size_t opix = (size_t)w.oy * a.Wo + 4 * w.qx;
    int rowoff[K];
    bool rok[K];
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const int yi = 2 * w.oy - G::P + i;
        rok[i] = yi >= 0 && yi < a.H;
        rowoff[i] = (rok[i] ? yi : 0) * a.W;
    }
    for (int c = c0; c < c1; c += CB) {
        RowRaw<T, K> raw[CB][K];
        float g[CB][4];
#pragma unroll
        for (int cc = 0; cc < CB; ++cc) {
            const int cu = min(c + cc, c1 - 1);
            const T* pl = in + ((size_t)b * a.C + cu) * iplane;
            ldq(gout + ((size_t)b * a.C + cu) * oplane + opix, g[cc]);
#pragma unroll
            for (int i = 0; i < K; ++i) issue_row<T, K>(pl + rowoff[i], w.qx, a.WQ, raw[cc][i]);
        }
#pragma unroll
        for (int cc = 0; cc < CB; ++cc) {
            if (w.live && c + cc < c1) {
                T* dst = gk + ((size_t)b * a.C + c + cc) * G::NT * oplane + opix;
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    float ext[G::EXT];
                    finish_row<T, K>(raw[cc][i], rok[i], w.qx, a.WQ, ext);
#pragma unroll
                    for (int j = 0; j < K; ++j)
                        st4(dst + (size_t)(i * K + j) * oplane,
                            make_float4(g[cc][0] * ext[j], g[cc][1] * ext[2 + j], g[cc][2] * ext[4 + j], g[cc][3] * ext[6 + j]));
                }
            }
        }
    }
}

// kernel_ch = 1: the window gradient is summed over the channels.  NW wavefronts of a workgroup own the SAME 64 quads and
// take every NW-th group of CB channels; the partial windows meet in LDS and are added in wavefront order by wavefront 0.
template <typename T, int K, int NW, int CB>
__global__ __launch_bounds__(64 * NW) void pac_s2_gk_shared(const T* __restrict__ gout, const T* __restrict__ in,
                                                           T* __restrict__ gk, const PacS2Args a) {
    typedef S2<K> G;
    __shared__ float red[NW > 1 ? NW - 1 : 1][G::NT * 4][64];
    const int wave = threadIdx.x >> 6;
    const Where w = where_am_i(blockIdx.x * 64 + (threadIdx.x & 63), a.Ho, a.WQ);
    const int b = blockIdx.z;
    const size_t iplane = (size_t)a.H * a.W, oplane = (size_t)a.Ho * a.Wo;
    const size_t opix = (size_t)w.oy * a.Wo + 4 * w.qx;
    int rowoff[K];
    bool rok[K];
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const int yi = 2 * w.oy - G::P + i;
        rok[i] = yi >= 0 && yi < a.H;
        rowoff[i] = (rok[i] ? yi : 0) * a.W;
    }
    float acc[G::NT][4];
#pragma unroll
    for (int t = 0; t < G::NT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = 0.f;
    for (int c = wave * CB; c < a.C; c += NW * CB) {
        RowRaw<T, K> raw[CB][K];
        float g[CB][4];
#pragma unroll
        for (int cc = 0; cc < CB; ++cc) {
            const int cu = min(c + cc, a.C - 1);
            const T* pl = in + ((size_t)b * a.C + cu) * iplane;
            ldq(gout + ((size_t)b * a.C + cu) * oplane + opix, g[cc]);
#pragma unroll
            for (int i = 0; i < K; ++i) issue_row<T, K>(pl + rowoff[i], w.qx, a.WQ, raw[cc][i]);
        }
#pragma unroll
        for (int cc = 0; cc < CB; ++cc) {
            const bool cok = c + cc < a.C;
#pragma unroll
            for (int e = 0; e < 4; ++e) g[cc][e] = cok ? g[cc][e] : 0.f;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                float ext[G::EXT];
                finish_row<T, K>(raw[cc][i], rok[i] && cok, w.qx, a.WQ, ext);
#pragma unroll
                for (int j = 0; j < K; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i * K + j][e] = fmaf(g[cc][e], ext[2 * e + j], acc[i * K + j][e]);
            }
        }
    }
    if constexpr (NW > 1) {
        if (wave > 0) {
#pragma unroll
            for (int t = 0; t < G::NT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) red[wave - 1][t * 4 + e][w.lane] = acc[t][e];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int v = 0; v < NW - 1; ++v)
#pragma unroll
            for (int t = 0; t < G::NT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t][e] += red[v][t * 4 + e][w.lane];
    }
    if (!w.live) return;
    T* dst = gk + (size_t)b * G::NT * oplane + opix;
#pragma unroll
    for (int t = 0; t < G::NT; ++t) st4(dst + (size_t)t * oplane, make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]));
}

// ------------------------------------------------------------------------------------------------ dL/dinput
template <typename T, int K, bool SHARED, int CB>
__global__ __launch_bounds__(256) void pac_s2_gi(const T* __restrict__ gout, const T* __restrict__ kern, T* __restrict__ gin,
                                                 const PacS2Args a) {
    typedef S2<K> G;
    const Chunked ck = chunked_id(a);
    if (!ck.any) return;
    const Where w = where_am_i(ck.bx * 256 + threadIdx.x, a.Ho, a.WQ);     // (yp, qx): input rows 2yp, 2yp+1, octet 8qx
    const int yp = w.oy;
    const int b = ck.b;
    const int c0 = ck.c0, c1 = min(a.C, c0 + a.cchunk);
    const size_t iplane = (size_t)a.H * a.W, oplane = (size_t)a.Ho * a.Wo;
    const bool patch_l = w.fixl && w.qx > 0, patch_r = w.fixr && w.qx < a.WQ - 1;     // a neighbour exists, in another wavefront
    size_t rpix[G::NR];                 // grad_out / kernel quad of row yp + DMIN + r (clamped to a valid row)
    bool rok[G::NR];
#pragma unroll
    for (int r = 0; r < G::NR; ++r) {
        const int oy = yp + G::DMIN + r;
        rok[r] = oy >= 0 && oy < a.Ho;
        rpix[r] = (size_t)(rok[r] ? oy : 0) * a.Wo + 4 * w.qx;
    }
    // taps, and for the strip-end lanes the tap values at the neighbouring quad's first / last column
    float kv[G::NT][4], kl[G::NT], kr[G::NT];
    auto load_taps = [&](const T* kb) {
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int r = G::del(i) - G::DMIN;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int t = i * K + j;
                ldq(kb + (size_t)t * oplane + rpix[r], kv[t]);
                kl[t] = kr[t] = 0.f;
                if (G::del(j) < 0 && patch_l) kl[t] = ld1(kb + (size_t)t * oplane + rpix[r] - 1);
                if (G::del(j) > 0 && patch_r) kr[t] = ld1(kb + (size_t)t * oplane + rpix[r] + 4);
            }
        }
    };
    if constexpr (SHARED) load_taps(kern + (size_t)b * G::NT * oplane);
    for (int c = c0; c < c1; c += CB) {
        float g[CB][G::NR][4], gl[CB][G::NR], gr[CB][G::NR];
#pragma unroll
        for (int cc = 0; cc < CB; ++cc) {
            const int cu = min(c + cc, c1 - 1);
            const T* gp = gout + ((size_t)b * a.C + cu) * oplane;
#pragma unroll
            for (int r = 0; r < G::NR; ++r) {
                ldq(gp + rpix[r], g[cc][r]);
                gl[cc][r] = gr[cc][r] = 0.f;
                if (G::DMIN < 0 && patch_l) gl[cc][r] = ld1(gp + rpix[r] - 1);
                if (patch_r) gr[cc][r] = ld1(gp + rpix[r] + 4);
            }
        }
#pragma unroll
        for (int cc = 0; cc < CB; ++cc) {
            const int cu = min(c + cc, c1 - 1);
            if constexpr (!SHARED) load_taps(kern + ((size_t)b * a.C + cu) * G::NT * oplane);
            float acc[2][8];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[r][m] = 0.f;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const int r = G::del(i) - G::DMIN, ry = G::par(i);
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const int t = i * K + j, cx = G::par(j), dx = G::del(j);
                    float pr[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) pr[e] = rok[r] ? g[cc][r][e] * kv[t][e] : 0.f;
                    if (dx == 0) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[ry][2 * e + cx] += pr[e];
                    } else if (dx > 0) {
                        float nx = dpp_from_next_lane(pr[0]);
                        if (w.fixr) nx = (patch_r && rok[r]) ? gr[cc][r] * kr[t] : 0.f;
#pragma unroll
                        for (int e = 0; e < 3; ++e) acc[ry][2 * e + cx] += pr[e + 1];
                        acc[ry][6 + cx] += nx;
                    } else {
                        float pv = dpp_from_prev_lane(pr[3]);
                        if (w.fixl) pv = (patch_l && rok[r]) ? gl[cc][r] * kl[t] : 0.f;
                        acc[ry][cx] += pv;
#pragma unroll
                        for (int e = 1; e < 4; ++e) acc[ry][2 * e + cx] += pr[e - 1];
                    }
                }
            }
            if (w.live && c + cc < c1) {
                T* dst = gin + ((size_t)b * a.C + cu) * iplane + (size_t)(2 * yp) * a.W + 8 * w.qx;
                st8(dst, acc[0]);
                if (2 * yp + 1 < a.H) st8(dst + a.W, acc[1]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ host
#ifndef CSPN_S2_WANT
#define CSPN_S2_WANT 1024      // workgroups a chunked launch aims for (channel chunks are split until there are that many)
#endif
// 1-D grid of the chunked launches: the spatial blocks of all images padded to a multiple of 8, times the chunks (see chunked_id)
inline dim3 chunked_grid(PacS2Args& a, int gx) {
    a.gx = gx;
    a.nchunk = ceil_div(a.C, a.cchunk);
    return dim3((unsigned)(ceil_div(gx * a.B, 8) * 8 * a.nchunk));
}

inline int chunk_for(int C, size_t spatial_blocks, int CB, size_t want) {
    if (spatial_blocks >= want) return C;
    size_t nchunk = (want + spatial_blocks - 1) / spatial_blocks;
    const size_t maxchunk = (size_t)ceil_div(C, CB);
    nchunk = std::min(nchunk, maxchunk);
    const int per = ceil_div(C, (int)nchunk);
    return ceil_div(per, CB) * CB;
}

template <typename T, int K>
int forward_k(const void* in_, const void* kern_, void* out_, PacS2Args a, hipStream_t st) {
    const T* in = static_cast<const T*>(in_);
    const T* kern = static_cast<const T*>(kern_);
    T* out = static_cast<T*>(out_);
    constexpr int CB = K == 3 ? 2 : 1;
    const int gx = ceil_div(a.Ho * a.WQ, 256);
    a.cchunk = chunk_for(a.C, (size_t)gx * a.B, CB, CSPN_S2_WANT);
    const dim3 grid = chunked_grid(a, gx), block(256);
    if (a.CK == 1) CSPN_PRE(st), pac_s2_fwd<T, K, true, CB><<<grid, block, 0, st>>>(in, kern, out, a);
    else CSPN_PRE(st), pac_s2_fwd<T, K, false, CB><<<grid, block, 0, st>>>(in, kern, out, a);
    HIP_OK(hipGetLastError());
    return 1;
}

template <typename T, int K>
int grad_input_k(const void* gout_, const void* kern_, void* gin_, PacS2Args a, hipStream_t st) {
    const T* gout = static_cast<const T*>(gout_);
    const T* kern = static_cast<const T*>(kern_);
    T* gin = static_cast<T*>(gin_);
    constexpr int CB = K == 3 ? 2 : 1;
    const int gx = ceil_div(a.Ho * a.WQ, 256);
    a.cchunk = chunk_for(a.C, (size_t)gx * a.B, CB, CSPN_S2_WANT);
    const dim3 grid = chunked_grid(a, gx), block(256);
    if (a.CK == 1) CSPN_PRE(st), pac_s2_gi<T, K, true, CB><<<grid, block, 0, st>>>(gout, kern, gin, a);
    else CSPN_PRE(st), pac_s2_gi<T, K, false, CB><<<grid, block, 0, st>>>(gout, kern, gin, a);
    HIP_OK(hipGetLastError());
    return 1;
}

template <typename T, int K>
int grad_kernel_k(const void* gout_, const void* in_, void* gk_, PacS2Args a, hipStream_t st) {
    const T* gout = static_cast<const T*>(gout_);
    const T* in = static_cast<const T*>(in_);
    T* gk = static_cast<T*>(gk_);
    if (a.CK == 1) {
        constexpr int NW = K == 3 ? 4 : 2, CB = K == 3 ? 2 : 1;
        const dim3 grid(ceil_div(a.Ho * a.WQ, 64), 1, a.B), block(64 * NW);
        if (a.C == 1) CSPN_PRE(st), pac_s2_gk_shared<T, K, 1, 1><<<grid, dim3(64), 0, st>>>(gout, in, gk, a);
        else CSPN_PRE(st), pac_s2_gk_shared<T, K, NW, CB><<<grid, block, 0, st>>>(gout, in, gk, a);
    } else {
        constexpr int CB = K == 3 ? 2 : 1;
        const int gx = ceil_div(a.Ho * a.WQ, 256);
        a.cchunk = chunk_for(a.C, (size_t)gx * a.B, CB, CSPN_S2_WANT);
        const dim3 grid = chunked_grid(a, gx), block(256);
        CSPN_PRE(st), pac_s2_gk_perch<T, K, CB><<<grid, block, 0, st>>>(gout, in, gk, a);
    }
    HIP_OK(hipGetLastError());
    return 1;
}

}  // namespace

namespace cspn_detail {

bool pac_s2_geometry(int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int W) {
    return sh == 2 && sw == 2 && dh == 1 && dw == 1 && kh == kw && (kh == 3 || kh == 5) && ph == kh / 2 && pw == kw / 2 &&
           W % 8 == 0;
}

int pac_s2_forward(const void* in, const void* kern, void* out, int dtype, int K, const PacS2Args& a, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == CSPN_F16) return K == 3 ? forward_k<__half, 3>(in, kern, out, a, st) : forward_k<__half, 5>(in, kern, out, a, st);
    return K == 3 ? forward_k<float, 3>(in, kern, out, a, st) : forward_k<float, 5>(in, kern, out, a, st);
}
int pac_s2_grad_input(const void* gout, const void* kern, void* gin, int dtype, int K, const PacS2Args& a, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == CSPN_F16)
        return K == 3 ? grad_input_k<__half, 3>(gout, kern, gin, a, st) : grad_input_k<__half, 5>(gout, kern, gin, a, st);
    return K == 3 ? grad_input_k<float, 3>(gout, kern, gin, a, st) : grad_input_k<float, 5>(gout, kern, gin, a, st);
}
int pac_s2_grad_kernel(const void* gout, const void* in, void* gk, int dtype, int K, const PacS2Args& a, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == CSPN_F16)
        return K == 3 ? grad_kernel_k<__half, 3>(gout, in, gk, a, st) : grad_kernel_k<__half, 5>(gout, in, gk, a, st);
    return K == 3 ? grad_kernel_k<float, 3>(gout, in, gk, a, st) : grad_kernel_k<float, 5>(gout, in, gk, a, st);
}

}  // namespace cspn_detail
