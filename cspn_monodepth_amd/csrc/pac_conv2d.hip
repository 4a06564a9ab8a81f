// pac_conv2d.hip — general pixel-adaptive convolution (network/libs/base/pac.py), SURVEY.md §8 f-3.
//
// One step of  out[b,c,y,x] = sum_ij kernel[b,c|0,i,j,y,x] * in0[b,c, y*sh-ph+i*dh, x*sw-pw+j*dw]  for any channel count,
// stride, padding, dilation, plus its two gradients and nd2col.  HBM-bound streaming work: per output pixel the op
// reads kh*kw kernel values (once, they are the bulk), C inputs (re-used kh*kw times out of L1/L2) and writes C
// outputs.  Every thread owns a quad of 4 consecutive output pixels so the kernel planes move as 16-byte accesses;
// channels are walked CC at a time against the same kernel registers.  The K x K recurrence of CSPN_ours
// (C = 1, stride 1, "same" padding, T steps) does NOT come through here — cspn_propagate keeps it in LDS.
#include "cspn_common.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>

namespace {

// Process-wide A/B switch: 1 = skip the LDS-tiled kernels and run the generic one-quad-per-thread kernels.  Initialised
// once from CSPN_PAC_SCALAR=1 in the environment, changed at run time through cspn_pac_force_generic (tests, A/B runs).
std::atomic<int>& force_generic_flag() {
    static std::atomic<int> flag{[] {
        const char* fs = getenv("CSPN_PAC_SCALAR");
        return (fs && fs[0] == '1') ? 1 : 0;
    }()};
    return flag;
}

// workgroups the any-geometry launches aim for (channel chunks / tap groups are split until there are that many): measured on
// the C = 32 dilated row — 512 / 1024 / 2048 workgroups: forward 57 / 47 / 54 us; dL/dkernel with 1 / 2 tap groups: 58 / 88 us
#ifndef CSPN_PAC_NT
#define CSPN_PAC_NT 0          // developer A/B: 1 = non-temporal stores of the tiled kernels' results
#endif
#ifndef CSPN_PAC_WANT_WGS
#define CSPN_PAC_WANT_WGS 1024
#endif
constexpr size_t ANY_WANT_WGS = CSPN_PAC_WANT_WGS, ANY_WANT_WGS_GK = 256;

struct ConvArgs {
    int B, C, CK, H, W, Ho, Wo, WQ;      // WQ = ceil(Wo / 4) output quads per row
    int kh, kw, sh, sw, ph, pw, dh, dw;
    int lead_h, lead_w;                  // transposed nd2col: leading pad (k-1)*d - p on the zero-inserted plane
    int transposed;
    int cchunk;                          // channels per blockIdx.y
    int vec;                             // Wo % 4 == 0 and 16-byte (8 for f16) aligned bases: quad loads/stores
    int force_scalar;                    // CSPN_PAC_SCALAR=1 in the environment: skip the tiled kernels (A/B, tests)
    int lin_tiles, lin_chunks;           // > 0: 1-D grid, decoded by wg_id (the channel chunks of a tile share an XCD)
};

// Workgroup -> (tile, channel chunk, image).  The chunked launches of a SHARED kernel read the tile's kh*kw kernel quads once per
// chunk; on a (tile, chunk, image) grid the chunks of one tile are `tiles` workgroups apart and land on different XCDs (the
// dispatcher deals workgroups out round-robin over the eight XCDs), i.e. on different private L2s — the kernel planes, the bulk of
// the traffic at K = 5 / 7, came from memory once per chunk.  With lin_chunks > 0 the grid is 1-D and decoded so that the chunks
// of a tile are the workgroups L, L + 8, L + 16, ...: same XCD, consecutive in time, every chunk but the first finds the taps in
// that XCD's L2.  (Speed only: any mapping is correct.)
struct WgId {
    int x, y, z;
    bool any;
};
__device__ __forceinline__ WgId wg_id(int lin_tiles, int lin_chunks, int B) {
    WgId w;
    if (lin_chunks <= 0) {
        w.x = blockIdx.x; w.y = blockIdx.y; w.z = blockIdx.z; w.any = true;
        return w;
    }
    const int L = blockIdx.x, r = L & 7, q = L >> 3;
    w.y = q % lin_chunks;
    const int sidx = (q / lin_chunks) * 8 + r;
    w.any = sidx < lin_tiles * B;
    w.z = w.any ? sidx / lin_tiles : 0;
    w.x = sidx - w.z * lin_tiles;
    return w;
}
inline dim3 lin_grid(int tiles, int chunks, int B) { return dim3((unsigned)(((tiles * B + 7) / 8) * 8 * chunks)); }

// source index along one axis for output index o and tap t, or -1 where the window sees a zero
__device__ __forceinline__ int src_plain(int o, int t, int S, int P, int D, int N) {
    const int v = o * S - P + t * D;
    return (unsigned)v < (unsigned)N ? v : -1;
}
__device__ __forceinline__ int src_transposed(int o, int t, int S, int lead, int D, int N) {
    int v = o + t * D - lead;                      // position on the zero-inserted plane (pac.py:53-56)
    if (v < 0 || v % S) return -1;
    v /= S;
    return v < N ? v : -1;
}

// Guarded scalar load: element `off` of `base` if ok, else 0.  A conditional load cannot be speculated, so it becomes an
// exec-masked block of its own.  For fp32 that is fine — the masked load simply completes later, the wait sits at the
// first use.  For fp16 the f16 -> f32 conversion sits INSIDE that block, right behind the load (the compiler sinks it
// there: cvt(0) = 0): one serialised memory round trip per element — 11 per staged patch and thread, 4 per tap of the
// transposed kernel, which is why the fp16 gradients ran at half the fp32 kernels' speed.  The tiled kernels therefore
// load RAW bits under the mask (ld1_raw_or0) and convert where the value is consumed (raw_to_float: an empty asm keeps the
// conversion from being sunk back), i.e. after every load of the batch has been issued.
template <typename I>
__device__ __forceinline__ float ld1_or0(const float* base, I off, bool ok) { return ok ? base[off] : 0.f; }
template <typename I>
__device__ __forceinline__ float ld1_or0(const __half* base, I off, bool ok) { return ok ? __half2float(base[off]) : 0.f; }
template <typename I>
__device__ __forceinline__ unsigned ld1_raw_or0(const float* base, I off, bool ok) { return ok ? __float_as_uint(base[off]) : 0u; }
template <typename I>
__device__ __forceinline__ unsigned ld1_raw_or0(const __half* base, I off, bool ok) {
    return ok ? (unsigned)reinterpret_cast<const unsigned short*>(base)[off] : 0u;
}
template <typename T>
__device__ __forceinline__ float raw_to_float(unsigned raw) {
    if constexpr (std::is_same<T, __half>::value) {
        asm("" : "+v"(raw));
        return __half2float(__ushort_as_half((unsigned short)raw));
    } else {
        return __uint_as_float(raw);
    }
}

template <typename T, bool VEC>
__device__ __forceinline__ void load_quad(const T* p, int x0, int Wo, float (&v)[4]) {
    if constexpr (VEC) {
        const float4 q = ld4(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ld1_or0(p, e, x0 + e < Wo);
    }
}
template <typename T, bool VEC>
__device__ __forceinline__ void store_quad(T* p, int x0, int Wo, const float (&v)[4]) {
    if constexpr (VEC) {
        st4(p, make_float4(v[0], v[1], v[2], v[3]));
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (x0 + e < Wo) st1(p + e, v[e]);
    }
}

constexpr int CC = 4;    // channels accumulated against one set of kernel registers

// ------------------------------------------------------------------------------------------------ forward
template <typename T, bool VEC, bool SHARED>
__global__ __launch_bounds__(256) void pac_conv2d_fwd(const T* __restrict__ in, const T* __restrict__ kern,
                                                      T* __restrict__ out, ConvArgs a) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= a.Ho * a.WQ) return;
    const int y = q / a.WQ, x0 = (q - y * a.WQ) * 4;
    const int b = blockIdx.z;
    const int c_begin = blockIdx.y * a.cchunk, c_end = min(a.C, c_begin + a.cchunk);
    const size_t oplane = (size_t)a.Ho * a.Wo, iplane = (size_t)a.H * a.W;
    const size_t opix = (size_t)y * a.Wo + x0;
    const int ntap = a.kh * a.kw;
    int xs[4];                                        // x*sw - pw for the four pixels
#pragma unroll
    for (int e = 0; e < 4; ++e) xs[e] = (x0 + e) * a.sw - a.pw;
    for (int c = c_begin; c < c_end; c += CC) {
        const int nc = min(CC, c_end - c);
        float acc[CC][4];
#pragma unroll
        for (int cc = 0; cc < CC; ++cc)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[cc][e] = 0.f;
        const T* inb = in + ((size_t)b * a.C + c) * iplane;
        for (int i = 0; i < a.kh; ++i) {
            const int yi = src_plain(y, i, a.sh, a.ph, a.dh, a.H);
            for (int j = 0; j < a.kw; ++j) {
                const int tap = i * a.kw + j;
                int xi[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int v = xs[e] + j * a.dw;
                    xi[e] = (yi >= 0 && (unsigned)v < (unsigned)a.W) ? v : -1;
                }
                float kv[4];
                if constexpr (SHARED) load_quad<T, VEC>(kern + ((size_t)b * ntap + tap) * oplane + opix, x0, a.Wo, kv);
#pragma unroll
                for (int cc = 0; cc < CC; ++cc) {
                    if (cc < nc) {
                        if constexpr (!SHARED)
                            load_quad<T, VEC>(kern + (((size_t)b * a.C + c + cc) * ntap + tap) * oplane + opix, x0, a.Wo, kv);
                        const T* row = inb + (size_t)cc * iplane + (size_t)(yi < 0 ? 0 : yi) * a.W;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            // a zero of the padding still multiplies the kernel value (0 * inf = NaN, as F.unfold * kernel)
                            const float v = ld1_or0(row, xi[e], xi[e] >= 0);
                            acc[cc][e] = fmaf(kv[e], v, acc[cc][e]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int cc = 0; cc < CC; ++cc)
            if (cc < nc) store_quad<T, VEC>(out + ((size_t)b * a.C + c + cc) * oplane + opix, x0, a.Wo, acc[cc]);
    }
}

// ------------------------------------------------------------------------------------------------ forward / dL/dinput, tiled
// Stride 1, dilation 1, square K in {3,5,7} (any padding): the case every model in the reference uses, and the one
// where the scalar kernels are bound by their K*K cached loads per channel rather than by HBM.  A workgroup owns a
// 64 x 16 tile of the destination plane (one quad per thread).  Channels go through LDS CB at a time: the
// (64+K-1) x (16+K-1) source patches of the NEXT batch are fetched into registers before the current batch is
// computed and committed to the other LDS buffer afterwards (one barrier per batch, global latency hidden behind the
// FMAs); window rows come back as aligned ds_read_b128.  Kernel taps: shared kernel and K <= 5 — loaded once,
// resident in registers for every channel (HOIST); K = 7 — one tap row at a time inside the row loop, each row
// applied to the whole batch.  CB = 1 serves single-channel chunks (fewer registers: the C = 1 case is a pure stream
// and lives on occupancy).
//
//   forward    : source = input,    destination = out [Ho,Wo]; tap (i,j) of destination pixel q is kernel[i,j] AT q.
//   TRANSPOSED : source = grad_out, destination = grad_input [H,W]; the fold of pac.py:104-113 as a gather — input
//                pixel q receives grad_out[s] * kernel[K-1-i', K-1-j'][s] from the source pixel s = q + (i',j') + origin,
//                so the taps are read where the SOURCE pixel lives (unaligned: 4 scalar loads per tap).
constexpr int TILE_W = 64, TILE_H = 16;

struct TiledArgs {
    int B, C, CK, cchunk;
    int src_h, src_w;        // plane the LDS patches come from
    int dst_h, dst_w;        // plane the tile lives on
    int org_y, org_x;        // patch origin relative to the tile origin (forward: -pad; transposed: pad - (K-1))
    int k_h, k_w;            // kernel planes: always [Ho, Wo]
    int k_vec, dst_vec;      // aligned quads possible on the kernel planes (forward only) / the destination
    int tiles_x;
    int lin_tiles, lin_chunks;   // > 0: 1-D grid, decoded by wg_id
};

// NBUF = 1: the whole channel chunk is ONE batch (cchunk <= CB, the host's promise) — no second LDS buffer, twice the workgroups per CU
template <typename T, int K, bool HOIST, int CB, bool TRANSPOSED, int NBUF = 2>
__global__ __launch_bounds__(256, (K > 5 && (CB <= 4 || NBUF == 1) ? 2 : 1)) void pac_conv2d_tiled(const T* __restrict__ src, const T* __restrict__ kern,
                                                                         T* __restrict__ dst, TiledArgs a) {
    constexpr int RW = TILE_W + ((K - 1 + 3) & ~3);     // LDS row pitch, a multiple of 4
    constexpr int RH = TILE_H + K - 1;
    constexpr int PATCH = RH * RW;
    constexpr int NLD = (PATCH + 255) / 256;            // patch elements staged per thread and channel
    constexpr int NQUAD = (K + 3 + 3) / 4;              // aligned quads covering the K+3 window columns
    constexpr bool ROWWISE = K > 5;
    constexpr int KR = ROWWISE ? K : K * K;
    constexpr int ROW_UNROLL = ROWWISE ? 1 : K;         // row-wise: a real loop, or the tap loads are hoisted and spill
    __shared__ __attribute__((aligned(16))) float tile[NBUF][CB][PATCH];
    const WgId wg = wg_id(a.lin_tiles, a.lin_chunks, a.B);
    if (!wg.any) return;
    const int tid = wg.x;
    const int ty = tid / a.tiles_x, tx = tid - ty * a.tiles_x;
    const int tx0 = tx * TILE_W, ty0 = ty * TILE_H;
    const int qx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int x0 = tx0 + 4 * qx, y = ty0 + ly;
    const bool live = y < a.dst_h && x0 < a.dst_w;
    const int b = wg.z;
    const int c_begin = wg.y * a.cchunk, c_end = min(a.C, c_begin + a.cchunk);
    const size_t kplane = (size_t)a.k_h * a.k_w, splane = (size_t)a.src_h * a.src_w, dplane = (size_t)a.dst_h * a.dst_w;
    const size_t dpix = (size_t)y * a.dst_w + x0;

    int goff[NLD];                                      // patch element -> offset in the source plane, -1 = zero
#pragma unroll
    for (int n = 0; n < NLD; ++n) {
        const int idx = threadIdx.x + 256 * n;
        const int ry = idx / RW, rx = idx - ry * RW;
        const int yi = ty0 + a.org_y + ry, xi = tx0 + a.org_x + rx;
        goff[n] = (idx < PATCH && (unsigned)yi < (unsigned)a.src_h && (unsigned)xi < (unsigned)a.src_w) ? yi * a.src_w + xi : -1;
    }
    unsigned pre[CB][NLD];                              // raw bits: converted at commit time (see ld1_raw_or0)
    auto fetch = [&](int c0) {
#pragma unroll
        for (int cc = 0; cc < CB; ++cc) {
            const T* sp = src + ((size_t)b * a.C + min(c0 + cc, a.C - 1)) * splane;
#pragma unroll
            for (int n = 0; n < NLD; ++n) pre[cc][n] = ld1_raw_or0(sp, goff[n], c0 + cc < c_end && goff[n] >= 0);
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int cc = 0; cc < CB; ++cc)
#pragma unroll
            for (int n = 0; n < NLD; ++n)
                if (256 * (n + 1) <= PATCH || threadIdx.x + 256 * n < PATCH) tile[buf][cc][threadIdx.x + 256 * n] = raw_to_float<T>(pre[cc][n]);
    };
    float kr[KR][4];
    // taps [first, first+n) in window order (row-major over (i',j')) of kernel channel kc -> kr[0..n)
    auto load_taps = [&](auto& kr, int kc, int first, int n) {
        const T* kb = kern + ((size_t)b * a.CK + kc) * (K * K) * kplane;
        if constexpr (!TRANSPOSED) {
            const T* kp = kb + (size_t)first * kplane + dpix;
#pragma unroll
            for (int t = 0; t < n; ++t) {
                if (a.k_vec) load_quad<T, true>(kp + (size_t)t * kplane, x0, a.k_w, kr[t]);
                else load_quad<T, false>(kp + (size_t)t * kplane, x0, a.k_w, kr[t]);
            }
        } else {
#pragma unroll
            for (int t = 0; t < n; ++t) {
                const int w = first + t, wi = w / K, wj = w - wi * K;      // window position (i',j'); constants after unrolling
                const int sy = y + a.org_y + wi;
                const bool rowok = (unsigned)sy < (unsigned)a.k_h;
                const T* kp = kb + (size_t)(K * K - 1 - w) * kplane + (size_t)(rowok ? sy : 0) * a.k_w;   // flipped tap, (safe) source row
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int sx = x0 + e + a.org_x + wj;
                    kr[t][e] = __uint_as_float(ld1_raw_or0(kp, sx, rowok && (unsigned)sx < (unsigned)a.k_w));
                }
            }
            if constexpr (std::is_same<T, __half>::value) {        // all n taps are in flight: now the raw halfs become floats
#pragma unroll
                for (int t = 0; t < n; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) kr[t][e] = raw_to_float<T>(__float_as_uint(kr[t][e]));
            }
        }
    };
    auto window = [&](int buf, int cc, int row, float (&win)[4 * NQUAD]) {
#pragma unroll
        for (int n = 0; n < NQUAD; ++n) {
            const v4f v = *(lds_cv4f_ptr)(&tile[buf][cc][row * RW + 4 * qx + 4 * n]);
            win[4 * n] = v.x; win[4 * n + 1] = v.y; win[4 * n + 2] = v.z; win[4 * n + 3] = v.w;
        }
    };
    auto store_dst = [&](int c, const float (&acc)[4]) {
        T* op = dst + ((size_t)b * a.C + c) * dplane + dpix;
#if CSPN_PAC_NT & 1
        if constexpr (std::is_same<T, float>::value) {
            if (a.dst_vec) { const v4f v = {acc[0], acc[1], acc[2], acc[3]}; __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(op)); return; }
        }
#endif
        if (a.dst_vec) store_quad<T, true>(op, x0, a.dst_w, acc);
        else store_quad<T, false>(op, x0, a.dst_w, acc);
    };

    if (HOIST && live) load_taps(kr, 0, 0, K * K);
    fetch(c_begin);
    commit(0);
    __syncthreads();
    int buf = 0;
    for (int c = c_begin; c < c_end; c += CB) {
        const bool more = NBUF > 1 && c + CB < c_end;
        if (more) fetch(c + CB);
        if (live) {
            if constexpr (ROWWISE) {
                float acc[CB][4];
#pragma unroll
                for (int cc = 0; cc < CB; ++cc)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[cc][e] = 0.f;
                // the eight-channel batches (always a shared kernel) double-buffer the tap rows: row i + 1 is requested before
                // row i is applied.  Compile-time: the 28 extra registers cost the four-channel transposed instance its
                // spill-free form.
                constexpr bool PREF = CB == 8;
                float kn[PREF ? KR : 1][4];
                if (PREF) load_taps(kr, 0, 0, K);
#pragma unroll 1
                for (int i = 0; i < K; ++i) {
                    if constexpr (PREF) {
                        if (i + 1 < K) load_taps(kn, 0, (i + 1) * K, K);
                    } else {
                        if (a.CK == 1) load_taps(kr, 0, i * K, K);
                    }
#pragma unroll
                    for (int cc = 0; cc < CB; ++cc) {
                        if (c + cc < c_end) {
                            if (!PREF && a.CK != 1) load_taps(kr, c + cc, i * K, K);
                            float win[4 * NQUAD];
                            window(buf, cc, ly + i, win);
#pragma unroll
                            for (int j = 0; j < K; ++j)
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[cc][e] = fmaf(kr[j][e], win[j + e], acc[cc][e]);
                        }
                    }
                    if constexpr (PREF) {
                        if (i + 1 < K) {
#pragma unroll
                            for (int j = 0; j < K; ++j)
#pragma unroll
                                for (int e = 0; e < 4; ++e) kr[j][e] = kn[j][e];
                        }
                    }
                }
#pragma unroll
                for (int cc = 0; cc < CB; ++cc)
                    if (c + cc < c_end) store_dst(c + cc, acc[cc]);
            } else {
#pragma unroll
                for (int cc = 0; cc < CB; ++cc) {
                    if (c + cc < c_end) {
                        if (!HOIST) load_taps(kr, a.CK == 1 ? 0 : c + cc, 0, K * K);
                        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll ROW_UNROLL
                        for (int i = 0; i < K; ++i) {
                            float win[4 * NQUAD];
                            window(buf, cc, ly + i, win);
#pragma unroll
                            for (int j = 0; j < K; ++j)
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[e] = fmaf(kr[i * K + j][e], win[j + e], acc[e]);
                        }
                        store_dst(c + cc, acc);
                    }
                }
            }
        }
        if (more) commit((buf + 1) % NBUF);
        __syncthreads();
        buf = (buf + 1) % NBUF;
    }
}

// ------------------------------------------------------------------------------------------------ forward, tiled, fp16 octs
// The fp16 form of pac_conv2d_tiled (forward, K <= 5): in fp16 the quad kernel moves its taps as 8-byte loads and runs at
// the fp32 kernel's instruction rate, i.e. at half its byte rate (30 % of the HBM peak).  Here a thread owns EIGHT
// consecutive output pixels: every tap plane is one 16-byte load per thread, the taps stay packed (two halfs per VGPR:
// K*K*4 registers, as many as the fp32 quad kernel uses for half the pixels) and feed v_fma_mix_f32 directly (f16 tap
// x f32 window value + f32 accumulator: the same arithmetic as converting the tap first), the eight results leave as
// one 16-byte store.  Tile 128 x 16 per workgroup, channels double-buffered through LDS exactly like pac_conv2d_tiled.
// (fma_h8: cspn_common.hpp)
constexpr int TILE_W8 = 128;

// HOIST (shared kernel, several channels): all K*K taps are loaded once and stay in registers for every channel.
// Otherwise: one tap ROW at a time, double-buffered (row i+1 is requested before row i is applied; row 0 before the
// source patch is staged) — 8 K tap registers instead of 4 K*K, so that five wavefronts per SIMD are resident and a
// single-channel launch (the CSPN_ours step: a pure stream) fits the chip in one round.
template <int K, bool HOIST>
__global__ __launch_bounds__(256, (HOIST ? 1 : 4)) void pac_conv2d_tiled_h8(const __half* __restrict__ src, const __half* __restrict__ kern,
                                                           __half* __restrict__ dst, TiledArgs a) {
    constexpr int RW = TILE_W8 + ((K - 1 + 3) & ~3);
    constexpr int RH = TILE_H + K - 1;
    constexpr int PATCH = RH * RW;
    constexpr int NLD = (PATCH + 255) / 256;
    constexpr int NQUAD = (K + 7 + 3) / 4;              // aligned quads covering the K+7 window columns
    __shared__ __attribute__((aligned(16))) float tile[2][PATCH];
    const WgId wg = wg_id(a.lin_tiles, a.lin_chunks, a.B);
    if (!wg.any) return;
    const int tid = wg.x;
    const int ty = tid / a.tiles_x, tx = tid - ty * a.tiles_x;
    const int tx0 = tx * TILE_W8, ty0 = ty * TILE_H;
    const int ox = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int x0 = tx0 + 8 * ox, y = ty0 + ly;
    const bool live = y < a.dst_h && x0 < a.dst_w;      // dst_w % 8 == 0 (launcher): an oct is inside or outside as a whole
    const int b = wg.z;
    const int c_begin = wg.y * a.cchunk, c_end = min(a.C, c_begin + a.cchunk);
    const size_t kplane = (size_t)a.k_h * a.k_w, splane = (size_t)a.src_h * a.src_w, dplane = (size_t)a.dst_h * a.dst_w;
    const size_t dpix = (size_t)y * a.dst_w + x0;

    // patch element -> offset in the source plane, -1 = zero.  Kept in registers across the channels only by the HOIST
    // instance; the row-wise instance recomputes it per fetch (it lives on occupancy: 11 registers matter)
    auto patch_off = [&](int n) -> int {
        const int idx = threadIdx.x + 256 * n;
        const int ry = idx / RW, rx = idx - ry * RW;
        const int yi = ty0 + a.org_y + ry, xi = tx0 + a.org_x + rx;
        return (idx < PATCH && (unsigned)yi < (unsigned)a.src_h && (unsigned)xi < (unsigned)a.src_w) ? yi * a.src_w + xi : -1;
    };
    int goff[HOIST ? NLD : 1];
    if (HOIST) {
#pragma unroll
        for (int n = 0; n < NLD; ++n) goff[HOIST ? n : 0] = patch_off(n);
    }
    unsigned pre[NLD];                                  // raw halfs: converted at commit time (see ld1_raw_or0)
    auto fetch = [&](int c) {
        const __half* sp = src + ((size_t)b * a.C + min(c, a.C - 1)) * splane;
#pragma unroll
        for (int n = 0; n < NLD; ++n) {
            const int o = HOIST ? goff[HOIST ? n : 0] : patch_off(n);
            pre[n] = ld1_raw_or0(sp, o, c < c_end && o >= 0);
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int n = 0; n < NLD; ++n)
            if (256 * (n + 1) <= PATCH || threadIdx.x + 256 * n < PATCH) tile[buf][threadIdx.x + 256 * n] = raw_to_float<__half>(pre[n]);
    };
    constexpr int KR = HOIST ? K * K : 2 * K;
    uint4 kr[KR];
    auto load_taps = [&](int kc, int first, int n, int at) {   // taps [first, first + n) of kernel channel kc -> kr[at ..]
        const __half* kp = kern + (((size_t)b * a.CK + kc) * (K * K) + first) * kplane + dpix;
#pragma unroll
        for (int t = 0; t < n; ++t) kr[at + t] = *reinterpret_cast<const uint4*>(kp + (size_t)t * kplane);
    };

    if (live) {
        if (HOIST) load_taps(0, 0, K * K, 0);
        else load_taps(a.CK == 1 ? 0 : c_begin, 0, K, 0);
    }
    fetch(c_begin);
    commit(0);
    __syncthreads();
    int buf = 0;
    for (int c = c_begin; c < c_end; ++c) {
        const bool more = c + 1 < c_end;
        if (more) fetch(c + 1);
        if (live) {
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                if (!HOIST) {
                    if (i + 1 < K) load_taps(a.CK == 1 ? 0 : c, (i + 1) * K, K, ((i + 1) & 1) * K);
                    // K is odd: the last row sits in half 0, so row 0 of the NEXT channel is prefetched into the free half 1
                    // and moved to half 0 (where every channel expects its row 0) at the channel boundary below
                    else if (more) load_taps(a.CK == 1 ? 0 : c + 1, 0, K, K);
                }
                float win[4 * NQUAD];
#pragma unroll
                for (int n = 0; n < NQUAD; ++n) {
                    const v4f v = *(lds_cv4f_ptr)(&tile[buf][(ly + i) * RW + 8 * ox + 4 * n]);
                    win[4 * n] = v.x; win[4 * n + 1] = v.y; win[4 * n + 2] = v.z; win[4 * n + 3] = v.w;
                }
#pragma unroll
                for (int j = 0; j < K; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] = fma_h8(kr[HOIST ? i * K + j : (i & 1) * K + j], e, win[j + e], acc[e]);
                if (!HOIST) __builtin_amdgcn_sched_barrier(0);   // one row of taps ahead, not all of them (registers = occupancy)
            }
            uint4 o;
            *reinterpret_cast<__half2*>(&o.x) = __floats2half2_rn(acc[0], acc[1]);
            *reinterpret_cast<__half2*>(&o.y) = __floats2half2_rn(acc[2], acc[3]);
            *reinterpret_cast<__half2*>(&o.z) = __floats2half2_rn(acc[4], acc[5]);
            *reinterpret_cast<__half2*>(&o.w) = __floats2half2_rn(acc[6], acc[7]);
            *reinterpret_cast<uint4*>(dst + ((size_t)b * a.C + c) * dplane + dpix) = o;
            if (!HOIST && more) {
                static_assert(K & 1, "the row double-buffer assumes an odd K");
#pragma unroll
                for (int t = 0; t < K; ++t) kr[t] = kr[K + t];
            }
        }
        if (more) commit(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
}

// Staging of a [nc][RH * RW] patch for the any-geometry kernels: patch element idx = threadIdx.x + 256 n -> (ry, rx),
// advanced incrementally (integer divisions by the runtime patch width per element had made the forward VALU-bound: 6.6 k
// VALU instructions per wavefront).  Four elements x nc channels are REQUESTED before the first is written to LDS: with one
// element per trip a wavefront had nc loads in flight, ~19 KB per CU at the occupancy of these kernels — 3 TB/s whatever
// the rest of the kernel did.  Raw bits under the mask, converted at the LDS write (see ld1_raw_or0).
template <typename T>
__device__ __forceinline__ void stage_patch_any(float* __restrict__ patch, const T* __restrict__ src, size_t splane, int nc,
                                                int psz, int RW, int gy0, int gx0, int src_h, int src_w, int t_ry, int t_rx,
                                                int step_ry, int step_rx) {
    constexpr int U = 4;
    int ry = t_ry, rx = t_rx;
    for (int idx0 = threadIdx.x; idx0 < psz; idx0 += 256 * U) {
        unsigned v[U][CC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int yi = gy0 + ry, xi = gx0 + rx;
            const bool ok = idx0 + 256 * u < psz && (unsigned)yi < (unsigned)src_h && (unsigned)xi < (unsigned)src_w;
            const size_t goff = (size_t)(ok ? yi : 0) * src_w + (ok ? xi : 0);
#pragma unroll
            for (int cc = 0; cc < CC; ++cc) v[u][cc] = (cc < nc) ? ld1_raw_or0(src + cc * splane, goff, ok) : 0u;
            rx += step_rx; ry += step_ry;
            if (rx >= RW) { rx -= RW; ++ry; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = idx0 + 256 * u;
            if (idx < psz) {
#pragma unroll
                for (int cc = 0; cc < CC; ++cc)
                    if (cc < nc) patch[cc * psz + idx] = raw_to_float<T>(v[u][cc]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ forward, tiled, any geometry
// Strided / dilated / non-square windows: same 64 x 16 output tile, the input patch it touches —
// ((64-1) sw + (kw-1) dw + 1) x ((16-1) sh + (kh-1) dh + 1) — staged in dynamic LDS for `cb` channels at a time and the
// taps read back as scalar ds_read_b32 (the window is not contiguous).  Thread t owns column x = t % 64 on the four
// rows (t / 64) + 4 e: a wavefront reads 64 consecutive patch columns (stride sw), which is bank-conflict free for
// sw = 1 — the quad-per-thread mapping of the other kernels puts 64 lanes on 8 banks here (measured 2.4x slower).
// The price is 4-byte kernel loads and output stores (coalesced: one 256-byte row segment per wavefront).
// TRANSPOSED (unit stride only) is dL/dinput as a gather: the tile lies on the INPUT plane, the patch holds grad_out, tap
// (i, j) of input pixel q looks at the output pixel p = q + pad - (i, j) * dilation — the patch is read at the flipped tap
// position and the kernel value is fetched where p lives (one coalesced, shifted row segment per wavefront and tap).
// HOIST (shared kernel of <= 9 taps): the tile's kernel values are loaded once and stay in registers across all channel batches.
template <typename T, bool SHARED, bool TRANSPOSED, bool HOIST>
__global__ __launch_bounds__(256) void pac_conv2d_fwd_tiled_any(const T* __restrict__ in, const T* __restrict__ kern,
                                                                T* __restrict__ out, ConvArgs a, int tiles_x, int RW,
                                                                int RH, int cb) {
    extern __shared__ __attribute__((aligned(16))) float patch[];       // [cb][RH * RW]
    const WgId wg = wg_id(a.lin_tiles, a.lin_chunks, a.B);
    if (!wg.any) return;
    const int tid = wg.x;
    const int ty = tid / tiles_x, tx = tid - ty * tiles_x;
    const int tx0 = tx * TILE_W, ty0 = ty * TILE_H;
    const int lx = threadIdx.x & 63, ly0 = threadIdx.x >> 6;
    const int x = tx0 + lx;
    const int b = wg.z;
    const int c_begin = wg.y * a.cchunk, c_end = min(a.C, c_begin + a.cchunk);
    const int src_h = TRANSPOSED ? a.Ho : a.H, src_w = TRANSPOSED ? a.Wo : a.W;     // the plane the patch is cut from
    const int dst_h = TRANSPOSED ? a.H : a.Ho, dst_w = TRANSPOSED ? a.W : a.Wo;     // the plane the tile lies on
    const size_t kplane = (size_t)a.Ho * a.Wo, splane = (size_t)src_h * src_w, dplane = (size_t)dst_h * dst_w;
    const int psz = RH * RW, ntap = a.kh * a.kw;
    // source coordinates of patch element (0,0)
    const int gy0 = TRANSPOSED ? ty0 + a.ph - (a.kh - 1) * a.dh : ty0 * a.sh - a.ph;
    const int gx0 = TRANSPOSED ? tx0 + a.pw - (a.kw - 1) * a.dw : tx0 * a.sw - a.pw;
    bool live[4];
    size_t opix[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int y = ty0 + ly0 + 4 * e;
        live[e] = x < dst_w && y < dst_h;
        opix[e] = (size_t)(live[e] ? y : 0) * dst_w + (live[e] ? x : 0);
    }
    const int step_ry = 256 / RW, step_rx = 256 - step_ry * RW;         // uniform: one division per workgroup
    const int t_ry = threadIdx.x / RW, t_rx = threadIdx.x - t_ry * RW;   // one division per thread
    const int base0 = TRANSPOSED ? ly0 * RW + lx : ly0 * a.sh * RW + lx * a.sw;
    const int estep = TRANSPOSED ? 4 * RW : 4 * a.sh * RW;
    // kernel value of tap (ti, tj) for the thread's four pixels: at the output pixel itself, or (transposed) at the output
    // pixel p that reads input pixel q through this tap — where there is no such p the value is 0 and so is the patch
    // element it multiplies (p lies outside grad_out)
    auto load_taps = [&](const T* kp, int ti, int tj, float (&kv)[4]) {
        if constexpr (TRANSPOSED) {
            const int kx = x + a.pw - tj * a.dw;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ky = ty0 + ly0 + 4 * e + a.ph - ti * a.dh;
                const bool ok = live[e] && (unsigned)ky < (unsigned)a.Ho && (unsigned)kx < (unsigned)a.Wo;
                kv[e] = ld1_or0(kp, ok ? (size_t)ky * a.Wo + kx : (size_t)0, ok);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) kv[e] = ld1_or0(kp, opix[e], live[e]);
        }
    };
    constexpr int NH = 9;
    float kh_[HOIST ? NH : 1][4];
    if constexpr (HOIST) {
        int ti = 0, tj = 0;
#pragma unroll
        for (int k = 0; k < NH; ++k) {
            if (k < ntap) {
                load_taps(kern + ((size_t)b * ntap + k) * kplane, ti, tj, kh_[k]);
                if (++tj == a.kw) { tj = 0; ++ti; }
            }
        }
    }
    for (int c = c_begin; c < c_end; c += cb) {
        const int nc = min(cb, c_end - c);
        __syncthreads();
        stage_patch_any<T>(patch, in + ((size_t)b * a.C + c) * splane, splane, nc, psz, RW, gy0, gx0, src_h, src_w, t_ry, t_rx,
                           step_ry, step_rx);
        __syncthreads();
        float acc[CC][4];
#pragma unroll
        for (int cc = 0; cc < CC; ++cc)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[cc][e] = 0.f;
        int ti = 0, tj = 0;
        auto one_tap = [&](int tap, const float (*hoisted)[4]) {
            const int base = TRANSPOSED ? base0 + (a.kh - 1 - ti) * a.dh * RW + (a.kw - 1 - tj) * a.dw
                                        : base0 + ti * a.dh * RW + tj * a.dw;
            float kv[4];
            if constexpr (HOIST) {
#pragma unroll
                for (int e = 0; e < 4; ++e) kv[e] = (*hoisted)[e];
            } else if constexpr (SHARED) {
                load_taps(kern + ((size_t)b * ntap + tap) * kplane, ti, tj, kv);
            }
#pragma unroll
            for (int cc = 0; cc < CC; ++cc) {
                if (cc < nc) {
                    if constexpr (!SHARED) load_taps(kern + (((size_t)b * a.C + c + cc) * ntap + tap) * kplane, ti, tj, kv);
                    const float* pp = patch + cc * psz + base;
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[cc][e] = fmaf(kv[e], pp[e * estep], acc[cc][e]);
                }
            }
            if (++tj == a.kw) { tj = 0; ++ti; }
        };
        if constexpr (HOIST) {
#pragma unroll
            for (int k = 0; k < NH; ++k)
                if (k < ntap) one_tap(k, &kh_[k]);
        } else {
#pragma unroll 2
            for (int tap = 0; tap < ntap; ++tap) one_tap(tap, nullptr);
        }
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) {
            if (cc < nc) {
                T* op = out + ((size_t)b * a.C + c + cc) * dplane;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (live[e]) st1(op + opix[e], acc[cc][e]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ dL/dkernel, tiled, any geometry
// The input patch of a 64 x 16 output tile staged as in pac_conv2d_fwd_tiled_any; every thread turns the grad_out values of
// its four pixels into the kernel gradients of `tpg` taps.  A shared kernel sums over ALL channels in registers (NTM taps at
// most per workgroup: blockIdx.y = tap group) and writes once; a per-channel kernel is an outer product (blockIdx.y =
// channel chunk, every tap written as it is formed).
template <typename T, bool SHARED, int NTM>
__global__ __launch_bounds__(256) void pac_conv2d_gk_any(const T* __restrict__ gout, const T* __restrict__ in,
                                                         T* __restrict__ gk, ConvArgs a, int tiles_x, int RW, int RH,
                                                         int cb, int tpg) {
    extern __shared__ __attribute__((aligned(16))) float patch[];       // [cb][RH * RW]
    const int tid = blockIdx.x;
    const int ty = tid / tiles_x, tx = tid - ty * tiles_x;
    const int tx0 = tx * TILE_W, ty0 = ty * TILE_H;
    const int lx = threadIdx.x & 63, ly0 = threadIdx.x >> 6;
    const int x = tx0 + lx;
    const int b = blockIdx.z;
    const int ntap = a.kh * a.kw;
    const int c_begin = SHARED ? 0 : blockIdx.y * a.cchunk, c_end = SHARED ? a.C : min(a.C, c_begin + a.cchunk);
    const int t_begin = SHARED ? blockIdx.y * tpg : 0, t_end = SHARED ? min(ntap, t_begin + tpg) : ntap;
    const size_t oplane = (size_t)a.Ho * a.Wo, iplane = (size_t)a.H * a.W;
    const int psz = RH * RW;
    const int gy0 = ty0 * a.sh - a.ph, gx0 = tx0 * a.sw - a.pw;
    bool live[4];
    size_t opix[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int y = ty0 + ly0 + 4 * e;
        live[e] = x < a.Wo && y < a.Ho;
        opix[e] = (size_t)(live[e] ? y : 0) * a.Wo + (live[e] ? x : 0);
    }
    const int step_ry = 256 / RW, step_rx = 256 - step_ry * RW;
    const int t_ry = threadIdx.x / RW, t_rx = threadIdx.x - t_ry * RW;
    const int base0 = ly0 * a.sh * RW + lx * a.sw, estep = 4 * a.sh * RW;
    const int ti0 = t_begin / a.kw, tj0 = t_begin - ti0 * a.kw;
    float acc[SHARED ? NTM : 1][4];
#pragma unroll
    for (int k = 0; k < (SHARED ? NTM : 1); ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[k][e] = 0.f;
    for (int c = c_begin; c < c_end; c += cb) {
        const int nc = min(cb, c_end - c);
        __syncthreads();
        // the batch's grad_out values are requested first: they travel while the patch is staged
        unsigned graw[CC][4];
#pragma unroll
        for (int cc = 0; cc < CC; ++cc)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                graw[cc][e] = ld1_raw_or0(gout + ((size_t)b * a.C + c + (cc < nc ? cc : 0)) * oplane, opix[e], live[e] && cc < nc);
        stage_patch_any<T>(patch, in + ((size_t)b * a.C + c) * iplane, iplane, nc, psz, RW, gy0, gx0, a.H, a.W, t_ry, t_rx,
                           step_ry, step_rx);
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) {
            if (cc >= nc) continue;
            float g[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = raw_to_float<T>(graw[cc][e]);
            const float* pc = patch + cc * psz + base0;
            int ti = ti0, tj = tj0;
            if constexpr (SHARED) {
#pragma unroll
                for (int k = 0; k < NTM; ++k) {
                    if (t_begin + k < t_end) {
                        const float* pp = pc + ti * a.dh * RW + tj * a.dw;
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[k][e] += g[e] * pp[e * estep];
                        if (++tj == a.kw) { tj = 0; ++ti; }
                    }
                }
            } else {
                T* dst = gk + ((size_t)b * a.C + c + cc) * ntap * oplane;
                for (int tap = 0; tap < ntap; ++tap) {
                    const float* pp = pc + ti * a.dh * RW + tj * a.dw;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (live[e]) st1(dst + (size_t)tap * oplane + opix[e], g[e] * pp[e * estep]);
                    if (++tj == a.kw) { tj = 0; ++ti; }
                }
            }
        }
    }
    if constexpr (SHARED) {
#pragma unroll
        for (int k = 0; k < NTM; ++k) {
            if (t_begin + k < t_end) {
                T* dst = gk + ((size_t)b * ntap + t_begin + k) * oplane;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (live[e]) st1(dst + opix[e], acc[k][e]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ dL/dkernel
// thread = (output quad, tap); blockIdx.y = tap.  grad_kernel[b,c|0,i,j,y,x] = (sum_c) g[b,c,y,x] * in0[b,c,...]
template <typename T, bool VEC, bool SHARED>
__global__ __launch_bounds__(256) void pac_conv2d_gk(const T* __restrict__ gout, const T* __restrict__ in,
                                                     T* __restrict__ gk, ConvArgs a) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= a.Ho * a.WQ) return;
    const int y = q / a.WQ, x0 = (q - y * a.WQ) * 4;
    const int b = blockIdx.z, tap = blockIdx.y;
    const int i = tap / a.kw, j = tap - i * a.kw;
    const size_t oplane = (size_t)a.Ho * a.Wo, iplane = (size_t)a.H * a.W;
    const size_t opix = (size_t)y * a.Wo + x0;
    const int ntap = a.kh * a.kw;
    const int yi = src_plain(y, i, a.sh, a.ph, a.dh, a.H);
    int xi[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int v = (x0 + e) * a.sw - a.pw + j * a.dw;
        xi[e] = (yi >= 0 && (unsigned)v < (unsigned)a.W) ? v : -1;
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < a.C; ++c) {
        float g[4];
        load_quad<T, VEC>(gout + ((size_t)b * a.C + c) * oplane + opix, x0, a.Wo, g);
        const T* row = in + ((size_t)b * a.C + c) * iplane + (size_t)(yi < 0 ? 0 : yi) * a.W;
        float p[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) p[e] = g[e] * ld1_or0(row, xi[e], xi[e] >= 0);
        if constexpr (SHARED) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += p[e];
        } else {
            store_quad<T, VEC>(gk + (((size_t)b * a.C + c) * ntap + tap) * oplane + opix, x0, a.Wo, p);
        }
    }
    if constexpr (SHARED) store_quad<T, VEC>(gk + ((size_t)b * ntap + tap) * oplane + opix, x0, a.Wo, acc);
}

// ------------------------------------------------------------------------------------------------ dL/dkernel, tiled
// Same tile and geometry restrictions as pac_conv2d_fwd_tiled.  blockIdx.y = tap row i: the workgroup stages, for CC
// channels at a time, the 16 input rows that tap row looks at, and every thread turns its grad_out quad into the K
// kernel-gradient quads of that row (accumulated over channels in registers when the kernel is shared).
template <typename T, int K, bool SHARED>
__global__ __launch_bounds__(256) void pac_conv2d_gk_tiled(const T* __restrict__ gout, const T* __restrict__ in,
                                                           T* __restrict__ gk, ConvArgs a, int tiles_x) {
    constexpr int RW = TILE_W + ((K - 1 + 3) & ~3);
    constexpr int NQUAD = (K + 3 + 3) / 4;
    __shared__ __attribute__((aligned(16))) float tile[CC][TILE_H * RW];
    const int tid = blockIdx.x;
    const int ty = tid / tiles_x, tx = tid - ty * tiles_x;
    const int tx0 = tx * TILE_W, ty0 = ty * TILE_H;
    const int qx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int x0 = tx0 + 4 * qx, y = ty0 + ly;
    const bool live = y < a.Ho && x0 < a.Wo;
    const int b = blockIdx.z, i = blockIdx.y;
    const size_t oplane = (size_t)a.Ho * a.Wo, iplane = (size_t)a.H * a.W;
    const size_t opix = (size_t)y * a.Wo + x0;
    float acc[K][4];
#pragma unroll
    for (int j = 0; j < K; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][e] = 0.f;
    for (int c = 0; c < a.C; c += CC) {
        const int nc = min(CC, a.C - c);
        float g[CC][4];
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) {
            if (live && cc < nc) {
                const T* gp = gout + ((size_t)b * a.C + c + cc) * oplane + opix;
                if (a.vec) load_quad<T, true>(gp, x0, a.Wo, g[cc]);
                else load_quad<T, false>(gp, x0, a.Wo, g[cc]);
            }
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < nc * TILE_H * RW; idx += 256) {
            const int cc = idx / (TILE_H * RW), r = idx - cc * (TILE_H * RW);
            const int ry = r / RW, rx = r - ry * RW;
            const int yi = ty0 - a.ph + i + ry, xi = tx0 - a.pw + rx;
            tile[cc][r] = ld1_or0(in + ((size_t)b * a.C + c + cc) * iplane, (size_t)yi * a.W + xi,
                                  (unsigned)yi < (unsigned)a.H && (unsigned)xi < (unsigned)a.W);
        }
        __syncthreads();
        if (live) {
#pragma unroll
            for (int cc = 0; cc < CC; ++cc) {
                if (cc < nc) {
                    float win[4 * NQUAD];
#pragma unroll
                    for (int n = 0; n < NQUAD; ++n) {
                        const v4f v = *(lds_cv4f_ptr)(&tile[cc][ly * RW + 4 * qx + 4 * n]);
                        win[4 * n] = v.x; win[4 * n + 1] = v.y; win[4 * n + 2] = v.z; win[4 * n + 3] = v.w;
                    }
                    if constexpr (SHARED) {
#pragma unroll
                        for (int j = 0; j < K; ++j)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[j][e] += g[cc][e] * win[j + e];   // mul then add, as :117-119
                    } else {
#pragma unroll
                        for (int j = 0; j < K; ++j) {
                            float pr[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) pr[e] = g[cc][e] * win[j + e];
                            T* dst = gk + (((size_t)b * a.C + c + cc) * (K * K) + i * K + j) * oplane + opix;
                            if (a.vec) store_quad<T, true>(dst, x0, a.Wo, pr);
                            else store_quad<T, false>(dst, x0, a.Wo, pr);
                        }
                    }
                }
            }
        }
    }
    if (SHARED && live) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            T* dst = gk + ((size_t)b * (K * K) + i * K + j) * oplane + opix;
            if (a.vec) store_quad<T, true>(dst, x0, a.Wo, acc[j]);
            else store_quad<T, false>(dst, x0, a.Wo, acc[j]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ dL/dkernel, fp16 octs
// dL/dkernel without a sum over channels (one kernel per channel, or a single channel: what a CSPN_ours step has) is an
// outer product  grad_kernel[c,i,j][q] = grad_out[c][q] * in[c][q + (i,j)]  — 2 + 2 bytes read and 2 K*K written per pixel,
// a pure store stream.  The fp16 quad kernel writes it as 8-byte stores (half the bytes per instruction); here a thread
// owns eight pixels: one 16-byte load of grad_out, window rows from the LDS patch, K*K 16-byte stores.
template <int K>
__global__ __launch_bounds__(256) void pac_conv2d_gk_h8(const __half* __restrict__ gout, const __half* __restrict__ in,
                                                        __half* __restrict__ gk, TiledArgs a) {
    constexpr int RW = TILE_W8 + ((K - 1 + 3) & ~3);
    constexpr int RH = TILE_H + K - 1;
    constexpr int PATCH = RH * RW;
    constexpr int NLD = (PATCH + 255) / 256;
    constexpr int NQUAD = (K + 7 + 3) / 4;
    __shared__ __attribute__((aligned(16))) float tile[2][PATCH];
    const int tid = blockIdx.x;
    const int ty = tid / a.tiles_x, tx = tid - ty * a.tiles_x;
    const int tx0 = tx * TILE_W8, ty0 = ty * TILE_H;
    const int ox = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int x0 = tx0 + 8 * ox, y = ty0 + ly;
    const bool live = y < a.dst_h && x0 < a.dst_w;      // dst_w % 8 == 0 (launcher)
    const int b = blockIdx.z;
    const int c_begin = blockIdx.y * a.cchunk, c_end = min(a.C, c_begin + a.cchunk);
    const size_t splane = (size_t)a.src_h * a.src_w, dplane = (size_t)a.dst_h * a.dst_w;
    const size_t dpix = (size_t)y * a.dst_w + x0;

    int goff[NLD];
#pragma unroll
    for (int n = 0; n < NLD; ++n) {
        const int idx = threadIdx.x + 256 * n;
        const int ry = idx / RW, rx = idx - ry * RW;
        const int yi = ty0 + a.org_y + ry, xi = tx0 + a.org_x + rx;
        goff[n] = (idx < PATCH && (unsigned)yi < (unsigned)a.src_h && (unsigned)xi < (unsigned)a.src_w) ? yi * a.src_w + xi : -1;
    }
    unsigned pre[NLD];
    uint4 gpre = make_uint4(0u, 0u, 0u, 0u), graw = gpre;
    auto fetch = [&](int c) {
        const __half* sp = in + ((size_t)b * a.C + min(c, a.C - 1)) * splane;
#pragma unroll
        for (int n = 0; n < NLD; ++n) pre[n] = ld1_raw_or0(sp, goff[n], c < c_end && goff[n] >= 0);
        if (live && c < c_end) gpre = *reinterpret_cast<const uint4*>(gout + ((size_t)b * a.C + c) * dplane + dpix);
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int n = 0; n < NLD; ++n)
            if (256 * (n + 1) <= PATCH || threadIdx.x + 256 * n < PATCH) tile[buf][threadIdx.x + 256 * n] = raw_to_float<__half>(pre[n]);
        graw = gpre;
    };
    fetch(c_begin);
    commit(0);
    __syncthreads();
    int buf = 0;
    for (int c = c_begin; c < c_end; ++c) {
        const bool more = c + 1 < c_end;
        const uint4 gc = graw;                              // this channel's grad_out oct (commit below overwrites graw)
        if (more) fetch(c + 1);
        if (live) {
            float g[8];
            {
                const float2 a0 = __half22float2(*reinterpret_cast<const __half2*>(&gc.x));
                const float2 a1 = __half22float2(*reinterpret_cast<const __half2*>(&gc.y));
                const float2 a2 = __half22float2(*reinterpret_cast<const __half2*>(&gc.z));
                const float2 a3 = __half22float2(*reinterpret_cast<const __half2*>(&gc.w));
                g[0] = a0.x; g[1] = a0.y; g[2] = a1.x; g[3] = a1.y; g[4] = a2.x; g[5] = a2.y; g[6] = a3.x; g[7] = a3.y;
            }
            __half* dp = gk + ((size_t)b * a.C + c) * (K * K) * dplane + dpix;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                float win[4 * NQUAD];
#pragma unroll
                for (int n = 0; n < NQUAD; ++n) {
                    const v4f v = *(lds_cv4f_ptr)(&tile[buf][(ly + i) * RW + 8 * ox + 4 * n]);
                    win[4 * n] = v.x; win[4 * n + 1] = v.y; win[4 * n + 2] = v.z; win[4 * n + 3] = v.w;
                }
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    uint4 o;
                    *reinterpret_cast<__half2*>(&o.x) = __floats2half2_rn(g[0] * win[j], g[1] * win[j + 1]);
                    *reinterpret_cast<__half2*>(&o.y) = __floats2half2_rn(g[2] * win[j + 2], g[3] * win[j + 3]);
                    *reinterpret_cast<__half2*>(&o.z) = __floats2half2_rn(g[4] * win[j + 4], g[5] * win[j + 5]);
                    *reinterpret_cast<__half2*>(&o.w) = __floats2half2_rn(g[6] * win[j + 6], g[7] * win[j + 7]);
                    *reinterpret_cast<uint4*>(dp + (size_t)(i * K + j) * dplane) = o;
                }
            }
        }
        if (more) commit(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
}

// ------------------------------------------------------------------------------------------------ dL/dkernel, whole window
// K <= 5: one workgroup produces all K*K kernel-gradient planes of its tile in one pass — the input patches and the
// grad_out quads of the next channel batch are prefetched exactly as in pac_conv2d_tiled; with a shared kernel the
// K*K quads accumulate over every channel in registers and are written once.
template <typename T, int K, bool SHARED, int CB>
__global__ __launch_bounds__(256) void pac_conv2d_gk_window(const T* __restrict__ gout, const T* __restrict__ in,
                                                            T* __restrict__ gk, TiledArgs a) {
    constexpr int RW = TILE_W + ((K - 1 + 3) & ~3);
    constexpr int RH = TILE_H + K - 1;
    constexpr int PATCH = RH * RW;
    constexpr int NLD = (PATCH + 255) / 256;
    constexpr int NQUAD = (K + 3 + 3) / 4;
    __shared__ __attribute__((aligned(16))) float tile[2][CB][PATCH];
    const int tid = blockIdx.x;
    const int ty = tid / a.tiles_x, tx = tid - ty * a.tiles_x;
    const int tx0 = tx * TILE_W, ty0 = ty * TILE_H;
    const int qx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int x0 = tx0 + 4 * qx, y = ty0 + ly;
    const bool live = y < a.dst_h && x0 < a.dst_w;
    const int b = blockIdx.z;
    const int c_begin = blockIdx.y * a.cchunk, c_end = min(a.C, c_begin + a.cchunk);
    const size_t splane = (size_t)a.src_h * a.src_w, dplane = (size_t)a.dst_h * a.dst_w;
    const size_t dpix = (size_t)y * a.dst_w + x0;

    int goff[NLD];
#pragma unroll
    for (int n = 0; n < NLD; ++n) {
        const int idx = threadIdx.x + 256 * n;
        const int ry = idx / RW, rx = idx - ry * RW;
        const int yi = ty0 + a.org_y + ry, xi = tx0 + a.org_x + rx;
        goff[n] = (idx < PATCH && (unsigned)yi < (unsigned)a.src_h && (unsigned)xi < (unsigned)a.src_w) ? yi * a.src_w + xi : -1;
    }
    unsigned pre[CB][NLD];                              // raw bits: converted at commit time (see ld1_raw_or0)
    float gpre[CB][4], g[CB][4];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int cc = 0; cc < CB; ++cc) {
            const int c = min(c0 + cc, a.C - 1);
            const T* sp = in + ((size_t)b * a.C + c) * splane;
#pragma unroll
            for (int n = 0; n < NLD; ++n) pre[cc][n] = ld1_raw_or0(sp, goff[n], c0 + cc < c_end && goff[n] >= 0);
            if (live && c0 + cc < c_end) {
                const T* gp = gout + ((size_t)b * a.C + c) * dplane + dpix;
                if (a.dst_vec) load_quad<T, true>(gp, x0, a.dst_w, gpre[cc]);
                else load_quad<T, false>(gp, x0, a.dst_w, gpre[cc]);
            }
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int cc = 0; cc < CB; ++cc) {
#pragma unroll
            for (int n = 0; n < NLD; ++n)
                if (256 * (n + 1) <= PATCH || threadIdx.x + 256 * n < PATCH) tile[buf][cc][threadIdx.x + 256 * n] = raw_to_float<T>(pre[cc][n]);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[cc][e] = gpre[cc][e];
        }
    };
    float acc[K * K][4];
#pragma unroll
    for (int t = 0; t < K * K; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = 0.f;
    fetch(c_begin);
    commit(0);
    __syncthreads();
    int buf = 0;
    for (int c = c_begin; c < c_end; c += CB) {
        const bool more = c + CB < c_end;
        float gc[CB][4];                                  // this batch's grad_out quads (g is overwritten by commit below)
#pragma unroll
        for (int cc = 0; cc < CB; ++cc)
#pragma unroll
            for (int e = 0; e < 4; ++e) gc[cc][e] = g[cc][e];
        if (more) fetch(c + CB);
        if (live) {
#pragma unroll
            for (int cc = 0; cc < CB; ++cc) {
                if (c + cc < c_end) {
#pragma unroll
                    for (int i = 0; i < K; ++i) {
                        float win[4 * NQUAD];
#pragma unroll
                        for (int n = 0; n < NQUAD; ++n) {
                            const v4f v = *(lds_cv4f_ptr)(&tile[buf][cc][(ly + i) * RW + 4 * qx + 4 * n]);
                            win[4 * n] = v.x; win[4 * n + 1] = v.y; win[4 * n + 2] = v.z; win[4 * n + 3] = v.w;
                        }
#pragma unroll
                        for (int j = 0; j < K; ++j) {
                            if constexpr (SHARED) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[i * K + j][e] = fmaf(gc[cc][e], win[j + e], acc[i * K + j][e]);
                            } else {
                                float pr[4];
#pragma unroll
                                for (int e = 0; e < 4; ++e) pr[e] = gc[cc][e] * win[j + e];
                                T* dp = gk + (((size_t)b * a.C + c + cc) * (K * K) + i * K + j) * dplane + dpix;
                                if (a.dst_vec) store_quad<T, true>(dp, x0, a.dst_w, pr);
                                else store_quad<T, false>(dp, x0, a.dst_w, pr);
                            }
                        }
                    }
                }
            }
        }
        if (more) commit(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    if (SHARED && live) {
#pragma unroll
        for (int t = 0; t < K * K; ++t) {
            T* dp = gk + ((size_t)b * (K * K) + t) * dplane + dpix;
            if (a.dst_vec) store_quad<T, true>(dp, x0, a.dst_w, acc[t]);
            else store_quad<T, false>(dp, x0, a.dst_w, acc[t]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ dL/dinput
// Gather form of the fold (pac.py:104-113): thread = quad of 4 consecutive INPUT pixels; for every tap the output
// pixel whose window puts that tap on this input pixel contributes g * kernel.  WQ here counts input quads.
template <typename T, bool SHARED, bool UNIT_STRIDE>
__global__ __launch_bounds__(256) void pac_conv2d_gi(const T* __restrict__ gout, const T* __restrict__ kern,
                                                     T* __restrict__ gin, ConvArgs a, int in_wq, int in_vec) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= a.H * in_wq) return;
    const int yy = q / in_wq, x0 = (q - yy * in_wq) * 4;
    const int b = blockIdx.z;
    const int c_begin = blockIdx.y * a.cchunk, c_end = min(a.C, c_begin + a.cchunk);
    const size_t oplane = (size_t)a.Ho * a.Wo, iplane = (size_t)a.H * a.W;
    const int ntap = a.kh * a.kw;
    for (int c = c_begin; c < c_end; c += CC) {
        const int nc = min(CC, c_end - c);
        float acc[CC][4];
#pragma unroll
        for (int cc = 0; cc < CC; ++cc)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[cc][e] = 0.f;
        for (int i = 0; i < a.kh; ++i) {
            int ty = yy + a.ph - i * a.dh;
            if (ty < 0) continue;
            if constexpr (!UNIT_STRIDE) {
                if (ty % a.sh) continue;
                ty /= a.sh;
            }
            if (ty >= a.Ho) continue;
            for (int j = 0; j < a.kw; ++j) {
                const int tap = i * a.kw + j;
                int xo[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int tx = x0 + e + a.pw - j * a.dw;
                    bool ok = tx >= 0 && x0 + e < a.W;
                    if constexpr (!UNIT_STRIDE) {
                        ok = ok && (tx % a.sw) == 0;
                        tx /= a.sw;
                    }
                    xo[e] = (ok && tx < a.Wo) ? tx : -1;
                }
                const size_t orow = (size_t)ty * a.Wo;
                float kv[4];
                if constexpr (SHARED) {
                    const T* kp = kern + ((size_t)b * ntap + tap) * oplane + orow;
#pragma unroll
                    for (int e = 0; e < 4; ++e) kv[e] = ld1_or0(kp, xo[e], xo[e] >= 0);
                }
#pragma unroll
                for (int cc = 0; cc < CC; ++cc) {
                    if (cc < nc) {
                        if constexpr (!SHARED) {
                            const T* kp = kern + (((size_t)b * a.C + c + cc) * ntap + tap) * oplane + orow;
#pragma unroll
                            for (int e = 0; e < 4; ++e) kv[e] = ld1_or0(kp, xo[e], xo[e] >= 0);
                        }
                        const T* gp = gout + ((size_t)b * a.C + c + cc) * oplane + orow;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (xo[e] >= 0) acc[cc][e] = fmaf(kv[e], ld1(gp + xo[e]), acc[cc][e]);
                    }
                }
            }
        }
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) {
            if (cc < nc) {
                T* dst = gin + ((size_t)b * a.C + c + cc) * iplane + (size_t)yy * a.W + x0;
                if (in_vec) store_quad<T, true>(dst, x0, a.W, acc[cc]);
                else store_quad<T, false>(dst, x0, a.W, acc[cc]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ nd2col
// thread = output quad of one (b*C + c, tap) plane.  blockIdx.y = tap, blockIdx.z = b*C + c.
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void pac_nd2col_kernel(const T* __restrict__ in, T* __restrict__ cols, ConvArgs a) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= a.Ho * a.WQ) return;
    const int y = q / a.WQ, x0 = (q - y * a.WQ) * 4;
    const int tap = blockIdx.y, bc = blockIdx.z;
    const int i = tap / a.kw, j = tap - i * a.kw;
    const size_t oplane = (size_t)a.Ho * a.Wo, iplane = (size_t)a.H * a.W;
    const int yi = a.transposed ? src_transposed(y, i, a.sh, a.lead_h, a.dh, a.H) : src_plain(y, i, a.sh, a.ph, a.dh, a.H);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int xi = a.transposed ? src_transposed(x0 + e, j, a.sw, a.lead_w, a.dw, a.W)
                                    : src_plain(x0 + e, j, a.sw, a.pw, a.dw, a.W);
        v[e] = ld1_or0(in + (size_t)bc * iplane, (size_t)yi * a.W + xi, yi >= 0 && xi >= 0 && x0 + e < a.Wo);
    }
    store_quad<T, VEC>(cols + ((size_t)bc * (a.kh * a.kw) + tap) * oplane + (size_t)y * a.Wo + x0, x0, a.Wo, v);
}

// ------------------------------------------------------------------------------------------------ host
bool out_size(int H, int W, const cspn_conv_geometry& g, int* Ho, int* Wo) {
    long ho, wo;
    if (g.transposed) {
        ho = (long)(H - 1) * g.sh - 2L * g.ph + (long)g.dh * (g.kh - 1) + 1 + g.oph;
        wo = (long)(W - 1) * g.sw - 2L * g.pw + (long)g.dw * (g.kw - 1) + 1 + g.opw;
    } else {
        const long nh = (long)H + 2L * g.ph - (long)g.dh * (g.kh - 1) - 1;
        const long nw = (long)W + 2L * g.pw - (long)g.dw * (g.kw - 1) - 1;
        if (nh < 0 || nw < 0) return false;
        ho = nh / g.sh + 1;
        wo = nw / g.sw + 1;
    }
    if (ho < 1 || wo < 1 || ho > (1L << 30) || wo > (1L << 30)) return false;
    *Ho = (int)ho;
    *Wo = (int)wo;
    return true;
}

int make_args(const char* who, int dtype, int B, int C, int CK, int H, int W, const cspn_conv_geometry* g,
              bool allow_transposed, ConvArgs* a) {
    if (!g) return fail("%s: geom is null", who);
    if (dtype != CSPN_F32 && dtype != CSPN_F16) return fail("%s: dtype must be CSPN_F32 or CSPN_F16", who);
    if (B < 1 || C < 1 || H < 1 || W < 1) return fail("%s: empty tensor (B=%d C=%d H=%d W=%d)", who, B, C, H, W);
    if (g->kh < 1 || g->kw < 1 || g->sh < 1 || g->sw < 1 || g->dh < 1 || g->dw < 1 || g->ph < 0 || g->pw < 0 ||
        g->oph < 0 || g->opw < 0)
        return fail("%s: bad geometry (k %dx%d, stride %d,%d, pad %d,%d, dilation %d,%d)", who, g->kh, g->kw, g->sh,
                    g->sw, g->ph, g->pw, g->dh, g->dw);
    if (g->transposed && !allow_transposed) return fail("%s: transposed geometry is for nd2col only", who);
    if (!g->transposed && (g->oph || g->opw)) return fail("%s: output_padding needs transposed", who);
    if (CK != 1 && CK != C) return fail("%s: Incompatible input and kernel sizes (kernel_ch=%d, C=%d)", who, CK, C);
    if (B > 65535) return fail("%s: B=%d exceeds the grid limit 65535", who, B);
    if ((long)g->kh * g->kw > 65535) return fail("%s: window %dx%d too large", who, g->kh, g->kw);
    ConvArgs r{};
    r.B = B; r.C = C; r.CK = CK; r.H = H; r.W = W;
    r.kh = g->kh; r.kw = g->kw; r.sh = g->sh; r.sw = g->sw; r.ph = g->ph; r.pw = g->pw; r.dh = g->dh; r.dw = g->dw;
    r.transposed = g->transposed ? 1 : 0;
    r.lead_h = (g->kh - 1) * g->dh - g->ph;
    r.lead_w = (g->kw - 1) * g->dw - g->pw;
    if (r.transposed && (r.lead_h < 0 || r.lead_w < 0))
        return fail("%s: transposed geometry with padding > (k-1)*dilation is not defined (negative pad)", who);
    if (!out_size(H, W, *g, &r.Ho, &r.Wo)) return fail("%s: geometry gives an empty output for input %dx%d", who, H, W);
    r.WQ = ceil_div(r.Wo, 4);
    r.force_scalar = force_generic_flag().load(std::memory_order_relaxed);
    if ((size_t)r.Ho * r.WQ > (size_t)1 << 30) return fail("%s: plane too large", who);
    *a = r;
    return 1;
}

// spread channels over blockIdx.y until the launch has enough workgroups to fill 256 CUs a few times over
int channel_chunk(int C, size_t spatial_blocks) {
    const size_t want = 2048;
    if (spatial_blocks >= want || C <= CC) return C;
    size_t nchunk = (want + spatial_blocks - 1) / spatial_blocks;
    const size_t maxchunk = (size_t)ceil_div(C, CC);
    if (nchunk > maxchunk) nchunk = maxchunk;
    int per = ceil_div(C, (int)nchunk);
    per = ceil_div(per, CC) * CC;
    return per;
}

bool aligned_for(const void* p, int dtype) {
    return (reinterpret_cast<uintptr_t>(p) & (dtype == CSPN_F16 ? 7 : 15)) == 0;
}

// stride 2 x 2, dilation 1, K in {3, 5}, padding K / 2, whole octets, 16-byte aligned bases: the register / DPP kernels of
// pac_conv2d_s2.hip (no LDS staging, no tile quantisation)
bool s2_fast(const ConvArgs& a, const void* p0, const void* p1, const void* p2) {
    if (a.force_scalar || a.transposed) return false;
    if (!cspn_detail::pac_s2_geometry(a.kh, a.kw, a.sh, a.sw, a.ph, a.pw, a.dh, a.dw, a.W)) return false;
    if ((size_t)a.H * a.W >= ((size_t)1 << 31)) return false;
    return ((reinterpret_cast<uintptr_t>(p0) | reinterpret_cast<uintptr_t>(p1) | reinterpret_cast<uintptr_t>(p2)) & 15) == 0;
}
cspn_detail::PacS2Args s2_args(const ConvArgs& a) {
    cspn_detail::PacS2Args r{};
    r.B = a.B; r.C = a.C; r.CK = a.CK; r.H = a.H; r.W = a.W; r.Ho = a.Ho; r.Wo = a.Wo; r.WQ = a.Wo / 4;
    r.cchunk = a.C;
    return r;
}

bool tiled_geometry(const ConvArgs& a) {
    if ((size_t)a.H * a.W >= ((size_t)1 << 31)) return false;      // the tiled kernels keep plane offsets in int
    return a.kh == a.kw && (a.kh == 3 || a.kh == 5 || a.kh == 7) && a.sh == 1 && a.sw == 1 && a.dh == 1 && a.dw == 1;
}

// Launch of pac_conv2d_tiled for the forward (transposed = false) or the input gradient (true).
template <typename T, int K, bool TRANSPOSED>
int launch_tiled(const T* src, const T* kern, T* dst, const ConvArgs& a, int dst_vec, hipStream_t st) {
    TiledArgs t{};
    t.B = a.B; t.C = a.C; t.CK = a.CK;
    t.k_h = a.Ho; t.k_w = a.Wo;
    if (TRANSPOSED) {
        t.src_h = a.Ho; t.src_w = a.Wo; t.dst_h = a.H; t.dst_w = a.W;
        t.org_y = a.ph - (K - 1); t.org_x = a.pw - (K - 1);
    } else {
        t.src_h = a.H; t.src_w = a.W; t.dst_h = a.Ho; t.dst_w = a.Wo;
        t.org_y = -a.ph; t.org_x = -a.pw;
    }
    t.k_vec = a.vec; t.dst_vec = dst_vec;
    if constexpr (std::is_same<T, __half>::value && !TRANSPOSED && K <= 5) {
        // fp16 forward with whole 16-byte octs everywhere: the eight-pixel kernel (2x the bytes per load instruction)
        if (a.vec && dst_vec && t.dst_w % 8 == 0 && ((reinterpret_cast<uintptr_t>(kern) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
            t.tiles_x = ceil_div(t.dst_w, TILE_W8);
            const int tiles8 = t.tiles_x * ceil_div(t.dst_h, TILE_H);
            const size_t want8 = 1024, have8 = (size_t)tiles8 * a.B;
            int nchunk8 = (int)std::min<size_t>((want8 + have8 - 1) / have8, (size_t)a.C);
            if (a.CK != 1) nchunk8 = (int)std::min<size_t>((4 * want8 + have8 - 1) / have8, (size_t)a.C);
            t.cchunk = ceil_div(a.C, std::max(nchunk8, 1));
            t.lin_tiles = tiles8; t.lin_chunks = ceil_div(a.C, t.cchunk);
            const dim3 grid8 = lin_grid(tiles8, t.lin_chunks, a.B), block8(256);
            if (a.CK == 1 && t.cchunk > 1) CSPN_PRE(st), pac_conv2d_tiled_h8<K, true><<<grid8, block8, 0, st>>>(src, kern, dst, t);
            else CSPN_PRE(st), pac_conv2d_tiled_h8<K, false><<<grid8, block8, 0, st>>>(src, kern, dst, t);
            HIP_OK(hipGetLastError());
            return 1;
        }
    }
    t.tiles_x = ceil_div(t.dst_w, TILE_W);
    const int tiles = t.tiles_x * ceil_div(t.dst_h, TILE_H);
    // every channel chunk re-reads the kernel planes, so only split as far as filling the chip needs (~4 x 256 groups)
    const size_t want = ANY_WANT_WGS, have = (size_t)tiles * a.B;
    int nchunk = (int)std::min<size_t>((want + have - 1) / have, (size_t)a.C);
    if (a.CK != 1) nchunk = (int)std::min<size_t>((4 * want + have - 1) / have, (size_t)a.C);   // nothing is re-read
    t.cchunk = ceil_div(a.C, std::max(nchunk, 1));
    t.lin_tiles = tiles; t.lin_chunks = ceil_div(a.C, t.cchunk);
    const dim3 grid = lin_grid(tiles, t.lin_chunks, a.B), block(256);
    constexpr bool CAN_HOIST = K <= 5;                   // the whole window in registers
    const bool hoist = CAN_HOIST && a.CK == 1;
    if constexpr (K > 5) {
        // a shared 7 x 7 kernel streams its 49 taps once per channel batch: batches of eight channels (101 KB of LDS, one
        // workgroup per CU) halve that stream
        if (a.CK == 1 && a.C >= 8) {
            int nchunk8 = (int)std::min<size_t>((want + have - 1) / have, (size_t)ceil_div(a.C, 8));
            t.cchunk = ceil_div(ceil_div(a.C, std::max(nchunk8, 1)), 8) * 8;
            t.lin_chunks = ceil_div(a.C, t.cchunk);
            const dim3 grid8 = lin_grid(tiles, t.lin_chunks, a.B);
            if (t.cchunk <= 8) CSPN_PRE(st), pac_conv2d_tiled<T, K, false, 8, TRANSPOSED, 1><<<grid8, block, 0, st>>>(src, kern, dst, t);
            else CSPN_PRE(st), pac_conv2d_tiled<T, K, false, 8, TRANSPOSED><<<grid8, block, 0, st>>>(src, kern, dst, t);
            HIP_OK(hipGetLastError());
            return 1;
        }
    }
    if (t.cchunk == 1) {
        if (hoist) CSPN_PRE(st), pac_conv2d_tiled<T, K, CAN_HOIST, 1, TRANSPOSED><<<grid, block, 0, st>>>(src, kern, dst, t);
        else CSPN_PRE(st), pac_conv2d_tiled<T, K, false, 1, TRANSPOSED><<<grid, block, 0, st>>>(src, kern, dst, t);
    } else {
        if (hoist) CSPN_PRE(st), pac_conv2d_tiled<T, K, CAN_HOIST, CC, TRANSPOSED><<<grid, block, 0, st>>>(src, kern, dst, t);
        else CSPN_PRE(st), pac_conv2d_tiled<T, K, false, CC, TRANSPOSED><<<grid, block, 0, st>>>(src, kern, dst, t);
    }
    HIP_OK(hipGetLastError());
    return 1;
}

template <typename T, bool TRANSPOSED>
int launch_tiled_k(const T* src, const T* kern, T* dst, const ConvArgs& a, int dst_vec, hipStream_t st) {
    return a.kh == 3 ? launch_tiled<T, 3, TRANSPOSED>(src, kern, dst, a, dst_vec, st)
         : a.kh == 5 ? launch_tiled<T, 5, TRANSPOSED>(src, kern, dst, a, dst_vec, st)
                     : launch_tiled<T, 7, TRANSPOSED>(src, kern, dst, a, dst_vec, st);
}

template <typename T>
int conv_forward_typed(const void* in, const void* kern, void* out, ConvArgs a, hipStream_t st) {
    if (s2_fast(a, in, kern, out))
        return cspn_detail::pac_s2_forward(in, kern, out, std::is_same<T, __half>::value ? CSPN_F16 : CSPN_F32, a.kh, s2_args(a), st);
    if (tiled_geometry(a) && !a.force_scalar) {
        const T* i = static_cast<const T*>(in);
        const T* k = static_cast<const T*>(kern);
        T* o = static_cast<T*>(out);
        return launch_tiled_k<T, false>(i, k, o, a, a.vec, st);
    }
    const T* i = static_cast<const T*>(in);
    const T* k = static_cast<const T*>(kern);
    T* o = static_cast<T*>(out);
    const bool shared = a.CK == 1;
    if (!a.force_scalar && (size_t)a.H * a.W < ((size_t)1 << 31)) {
        // every other geometry (stride, dilation, non-square or even windows): LDS-tiled when the input patch of a 64 x 16
        // output tile fits 64 KiB for at least one channel.  (Strided windows used to stay on the scalar kernel — their
        // patches are stride^2 larger per output and the tiled form measured 87 vs 77 us at stride 2; that was the old
        // staging loop, four loads in flight per wavefront: with stage_patch_any it is 42 vs 76 us.)
        const long RWl = ((long)(TILE_W - 1) * a.sw + (long)(a.kw - 1) * a.dw + 1 + 3) & ~3L;
        const long RHl = (long)(TILE_H - 1) * a.sh + (long)(a.kh - 1) * a.dh + 1;
        const long psz = RWl * RHl;
        int cb = (int)std::min<long>(CC, (64 * 1024 / 4) / std::max(psz, 1L));
        cb = std::min(cb, a.C);
        if (cb >= 1) {
            const int tiles_x = ceil_div(a.Wo, TILE_W), tiles = tiles_x * ceil_div(a.Ho, TILE_H);
            const size_t want = ANY_WANT_WGS, have = (size_t)tiles * a.B;
            int nchunk = (int)std::min<size_t>((want + have - 1) / have, (size_t)ceil_div(a.C, cb));
            a.cchunk = ceil_div(ceil_div(a.C, std::max(nchunk, 1)), cb) * cb;
            a.lin_tiles = tiles; a.lin_chunks = ceil_div(a.C, a.cchunk);
            const dim3 grid = lin_grid(tiles, a.lin_chunks, a.B), block(256);
            const size_t lds = (size_t)cb * psz * sizeof(float);
            if (shared && a.kh * a.kw <= 9) CSPN_PRE(st), pac_conv2d_fwd_tiled_any<T, true, false, true><<<grid, block, lds, st>>>(i, k, o, a, tiles_x, (int)RWl, (int)RHl, cb);
            else if (shared) CSPN_PRE(st), pac_conv2d_fwd_tiled_any<T, true, false, false><<<grid, block, lds, st>>>(i, k, o, a, tiles_x, (int)RWl, (int)RHl, cb);
            else CSPN_PRE(st), pac_conv2d_fwd_tiled_any<T, false, false, false><<<grid, block, lds, st>>>(i, k, o, a, tiles_x, (int)RWl, (int)RHl, cb);
            HIP_OK(hipGetLastError());
            return 1;
        }
    }
    const int gx = ceil_div(a.Ho * a.WQ, 256);
    a.cchunk = channel_chunk(a.C, (size_t)gx * a.B);
    const dim3 grid(gx, ceil_div(a.C, a.cchunk), a.B), block(256);
    if (a.vec && shared) CSPN_PRE(st), pac_conv2d_fwd<T, true, true><<<grid, block, 0, st>>>(i, k, o, a);
    else if (a.vec) CSPN_PRE(st), pac_conv2d_fwd<T, true, false><<<grid, block, 0, st>>>(i, k, o, a);
    else if (shared) CSPN_PRE(st), pac_conv2d_fwd<T, false, true><<<grid, block, 0, st>>>(i, k, o, a);
    else CSPN_PRE(st), pac_conv2d_fwd<T, false, false><<<grid, block, 0, st>>>(i, k, o, a);
    HIP_OK(hipGetLastError());
    return 1;
}

template <typename T, int K>
int conv_gk_tiled(const T* g, const T* in, T* gk, ConvArgs a, hipStream_t st) {
    const int tiles_x = ceil_div(a.Wo, TILE_W), tiles = tiles_x * ceil_div(a.Ho, TILE_H);
    if constexpr (std::is_same<T, __half>::value && K <= 5) {
        // fp16, no sum over channels (CK == C), whole 16-byte octs: the eight-pixel outer-product kernel
        if (a.CK == a.C && a.vec && a.Wo % 8 == 0 &&
            ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(gk)) & 15) == 0) {
            TiledArgs t{};
            t.B = a.B; t.C = a.C; t.CK = a.CK;
            t.src_h = a.H; t.src_w = a.W; t.dst_h = a.Ho; t.dst_w = a.Wo; t.k_h = a.Ho; t.k_w = a.Wo;
            t.org_y = -a.ph; t.org_x = -a.pw;
            t.k_vec = t.dst_vec = a.vec;
            t.tiles_x = ceil_div(a.Wo, TILE_W8);
            const int tiles8 = t.tiles_x * ceil_div(a.Ho, TILE_H);
            const int nchunk = (int)std::min<size_t>((4096 + (size_t)tiles8 * a.B - 1) / ((size_t)tiles8 * a.B), (size_t)a.C);
            t.cchunk = ceil_div(a.C, std::max(nchunk, 1));
            const dim3 grid8(tiles8, ceil_div(a.C, t.cchunk), a.B), block8(256);
            CSPN_PRE(st), pac_conv2d_gk_h8<K><<<grid8, block8, 0, st>>>(g, in, gk, t);
            HIP_OK(hipGetLastError());
            return 1;
        }
    }
    if constexpr (K <= 5) {
        // whole-window kernel; a shared kernel pins all channels to one workgroup, so it needs enough tiles to fill the
        // chip — small launches keep the tap-row split below (K x the workgroups)
        if (a.CK != 1 || (size_t)tiles * a.B >= 256) {
            TiledArgs t{};
            t.B = a.B; t.C = a.C; t.CK = a.CK;
            t.src_h = a.H; t.src_w = a.W; t.dst_h = a.Ho; t.dst_w = a.Wo; t.k_h = a.Ho; t.k_w = a.Wo;
            t.org_y = -a.ph; t.org_x = -a.pw;
            t.k_vec = t.dst_vec = a.vec;
            t.tiles_x = tiles_x;
            int nchunk = 1;
            if (a.CK != 1) nchunk = (int)std::min<size_t>((4096 + (size_t)tiles * a.B - 1) / ((size_t)tiles * a.B), (size_t)a.C);
            t.cchunk = ceil_div(a.C, nchunk);
            const dim3 grid(tiles, ceil_div(a.C, t.cchunk), a.B), block(256);
            if (a.CK == 1) {
                if (a.C == 1) CSPN_PRE(st), pac_conv2d_gk_window<T, K, true, 1><<<grid, block, 0, st>>>(g, in, gk, t);
                else CSPN_PRE(st), pac_conv2d_gk_window<T, K, true, CC><<<grid, block, 0, st>>>(g, in, gk, t);
            } else {
                if (t.cchunk == 1) CSPN_PRE(st), pac_conv2d_gk_window<T, K, false, 1><<<grid, block, 0, st>>>(g, in, gk, t);
                else CSPN_PRE(st), pac_conv2d_gk_window<T, K, false, CC><<<grid, block, 0, st>>>(g, in, gk, t);
            }
            HIP_OK(hipGetLastError());
            return 1;
        }
    }
    if constexpr (K == 7) {
        // shared 7 x 7: the tap-row split below reads grad_out and the input seven times; with enough tiles the whole window
        // (49 accumulator quads) stays in the registers of one workgroup instead
        if (a.CK == 1 && (size_t)tiles * a.B >= 256) {
            TiledArgs t{};
            t.B = a.B; t.C = a.C; t.CK = a.CK;
            t.src_h = a.H; t.src_w = a.W; t.dst_h = a.Ho; t.dst_w = a.Wo; t.k_h = a.Ho; t.k_w = a.Wo;
            t.org_y = -a.ph; t.org_x = -a.pw;
            t.k_vec = t.dst_vec = a.vec;
            t.tiles_x = tiles_x;
            t.cchunk = a.C;
            const dim3 gridw(tiles, 1, a.B), blockw(256);
            // one channel per batch: 247 VGPRs, two wavefronts per SIMD (two channels: 256 + spills to AGPRs, one wavefront —
            // 67.8 vs 52.3 us; the tap-row split: 125.9 us)
            CSPN_PRE(st), pac_conv2d_gk_window<T, K, true, 1><<<gridw, blockw, 0, st>>>(g, in, gk, t);
            HIP_OK(hipGetLastError());
            return 1;
        }
    }
    const dim3 grid(tiles, K, a.B), block(256);
    if (a.CK == 1) CSPN_PRE(st), pac_conv2d_gk_tiled<T, K, true><<<grid, block, 0, st>>>(g, in, gk, a, tiles_x);
    else CSPN_PRE(st), pac_conv2d_gk_tiled<T, K, false><<<grid, block, 0, st>>>(g, in, gk, a, tiles_x);
    HIP_OK(hipGetLastError());
    return 1;
}

template <typename T>
int conv_gk_typed(const void* gout, const void* in, void* gk, ConvArgs a, hipStream_t st) {
    if (s2_fast(a, gout, in, gk))
        return cspn_detail::pac_s2_grad_kernel(gout, in, gk, std::is_same<T, __half>::value ? CSPN_F16 : CSPN_F32, a.kh, s2_args(a), st);
    if (tiled_geometry(a) && !a.force_scalar) {
        const T* g = static_cast<const T*>(gout);
        const T* i = static_cast<const T*>(in);
        T* o = static_cast<T*>(gk);
        return a.kh == 3 ? conv_gk_tiled<T, 3>(g, i, o, a, st)
             : a.kh == 5 ? conv_gk_tiled<T, 5>(g, i, o, a, st) : conv_gk_tiled<T, 7>(g, i, o, a, st);
    }
    if (!a.force_scalar && a.sh == 1 && a.sw == 1 && (size_t)a.H * a.W < ((size_t)1 << 31) &&
        (size_t)a.Ho * a.Wo < ((size_t)1 << 31)) {
        // other unit-stride windows: input patch in LDS, all taps (per-channel kernel) or a group of <= 9 taps (shared kernel:
        // the sum over channels stays in registers) per workgroup.  (The kernel takes strides too, but a strided window's
        // patch is stride^2 larger per output: 85 vs 70 us for the scalar kernel at stride 2.)
        constexpr int NTM = 9;
        const long RWl = ((long)(TILE_W - 1) * a.sw + (long)(a.kw - 1) * a.dw + 1 + 3) & ~3L;
        const long RHl = (long)(TILE_H - 1) * a.sh + (long)(a.kh - 1) * a.dh + 1;
        const long psz = RWl * RHl;
        int cb = (int)std::min<long>(CC, (64 * 1024 / 4) / std::max(psz, 1L));
        cb = std::min(cb, a.C);
        if (cb >= 1) {
            const int tiles_x = ceil_div(a.Wo, TILE_W), tiles = tiles_x * ceil_div(a.Ho, TILE_H);
            const int ntap = a.kh * a.kw;
            const size_t have = (size_t)tiles * a.B;
            const size_t lds = (size_t)cb * psz * sizeof(float);
            const T* g = static_cast<const T*>(gout);
            const T* i = static_cast<const T*>(in);
            T* o = static_cast<T*>(gk);
            if (a.CK == 1) {
                // enough workgroups to fill the chip: split the taps into more groups when there are few tiles
                const int want_groups = (int)std::min<size_t>((size_t)ntap, (ANY_WANT_WGS_GK + have - 1) / have);
                const int tpg = std::max(1, std::min(NTM, ceil_div(ntap, std::max(want_groups, 1))));
                const dim3 grid(tiles, ceil_div(ntap, tpg), a.B), block(256);
                CSPN_PRE(st), pac_conv2d_gk_any<T, true, NTM><<<grid, block, lds, st>>>(g, i, o, a, tiles_x, (int)RWl, (int)RHl, cb, tpg);
            } else {
                int nchunk = (int)std::min<size_t>((1024 + have - 1) / have, (size_t)ceil_div(a.C, cb));
                a.cchunk = ceil_div(ceil_div(a.C, std::max(nchunk, 1)), cb) * cb;
                const dim3 grid(tiles, ceil_div(a.C, a.cchunk), a.B), block(256);
                CSPN_PRE(st), pac_conv2d_gk_any<T, false, NTM><<<grid, block, lds, st>>>(g, i, o, a, tiles_x, (int)RWl, (int)RHl, cb, 0);
            }
            HIP_OK(hipGetLastError());
            return 1;
        }
    }
    const dim3 grid(ceil_div(a.Ho * a.WQ, 256), a.kh * a.kw, a.B), block(256);
    const T* g = static_cast<const T*>(gout);
    const T* i = static_cast<const T*>(in);
    T* o = static_cast<T*>(gk);
    const bool shared = a.CK == 1;
    if (a.vec && shared) CSPN_PRE(st), pac_conv2d_gk<T, true, true><<<grid, block, 0, st>>>(g, i, o, a);
    else if (a.vec) CSPN_PRE(st), pac_conv2d_gk<T, true, false><<<grid, block, 0, st>>>(g, i, o, a);
    else if (shared) CSPN_PRE(st), pac_conv2d_gk<T, false, true><<<grid, block, 0, st>>>(g, i, o, a);
    else CSPN_PRE(st), pac_conv2d_gk<T, false, false><<<grid, block, 0, st>>>(g, i, o, a);
    HIP_OK(hipGetLastError());
    return 1;
}

template <typename T>
int conv_gi_typed(const void* gout, const void* kern, void* gin, ConvArgs a, int in_vec, hipStream_t st) {
    if (s2_fast(a, gout, kern, gin))
        return cspn_detail::pac_s2_grad_input(gout, kern, gin, std::is_same<T, __half>::value ? CSPN_F16 : CSPN_F32, a.kh, s2_args(a), st);
    if (tiled_geometry(a) && !a.force_scalar && (size_t)a.Ho * a.Wo < ((size_t)1 << 31))
        return launch_tiled_k<T, true>(static_cast<const T*>(gout), static_cast<const T*>(kern), static_cast<T*>(gin), a,
                                       in_vec, st);
    if (!a.force_scalar && a.sh == 1 && a.sw == 1 && (size_t)a.H * a.W < ((size_t)1 << 31) &&
        (size_t)a.Ho * a.Wo < ((size_t)1 << 31)) {
        // other unit-stride windows (dilated, non-square, even): the any-geometry tiled kernel, transposed — the tile lies
        // on the input plane, the grad_out patch it gathers from has the extents of the forward's input patch
        const long RWl = ((long)(TILE_W - 1) + (long)(a.kw - 1) * a.dw + 1 + 3) & ~3L;
        const long RHl = (long)(TILE_H - 1) + (long)(a.kh - 1) * a.dh + 1;
        const long psz = RWl * RHl;
        int cb = (int)std::min<long>(CC, (64 * 1024 / 4) / std::max(psz, 1L));
        cb = std::min(cb, a.C);
        if (cb >= 1) {
            const int tiles_x = ceil_div(a.W, TILE_W), tiles = tiles_x * ceil_div(a.H, TILE_H);
            const size_t want = ANY_WANT_WGS, have = (size_t)tiles * a.B;
            int nchunk = (int)std::min<size_t>((want + have - 1) / have, (size_t)ceil_div(a.C, cb));
            a.cchunk = ceil_div(ceil_div(a.C, std::max(nchunk, 1)), cb) * cb;
            a.lin_tiles = tiles; a.lin_chunks = ceil_div(a.C, a.cchunk);
            const dim3 grid = lin_grid(tiles, a.lin_chunks, a.B), block(256);
            const size_t lds = (size_t)cb * psz * sizeof(float);
            const T* g = static_cast<const T*>(gout);
            const T* k = static_cast<const T*>(kern);
            T* o = static_cast<T*>(gin);
            if (a.CK == 1 && a.kh * a.kw <= 9) CSPN_PRE(st), pac_conv2d_fwd_tiled_any<T, true, true, true><<<grid, block, lds, st>>>(g, k, o, a, tiles_x, (int)RWl, (int)RHl, cb);
            else if (a.CK == 1) CSPN_PRE(st), pac_conv2d_fwd_tiled_any<T, true, true, false><<<grid, block, lds, st>>>(g, k, o, a, tiles_x, (int)RWl, (int)RHl, cb);
            else CSPN_PRE(st), pac_conv2d_fwd_tiled_any<T, false, true, false><<<grid, block, lds, st>>>(g, k, o, a, tiles_x, (int)RWl, (int)RHl, cb);
            HIP_OK(hipGetLastError());
            return 1;
        }
    }
    const int in_wq = ceil_div(a.W, 4);
    const int gx = ceil_div(a.H * in_wq, 256);
    a.cchunk = channel_chunk(a.C, (size_t)gx * a.B);
    const dim3 grid(gx, ceil_div(a.C, a.cchunk), a.B), block(256);
    const T* g = static_cast<const T*>(gout);
    const T* k = static_cast<const T*>(kern);
    T* o = static_cast<T*>(gin);
    const bool shared = a.CK == 1, unit = a.sh == 1 && a.sw == 1;
    if (shared && unit) CSPN_PRE(st), pac_conv2d_gi<T, true, true><<<grid, block, 0, st>>>(g, k, o, a, in_wq, in_vec);
    else if (shared) CSPN_PRE(st), pac_conv2d_gi<T, true, false><<<grid, block, 0, st>>>(g, k, o, a, in_wq, in_vec);
    else if (unit) CSPN_PRE(st), pac_conv2d_gi<T, false, true><<<grid, block, 0, st>>>(g, k, o, a, in_wq, in_vec);
    else CSPN_PRE(st), pac_conv2d_gi<T, false, false><<<grid, block, 0, st>>>(g, k, o, a, in_wq, in_vec);
    HIP_OK(hipGetLastError());
    return 1;
}

template <typename T>
int nd2col_typed(const void* in, void* cols, ConvArgs a, hipStream_t st) {
    if ((long)a.B * a.C > 65535) return fail("cspn_pac_nd2col: B*C=%ld exceeds the grid limit 65535", (long)a.B * a.C);
    const dim3 grid(ceil_div(a.Ho * a.WQ, 256), a.kh * a.kw, a.B * a.C), block(256);
    if (a.vec) CSPN_PRE(st), pac_nd2col_kernel<T, true><<<grid, block, 0, st>>>(static_cast<const T*>(in), static_cast<T*>(cols), a);
    else CSPN_PRE(st), pac_nd2col_kernel<T, false><<<grid, block, 0, st>>>(static_cast<const T*>(in), static_cast<T*>(cols), a);
    HIP_OK(hipGetLastError());
    return 1;
}

}  // namespace

extern "C" {

int cspn_pac_force_generic(int on, int* previous_or_null) {
    const int prev = force_generic_flag().exchange(on ? 1 : 0, std::memory_order_relaxed);
    if (previous_or_null) *previous_or_null = prev;
    return 1;
}

int cspn_pac_out_size(int H, int W, const cspn_conv_geometry* geom, int* Ho, int* Wo) {
    if (!geom || !Ho || !Wo) return fail("cspn_pac_out_size: null argument");
    if (H < 1 || W < 1 || geom->kh < 1 || geom->kw < 1 || geom->sh < 1 || geom->sw < 1 || geom->dh < 1 || geom->dw < 1)
        return fail("cspn_pac_out_size: bad geometry");
    if (!out_size(H, W, *geom, Ho, Wo)) return fail("cspn_pac_out_size: empty output");
    return 1;
}

int cspn_pac_conv2d(const void* input, const void* kernel, void* out, int dtype, int B, int C, int kernel_ch,
                    int H, int W, const cspn_conv_geometry* geom, cspn_stream_t stream) {
    ConvArgs a;
    if (!make_args("cspn_pac_conv2d", dtype, B, C, kernel_ch, H, W, geom, false, &a)) return 0;
    if (!input || !kernel || !out) return fail("cspn_pac_conv2d: null pointer");
    a.vec = a.Wo % 4 == 0 && aligned_for(kernel, dtype) && aligned_for(out, dtype);
    hipStream_t st = static_cast<hipStream_t>(stream);
    return dtype == CSPN_F16 ? conv_forward_typed<__half>(input, kernel, out, a, st)
                             : conv_forward_typed<float>(input, kernel, out, a, st);
}

int cspn_pac_conv2d_grad_kernel(const void* grad_out, const void* input, void* grad_kernel, int dtype, int B, int C,
                                int kernel_ch, int H, int W, const cspn_conv_geometry* geom, cspn_stream_t stream) {
    ConvArgs a;
    if (!make_args("cspn_pac_conv2d_grad_kernel", dtype, B, C, kernel_ch, H, W, geom, false, &a)) return 0;
    if (!grad_out || !input || !grad_kernel) return fail("cspn_pac_conv2d_grad_kernel: null pointer");
    a.vec = a.Wo % 4 == 0 && aligned_for(grad_out, dtype) && aligned_for(grad_kernel, dtype);
    hipStream_t st = static_cast<hipStream_t>(stream);
    return dtype == CSPN_F16 ? conv_gk_typed<__half>(grad_out, input, grad_kernel, a, st)
                             : conv_gk_typed<float>(grad_out, input, grad_kernel, a, st);
}

int cspn_pac_conv2d_grad_input(const void* grad_out, const void* kernel, void* grad_input, int dtype, int B, int C,
                               int kernel_ch, int H, int W, const cspn_conv_geometry* geom, cspn_stream_t stream) {
    ConvArgs a;
    if (!make_args("cspn_pac_conv2d_grad_input", dtype, B, C, kernel_ch, H, W, geom, false, &a)) return 0;
    if (!grad_out || !kernel || !grad_input) return fail("cspn_pac_conv2d_grad_input: null pointer");
    const int in_vec = W % 4 == 0 && aligned_for(grad_input, dtype);
    hipStream_t st = static_cast<hipStream_t>(stream);
    return dtype == CSPN_F16 ? conv_gi_typed<__half>(grad_out, kernel, grad_input, a, in_vec, st)
                             : conv_gi_typed<float>(grad_out, kernel, grad_input, a, in_vec, st);
}

int cspn_pac_nd2col(const void* input, void* cols, int dtype, int B, int C, int H, int W,
                    const cspn_conv_geometry* geom, cspn_stream_t stream) {
    ConvArgs a;
    if (!make_args("cspn_pac_nd2col", dtype, B, C, 1, H, W, geom, true, &a)) return 0;
    if (!input || !cols) return fail("cspn_pac_nd2col: null pointer");
    a.vec = a.Wo % 4 == 0 && aligned_for(cols, dtype);
    hipStream_t st = static_cast<hipStream_t>(stream);
    return dtype == CSPN_F16 ? nd2col_typed<__half>(input, cols, a, st) : nd2col_typed<float>(input, cols, a, st);
}

}  // extern "C"
