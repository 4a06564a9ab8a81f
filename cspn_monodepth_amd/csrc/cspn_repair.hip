// cspn_repair.hip — the guard behind a weight-resident inference launch (include/cspn_hip.h: cspn_resident_plan.guard).
//
// A resident launch whose workgroups could not all become co-resident in time (the GPU shared with a tenant that holds CUs for
// seconds) gives up: its abort word receives the call's sequence number and the tiles that gave up leave the poison NaN.  The host
// repairs that where the package is the next consumer (functional.py: journal) — but the reference module is plain ATen
// (network/libs/post_process/CSPN_new.py:80-92): whatever consumes its output ON THE GPU sees the finished tensor.  This kernel
// restores that property from the stream: launched behind the resident launch(es) of a call, every workgroup reads the abort word and
// returns at once when it is not this call's sequence number (the success path: one empty launch, priced in DESIGN.md §4.1b);
// otherwise the whole batch is re-computed here — slowly, but with the resident kernel's arithmetic, operation for operation
// (|g| gathered with the reference's shifts, S summed in channel order, div8_shared_reciprocal, the blend folded into the taps,
// the eight FMAs in tap order from 0, + m * d0), so the repaired result is bit-identical to both schedules.
//
// Re-computation: a workgroup takes a 32 x 32 output tile and the T-pixel halo its T steps depend on ((32 + 2T)^2 floats twice in
// LDS: T <= 54), and advances the region step by step; the taps of a pixel are rebuilt from the guidance at every step (no
// register residency, no exchange between workgroups — nothing here can wait for anybody).
#include "cspn_common.hpp"
#include "cspnk_helpers.hpp"      // the dot-product form's arithmetic (shared with cspnk_d2.hip: the same code, the same bits)

#include <atomic>

namespace {

constexpr int REP_TILE = 32, REP_THREADS = 256;

struct RepArgs {
    const float* g; long g_bs, g_cs;
    const float* d0; const float* sparse; float* out;
    float* hist; float* s_out; float* w_out; const float* s_in;
    const float* target; double* macc; int nslots;      // MODE 1: the fused depth metrics of the pixels the failed launch did not score
    const unsigned* abort_word; unsigned seq;
    int B, H, W, Wv, T, tiles_x, tiles_y;
};

// the forward's refined reciprocal of the normaliser (div8_shared_reciprocal's recipe), as cspn3_resident MODE 4 forms it
__device__ __forceinline__ float rcp_as_forward(float S) {
    const bool okr = (S <= 0x1p+100f) && (S >= 0x1p-100f || S == 0.f);
    float r = __builtin_amdgcn_rcpf(S);
    r = fmaf(fmaf(-S, r, 1.0f), r, r);
    return okr ? r : 1.0f / S;
}

// MODE 0: inference (refined depth -> out).  MODE 1: inference with the depth metrics fused — the launch that gave up has scored the
// tiles that finished and poisoned the others, so the re-computation adds the metric terms of exactly the pixels it finds poisoned
// (summation order differs from the fused epilogue's: the sums agree to ~1e-7, the refined depth to the bit).  MODE 2: the training forward — every step's state goes to its history plane, the
// normaliser S (and, when asked for, the taps) is published.  MODE 4: the volume-free reverse sweep G_t = stencil^T((1-m) G_{t+1}) on
// the taps |g_j[p]| / S[p + off_j]; d0 is G_T, the history receives G_{T-1} .. G_0, (1-m) G travels.  MODE 3: the same sweep on taps
// gathered from a forward tap volume (a.g = [B,8,H,W]): tap j = w_{7-j}[p + off_j].
// PAC = 1 (MODE 0 / 2): the K = 3 softmax form of CSPN_ours.py:35-41 — tap j = softmax over the 8 guidance channels at the pixel
// itself, with the arithmetic of cspn3_resident's PAC instances (maximum, two-piece exponential, sum in channel order, one refined
// reciprocal); MODE 2 publishes the taps (no S).
template <int BLEND, int MODE, int PAC>
__global__ __launch_bounds__(REP_THREADS) void cspn3_resident_repair(const RepArgs a) {
    if (__hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.seq) return;      // the call finished: nothing to do
    constexpr bool TRANSG = MODE == 4, TRANS = MODE == 3 || MODE == 4, HIST = MODE >= 2, SCORE = MODE == 1;
    float mf[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) mf[k] = 0.f;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int T = a.T, R = REP_TILE + 2 * T, H = a.H, W = a.W, Wv = a.Wv;
    float* cur = lds;
    float* nxt = lds + (size_t)R * R;
    const int tiles = a.tiles_x * a.tiles_y;
    const size_t HW = (size_t)H * W, plane = (size_t)a.B * HW;
    for (int t = blockIdx.x; t < a.B * tiles; t += gridDim.x) {
        const int b = t / tiles, tr = t - b * tiles, ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
        const int ry0 = ty * REP_TILE - T, rx0 = tx * REP_TILE - T;       // image coordinates of region (0, 0)
        const float* __restrict__ gb = a.g + (size_t)b * a.g_bs;
        const float* __restrict__ db = a.d0 + b * HW;
        const float* __restrict__ sb = BLEND ? a.sparse + b * HW : nullptr;
        const float* __restrict__ sib = TRANSG ? a.s_in + b * HW : nullptr;
        __syncthreads();                                                   // (the previous tile's readers are done)
        for (int i = threadIdx.x; i < R * R; i += REP_THREADS) {
            const int ry = i / R, rx = i - ry * R, y = ry0 + ry, x = rx0 + rx;
            const bool in = y >= 0 && y < H && x >= 0 && x < Wv;
            float v = in ? db[(size_t)y * W + x] : 0.f;
            if (TRANS && BLEND && in) v *= 1.f - sgnf(sb[(size_t)y * W + x]);      // the sweep starts from (1-m) G_T
            cur[i] = v;
            nxt[i] = 0.f;
        }
        __syncthreads();
        for (int s = 1; s <= T; ++s) {
            // after step s the pixels within T - s of the tile are exact: only those are advanced
            const int lo = s, n = R - 2 * s;
            float* const hp = HIST ? a.hist + (size_t)(s - 1) * plane + b * HW : nullptr;
            for (int i = threadIdx.x; i < n * n; i += REP_THREADS) {
                const int ry = lo + i / n, rx = lo + i % n, y = ry0 + ry, x = rx0 + rx;
                float keep = 0.f;
                if (y >= 0 && y < H && x >= 0 && x < Wv) {
                    const bool mine = ry >= T && ry < T + REP_TILE && rx >= T && rx < T + REP_TILE;      // a pixel of the tile itself
                    float qv[8];
                    float m = 0.f;
                    if (BLEND) m = sgnf(sb[(size_t)y * W + x]);
                    if (TRANSG) {
                        // tap j = |channel j at p| x 1 / S[p + off_j] (the forward's refined reciprocal), 0 where p + off_j is outside
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int lin = j < 4 ? j : j + 1, yy = y + lin / 3 - 1, xx = x + lin % 3 - 1;
                            const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < Wv;
                            const float rs = ok ? rcp_as_forward(sib[(size_t)yy * W + xx]) : 0.f;
                            qv[j] = fabsf(gb[(size_t)j * a.g_cs + (size_t)y * W + x]) * rs;
                        }
                    } else if (TRANS) {
                        // tap j = w_{7-j}[p + off_j] from the forward's tap volume, 0 outside the image
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int lin = j < 4 ? j : j + 1, yy = y + lin / 3 - 1, xx = x + lin % 3 - 1;
                            const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
                            qv[j] = ok ? gb[(size_t)(7 - j) * a.g_cs + (size_t)yy * W + xx] : 0.f;
                        }
                    } else if (PAC) {
                        float mx = -INFINITY, den = 0.f;
#pragma unroll
                        for (int j = 0; j < 8; ++j) { qv[j] = gb[(size_t)j * a.g_cs + (size_t)y * W + x]; mx = fmaxf(mx, qv[j]); }
#pragma unroll
                        for (int j = 0; j < 8; ++j) { qv[j] = softmax_exp<float>(qv[j] - mx); den += qv[j]; }
                        const float inv = reciprocal_refined(den);
#pragma unroll
                        for (int j = 0; j < 8; ++j) qv[j] = softmax_weight<float>(qv[j], inv);
                        if (MODE == 2 && mine && s == 1) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) a.w_out[((size_t)b * 8 + j) * HW + (size_t)y * W + x] = qv[j];
                        }
                        if (BLEND) {
                            const float om = 1.f - m;
#pragma unroll
                            for (int j = 0; j < 8; ++j) qv[j] *= om;
                        }
                    } else {
                        // taps as cspn3_resident derives them: tap j (row-major without the centre) = |channel 7-j| at p + off_j, 0
                        // outside the image (rows: [0, H); columns: [0, W) — the row padding [Wv, W) holds the caller's zeros)
                        float av[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int lin = j < 4 ? j : j + 1, yy = y + lin / 3 - 1, xx = x + lin % 3 - 1;
                            const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
                            av[j] = ok ? fabsf(gb[(size_t)(7 - j) * a.g_cs + (size_t)yy * W + xx]) : 0.f;
                        }
                        float S = av[7];
#pragma unroll
                        for (int k = 1; k < 8; ++k) S += av[7 - k];          // the reference's channel order (CSPN_new.py:124-127)
                        div8_shared_reciprocal(av, S, qv);
                        if (MODE == 2 && mine && s == 1) {                      // published before the blend is folded in
                            a.s_out[b * HW + (size_t)y * W + x] = S;
                            if (a.w_out) {
#pragma unroll
                                for (int j = 0; j < 8; ++j) a.w_out[((size_t)b * 8 + j) * HW + (size_t)y * W + x] = qv[j];
                            }
                        }
                        if (BLEND) {
                            const float om = 1.f - m;                          // 0, 1 or 2: exact
#pragma unroll
                            for (int j = 0; j < 8; ++j) qv[j] *= om;
                        }
                    }
                    float u = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int lin = j < 4 ? j : j + 1;
                        u = fmaf(qv[j], cur[(ry + lin / 3 - 1) * R + rx + lin % 3 - 1], u);
                    }
                    if (TRANS) {
                        keep = BLEND ? __fmul_rn(1.f - m, u) : u;              // (1-m) G travels, G itself goes to the history
                    } else {
                        // + m * d0: a product rounded on its own, then an addition (the resident kernel keeps m * d0 in LDS) — never one FMA
                        if (BLEND) u = __fadd_rn(u, __fmul_rn(m, db[(size_t)y * W + x]));
                        keep = u;
                    }
                    if (HIST && mine) hp[(size_t)y * W + x] = u;
                }
                else if (MODE == 2 && !PAC && s == 1 && y >= 0 && y < H && x >= Wv && x < W && (x & ~3) < Wv && ry >= T && ry < T + REP_TILE && rx >= T && rx < T + REP_TILE) {
                    // row padding inside the last valid quad: the resident launch stores that quad's S (and zero taps) as a whole
                    float S = 0.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int j = 7 - k, lin = j < 4 ? j : j + 1, yy = y + lin / 3 - 1, xx = x + lin % 3 - 1;
                        const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
                        const float av = ok ? fabsf(gb[(size_t)(7 - j) * a.g_cs + (size_t)yy * W + xx]) : 0.f;
                        S = k == 0 ? av : S + av;
                    }
                    a.s_out[b * HW + (size_t)y * W + x] = S;
                    if (a.w_out) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) a.w_out[((size_t)b * 8 + j) * HW + (size_t)y * W + x] = 0.f;
                    }
                }
                // row padding inside the last valid quad: the resident launch stores (and, giving up, poisons) that quad as a whole, with
                // zeros in its padding columns — a C-API reader of the repaired history plane must find them again
                if (HIST && y >= 0 && y < H && x >= Wv && x < W && (x & ~3) < Wv && ry >= T && ry < T + REP_TILE && rx >= T && rx < T + REP_TILE)
                    hp[(size_t)y * W + x] = 0.f;
                nxt[ry * R + rx] = keep;
            }
            __syncthreads();
            float* tmp = cur; cur = nxt; nxt = tmp;
        }
        if (!HIST) {
            float* __restrict__ ob = a.out + b * HW;
            for (int i = threadIdx.x; i < REP_TILE * REP_TILE; i += REP_THREADS) {
                const int ly = i / REP_TILE, lx = i - ly * REP_TILE, y = ty * REP_TILE + ly, x = tx * REP_TILE + lx;
                if (y < H && x < Wv) {
                    const float v = cur[(T + ly) * R + T + lx];
                    if (SCORE && __float_as_uint(ob[(size_t)y * W + x]) == CSPN_POISON_F32) metric_terms(v, a.target[b * HW + (size_t)y * W + x], mf);
                    ob[(size_t)y * W + x] = v;
                } else if (y < H && x < W && (x & ~3) < Wv) {
                    ob[(size_t)y * W + x] = 0.f;      // padding columns of the last valid quad: zeros, as the clean launch stores them
                }
            }
        }
    }
    if (SCORE) {
        __syncthreads();
        float* part = lds;
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const float v = wave_sum_to_lane63(mf[k]);
            if (lane == 63) part[wave * 10 + k] = v;
        }
        __syncthreads();
        if (threadIdx.x < 10) {
            double v = 0.0;
            for (int w = 0; w < REP_THREADS / 64; ++w) v += (double)part[w * 10 + threadIdx.x];
            if (v != 0.0) atomicAdd(a.macc + (size_t)(blockIdx.x % a.nslots) * 10 + threadIdx.x, v);
        }
    }
}


// ------------------------------------------------------------------------------------------------ K x K softmax forms (CSPN_ours)
// The guard of cspnk_forward_resident's unscored inference calls (round 5): same structure — 32 x 32 output tiles with the T * (K / 2)
// pixel halo of T steps in LDS, nothing to wait for — with the arithmetic of the FMA kernel cspnk_resident: softmax over the K*K-1
// channels at the pixel itself (fp16 guidance: float(half) - max, 2^(d log2 e), sum in channel order, refined reciprocal, ONE rounding
// to half; fp32 guidance: the two-piece exponential), (1-m) folded into the taps, one FMA per tap in row-major tap order from 0,
// + m * x0, the state rounded to the plane dtype where that kernel rounds it: after every `round_every` steps (its phase length) and
// at the end.  For the FMA step form the re-computed depth is therefore the multi-launch schedule's bits; for the dot-product form
// (cspnk_d2: state rounded after EVERY step, two taps per v_dot2_f32_f16) round_every = 1 gives the half-precision recurrence with
// one FMA per tap — within the fp16 tolerance of the configuration, like the host repair of that form.
struct KRepArgs {
    const void* g; const void* x0; const void* sparse; void* out;
    const unsigned* abort_word; unsigned seq;
    int B, H, W, T, round_every, tiles_x, tiles_y;
};
template <typename T> __device__ __forceinline__ float ldf(const void* p, size_t i);
template <> __device__ __forceinline__ float ldf<float>(const void* p, size_t i) { return static_cast<const float*>(p)[i]; }
template <> __device__ __forceinline__ float ldf<__half>(const void* p, size_t i) { return __half2float(static_cast<const __half*>(p)[i]); }
template <typename T> __device__ __forceinline__ float round_to(float v);
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_to<__half>(float v) { return __half2float(__float2half_rn(v)); }
__device__ __forceinline__ void stf(float* p, size_t i, float v) { p[i] = v; }
__device__ __forceinline__ void stf(__half* p, size_t i, float v) { p[i] = __float2half_rn(v); }

// D2 = 1 (K = 5, fp16 guidance, fp16 planes; round 6): the re-computation of the DOT-PRODUCT form (cspnk_d2.hip) with that kernel's own
// arithmetic — the softmax as softmax_to_pairs forms it (max folded into the exponent's scale, v_exp_f32, sum in channel order, refined
// reciprocal, numerator x 1 / sum rounded by v_cvt_pk_f16_f32), the taps as tap PAIRS, a window row as dot2(pair dx -2 -1), dot2(pair dx 0 +1)
// [centre row: dx +1 +2], then the single dx +2 tap as v_fma_mix_f32, rows top to bottom, + m x0 as the sum's start value, one v_cvt_pk_f16_f32
// per step — so that a timed-out call of config 3's shape returns the BITS of a clean one (until round 5 "within its
// fp16 tolerance").
template <int K, typename GT, typename ST, int BLEND, int D2 = 0>
__global__ __launch_bounds__(REP_THREADS) void cspnk_resident_repair(const KRepArgs a) {
    if (__hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.seq) return;      // the call finished: nothing to do
    constexpr int RK = K / 2, NT = K * K - 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int T = a.T, HALO = T * RK, R = REP_TILE + 2 * HALO, H = a.H, W = a.W;
    float* cur = lds;
    float* nxt = lds + (size_t)R * R;
    const int tiles = a.tiles_x * a.tiles_y;
    const size_t HW = (size_t)H * W;
    for (int t = blockIdx.x; t < a.B * tiles; t += gridDim.x) {
        const int b = t / tiles, tr = t - b * tiles, ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
        const int ry0 = ty * REP_TILE - HALO, rx0 = tx * REP_TILE - HALO;
        const size_t gbase = (size_t)b * NT * HW, pbase = (size_t)b * HW;
        __syncthreads();
        for (int i = threadIdx.x; i < R * R; i += REP_THREADS) {
            const int ry = i / R, rx = i - ry * R, y = ry0 + ry, x = rx0 + rx;
            const bool in = y >= 0 && y < H && x >= 0 && x < W;
            cur[i] = in ? ldf<ST>(a.x0, pbase + (size_t)y * W + x) : 0.f;
            nxt[i] = 0.f;
        }
        __syncthreads();
        for (int s = 1; s <= T; ++s) {
            const int lo = s * RK, n = R - 2 * lo;
            const bool round_now = (s % a.round_every) == 0 || s == T;
            for (int i = threadIdx.x; i < n * n; i += REP_THREADS) {
                const int ry = lo + i / n, rx = lo + i % n, y = ry0 + ry, x = rx0 + rx;
                float u = 0.f;
                if constexpr (D2 != 0) {
                    if (y >= 0 && y < H && x >= 0 && x < W) {
                        const size_t p = (size_t)y * W + x;
                        unsigned gw[24], tp[12];
                        unsigned mx2 = 0xfc00fc00u;
#pragma unroll
                        for (int c = 0; c < 24; ++c) {
                            gw[c] = (unsigned)__half_as_ushort(static_cast<const __half*>(a.g)[gbase + (size_t)c * HW + p]);
                            mx2 = pk_max_f16(mx2, gw[c] | 0xfc000000u);           // (the high half stays -inf)
                        }
                        constexpr float L2E = 1.44269502162933349609375f;
                        softmax_to_pairs<0>(gw, -h2f_lo(mx2) * L2E, tp);
                        float acc = 0.f;
                        if (BLEND) {
                            const float m = sgnf(ldf<ST>(a.sparse, pbase + p));
                            acc = m * ldf<ST>(a.x0, pbase + p);
                            const unsigned om = pack_h2(1.f - m, 1.f - m);
#pragma unroll
                            for (int t = 0; t < 12; ++t) tp[t] = pk_mul_f16(tp[t], om);
                        }
                        asm volatile("" : "+v"(tp[10]), "+v"(tp[11]));        // (as the step loop of cspnk_d2: see mix_hh)
#pragma unroll
                        for (int rr = 0; rr < 5; ++rr) {
                            const float* row = cur + (ry + rr - 2) * R + rx;
                            const unsigned pm = pack_h2(row[-2], row[-1]);       // the state is half-rounded every step: packing is exact
                            if (rr == 2) {
                                acc = dot2(tp[4], pm, acc);
                                acc = dot2(tp[5], pack_h2(row[1], row[2]), acc);
                            } else {
                                const int s0 = rr < 2 ? 2 * rr : 6 + 2 * (rr - 3);
                                acc = dot2(tp[s0], pm, acc);
                                acc = dot2(tp[s0 + 1], pack_h2(row[0], row[1]), acc);
                                const unsigned ps = pack_h2(row[2], 0.f);
                                if (rr == 0) acc = mix_hh<0, 0>(tp[10], ps, acc);
                                if (rr == 1) acc = mix_hh<1, 0>(tp[10], ps, acc);
                                if (rr == 3) acc = mix_hh<0, 0>(tp[11], ps, acc);
                                if (rr == 4) acc = mix_hh<1, 0>(tp[11], ps, acc);
                            }
                        }
                        u = h2f_lo(cvt_pk_f16(acc, 0.f));
                    }
                } else
                if (y >= 0 && y < H && x >= 0 && x < W) {
                    const size_t p = (size_t)y * W + x;
                    float w[NT];
                    float mx = -INFINITY, den = 0.f;
#pragma unroll
                    for (int c = 0; c < NT; ++c) { w[c] = ldf<GT>(a.g, gbase + (size_t)c * HW + p); mx = fmaxf(mx, w[c]); }
#pragma unroll
                    for (int c = 0; c < NT; ++c) { w[c] = softmax_exp<GT>(w[c] - mx); den += w[c]; }
                    const float inv = reciprocal_refined(den);
                    float m = 0.f;
                    if (BLEND) m = sgnf(ldf<ST>(a.sparse, pbase + p));
#pragma unroll
                    for (int c = 0; c < NT; ++c) {
                        w[c] = softmax_weight<GT>(w[c], inv);
                        if (BLEND) w[c] = round_to<GT>(w[c] * (1.f - m));      // 1-m in {0, 1, 2}: exact in either tap dtype
                    }
#pragma unroll
                    for (int c = 0; c < NT; ++c) {
                        const int lin = c < NT / 2 ? c : c + 1, dy = lin / K - RK, dx = lin % K - RK;
                        u = fmaf(w[c], cur[(ry + dy) * R + rx + dx], u);
                    }
                    if (BLEND) u = __fadd_rn(u, __fmul_rn(m, ldf<ST>(a.x0, pbase + p)));
                    // (opaque: left to the compiler, the last FMA and the conversion fuse into ONE v_fma_mixlo_f16 — a single rounding of
                    // the exact sum where cspnk_resident rounds twice, fp32 then half: 2e-4 of the pixels differed by a half ulp)
                    asm volatile("" : "+v"(u));
                    if (round_now) u = round_to<ST>(u);
                }
                nxt[ry * R + rx] = u;
            }
            __syncthreads();
            float* tmp = cur; cur = nxt; nxt = tmp;
        }
        ST* ob = static_cast<ST*>(a.out) + pbase;
        for (int i = threadIdx.x; i < REP_TILE * REP_TILE; i += REP_THREADS) {
            const int ly = i / REP_TILE, lx = i - ly * REP_TILE, y = ty * REP_TILE + ly, x = tx * REP_TILE + lx;
            if (y < H && x < W) stf(ob, (size_t)y * W + x, cur[(HALO + ly) * R + HALO + lx]);
        }
    }
}


// ---- the K = 5 fp16 training forms (BASELINE config 3's shape; round 5) ------------------------------------------------------------
// Guard of cspnk_forward_resident_history's dot-product launch (cspnk_d2<HIST>): history [T,B,H,W] receives x_1 .. x_T, the state rounded
// to the plane dtype after EVERY step, wk_out the softmax taps BEFORE the blend is folded in, in the Taps<__half> layout — the arithmetic
// of cspn_pac_prepare + cspn_propagate (history) at one step per launch (one FMA per tap), i.e. what the multi-launch training path
// stores; the dot-product kernel's own bits are within the fp16 tolerance of it, and the backward differentiates whichever history and
// taps it is handed.
struct KHistRepArgs {
    const void* g; const void* x0; const void* sparse; void* hist; void* wk_out;
    const unsigned* abort_word; unsigned seq;
    int B, H, W, T, tiles_x, tiles_y;
};

template <int K, typename GT, typename ST, int BLEND>
__global__ __launch_bounds__(REP_THREADS) void cspnk_history_repair(const KHistRepArgs a) {
    if (__hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.seq) return;
    constexpr int RK = K / 2, NT = K * K - 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int T = a.T, HALO = T * RK, R = REP_TILE + 2 * HALO, H = a.H, W = a.W;
    float* cur = lds;
    float* nxt = lds + (size_t)R * R;
    const int tiles = a.tiles_x * a.tiles_y;
    const size_t HW = (size_t)H * W, plane = (size_t)a.B * HW;
    for (int t = blockIdx.x; t < a.B * tiles; t += gridDim.x) {
        const int b = t / tiles, tr = t - b * tiles, ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
        const int ry0 = ty * REP_TILE - HALO, rx0 = tx * REP_TILE - HALO;
        const size_t gbase = (size_t)b * NT * HW, pbase = (size_t)b * HW;
        GT* const wkb = static_cast<GT*>(a.wk_out) + (size_t)b * Taps<GT>::image_elems(NT, HW);
        __syncthreads();
        for (int i = threadIdx.x; i < R * R; i += REP_THREADS) {
            const int ry = i / R, rx = i - ry * R, y = ry0 + ry, x = rx0 + rx;
            const bool in = y >= 0 && y < H && x >= 0 && x < W;
            cur[i] = in ? ldf<ST>(a.x0, pbase + (size_t)y * W + x) : 0.f;
            nxt[i] = 0.f;
        }
        __syncthreads();
        for (int s = 1; s <= T; ++s) {
            const int lo = s * RK, n = R - 2 * lo;
            for (int i = threadIdx.x; i < n * n; i += REP_THREADS) {
                const int ry = lo + i / n, rx = lo + i % n, y = ry0 + ry, x = rx0 + rx;
                float u = 0.f;
                if (y >= 0 && y < H && x >= 0 && x < W) {
                    const size_t p = (size_t)y * W + x;
                    const bool own = ry >= HALO && ry < HALO + REP_TILE && rx >= HALO && rx < HALO + REP_TILE;
                    float w[NT];
                    float mx = -INFINITY, den = 0.f;
#pragma unroll
                    for (int c = 0; c < NT; ++c) { w[c] = ldf<GT>(a.g, gbase + (size_t)c * HW + p); mx = fmaxf(mx, w[c]); }
#pragma unroll
                    for (int c = 0; c < NT; ++c) { w[c] = softmax_exp<GT>(w[c] - mx); den += w[c]; }
                    const float inv = reciprocal_refined(den);
                    float m = 0.f;
                    if (BLEND) m = sgnf(ldf<ST>(a.sparse, pbase + p));
#pragma unroll
                    for (int c = 0; c < NT; ++c) {
                        w[c] = softmax_weight<GT>(w[c], inv);
                        if (s == 1 && own) stf(wkb, Taps<GT>::idx(c, p, HW), w[c]);      // the published taps: before the fold
                        if (BLEND) w[c] = round_to<GT>(w[c] * (1.f - m));
                    }
#pragma unroll
                    for (int c = 0; c < NT; ++c) {
                        const int lin = c < NT / 2 ? c : c + 1, dy = lin / K - RK, dx = lin % K - RK;
                        u = fmaf(w[c], cur[(ry + dy) * R + rx + dx], u);
                    }
                    if (BLEND) u = __fadd_rn(u, __fmul_rn(m, ldf<ST>(a.x0, pbase + p)));
                    asm volatile("" : "+v"(u));                  // (no fma + cvt fusion: cspnk_resident_repair has the note)
                    u = round_to<ST>(u);
                    if (own) stf(static_cast<ST*>(a.hist) + (size_t)(s - 1) * plane + pbase, p, u);
                }
                nxt[ry * R + rx] = u;
            }
            __syncthreads();
            float* tmp = cur; cur = nxt; nxt = tmp;
        }
    }
}

// Guard of cspnk_transposed_resident: G_t = stencil^T((1-m) G_{t+1}) with the transposed taps w_{NT-1-j}[p + off_j] read from the forward's
// tap volume (zero where p + off_j leaves the image), one FMA per tap in row-major tap order from 0, fp32 state; history [T,B,H,W] f32
// receives G_{T-1} .. G_0 (the UNMASKED G_t; what travels is (1-m) G_t), g32_out — when given — G_T as fp32.  The arithmetic of
// cspnk_resident<TRANS>, i.e. of cspn_transpose_weights + cspn_propagate (history, CSPN_BLEND_PREMASK), bit for bit.
struct KSweepRepArgs {
    const void* wk; const void* g_T; const void* sparse; float* g32_out; float* hist;
    const unsigned* abort_word; unsigned seq;
    int B, H, W, T, tiles_x, tiles_y;
};

template <int K, typename WT, typename INT, int PREMASK>
__global__ __launch_bounds__(REP_THREADS) void cspnk_sweep_repair(const KSweepRepArgs a) {
    if (__hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.seq) return;
    constexpr int RK = K / 2, NT = K * K - 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int T = a.T, HALO = T * RK, R = REP_TILE + 2 * HALO, H = a.H, W = a.W;
    float* cur = lds;
    float* nxt = lds + (size_t)R * R;
    const int tiles = a.tiles_x * a.tiles_y;
    const size_t HW = (size_t)H * W, plane = (size_t)a.B * HW;
    for (int t = blockIdx.x; t < a.B * tiles; t += gridDim.x) {
        const int b = t / tiles, tr = t - b * tiles, ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
        const int ry0 = ty * REP_TILE - HALO, rx0 = tx * REP_TILE - HALO;
        const size_t pbase = (size_t)b * HW;
        const WT* const wkb = static_cast<const WT*>(a.wk) + (size_t)b * Taps<WT>::image_elems(NT, HW);
        __syncthreads();
        for (int i = threadIdx.x; i < R * R; i += REP_THREADS) {
            const int ry = i / R, rx = i - ry * R, y = ry0 + ry, x = rx0 + rx;
            const bool in = y >= 0 && y < H && x >= 0 && x < W;
            float v = 0.f;
            if (in) {
                const size_t p = (size_t)y * W + x;
                v = ldf<INT>(a.g_T, pbase + p);
                if (a.g32_out && ry >= HALO && ry < HALO + REP_TILE && rx >= HALO && rx < HALO + REP_TILE) a.g32_out[pbase + p] = v;
                if (PREMASK) v = __fmul_rn(v, 1.f - sgnf(ldf<INT>(a.sparse, pbase + p)));
            }
            cur[i] = v;
            nxt[i] = 0.f;
        }
        __syncthreads();
        for (int s = 1; s <= T; ++s) {
            const int lo = s * RK, n = R - 2 * lo;
            for (int i = threadIdx.x; i < n * n; i += REP_THREADS) {
                const int ry = lo + i / n, rx = lo + i % n, y = ry0 + ry, x = rx0 + rx;
                float u = 0.f;
                if (y >= 0 && y < H && x >= 0 && x < W) {
                    const size_t p = (size_t)y * W + x;
#pragma unroll
                    for (int c = 0; c < NT; ++c) {
                        const int lin = c < NT / 2 ? c : c + 1, dy = lin / K - RK, dx = lin % K - RK;
                        const int ys = y + dy, xs = x + dx;
                        const bool tin = ys >= 0 && ys < H && xs >= 0 && xs < W;
                        const float wt = tin ? ldf<WT>(wkb, Taps<WT>::idx(NT - 1 - c, (size_t)ys * W + xs, HW)) : 0.f;
                        u = fmaf(wt, cur[(ry + dy) * R + rx + dx], u);
                    }
                    if (ry >= HALO && ry < HALO + REP_TILE && rx >= HALO && rx < HALO + REP_TILE) a.hist[(size_t)(s - 1) * plane + pbase + p] = u;
                    if (PREMASK) u = __fmul_rn(1.f - sgnf(ldf<INT>(a.sparse, pbase + p)), u);
                }
                nxt[ry * R + rx] = u;
            }
            __syncthreads();
            float* tmp = cur; cur = nxt; nxt = tmp;
        }
    }
}

}  // namespace

namespace cspn_detail {

bool resident_repair_fits(int T) { return T >= 1 && (size_t)2 * (REP_TILE + 2 * T) * (REP_TILE + 2 * T) * sizeof(float) <= 160 * 1024; }

int resident_repair_launch(const float* g, long bs, long cs, const float* d0, const float* sparse, float* out, float* hist, float* s_out,
                           float* w_out, const float* s_in, int mode, const unsigned* abort_word, unsigned seq, int B, int H, int W, int Wv,
                           int T, int blend, int n_cu, void* stream, const float* target, double* acc, int nslots) {
    if (!resident_repair_fits(T)) return fail("cspn3_forward_resident: the guard re-computes at most 54 steps (T=%d)", T);
    if (mode != 0 && mode != 1 && mode != 2 && mode != 3 && mode != 4 && mode != 10 && mode != 12) return fail("cspn3_forward_resident: the guard has no form for this launch");
    RepArgs a{g, bs, cs, d0, sparse, out, hist, s_out, w_out, s_in, target, acc, nslots > 0 ? nslots : 1, abort_word, seq, B, H, W, Wv, T, ceil_div(W, REP_TILE), ceil_div(H, REP_TILE)};
    const size_t lds = (size_t)2 * (REP_TILE + 2 * T) * (REP_TILE + 2 * T) * sizeof(float);
    static std::atomic<size_t> granted[14][64];
    int dev = 0;
    HIP_OK(hipGetDevice(&dev));
    // slots: (inference, training forward, sweep from guidance + S, sweep from a tap volume, softmax inference, softmax training forward) x blend
    const int form = mode == 0 ? 0 : mode == 2 ? 1 : mode == 4 ? 2 : mode == 3 ? 3 : mode == 10 ? 4 : mode == 12 ? 5 : 6;
    const int slot = 2 * form + (blend ? 1 : 0);
    void (*kern)(RepArgs) = nullptr;
    switch (slot) {
        case 0: kern = cspn3_resident_repair<0, 0, 0>; break;
        case 1: kern = cspn3_resident_repair<1, 0, 0>; break;
        case 2: kern = cspn3_resident_repair<0, 2, 0>; break;
        case 3: kern = cspn3_resident_repair<1, 2, 0>; break;
        case 4: kern = cspn3_resident_repair<0, 4, 0>; break;
        case 5: kern = cspn3_resident_repair<1, 4, 0>; break;
        case 6: kern = cspn3_resident_repair<0, 3, 0>; break;
        case 7: kern = cspn3_resident_repair<1, 3, 0>; break;
        case 8: kern = cspn3_resident_repair<0, 0, 1>; break;
        case 9: kern = cspn3_resident_repair<1, 0, 1>; break;
        case 10: kern = cspn3_resident_repair<0, 2, 1>; break;
        case 11: kern = cspn3_resident_repair<1, 2, 1>; break;
        case 12: kern = cspn3_resident_repair<0, 1, 0>; break;
        default: kern = cspn3_resident_repair<1, 1, 0>; break;
    }
    if (lds > 64 * 1024 && granted[slot][dev & 63].load(std::memory_order_acquire) < lds) {
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        granted[slot][dev & 63].store(lds, std::memory_order_release);
    }
    // a few workgroups that loop over the tiles: the success path — every launch of a healthy deployment — is `grid` workgroups
    // that read one word and return, and its cost is their dispatch (rocprofv3, config 2's training step: 5.1 us with one workgroup
    // per CU and 51 KB of LDS each; n_cu / 8 of them — 32 on an MI355X — keep the failure path at tens of milliseconds)
    int grid = B * a.tiles_x * a.tiles_y;
    const int cap = n_cu >= 64 ? n_cu / 8 : 8;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(REP_THREADS), lds, static_cast<hipStream_t>(stream), a);
    HIP_OK(hipGetLastError());
    return 1;
}


bool kres_repair_fits(int K, int T) {
    const int R = REP_TILE + 2 * T * (K / 2);
    return T >= 1 && (K == 3 || K == 5) && (size_t)2 * R * R * sizeof(float) <= 160 * 1024;
}

template <int K, typename GT, typename ST, int D2 = 0>
static int kres_repair_launch_t(const KRepArgs& a, int blend, size_t lds, int grid, hipStream_t st) {
    static std::atomic<size_t> granted[2][64];
    int dev = 0;
    HIP_OK(hipGetDevice(&dev));
    void (*kern)(KRepArgs) = blend ? cspnk_resident_repair<K, GT, ST, 1, D2> : cspnk_resident_repair<K, GT, ST, 0, D2>;
    if (lds > 64 * 1024 && granted[blend ? 1 : 0][dev & 63].load(std::memory_order_acquire) < lds) {
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        granted[blend ? 1 : 0][dev & 63].store(lds, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(REP_THREADS), lds, st, a);
    HIP_OK(hipGetLastError());
    return 1;
}

int kres_repair_launch(const void* g, int g_dtype, int K, const void* x0, const void* sparse, void* out, int state_dtype,
                       const unsigned* abort_word, unsigned seq, int B, int H, int W, int T, int round_every, int blend, int n_cu, void* stream) {
    if (!kres_repair_fits(K, T)) return fail("cspnk_forward_resident: the guard re-computes at most T * (K / 2) = 54 halo pixels (K=%d, T=%d)", K, T);
    KRepArgs a{g, x0, sparse, out, abort_word, seq, B, H, W, T, round_every > 0 ? round_every : T, ceil_div(W, REP_TILE), ceil_div(H, REP_TILE)};
    const int R = REP_TILE + 2 * T * (K / 2);
    const size_t lds = (size_t)2 * R * R * sizeof(float);
    int grid = B * a.tiles_x * a.tiles_y;
    const int cap = n_cu >= 64 ? n_cu / 8 : 8;
    if (grid > cap) grid = cap;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool gh = g_dtype == CSPN_F16, sh = state_dtype == CSPN_F16;
    if (!gh && sh) return fail("cspnk_forward_resident: fp32 guidance with fp16 planes has no kernel");
    if (K == 3) {
        if (gh) return sh ? kres_repair_launch_t<3, __half, __half>(a, blend, lds, grid, st) : kres_repair_launch_t<3, __half, float>(a, blend, lds, grid, st);
        return kres_repair_launch_t<3, float, float>(a, blend, lds, grid, st);
    }
    // round_every == 1 with fp16 guidance and fp16 planes is the dot-product form's launch (cspnk_d2.hip): re-computed with ITS arithmetic
    if (gh && sh && round_every == 1) return kres_repair_launch_t<5, __half, __half, 1>(a, blend, lds, grid, st);
    if (gh) return sh ? kres_repair_launch_t<5, __half, __half>(a, blend, lds, grid, st) : kres_repair_launch_t<5, __half, float>(a, blend, lds, grid, st);
    return kres_repair_launch_t<5, float, float>(a, blend, lds, grid, st);
}

template <typename KernT, typename ArgsT>
static int guarded_launch(KernT kern, std::atomic<size_t>* granted, const ArgsT& a, size_t lds, int n_tiles, int n_cu, hipStream_t st) {
    int dev = 0;
    HIP_OK(hipGetDevice(&dev));
    if (lds > 64 * 1024 && granted[dev & 63].load(std::memory_order_acquire) < lds) {
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        granted[dev & 63].store(lds, std::memory_order_release);
    }
    const int cap = n_cu >= 64 ? n_cu / 8 : 8;
    hipLaunchKernelGGL(kern, dim3(n_tiles > cap ? cap : n_tiles), dim3(REP_THREADS), lds, st, a);
    HIP_OK(hipGetLastError());
    return 1;
}

int kres_history_repair_launch(const void* g, const void* x0, const void* sparse, void* hist, void* wk_out, const unsigned* abort_word, unsigned seq,
                               int B, int H, int W, int T, int blend, int n_cu, void* stream) {
    if (!kres_repair_fits(5, T)) return fail("cspnk_forward_resident_history: the guard re-computes at most 27 steps of a 5 x 5 stencil (T=%d)", T);
    KHistRepArgs a{g, x0, sparse, hist, wk_out, abort_word, seq, B, H, W, T, ceil_div(W, REP_TILE), ceil_div(H, REP_TILE)};
    const int R = REP_TILE + 4 * T;
    static std::atomic<size_t> granted[2][64];
    if (blend) return guarded_launch(cspnk_history_repair<5, __half, __half, 1>, granted[1], a, (size_t)2 * R * R * sizeof(float), B * a.tiles_x * a.tiles_y, n_cu, static_cast<hipStream_t>(stream));
    return guarded_launch(cspnk_history_repair<5, __half, __half, 0>, granted[0], a, (size_t)2 * R * R * sizeof(float), B * a.tiles_x * a.tiles_y, n_cu, static_cast<hipStream_t>(stream));
}

int kres_sweep_repair_launch(const void* wk, const void* g_T, const void* sparse, int in_dtype, float* g32_out, float* hist, const unsigned* abort_word,
                             unsigned seq, int B, int H, int W, int T, int premask, int n_cu, void* stream) {
    if (!kres_repair_fits(5, T)) return fail("cspnk_transposed_resident: the guard re-computes at most 27 steps of a 5 x 5 stencil (T=%d)", T);
    KSweepRepArgs a{wk, g_T, sparse, g32_out, hist, abort_word, seq, B, H, W, T, ceil_div(W, REP_TILE), ceil_div(H, REP_TILE)};
    const int R = REP_TILE + 4 * T;
    const size_t lds = (size_t)2 * R * R * sizeof(float);
    const int n = B * a.tiles_x * a.tiles_y;
    hipStream_t st = static_cast<hipStream_t>(stream);
    static std::atomic<size_t> granted[4][64];
    if (in_dtype == CSPN_F16)
        return premask ? guarded_launch(cspnk_sweep_repair<5, __half, __half, 1>, granted[0], a, lds, n, n_cu, st)
                       : guarded_launch(cspnk_sweep_repair<5, __half, __half, 0>, granted[1], a, lds, n, n_cu, st);
    return premask ? guarded_launch(cspnk_sweep_repair<5, __half, float, 1>, granted[2], a, lds, n, n_cu, st)
                   : guarded_launch(cspnk_sweep_repair<5, __half, float, 0>, granted[3], a, lds, n, n_cu, st);
}

}  // namespace cspn_detail
