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
#include "cspn_common.hpp"

#include <atomic>

namespace {

// ------------------------------------------------------------------------------------------------
// the fused propagation kernel
// ------------------------------------------------------------------------------------------------
struct PropArgs {
    const void* w;       // [B,NT,H,W] tap planes, or (WSRC=1) the guidance tensor itself
    long g_bs, g_cs;     // WSRC=1: guidance batch / channel strides in elements
    void* w_out;         // WSRC=1: optional tap volume receiving the derived weights of the interior quads
    float* s_out;        // WSRC=1: optional [B,H,W] f32 receiving the normaliser S of the interior quads (backward needs it)
    const void* target;  // SCORE=1: ground-truth depth [B,H,W] (DT) scored against the final state
    double* macc;        // SCORE=1: [nslots][10] metric accumulators (cspn_metrics_accumulate layout)
    int nslots;
    const void* d_in;    // [B,H,W]
    void* d_out;         // [B,H,W] state after the last fused step (may be null when hist != null)
    void* hist;          // null, or plane s-1 (stride B*H*W) receives the state after fused step s
    const void* sparse;  // [B,H,W] (blend 1, 2)
    const void* d0;      // [B,H,W] (blend 1)
    int B, H, W, S;
    int Wv;              // logical image width (<= W): columns [Wv, W) are row padding and are kept at exactly 0
    int tw, th, tiles_x, tiles_y;
    int wq, wr;          // weight region: quad columns, rows
    int hxw, hyw;        // weight-region halo (pixels) left/right, top/bottom
    int dr, ls;          // depth region rows, LDS row stride (floats)
};

// Minimum waves per SIMD requested from the register allocator: the one-quad 3x3 instance is held to
// 64 VGPRs (8 waves/SIMD, i.e. 8/4/2 workgroups of 256/512/1024 threads per CU); the others take what they need.
template <int K, int NQ> struct MinWaves { static constexpr int value = (K == 3 && NQ == 1) ? 8 : 1; };

// WSRC = 0: weights are read from prepared tap planes.  WSRC = 1 (3x3 only): the launch derives them from the
// raw guidance itself — |g| of the 8 shifted channels, their sum, the normalisation (exactly the arithmetic of
// cspn3_prepare_kernel, CSPN_new.py:29-70/:124-127) — so inference needs no prepare pass and never
// materialises the 8 weight planes (saves 53 MB written + 53 MB re-read per forward at config 2).
// SCORE = 1: the launch that produces the final state also accumulates the depth metrics of its interior pixels
// against `target` (the reduction cspn_metrics_kernel would do in a separate pass over the output).
template <int K, int NQ, int NTHREADS, typename WT, typename DT, int BLEND, int WSRC, int SCORE = 0>
__global__ __launch_bounds__(NTHREADS, (MinWaves<K, NQ>::value)) void cspn_prop_fused(const PropArgs a) {
    static_assert(WSRC != 1 || K == 3, "on-the-fly weights exist for the 3x3 variant only");
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
    const bool x_in = (xq >= 0) && (xq < a.Wv);   // W % 4 == 0: a quad lies inside the pitch or outside
    const int nval = a.Wv - xq;                   // elements e < nval are inside the logical width (>= 4: all)
    const int lane = tid & 63;
    const bool fix_left = (sx == 0) || (lane == 0);          // left neighbour quad is not lane-1's
    const bool fix_right = (sx == wq - 1) || (lane == 63);   // right neighbour quad is not lane+1's

    // ---- 1. issue the weight stream first (independent of LDS): NQ x NT 16-byte loads -----------
    // f16 tap volumes streamed as prepared weights stay PACKED in registers (one 16-byte load = uint4 = taps 2i, 2i+1
    // of the quad, two halfs per VGPR) and feed v_fma_mix_f32 directly: half the weight registers of the fp32
    // form (higher occupancy / more bytes in flight per wave) and no conversion instructions.
    constexpr bool PACKED = std::is_same<WT, __half>::value && WSRC == 0;
    constexpr bool FOLD = (BLEND == CSPN_BLEND_SPARSE) && !PACKED;   // sparse blend folded into the weight registers
    float wreg[PACKED ? 1 : NQ][PACKED ? 1 : NT][4];
    uint4 wpk[PACKED ? NQ : 1][PACKED ? NT / 2 : 1];
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
            if constexpr (PACKED) {
                const size_t pair_stride = 2 * Taps<__half>::hw4(HW);
#pragma unroll
                for (int jp = 0; jp < NT / 2; ++jp)
                    wpk[i][jp] = ok ? *reinterpret_cast<const uint4*>(wg + (size_t)jp * pair_stride + 2 * off)
                                    : make_uint4(0u, 0u, 0u, 0u);
            } else {
                load_taps_quad<NT>(wg, off, HW, ok, wreg[i]);
            }
        } else if constexpr (WSRC == 2) {
            // Transposed stencil (backward recurrence): tap j = w_{NT-1-j}[p + off_j], read straight from the
            // forward tap volume — one aligned quad of plane NT-1-j at row y+dy, shifted by dx columns with the
            // neighbouring lanes' quads (DPP), strip-end lanes patch with scalar loads.  No transposed copy of
            // the weights ever exists.
            float lq[NT][R], rq[NT][R];      // lq[j][c] = column xq-R+c, rq[j][c] = column xq+4+c of the source row
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int lin = j < NT / 2 ? j : j + 1;
                const int dy = lin / K - R;
                const int row = y + dy;
                const bool rok = ok && row >= 0 && row < H;
                const float4 v = rok ? ld4(wg + Taps<WT>::idx(NT - 1 - j, (size_t)(rok ? row : 0) * W + xq, HW)) : z4;
                wreg[i][j][0] = v.x; wreg[i][j][1] = v.y; wreg[i][j][2] = v.z; wreg[i][j][3] = v.w;
#pragma unroll
                for (int c = 0; c < R; ++c) {
                    lq[j][c] = dpp_from_prev_lane(wreg[i][j][4 - R + c]);
                    rq[j][c] = dpp_from_next_lane(wreg[i][j][c]);
                }
            }
            if (fix_left) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int lin = j < NT / 2 ? j : j + 1;
                    const int dy = lin / K - R, dx = lin % K - R;
                    const int row = y + dy;
                    if (dx < 0) {
#pragma unroll
                        for (int c = 0; c < R; ++c) {
                            const int xx = xq - R + c;
                            const bool cnd = ok && row >= 0 && row < H && xx >= 0;
                            lq[j][c] = cnd ? ld1(wg + Taps<WT>::idx(NT - 1 - j, (size_t)row * W + xx, HW)) : 0.f;
                        }
                    }
                }
            }
            if (fix_right) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int lin = j < NT / 2 ? j : j + 1;
                    const int dy = lin / K - R, dx = lin % K - R;
                    const int row = y + dy;
                    if (dx > 0) {
#pragma unroll
                        for (int c = 0; c < R; ++c) {
                            const int xx = xq + 4 + c;
                            const bool cnd = ok && row >= 0 && row < H && xx < W;
                            rq[j][c] = cnd ? ld1(wg + Taps<WT>::idx(NT - 1 - j, (size_t)row * W + xx, HW)) : 0.f;
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int lin = j < NT / 2 ? j : j + 1;
                const int dx = lin % K - R;
                // window of the source row: [xq-R, xq+4+R) = lq | own quad | rq; tap value e = window[R + e + dx]
                float win[4 + 2 * R];
#pragma unroll
                for (int c = 0; c < R; ++c) { win[c] = lq[j][c]; win[R + 4 + c] = rq[j][c]; }
#pragma unroll
                for (int e = 0; e < 4; ++e) win[R + e] = wreg[i][j][e];
#pragma unroll
                for (int e = 0; e < 4; ++e) wreg[i][j][e] = ok ? win[R + e + dx] : 0.f;
            }
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
            float Sq[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float S = wreg[i][7][e];
#pragma unroll
                for (int k = 1; k < 8; ++k) S += wreg[i][7 - k][e];
                Sq[e] = S;
                float av[8], qv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) av[j] = wreg[i][j][e];
                div8_shared_reciprocal(av, S, qv);
#pragma unroll
                for (int j = 0; j < NT; ++j) wreg[i][j][e] = (ok && e < nval) ? qv[j] : 0.f;
            }
            // the launch that derives the weights can also publish them (tap-volume layout) for the launches
            // that follow, which then stream them like a prepared volume
            if (a.w_out && ((interior >> i) & 1u)) {
                WT* wo = static_cast<WT*>(a.w_out) + (size_t)b * Taps<WT>::image_elems(NT, HW);
                store_taps_quad<NT>(wo, off, HW, wreg[i]);
            }
            if (a.s_out && ((interior >> i) & 1u)) st4(a.s_out + (size_t)b * HW + off, make_float4(Sq[0], Sq[1], Sq[2], Sq[3]));
        }
        if (BLEND && r < wr) {
            const float4 m = ok ? sgn4(ld4(spg + off)) : z4;
            const int qoff = (r * wq + sx) * 4;
            if constexpr (FOLD) {
                // (1-m) u + m d0 with u = sum_j w_j d_j  ==  sum_j ((1-m) w_j) d_j + m d0: 1-m is 0, 1 or 2, so the scaled
                // weights and every product are exact and the result is bit-identical — the steps then only add m d0
                const float omq[4] = {1.f - m.x, 1.f - m.y, 1.f - m.z, 1.f - m.w};
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) wreg[i][j][e] *= omq[e];
            } else {
                *reinterpret_cast<float4*>(om_lds + qoff) = make_float4(1.f - m.x, 1.f - m.y, 1.f - m.z, 1.f - m.w);
            }
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

    // SCORE: the final quads are kept and scored AFTER the step loop, when the weight and window registers are dead —
    // ten accumulators live across the loop cost the 64-VGPR instance spills and several us.
    float4 scored_q[SCORE ? NQ : 1];
    float4 scored_t[SCORE ? NQ : 1];     // the target quads, requested at the top of the last step: their latency hides behind it
    // The centre rows of a thread's window are its own previous outputs: they stay in registers from step to step
    // instead of being read back from LDS (NQ of the NQ + 2R window reads per step: -3 % per forward at config 2;
    // the step loop is bound by LDS traffic and barrier latency rather than by VALU issue).
    float own[NQ][4];
    if (active) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            int drow = r0 + i + R;
            if (NQ > 1) drow = drow < dr ? drow : dr - 1;
            const v4f mid = *(lds_cv4f_ptr)(cur + drow * ls + cb);
            own[i][0] = mid.x; own[i][1] = mid.y; own[i][2] = mid.z; own[i][3] = mid.w;
        }
    }
    for (int s = 1; s <= a.S; ++s) {
        const bool last = (s == a.S);
        if (SCORE && last && active) {
#pragma unroll
            for (int i = 0; i < NQ; ++i)
                if ((interior >> i) & 1u)
                    scored_t[SCORE ? i : 0] = ld4(static_cast<const DT*>(a.target) + (size_t)b * HW + (size_t)(yq0 + i) * W + xq);
        }
        if (active) {
            // Window fetch.  One aligned ds_read_b128 per neighbour row (the thread's own rows are in `own`); the R pixels
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
                float m4[4];
                if (rr >= R && rr < R + NQ) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) m4[c] = own[rr - R][c];
                } else {
                    const v4f mid = *(lds_cv4f_ptr)(row_ptr(rr));
                    m4[0] = mid.x; m4[1] = mid.y; m4[2] = mid.z; m4[3] = mid.w;
                }
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
                                if constexpr (PACKED)
                                    u[e] = fma_packed_tap(wpk[i][j >> 1], j & 1, e, win[i + dy + R][e + dx + R], u[e]);
                                else
                                    u[e] = fmaf(wreg[i][j][e], win[i + dy + R][e + dx + R], u[e]);
                        }
                    float keep[4];   // value carried to the next step through LDS
                    float om[4] = {1.f, 1.f, 1.f, 1.f}, md[4] = {0.f, 0.f, 0.f, 0.f};
                    if (BLEND) {
                        const int qoff = ((r0 + i) * wq + sx) * 4;
                        if constexpr (!FOLD) {
                            const float4 o4 = *reinterpret_cast<const float4*>(om_lds + qoff);
                            om[0] = o4.x; om[1] = o4.y; om[2] = o4.z; om[3] = o4.w;
                        }
                        if (BLEND == CSPN_BLEND_SPARSE) {
                            const float4 m4 = *reinterpret_cast<const float4*>(md_lds + qoff);
                            md[0] = m4.x; md[1] = m4.y; md[2] = m4.z; md[3] = m4.w;
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (BLEND == CSPN_BLEND_SPARSE) {
                            // (1-m) u + m d0     CSPN_new.py:90 ((1-m) lives in the weights when FOLD)
                            u[e] = FOLD ? u[e] + md[e] : om[e] * u[e] + md[e];
                            keep[e] = u[e];
                        } else if (BLEND == CSPN_BLEND_PREMASK) {
                            keep[e] = om[e] * u[e];
                        } else {
                            keep[e] = u[e];
                        }
                        // zero padding (outside the image, incl. row-padding columns) stays exactly zero
                        if (!((in_img >> i) & 1u) || e >= nval) { u[e] = 0.f; keep[e] = 0.f; }
                    }
                    if (!last)
                        *reinterpret_cast<float4*>(&nxt[(r0 + i + R) * ls + cb]) =
                            make_float4(keep[0], keep[1], keep[2], keep[3]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) own[i][e] = keep[e];
                    if ((interior >> i) & 1u) {
                        const size_t off = (size_t)(yq0 + i) * W + xq;
                        const float4 uv = make_float4(u[0], u[1], u[2], u[3]);
                        if (hist) st4(hist + (size_t)(s - 1) * plane + off, uv);
                        else if (last) st4(dout + off, uv);
                        if (SCORE && last) scored_q[SCORE ? i : 0] = uv;
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
        float mf[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) mf[k] = 0.f;
        if (active) {
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                if ((interior >> i) & 1u) {
                    const float4 tg = scored_t[i];
                    const float t4[4] = {tg.x, tg.y, tg.z, tg.w};
                    const float o4[4] = {scored_q[i].x, scored_q[i].y, scored_q[i].z, scored_q[i].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float o = o4[e];
                        if (sizeof(DT) == 2) o = __half2float(__float2half_rn(o));   // score the stored value
                        metric_terms(o, t4[e], mf);
                    }
                }
            }
        }
        // fp32 wave reduction (<= 256 pixels per wave), fp64 across the waves, 10 atomics per workgroup
        float* part = lds + (size_t)2 * a.dr * a.ls + (size_t)(BLEND == CSPN_BLEND_SPARSE ? 2 : (BLEND ? 1 : 0)) * a.wr * 4 * a.wq;
        const int wave = tid >> 6;
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const float v = wave_sum_to_lane63(mf[k]);
            if (lane == 63) part[wave * 10 + k] = v;
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
// host side: plan selection and launches
// ------------------------------------------------------------------------------------------------
struct Launch {
    PropArgs a;
    int grid, threads, nq;
    size_t lds_bytes;
};

// Geometry of one fused launch.  Returns false if (plan, S) does not fit the machine limits.
bool make_geometry(int K, int B, int H, int W, int S, int tw, int th, int nq, int threads, int blend, Launch* L,
                   int Wv = 0) {
    const int R = K / 2;
    if (tw <= 0 || th <= 0 || (tw & 3) || nq <= 0) return false;
    PropArgs& a = L->a;
    a.B = B; a.H = H; a.W = W; a.S = S;
    a.Wv = (Wv > 0 && Wv < W) ? Wv : W;
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
//     weight-region cache lines (tiles x region rows x lines per row) wins, wider tile on ties.
// Fields the caller pins (steps_per_launch, quads_per_thread, threads > 0 in `user`) are taken as given and the tile
// search runs for THEM, so a partial plan such as {S = 4, NQ = 3, 256 threads} still gets the cheapest tiling.
void default_plan(int K, int B, int H, int W, int T, int keep_history, const cspn_plan* user, cspn_plan* p) {
    (void)keep_history;
    const int R = K / 2;
    p->force_scalar = 0;
    p->quads_per_thread = (user && user->quads_per_thread > 0) ? user->quads_per_thread : 1;
    p->tile_w = 0;
    p->tile_h = 0;
    int S0 = (K == 3) ? 8 : (K == 5 ? 3 : 2);
    const bool s_pinned = user && user->steps_per_launch > 0;
    if (s_pinned) S0 = user->steps_per_launch;
    if (T < 1) T = 1;
    if (S0 > T) S0 = T;
    if (!s_pinned) S0 = ceil_div(T, ceil_div(T, S0));   // balance the launches (T=24, S0=8 -> 3 x 8)
    // Largest workgroup first (least halo); small batches step down to 512 threads when the launch would leave
    // most CUs without a tile (B=3 at 304x228 with 1024-thread tiles occupies 90 of the 256 CUs).  Going further
    // down (256 threads) costs more in halo work than it gains in occupancy (plan sweeps at B=1..3).
    int thread_opts[3] = {(K == 3) ? 1024 : 256, (K == 3) ? 512 : 0, 0};
    if (user && user->threads > 0) { thread_opts[0] = user->threads; thread_opts[1] = 0; }
    for (int ti = 0; ti < 3; ++ti) {
        const int threads = thread_opts[ti];
        if (!threads) break;
        int S = S0, tw_best = 0, th_best = 0;
        long tiles_best = 0;
        for (;; --S) {                               // shrink S until some tiling fits the workgroup
            const int hyw = (S - 1) * R, hxw = round_up4(hyw);
            long best_cost = -1;
            for (int n = 1; n <= 64; ++n) {
                const int tw = round_up4(ceil_div(W, n));
                if (n > 1 && tw < 16) break;
                const int wq = (tw + 2 * hxw) / 4;
                if (wq > threads) continue;
                int th = p->quads_per_thread * (threads / wq) - 2 * hyw;
                if (th > H) th = H;
                if (th < 1 || (th < 8 && th < H)) continue;
                th = ceil_div(H, ceil_div(H, th));   // even out the tile rows
                const long tiles = (long)ceil_div(W, tw) * ceil_div(H, th);
                // region rows x 128-byte lines per row (a 16-byte-aligned row segment of wq quads touches wq/8 + 7/8
                // lines on average): plain pixel counts favoured narrow tiles whose rows waste most of their last line
                // (NYU: 52x46 modelled 5 % cheaper than 76x30 but measured 7 % slower with history, 2 % without).  The
                // 24/48-tap kernels are VALU-bound — their cost follows the pixel count — so the line term is small there
                // (K = 5, fp16, S = 4 / NQ = 3: the full term picks 76x21, 179 k maps/s, against 190-194 k for 44x38 / 52x33).
                const long cost = tiles * (long)(wq + (K == 3 ? 7 : 2)) * (th + 2 * hyw);
                if (best_cost < 0 || cost < best_cost) { best_cost = cost; tw_best = tw; th_best = th; tiles_best = tiles; }
            }
            if (best_cost >= 0 || S == 1) break;
        }
        if (tw_best > 0) {
            p->threads = threads; p->steps_per_launch = S; p->tile_w = tw_best; p->tile_h = th_best;
            if ((long)B * tiles_best >= 160) break;  // enough tiles to keep most of the 256 CUs busy
        }
    }
    if (p->tile_w <= 0) {
        p->threads = 256;
        p->steps_per_launch = 1;
        p->tile_w = round_up4(W < 64 ? W : 64);
        p->tile_h = p->threads / (p->tile_w / 4);
    }
}

void resolve_plan(int K, int B, int H, int W, int T, int keep_history, const cspn_plan* user, cspn_plan* p) {
    default_plan(K, B, H, W, T, keep_history, user, p);
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

// (quads_per_thread, threads) pairs that have a compiled cspn_prop_fused instance (keep in step with launch_fused)
bool has_instance(int K, int nq, int threads) {
    if (K == 3) return ((nq == 1 || nq == 2) && (threads == 256 || threads == 512 || threads == 1024)) ||
                       (nq == 4 && (threads == 256 || threads == 512)) || (nq == 8 && threads == 256);
    if (K == 5) return (threads == 256 && nq >= 1 && nq <= 3) || (threads == 512 && nq == 1);
    return K == 7 && nq == 1 && threads == 256;
}

// Launches with more than 64 KiB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised once per
// (kernel instance, device); the granted size is remembered so the hot loop does not repeat the driver call.
template <auto Kern>
int ensure_dynamic_lds(size_t bytes) {
    static std::atomic<size_t> granted[64];
    int dev = 0;
    HIP_OK(hipGetDevice(&dev));
    std::atomic<size_t>& slot = granted[dev & 63];
    if (slot.load(std::memory_order_acquire) >= bytes) return 1;
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(Kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    size_t seen = slot.load(std::memory_order_relaxed);
    while (seen < bytes && !slot.compare_exchange_weak(seen, bytes, std::memory_order_release)) {}
    return 1;
}

template <int K, int NQ, int NTHREADS, typename WT, typename DT, int WSRC, int SCORE = 0>
int launch_fused_blend(const Launch& L, int blend, hipStream_t st) {
#define CSPN_LAUNCH(BL)                                                                               \
    do {                                                                                              \
        constexpr auto kern = cspn_prop_fused<K, NQ, NTHREADS, WT, DT, BL, WSRC, SCORE>;              \
        if (L.lds_bytes > 64 * 1024 && !ensure_dynamic_lds<kern>(L.lds_bytes)) return 0;              \
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
        if (wsrc == 1) { // on-the-fly weights: one- and two-quad instances only
#define CSPN_CASE_G(NQV, NTV) \
    if (L.nq == NQV && L.threads == NTV) return launch_fused_blend<K, NQV, NTV, WT, DT, 1>(L, blend, st)
            CSPN_CASE_G(1, 256); CSPN_CASE_G(2, 256); CSPN_CASE_G(1, 512); CSPN_CASE_G(2, 512);
            CSPN_CASE_G(1, 1024); CSPN_CASE_G(2, 1024);
#undef CSPN_CASE_G
            return fail("no from-guidance kernel instance for quads_per_thread=%d threads=%d", L.nq, L.threads);
        }
    } else if (wsrc == 1) {
        return fail("on-the-fly weights exist for K=3 only");
    }
    if (wsrc == 2) {     // transposed recurrence: one-quad (and, K=3, two-quad) instances, f32 state
        if constexpr (std::is_same<DT, float>::value) {
#define CSPN_CASE_T(NQV, NTV) \
    if (L.nq == NQV && L.threads == NTV) return launch_fused_blend<K, NQV, NTV, WT, DT, 2>(L, blend, st)
            CSPN_CASE_T(1, 256);
            if constexpr (K <= 5) { CSPN_CASE_T(1, 512); }
            if constexpr (K == 3) { CSPN_CASE_T(2, 256); CSPN_CASE_T(2, 512); CSPN_CASE_T(1, 1024); CSPN_CASE_T(2, 1024); }
#undef CSPN_CASE_T
        }
        return fail("no transposed kernel instance for K=%d quads_per_thread=%d threads=%d", K, L.nq, L.threads);
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
                    int nslots = 0, void* w_out = nullptr, int Wv = 0, float* s_out = nullptr) {
    if (Wv < 0 || Wv > W) return fail("W_valid=%d outside (0, W=%d]", Wv, W);
    if (Wv > 0 && Wv < W && (W % 4 != 0)) return fail("row padding (W_valid < W) needs a pitch W %% 4 == 0");
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
            if (!make_geometry(K, B, H, W, S, p.tile_w, p.tile_h, p.quads_per_thread, p.threads, blend, &L, Wv))
                return fail("plan does not fit: K=%d S=%d tile=%dx%d nq=%d threads=%d", K, S, p.tile_w, p.tile_h,
                            p.quads_per_thread, p.threads);
            // from-guidance with a weight buffer: the first launch derives + publishes the weights, the rest stream them
            const bool derive = (wsrc == 1) && (launch_idx == 0 || !w_out);
            L.a.w = (wsrc == 1 && !derive) ? w_out : w;
            L.a.w_out = (derive && (n_launch > 1 || s_out)) ? w_out : nullptr;     // s_out: the backward wants the volume too
            L.a.s_out = (derive && launch_idx == 0) ? s_out : nullptr;
            L.a.g_bs = g_bs; L.a.g_cs = g_cs; L.a.d_in = src; L.a.sparse = sparse; L.a.d0 = d0;
            L.a.d_out = history ? nullptr : dst;
            L.a.hist = hist_base;
            const bool final_launch = (t + S >= T);
            if (final_launch && macc && derive)
                return fail("scored from-guidance propagation needs more than one launch (T > steps_per_launch)");
            L.a.target = final_launch ? target : nullptr;
            L.a.macc = final_launch ? macc : nullptr;
            L.a.nslots = nslots;
            if (!launch_fused<K, WT, DT>(L, blend, wsrc == 2 ? 2 : (derive ? 1 : 0), st)) return 0;
        } else {
            if (Wv > 0 && Wv < W) return fail("row padding (W_valid < W) needs the vector path (16-byte aligned tensors)");
            if (wsrc == 2) return fail("transposed propagation needs W %% 4 == 0 and 16-byte aligned tensors; use "
                                       "cspn_transpose_weights + cspn_propagate");
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
                void* work, int d_dtype, int B, int H, int W, int Wv, int T, int blend, const cspn_plan* plan,
                hipStream_t st) {
    if (w_dtype == CSPN_F32 && d_dtype == CSPN_F32)
        return propagate_typed<K, float, float>(w, d0, sparse, out, history, work, B, H, W, T, blend, plan, st, 0, 0, 0,
                                                nullptr, nullptr, 0, nullptr, Wv);
    if (w_dtype == CSPN_F16 && d_dtype == CSPN_F16)
        return propagate_typed<K, __half, __half>(w, d0, sparse, out, history, work, B, H, W, T, blend, plan, st, 0, 0, 0,
                                                  nullptr, nullptr, 0, nullptr, Wv);
    if (w_dtype == CSPN_F16 && d_dtype == CSPN_F32)
        return propagate_typed<K, __half, float>(w, d0, sparse, out, history, work, B, H, W, T, blend, plan, st, 0, 0, 0,
                                                 nullptr, nullptr, 0, nullptr, Wv);
    return fail("unsupported dtype combination w=%d d=%d", w_dtype, d_dtype);
}

}  // namespace

extern "C" {

int cspn_plan_resolve(int K, int B, int H, int W, int T, int keep_history, const cspn_plan* plan_or_null,
                      cspn_plan* resolved) {
    if (!resolved) return fail("cspn_plan_resolve: NULL output");
    if (K != 3 && K != 5 && K != 7) return fail("cspn_plan_resolve: unsupported K=%d", K);
    resolve_plan(K, B, H, W, T, keep_history, plan_or_null, resolved);
    if (!resolved->force_scalar) {
        if (!has_instance(K, resolved->quads_per_thread, resolved->threads))
            return fail("no kernel instance for K=%d quads_per_thread=%d threads=%d", K, resolved->quads_per_thread,
                        resolved->threads);
        Launch L;
        if (!make_geometry(K, B, H, W, resolved->steps_per_launch, resolved->tile_w, resolved->tile_h,
                           resolved->quads_per_thread, resolved->threads, CSPN_BLEND_SPARSE /* worst-case LDS */, &L))
            return fail("plan does not fit: K=%d S=%d tile=%dx%d nq=%d threads=%d", K, resolved->steps_per_launch,
                        resolved->tile_w, resolved->tile_h, resolved->quads_per_thread, resolved->threads);
    }
    return 1;
}
size_t cspn_propagate_workspace_bytes(int B, int H, int W, int T, int d_dtype, int keep_history) {
    if (keep_history || T <= 1) return 0;
    return (size_t)2 * B * H * W * esize(d_dtype);
}

int cspn_propagate(const void* w, int w_dtype, const void* d0, const void* sparse, void* out, void* history,
                   void* work, int d_dtype, int B, int H, int W, int W_valid, int K, int T, int blend,
                   const cspn_plan* plan, cspn_stream_t stream) {
    if (!w || !d0 || B <= 0 || H <= 0 || W <= 0 || T < 0) return fail("cspn_propagate: bad arguments");
    if (blend != CSPN_BLEND_NONE && !sparse) return fail("cspn_propagate: blend=%d needs sparse", blend);
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (K) {
        case 3: return propagate_k<3>(w, w_dtype, d0, sparse, out, history, work, d_dtype, B, H, W, W_valid, T, blend, plan, st);
        case 5: return propagate_k<5>(w, w_dtype, d0, sparse, out, history, work, d_dtype, B, H, W, W_valid, T, blend, plan, st);
        case 7: return propagate_k<7>(w, w_dtype, d0, sparse, out, history, work, d_dtype, B, H, W, W_valid, T, blend, plan, st);
        default: return fail("cspn_propagate: unsupported K=%d (3, 5, 7)", K);
    }
}

int cspn_propagate_scored(const void* w, int w_dtype, const void* d0, const void* sparse, void* out, void* work,
                          int d_dtype, int B, int H, int W, int W_valid, int K, int T, int blend, const void* target,
                          double* acc, int nslots, const cspn_plan* plan, cspn_stream_t stream) {
    if (!w || !d0 || !out || !target || !acc || nslots < 1 || B <= 0 || H <= 0 || W <= 0 || T < 1)
        return fail("cspn_propagate_scored: bad arguments");
    if (blend != CSPN_BLEND_NONE && blend != CSPN_BLEND_SPARSE) return fail("cspn_propagate_scored: blend %d", blend);
    if (blend != CSPN_BLEND_NONE && !sparse) return fail("cspn_propagate_scored: blend needs sparse");
    if (!aligned16(target)) return fail("cspn_propagate_scored: target must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
#define SCORED(KV, WTT, DTT) \
    return propagate_typed<KV, WTT, DTT>(w, d0, sparse, out, nullptr, work, B, H, W, T, blend, plan, st, 0, 0, 0, target, acc, \
                                         nslots, nullptr, W_valid)
    if (K == 3 && w_dtype == CSPN_F32 && d_dtype == CSPN_F32) SCORED(3, float, float);
    if (K == 3 && w_dtype == CSPN_F16 && d_dtype == CSPN_F16) SCORED(3, __half, __half);
    if (K == 5 && w_dtype == CSPN_F32 && d_dtype == CSPN_F32) SCORED(5, float, float);
    if (K == 5 && w_dtype == CSPN_F16 && d_dtype == CSPN_F16) SCORED(5, __half, __half);
#undef SCORED
    return fail("cspn_propagate_scored: unsupported K=%d / dtypes w=%d d=%d", K, w_dtype, d_dtype);
}

int cspn_propagate_transposed(const void* w, int w_dtype, const float* g_T, const float* sparse_f32, float* history,
                              int B, int H, int W, int W_valid, int K, int T, int premask, const cspn_plan* plan,
                              cspn_stream_t stream) {
    if (!w || !g_T || !history || B <= 0 || H <= 0 || W <= 0 || T < 1) return fail("cspn_propagate_transposed: bad arguments");
    if (premask && !sparse_f32) return fail("cspn_propagate_transposed: premask needs sparse");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int blend = premask ? CSPN_BLEND_PREMASK : CSPN_BLEND_NONE;
#define TRANSPOSED(KV, WTT) \
    return propagate_typed<KV, WTT, float>(w, g_T, sparse_f32, nullptr, history, nullptr, B, H, W, T, blend, plan, st, 2, 0, 0, \
                                           nullptr, nullptr, 0, nullptr, W_valid)
    if (w_dtype == CSPN_F32) {
        if (K == 3) TRANSPOSED(3, float);
        if (K == 5) TRANSPOSED(5, float);
        if (K == 7) TRANSPOSED(7, float);
    } else if (w_dtype == CSPN_F16) {
        if (K == 3) TRANSPOSED(3, __half);
        if (K == 5) TRANSPOSED(5, __half);
        if (K == 7) TRANSPOSED(7, __half);
    }
#undef TRANSPOSED
    return fail("cspn_propagate_transposed: unsupported K=%d / w_dtype=%d", K, w_dtype);
}

int cspn3_propagate_from_guidance(const void* guidance, int g_dtype, long bs, long cs, void* w8_out, float* s_out,
                                  const void* d0, const void* sparse, void* out, void* history, void* work, int d_dtype, int B, int H,
                                  int W, int W_valid, int T, int blend, const void* target, double* acc, int nslots,
                                  const cspn_plan* plan, cspn_stream_t stream) {
    if ((target || acc) && (!target || !acc || nslots < 1 || !w8_out || history || !aligned16(target) || g_dtype != d_dtype))
        return fail("cspn3_propagate_from_guidance: scoring needs target, acc, nslots >= 1, w8_out, no history, "
                    "one dtype and a 16-byte aligned target");
    if (!guidance || !d0 || B <= 0 || H <= 0 || W <= 0 || T < 0) return fail("cspn3_propagate_from_guidance: bad arguments");
    if (blend != CSPN_BLEND_NONE && blend != CSPN_BLEND_SPARSE) return fail("cspn3_propagate_from_guidance: blend %d", blend);
    if (blend != CSPN_BLEND_NONE && !sparse) return fail("cspn3_propagate_from_guidance: blend needs sparse");
    if ((cs & 3) || (bs & 3)) return fail("cspn3_propagate_from_guidance: guidance strides must be multiples of 4 elements");
    if (w8_out && !aligned16(w8_out)) return fail("cspn3_propagate_from_guidance: w8_out must be 16-byte aligned");
    if (s_out && (!aligned16(s_out) || !w8_out)) return fail("cspn3_propagate_from_guidance: s_out needs w8_out and 16-byte alignment");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (g_dtype == CSPN_F32 && d_dtype == CSPN_F32)
        return propagate_typed<3, float, float>(guidance, d0, sparse, out, history, work, B, H, W, T, blend, plan, st, 1, bs, cs, target, acc, nslots, w8_out, W_valid, s_out);
    if (g_dtype == CSPN_F16 && d_dtype == CSPN_F16)
        return propagate_typed<3, __half, __half>(guidance, d0, sparse, out, history, work, B, H, W, T, blend, plan, st, 1, bs, cs, target, acc, nslots, w8_out, W_valid, s_out);
    if (g_dtype == CSPN_F16 && d_dtype == CSPN_F32)
        return propagate_typed<3, __half, float>(guidance, d0, sparse, out, history, work, B, H, W, T, blend, plan, st, 1, bs, cs, target, acc, nslots, w8_out, W_valid, s_out);
    return fail("cspn3_propagate_from_guidance: unsupported dtypes g=%d d=%d", g_dtype, d_dtype);
}

}  // extern "C"
