// cspnk_d2.hip — BASELINE config 3's kernel: the 5 x 5 softmax propagation with fp16 guidance AND fp16 state as ONE
// weight-resident launch whose steps run on v_dot2_f32_f16.
//
// Reference path: network/libs/post_process/CSPN_ours.py:24-54 (softmax over the 24 guidance channels :35, zero centre tap
// :37-39, prop_time x { pac.conv2d :49 = network/libs/base/pac.py:89-92, sparse blend :51-53 }) run in half precision, where the
// state x is a half tensor that is rounded after every step.
//
// cspnk_resident.hip keeps the state as fp32 in LDS and spends one v_fma_mix_f32 per tap and pixel: 192 per oct-step at 4.6 cycles
// per wavefront (tools/probes/valu_rate_probe.hip) — the step phases were 2 x 17 of the 86 us of a forward.  Here
//   * the state lives in LDS as PACKED fp16 pairs (what the reference's half tensor holds between steps): a window row of an oct is
//     one ds_read_b128 + two ds_read_b32, the five odd-aligned pairs come from v_alignbit_b32;
//   * the softmax taps are packed as TAP pairs of one pixel — (dx-2, dx-1), (dx, dx+1) of a window row — so that two taps are one
//     v_dot2_f32_f16 (same issue cost as one v_fma_mix_f32, measured): 10 dot products + 4 single taps per pixel-step instead of
//     24 FMAs, fp32 accumulation, ONE rounding to fp16 per step (v_cvt_pk_f16_f32);
//   * the batch goes through ONE launch: the register files hold the taps of 12 of config 3's 24 frames, so a workgroup refines
//     its tile of frame b and then of frame b + 12 back to back (a.rounds) — no second launch, no drain between the halves;
//   * the softmax is cheaper per channel (max folded into the exponent's scale: one v_fma_mix_f32 + v_exp_f32; numerator x 1/sum
//     rounded by v_mul_f32 + v_cvt_pk_f16_f32 for two taps at a time).
// Results are those of the half-precision recurrence with fp32 accumulation inside a step; they differ from the phase-rounded
// schedules of cspnk_resident.hip / the multi-launch plans by fp16 rounding of the state (tests: oracle tolerance of the fp16
// configuration, tests/test_hip_kres.py).  Exchange protocol, flags, bounded wait and NaN poisoning: those of cspnk_resident.hip.
#include "cspnk_helpers.hpp"

#include <atomic>

namespace {

typedef unsigned v4uu __attribute__((ext_vector_type(4)));
typedef const volatile __attribute__((address_space(3))) v4uu* lds_cv4u_ptr;
typedef const volatile __attribute__((address_space(3))) unsigned* lds_cu_ptr;

// (the arithmetic of a step and of the softmax — dot2, mix_hh, cvt_pk_f16, half_scaled, the tap-pair slots, softmax_to_pairs — lives in
//  cspnk_helpers.hpp: the guard's re-computation of this kernel, csrc/cspn_repair.hip, uses the same code and so produces the same bits)

constexpr int D2_R = 2, D2_NT = 24;
constexpr int D2_NPF = 8;           // guidance channels of a round that are staged through LDS (a third of the 24)
#ifndef CSPN_D2_PREFETCH
#define CSPN_D2_PREFETCH 0          // round 6, BUILT, measured and NOT the default (NEGATIVE_RESULTS #57): see cspnk_d2's NPF
#endif
#ifndef CSPN_D2_DPP_HALO
#define CSPN_D2_DPP_HALO 0          // round 6, measured and NOT the default (1 for A/B): see the step
#endif

// MODE 0: inference; 1: inference + fused depth metrics; 2: the training forward — every step's state goes to its fp16 history
// plane and the softmax taps are published once as the fp16 tap volume (pairs (2i, 2i+1) interleaved per quad: cspn_common.hpp
// Taps<__half>) that cspn_transpose_kernel / cspn_grad_tail<5, __half, __half> stream in the backward.
// NPF (round 6; 0 or D2_NPF): the first NPF guidance channels of a round reach the registers THROUGH LDS (global_load_lds_dwordx4: no
// destination registers) — and those of round r + 1 are requested as soon as round r's taps are derived, so that they stream while round
// r's steps run (VALU-bound, the memory system idle) instead of in front of round r + 1's softmax.  Round 4 priced this (NEGATIVE_RESULTS
// #35: "40 % of the region fits the free LDS ... a second staging protocol"); round 4's register form of the idea spilled (#38).
// BUILT in round 6 (-DCSPN_D2_PREFETCH=1: a third of the guidance, 96 KB per workgroup; parity suite green, same bits) and SLOWER: 69.9 ->
// 72.6 us per scored forward (profiles/r06_d2_prefetch_ab.txt).  The in-kernel stamps of the second round say why (r06_d2_prefetch_stamps.txt):
// parked + derive + the barrier behind them 11.1 us without, 11.6 us with the prefetch — what a round spends in front of its steps is NOT
// the guidance stream but the softmax itself, 192 v_exp_f32 per thread at a quarter of the VALU rate with three wavefronts per SIMD taking
// turns (the loads hide under the other wavefronts' exponentials); the staging through LDS only adds instructions and 5 spilled registers.
template <int BLEND, int MODE, int CLEAN, int NTH, int NPF>
__global__ __launch_bounds__(NTH, NTH / 256) void cspnk_d2(const KResArgs a) {
    constexpr int R = D2_R, NT = D2_NT;
    constexpr bool SCORE = MODE == 1, HIST = MODE == 2;
    using IO = StateIO<__half>;
    using Oct = IO::Oct;
    using Pair = IO::Pair;
    extern __shared__ __attribute__((aligned(16))) unsigned ldsu[];
    __shared__ int wg_bad;

    const int tid = threadIdx.x;
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int tile = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int bl = tile / tiles_per_img;
    const int trem = tile - bl * tiles_per_img;
    const int ty = trem / a.tiles_x;
    const int tx = trem - ty * a.tiles_x;
    const int H = a.H, W = a.W;
    const int y0 = ty * a.th, x0 = tx * a.tw;
    const unsigned HW = (unsigned)(H * W);
    const size_t plane = (size_t)a.B * HW;
    int n_stamp = 0;
    // Training form: every workgroup counts itself out when it has finished all its rounds (or given up); the last one re-arms the
    // counter and stores `seq` to the second host word — the completion report the end-of-backward check polls (cspn_resident.hip).
    auto count_out = [&]() {
        if (HIST && tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(a.status + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1u == gridDim.x) {
                __hip_atomic_store(a.status + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.host_err) __hip_atomic_store(a.host_err + 1, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    };

    // ---- ownership: thread (sy, sx) owns oct sx of region row sy (tile + halo), one oct per thread ---------------------
    const int wo = a.wo, wr = a.wr;
    const int sp = tid / wo;                                // strip row in thread order ...
    const int sx = tid - sp * wo;
    const bool exch = a.T > a.S;
    const int rx0 = max(exch ? max(0, x0 - a.tw) : 0, min(x0 - a.hxw, W - 8 * wo));
    const int ry0 = max(exch ? max(0, y0 - a.th) : 0, min(y0 - a.hyw, H - wr));
    // ... and the region row it owns.  -DCSPN_D2_HALO_ROWS_FIRST (an experiment that did not pay): the HALO rows (above
    // and below the tile) come first in thread order, the tile's own rows after them; the last step of a phase is only needed on
    // the tile's rows (the halo is re-staged from the neighbours, or the forward is over), so the wavefronts that own nothing but
    // halo rows sit it out — at config 3 the 12 halo rows are the first four wavefronts, one per SIMD.  The step phases shrink
    // (4.22 -> 3.74 us) but the neighbour waits grow by as much: 71.3-72.3 vs 72.0-72.5 us per forward, B = 3 29.4 vs 28.5 us.
    const int i0 = y0 - ry0;                                // first tile row inside the region
    const int th_in = min(a.th, ry0 + wr - y0);             // tile rows inside the region (the last tile may be cut short)
#ifdef CSPN_D2_HALO_ROWS_FIRST
    const int nh = wr - th_in;
    const int sy = sp >= wr ? sp : (sp < nh ? (sp < i0 ? sp : sp + th_in) : i0 + (sp - nh));
#else
    const int sy = sp;
#endif
    const int xo = rx0 + 8 * sx;
    const int yo = ry0 + sy;
    const bool active = sy < wr;
    const bool in_img = active && xo < W && yo < H;        // W % 8 == 0: an oct is inside the image or outside as a whole
    const bool interior = in_img && yo >= y0 && yo < y0 + a.th && xo >= x0 && xo < x0 + a.tw;
    const unsigned off_own = in_img ? (unsigned)(yo * W + xo) : 0u;
#if CSPN_D2_DPP_HALO
    const bool fix_left = (sx == 0) || ((tid & 63) == 0), fix_right = (sx == wo - 1) || ((tid & 63) == 63);
    const int colL = fix_left ? 4 * (sx + 1) - 1 : 0, colR = fix_right ? 4 * (sx + 1) + 4 : 0;      // dword 0 of a row: padding, one address per row
#endif

    // LDS: two depth buffers of dr rows x ls dwords; a row is [3 pad][ring pair][wo octs x 4 dwords][ring pair][pad]: oct k at
    // dword 4 (k + 1) (16-byte aligned), the pixel pair left of the row at dword 3, the pair right of it at dword 4 (wo + 1)
    const int dr = a.dr, ls = a.ls;
    const int pp = dr * ls;
    const int yd0 = ry0 - R;
    const int step_r = NTH / wo, step_q = NTH - step_r * wo;
    const int n_phase = (a.T + a.S - 1) / a.S;
    // the prefetch area: behind the two depth buffers and the scoring's partial sums, [NPF channels][NTH lanes] x 16 bytes — lane l of
    // wavefront w finds channel c of its own oct at slot c * NTH + 64 w + l (the DMA writes M0 + 16 x lane: a wavefront reads what it wrote)
    unsigned* const pf_lds = ldsu + ((2 * pp + (NTH / 64) * 10 + 16 + 3) & ~3);
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto prefetch = [&](int bimg, unsigned off) __attribute__((always_inline)) {
        const __half* __restrict__ gbn = kuniform_ptr(static_cast<const __half*>(a.g) + (size_t)bimg * NT * HW);
#pragma unroll
        for (int c = 0; c < NPF; ++c)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const GLB void*>(atb(gbn, ((unsigned)c * HW + off) * 2u)),
                                             (__attribute__((address_space(3))) void*)(pf_lds + (c * NTH + wave_u * 64) * 4), 16, 0, 0);
    };

    for (int round = 0; round < a.rounds; ++round) {
        const int b = a.b0 + round * a.nb + bl;
        if (b >= a.B) break;                                  // the last round of a ragged batch has fewer images
        auto stamp = [&]() {
            if (a.dbg && tid == 0 && n_stamp < 16 && round == a.rounds - 1) a.dbg[(size_t)blockIdx.x * 16 + n_stamp++] = wall_clock64();
        };
        if (round > 0) __syncthreads();                       // the previous round's readers of LDS are done
        if (tid == 0) wg_bad = 0;
        stamp();
        const __half* __restrict__ x0b = kuniform_ptr(static_cast<const __half*>(a.x0) + (size_t)b * HW);
        const __half* __restrict__ spb = BLEND ? kuniform_ptr(static_cast<const __half*>(a.sparse) + (size_t)b * HW) : nullptr;

        // ---- 0. the depth region of phase 0 is requested FIRST (loads return in order) ---------------------------------
        Oct st0[2];                                           // dr * wo <= 2 * NTH octs (the host checks)
        unsigned st0_in = 0;
        {
            int row = sp, oc = sx;                            // (the staging enumerates the region linearly by thread id)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int y = yd0 + row, x = rx0 + 8 * oc;
                const bool in = (row < dr) && y >= 0 && y < H && x < W;
                if (in) st0_in |= 1u << u;
                st0[u] = IO::ld_oct(x0b, in ? (unsigned)(y * W + x) : 0u);
                row += step_r; oc += step_q;
                if (oc >= wo) { oc -= wo; ++row; }
            }
        }
        Pair rg[2];
        unsigned rg_in = 0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int t = tid + k * NTH;
            const int row = t >> 1, side = t & 1;
            const int y = yd0 + row, x = side ? rx0 + 8 * wo : rx0 - 2;
            const bool in = (t < 2 * dr) && y >= 0 && y < H && x >= 0 && x < W;
            if (in) rg_in |= 1u << k;
            rg[k] = IO::ld_pair(x0b, in ? (unsigned)(y * W + x) : 0u);
        }

        // ---- 1. guidance of the owned oct: 24 16-byte loads, all requested before the arithmetic ----------------------
        uint4 graw[NT];
        const __half* __restrict__ gb = kuniform_ptr(static_cast<const __half*>(a.g) + (size_t)b * NT * HW);
        // (an opaque per-round copy of the offset: left loop-invariant, the 24 load addresses are hoisted out of the round loop as
        // 64-bit pairs, spilled, and every scratch reload between the loads waits for the whole in-order stream)
        unsigned off_r = off_own;
        asm volatile("" : "+v"(off_r));
        if (NPF > 0 && round == 0) prefetch(b, off_r);        // (later rounds: requested during the previous round's steps)
#pragma unroll
        for (int c = NPF; c < NT; ++c) graw[c] = ld16(atb(gb, ((unsigned)c * HW + off_r) * 2u));
        // sparse blend operands of the owned oct, behind the guidance
        Oct spo, x0o;
        if (BLEND) { spo = IO::ld_oct(spb, off_r); x0o = IO::ld_oct(x0b, off_r); }

        // ---- park the depth region (its loads came first: only those are waited for here) ------------------------------
        unsigned* const cur = ldsu;
        unsigned* const nxt = ldsu + pp;
        {
            int tidk = tid, prow = sp, poc = sx;
            asm volatile("" : "+v"(tidk), "+v"(prow), "+v"(poc));
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bool in = (st0_in >> u) & 1u;
                if (prow < dr) {
                    const uint4 v = st0[u].a;
                    *reinterpret_cast<uint4*>(cur + prow * ls + 4 * (poc + 1)) = make_uint4(in ? v.x : 0u, in ? v.y : 0u, in ? v.z : 0u, in ? v.w : 0u);
                }
                prow += step_r; poc += step_q;
                if (poc >= wo) { poc -= wo; ++prow; }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int t = tidk + k * NTH;
                if (t < 2 * dr) {
                    const int row = t >> 1, side = t & 1;
                    const bool in = (rg_in >> k) & 1u;
                    const int at = row * ls + (side ? 4 * (wo + 1) : 3);
                    cur[at] = in ? rg[k].a : 0u;
                    nxt[at] = 0u;                              // the ring of the second buffer is never computed: it must read as 0
                }
            }
            // ... and so must its ring ROWS (with shifted regions they are the zero padding above / below the image)
            for (int c = tidk; c < 2 * R * ls; c += NTH) {
                const int rr = c / ls, col = c - rr * ls;
                nxt[(rr < R ? rr : dr - 2 * R + rr) * ls + col] = 0u;
            }
        }
        stamp();                                              // depth region parked

        // ---- 2. softmax of the 8 owned pixels -> tap-pair registers -----------------------------------------------------
        if (NPF > 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA of this wavefront's slots has landed (the softmax needs every channel anyway)
#pragma unroll
            for (int c = 0; c < NPF; ++c) {
                const v4uu v = *(lds_cv4u_ptr)(pf_lds + (c * NTH + tid) * 4);
                graw[c] = make_uint4(v.x, v.y, v.z, v.w);
            }
        }
        unsigned wq[8][12];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned w[NT];
#pragma unroll
            for (int c = 0; c < NT; ++c) w[c] = comp(graw[c], q);
            unsigned mx2 = 0xfc00fc00u;                       // (-inf, -inf): the channel maximum of both pixels at once (exact)
#pragma unroll
            for (int c = 0; c < NT; ++c) mx2 = pk_max_f16(mx2, w[c]);
            constexpr float L2E = 1.44269502162933349609375f;
            softmax_to_pairs<0>(w, -h2f_lo(mx2) * L2E, wq[2 * q]);
            softmax_to_pairs<1>(w, -h2f_hi(mx2) * L2E, wq[2 * q + 1]);
        }
        if (HIST && interior) {
            // publish the taps (before the blend is folded in): pair i = taps (2i, 2i+1), one 16-byte store per pair and quad —
            // dwords (tap 2i: px 0,1 | px 2,3), (tap 2i+1: px 0,1 | px 2,3) — rebuilt from the tap-pair registers by byte permutes
            __half* wkb = static_cast<__half*>(a.wk_out) + (size_t)b * Taps<__half>::image_elems(NT, HW);
            const unsigned hw4 = (unsigned)Taps<__half>::hw4(HW);
            const unsigned qd = off_r >> 2;                  // (the per-round opaque copy: see the guidance loads)
#pragma unroll
            for (int i = 0; i < NT / 2; ++i) {
#pragma unroll
                for (int hq = 0; hq < 2; ++hq) {
                    unsigned dw[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int c = 2 * i + (k >> 1), e = 4 * hq + 2 * (k & 1);
                        const unsigned lo = wq[e][d2_slot(c)], hi = wq[e + 1][d2_slot(c)];
                        dw[k] = d2_half(c) ? __builtin_amdgcn_perm(hi, lo, 0x07060302u) : __builtin_amdgcn_perm(hi, lo, 0x05040100u);
                    }
                    st16_hist(atb(wkb, ((unsigned)i * 2u * hw4 + (qd + (unsigned)hq) * 8u) * 2u), make_uint4(dw[0], dw[1], dw[2], dw[3]));
                }
            }
        }
        // sparse blend (CSPN_ours.py:51-53): (1-m) u + m x0, m = sign(sparse); 1-m in {0, 1, 2} is folded into the taps (exact in
        // fp16), the steps start their sums from md = m x0
        float md[BLEND ? 8 : 1];
        if (BLEND) {
            float sp[8], dv[8];
            IO::to_f8(spo, sp);
            IO::to_f8(x0o, dv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float m = in_img ? sgnf(sp[e]) : 0.f;
                md[e] = m * (in_img ? dv[e] : 0.f);
                const unsigned om = pack_h2(1.f - m, 1.f - m);
#pragma unroll
                for (int s = 0; s < 12; ++s) wq[e][s] = pk_mul_f16(wq[e][s], om);
            }
        }
        stamp();                                              // weights derived
        if (NPF > 0 && round + 1 < a.rounds) {
            // the next round's first NPF channels stream into LDS while this round's steps run (the slots were read above; a workgroup's
            // next image is b + nb, same tile, same offsets).  The phase boundaries' s_waitcnt vmcnt(0) also wait for them.
            const int bn = b + a.nb;
            if (bn < a.B) prefetch(bn, off_r);
        }

        // ---- 3. phases of S steps; between phases the tile borders travel through the exchange planes --------------------
        __half* __restrict__ outb = HIST ? nullptr : kuniform_ptr(static_cast<__half*>(a.out) + (size_t)b * HW);
        __half* hist_step = HIST ? kuniform_ptr(static_cast<__half*>(a.hist) + (size_t)b * HW) : nullptr;   // plane of the step being computed
        const int tile_global = b * tiles_per_img + trem;
        uint4 fin = make_uint4(0u, 0u, 0u, 0u);               // the final step's packed result (scored after the loop)
        bool aborted = false;

        for (int p = 0; p < n_phase; ++p) {
            const int steps = (a.T - p * a.S) < a.S ? (a.T - p * a.S) : a.S;
            const bool last_phase = (p == n_phase - 1);
            __half* __restrict__ xout = kuniform_ptr(static_cast<__half*>(a.xbuf) + (size_t)(p & 1) * plane + (size_t)b * HW);
            if (p > 0) {
                // halo octs only (the tile's own interior is in LDS already), device-scope loads, two per thread and trip
                const __half* __restrict__ xin = kuniform_ptr(static_cast<const __half*>(a.xbuf) + (size_t)((p + 1) & 1) * plane + (size_t)b * HW);
                const int tq_in = min(a.tw, rx0 + 8 * wo - x0) >> 3;
                const int th_in = min(a.th, ry0 + wr - y0);
                const int nl = (x0 - rx0) >> 3;
                const int nside = wo - tq_in;
                const int nrow_t = (y0 - ry0) + R;
                const int n_top = nrow_t * wo;
                const int n_bot = (dr - nrow_t - th_in) * wo;
                const int n_halo = n_top + n_bot + th_in * nside;
                Pair rgp[2];
                unsigned rgp_in = 0;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int t = tid + k * NTH;
                    const int row = t >> 1, side = t & 1;
                    const int y = yd0 + row, x = side ? rx0 + 8 * wo : rx0 - 2;
                    const bool in = (t < 2 * dr) && y >= 0 && y < H && x >= 0 && x < W;
                    if (in) rgp_in |= 1u << k;
                    rgp[k] = IO::ld_pair_dev(xin, in ? (unsigned)(y * W + x) : 0u);
                }
                for (int base = 0; base < n_halo; base += 2 * NTH) {
                    Oct hv[2];
                    int at[2];
                    bool hin[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int h = base + u * NTH + tid;
                        int row, oc;
                        if (h < n_top) { row = h / wo; oc = h - row * wo; }
                        else if (h < n_top + n_bot) { const int h2 = h - n_top; row = h2 / wo; oc = h2 - row * wo; row += nrow_t + th_in; }
                        else {
                            const int h3 = h - n_top - n_bot;
                            const int ns = nside > 0 ? nside : 1;
                            row = h3 / ns;
                            const int c = h3 - row * ns;
                            row += nrow_t;
                            oc = c < nl ? c : c + tq_in;
                        }
                        const int y = yd0 + row, x = rx0 + 8 * oc;
                        const bool valid = h < n_halo;
                        hin[u] = valid && y >= 0 && y < H && x < W;
                        hv[u] = IO::ld_oct_dev(xin, hin[u] ? (unsigned)(y * W + x) : 0u);
                        at[u] = valid ? row * ls + 4 * (oc + 1) : -1;
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        if (at[u] >= 0) {
                            const uint4 v = hv[u].a;
                            const bool in = hin[u];
                            *reinterpret_cast<uint4*>(cur + at[u]) = make_uint4(in ? v.x : 0u, in ? v.y : 0u, in ? v.z : 0u, in ? v.w : 0u);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int t = tid + k * NTH;
                    if (t < 2 * dr) {
                        const int row = t >> 1, side = t & 1;
                        const bool in = (rgp_in >> k) & 1u;
                        cur[row * ls + (side ? 4 * (wo + 1) : 3)] = in ? rgp[k].a : 0u;
                    }
                }
            }
            __syncthreads();
            stamp();                                          // depth staged

            // One propagation step on the packed LDS tile: kind 0 = plain, 1 = last step of a phase (the interior is published),
            // 2 = the final step of the forward (refined depth stored, kept for scoring).
            auto step = [&](const int kind, const unsigned* rd, unsigned* wrb) __attribute__((always_inline)) {
                float acc[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(wq[e][10]), "+v"(wq[e][11]));      // see mix_hh
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = BLEND ? md[BLEND ? e : 0] : 0.f;
#pragma unroll
                for (int rr = 0; rr < 2 * R + 1; ++rr) {
                    int drow = sy + rr;                       // depth-region row of window row rr; idle threads past the region read
                    drow = drow < dr ? drow : dr - 1;         // the clamped last row and never store
                    const unsigned* rowp = rd + drow * ls + 4 * (sx + 1);
                    unsigned D[6], U[5];
                    const v4uu d4 = *(lds_cv4u_ptr)(rowp);
#if CSPN_D2_DPP_HALO
                    // the pixel pairs left / right of the oct are the neighbouring lanes' d4.w / d4.x (DPP wave shift: VALU, no LDS); only the
                    // strip-end / wave-edge lanes need LDS, and every other lane reads one dword per region row there (a broadcast) — the two
                    // 4-byte reads per lane at a 16-byte stride are 4-way bank conflicts: 54 % of this kernel's LDS cycles (profiles/r05_sq_pac5.json).
                    // MEASURED (NEGATIVE_RESULTS #56): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.539 -> 0.059, LDS cycles halved — and the forward
                    // 69.8 -> 74.4 us: the step is VALU-bound (v_dot2 / v_fma_mix issue at ~4.6 cycles), the conflicts hide under it, and the 10
                    // DPP moves + 10 selects per step are 13 % more VALU work (profiles/r06_d2_dpp_halo_ab.txt, r06_sq_pac5_dpp_halo.json)
                    const unsigned* rowb = rd + drow * ls;
                    const unsigned lfix = *(lds_cu_ptr)(rowb + colL), rfix = *(lds_cu_ptr)(rowb + colR);
                    const unsigned ldpp = (unsigned)__builtin_amdgcn_update_dpp(0, (int)d4.w, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
                    const unsigned rdpp = (unsigned)__builtin_amdgcn_update_dpp(0, (int)d4.x, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
                    D[0] = fix_left ? lfix : ldpp;
                    D[5] = fix_right ? rfix : rdpp;
#else
                    D[0] = *(lds_cu_ptr)(rowp - 1);
                    D[5] = *(lds_cu_ptr)(rowp + 4);
#endif
                    D[1] = d4.x; D[2] = d4.y; D[3] = d4.z; D[4] = d4.w;
#pragma unroll
                    for (int k = 0; k < 5; ++k) U[k] = __builtin_amdgcn_alignbit(D[k + 1], D[k], 16);      // pixels (2k - 1, 2k)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float& ae = acc[2 * q];
                        float& ao = acc[2 * q + 1];
                        const unsigned (&we)[12] = wq[2 * q];
                        const unsigned (&wo_)[12] = wq[2 * q + 1];
                        if (rr == 2) {
                            ae = dot2(we[4], D[q], ae);       ae = dot2(we[5], U[q + 1], ae);
                            ao = dot2(wo_[4], U[q], ao);      ao = dot2(wo_[5], D[q + 2], ao);
                        } else {
                            const int s0 = rr < 2 ? 2 * rr : 6 + 2 * (rr - 3);
                            ae = dot2(we[s0], D[q], ae);      ae = dot2(we[s0 + 1], D[q + 1], ae);
                            ao = dot2(wo_[s0], U[q], ao);     ao = dot2(wo_[s0 + 1], U[q + 1], ao);
                            if (rr == 0) { ae = mix_hh<0, 0>(we[10], D[q + 2], ae); ao = mix_hh<0, 1>(wo_[10], D[q + 2], ao); }
                            if (rr == 1) { ae = mix_hh<1, 0>(we[10], D[q + 2], ae); ao = mix_hh<1, 1>(wo_[10], D[q + 2], ao); }
                            if (rr == 3) { ae = mix_hh<0, 0>(we[11], D[q + 2], ae); ao = mix_hh<0, 1>(wo_[11], D[q + 2], ao); }
                            if (rr == 4) { ae = mix_hh<1, 0>(we[11], D[q + 2], ae); ao = mix_hh<1, 1>(wo_[11], D[q + 2], ao); }
                        }
                    }
                }
                uint4 o = make_uint4(cvt_pk_f16(acc[0], acc[1]), cvt_pk_f16(acc[2], acc[3]), cvt_pk_f16(acc[4], acc[5]), cvt_pk_f16(acc[6], acc[7]));
                if (!CLEAN) { if (!in_img) o = make_uint4(0u, 0u, 0u, 0u); }        // zero padding stays exactly zero
                if (kind == 1) { if (interior) st16_dev(xout, off_own * 2u, o); }
                else if (kind == 2 && !HIST) { if (interior) st16(atb(outb, off_own * 2u), o); fin = o; }
                if (HIST && interior) st16_hist(atb(hist_step, off_own * 2u), o);
                if (kind != 2 && active) *reinterpret_cast<uint4*>(wrb + (sy + R) * ls + 4 * (sx + 1)) = o;
            };
            const bool any = __ballot(active) != 0ull;       // wavefronts without a single owned row only keep the barriers company
#ifdef CSPN_D2_HALO_ROWS_FIRST
            const bool any_tile = __ballot(active && sy >= i0 && sy < i0 + th_in) != 0ull;      // ... without a tile row: the last step
#else
            const bool any_tile = any;
#endif
            for (int s = 0; s < steps; ++s) {
                const int kind = (s == steps - 1) ? (last_phase ? 2 : 1) : 0;
                if (kind == 0 ? any : any_tile) step(kind, ldsu + (s & 1) * pp, ldsu + ((s + 1) & 1) * pp);
                if (HIST) hist_step += plane;
                if (kind == 0) __syncthreads();
            }
            stamp();                                          // steps of the phase done
            if (!last_phase) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this thread's device-scope stores have landed
                __syncthreads();                                       // ... and so have everybody else's in the workgroup
                const unsigned want = a.seq + (unsigned)p + 1u;
                if (tid == 0) __hip_atomic_store(a.flags + tile_global, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (tid < 9 && tid != 4) {
                    const int ny = ty + tid / 3 - 1, nx = tx + tid % 3 - 1;
                    if (ny >= 0 && ny < a.tiles_y && nx >= 0 && nx < a.tiles_x) {
                        const unsigned* f = a.flags + b * tiles_per_img + ny * a.tiles_x + nx;
                        unsigned spins = 0;
                        while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
                            ++spins;
                            if ((spins & 255u) == 0u && __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.seq) {
                                wg_bad = 1;
                                break;
                            }
                            if (spins >= a.spin_limit) {                // a neighbour never became resident / finished: give up
                                __hip_atomic_store(a.status, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                wg_bad = 1;
                                break;
                            }
                            __builtin_amdgcn_s_sleep(2);
                        }
                    }
                }
                __syncthreads();
                stamp();                                      // neighbours' borders published
                if (wg_bad) { aborted = true; break; }
            }
        }
        if (aborted) {
            if (tid == 0) {
                __hip_atomic_store(a.status + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.host_err) __hip_atomic_store(a.host_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __atomic_thread_fence(__ATOMIC_SEQ_CST);
            }
            // poison what this workgroup will never produce — this round's tile and the tiles of its remaining rounds — with NaN
            constexpr unsigned pz2 = CSPN_POISON_F16 | (CSPN_POISON_F16 << 16);      // NaN with the payload the host looks for
            const uint4 qn = make_uint4(pz2, pz2, pz2, pz2);
            __half* const pz = HIST ? static_cast<__half*>(a.hist) + (size_t)(a.T - 1) * plane : static_cast<__half*>(a.out);   // x_T
            for (int r2 = round; r2 < a.rounds; ++r2) {
                const int b2 = a.b0 + r2 * a.nb + bl;
                if (b2 < a.B && interior) st16(atb(pz + (size_t)b2 * HW, off_own * 2u), qn);
            }
            count_out();
            return;
        }
        if (SCORE) {
            float mf[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) mf[k] = 0.f;
            if (interior) {
                const __half* tgt_b = kuniform_ptr(static_cast<const __half*>(a.target) + (size_t)b * HW);
                float t8[8], f8[8];
                IO::to_f8(IO::ld_oct(tgt_b, off_own), t8);
                IO::to_f8(Oct{fin}, f8);
#pragma unroll
                for (int e = 0; e < 8; ++e) metric_terms(f8[e], t8[e], mf);
            }
            float* part = reinterpret_cast<float*>(ldsu + 2 * pp);
            const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                const float v = wave_sum_to_lane63(mf[k]);
                if (lane == 63) part[wave * 10 + k] = v;
            }
            __syncthreads();
            if (tid < 10) {
                double v = 0.0;
                for (int w = 0; w < NTH / 64; ++w) v += (double)part[w * 10 + tid];
                if (v != 0.0) atomicAdd(a.macc + (size_t)(blockIdx.x % a.nslots) * 10 + tid, v);
            }
        }
        stamp();                                              // epilogue done
    }
    count_out();
}

template <int BLEND, int MODE, int CLEAN, int NTH, int NPF>
int d2_launch_inst(const KResArgs& a, int grid, size_t lds_bytes, hipStream_t st) {
    constexpr auto kern = cspnk_d2<BLEND, MODE, CLEAN, NTH, NPF>;
    static std::atomic<size_t> granted[64];
    int dev = 0;
    HIP_OK(hipGetDevice(&dev));
    if (lds_bytes > 64 * 1024 && granted[dev & 63].load(std::memory_order_acquire) < lds_bytes) {
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        granted[dev & 63].store(lds_bytes, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTH), lds_bytes, st, a);
    HIP_OK(hipGetLastError());
    return 1;
}
template <int NTH, int NPF>
int d2_launch_nth(const KResArgs& a, int grid, size_t lds, int blend, int mode, int clean, hipStream_t st) {
#define D2_CASE(BL, MD, CL) \
    if (blend == BL && mode == MD && clean == CL) return d2_launch_inst<BL, MD, CL, NTH, NPF>(a, grid, lds, st)
    D2_CASE(0, 0, 0); D2_CASE(0, 0, 1); D2_CASE(0, 1, 0); D2_CASE(0, 1, 1); D2_CASE(0, 2, 0); D2_CASE(0, 2, 1);
    D2_CASE(1, 0, 0); D2_CASE(1, 0, 1); D2_CASE(1, 1, 0); D2_CASE(1, 1, 1); D2_CASE(1, 2, 0); D2_CASE(1, 2, 1);
#undef D2_CASE
    return fail("cspnk_forward_resident (dot2 form): internal dispatch");
}

}  // namespace

namespace cspn_detail {

// Row stride (dwords) of a packed depth buffer: [3 pad][ring pair][wo octs][ring pair] rounded up to whole quads, and such that the
// 16-byte slot a lane reads continues the bank pattern of the row above it inside a wavefront: ls = 4 wo (mod 64 dwords).
int kres_d2_row_stride(int wo) {
    const int lo = 4 * (wo + 2);
    int best = lo, best_score = 1 << 30;
    for (int cand = lo; cand < lo + 64; cand += 4) {
        const int score = (((cand - 4 * wo) % 64) + 64) % 64;
        if (score < best_score) { best_score = score; best = cand; }
    }
    return best;
}

size_t kres_d2_lds_bytes(int dr, int ls, int threads, int npf) {
    const size_t base = (((size_t)2 * dr * ls + (size_t)(threads / 64) * 10 + 16 + 3) & ~(size_t)3) * sizeof(unsigned);
    return base + (size_t)npf * threads * 16;
}

// Guidance channels that travel through LDS (cspnk_d2's NPF): D2_NPF when the launch has more than one round — only then is there a next
// round to request early — and the prefetch area fits the 160 KB next to the depth buffers; else 0 (-DCSPN_D2_PREFETCH=0: never, for A/B).
int kres_d2_prefetch_channels(int dr, int ls, int threads, int rounds) {
    if (!CSPN_D2_PREFETCH || rounds < 2) return 0;
    return kres_d2_lds_bytes(dr, ls, threads, D2_NPF) <= (size_t)160 * 1024 ? D2_NPF : 0;
}

int kres_d2_launch(const void* kres_args, int threads, int grid, size_t lds_bytes, int blend, int mode, int clean, int npf, void* stream) {
    const KResArgs& a = *static_cast<const KResArgs*>(kres_args);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (npf != 0 && npf != D2_NPF) return fail("cspnk_forward_resident (dot2 form): %d prefetched channels (0 or %d)", npf, D2_NPF);
#if CSPN_D2_PREFETCH       // (the prefetching instances are only compiled into A/B builds)
    if (npf && threads == 768) return d2_launch_nth<768, D2_NPF>(a, grid, lds_bytes, blend, mode, clean, st);
    if (npf && threads == 512) return d2_launch_nth<512, D2_NPF>(a, grid, lds_bytes, blend, mode, clean, st);
#endif
    if (threads == 768) return d2_launch_nth<768, 0>(a, grid, lds_bytes, blend, mode, clean, st);
    if (threads == 512) return d2_launch_nth<512, 0>(a, grid, lds_bytes, blend, mode, clean, st);
    return fail("cspnk_forward_resident (dot2 form): %d threads (512 or 768)", threads);
}

}  // namespace cspn_detail
